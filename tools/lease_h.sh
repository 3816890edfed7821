#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
timeout 600 python bench.py --ddp-probe 2>/dev/null | grep "^{" | cut -c1-120
rm -rf gpurun_out/$R/prof_ddp; rocprofv3 --kernel-trace --stats -d gpurun_out/$R/prof_ddp -o b --output-format csv -- python bench.py --ddp-probe > /dev/null 2>&1
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r05/prof_ddp/b_kernel_stats.csv")))
steps = 5.0   # 2 warm-up + 3 timed
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
print("DDP probe: kernel time %.1f ms/step, %d launches/step" % (tot, sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:22]:
    print("%-70s %7.1f/step %8.3f ms/step %9.1f us" % (r["Name"].split("(")[0].replace("void ", "")[:70], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3))
PY
