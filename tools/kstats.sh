#!/bin/bash
# quick per-kernel table of the eager bench step (GPU box): bash tools/kstats.sh [rows]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/kstats; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o b --output-format csv -- python bench.py --graph 0 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs > $OUT/stdout.log 2>&1
python - "$1" <<'PY'
import csv, sys
rows = list(csv.DictReader(open("gpurun_out/kstats/b_kernel_stats.csv")))
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else 45
steps = 13.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
print("kernel time %.2f ms/step, %d launches/step" % (tot, sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:n]:
    print("%-72s %6.1f calls  %7.3f ms/step  %8.1f us" % (r["Name"].split("(")[0].replace("void ", "")[:72], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3))
PY
