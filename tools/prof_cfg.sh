#!/bin/bash
# rocprofv3 kernel stats of BASELINE config 4 or 5 (GPU box): tools/prof_cfg.sh <4|5>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
C=$1; rm -rf gpurun_out/prof_cfg$C
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg$C -o b --output-format csv -- python tools/prof_cfg.py $C 4 > gpurun_out/prof_cfg${C}_stdout.log 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_cfg$C/b_kernel_stats.csv")))
steps = 4.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
print("config $C: kernel time %.1f ms/step" % tot)
for r in rows[:28]:
    print("%-60s %7.1f/step %8.3f ms/step %9.1f us" % (r["Name"].split("(")[0].replace("void ", "")[:60], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3))
PY
