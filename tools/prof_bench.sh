#!/bin/bash
# rocprofv3 kernel stats of the default bench step (GPU box) -> gpurun_out/prof_bench/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o b --output-format csv -- python bench.py --no-cpu-baseline > gpurun_out/prof_bench_stdout.log 2>&1
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_bench/b_kernel_stats.csv")))
steps = 26.0  # 5 warm-up + 20 timed + 1 instrumented
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
g = sum(float(r["TotalDurationNs"]) for r in rows if "vptr_gemm" in r["Name"]) / 1e6 / steps
print("kernel time %.1f ms/step, gemm %.1f, other %.1f" % (tot, g, tot - g))
for r in rows[:45]:
    print("%-64s %7.1f/step %8.3f ms/step %9.1f us" % (r["Name"].split("(")[0].replace("void ", "")[:64], int(r["Calls"]) / steps,
                                                      float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3))
PY
