#!/bin/bash
# idle time between consecutive kernels of the hipGraph step (GPU box): rocprofv3 kernel trace of the default (graph) bench run
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r04/gaps; mkdir -p gpurun_out/r04/gaps
rocprofv3 --kernel-trace -d gpurun_out/r04/gaps -o g --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs > gpurun_out/r04/gaps/stdout.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04/gaps/**/g_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-50:]) for r in csv.DictReader(open(f))]
rows.sort()
# the last 10 steps: find the adamw kernel launches (one per step) as step delimiters
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
print("kernels traced", len(rows), " adamw launches", len(idx))
out = open("gpurun_out/r04/graph_gaps.log", "w")
def P(*a):
    s = " ".join(str(x) for x in a); print(s); out.write(s + "\n")
for a, b in list(zip(idx[:-1], idx[1:]))[-5:]:
    seg = rows[a + 1:b + 1]
    wall = (seg[-1][1] - seg[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in seg) / 1e6
    gaps = [(seg[i + 1][0] - seg[i][1]) / 1e3 for i in range(len(seg) - 1)]
    pos = [g for g in gaps if g > 0]
    P("step: %d kernels  wall %.2f ms  sum of kernel durations %.2f ms  sum of positive gaps %.2f ms  (overlaps %.2f ms)  median gap %.2f us  gaps > 5 us: %d (%.2f ms)" % (
        len(seg), wall, busy, sum(pos) / 1e3, -sum(g for g in gaps if g < 0) / 1e3, sorted(gaps)[len(gaps) // 2], sum(1 for g in gaps if g > 5), sum(g for g in gaps if g > 5) / 1e3))
seg = rows[idx[-2] + 1:idx[-1] + 1]
big = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2], seg[i + 1][2]) for i in range(len(seg) - 1))[-12:]
for g, a, b in big:
    P("  gap %.1f us after %s before %s" % (g, a, b))
PY
