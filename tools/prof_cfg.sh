#!/bin/bash
# rocprofv3 kernel stats (+ HBM-side traffic of the top kernels) of BASELINE config 4 or 5 (GPU box): tools/prof_cfg.sh <4|5> [pmc]
# writes gpurun_out/$R/${R}_cfg<C>_kernel_stats.md (copy into profiles/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
C=$1; R=${R:-r05}; OUT=gpurun_out/$R; mkdir -p $OUT; rm -rf $OUT/prof_cfg$C
STEPS=4
rocprofv3 --kernel-trace --stats -d $OUT/prof_cfg$C -o b --output-format csv -- python tools/prof_cfg.py $C $STEPS > $OUT/prof_cfg${C}_stdout.log 2>&1
if [ "$2" = "pmc" ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_cfg$C -o f --output-format csv -- python tools/prof_cfg.py $C 2 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_cfg$C -o w --output-format csv -- python tools/prof_cfg.py $C 2 > /dev/null 2>&1
fi
C=$C R=$R STEPS=$STEPS python - <<'PY'
import csv, os, collections
C, R, steps = os.environ["C"], os.environ["R"], float(os.environ["STEPS"])
OUT = "gpurun_out/%s" % R
def short(n): return n.split("(")[0].replace("void ", "")
rows = list(csv.DictReader(open(OUT + "/prof_cfg%s/b_kernel_stats.csv" % C)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
gemm = sum(float(r["TotalDurationNs"]) for r in rows if "gemm" in r["Name"] or "conv_planes" in r["Name"] or "wgrad" in r["Name"]) / 1e6 / steps
traffic = {}
for tag, cname in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    fn = OUT + "/prof_cfg%s/%s_counter_collection.csv" % (C, tag)
    if not os.path.exists(fn):
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] != cname: continue
        k = short(r["Kernel_Name"]); agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        traffic.setdefault(k, {})[cname] = v / n
what = {"4": "BAIR FAR 2->28 (FARTrainer, 3-ch zero-pad AE, 12 layers, T_in = 29, per-GPU batch 16)",
        "5": "KTH 128x128 NAR 10->40 (NARTrainer, 16x16 features, 8x8 windows, per-GPU batch 2)"}[C]
with open(OUT + "/%s_cfg%s_kernel_stats.md" % (R, C), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python tools/prof_cfg.py %s %d   (eager, bf16x3, dropout 0.1; %d steps incl. the first; per-step = totals / %d)\n" % (C, steps, steps, steps))
    f.write("# config %s: %s\n# kernel time %.1f ms/step: MFMA GEMM kernels %.1f, everything else %.1f; %.0f launches/step\n" % (
        C, what, tot, gemm, tot - gemm, sum(int(r["Calls"]) for r in rows) / steps))
    if traffic:
        f.write("# HBM-side MB per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB / 1024 (separate --pmc passes; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md)\n")
    f.write("\n| kernel | calls/step | ms/step | avg us | % | HBM-side MB/launch |\n|---|---|---|---|---|---|\n")
    for r in rows[:45]:
        t = traffic.get(short(r["Name"]))
        mb = "%.1f" % ((2 * t.get("FETCH_SIZE", 0) + t.get("WRITE_SIZE", 0)) / 1024.0) if t else ""
        f.write("| %s | %.1f | %.3f | %.1f | %s | %s |\n" % (short(r["Name"])[:70], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps,
                                                        float(r["AverageNs"]) / 1e3, r["Percentage"], mb))
print(open(OUT + "/%s_cfg%s_kernel_stats.md" % (R, C)).read())
PY
