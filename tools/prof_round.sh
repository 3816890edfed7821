#!/bin/bash
# Per-round evidence for profiles/ (GPU box; R=r03 bash tools/prof_round.sh): (1) rocprofv3 kernel stats of the default bench step, (2) HBM-side traffic per
# kernel (FETCH_SIZE / WRITE_SIZE in separate passes, KB; FETCH_SIZE doubled for 16-B/lane streaming reads per the gfx950 note of
# MI355X_MICROARCH.md), (3) MFMA utilisation / wait / LDS counters of the top kernels (own passes, --kernel-trace only).
# Writes gpurun_out/r02/*; copy the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${R:-r06}; export R
OUT=gpurun_out/$R; mkdir -p $OUT; rm -rf $OUT/stats $OUT/pmc $OUT/rccl
BENCH="python bench.py --graph 0 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"   # eager launches: one traced kernel per launch
SHORT="python bench.py --graph 0 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o b --output-format csv -- $BENCH > $OUT/bench_stdout.log 2>&1
tail -1 $OUT/bench_stdout.log > $OUT/bench_line.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc -o f --output-format csv -- $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc -o w --output-format csv -- $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/pmc -o m1 --output-format csv -- $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc -o m2 --output-format csv -- $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $OUT/pmc -o m3 --output-format csv -- $SHORT > /dev/null 2>&1
XCH="python bench.py --force-exchange --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs"   # the multi-rank code path: 4 plain weight-gradient chunks
rm -rf $OUT/pmcx
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmcx -o f --output-format csv -- $XCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmcx -o w --output-format csv -- $XCH > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/rccl -o r --output-format csv -- python bench.py --force-exchange --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs > $OUT/rccl_stdout.log 2>&1
python - <<'PY'
import csv, collections, json, glob, os
R = os.environ.get("R", "r03")
OUT = "gpurun_out/" + R
def short(n): return n.split("(")[0].replace("void ", "")
def fam(n):   # PMC tables: the instantiations of the P16 kernels as one family (bench.py looks the dominant kernel up by this name)
    s = short(n)
    if s.startswith("vptr_gemm_p16_kernel"): return "vptr_gemm_p16_kernel"   # (the two wgrad instantiations stay separate rows: slab launch / sub-problem launches)
    if s.startswith("vptr_wgrad_p16_sync_kernel"): return s.replace("<16, 8, 2, 2>", "<16>")   # the 128-row persistent launch keeps its round-4 name
    return s
# ---- (1) kernel stats
rows = list(csv.DictReader(open(OUT + "/stats/b_kernel_stats.csv")))
steps = 28.0   # 5 warm-up + 20 timed + 3 instrumented
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
gemm = sum(float(r["TotalDurationNs"]) for r in rows if "gemm" in r["Name"] or "conv_planes" in r["Name"] or "wgrad" in r["Name"]) / 1e6 / steps
with open(OUT + "/%s_bench_kernel_stats.md" % R, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --graph 0 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs   (eager, N=16, bf16x3, dropout 0.1;\n")
    f.write("# 5 warm-up + 20 timed + 3 instrumented steps = 28 steps; per-step figures = totals / 28)\n")
    f.write("# kernel time %.1f ms/step: MFMA GEMM kernels %.1f, everything else %.1f\n\n" % (tot, gemm, tot - gemm))
    f.write("| kernel | calls/step | ms/step | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:60]:
        f.write("| %s | %.1f | %.3f | %.1f | %s |\n" % (short(r["Name"])[:70], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps,
                                                    float(r["AverageNs"]) / 1e3, r["Percentage"]))
    aten = [r for r in rows if "at::native" in r["Name"] or "at::cuda" in r["Name"] or "Cijk_" in r["Name"] or "copyBuffer" in r["Name"] or "fillBuffer" in r["Name"]]
    f.write("\nstock ATen / runtime kernels (at::native, Tensile, copyBuffer, fillBuffer): %.1f launches/step, %.3f ms/step\n" % (
        sum(int(r["Calls"]) for r in aten) / steps, sum(float(r["TotalDurationNs"]) for r in aten) / 1e6 / steps))
    f.write("all kernels: %.1f launches/step\n" % (sum(int(r["Calls"]) for r in rows) / steps))
print(open(OUT + "/%s_bench_kernel_stats.md" % R).read()[:6000])
# ---- (1b) forced one-rank RCCL exchange
try:
    rr = list(csv.DictReader(open(OUT + "/rccl/r_kernel_stats.csv")))
    with open(OUT + "/%s_rccl_world1_kernel_stats.md" % R, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --force-exchange --steps 5 --warmup 2 ...  (ONE-rank RCCL group, the step through the\n")
        f.write("# multi-rank code path: forward + backward as one hipGraph (capture_front), then 4 chunked weight-gradient launches, async all-reduces of the\n# 473.5 MB slab in <= 64 MB pieces, optimizer; the trace covers capture warm-up, verify_graph and the 7 bench steps)\n")
        f.write("# RCCL / c10d kernels in the trace: %s\n\n" % ([short(r["Name"])[:60] for r in rr if ("nccl" in r["Name"].lower() or "rccl" in r["Name"].lower()) and "rocclr" not in r["Name"].lower()] or "none (in-place all-reduce on one rank launches no kernel)"))
        f.write("| kernel | calls | total ms | avg us |\n|---|---|---|---|\n")
        for r in rr[:25]:
            f.write("| %s | %s | %.3f | %.1f |\n" % (short(r["Name"])[:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
    print(open(OUT + "/%s_rccl_world1_kernel_stats.md" % R).read()[:1500])
except Exception as e:
    print("rccl stats failed:", e)
# ---- (2) traffic
res = {}
for tag, cname in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(OUT + "/pmc/%s_counter_collection.csv" % tag)):
        if r["Counter_Name"] != cname: continue
        k = fam(r["Kernel_Name"]); agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        res.setdefault(k, {})[cname] = v / n; res[k]["launches"] = n
out = {k: v for k, v in sorted(res.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) * kv[1].get("launches", 0)))
       if "gemm" in k or "wgrad" in k or "conv_planes" in k or v.get("FETCH_SIZE", 0) * v.get("launches", 0) > 1e5}
import hashlib
src_hash = hashlib.sha256(b"".join(open(os.path.join("vptr_amd", "csrc", f), "rb").read() for f in ("gemm_p16.hip", "gemm_shared.h", "gemm.hip"))).hexdigest()[:16]
json.dump({"gemm_source_sha16": src_hash, "unit": "KB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes); HBM-side bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024",
           "command": "python bench.py --graph 0 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs", "kernels": out},
          open(OUT + "/%s_pmc_traffic.json" % R, "w"), indent=1)
for k, v in list(out.items())[:12]:
    print("%-46s launches %5d  FETCH_SIZE %12.1f KB  WRITE_SIZE %12.1f KB per launch" % (k[:46], v.get("launches", 0), v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)))
# ---- (2b) the same counters on the exchange path: fabric bytes of the PLAIN chunked weight-gradient launches that an
# all-reduce shares the fabric with
try:
    resx = {}
    for tag, cname in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(OUT + "/pmcx/%s_counter_collection.csv" % tag)):
            if r["Counter_Name"] != cname or "wgrad" not in r["Kernel_Name"]: continue
            k = short(r["Kernel_Name"]); agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
        for k, (n, v) in agg.items():
            resx.setdefault(k, {})[cname] = v / n; resx[k]["launches"] = n
    for k, v in resx.items():
        v["hbm_side_MB_per_launch"] = round((2.0 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 / 1e6, 1)
    json.dump({"unit": "KB per launch; HBM-side bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024", "command": "python bench.py --force-exchange --steps 2 --warmup 1 ...",
               "kernels": resx}, open(OUT + "/%s_pmc_traffic_exchange.json" % R, "w"), indent=1)
    print(json.dumps(resx, indent=1)[:1500])
except Exception as e:
    print("exchange traffic failed:", e)
# ---- (3) MFMA utilisation
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in sorted(glob.glob(OUT + "/pmc/m*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = fam(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
tops = sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CYCLES", 0))[:8]
with open(OUT + "/%s_mfma_util.md" % R, "w") as f:
    f.write("# SQ counters per launch (rocprofv3 --pmc, own passes with --kernel-trace only), top kernels of the bench step by SQ_BUSY_CYCLES\n")
    f.write("# MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES * 4 SIMDs per CU-equivalent unit as reported); wave-level shares are of SQ_WAVE_CYCLES\n\n")
    f.write("| kernel | launches | MFMA_MOPS_BF16 | MFMA_BUSY_CYCLES | BUSY_CYCLES | WAVE_CYCLES | mfma_busy/busy | wait_inst/wave | active/wave | LDS_BANK_CONFLICT/LDS_IDX_ACTIVE | INSTS_MFMA | INSTS_VALU | INSTS_LDS | INSTS_VMEM |\n|" + "---|" * 14 + "\n")
    for k in tops:
        g = lambda c: agg[k].get(c, 0.0) / max(cnt[(k, c)], 1)
        f.write("| %s | %d | %.3g | %.3g | %.3g | %.3g | %.3f | %.3f | %.3f | %.4f | %.3g | %.3g | %.3g | %.3g |\n" % (
            k[:48], cnt[(k, "SQ_BUSY_CYCLES")], g("SQ_INSTS_VALU_MFMA_MOPS_BF16"), g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_BUSY_CYCLES"), g("SQ_WAVE_CYCLES"),
            g("SQ_VALU_MFMA_BUSY_CYCLES") / max(g("SQ_BUSY_CYCLES"), 1), g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
            g("SQ_ACTIVE_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1),
            g("SQ_INSTS_MFMA"), g("SQ_INSTS_VALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM")))
print(open(OUT + "/%s_mfma_util.md" % R).read())
PY
