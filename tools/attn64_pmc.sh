#!/bin/bash
# SQ counters of the 17 ... 64-row attention kernels alone (tools/attn_bench64.py), one rocprofv3 --pmc pass per counter set (--kernel-trace only)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/attn64_pmc; rm -rf $O; mkdir -p $O
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c -d $O -o p$i --output-format csv -- python tools/attn_bench64.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in sorted(glob.glob("gpurun_out/attn64_pmc/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "attn_mfma" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    g = lambda c: agg[k].get(c, 0.0) / max(cnt[(k, c)], 1)
    print(k)
    print("  launches/counter %d  waves %.0f  WAVE_CYCLES %.3g  BUSY %.3g" % (cnt[(k, "SQ_WAVE_CYCLES")], g("SQ_WAVES"), g("SQ_WAVE_CYCLES"), g("SQ_BUSY_CYCLES")))
    wc = max(g("SQ_WAVE_CYCLES"), 1)
    print("  of wave cycles: WAIT_ANY (parked: waitcnt / barrier) %.3f  WAIT_INST_ANY (issue stall) %.3f  ACTIVE_INST_ANY %.3f  WAIT_INST_LDS %.3f  ACTIVE_INST_LDS %.3f" % (
        g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc, g("SQ_ACTIVE_INST_LDS") / wc))
    print("  LDS bank conflict / LDS active %.3f" % (g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1)))
    w = max(g("SQ_WAVES"), 1)
    print("  per wave: VALU %.0f  SALU %.0f  LDS %.0f  VMEM %.0f  SMEM %.0f  MFMA %.0f  cycles(quad) %.0f" % (g("SQ_INSTS_VALU") / w, g("SQ_INSTS_SALU") / w, g("SQ_INSTS_LDS") / w,
          g("SQ_INSTS_VMEM") / w, g("SQ_INSTS_SMEM") / w, g("SQ_INSTS_MFMA") / w, g("SQ_WAVE_CYCLES") / w))
PY
