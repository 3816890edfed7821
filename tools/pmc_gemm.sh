#!/bin/bash
# PMC profile of one GEMM shape (GPU box).  usage: tools/pmc_gemm.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1
export ONLY="${ONLY:-fwd  528x2112}" PRECS=${PRECS:-1,3}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d gpurun_out/pmc_$TAG -o a --output-format csv -- python tools/gemm_bench.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d gpurun_out/pmc_$TAG -o b --output-format csv -- python tools/gemm_bench.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, os
tag = os.environ.get("TAG_", "")
for f in sorted(glob.glob("gpurun_out/pmc_*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "vptr_gemm" not in r["Kernel_Name"]: continue
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(r["Kernel_Name"][:60], r["Counter_Name"])] += 1
    for k, d in agg.items():
        print(f, k)
        for c, v in d.items(): print("    %-28s %14.0f per launch" % (c, v / cnt[(k, c)]))
PY
