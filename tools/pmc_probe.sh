#!/bin/bash
# L2 hit rate / traffic of one gemm_probe problem (GPU box):  tools/pmc_probe.sh M N K am bm split
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_probe
for c in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM"; do
  t=$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_probe -o $t --output-format csv -- ./gemm_probe.bin "$@" > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc_probe/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %16.0f per launch" % (c, v / cnt[(k, c)]))
PY
