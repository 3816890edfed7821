#!/bin/bash
# L2 / fabric counters of the stand-alone P16 GEMM probe (GPU box): tools/pmc_p16_probe.sh <nt|tn>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
MODE=${1:-tn}
rm -rf gpurun_out/pmc_p16; mkdir -p gpurun_out/pmc_p16
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  t=$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_p16 -o $t --output-format csv -- ./tools/_bin/gemm_p16_probe $MODE > gpurun_out/pmc_p16/stdout_$t.log 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc_p16/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:] + " grid " + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items()):
    if "gemm" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %16.0f per launch" % (c, v / cnt[(k, c)]))
PY
