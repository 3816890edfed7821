#!/bin/bash
# round 6, lease 3: fused norm1 + GELU + depthwise forward (VPTR_FUSED_NORM_DW) -- op tests, model parity, step A/B; counters of the two operand paths
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease3.log && : > $O
export PYTHONPATH=.
echo "### op tests" >> $O
timeout 600 python -m pytest tests/test_01_p16_gpu.py -x -q -m gpu -k "norm_dwconv or frame_stats" 2>&1 | tail -15 >> $O
echo "### model parity (test_02, test_03, test_05, test_20)" >> $O
timeout 1500 python -m pytest tests/test_02_model_gpu.py tests/test_03_dropout_parity_gpu.py tests/test_05_config_steps_gpu.py tests/test_20_graph_gpu.py -x -q -m gpu 2>&1 | tail -15 >> $O
for i in 1 2 3; do for v in 1 0; do
  echo "VPTR_FUSED_NORM_DW=$v $(VPTR_FUSED_NORM_DW=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> $O
done; done
echo "### counters" >> $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA_[A-Z_0-9a-z]+|TCP_[A-Z_0-9a-z]+|TCC_[A-Z_0-9a-z]+)\b" | sort -u | tr '\n' ' ' >> $O; echo >> $O
for v in 0 1; do
  for c in "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCC_BUSY_sum"; do
    t=$(echo $c | tr ' ' '_' | cut -c1-24)
    rm -rf gpurun_out/pmc_rs$v_$t
    VPTR_WGRAD_ROWS=256 VPTR_WGRAD_RS=$v timeout 300 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_rs${v}_$t -o x --output-format csv -- python tools/wgrad_standalone.py --reps 3 > /dev/null 2>&1
  done
  echo "## VPTR_WGRAD_RS=$v (per launch, wgrad kernels)" >> $O
  python - $v >> $O <<'PY'
import csv, glob, collections, sys
v = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc_rs%s_*/**/*counter_collection.csv" % v, recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_p16" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][-46:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, x in sorted(d.items()): print("    %-32s %16.0f" % (c, x / cnt[(k, c)]))
PY
done
tail -120 $O
