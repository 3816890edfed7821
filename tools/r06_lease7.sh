#!/bin/bash
# round 6, lease 7: LDS-slab fused norm + depthwise forward, LDS-fused Winograd output -> input transform; the 8-rank test's first error; ATen sources
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease7.log && : > $O
export PYTHONPATH=.
echo "### op tests" >> $O
timeout 900 python -m pytest tests/test_06_winograd_gpu.py tests/test_01_p16_gpu.py tests/test_00_ops_gpu.py -q -m gpu 2>&1 | tail -12 >> $O
echo "### model parity" >> $O
timeout 1500 python -m pytest tests/test_02_model_gpu.py tests/test_03_dropout_parity_gpu.py tests/test_05_config_steps_gpu.py tests/test_20_graph_gpu.py -q -m gpu 2>&1 | tail -8 >> $O
for i in 1 2 3; do for v in "VPTR_WINO_FUSE=1 VPTR_DWN_LDS=1" "VPTR_WINO_FUSE=0 VPTR_DWN_LDS=1" "VPTR_WINO_FUSE=1 VPTR_DWN_LDS=0" "VPTR_LN_ROWS=1 VPTR_LN_BWD_RPB=32" "VPTR_LN_ROWS=4"; do
  echo "$v $(env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> $O
done; done
echo "### kstats" >> $O
bash tools/kstats.sh 36 >> $O 2>&1
echo "### 8-rank test" >> $O
timeout 1500 python -m pytest tests/test_21_dp_gpu.py -k eight -x -q 2>&1 | grep -v "^  File\|^    " | head -120 >> $O
echo "### aten sources" >> $O
timeout 600 python tools/aten_sources.py 2>&1 | tail -45 >> $O
tail -150 $O
