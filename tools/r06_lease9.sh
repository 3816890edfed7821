#!/bin/bash
# round 6, lease 9: frame-statistics buffers padded to one line per frame -- probe, op + model tests, step A/B
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease9.log && : > $O
export PYTHONPATH=.
for v in "VPTR_DWN_LDS=1" "VPTR_DWN_LDS=0" "VPTR_DWN_LDS=1 VPTR_DWN_DBG=2" "VPTR_DWN_LDS=1 VPTR_DWN_DBG=1"; do
  env $v timeout 200 python tools/dwn_probe.py 2>&1 | grep "^env" >> $O
done
echo "### op + model tests" >> $O
timeout 1500 python -m pytest tests/test_01_p16_gpu.py tests/test_02_model_gpu.py tests/test_03_dropout_parity_gpu.py tests/test_05_config_steps_gpu.py tests/test_20_graph_gpu.py -q -m gpu 2>&1 | tail -8 >> $O
for i in 1 2 3; do for v in "VPTR_DWN_LDS=1" "VPTR_DWN_LDS=0" "VPTR_FUSED_NORM_DW=0"; do
  echo "$v $(env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> $O
done; done
echo "### kstats" >> $O
bash tools/kstats.sh 40 >> $O 2>&1
tail -80 $O
