#!/bin/bash
# round 6, lease 15: 16-quad depthwise weight-gradient blocks, 256-thread frame_final, geometry gate of the fused normalise + depthwise launch
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease15.log && : > $O
export PYTHONPATH=.
timeout 1500 python -m pytest tests/test_00_ops_gpu.py tests/test_01_p16_gpu.py tests/test_02_model_gpu.py tests/test_05_config_steps_gpu.py -q -m gpu 2>&1 | tail -4 >> $O
B="--steps 20 --warmup 4 --no-cpu-baseline --no-roofline --no-other-configs"
for i in 1 2; do
for v in "VPTR_DWB_CQ=16" "VPTR_DWB_CQ=32"; do
  echo "k64 $v $(env $v timeout 300 python bench.py $B 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> $O
done
for c in kth128 bair_far; do for v in "VPTR_DWB_CQ=16 VPTR_FUSED_NORM_DW=1" "VPTR_DWB_CQ=32 VPTR_FUSED_NORM_DW=1" "VPTR_DWB_CQ=16 VPTR_FUSED_NORM_DW=2"; do
  echo "$c $v $(env $v timeout 600 python bench.py --config $c $B 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> $O
done; done
done
cat $O
