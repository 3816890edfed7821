#!/bin/bash
# round 6, lease 16: depthwise weight-gradient block shapes (quads per block x frames per block), same-box A/B on K64 and KTH128
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease16.log && : > $O
export PYTHONPATH=.
timeout 600 python -m pytest tests/test_00_ops_gpu.py -q -m gpu -k dwconv 2>&1 | tail -2 >> $O
VPTR_DWB_CQ=8 timeout 600 python -m pytest tests/test_00_ops_gpu.py tests/test_01_p16_gpu.py -q -m gpu -k "dwconv" 2>&1 | tail -2 >> $O
B="--steps 20 --warmup 4 --no-cpu-baseline --no-roofline --no-other-configs"
for i in 1 2; do for v in "VPTR_DWB_CQ=16" "VPTR_DWB_CQ=8" "VPTR_DWB_CQ=16 VPTR_DWB_FPB=4" "VPTR_DWB_CQ=8 VPTR_DWB_FPB=16"; do
  echo "k64 $v $(env $v timeout 300 python bench.py $B 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> $O
  echo "kth128 $v $(env $v timeout 600 python bench.py --config kth128 $B 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> $O
done; done
cat $O
