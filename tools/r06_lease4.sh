#!/bin/bash
# round 6, lease 4: per-kernel tables of the eager step with the fused norm1 + GELU + depthwise forward on / off; op tests of the fused op
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease4.log && : > $O
export PYTHONPATH=.
timeout 600 python -m pytest tests/test_01_p16_gpu.py -x -q -m gpu -k "norm_dwconv or frame_stats" 2>&1 | tail -3 >> $O
for v in 1 0; do
  echo "### VPTR_FUSED_NORM_DW=$v" >> $O
  VPTR_FUSED_NORM_DW=$v bash tools/kstats.sh 40 >> $O 2>&1
done
tail -100 $O
