#!/bin/bash
# round 6, lease 17: LayerNorm(528) backward rows per workgroup (8 / 16 / 32), same-box A/B
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease17.log && : > $O
export PYTHONPATH=.
VPTR_LN_BWD_RPB=8 timeout 600 python -m pytest tests/test_00_ops_gpu.py -q -m gpu -k "layernorm" 2>&1 | tail -2 >> $O
B="--steps 20 --warmup 4 --no-cpu-baseline --no-roofline --no-other-configs"
for i in 1 2 3; do for v in 16 8 32; do
  echo "k64 VPTR_LN_BWD_RPB=$v $(VPTR_LN_BWD_RPB=$v timeout 300 python bench.py $B 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> $O
done; done
cat $O
