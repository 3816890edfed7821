#!/bin/bash
# round 6, lease 12: chunking of the two-phase LayerNorm((F,H,W)) backward (VPTR_NORM_SPLIT = "big,small" frame chunks), same-box A/B
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease12.log && : > $O
export PYTHONPATH=.
for i in 1 2; do for v in "4,16" "8,16" "2,16" "4,8" "4,32" "8,32" "16,16"; do
  echo "VPTR_NORM_SPLIT=$v $(VPTR_NORM_SPLIT=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> $O
done; done
cat $O
