#!/bin/bash
# round 6, lease 2: register-staged operand path of the grouped weight-gradient (tn) launches, VPTR_WGRAD_RS
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_wgrad_rs_ab.log && : > $O
export PYTHONPATH=.
for m in 1 2 3; do
  echo "### correctness VPTR_WGRAD_RS=$m" >> $O
  VPTR_WGRAD_RS=$m timeout 900 python -m pytest tests/test_01_p16_gpu.py tests/test_00_ops_gpu.py -x -q -m gpu -k "wgrad or linear or mlp or grouped" 2>&1 | tail -3 >> $O
done
for r in 1 2; do
for v in "VPTR_WGRAD_RS=0" "VPTR_WGRAD_RS=1" "VPTR_WGRAD_RS=2" "VPTR_WGRAD_RS=3" "VPTR_WGRAD_RS=0 VPTR_WGRAD_WAVES=4"; do
  echo "### wgrad_standalone rows=256 $v" >> $O
  env VPTR_WGRAD_ROWS=256 $v timeout 300 python tools/wgrad_standalone.py --reps 20 2>&1 | tail -2 >> $O
done; done
for v in "VPTR_WGRAD_RS=0" "VPTR_WGRAD_RS=2"; do
  echo "### wgrad_standalone rows=128 $v" >> $O
  env VPTR_WGRAD_ROWS=128 $v timeout 300 python tools/wgrad_standalone.py --reps 20 2>&1 | tail -2 >> $O
done
echo "### nt standalone FORCE_LONE DMA" >> $O; VPTR_GEMM_FORCE_LONE=1 timeout 300 python tools/gemm_standalone.py 2>&1 | tail -3 >> $O
echo "### nt standalone FORCE_LONE RS=1" >> $O; VPTR_GEMM_FORCE_LONE=1 VPTR_GEMM_RS=1 timeout 300 python tools/gemm_standalone.py 2>&1 | tail -3 >> $O
for i in 1 2; do for v in "VPTR_WGRAD_RS=0" "VPTR_WGRAD_RS=1" "VPTR_WGRAD_RS=3" "VPTR_WGRAD_RS=1 VPTR_GEMM_RS=1"; do
  echo "$v $(env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> $O
done; done
tail -70 $O
