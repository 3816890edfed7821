#!/bin/bash
# round 6, lease 22: soak of the multi-process tests (the ones that share the GPU between processes), 5 iterations
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease22.log && : > $O
export PYTHONPATH=.
for i in 1 2 3 4 5; do
  echo "### iteration $i" >> $O
  timeout 1500 python -m pytest tests/test_21_dp_gpu.py tests/test_22_rccl_gpu.py tests/test_23_ddp_gpu.py tests/test_24_bench_launch_gpu.py -q -m gpu -s --timeout 1100 2>&1 | grep -i "attempt\|passed\|failed\|xfail\|HSA_STATUS\|Error" | cut -c1-200 >> $O
done
cat $O
