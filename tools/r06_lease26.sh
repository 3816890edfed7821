#!/bin/bash
# round 6, lease 26: test_24 x 4 and the full -x suite once with the final test harness
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease26.log && : > $O
export PYTHONPATH=.
for i in 1 2 3; do
  timeout 2400 python -m pytest tests/test_24_bench_launch_gpu.py -q -m gpu -s 2>&1 | grep -i "attempt\|passed\|failed\|Thread\|File" | cut -c1-200 | tail -40 >> $O
done
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 >> $O
cat $O
