#!/bin/bash
# round 6, lease 6: full GPU suite with the Winograd encoder; 256-row tiles for the Winograd GEMM A/B; the full default bench line
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease6.log && : > $O
export PYTHONPATH=.
echo "### full GPU suite" >> $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -25 >> $O
for i in 1 2; do for v in 128 256; do
  echo "VPTR_WINO_ROWS=$v $(VPTR_WINO_ROWS=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> $O
done; done
echo "### default bench" >> $O
( time timeout 1200 python bench.py ) > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
tail -3 gpurun_out/r06_bench_default.err >> $O
tail -c 6000 gpurun_out/r06_bench_default.json >> $O
tail -60 $O
