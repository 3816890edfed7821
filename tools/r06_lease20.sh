#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_lease20.log && : > $O
export PYTHONPATH=.
timeout 1500 python -m pytest tests/test_24_bench_launch_gpu.py tests/test_21_dp_gpu.py -q -m gpu 2>&1 | tail -15 >> $O
VPTR_BENCH_SHARE_GPU=1 VPTR_BENCH_BACKEND=gloo GPU_MAX_HW_QUEUES=2 timeout 600 python bench.py --gpus 2 --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-roofline --dp-chunks 8 --bucket-mb 32 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['comm']; print(c['dp_chunks'], c['bucket_mb'], json.dumps(c['exchange_timeline'])[:1500])" >> $O 2>&1
cat $O
