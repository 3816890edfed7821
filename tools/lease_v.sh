#!/bin/bash
cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
for n in 4 8; do
VPTR_BENCH_SHARE_GPU=1 VPTR_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus $n --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/$R/bench_share$n.log 2> gpurun_out/$R/bench_share$n.err; echo "rc=$?"
tail -1 gpurun_out/$R/bench_share$n.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('n_gpus', d['n_gpus'], 'ms', d['ms_per_step'], d['config']['launch'], d['config']['global_batch'], d['per_rank_ms_per_step'], 'loss', d['final_loss'], d['loss_sane']); c=d['comm']; print({k:c[k] for k in ('backend','rccl_ranks','rank_sum_check','allreduce_bytes_per_step','allreduce_calls_per_step','exposed_comm_ms_per_step','wait_host_ms_per_step','launched_by')})"
tail -3 gpurun_out/$R/bench_share$n.err | cut -c1-300
done
