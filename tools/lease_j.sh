#!/bin/bash
cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
L=gpurun_out/$R/wgrad_rows_ab.log; : > $L
for w in 128 256 192 128 256 192; do echo "### VPTR_WGRAD_ROWS=$w" >> $L; VPTR_WGRAD_ROWS=$w timeout 300 python tools/wgrad_standalone.py --reps 20 2>&1 | grep -v amdgpu.ids | tail -2 >> $L; done
cat $L
VPTR_WGRAD_ROWS=192 timeout 1200 python -m pytest tests/test_01_p16_gpu.py tests/test_02_model_gpu.py -x -q -k "p16 or wgrad or grouped or train_step or full_size_digest" 2>&1 | tail -8
for w in 128 256 192; do VPTR_WGRAD_ROWS=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ROWS=$w', d['ms_per_step'], r['all_gemm'], {k:v for k,v in r['per_kernel'].items() if 'wgrad' in k})"; done
