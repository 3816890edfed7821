#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_00_ops_gpu.py tests/test_01_p16_gpu.py tests/test_02_model_gpu.py tests/test_03_dropout_parity_gpu.py -x -q 2>&1 | tail -3
for e in 1 0 1 0; do if [ $e = 1 ]; then export VPTR_GEMM_NO_EPI4=1; else unset VPTR_GEMM_NO_EPI4; fi; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('NO_EPI4=$e', d['ms_per_step'], {k:v for k,v in r['per_kernel'].items() if '<0, ' in k or '<4, ' in k})"; done
