#!/bin/bash
cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
VPTR_GEMM_ROWS=256 timeout 1500 python -m pytest tests/test_00_ops_gpu.py tests/test_01_p16_gpu.py tests/test_02_model_gpu.py -x -q 2>&1 | tail -6
L=gpurun_out/$R/nt_rows_ab.log; : > $L
for m in 128 model 256; do echo "### VPTR_GEMM_ROWS=$m" >> $L; VPTR_GEMM_ROWS=$m timeout 300 python tools/gemm_shapes.py 2>&1 | grep -E "^10240 |total" >> $L; done
cat $L
for m in 128 model 256; do VPTR_GEMM_ROWS=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('GEMM_ROWS=$m', d['ms_per_step'], r['all_gemm'])"; done
