#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_00_ops_gpu.py tests/test_02_model_gpu.py tests/test_03_dropout_parity_gpu.py -x -q -k "attention or digest or dropout or rollouts or transformer" 2>&1 | tail -3
for e in 0 1 0 1; do echo "### VPTR_ATTN_PERSIST=$e"; VPTR_ATTN_PERSIST=$e timeout 300 python tools/attn_bench64.py 2>&1 | grep -v "amdgpu\|lib:"; done
