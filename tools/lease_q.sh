#!/bin/bash
cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
timeout 2400 python -m pytest tests/test_00_ops_gpu.py tests/test_02_model_gpu.py tests/test_03_dropout_parity_gpu.py -x -q -k "attention or digest or dropout or rollouts or transformer" 2>&1 | tail -3
timeout 300 python tools/attn_bench64.py 2>&1 | grep -v amdgpu
bash tools/attn64_pmc.sh 2>&1 | tail -12
