#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_00_ops_gpu.py tests/test_01_p16_gpu.py tests/test_02_model_gpu.py tests/test_04_dropin_gpu.py tests/test_21_dp_gpu.py tests/test_23_ddp_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --ddp-probe 2>/dev/null | grep "^{" | cut -c1-60; done
