#!/bin/bash
cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
timeout 2400 python -m pytest tests/test_00_ops_gpu.py tests/test_02_model_gpu.py tests/test_04_dropin_gpu.py tests/test_05_config_steps_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/$R/tests_b.log; cat gpurun_out/$R/tests_b.log
bash tools/prof_cfg.sh 4 | head -40
bash tools/prof_cfg.sh 5 | head -40
