#!/bin/bash
cd $GRAFT_REPO_ROOT; export R=r05
bash tools/prof_round.sh > gpurun_out/prof_round_r05.log 2>&1; tail -100 gpurun_out/prof_round_r05.log | head -60
bash tools/prof_cfg.sh 4 pmc > gpurun_out/r05/prof_cfg4.log 2>&1
bash tools/prof_cfg.sh 5 pmc > gpurun_out/r05/prof_cfg5.log 2>&1
head -12 gpurun_out/r05/r05_cfg4_kernel_stats.md; head -12 gpurun_out/r05/r05_cfg5_kernel_stats.md
