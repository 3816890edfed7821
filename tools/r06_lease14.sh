#!/bin/bash
# round 6, lease 14: kernel tables of BASELINE configs 4 (BAIR FAR 2->28) and 5 (KTH 128x128 NAR 10->40) with the round-6 kernels
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export PYTHONPATH=.
R=r06 bash tools/prof_cfg.sh 4 > gpurun_out/r06_cfg4.log 2>&1
R=r06 bash tools/prof_cfg.sh 5 > gpurun_out/r06_cfg5.log 2>&1
rm -rf gpurun_out/r06/prof_cfg4 gpurun_out/r06/prof_cfg5
tail -5 gpurun_out/r06_cfg4.log gpurun_out/r06_cfg5.log
