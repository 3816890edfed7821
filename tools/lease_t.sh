#!/bin/bash
cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
for i in 1 2; do VPTR_MARGIN_LOG=$PWD/gpurun_out/$R/margins_$i.log timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2; done
python tools/margins_report.py gpurun_out/$R/margins_1.log gpurun_out/$R/margins_2.log > gpurun_out/$R/r05_margins_report.md 2>&1; head -30 gpurun_out/$R/r05_margins_report.md
