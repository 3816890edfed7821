#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/graph_bisect.log
: > $L
run() { echo "### $ENVV $*" >> $L; timeout 600 env $ENVV python tools/graph_bisect.py "$@" >> $L 2>&1; echo "rc=$?" >> $L; }
ENVV="X=1" run --mode graph --batch 16 --steps 6 --top 2
cat $L | grep -v amdgpu.ids
