#!/bin/bash
# hipGraph replays vs eager steps of the K64 bench step, one variant per process (the bisect that found the round-2 corruption:
# profiles/r03_graph_postmortem.md).  Output -> gpurun_out/graph_bisect.log
mkdir -p gpurun_out
L=gpurun_out/graph_bisect.log
: > $L
run() { echo "### $*" >> $L; timeout 600 python tools/graph_bisect.py "$@" >> $L 2>&1; echo "rc=$?" >> $L; }
run --mode eager --batch 16 --steps 3
run --mode graph --batch 16 --steps 6 --top 2
run --mode graph --batch 4 --steps 4 --layers 1,1 --top 3
run --mode graph --batch 4 --steps 2 --layers 1,1 --top 2 --nce hooks
cat $L | grep -v amdgpu.ids
