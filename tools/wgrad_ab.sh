#!/bin/bash
L=gpurun_out/wgrad_ab.log; : > $L
run() { echo "### $ENVV $*" >> $L; env $ENVV timeout 300 python tools/wgrad_standalone.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 >> $L; }
ENVV="VPTR_WGRAD_GEN=0" run --order shape --reps 20
ENVV="VPTR_WGRAD_GEN=512" run --order shape --reps 20
ENVV="VPTR_WGRAD_GEN=1024" run --order shape --reps 20
ENVV="VPTR_WGRAD_GEN=408" run --order shape --reps 20
ENVV="VPTR_WGRAD_GEN=1536" run --order shape --reps 20
cat $L
