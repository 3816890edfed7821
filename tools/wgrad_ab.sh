#!/bin/bash
# A/B of the grouped weight-gradient launch variants on the step's real problem set (one process per variant)
L=gpurun_out/wgrad_ab.log; : > $L
run() { echo "### $ENVV $*" >> $L; env $ENVV timeout 300 python tools/wgrad_standalone.py "$@" 2>&1 | grep -v amdgpu.ids | tail -2 >> $L; }
ENVV="VPTR_WGRAD_FLIP=0 VPTR_WGRAD_XCD=0" run --order recorded
ENVV="VPTR_WGRAD_FLIP=1 VPTR_WGRAD_XCD=0" run --order recorded
ENVV="VPTR_WGRAD_FLIP=1 VPTR_WGRAD_XCD=1" run --order recorded
ENVV="VPTR_WGRAD_FLIP=1 VPTR_WGRAD_XCD=0" run --order address
ENVV="VPTR_WGRAD_FLIP=1 VPTR_WGRAD_XCD=0" run --order shape
ENVV="VPTR_WGRAD_FLIP=1 VPTR_WGRAD_XCD=1" run --order shape
cat $L
