#!/bin/bash
cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
L=gpurun_out/$R/attn64_ab.log; : > $L
for v in attnv0 attnv1 intree attnv0 attnv1 intree; do
  if [ $v = intree ]; then timeout 300 python tools/attn_bench64.py 2>&1 | grep -v amdgpu >> $L; else VPTR_HIP_LIB=$PWD/vptr_amd/_variants/libvptr_$v.so timeout 300 python tools/attn_bench64.py 2>&1 | grep -v amdgpu >> $L; fi
done
cat $L
