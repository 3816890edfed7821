#!/bin/bash
# round 6, lease 1: register-staged nt operand path (VPTR_GEMM_RS) -- correctness, per-shape A/B, elimination builds, step A/B
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && O=gpurun_out/r06_rs_ab.log && : > $O
run() { echo "### $*" >> $O; "$@" >> $O 2>&1; echo "rc=$?" >> $O; }
export PYTHONPATH=.
for m in "1 2" "2 1" "2 2"; do set -- $m
  echo "### correctness VPTR_GEMM_RS=$1 SETS=$2" >> $O
  VPTR_GEMM_RS=$1 VPTR_GEMM_RS_SETS=$2 timeout 900 python -m pytest tests/test_01_p16_gpu.py tests/test_00_ops_gpu.py -x -q -m gpu -k "gemm or linear or mlp or p16" 2>&1 | tail -4 >> $O
done
echo "### standalone DMA (default)" >> $O; timeout 300 python tools/gemm_standalone.py >> $O 2>&1
echo "### standalone RS=1 sets=2" >> $O; VPTR_GEMM_RS=1 timeout 300 python tools/gemm_standalone.py >> $O 2>&1
echo "### standalone RS=1 sets=1" >> $O; VPTR_GEMM_RS=1 VPTR_GEMM_RS_SETS=1 timeout 300 python tools/gemm_standalone.py >> $O 2>&1
echo "### standalone RS=2 (every grid)" >> $O; VPTR_GEMM_RS=2 timeout 300 python tools/gemm_standalone.py >> $O 2>&1
echo "### standalone RS=2 NOMFMA build (loads + LDS stores + fragment reads + barriers)" >> $O; VPTR_HIP_LIB=vptr_amd/_variants/libvptr_rs_nomfma.so VPTR_GEMM_RS=2 timeout 300 python tools/gemm_standalone.py >> $O 2>&1
echo "### standalone RS=2 NOSTAGE build (MFMA + fragment reads + barriers)" >> $O; VPTR_HIP_LIB=vptr_amd/_variants/libvptr_rs_nostage.so VPTR_GEMM_RS=2 timeout 300 python tools/gemm_standalone.py >> $O 2>&1
for i in 1 2; do for v in 0 1 2; do
  echo "VPTR_GEMM_RS=$v $(VPTR_GEMM_RS=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> $O
done; done
echo "### full GPU suite (default env)" >> $O
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 >> $O
tail -60 $O
