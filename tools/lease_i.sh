#!/bin/bash
cd $GRAFT_REPO_ROOT; export R=r05; mkdir -p gpurun_out/$R
L=gpurun_out/$R/ingest_roofline.log; : > $L
echo "# grouped weight-gradient launch of the K64 step alone (tools/wgrad_standalone.py, 20 reps): full kernel vs the DMA-ONLY elimination build" >> $L
echo "# (-DVPTR_TN_DMA_ONLY: every workgroup stages its operand tiles with the same global_load_lds pieces, waits and barriers, and does nothing else)" >> $L
for rows in 128 256 192; do
echo "### full kernel, VPTR_WGRAD_ROWS=$rows" >> $L; VPTR_WGRAD_ROWS=$rows timeout 300 python tools/wgrad_standalone.py --reps 20 2>&1 | grep -v amdgpu.ids | tail -2 >> $L
echo "### DMA only, VPTR_WGRAD_ROWS=$rows" >> $L; VPTR_HIP_LIB=$PWD/vptr_amd/_variants/libvptr_dmaonly.so VPTR_WGRAD_ROWS=$rows timeout 300 python tools/wgrad_standalone.py --reps 20 2>&1 | grep -v amdgpu.ids | tail -2 >> $L
done
cat $L
