mkdir -p gpurun_out/r04
L=gpurun_out/r04/stages_ab.log; : > $L
run() { echo "### $ENVV" >> $L; env $ENVV timeout 300 python tools/wgrad_standalone.py --reps 30 2>&1 | grep -v amdgpu.ids | tail -1 >> $L; }
for i in 1 2; do
ENVV="VPTR_WGRAD_SYNC=0" run
ENVV="VPTR_WGRAD_SYNC=0 VPTR_WGRAD_STAGES=3" run
ENVV="VPTR_WGRAD_SYNC=0 VPTR_WGRAD_STAGES=4" run
ENVV="VPTR_WGRAD_SYNC=16" run
done
echo "### wgrad test stages 4" >> $L
VPTR_WGRAD_SYNC=0 VPTR_WGRAD_STAGES=4 timeout 300 python -m pytest tests/test_01_p16_gpu.py -x -q 2>&1 | tail -1 >> $L
echo "### force lone 4" >> $L
VPTR_GEMM_FORCE_LONE=1 VPTR_GEMM_LONE_STAGES=4 timeout 300 python tools/gemm_shapes.py 2>&1 | grep -E "^10240 (528|2112) (528|2112|1584|1056) 5 3" >> $L
echo "### force lone 3" >> $L
VPTR_GEMM_FORCE_LONE=1 VPTR_GEMM_LONE_STAGES=3 timeout 300 python tools/gemm_shapes.py 2>&1 | grep -E "^10240 (528|2112) (528|2112|1584|1056) 5 3" >> $L
echo "### default" >> $L
timeout 300 python tools/gemm_shapes.py 2>&1 | grep -E "^10240 (528|2112) (528|2112|1584|1056) 5 3" >> $L
cat $L
