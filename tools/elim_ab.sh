mkdir -p gpurun_out/r04
L=gpurun_out/r04/nt_elim.log; : > $L
for v in elim4 elim3 elim2; do
echo "### $v" >> $L
VPTR_HIP_LIB=$PWD/vptr_amd/_variants/libvptr_$v.so timeout 300 python tools/gemm_shapes.py 2>&1 | grep -E "^10240 (528|2112) (528|2112|1584|1056) 5 3|Error|error" >> $L
done
echo "### base" >> $L
timeout 300 python tools/gemm_shapes.py 2>&1 | grep -E "^10240 (528|2112) (528|2112|1584|1056) 5 3" >> $L
cat $L
