# A/B of the GEMM epilogue variants on one box (VPTR_GEMM_EPI_ROWS bit 0: pipelined kernels, bit 1: single-image kernel)
for i in 1 2 3; do
for m in 3 1 0; do echo "EPI_ROWS=$m $(VPTR_GEMM_EPI_ROWS=$m timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')"; done
done
