"""The product nt P16 kernel alone (through the C ABI), cache-cold: successive launches rotate through ROT operand / output sets.
Compare with tools/gemm_p16_probe (ROT=6) and with the per-shape figures inside the step (tools/gemm_shapes.py)."""
import ctypes, os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import vptr_amd.ops as ops
from vptr_amd._lib import GemmDesc, lib, ptr, stream

dev = torch.device("cuda:0")
ROT = int(os.environ.get("ROT", "6"))
for (M, N, K, flags) in [(10240, 528, 528, ""), (10240, 528, 528, "b"), (10240, 528, 528, "br"), (10240, 528, 528, "o"), (10240, 528, 2112, "b"),
                         (10240, 2112, 528, "b"), (10240, 2112, 528, "bo")]:
    descs = []
    keep = []
    for r in range(ROT):
        A = ops.to_p16(torch.randn(M, K, device=dev))
        B = ops.to_p16(torch.randn(N, K, device=dev) * 0.05)
        D = torch.empty(M, N, device=dev)
        bias = torch.randn(N, device=dev) if "b" in flags else None
        res = torch.randn(M, N, device=dev) if "r" in flags else None
        keep += [A, B, D, bias, res]
        d = GemmDesc()
        d.A, d.B, d.D = ptr(A), ptr(B), ptr(D)
        d.lda, d.ldb, d.ldd = K, K, N
        d.M, d.N, d.K = M, N, K
        d.a_mode, d.b_mode = ops.A_P16, ops.B_P16
        d.precision = 3
        d.split_k = 1
        d.alpha = 1.0
        d.bias = ptr(bias)
        d.residual = ptr(res)
        d.ldr = N if res is not None else 0
        d.d_p16 = int("o" in flags)
        d.rs_div = d.rs_mod = 1
        descs.append(d)
    st = stream()
    for d in descs:
        assert lib.vptr_gemm(ctypes.byref(d), st) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        for d in descs:
            lib.vptr_gemm(ctypes.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * ROT)
    print("nt P16 product  M %d N %d K %d flags '%s' rot%d  %7.1f us  %6.1f TFLOP/s" % (M, N, K, flags, ROT, us, 2.0 * M * N * K / us / 1e6))
