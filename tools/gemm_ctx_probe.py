"""Why are GEMMs ~30 % slower inside the train step than in a back-to-back micro-benchmark?  (GPU box)
Times the 10240 x 528 x 528 forward GEMM (a) re-using one buffer set, (b) rotating over R buffer sets (cold L2 / MALL / TLB),
(c) for many iterations (sustained clocks), (d) interleaved with an HBM-bound elementwise kernel as in the step."""
import os, sys
import torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
import vptr_amd.ops as ops

dev = torch.device("cuda:0")
M, N, K = 10240, 528, 528


def timed(fn, n):
    for _ in range(5):
        fn(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for R in (1, 4, 16, 64, 256):
    As = [torch.randn(M, K, device=dev) for _ in range(R)]
    Ds = [torch.empty(M, N, device=dev) for _ in range(R)]
    W = torch.randn(N, K, device=dev)
    for n in (20, 400):
        us = timed(lambda i: ops.gemm_raw(As[i % R], W, Ds[i % R], M, N, K, 0, 0), n)
        print("rotate %3d sets, %4d iters: %7.1f us  %6.1f TF/s" % (R, n, us, 2.0 * M * N * K / us / 1e6))
    del As, Ds

# producer -> GEMM -> consumer chain like the step: LN writes A, GEMM reads it
x = torch.randn(M, K, device=dev)
g = torch.ones(K, device=dev)
b = torch.zeros(K, device=dev)
W = torch.randn(N, K, device=dev)
ev = []
def chain(i):
    a = ops.layernorm(x, g, b)
    y = torch.empty(M, N, device=dev)
    if len(ev) < 400:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm_raw(a, W, y, M, N, K, 0, 0); e1.record()
        ev.append((e0, e1))
    else:
        ops.gemm_raw(a, W, y, M, N, K, 0, 0)
with torch.no_grad():
    for i in range(200):
        chain(i)
torch.cuda.synchronize()
ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev[20:])
print("LN -> GEMM chain, fresh allocations: median %.1f us, p10 %.1f, p90 %.1f" % (ts[len(ts) // 2], ts[len(ts) // 10], ts[9 * len(ts) // 10]))
