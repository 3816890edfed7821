"""wgrad split-K sweep (GPU box)."""
import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import vptr_amd.ops as ops
dev = torch.device("cuda:0")
K = 10240
for (M, N) in [(528, 528), (2112, 528), (528, 2112)]:
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); D = torch.zeros(M, N, device=dev)
    for sk in [1, 2, 4, 8, 16, 32, 64]:
        for _ in range(2): ops.gemm_raw(A, B, D, M, N, K, 1, 1, precision=3, atomic=True, split_k=sk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10): ops.gemm_raw(A, B, D, M, N, K, 1, 1, precision=3, atomic=True, split_k=sk)
        g.replay(); torch.cuda.synchronize()
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print("wgrad %4dx%4d split %2d: %7.1f us %6.1f TF/s" % (M, N, sk, us, 2.0 * M * N * K / us / 1e6))
