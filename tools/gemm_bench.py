"""GEMM micro-benchmark (GPU box): TF/s per model shape, mode and precision, HIP-event timed."""
import sys, os, json
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import vptr_amd.ops as ops

dev = torch.device("cuda:0")
Mtok = int(os.environ.get("MTOK", 10240))
shapes = [
    ("fwd  528x528", 0, 0, Mtok, 528, 528), ("fwd  2112x528", 0, 0, Mtok, 2112, 528), ("fwd  528x2112", 0, 0, Mtok, 528, 2112),
    ("dgrd 528x528", 0, 1, Mtok, 528, 528), ("dgrd 528<-2112", 0, 1, Mtok, 528, 2112), ("dgrd 2112<-528", 0, 1, Mtok, 2112, 528),
    ("wgrd 528x528", 1, 1, 528, 528, Mtok), ("wgrd 2112x528", 1, 1, 2112, 528, Mtok), ("wgrd 528x2112", 1, 1, 528, 2112, Mtok),
]
if os.environ.get("EXTRA"):
    shapes += [("fwd  1056x528", 0, 0, Mtok, 1056, 528), ("fwd  1584x528", 0, 0, Mtok, 1584, 528),
               ("dgrd 528<-1056", 0, 1, Mtok, 528, 1056), ("dgrd 528<-1584", 0, 1, Mtok, 528, 1584)]
precs = [int(p) for p in os.environ.get("PRECS", "3,1").split(",")]
res = []
only = os.environ.get("ONLY")
for name, am, bm, M, N, K in shapes:
    if only and name != only:
        continue
    for prec in precs:
        if am == 0:
            A = torch.randn(M, K, device=dev)
        else:
            A = torch.randn(K, M, device=dev)
        B = torch.randn(N, K, device=dev) if bm == 0 else torch.randn(K, N, device=dev)
        D = torch.zeros(M, N, device=dev)
        kw = {}
        if am == 1:
            tiles = ((M + 127) // 128) * ((N + 175) // 176)
            kw = dict(atomic=True, split_k=ops._split_k_for(tiles, K))
        for _ in range(3):
            ops.gemm_raw(A, B, D, M, N, K, am, bm, precision=prec, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.gemm_raw(A, B, D, M, N, K, am, bm, precision=prec, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        tf = 2.0 * M * N * K / us / 1e6
        print("%-16s prec %d  M %6d N %5d K %6d  %8.1f us  %7.1f TF/s  %s" % (name, prec, M, N, K, us, tf, kw.get("split_k", "")))
