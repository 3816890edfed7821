"""Attention-core micro-benchmark at the step's shapes (GPU box): window / temporal, fwd / bwd, with and without dropout and bias."""
import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import vptr_amd.ops as ops
from vptr_amd._lib import lib, ptr, stream, check
from oracle import vptr_oracle as O

dev = torch.device("cuda:0")
B, H, W, C, nh, ws = 160, 8, 8, 528, 8, 4
M = B * H * W
q, k, v, do = (torch.randn(M, C, device=dev) for _ in range(4))
o, dq, dk, dv = (torch.empty(M, C, device=dev) for _ in range(4))
table = torch.randn((2 * ws - 1) ** 2, nh, device=dev)
dtable = torch.zeros_like(table)
idx = O.rpe_index(ws).to(dev)
seed = ops.seed_tensor(dev)
P16 = int(os.environ.get("P16", "1"))   # outputs in the P16 plane format, as the model requests them
print("VPTR_ATTN_MFMA =", os.environ.get("VPTR_ATTN_MFMA"), " P16 =", P16)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for p in (0.1, 0.0):
    for tb in (table, None):
        f = timed(lambda: check(lib.vptr_winattn_fwd(ptr(q), ptr(k), ptr(v), ptr(tb), ptr(idx), ptr(o), B, H, W, C, nh, ws, p, ptr(seed), 3, P16, stream()), "f"))
        b = timed(lambda: check(lib.vptr_winattn_bwd(ptr(q), ptr(k), ptr(v), ptr(tb), ptr(idx), ptr(do), ptr(dq), ptr(dk), ptr(dv),
                                                     ptr(dtable) if tb is not None else None, B, H, W, C, nh, ws, p, ptr(seed), 3, 1.0, P16, stream()), "b"))
        wsp = torch.empty(lib.vptr_winattn_bwd_workspace(nh), device=dev)
        b2 = timed(lambda: check(lib.vptr_winattn_bwd_ws(ptr(q), ptr(k), ptr(v), ptr(tb), ptr(idx), ptr(do), ptr(dq), ptr(dk), ptr(dv),
                                                         ptr(dtable) if tb is not None else None, B, H, W, C, nh, ws, p, ptr(seed), 3, 1.0, P16,
                                                         ptr(wsp), wsp.numel(), stream()), "b"))
        print("window   p=%.1f bias=%d  fwd %6.1f us  bwd %6.1f us  bwd with workspace %6.1f us" % (p, tb is not None, f, b, b2))
N, T, HW = 16, 10, 64
for p in (0.1, 0.0):
    f = timed(lambda: check(lib.vptr_tattn_fwd(ptr(q), ptr(k), ptr(v), ptr(o), N, T, T, HW, C, nh, 0, p, ptr(seed), 3, P16, stream()), "f"))
    b = timed(lambda: check(lib.vptr_tattn_bwd(ptr(q), ptr(k), ptr(v), ptr(do), ptr(dq), ptr(dk), ptr(dv), N, T, T, HW, C, nh, 0, p, ptr(seed), 3, 1.0,
                                               P16, stream()), "b"))
    print("temporal p=%.1f          fwd %6.1f us  bwd %6.1f us" % (p, f, b))
# reference points: a plain copy of the same bytes
src = torch.randn(4 * M * C, device=dev); dst = torch.empty(3 * M * C, device=dev)
print("copy of 3 tensors (6 x 21.6 MB moved): %6.1f us" % timed(lambda: dst.copy_(src[:3 * M * C])))
