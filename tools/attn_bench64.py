"""Attention-core micro-benchmark at the 17 ... 64-row shapes of BASELINE configs 4 / 5 (GPU box): 8 x 8 windows of KTH 128 x 128 (2 x 40 frames of
16 x 16 tokens), temporal T = 29 causal (BAIR FAR, 16 x 64 pixels), T = 40 self and 40 x 10 cross (KTH128 decoder).  P16 outputs, dropout 0.1."""
import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import vptr_amd.ops as ops
from vptr_amd._lib import lib, ptr, stream, check
from oracle import vptr_oracle as O
dev = torch.device("cuda:0")
C, nh = 528, 8
seed = ops.seed_tensor(dev)


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


def win(B, H, W, ws):
    M = B * H * W
    q, k, v, do = (torch.randn(M, C, device=dev) for _ in range(4))
    o, dq, dk, dv = (torch.empty(M, C, device=dev) for _ in range(4))
    table = torch.randn((2 * ws - 1) ** 2, nh, device=dev)
    dtable = torch.zeros_like(table)
    idx = O.rpe_index(ws).to(dev)
    f = timed(lambda: check(lib.vptr_winattn_fwd(ptr(q), ptr(k), ptr(v), ptr(table), ptr(idx), ptr(o), B, H, W, C, nh, ws, 0.1, ptr(seed), 3, 1, stream()), "f"))
    b = timed(lambda: check(lib.vptr_winattn_bwd(ptr(q), ptr(k), ptr(v), ptr(table), ptr(idx), ptr(do), ptr(dq), ptr(dk), ptr(dv), ptr(dtable), B, H, W, C, nh, ws, 0.1,
                                                 ptr(seed), 3, 1.0, 1, stream()), "b"))
    print("window %dx%d frames %d (M %d): fwd %6.1f us (ideal %.0f)  bwd %6.1f us (ideal %.0f)" % (ws, ws, B, M, f, 4 * M * C * 4 / 5e6, b, 7 * M * C * 4 / 5e6))


def temporal(N, Tq, Tk, HW, causal):
    Mq, Mk = N * Tq * HW, N * Tk * HW
    q, do = torch.randn(Mq, C, device=dev), torch.randn(Mq, C, device=dev)
    k, v = torch.randn(Mk, C, device=dev), torch.randn(Mk, C, device=dev)
    o, dq = torch.empty(Mq, C, device=dev), torch.empty(Mq, C, device=dev)
    dk, dv = torch.empty(Mk, C, device=dev), torch.empty(Mk, C, device=dev)
    f = timed(lambda: check(lib.vptr_tattn_fwd(ptr(q), ptr(k), ptr(v), ptr(o), N, Tq, Tk, HW, C, nh, causal, 0.1, ptr(seed), 3, 1, stream()), "f"))
    b = timed(lambda: check(lib.vptr_tattn_bwd(ptr(q), ptr(k), ptr(v), ptr(do), ptr(dq), ptr(dk), ptr(dv), N, Tq, Tk, HW, C, nh, causal, 0.1, ptr(seed), 3, 1.0, 1, stream()), "b"))
    print("temporal N %d Tq %d Tk %d HW %d causal %d: fwd %6.1f us (ideal %.0f)  bwd %6.1f us (ideal %.0f)" % (
        N, Tq, Tk, HW, causal, f, (2 * Mq + 2 * Mk) * C * 4 / 5e6, b, (3 * Mq + 4 * Mk) * C * 4 / 5e6))


print("lib:", os.environ.get("VPTR_HIP_LIB", "in-tree"))
win(80, 16, 16, 8)
temporal(16, 29, 29, 64, 1)
temporal(2, 40, 40, 256, 0)
temporal(2, 40, 10, 256, 0)
