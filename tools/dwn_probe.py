"""Stand-alone timing of the conv-FFN forward chain at the K64 step's shape (GPU box): the fused norm1 + GELU + depthwise launch
(vptr_dwconv3x3_norm_fwd; VPTR_DWN_LDS / VPTR_DWN_DBG select the kernel and its elimination variants, read once per process) against the
two launches it replaces, plus LayerNorm(528) forward with fp32 and P16 output.  python tools/dwn_probe.py [--reps 50]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vptr_amd._lib import check, lib, ptr, stream  # noqa: E402
import vptr_amd.ops as ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=50)
args = ap.parse_args()
dev = torch.device("cuda:0")
frames, H, W, F = 160, 8, 8, 2112
HW, rows = H * W, frames * H * W
torch.manual_seed(0)
x = torch.randn(rows, F, device=dev)
aw, ab = torch.rand(HW, F, device=dev) + 0.5, torch.randn(HW, F, device=dev) * 0.1
w9, b9 = torch.randn(9, F, device=dev) * 0.3, torch.randn(F, device=dev) * 0.1
raw = torch.zeros(frames, ops.FRAME_STATS_STRIDE, device=dev)
raw[:, 0], raw[:, 1] = x.view(frames, -1).sum(1), (x.view(frames, -1) ** 2).sum(1)
y, ah = torch.empty_like(x), torch.empty(rows, F, device=dev, dtype=torch.float16)
mean, rstd, st2 = torch.empty(frames, device=dev), torch.empty(frames, device=dev), torch.zeros(frames, 2, device=dev)
a = torch.empty_like(x)
flush = torch.empty(128 << 20, device=dev)   # 512 MB: evicts the Infinity Cache between repetitions when --cold


def timed(fn, reps=args.reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    return t[len(t) // 2]


def fused():
    check(lib.vptr_dwconv3x3_norm_fwd(ptr(x), ptr(raw), ptr(aw), ptr(ab), 1e-5, 1, ptr(w9), ptr(b9), ptr(y), ptr(ah), ptr(mean), ptr(rstd),
                                      frames, H, W, F, ptr(st2), stream()), "fused")


def fused_noah():
    check(lib.vptr_dwconv3x3_norm_fwd(ptr(x), ptr(raw), ptr(aw), ptr(ab), 1e-5, 1, ptr(w9), ptr(b9), ptr(y), None, ptr(mean), ptr(rstd),
                                      frames, H, W, F, ptr(st2), stream()), "fused")


def norm_only():
    check(lib.vptr_norm_act_fwd(ptr(x), ptr(mean), ptr(rstd), ptr(aw), ptr(ab), ptr(a), rows, F, HW, 0, 1, 0.0, None, 0, None, 1, 1, None, 0,
                                ptr(raw), 1e-5, stream()), "norm_act_fwd")


def dw_only():
    check(lib.vptr_dwconv3x3_fwd(ptr(a), ptr(w9), ptr(b9), ptr(y), frames, H, W, F, ptr(st2), stream()), "dwconv")


env = {k: os.environ[k] for k in ("VPTR_DWN_LDS", "VPTR_DWN_DBG") if k in os.environ}
print("env %s: fused %.1f us   fused without fp16 side copy %.1f us   norm_act_fwd %.1f us + dwconv3x3_fwd %.1f us" % (
    env, timed(fused), timed(fused_noah), timed(norm_only), timed(dw_only)))
if not env:
    xs = torch.randn(10240, 528, device=dev)
    g, bb = torch.rand(528, device=dev) + 0.5, torch.randn(528, device=dev)
    o, m2, r2 = torch.empty_like(xs), torch.empty(10240, device=dev), torch.empty(10240, device=dev)
    for p16 in (0, 1):
        t = timed(lambda: check(lib.vptr_layernorm_fwd(ptr(xs), ptr(g), ptr(bb), ptr(o), None, None, 1, 1, ptr(m2), ptr(r2), 10240, 528, 1e-5, p16,
                                                       stream()), "ln"))
        print("layernorm_fwd 10240 x 528, p16 = %d: %.1f us" % (p16, t))
