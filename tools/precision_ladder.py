#!/usr/bin/env python
"""Precision ladder of the GEMM operand schemes (CPU only; VERDICT r5 "judge's probe").

Every nn.Linear / 1x1-convolution product of the K64 VPTRFormerNAR (forward, input gradient, weight gradient) is emulated inside the fp32
oracle with the operands rounded the way a cheaper MFMA scheme would round them, fp32 accumulation throughout; the result is compared with
an fp64 run of the same oracle on the same weights and inputs (N = 1 clip, random-init weights of the K64 architecture, seed 3407).

    bf16 x1          one bf16 MFMA pass
    fp16 x1          one fp16 pass, per-tensor power-of-two scaling
    fp16 act-split   activations hi + lo fp16, weights single fp16 (2 passes; the weight-gradient product has two activation operands
                     and is run as 3 passes)
    bf16 x3          hi*hi + hi*lo + lo*hi of a bf16 hi / lo split -- the shipped scheme (csrc/gemm_p16.hip)

Reported: relative l2 error of the output, of the input gradient and of the parameter gradients (all tensors together / worst tensor).
The parity bar of the build is 1e-3 (BASELINE.json north_star): only the 3-pass split holds it on the gradients.

    python tools/precision_ladder.py [--layers-enc 4 --layers-dec 8]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _fp16_scaled(x):
    s = 2.0 ** torch.floor(torch.log2(x.abs().max().clamp_min(1e-30)))   # per-tensor power of two: |x / s| < 2
    return (x / s).to(torch.float16).to(torch.float32) * s


def _split(x, rnd):
    hi = rnd(x)
    return hi, rnd(x - hi)


def product(a, b, scheme, a_is_act=True, b_is_act=False):
    """a [M, K] @ b [K, N] under `scheme`, fp32 accumulate"""
    if scheme == "fp32":
        return a @ b
    if scheme == "bf16x1":
        return _bf16(a) @ _bf16(b)
    if scheme == "fp16x1":
        return _fp16_scaled(a) @ _fp16_scaled(b)
    if scheme == "fp16_actsplit":
        ah, al = _split(a, _fp16_scaled)
        if b_is_act:
            bh, bl = _split(b, _fp16_scaled)
            return ah @ bh + ah @ bl + al @ bh
        bh = _fp16_scaled(b)
        return ah @ bh + al @ bh
    if scheme == "bf16x3":
        ah, al = _split(a, _bf16)
        bh, bl = _split(b, _bf16)
        return ah @ bh + ah @ bl + al @ bh
    raise ValueError(scheme)


class _Lin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, scheme):
        ctx.save_for_backward(x, w)
        ctx.scheme, ctx.has_b = scheme, b is not None
        y = product(x.reshape(-1, x.shape[-1]), w.t(), scheme).reshape(*x.shape[:-1], w.shape[0])
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        d2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        dx = product(d2, w, ctx.scheme).reshape(x.shape)
        dw = product(d2.t(), x2, ctx.scheme, b_is_act=True)
        return dx, dw, (d2.sum(0) if ctx.has_b else None), None


class _FProxy:
    """oracle.vptr_oracle's `F` with linear / 1x1 conv2d routed through the emulated product"""

    def __init__(self, scheme):
        self.scheme = scheme

    def __getattr__(self, name):
        return getattr(F, name)

    def linear(self, x, w, b=None):
        if self.scheme == "fp32" or x.dtype != torch.float32:
            return F.linear(x, w, b)
        return _Lin.apply(x, w, b, self.scheme)

    def conv2d(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if self.scheme == "fp32" or x.dtype != torch.float32 or groups != 1 or tuple(w.shape[2:]) != (1, 1):
            return F.conv2d(x, w, b, stride, padding, dilation, groups)
        y = _Lin.apply(x.permute(0, 2, 3, 1), w.view(w.shape[0], w.shape[1]), b, self.scheme)
        return y.permute(0, 3, 1, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers-enc", type=int, default=4)
    ap.add_argument("--layers-dec", type=int, default=8)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    from oracle import vptr_oracle as O
    import vptr_amd.model as M
    torch.manual_seed(3407)
    Tp = Tf = 10
    cfg = dict(Tp=Tp, Tf=Tf, H=8, W=8, C=528, nhead=8, window_size=4, num_encoder_layers=args.layers_enc, num_decoder_layers=args.layers_dec, rpe=True)
    T = M.VPTRFormerNAR(Tp, Tf, 8, 8, 528, 8, args.layers_enc, args.layers_dec, 0.0, 4, 4, False, True)
    sd = {k: v.detach().clone() for k, v in T.state_dict().items()}
    feat = torch.relu(torch.randn(1, Tp, 528, 8, 8))
    cot = torch.randn(1, Tf, 528, 8, 8)
    buffers = ("temporal_pos", "lw_pos", "Tlw_pos")

    mask = [None]

    def run(scheme, dtype):
        P = {}
        for k, v in sd.items():
            t = v.to(dtype).clone() if v.is_floating_point() else v.clone()
            if v.is_floating_point() and k not in buffers and not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                t.requires_grad_(True)
            P[k] = t
        x = feat.to(dtype).clone().requires_grad_(True)
        O.F = _FProxy(scheme)
        try:
            out, pre = O.nar_forward(P, x, cfg, training=False, return_pre=True)
            # cotangent zeroed near the final ReLU's kink (as oracle/make_golden.py does) so that a rounding-flipped sign is not counted
            if mask[0] is None:   # from the fp64 reference run, shared by every scheme
                mask[0] = pre.detach().abs() > 2e-3
            c = cot.to(dtype) * mask[0]
            (out * c).sum().backward()
        finally:
            O.F = F
        return out.detach().double(), x.grad.double(), {k: p.grad.double() for k, p in P.items() if p.requires_grad and p.grad is not None}

    t0 = time.time()
    ref = run("fp32", torch.float64)
    print("fp64 reference: %.1f s" % (time.time() - t0))

    def rel(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-300))

    print("| scheme (MFMA passes) | out | dx | param grads (global / worst tensor) |\n|---|---|---|---|")
    for scheme, label in (("fp32", "fp32 (oracle itself)"), ("bf16x1", "bf16 x1"), ("fp16x1", "fp16 x1"),
                          ("fp16_actsplit", "fp16, activations split hi+lo, weights single (x2)"), ("bf16x3", "bf16 hi+lo both (x3, shipped)")):
        o, dx, g = run(scheme, torch.float32)
        med = sorted(v.norm().item() for v in ref[2].values())[len(ref[2]) // 2]
        live = [k for k in g if ref[2][k].norm().item() > 1e-4 * med]   # analytically-zero gradients (k_proj bias ...) have no relative error
        num = sum(float((g[k] - ref[2][k]).norm() ** 2) for k in live) ** 0.5
        den = sum(float(ref[2][k].norm() ** 2) for k in live) ** 0.5
        worst = max(live, key=lambda k: rel(g[k], ref[2][k]))
        print("| %s | %.1e | %.1e | %.1e / %.1e (%s) |" % (label, rel(o, ref[0]), rel(dx, ref[1]), num / den, rel(g[worst], ref[2][worst]), worst))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
