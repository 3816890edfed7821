#!/usr/bin/env python
"""Bisect of the whole-step hipGraph corruption at the bench configuration.

    python tools/graph_bisect.py --mode graph --batch 16 --steps 4 [--no-droppath] [--stash] [--dropout 0.1]

Prints every loss term per step; with --stash the decoder output / target tensors of the captured step are kept as static
outputs and the loss terms are recomputed EAGERLY from them after each replay (tells a corrupted loss intermediate from a
corrupted input)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="graph", choices=["graph", "eager"])
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-droppath", action="store_true")
    ap.add_argument("--stash", action="store_true")
    ap.add_argument("--top", type=int, default=0)
    ap.add_argument("--nce", default="orig", choices=["orig", "elementwise", "off", "hooks"])
    ap.add_argument("--layers", type=str, default="4,8")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    import vptr_amd.model as M
    import vptr_amd.model.vidhrformer as V
    import vptr_amd.ops as ops
    from vptr_amd.train import NARTrainer
    if args.no_droppath:
        V._droppath_scale = lambda p, training, count, device: None
    ops.config.gemm_precision = 3
    le, ld = [int(v) for v in args.layers.split(",")]
    torch.manual_seed(3407)
    enc = M.VPTREnc(1, feat_dim=528, n_downsampling=3, padding_type="reflect")
    dec = M.VPTRDec(1, feat_dim=528, n_downsampling=3, out_layer="Tanh", padding_type="reflect")
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        M.init_weights(enc)
        M.init_weights(dec)
    T = M.VPTRFormerNAR(10, 10, 8, 8, 528, 8, le, ld, args.dropout, 4, 4, False, True)
    enc, dec, T = enc.to(dev), dec.to(dev), T.to(dev)

    stash = {}

    class Tr(NARTrainer):
        def losses(self, pred_frames, future, pred_feats, future_feats):
            if args.stash:
                stash["pred"], stash["future"] = pred_frames, future
            return super().losses(pred_frames, future, pred_feats, future_feats)

    if args.nce == "elementwise":
        import torch.nn.functional as F
        import vptr_amd.model.criterion as Cr

        def fwd(self, gt_f, pred_f):
            N, T_, C, h, w = gt_f.shape
            g = gt_f.permute(0, 1, 3, 4, 2).reshape(N * T_, h * w, C)
            p = pred_f.permute(0, 1, 3, 4, 2).reshape(N * T_, h * w, C)
            pos = self.mask.to(g.dtype)
            neg = 1.0 - pos

            def mm(a, b):   # a [B, L, C] x b [B, L, C] -> [B, L, L] without a library GEMM
                return (a.unsqueeze(2) * b.unsqueeze(1)).sum(-1)
            s1 = (mm(g, p) * pos + mm(g, p.detach()) * neg) / self.temperature
            s2 = (mm(p, g) * pos + mm(p, g.detach()) * neg) / self.temperature
            target = torch.arange(h * w, device=g.device).repeat(N * T_)
            return 0.5 * (F.cross_entropy(s1.flatten(0, 1), target) + F.cross_entropy(s2.flatten(0, 1), target))
        Cr.BiPatchNCE.forward = fwd
    hookbuf = {}
    if args.nce == "hooks":
        import torch.nn.functional as F
        import vptr_amd.model.criterion as Cr

        def tap(name, t):
            if name not in hookbuf:
                hookbuf[name] = torch.zeros(t.shape, device=t.device, dtype=t.dtype)
            t.register_hook(lambda gr, name=name: (hookbuf[name].copy_(gr), None)[1])
            return t

        def fwd(self, gt_f, pred_f):
            N, T_, C, h, w = gt_f.shape
            tap("gt_f", gt_f), tap("pred_f", pred_f)
            g = tap("g", gt_f.permute(0, 1, 3, 4, 2).reshape(N * T_, h * w, C))
            p = tap("p", pred_f.permute(0, 1, 3, 4, 2).reshape(N * T_, h * w, C))
            pos = self.mask.to(g.dtype)
            neg = 1.0 - pos
            m1, m2 = tap("m1", torch.matmul(g, p.transpose(1, 2))), tap("m2", torch.matmul(g, p.detach().transpose(1, 2)))
            m3, m4 = tap("m3", torch.matmul(p, g.transpose(1, 2))), tap("m4", torch.matmul(p, g.detach().transpose(1, 2)))
            s1 = tap("s1", (m1 * pos + m2 * neg) / self.temperature)
            s2 = tap("s2", (m3 * pos + m4 * neg) / self.temperature)
            target = torch.arange(h * w, device=g.device).repeat(N * T_)
            return 0.5 * (F.cross_entropy(s1.flatten(0, 1), target) + F.cross_entropy(s2.flatten(0, 1), target))
        Cr.BiPatchNCE.forward = fwd
    if args.nce == "hooks":
        import vptr_amd.model.modules as Mod
        orig_fwd = Mod._NCEProjector.forward
        callno = [0]

        def pf(self, x):
            y = orig_fwd(self, x)
            if y.requires_grad:
                tap("projout%d" % (callno[0] % 2), y)
            callno[0] += 1
            return y
        Mod._NCEProjector.forward = pf
        orig_p16 = ops.to_p16
        p16no = [0]

        def dbg_to_p16(x):
            out = orig_p16(x)
            if getattr(dbg_to_p16, "on", False):
                k = "to_p16_%02d" % p16no[0]
                p16no[0] += 1
                if k + "_in" not in hookbuf:
                    hookbuf[k + "_in"] = torch.zeros_like(x)
                    hookbuf[k + "_out"] = torch.zeros_like(x)
                hookbuf[k + "_in"].copy_(x)
                hookbuf[k + "_out"].copy_(ops.p16_decode(out))
            return out
        ops.to_p16 = dbg_to_p16
        orig_bwd = ops._LinearFn.backward

        def bwd(ctx, dy):
            dbg_to_p16.on = getattr(bwd, "n", 0) < 4
            bwd.n = getattr(bwd, "n", 0) + 1
            r = orig_bwd(ctx, dy)
            dbg_to_p16.on = False
            return r
        ops._LinearFn.backward = staticmethod(bwd)
        orig_step = NARTrainer._step_impl

        def step_impl(self, past, future):
            bwd.n = 0
            p16no[0] = 0
            return orig_step(self, past, future)
        NARTrainer._step_impl = step_impl
    tr = Tr(enc, dec, T, batch_size=args.batch, lr=1e-4, max_grad_norm=1.0, lam_pc=0.0 if args.nce in ("off", "hooks") else 0.1)
    past, fut = bench.synth_batch(args.batch, 0, dev)
    if args.mode == "graph":
        tr.capture(past, fut, warmup=2)
    for s in range(args.steps):
        out = tr.step(past, fut)
        torch.cuda.synchronize()
        line = {k: round(float(v), 5) for k, v in out.items()}
        if args.stash:
            with torch.no_grad():
                p, f = stash["pred"], stash["future"]
                line["re_MSE"] = round(float(tr.mse(p, f)), 5)
                line["re_GDL"] = round(float(tr.gdl(f, p)), 5)
                line["pred_absmax"] = round(float(p.abs().max()), 4)
                line["fut_absmax"] = round(float(f.abs().max()), 4)
        if args.top:
            names = {id(p_): n for n, p_ in tr.T.named_parameters()}
            rows = []
            for i, p_ in enumerate(tr.opt.params):
                g = tr.opt._logical(tr.opt.grad, i)
                rows.append((float(g.double().pow(2).sum().sqrt()), float(g.abs().max()), names[id(p_)], tuple(p_.shape)))
            rows.sort(reverse=True)
            for r in rows[:args.top]:
                print("    |g| %.4e  max %.4e  %s %s" % r)
        for k_, v_ in hookbuf.items():
            print("    grad tap %-7s norm %.4e" % (k_, float(v_.double().norm())))
        print(args.mode, "N", args.batch, "dp_off" if args.no_droppath else "dp_on", "stash" if args.stash else "", s, line, flush=True)


if __name__ == "__main__":
    main()
