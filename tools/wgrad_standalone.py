#!/usr/bin/env python
"""The grouped weight-gradient launch of the K64 bench step ALONE, on the step's real problem set: one forward + backward of the
NARTrainer with the end-of-backward flush held back, then the recorded (dY, X, dW) problems are launched `--reps` times and timed
with HIP events.  Variants are selected through the environment of this process (VPTR_WGRAD_* switches read by the launcher /
ops._launch_wgrad_group); run under rocprofv3 --pmc for FETCH_SIZE / TCC_HIT / TCC_MISS of exactly this launch.

    python tools/wgrad_standalone.py --reps 10 [--order recorded|address|shape]
"""
import argparse
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--order", default="recorded", choices=["recorded", "address", "shape"])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    import vptr_amd.ops as ops
    from vptr_amd.train import NARTrainer
    enc, dec, T = bench.build_models(dev, 0.1)
    tr = NARTrainer(enc, dec, T, batch_size=args.batch, lr=1e-4, max_grad_norm=1.0, lam_pc=0.1)
    past, fut = bench.synth_batch(args.batch, 0, dev)
    tr.step(past, fut)
    # one more forward / backward with the flush held back: the queue then holds the step's problems in backward order
    with torch.no_grad():
        feats = tr.enc(torch.cat([past, fut], dim=1))
    pf, ff = feats[:, :10], feats[:, 10:]
    tr.opt.zero_grad()
    pred_feats = tr.T(pf)
    pred_frames = tr.dec(pred_feats)
    loss = tr.losses(pred_frames, fut, pred_feats, ff)[0]
    with ops.hold_wgrads():
        loss.backward()
    items = list(ops._wgrad_q)
    del ops._wgrad_q[:]
    if args.order == "address":
        items.sort(key=lambda it: it[2].data_ptr())
    elif args.order == "shape":
        items.sort(key=lambda it: (-it[3] * it[4], it[1].data_ptr(), it[2].data_ptr()))
    flops = sum(2.0 * it[5] * it[3] * it[4] for it in items)
    shapes = {}
    for it in items:
        shapes[(it[3], it[4])] = shapes.get((it[3], it[4]), 0) + 1
    print("problems %d  shapes %s  GF %.1f" % (len(items), shapes, flops / 1e9))
    # build the descriptor tables ONCE (host work + uploads), then time bare launches of the kernel
    keep, calls = [], []
    orig_upload, orig_launch = ops.wgrad._to_device_async, ops.lib.vptr_gemm_grouped

    def upload(b, d):
        t = orig_upload(b, d)
        keep.append(t)
        return t

    def launch(proto, raw, st, n, total, stream):
        import ctypes
        pc = type(proto._obj)()
        ctypes.memmove(ctypes.byref(pc), proto, ctypes.sizeof(pc))
        calls.append((pc, raw, st, n, total))
        return orig_launch(proto, raw, st, n, total, stream)
    ops.wgrad._to_device_async, ops.lib.vptr_gemm_grouped = upload, launch
    ops._launch_wgrad_group(items)
    ops.wgrad._to_device_async, ops.lib.vptr_gemm_grouped = orig_upload, orig_launch
    torch.cuda.synchronize()
    import ctypes
    from vptr_amd._lib import stream
    print("launches per group flush: %d, tiles %s, descriptors %s" % (len(calls), [c[4] for c in calls], [c[3] for c in calls]))

    def fire():
        for (pc, raw, st, n, total) in calls:
            assert orig_launch(ctypes.byref(pc), raw, st, n, total, stream()) == 0
    for _ in range(3):
        fire()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fire()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    med = ts[len(ts) // 2]
    env = {k: v for k, v in os.environ.items() if k.startswith("VPTR_WGRAD")}
    print("wgrad grouped launch: order %s env %s  median %.3f ms  min %.3f  max %.3f  -> %.1f TFLOP/s" % (
        args.order, env, med, ts[0], ts[-1], flops / med / 1e9))


if __name__ == "__main__":
    main()
