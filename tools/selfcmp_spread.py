"""Run-to-run spread of the train step on ONE box (the numbers the self-comparison tests' bounds are derived from).

The step is deterministic up to the order of fp32 atomics.  This script repeats R eager runs of S steps from the same state and
reports, per model, (a) the single-step spread (step 0: same parameters, same masks), (b) the spread of the TRAJECTORY after 5 and
10 steps (AdamW's ~lr * sign(g) first updates amplify (a)), (c) what `verify_graph` (lock-step + trajectory) measures for the
captured graph.  Bounds used in tests/test_2*_gpu.py are >= 10x the largest value this prints across the boxes of
profiles/r04_selfcmp_spread.log.

    python tools/selfcmp_spread.py [R] [S]
"""
import json
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import bench  # noqa: E402
import vptr_amd.model as pkg  # noqa: E402
from helpers import build_transformer, jload, load  # noqa: E402
from oracle import fill  # noqa: E402
from vptr_amd import ops  # noqa: E402
from vptr_amd.train import FARTrainer, NARTrainer  # noqa: E402

dev = torch.device("cuda:0")
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def tiny(far, dropout):
    z = load("step_far_tiny" if far else "step_tiny")
    cfg, meta = jload(z, "cfg"), jload(z, "meta")
    enc = pkg.VPTREnc(1, meta["feat"], 3, "reflect")
    dec = pkg.VPTRDec(1, meta["feat"], 3, meta.get("out_layer", "Tanh"), "reflect")
    T = build_transformer(pkg, cfg, far, dropout=dropout)
    fill.apply_fill(enc, meta["seed"]); fill.apply_fill(dec, meta["seed"] + 10); fill.apply_fill(T, meta["seed"] + 20)
    n = meta.get("N", 4)
    past = ((fill.rand_input((n, cfg["Tp"], 1, meta["HW"], meta["HW"]), meta["seed"] + 100) - 0.6013795) / 2.7570653).to(dev)
    fut = ((fill.rand_input((n, cfg["Tf"], 1, meta["HW"], meta["HW"]), meta["seed"] + 200) - 0.6013795) / 2.7570653).to(dev)
    if far:
        tr = FARTrainer(enc.to(dev), dec.to(dev), T.to(dev), lr=1e-4, max_grad_norm=1.0)
    else:
        tr = NARTrainer(enc.to(dev), dec.to(dev), T.to(dev), batch_size=n, lr=1e-4, max_grad_norm=1.0, lam_pc=0.1)
    return tr, past, fut


def k64(n):
    enc, dec, T = bench.build_models(dev, 0.1)
    tr = NARTrainer(enc, dec, T, batch_size=n, lr=1e-4, max_grad_norm=1.0, lam_pc=0.1)
    past, fut = bench.synth_batch(n, 0, dev)
    return tr, past, fut


def spread(name, tr, past, fut, graph=True):
    snap = tr._snapshot()
    recs, grads, params = [], [], []
    for r in range(NR):
        tr._restore(snap)
        rr = []
        for s in range(NS):
            out = tr.step(past, fut)
            rr.append({k: float(v) for k, v in out.items()})
            if s == 0:
                grads.append(tr.opt.grad.double().clone())
                params.append(tr.opt.flat.double().clone())
        recs.append(rr)
    res = {"model": name}
    for s in sorted({0, min(4, NS - 1), NS - 1}):
        worst = {}
        for k in recs[0][s]:
            vals = [recs[r][s][k] for r in range(NR)]
            worst[k] = (max(vals) - min(vals)) / (abs(vals[0]) + 1e-6)
        res["term_spread_step%d" % s] = {k: float("%.3g" % v) for k, v in worst.items()}
    res["grad_slab_rel_step0"] = float("%.3g" % max(float((g - grads[0]).norm() / grads[0].norm()) for g in grads[1:]))
    res["param_rel_step0"] = float("%.3g" % max(float((p - params[0]).norm() / params[0].norm()) for p in params[1:]))
    upd = params[0] - snap["flat"].double()
    res["update_rel_step0"] = float("%.3g" % max(float((p - params[0]).norm() / upd.norm()) for p in params[1:]))
    if graph:
        tr._restore(snap)
        tr.capture(past, fut, warmup=2)
        tr._restore(snap)
        ok, rep = tr.verify_graph(past, fut, steps=5)
        res["verify_graph"] = {"ok": ok, "lockstep_worst": float("%.3g" % rep["worst_term_rel_diff"]), "where": str(rep["worst_term"]),
                               "param_rel_l2": float("%.3g" % rep["param_rel_l2"]),
                               "trajectory_worst": float("%.3g" % rep["trajectory_worst_rel_diff"])}
    print(json.dumps(res), flush=True)
    del tr
    ops.unregister_flat_slabs()
    torch.cuda.empty_cache()


print(json.dumps({"device": torch.cuda.get_device_name(0), "runs": NR, "steps": NS, "fused_stats": ops.config.fused_frame_stats}))
for far in (False, True):
    for dp in (0.0, 0.1):
        ops.unregister_flat_slabs()
        ops.manual_seed(dev, 1234)
        spread("tiny_%s_dropout%g" % ("far" if far else "nar", dp), *tiny(far, dp))
ops.unregister_flat_slabs()
ops.manual_seed(dev, 99)
spread("k64_n2", *k64(2))
ops.unregister_flat_slabs()
ops.manual_seed(dev, 99)
spread("k64_n4", *k64(4))
