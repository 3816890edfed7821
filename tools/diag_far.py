"""Diagnostic (GPU box): per-tensor gradient errors of a tiny FAR/NAR config against the CPU oracle."""
import json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import build_transformer, load, jload, rel
from oracle import fill, vptr_oracle as O
import vptr_amd.model as pkg

def run(cfg, far, N, seed, tag):
    dev = torch.device("cuda:0")
    m = build_transformer(pkg, cfg, far); fill.apply_fill(m, seed)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    Tin = cfg.get("Tin", cfg["Tp"]); Tout = Tin if far else cfg["Tf"]
    x = fill.rand_normal((N, Tin, cfg["C"], cfg["H"], cfg["W"]), seed + 1).abs()
    g = fill.rand_normal((N, Tout, cfg["C"], cfg["H"], cfg["W"]), seed + 2)
    P = {k: v.clone() for k, v in sd.items()}
    leaves = {}
    for k, _ in m.named_parameters():
        P[k] = P[k].clone().requires_grad_(True); leaves[k] = P[k]
    xo = x.clone().requires_grad_(True)
    fwd = O.far_forward if far else O.nar_forward
    oo = fwd(P, xo, cfg, training=True); (oo * g).sum().backward()
    m = m.to(dev).train()
    xd = x.to(dev).requires_grad_(True)
    od = m(xd); (od * g.to(dev)).sum().backward()
    print("==", tag, "out", "%.2e" % rel(od, oo), "dx", "%.2e" % rel(xd.grad, xo.grad))
    errs = []
    norms = [float(v.grad.norm()) for v in leaves.values() if v.grad is not None]
    floor = 1e-2 * float(np.median(norms))
    for k, p in m.named_parameters():
        if leaves[k].grad is None: continue
        errs.append((rel(p.grad, leaves[k].grad, floor), k))
    errs.sort(reverse=True)
    for e, k in errs[:8]: print("   %.2e %s" % (e, k))

tiny = dict(Tp=3, Tf=3, H=8, W=8, C=48, nhead=8, window_size=4, num_encoder_layers=1, num_decoder_layers=1, rpe=True)
run(dict(tiny, Tin=5, num_encoder_layers=2), True, 2, 13, "far 2L N2 T5")
run(dict(tiny, Tin=5, num_encoder_layers=1), True, 2, 13, "far 1L N2 T5")
run(dict(tiny, Tin=3, num_encoder_layers=1), True, 2, 13, "far 1L N2 T3")
run(dict(tiny, Tin=5, num_encoder_layers=1), True, 1, 13, "far 1L N1 T5")
run(dict(tiny, Tin=4, num_encoder_layers=1), True, 2, 13, "far 1L N2 T4 (8 frames)")
run(dict(tiny, Tin=5, num_encoder_layers=1, C=96), True, 2, 13, "far 1L N2 T5 C96")
