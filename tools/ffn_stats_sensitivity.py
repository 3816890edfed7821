"""Tiny FAR train step: gradient reproducibility with the conv-FFN LayerNorm statistics accumulated by the producers (atomics) vs the
separate deterministic pass, the accuracy of the accumulated sums, and the fixture's sensitivity to a 1e-7 input perturbation.  GPU box."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import vptr_amd.model as pkg
from helpers import build_transformer, jload, load
from oracle import fill
from vptr_amd import ops
from vptr_amd.train import FARTrainer
dev = torch.device("cuda:0")
z = load("step_far_tiny")
cfg, meta = jload(z, "cfg"), jload(z, "meta")
print(cfg, meta)
_orig = ops.norm_act
def _patched(x, w, b, mode, HW, training, *a, **kw):
    raw = kw.get("raw_stats")
    if raw is not None:
        n = HW * x.shape[1]
        xd = x.detach().double().view(-1, n)
        m_ref, v_ref = xd.mean(1), xd.var(1, unbiased=False)
        m = raw[:, 0].double() / n
        v = raw[:, 1].double() / n - m * m
        print("  norm F=%d frames=%d: mean^2/var max %.1f  var rel err max %.2e  rstd rel err %.2e" % (
            x.shape[1], xd.shape[0], float((m_ref ** 2 / v_ref).max()), float(((v - v_ref).abs() / v_ref).max()),
            float((((v + 1e-5).rsqrt() - (v_ref + 1e-5).rsqrt()).abs() * (v_ref + 1e-5).sqrt()).max())))
    return _orig(x, w, b, mode, HW, training, *a, **kw)
ops.norm_act = _patched
def run(fused, G=4, noise=0.0):
    ops.config.fused_frame_stats = fused
    ops.unregister_flat_slabs()
    enc = pkg.VPTREnc(1, meta["feat"], 3, "reflect"); dec = pkg.VPTRDec(1, meta["feat"], 3, meta["out_layer"], "reflect")
    T = build_transformer(pkg, cfg, True)
    fill.apply_fill(enc, meta["seed"]); fill.apply_fill(dec, meta["seed"] + 10); fill.apply_fill(T, meta["seed"] + 20)
    tr = FARTrainer(enc.to(dev), dec.to(dev), T.to(dev), lr=1e-4, max_grad_norm=1.0)
    past = fill.rand_input((G, cfg["Tp"], 1, meta["HW"], meta["HW"]), meta["seed"] + 100).to(dev)
    fut = fill.rand_input((G, cfg["Tf"], 1, meta["HW"], meta["HW"]), meta["seed"] + 200).to(dev)
    if noise:
        past = past * (1 + noise * torch.randn_like(past))
    tr.step(past, fut)
    global last_named
    last_named = {n: ops.flat_grad_for(p).detach().clone() for n, p in tr.T.named_parameters() if ops.flat_grad_for(p) is not None}
    return tr.opt.grad.detach().clone()
def rel(a, b): return float((a - b).norm() / b.norm())
ops.norm_act = _orig
a = run(True); na = last_named; b = run(True); nb = last_named; c = run(False); nc = last_named; d = run(False)
worst = sorted(((float((na[k] - nb[k]).norm() / (nb[k].norm() + 1e-30)), float((na[k] - nc[k]).norm() / (nc[k].norm() + 1e-30)), k) for k in na), reverse=True)[:12]
for w in worst: print("  fused/fused %.2e  fused/plain %.2e  %s" % w)
print("fused vs fused %.2e  plain vs plain %.2e  fused vs plain %.2e" % (rel(a, b), rel(c, d), rel(a, c)))
a2 = run(True, 2); c2 = run(False, 2)
print("G=2 fused vs plain %.2e" % rel(a2, c2))

e = run(False, 4, 1e-7); f = run(False, 4, 1e-7)
print("plain, inputs perturbed by 1e-7 relative: grad change %.2e %.2e" % (rel(e, c), rel(f, c)))
