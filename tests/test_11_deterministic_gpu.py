"""`ops.set_deterministic(True)` (VPTR_DETERMINISTIC=1; C ABI: vptr_set_deterministic + vptr_sumsq_ws): the stage-2 train step is
reproducible BIT FOR BIT on one device -- the same state and batch give the same loss terms, the same gradient slab and the same
post-step parameters, run after run, eager and as hipGraph replays.  (The default path lets workgroups meet in fp32 atomics in a few
backward kernels: faster, reproducible to ~1e-7 only -- DESIGN.md section 8; the reference's cuDNN / cuBLAS path is not
bit-deterministic either.)  Tiny NAR / FAR models with dropout 0.1 and the full-size K64 model at batch 2 and at the bench's batch 16."""
import pytest
import torch

from helpers import build_transformer, jload, load
from oracle import fill

pytestmark = pytest.mark.gpu


def _tiny(dev, far):
    import vptr_amd.model as pkg
    from vptr_amd.train import FARTrainer, NARTrainer
    z = load("step_far_tiny" if far else "step_tiny")
    cfg, meta = jload(z, "cfg"), jload(z, "meta")
    enc = pkg.VPTREnc(1, meta["feat"], 3, "reflect")
    dec = pkg.VPTRDec(1, meta["feat"], 3, meta.get("out_layer", "Tanh"), "reflect")
    T = build_transformer(pkg, cfg, far, dropout=0.1)
    fill.apply_fill(enc, meta["seed"]); fill.apply_fill(dec, meta["seed"] + 10); fill.apply_fill(T, meta["seed"] + 20)
    n = meta["N"]
    past = fill.clip_input((n, cfg["Tp"], 1, meta["HW"], meta["HW"]), meta["seed"] + 100, "kth").to(dev)
    fut = fill.clip_input((n, cfg["Tf"], 1, meta["HW"], meta["HW"]), meta["seed"] + 200, "kth").to(dev)
    if far:
        return FARTrainer(enc.to(dev), dec.to(dev), T.to(dev), lr=1e-4, max_grad_norm=1.0), past, fut
    return NARTrainer(enc.to(dev), dec.to(dev), T.to(dev), batch_size=n, lr=1e-4, max_grad_norm=1.0, lam_pc=0.1), past, fut


def _k64(dev, n=2):
    import bench
    from vptr_amd.train import NARTrainer
    enc, dec, T = bench.build_models(dev, 0.1)
    tr = NARTrainer(enc, dec, T, batch_size=n, lr=1e-4, max_grad_norm=1.0, lam_pc=0.1)
    past, fut = bench.synth_batch(n, 0, dev)
    return tr, past, fut


def _run(tr, snap, past, fut, steps):
    tr._restore(snap)
    terms, grads = [], []
    for _ in range(steps):
        out = tr.step(past, fut)
        terms.append({k: float(v) for k, v in out.items()})
        grads.append(tr.opt.grad.clone())
    return terms, grads, tr.opt.flat.clone()


@pytest.mark.parametrize("model", ["tiny_nar", "tiny_far", "k64", "k64_n16"])
def test_train_step_is_bit_reproducible(dev, model):
    from vptr_amd import ops
    ops.unregister_flat_slabs()
    ops.manual_seed(dev, 77)
    ops.set_deterministic(True)
    try:
        if model.startswith("k64"):    # batch 2: the small-geometry launches (atomics by default); batch 16: the bench step's launches
            tr, past, fut = _k64(dev, 16 if model == "k64_n16" else 2)
        else:
            tr, past, fut = _tiny(dev, model == "tiny_far")
        snap = tr._snapshot()
        a = _run(tr, snap, past, fut, 3)
        b = _run(tr, snap, past, fut, 3)
        for s, (ta, tb) in enumerate(zip(a[0], b[0])):
            assert ta == tb, (s, ta, tb)
        for s, (ga, gb) in enumerate(zip(a[1], b[1])):
            assert torch.equal(ga, gb), "gradient slab of step %d differs in %d of %d elements (max |d| %.3e)" % (
                s, int((ga != gb).sum()), ga.numel(), float((ga - gb).abs().max()))
        assert torch.equal(a[2], b[2]), "post-step parameters differ"
        # ... and hipGraph replays reproduce the eager steps bit for bit (same kernels, same order, same inputs)
        tr._restore(snap)
        tr.capture(past, fut, warmup=1)
        c = _run(tr, snap, past, fut, 3)
        for s, (ta, tc) in enumerate(zip(a[0], c[0])):
            assert ta == tc, ("graph replay vs eager", s, ta, tc)
        assert torch.equal(a[2], c[2]), "post-step parameters of the replays differ from the eager run's"
        del tr
    finally:
        ops.set_deterministic(False)
        ops.unregister_flat_slabs()
        torch.cuda.empty_cache()


def test_default_mode_is_not_claimed_deterministic(dev):
    """the switch is off by default and `set_deterministic` restores the atomics-accumulated conv-FFN statistics when it is turned off"""
    from vptr_amd import _lib, ops
    assert not ops.config.deterministic and _lib.lib.vptr_get_deterministic() == 0
    before = ops.config.fused_frame_stats
    ops.set_deterministic(True)
    assert ops.config.deterministic and _lib.lib.vptr_get_deterministic() == 1 and not ops.config.fused_frame_stats
    ops.set_deterministic(False)
    assert _lib.lib.vptr_get_deterministic() == 0 and ops.config.fused_frame_stats == before
