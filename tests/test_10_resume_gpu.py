"""Optimizer-state interchange on the GPU: `FlatAdamW` (flat slabs, channel-last STORED LayerNorm((C,H,W)) affines and depthwise
weights) <-> `torch.optim.AdamW` (the format of the reference's `optimizer_T` checkpoints, utils/train_summary.py:10-38,130-160).

Two NARTrainer steps, export the state into a torch.optim.AdamW over a plain-parameter clone, third step on both from the
same gradients -> identical parameters; then the reverse (torch state -> a fresh FlatAdamW) for a fourth step."""
import pytest
import torch

from helpers import build_transformer, jload, load
from oracle import fill

pytestmark = pytest.mark.gpu


def _batch(meta, cfg, s, dev):
    past = ((fill.rand_input((meta["N"], cfg["Tp"], 1, meta["HW"], meta["HW"]), meta["seed"] + 100 + s) - 0.6013795) / 2.7570653).to(dev)
    fut = ((fill.rand_input((meta["N"], cfg["Tf"], 1, meta["HW"], meta["HW"]), meta["seed"] + 200 + s) - 0.6013795) / 2.7570653).to(dev)
    return past, fut


def _update_err(ours, theirs, before, names):
    """|| ours - theirs || relative to the size of the update || theirs - before || over all PARAMETERS (`names`; buffers such
    as the BatchNorm running statistics move in the replica that runs the forward pass only)"""
    num = den = 0.0
    for k in names:
        v = theirs[k]
        num += float((ours[k].double() - v.double()).pow(2).sum())
        den += float((v.double() - before[k].double()).pow(2).sum())
    return (num / max(den, 1e-300)) ** 0.5


@pytest.mark.parametrize("fixture", ["step_tiny", "step_k64_digest"])
def test_flat_adamw_state_round_trip_through_torch_adamw(dev, fixture):
    import vptr_amd.model as pkg
    from vptr_amd import ops
    from vptr_amd.train import NARTrainer
    z = load(fixture)
    cfg, meta = jload(z, "cfg"), jload(z, "meta")

    def models():
        enc = pkg.VPTREnc(1, meta["feat"], 3, "reflect")
        dec = pkg.VPTRDec(1, meta["feat"], 3, "Tanh", "reflect")
        T = build_transformer(pkg, cfg, False)
        fill.apply_fill(enc, meta["seed"])
        fill.apply_fill(dec, meta["seed"] + 10)
        fill.apply_fill(T, meta["seed"] + 20)
        return enc.to(dev), dec.to(dev), T.to(dev)

    enc, dec, T = models()
    tr = NARTrainer(enc, dec, T, batch_size=meta["N"], lr=1e-4, max_grad_norm=1.0, lam_pc=0.1)
    n_cl = sum(1 for lay in tr.opt._layout if lay[2] is not None)
    assert n_cl > 0, "the test model must contain channel-last stored parameters"
    for s in range(2):
        tr.step(*_batch(meta, cfg, s, dev))

    # ---- ours -> torch.optim.AdamW
    T2 = build_transformer(pkg, cfg, False).to(dev)
    T2.load_state_dict(T.state_dict())
    topt = torch.optim.AdamW(T2.parameters(), lr=1e-4, weight_decay=1e-2)
    topt.load_state_dict(tr.opt.state_dict())
    before = {k: v.detach().clone() for k, v in T.state_dict().items()}
    tr.step(*_batch(meta, cfg, 2, dev))                      # third step on the HIP path; its gradients stay in the slab
    for p2, p in zip(T2.parameters(), T.parameters()):
        p2.grad = p.grad.detach().clone().contiguous()       # logical layout (channel-last stored ones via their permuted view)
    torch.nn.utils.clip_grad_norm_(T2.parameters(), 1.0)
    topt.step()
    e = _update_err(T.state_dict(), T2.state_dict(), before, [k for k, _ in T2.named_parameters()])
    # parameters are O(1) fp32 numbers and one update is ~lr = 1e-4: one ulp of rounding difference between the two
    # implementations (fma contraction) is already ~5e-4 of the update; a scrambled or stale optimizer state gives O(1)
    assert e < 2e-3, "third step after exporting the state to torch.optim.AdamW: update differs by %.3e" % e

    # ---- torch.optim.AdamW -> a fresh FlatAdamW (what resuming from a reference checkpoint does)
    enc3, dec3, T3 = models()
    T3.load_state_dict(T2.state_dict())
    ops.unregister_flat_slabs()
    tr3 = NARTrainer(enc3, dec3, T3, batch_size=meta["N"], lr=1e-4, max_grad_norm=1.0, lam_pc=0.1)
    tr3.opt.load_state_dict(topt.state_dict())
    assert float(tr3.opt.step_dev) == 3.0
    before = {k: v.detach().clone() for k, v in T3.state_dict().items()}
    tr3.step(*_batch(meta, cfg, 3, dev))
    for p2, p in zip(T2.parameters(), T3.parameters()):
        p2.grad = p.grad.detach().clone().contiguous()
    torch.nn.utils.clip_grad_norm_(T2.parameters(), 1.0)
    topt.step()
    e = _update_err(T3.state_dict(), T2.state_dict(), before, [k for k, _ in T2.named_parameters()])
    assert e < 2e-3, "fourth step after importing torch.optim.AdamW state: update differs by %.3e" % e


def test_two_optimizers_and_collection_keep_gradient_destinations(dev):
    """process-wide slab registry: two FlatAdamW slabs alive at once route gradients by address; closing / collecting one (even after
    a new slab was allocated at its address) leaves the other's registration intact"""
    import gc
    from vptr_amd import ops
    from vptr_amd.train import FlatAdamW
    ops.unregister_flat_slabs()
    a = [torch.nn.Parameter(torch.randn(32, 16, device=dev)), torch.nn.Parameter(torch.randn(32, device=dev))]
    b = [torch.nn.Parameter(torch.randn(48, 16, device=dev))]
    oa, ob = FlatAdamW(a), FlatAdamW(b)
    assert ops.flat_grad_for(a[0]).data_ptr() == oa.grad.data_ptr() and ops.flat_grad_for(b[0]).data_ptr() == ob.grad.data_ptr()
    oa.close()
    assert ops.flat_grad_for(a[0]) is None and ops.flat_grad_for(b[0]).data_ptr() == ob.grad.data_ptr()
    del oa, a
    gc.collect()
    torch.cuda.empty_cache()
    c = [torch.nn.Parameter(torch.randn(32, 16, device=dev)), torch.nn.Parameter(torch.randn(32, device=dev))]
    oc = FlatAdamW(c)          # may reuse the collected slab's address
    del ob
    gc.collect()
    assert ops.flat_grad_for(c[0]) is not None and ops.flat_grad_for(c[0]).data_ptr() == oc.grad.data_ptr()
    oc.close()
    ops.unregister_flat_slabs()


def test_flat_adamw_grad_scale_equals_prescaled_gradients(dev):
    """the data-parallel trainers all-reduce a SUM and fold the mean's 1 / world into the optimizer kernel (`FlatAdamW.step(grad_scale)`,
    train.py `_exchange_held_wgrads`): step(grad_scale = s) on the summed gradients == step() on gradients multiplied by s beforehand --
    parameters, Adam moments and the reported (clipped-against) gradient norm, with clipping active and inactive"""
    from vptr_amd import ops
    from vptr_amd.train import FlatAdamW
    for max_norm, gmag in ((1.0, 3.0), (1.0, 1e-3), (None, 1.0)):      # clip engaged / not engaged / no clipping
        ops.unregister_flat_slabs()
        torch.manual_seed(11)
        shapes = [(48, 32), (48,), (7, 5, 3), (1000,)]
        pa = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
        pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
        oa = FlatAdamW(pa, lr=1e-3, max_grad_norm=max_norm)
        ob = FlatAdamW(pb, lr=1e-3, max_grad_norm=max_norm)
        world = 4
        for it in range(3):
            g = torch.randn_like(oa.grad) * gmag * world                 # "summed over 4 ranks"
            oa.grad.copy_(g)
            ob.grad.copy_(g * (1.0 / world))
            oa.step(grad_scale=1.0 / world)
            ob.step()
            na, nb = float(oa.grad_norm()), float(ob.grad_norm())
            assert abs(na - nb) <= 1e-5 * nb, (max_norm, gmag, it, na, nb)
            assert abs(nb - float((g / world).double().norm())) <= 1e-4 * nb
        for name in ("flat", "m", "v"):
            a, b = getattr(oa, name).double(), getattr(ob, name).double()
            assert float((a - b).norm()) <= 1e-6 * float(b.norm()) + 1e-12, (max_norm, gmag, name)
        oa.close(); ob.close()
    ops.unregister_flat_slabs()
