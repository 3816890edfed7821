"""Whole-step hipGraph (`NARTrainer.capture`): replays must equal eager steps.  The step is deterministic up to the order of fp32
atomics -- also with dropout 0.1 and DropPath ON, because every mask and stochastic-depth vector comes from the device-resident
counter seed that eager steps and replays advance identically (no torch generator in the step since round 3).  Losses, gradient
norms and post-step parameters of 10 replays (5 with a device -> host read after each, 5 back-to-back) are compared with 10 eager
steps from the same initial state -- on the tiny model and at the bench configuration (K64, 4 + 8 layers, batch 4, dropout 0.1).

Round-2 post-mortem: the bench-size graph trained on garbage from its second replay on, because ATen's
F.normalize backward (BiPatchNCE branch) zeroes reduction semaphores with cudaMemsetAsync and a captured memset node is broken on
this HIP runtime; the tiny test could not see it (its reductions are single-block).  `capture()` now refuses graphs with memset
nodes and the losses are plain kernels; both are tested here."""
import pytest
import torch

from helpers import build_transformer, jload, load
from oracle import fill

pytestmark = pytest.mark.gpu


def _make(pkg, cfg, meta, dev, dropout):
    enc = pkg.VPTREnc(1, meta["feat"], 3, "reflect")
    dec = pkg.VPTRDec(1, meta["feat"], 3, "Tanh", "reflect")
    T = build_transformer(pkg, cfg, False, dropout=dropout)
    fill.apply_fill(enc, meta["seed"])
    fill.apply_fill(dec, meta["seed"] + 10)
    fill.apply_fill(T, meta["seed"] + 20)
    return enc.to(dev), dec.to(dev), T.to(dev)


def _batch(meta, cfg, s, dev):
    past = ((fill.rand_input((meta["N"], cfg["Tp"], 1, meta["HW"], meta["HW"]), meta["seed"] + 100 + s) - 0.6013795) / 2.7570653).to(dev)
    fut = ((fill.rand_input((meta["N"], cfg["Tf"], 1, meta["HW"], meta["HW"]), meta["seed"] + 200 + s) - 0.6013795) / 2.7570653).to(dev)
    return past, fut


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_graph_replays_match_eager_steps(dev, dropout, monkeypatch):
    import vptr_amd.model as pkg
    import vptr_amd.model.vidhrformer as V
    from vptr_amd import ops
    from vptr_amd.train import NARTrainer
    if dropout == 0.0:   # the 2e-4 comparison needs a reproducible forward: conv-FFN statistics on the separate deterministic pass (the
        monkeypatch.setattr(ops.config, "fused_frame_stats", False)   # atomics-accumulated default is covered by the dropout 0.1 case)
    z = load("step_tiny")
    cfg, meta = jload(z, "cfg"), jload(z, "meta")
    nstep = 5
    runs = {}
    for mode in ("eager", "graph"):
        ops.unregister_flat_slabs()
        ops.manual_seed(dev, 1234)
        torch.manual_seed(7)
        enc, dec, T = _make(pkg, cfg, meta, dev, dropout)
        tr = NARTrainer(enc, dec, T, batch_size=meta["N"], lr=1e-4, max_grad_norm=1.0, lam_pc=0.1)
        start = {k: v.detach().clone() for k, v in T.state_dict().items()}
        if mode == "graph":
            tr.capture(*_batch(meta, cfg, 0, dev), warmup=2)
            T.load_state_dict(start)                                   # the warm-up and capture passes stepped the model
            tr.opt.m.zero_(); tr.opt.v.zero_(); tr.opt.step_dev.zero_()
            if tr.opt.planes is not None:
                tr.opt.planes.refresh()
            ops.manual_seed(dev, 1234)
        recs = []
        for s in range(nstep):
            out = tr.step(*_batch(meta, cfg, s, dev))
            recs.append({k: float(v) for k, v in out.items()})       # a device -> host read per step ...
        for s in range(nstep, 2 * nstep):                               # ... and a run of replays with no read in between
            out = tr.step(*_batch(meta, cfg, s, dev))
        torch.cuda.synchronize()
        recs.append({k: float(v) for k, v in out.items()})
        runs[mode] = (recs, {k: v.detach().clone() for k, v in T.state_dict().items()})
    e, g = runs["eager"], runs["graph"]
    for re_, rg in zip(e[0], g[0]):
        for k in re_:
            assert rg[k] == rg[k] and abs(rg[k]) < 1e6, ("graph replay produced a non-finite / absurd value", k, rg[k])
            # identical masks; residual differences come from the order of fp32 atomics, amplified over the AdamW steps
            assert abs(re_[k] - rg[k]) <= (2e-4 if dropout == 0.0 else 2e-3) * abs(re_[k]) + 1e-6, (k, re_[k], rg[k])
    if True:
        num = den = 0.0
        for k, v in e[1].items():
            if v.is_floating_point() and "running" not in k:
                num += float((g[1][k].double() - v.double()).pow(2).sum())
                den += float(v.double().pow(2).sum())
        assert (num / den) ** 0.5 < 1e-5, (num / den) ** 0.5


def test_graph_k64_bench_config_droppath_on(dev):
    """the bench configuration itself (VPTRFormerNAR 4 + 8 layers, d 528, dropout 0.1, DropPath on, random init as bench.py builds it)
    at batch 4: 10 replays == 10 eager steps term by term, every loss term inside its mathematical range"""
    import bench
    from vptr_amd import ops
    from vptr_amd.train import NARTrainer
    ops.unregister_flat_slabs()
    ops.manual_seed(dev, 99)
    enc, dec, T = bench.build_models(dev, 0.1)
    n = 4
    tr = NARTrainer(enc, dec, T, batch_size=n, lr=1e-4, max_grad_norm=1.0, lam_pc=0.1)
    past, fut = bench.synth_batch(n, 0, dev)
    tr.capture(past, fut, warmup=2)
    assert "memset" not in tr.graph_nodes and tr.graph_nodes.get("kernel", 0) > 500, tr.graph_nodes
    ok, rep = tr.verify_graph(past, fut, steps=10, rtol=2e-3)
    assert ok, rep
    g = rep["graph_last"]
    assert 0.0 <= g["T_GDL"] <= 4.0 and 0.0 <= g["T_MSE"] <= 1.5 and 0.0 < g["T_bpc"] < 8.0 and g["grad_norm"] < 100.0, g
    del tr
    ops.unregister_flat_slabs()
    torch.cuda.empty_cache()


def test_graph_census_sees_memset_nodes(dev):
    """the guard itself: a captured hipMemsetAsync shows up as a memset node in graph_node_census (capture() raises on those)"""
    import ctypes
    from vptr_amd.train import graph_node_census
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    x = torch.ones(1024, device=dev)
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        x.add_(1.0)
        rc = hip.hipMemsetAsync(ctypes.c_void_p(x.data_ptr()), 0, 4096, ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(dev.index or 0)))
        x.mul_(2.0)
    assert rc == 0
    census = graph_node_census(g)
    assert census.get("memset") == 1 and census.get("kernel") == 2, census
    g.reset()
