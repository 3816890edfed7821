"""Stock `torch.nn.parallel.DistributedDataParallel` around this package's modules: the reference's data-parallel script wraps Enc,
Dec and the transformer in DDP and reaches the projector through `.module` (train_NAR_mp.py:94-118,132-189).  2 ranks sharing
cuda:0 over gloo (what a one-GPU box can run; RCCL takes the same calls) run that `single_iter` recipe for two iterations:

  DDP replica  ==  plain replica on the same shard whose gradients are averaged by hand before clipping
  (gradients after iteration 0, parameters after two iterations, identical on both ranks).

What it pins: every parameter's gradient reaches DDP's reducer through its AccumulateGrad hook -- inside a torch.distributed job
the weight-gradient kernels hand their results to autograd instead of accumulating into `.grad` behind the engine's back
(ops._loose_grad_for) -- including the projector used outside DDP's forward and the decoder's never-stepped gradients; a missed hook
shows up as DDP's "expected to have finished reduction" error in iteration 1.  SELF-comparison (noise-derived bounds,
tools/selfcmp_spread.py)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ["VPTR_FUSED_STATS"] = "0"      # reproducible forward: the comparison below is between two replicas on ONE shard
    import torch.distributed as dist
    import torch.nn.functional as F
    from torch.nn.parallel import DistributedDataParallel as DDP
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import vptr_amd.model as pkg
        from helpers import build_transformer, jload, load
        from oracle import fill
        from vptr_amd import ops
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        ops.unregister_flat_slabs()
        z = load("step_tiny")
        cfg, meta = jload(z, "cfg"), jload(z, "meta")
        G = 4
        per = G // world

        def make():
            enc = pkg.VPTREnc(1, meta["feat"], 3, "reflect")
            dec = pkg.VPTRDec(1, meta["feat"], 3, "Tanh", "reflect")
            T = build_transformer(pkg, cfg, False)
            fill.apply_fill(enc, meta["seed"]); fill.apply_fill(dec, meta["seed"] + 10); fill.apply_fill(T, meta["seed"] + 20)
            return enc.to(dev), dec.to(dev), T.to(dev)

        mse, gdl = pkg.MSELoss(), pkg.GDL(alpha=1)
        bpnce = pkg.BiPatchNCE(per, cfg["Tf"], cfg["H"], cfg["W"], 1.0).to(dev)        # train_NAR_mp.py:127: batch_size // world_size

        def single_iter(Enc, Dec, T, proj, opt, past, fut, sync_grads):
            """train_NAR_mp.py:132-167 (no discriminator); `proj` = VPTR_Transformer.module.NCE_projector"""
            with torch.no_grad():
                pf, ff = Enc(past), Enc(fut)
            T.train()
            T.zero_grad(set_to_none=True)
            Dec.zero_grad(set_to_none=True)
            pred_f = T(pf)
            pred = Dec(pred_f)
            a = proj(pred_f.permute(0, 1, 3, 4, 2)).permute(0, 1, 4, 2, 3)
            b = proj(ff.permute(0, 1, 3, 4, 2)).permute(0, 1, 4, 2, 3)
            loss = gdl(fut, pred) + mse(pred, fut) + 0.1 * bpnce(F.normalize(b, p=2.0, dim=2), F.normalize(a, p=2.0, dim=2))
            loss.backward()
            if sync_grads is not None:
                sync_grads()
            torch.nn.utils.clip_grad_norm_(T.parameters(), max_norm=1.0, norm_type=2)
            opt.step()
            return float(loss)

        # ---- A: the script's objects, DDP-wrapped
        encA, decA, TA = make()
        EncA, DecA = DDP(encA, device_ids=[0]).eval(), DDP(decA, device_ids=[0]).eval()
        optA = torch.optim.AdamW(TA.parameters(), lr=1e-4)
        TA = DDP(TA, device_ids=[0])
        # ---- B: plain modules, gradients averaged by hand
        encB, decB, TB = make()
        encB.eval(); decB.eval()
        optB = torch.optim.AdamW(TB.parameters(), lr=1e-4)

        def mean_grads():
            for p in list(TB.parameters()) + list(decB.parameters()):
                if p.grad is not None:
                    dist.all_reduce(p.grad)
                    p.grad.div_(world)

        def rel(a, b):
            return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))

        res = {"rank": rank, "loss": []}
        for s in range(2):
            past = fill.clip_input((G, cfg["Tp"], 1, meta["HW"], meta["HW"]), meta["seed"] + 100 + s, "kth")[rank * per:(rank + 1) * per].to(dev)
            fut = fill.clip_input((G, cfg["Tf"], 1, meta["HW"], meta["HW"]), meta["seed"] + 200 + s, "kth")[rank * per:(rank + 1) * per].to(dev)
            la = single_iter(EncA, DecA, TA, TA.module.NCE_projector, optA, past, fut, None)
            lb = single_iter(encB, decB, TB, TB.NCE_projector, optB, past, fut, mean_grads)
            res["loss"].append((la, lb))
            ga = torch.cat([p.grad.flatten() for p in TA.module.parameters()])
            gb = torch.cat([p.grad.flatten() for p in TB.parameters()])
            res["grad_rel_step%d" % s] = rel(ga, gb)
            if s == 0:
                res["dec_grad_rel"] = rel(torch.cat([p.grad.flatten() for p in decA.parameters()]),
                                          torch.cat([p.grad.flatten() for p in decB.parameters()]))
                res["none_grads"] = sum(1 for p in TA.module.parameters() if p.grad is None)
        pa = torch.cat([p.detach().flatten() for p in TA.module.parameters()])
        pb = torch.cat([p.detach().flatten() for p in TB.parameters()])
        res["param_rel"] = rel(pa, pb)
        res["param_digest"] = float(pa.double().sum())
        q.put(res)
        dist.barrier()
    except Exception as e:  # noqa
        import traceback
        q.put({"rank": rank, "error": "%s\n%s" % (e, traceback.format_exc()[-2500:])})
    finally:
        dist.destroy_process_group()


def test_stock_ddp_wrapped_modules_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    from helpers import collect
    res = sorted(collect(q, procs, world, 600), key=lambda r: r["rank"])
    for p in procs:
        p.join(120)
    for r in res:
        assert "error" not in r, r["error"]
    from helpers import margin
    for r in res:
        for k, b in (("grad_rel_step0", 1e-4), ("dec_grad_rel", 1e-4), ("grad_rel_step1", 5e-2), ("param_rel", 1e-4)):
            margin("ddp:%s:rank%d" % (k, r["rank"]), r[k], b)
        assert r["none_grads"] == 0, r
        # iteration 0: same shard, same parameters, same reduction -> the noise of one backward pass
        assert r["grad_rel_step0"] < 1e-4 and r["dec_grad_rel"] < 1e-4, r
        # iteration 1 starts from parameters that already differ by sign flips of the first AdamW update (~lr * sign(g)): loose
        assert r["grad_rel_step1"] < 5e-2, r
        assert r["param_rel"] < 1e-4, r
        for la, lb in r["loss"]:
            assert abs(la - lb) < 1e-3 * abs(lb), r
    assert res[0]["param_digest"] == res[1]["param_digest"], "DDP replicas diverged"
