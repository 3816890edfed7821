"""The data-parallel train step on real kernels: 2 ranks sharing cuda:0, gradient exchange through gloo (what one-GPU boxes can
run; RCCL takes the same torch.distributed calls).  FARTrainer (no train-mode BatchNorm, so DP == one replica on the
concatenated batch):

  overlapped chunked exchange == plain exchange after backward == single replica on the global batch
  (gradient slab before the optimizer step and parameters after two steps)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stderr_path(port, rank):
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "vptr_dp_%d_rank%d.err" % (port, rank))


def _dead_worker_tails(port, world, procs):
    """last stderr lines of the workers that died without a Python exception (an abort inside a runtime library says why only there)"""
    out = []
    for r in range(world):
        if procs[r].exitcode not in (0, 1, None):
            try:
                out.append("rank %d (exit code %s) stderr tail:\n%s" % (r, procs[r].exitcode, "".join(open(_stderr_path(port, r)).readlines()[-30:])))
            except OSError:
                pass
    return "\n".join(out)


def _worker(rank, world, port, q, G=4, full=True):
    fd = os.open(_stderr_path(port, rank), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)   # this rank's stderr (runtime aborts included) to a file
    os.dup2(fd, 2)
    if world > 2:
        # eight processes on ONE GPU oversubscribe its hardware queues; the scheduler then time-slices them with wave save / restore, under which
        # a queue intermittently aborts with HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION (4 of 8 runs, profiles/r06_dp8_repeat.log) -- an artefact of
        # sharing the device that one-process-per-GPU jobs do not have.  One hardware queue per process keeps the total under the limit.
        if os.environ.get("VPTR_TEST_KEEP_HW_QUEUES") != "1":   # (tools/r06_lease19.sh measures the failure rate without it)
            os.environ["GPU_MAX_HW_QUEUES"] = "1"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank)
    # The comparisons below are at 1e-5, i.e. they need a bit-reproducible forward pass: keep the conv-FFN statistics on the separate
    # deterministic pass.  (Accumulated by atomics in the producers' epilogues -- the default -- they are reproducible to ~1e-7 only,
    # and this tiny random-filled fixture turns a 1e-7 input perturbation into a 3e-4 gradient change: tools/ffn_stats_sensitivity.py.)
    os.environ["VPTR_FUSED_STATS"] = "0"
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_per_process_memory_fraction(min(1.0, 0.9 / world), 0) if world > 2 else None   # a runaway rank fails itself, not its peers
        import vptr_amd.model as pkg
        from helpers import build_transformer, jload, load
        from oracle import fill
        from vptr_amd import ops
        from vptr_amd.parallel import broadcast_module, shard_batch
        from vptr_amd.train import FARTrainer
        dev = torch.device("cuda:0")
        z = load("step_far_tiny")
        cfg, meta = jload(z, "cfg"), jload(z, "meta")
        # G = global batch (4: 2 per rank at world 2; 8: 1 per rank at world 8)

        def make(seed_shift):
            enc = pkg.VPTREnc(1, meta["feat"], 3, "reflect")
            dec = pkg.VPTRDec(1, meta["feat"], 3, meta["out_layer"], "reflect")
            T = build_transformer(pkg, cfg, True)
            fill.apply_fill(enc, meta["seed"])
            fill.apply_fill(dec, meta["seed"] + 10)
            fill.apply_fill(T, meta["seed"] + 20 + seed_shift)     # rank-dependent init: the broadcast must equalise it
            return enc.to(dev), dec.to(dev), T.to(dev)

        def batch(s):
            past = fill.rand_input((G, cfg["Tp"], 1, meta["HW"], meta["HW"]), meta["seed"] + 100 + s).to(dev)
            fut = fill.rand_input((G, cfg["Tf"], 1, meta["HW"], meta["HW"]), meta["seed"] + 200 + s).to(dev)
            return past, fut

        def run(parallel, overlap, front=False):
            os.environ["VPTR_DP_OVERLAP"] = "1" if overlap else "0"
            ops.unregister_flat_slabs()
            enc, dec, T = make(rank if parallel else 0)
            if parallel:
                broadcast_module(T, 0)
            tr = FARTrainer(enc, dec, T, lr=1e-4, max_grad_norm=1.0, process_group=dist.group.WORLD if parallel else None)
            off, per = shard_batch(G, rank, world) if parallel else (0, G)
            if front:   # forward + backward as one hipGraph, exchange + optimizer eager (the multi-rank launch mode of bench.py)
                snap = tr._snapshot()
                p0, f0 = batch(0)
                tr.capture_front(p0[off:off + per], f0[off:off + per], warmup=1)
                tr._restore(snap)
            grads = []
            for s in range(2):
                past, fut = batch(s)
                out = tr.step(past[off:off + per], fut[off:off + per])
                grads.append(tr.opt.grad.detach().clone() * tr._grad_scale)   # the mean's 1 / world lives in the optimizer kernel
            return grads, tr.opt.flat.detach().clone(), float(out["grad_norm"])

        g_ov, p_ov, n_ov = run(True, True)
        g_1, p_1, n_1 = run(False, False)
        if full:
            g_pl, p_pl, n_pl = run(True, False)
            g_fr, p_fr, n_fr = run(True, True, front=True)
        else:   # the 8-rank test: chunked overlapped exchange and the front-graph step against ONE replica on the global batch
            g_fr, p_fr, n_fr = run(True, True, front=True)
            g_pl, p_pl, n_pl = g_ov, p_ov, n_ov
        offs = [shard_batch(G, r, world) for r in range(world)]
        assert offs[0][0] == 0 and all(offs[i][0] + offs[i][1] == offs[i + 1][0] for i in range(world - 1)) and offs[-1][0] + offs[-1][1] == G

        def rel(a, b):
            return float((a.double() - b.double()).norm() / b.double().norm())
        res = {"rank": rank,
               # step 0 is the clean comparison; step 1 starts from parameters that differ by sign flips of the first AdamW update
               # (~lr * sign(g)) wherever fp32 atomics reordered a near-zero gradient: ~5e-5 even between two runs of one mode
               "grad_overlap_vs_plain": rel(g_ov[0], g_pl[0]),
               "grad_dp_vs_single": rel(g_pl[0], g_1[0]),
               "grad_step1": max(rel(g_ov[1], g_pl[1]), rel(g_pl[1], g_1[1])),
               "param_overlap_vs_plain": rel(p_ov, p_pl), "param_dp_vs_single": rel(p_pl, p_1),
               "grad_front_vs_plain": rel(g_fr[0], g_pl[0]), "param_front_vs_plain": rel(p_fr, p_pl),
               "norms": (n_ov, n_pl, n_1), "param_digest": float(p_ov.double().sum())}
        q.put(res)
        dist.barrier()
    except Exception as e:  # noqa: the FIRST failing rank names its own error (its peers only see "connection closed by peer")
        import traceback
        q.put({"rank": rank, "error": "%r\n%s" % (e, traceback.format_exc()[-1500:])})
        raise
    finally:
        dist.destroy_process_group()


def test_dp_two_ranks_on_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    from helpers import collect
    res = sorted(collect(q, procs, world, 600), key=lambda r: r["rank"])
    errs = sorted((r for r in res if "error" in r), key=lambda r: "closed by peer" in r["error"])   # the cause before its echoes
    assert not errs, "rank %d failed first: %s" % (errs[0]["rank"], errs[0]["error"])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    from helpers import margin
    for r in res:
        for k, b in (("grad_overlap_vs_plain", 1e-5), ("grad_dp_vs_single", 1e-4), ("grad_step1", 5e-3), ("param_overlap_vs_plain", 1e-5),
                     ("param_dp_vs_single", 1e-5), ("grad_front_vs_plain", 1e-5), ("param_front_vs_plain", 1e-5)):
            margin("dp:%s:rank%d" % (k, r["rank"]), r[k], b)
        assert r["grad_overlap_vs_plain"] < 1e-5, r
        assert r["grad_dp_vs_single"] < 1e-4, r          # fp32 atomics + a different reduction order over the batch
        assert r["grad_step1"] < 5e-3, r          # second step, after sign-flipped first AdamW updates: measured 1.1e-4 (profiles/r04_margins.log)
        assert r["param_overlap_vs_plain"] < 1e-5, r
        assert r["param_dp_vs_single"] < 1e-5, r
        assert r["grad_front_vs_plain"] < 1e-5 and r["param_front_vs_plain"] < 1e-5, r     # front-graph step == eager step
        assert abs(r["norms"][0] - r["norms"][2]) < 1e-3 * r["norms"][2], r
    assert res[0]["param_digest"] == res[1]["param_digest"], "replicas diverged"


def test_dp_eight_ranks_on_one_gpu():
    """world 8 (the node the scaling bench runs on), one clip per rank, all ranks on cuda:0 over gloo: rank logic, shard_batch, the
    DP_CHUNKS boundaries, the rank-0 broadcast and the front-graph step at the real world size -- chunked overlapped exchange == one
    replica on the 8-clip batch.  Sharing one GPU between nine processes is what this test adds to the product's
    one-process-per-GPU contract: a run in which the RUNTIME aborts a rank's queue (HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION under time-slicing,
    see _worker) is repeated, and three such runs in a row are reported as xfail with the runtime's message, not as a product failure."""
    import gc
    from helpers import collect, margin
    world, res, tails = 8, None, ""
    for attempt in range(3):
        port = _free_port()
        gc.collect()
        torch.cuda.empty_cache()     # eight more contexts share this GPU with the pytest process: hand its cached blocks back first
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q, 8, False)) for r in range(world)]
        for p in procs:
            p.start()
        import queue
        try:
            res = sorted(collect(q, procs, world, 180), key=lambda r: r["rank"])    # (a healthy run takes 25 s)
        except (RuntimeError, queue.Empty) as e:
            for p in procs:
                p.join(5)
                if p.is_alive():
                    p.kill()
            tails = _dead_worker_tails(port, world, procs)
            # the runtime killed a queue of a time-sliced process, or the ranks that lost a peer that way never came back: not a statement about
            # the exchange path (the 2-rank test above drives the same code without sharing queues)
            if "HSA_STATUS_ERROR" in tails or isinstance(e, queue.Empty):
                tails = tails or "no report within 180 s (workers killed)"
                print("attempt %d: %s" % (attempt, tails[-600:]))
                res = None
                continue
            raise RuntimeError("%s\n%s" % (e, tails))
        errs = sorted((r for r in res if "error" in r), key=lambda r: "closed by peer" in r["error"])   # the cause before its echoes
        assert not errs, "rank %d failed first: %s" % (errs[0]["rank"], errs[0]["error"])
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        break
    if res is None:
        pytest.xfail("three runs in a row lost a rank to a runtime queue abort while nine processes shared the GPU: " + tails[-400:])
    assert [r["rank"] for r in res] == list(range(world))
    for r in res:
        for k, b in (("grad_dp_vs_single", 2e-4), ("param_dp_vs_single", 1e-5), ("grad_front_vs_plain", 1e-5), ("param_front_vs_plain", 1e-5)):
            margin("dp8:%s:rank%d" % (k, r["rank"]), r[k], b)
        assert r["grad_dp_vs_single"] < 2e-4, r          # fp32 atomics + an 8-way reduction order over the batch
        assert r["param_dp_vs_single"] < 1e-5, r
        assert r["grad_front_vs_plain"] < 1e-5 and r["param_front_vs_plain"] < 1e-5, r
        assert abs(r["norms"][0] - r["norms"][2]) < 1e-3 * r["norms"][2], r
    assert len({r["param_digest"] for r in res}) == 1, "replicas diverged"
