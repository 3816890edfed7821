"""world_size-2 gloo tests (CPU) of the data-parallel path: parameter broadcast and bucketed gradient mean, checked
against the definition 'mean of the per-rank gradients' and against a single replica on the concatenated batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vptr_amd.parallel import allreduce_mean_, broadcast_module, shard_batch
    torch.manual_seed(100 + rank)  # different init per rank -> broadcast must fix it
    model = torch.nn.Sequential(torch.nn.Linear(12, 7), torch.nn.BatchNorm1d(7), torch.nn.Linear(7, 3))
    broadcast_module(model, 0)
    sd = torch.cat([v.double().reshape(-1) for v in model.state_dict().values()])
    # a tiny eval-mode model so that DP == single replica on the concatenated batch
    model.eval()
    g = torch.Generator().manual_seed(7)
    xs = torch.randn(8, 12, generator=g)
    ys = torch.randn(8, 3, generator=g)
    off, per = shard_batch(8, rank, world)
    loss = ((model(xs[off:off + per]) - ys[off:off + per]) ** 2).mean()
    loss.backward()
    params = [p for p in model.parameters()]
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    local = flat.clone()
    allreduce_mean_(flat, None, bucket_elems=17)  # several ragged buckets
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = torch.stack(gathered).mean(0)
    # single replica on the whole batch
    model.zero_grad()
    ((model(xs) - ys) ** 2).mean().backward()
    full = torch.cat([p.grad.reshape(-1) for p in params])
    # DDP's per-forward buffer broadcast as one flat collective: rank 0's BatchNorm statistics replace everybody's
    from vptr_amd.parallel import BufferSync
    bn = model[1]
    with torch.no_grad():
        bn.running_mean.fill_(float(rank + 1))
        bn.running_var.fill_(float(10 * (rank + 1)))
        bn.num_batches_tracked.fill_(7 + rank)
    bs = BufferSync(model, 0, None)
    assert bool(bs)
    bs.sync()
    assert float(bn.running_mean[0]) == 1.0 and float(bn.running_var[-1]) == 10.0 and int(bn.num_batches_tracked) == 7
    # the job-start broadcast of several modules as one message per dtype (bench.py's multi-rank start-up)
    from vptr_amd.parallel import broadcast_modules_flat
    torch.manual_seed(200 + rank)
    m1 = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.BatchNorm1d(4))
    m2 = torch.nn.Linear(4, 2)
    with torch.no_grad():
        m1[1].num_batches_tracked.fill_(3 + rank)
    broadcast_modules_flat([m1, m2], 0, None)
    flat_sd = torch.cat([v.double().reshape(-1) for m in (m1, m2) for v in m.state_dict().values()])
    both = [torch.zeros_like(flat_sd) for _ in range(world)]
    dist.all_gather(both, flat_sd)
    assert torch.equal(both[0], both[1]) and int(m1[1].num_batches_tracked) == 3
    q.put((rank, sd, float((flat - expect).abs().max()), float((flat - full).abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_broadcast_and_bucketed_mean_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert torch.equal(res[0][1], res[1][1]), "broadcast did not equalise the replicas"
    for _, _, e_mean, e_full in res:
        assert e_mean < 1e-6      # bucketed all-reduce == mean of per-rank gradients
        assert e_full < 1e-6      # == single replica on the concatenated batch (no train-mode BN)


def test_shard_batch_rejects_reference_bug():
    from vptr_amd.parallel import shard_batch
    assert shard_batch(64, 3, 4) == (48, 16)
    with pytest.raises(ValueError):
        shard_batch(2, 0, 4)  # train_NAR_mp.py:297,313 ships batch 2 on 4 ranks
