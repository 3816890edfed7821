"""Data-parallel plumbing for the VPTR train step: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on the MI355X node; "gloo" for the CPU tests).

The reference wraps its modules in DistributedDataParallel over NCCL (train_NAR_mp.py:94-118, 191-198).  What that
does for this model is (a) a one-time broadcast of rank 0's parameters and buffers, (b) an all-reduce (mean) of the
transformer's gradients every step.  Here (b) runs on the flat fp32 gradient slab of `FlatAdamW` in a few large
buckets -- xGMI is point-to-point (7 links x ~153 GB/s per GPU), so collectives are per-link bound and large
messages amortise the ring latency; 118 M fp32 gradients = 473.5 MB go out as ceil(473.5 / bucket_mb) calls instead
of DDP's ~20 x 25 MB buckets over 664 tensors.  The decoder/encoder are frozen in stage 2 and are not reduced (the
reference reduces decoder gradients nobody steps).  BatchNorm batch statistics stay per rank, like the reference (no
SyncBN); running statistics are kept identical by broadcasting rank 0's buffers on demand (`broadcast_buffers`).
"""
import torch
import torch.distributed as dist


def _broadcast_any(t, src, group):
    """dist.broadcast for tensors that may be non-contiguous views (FlatAdamW exposes channel-last stored parameters as permuted
    views of its slab; NCCL and gloo reject those): broadcast a contiguous temporary and copy it back."""
    if t.is_contiguous():
        dist.broadcast(t, src, group=group)
    else:
        tmp = t.contiguous()
        dist.broadcast(tmp, src, group=group)
        t.copy_(tmp)
    _planes_dirty()


def _planes_dirty():
    """c10d collectives write tensors without bumping autograd's version counters: cached P16 weight images must be rebuilt"""
    try:
        from . import ops
    except ImportError:      # CPU-only host (gloo tests of this module): no HIP library, no planes
        return
    ops.invalidate_weight_planes()


def broadcast_module(module, src=0, group=None):
    """Make every rank hold rank `src`'s parameters and buffers (DDP constructor semantics)."""
    with torch.no_grad():
        for t in module.state_dict().values():
            if t.is_floating_point() or t.dtype in (torch.int64, torch.int32):
                _broadcast_any(t, src, group)


def broadcast_modules_flat(modules, src=0, group=None):
    """`broadcast_module` for several modules as ONE collective per dtype: every tensor of the state_dicts is packed into a flat
    buffer, broadcast, and copied back (the K64 models are ~800 tensors: one 0.6 GB fp32 message and one small int64 message
    instead of ~800 ring start-ups at job start)."""
    with torch.no_grad():
        by_dtype = {}
        for m in modules:
            for t in m.state_dict().values():
                if t.is_floating_point() or t.dtype in (torch.int64, torch.int32):
                    by_dtype.setdefault(t.dtype, []).append(t)
        for dt, ts in by_dtype.items():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src, group=group)
            off = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view(t.shape))
                off += n
    _planes_dirty()


def broadcast_buffers(module, src=0, group=None):
    """DDP's per-forward buffer broadcast (BatchNorm running statistics of the NAR-encoder conv-FFNs)."""
    with torch.no_grad():
        for b in module.buffers():
            _broadcast_any(b, src, group)


class BufferSync:
    """DDP's per-forward buffer broadcast (`broadcast_buffers=True`, the default the reference relies on: train_NAR_mp.py:94-95,118)
    for the only buffers that change during training -- the BatchNorm running statistics of the NAR-encoder conv-FFNs -- as ONE
    flat collective per step instead of one per buffer: rank `src`'s running_mean / running_var / num_batches_tracked replace
    everybody's before the forward pass, so evaluation and checkpoints agree across ranks."""

    def __init__(self, module, src=0, group=None):
        self.src, self.group = src, group
        self.bufs = []
        for m in module.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.track_running_stats:
                self.bufs += [m.running_mean, m.running_var, m.num_batches_tracked]

    def __bool__(self):
        return bool(self.bufs)

    def sync(self):
        if not self.bufs or dist.get_world_size(self.group) == 1:
            return
        with torch.no_grad():
            flat = torch.cat([b.reshape(-1).to(torch.float32) for b in self.bufs])   # num_batches_tracked: exact in fp32 below 2^24
            dist.broadcast(flat, self.src, group=self.group)
            off = 0
            for b in self.bufs:
                n = b.numel()
                b.copy_(flat[off:off + n].view(b.shape))
                off += n


def allreduce_mean_(flat, group=None, bucket_elems=16 << 20, force=False):
    """In-place mean over ranks of a flat tensor, in buckets of `bucket_elems` elements (64 MB fp32 by default).  force: issue the
    collectives on a one-rank group too (functional runs of the RCCL path on a one-GPU box)."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return flat
    n = flat.numel()
    for off in range(0, n, bucket_elems):
        dist.all_reduce(flat[off:off + bucket_elems], op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / world)
    return flat


def allreduce_sum_(flat, group=None, bucket_elems=16 << 20):
    """In-place SUM over ranks of a flat tensor in buckets of `bucket_elems` elements; issued on one-rank groups too.  The trainers fold
    the 1 / world of the mean into the optimizer kernel's gradient scale instead of running another pass over the slab."""
    n = flat.numel()
    for off in range(0, n, bucket_elems):
        dist.all_reduce(flat[off:off + bucket_elems], op=dist.ReduceOp.SUM, group=group)
    return flat


def shard_batch(global_batch, rank, world):
    """The reference's DistributedSampler split: per-rank batch = global // world (utils/dataset.py:71-77).  Raises on
    the degenerate configuration the reference script ships (batch 2 on 4 ranks -> 0 per rank, train_NAR_mp.py:297,313)."""
    per = global_batch // world
    if per < 1:
        raise ValueError("global batch %d cannot be split over %d ranks" % (global_batch, world))
    return rank * per, per
