"""Training-step recipes of VPTR on the MI355X path.

`NARTrainer.step` reproduces `single_iter` of the reference's stage-2 NAR trainer (train_NAR.py:49-107 and its
data-parallel twin train_NAR_mp.py:132-189): two no-grad encoder passes, transformer + decoder forward,
MSE + GDL + lam_pc * BiPatchNCE, backward, clip_grad_norm_(max_norm) over the transformer parameters, AdamW.

MI355X-first differences in *how* (not *what*):
  * parameters / gradients / Adam moments of the transformer live in three flat fp32 slabs, so that global-norm
    clipping + AdamW are two HIP launches (vptr_sumsq, vptr_adamw) and the data-parallel gradient exchange is a
    handful of large RCCL all-reduces over xGMI instead of 664 small tensors;
  * losses are returned as device tensors (the reference's 8 `.item()` syncs per step are left to the caller);
  * the weight (and bias) gradients of every nn.Linear of a backward pass run as ONE grouped MFMA launch at its end
    (ops.defer_wgrad / vptr_gemm_grouped);
  * `capture()` turns the whole step into one hipGraph (single GPU; bench.py's default launch mode).

`FARTrainer` is the same for `single_iter` of train_FAR.py:48-101, `AETrainer` for the stage-1 auto-encoder + PatchGAN step of
train_AutoEncoder.py:44-78; both NAR and FAR trainers take the optional adversarial branch (`disc=`, `lam_gan=`).
"""
import os
import time

import torch
import torch.nn.functional as F

from . import ops
from ._lib import check, lib, ptr, stream
from .model.criterion import GANLoss


_HIP_NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record",
                   10: "mem_alloc", 11: "mem_free"}


def graph_node_census(g):
    """{node type: count} of a torch.cuda.CUDAGraph captured with keep_graph=True (hipGraphGetNodes / hipGraphNodeGetType on the
    runtime torch itself loaded)"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    graph = ctypes.c_void_p(int(g.raw_cuda_graph()))
    n = ctypes.c_size_t(0)
    if hip.hipGraphGetNodes(graph, None, ctypes.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    nodes = (ctypes.c_void_p * max(n.value, 1))()
    if n.value and hip.hipGraphGetNodes(graph, nodes, ctypes.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    out = {}
    for i in range(n.value):
        t = ctypes.c_int(-1)
        if hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) != 0:
            raise RuntimeError("hipGraphNodeGetType failed")
        name = _HIP_NODE_TYPES.get(t.value, "type%d" % t.value)
        out[name] = out.get(name, 0) + 1
    return out


DP_CHUNKS = 4  # grouped weight-gradient launches per step when gradients are exchanged between ranks


class FlatAdamW:
    """torch.optim.AdamW semantics (lr, betas, eps, decoupled weight decay, bias correction) on one flat slab, with
    clip_grad_norm_ folded in.  Every parameter must receive a gradient each step (true for the VPTR transformers)."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None, channel_last=()):
        """channel_last: ids of [C, ...] parameters that the kernels consume channel-last (the (C,H,W) LayerNorm affines and
        the depthwise 3x3 weights of MlpDWBN): they are STORED channel-last in the slab and exposed as a permuted view of
        their usual shape, so no per-step transposes of the parameter or its gradient remain (AdamW is elementwise, the
        order inside the slab is irrelevant; state_dict / load_state_dict go through strided copies)."""
        self.params = [p for p in params if p.requires_grad]
        channel_last = set(channel_last)
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        pad = (-total) % 4
        self.flat = torch.zeros(total + pad, device=dev, dtype=torch.float32)
        self.grad = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        off = 0
        self._layout = []  # per parameter: (offset, numel, None | (channel-last storage shape, permutation back to the logical shape))
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                if id(p) in channel_last and p.dim() >= 2:
                    perm = list(range(1, p.dim())) + [0]                  # storage order: trailing dims, then channels
                    inv = [p.dim() - 1] + list(range(p.dim() - 1))
                    shape_cl = [p.shape[i] for i in perm]
                    val = p.detach().clone()
                    p.data = self.flat[off:off + n].view(shape_cl).permute(inv)
                    p.data.copy_(val)
                    p.grad = self.grad[off:off + n].view(shape_cl).permute(inv)
                    self._layout.append((off, n, (shape_cl, inv)))
                else:
                    self._layout.append((off, n, None))
                    self.flat[off:off + n].copy_(p.detach().reshape(-1))
                    p.data = self.flat[off:off + n].view(p.shape)
                    p.grad = self.grad[off:off + n].view(p.shape)
                off += n
        self.total = total
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.float32)
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float32)
        ops.register_flat_slab(self.flat, self.grad)  # backward kernels accumulate weight gradients in place
        # P16 planes of every nn.Linear-shaped weight of the slab (forward operand and transposed input-gradient operand), rebuilt
        # by one launch after each step: the GEMMs then stage weights by DMA instead of splitting fp32 per tile (ops.WeightPlanes)
        self.planes = None
        lin = ops.linear_weights_of(self.params) if ops.config.use_p16 else []
        if lin:
            self.planes = ops.WeightPlanes(lin)
            ops.register_weight_planes(self.planes)

    # -- torch.optim.AdamW-compatible state (checkpoints of the reference carry `optimizer_T.state_dict()`) ------------------
    def _logical(self, slab, i):
        """view of parameter i's range of `slab` (m, v, ...) in the parameter's LOGICAL shape: channel-last stored parameters
        come back through the same permuted view their .data uses, so state crosses the torch.optim format unscrambled"""
        off, n, cl = self._layout[i]
        if cl is None:
            return slab[off:off + n].view(self.params[i].shape)
        return slab[off:off + n].view(cl[0]).permute(cl[1])

    def state_dict(self):
        state = {}
        step = self.step_dev.detach().clone().reshape(())
        for i in range(len(self.params)):
            state[i] = {"step": step.clone(), "exp_avg": self._logical(self.m, i).contiguous().clone(),
                        "exp_avg_sq": self._logical(self.v, i).contiguous().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(self.params):
            raise ValueError("FlatAdamW.load_state_dict: %d parameters in the checkpoint, %d here" % (len(ids), len(self.params)))
        g0 = groups[0]
        self.lr, self.betas, self.eps, self.weight_decay = g0["lr"], tuple(g0["betas"]), g0["eps"], g0["weight_decay"]
        step = None
        with torch.no_grad():
            for i, (pid, p) in enumerate(zip(ids, self.params)):
                st = sd["state"].get(pid)
                if st is None:  # parameter never stepped
                    self._logical(self.m, i).zero_()
                    self._logical(self.v, i).zero_()
                else:
                    if tuple(st["exp_avg"].shape) != tuple(p.shape):
                        raise ValueError("FlatAdamW.load_state_dict: parameter %d has shape %s, its state %s"
                                         % (i, tuple(p.shape), tuple(st["exp_avg"].shape)))
                    self._logical(self.m, i).copy_(st["exp_avg"])
                    self._logical(self.v, i).copy_(st["exp_avg_sq"])
                    s_i = float(st["step"])
                    if step is not None and s_i != step:
                        raise ValueError("FlatAdamW.load_state_dict: per-parameter step counts differ (%g vs %g)" % (s_i, step))
                    step = s_i
            self.step_dev.fill_(0.0 if step is None else step)
        ops.invalidate_weight_planes()   # a checkpoint load usually rewrites the parameters too (possibly through .data)

    def close(self):
        """drop this optimizer's process-wide registrations (gradient-slab lookup, weight planes); also runs when it is collected"""
        if getattr(self, "flat", None) is not None:
            ops.unregister_flat_slab(self.flat)
        self.planes = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: interpreter shutdown
            pass

    def zero_grad(self):
        p0 = self.params[0]
        if p0.grad is None or p0.grad.data_ptr() != self.grad.data_ptr():
            raise RuntimeError("FlatAdamW: a parameter lost its flat gradient view (do not use set_to_none=True)")
        ops.discard_wgrads()  # leftovers of a backward pass that raised must not leak into this step
        self.grad.zero_()

    def grad_norm(self):
        """global L2 norm of the (scaled) gradients of the last step (device tensor), as clip_grad_norm_ returns"""
        return self.sumsq.sqrt() * self._last_scale

    _last_scale = 1.0
    _sumsq_ws = None

    def step(self, grad_scale=1.0):
        """grad_scale multiplies every gradient before clipping (e.g. 1 / accumulation steps)"""
        n = self.flat.numel()
        self._last_scale = float(grad_scale)
        if ops.config.deterministic:   # fixed-order two-pass sum (no atomics): the clip coefficient is reproducible bit for bit
            if self._sumsq_ws is None:
                self._sumsq_ws = torch.empty(1024, device=self.flat.device, dtype=torch.float32)
            check(lib.vptr_sumsq_ws(ptr(self.grad), n, ptr(self.sumsq), ptr(self._sumsq_ws), self._sumsq_ws.numel(), stream()), "vptr_sumsq_ws")
        else:
            self.sumsq.zero_()
            check(lib.vptr_sumsq(ptr(self.grad), n, ptr(self.sumsq), stream()), "vptr_sumsq")
        self.step_dev.add_(1.0)
        prof = ops.profiling.opt
        if prof is not None:   # bench.py's HBM roofline pass: HIP events on the launch stream around the optimizer's streaming kernels
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
        check(lib.vptr_adamw(ptr(self.flat), ptr(self.grad), ptr(self.m), ptr(self.v), n, self.lr, self.betas[0], self.betas[1],
                             self.eps, self.weight_decay, ptr(self.step_dev),
                             ptr(self.sumsq) if self.max_grad_norm is not None else None,
                             float(self.max_grad_norm or 0.0), float(grad_scale), stream()), "vptr_adamw")
        if prof is not None:
            ev[1].record()
        if self.planes is not None:
            self.planes.refresh()
        if prof is not None:
            ev[2].record()
            prof.append((n, self.planes.wp.numel() if self.planes is not None else 0, ev))


def _channel_last_ids(transformer):
    """parameters of the conv-FFNs that the HIP kernels read channel-last: LayerNorm((C,H,W)) affines and depthwise 3x3 weights"""
    from .model.vidhrformer import MlpDWBN
    ids = []
    for m in transformer.modules():
        if isinstance(m, MlpDWBN):
            ids.append(id(m.dw3x3.weight))
            if m.layer_norm:
                for norm in (m.norm1, m.norm2, m.norm3):
                    ids += [id(norm.weight), id(norm.bias)]
    return ids


class NARTrainer:
    """One stage-2 NAR training step; see module docstring.  `enc`/`dec` are frozen (eval), `transformer` trains."""

    def __init__(self, enc, dec, transformer, batch_size, lr=1e-4, max_grad_norm=1.0, lam_pc=0.1, process_group=None,
                 bucket_mb=64, dec_weight_grads=True, disc=None, lam_gan=None, gan_mode="vanilla"):
        self.enc, self.dec, self.T = enc.eval(), dec.eval(), transformer
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.bucket_elems = bucket_mb * (1 << 20) // 4
        self._init_gan(disc, lam_gan, lr, gan_mode)
        # Stage 2 optimises the transformer only (train_NAR.py:205).  The reference nevertheless leaves the decoder's
        # parameters trainable (:190-191), so its backward computes decoder weight gradients nobody consumes; that work
        # is reproduced by default (dec_weight_grads=True) so that the measured step does everything the reference's does.
        for p in enc.parameters():
            p.requires_grad_(False)  # the encoder runs under no_grad (:54-56)
        for p in dec.parameters():
            p.requires_grad_(bool(dec_weight_grads))
        self.dec_weight_grads = bool(dec_weight_grads)
        for mod in list(self.enc.modules()) + list(self.dec.modules()):
            mod._vptr_frozen = True  # never stepped here: packed conv weights / eval-BN folds are cached (ops.frozen_weights)
        self.opt = FlatAdamW(self.T.parameters(), lr=lr, max_grad_norm=max_grad_norm, channel_last=_channel_last_ids(self.T))
        dev = self.opt.flat.device
        # loss_name_list MSE / GDL(alpha=1) / BiPatchNCE(N, Tf, 8, 8, temperature 1.0) of train_NAR.py:176-179 run as the fused kernels of
        # csrc/losses.hip (ops.mse_gdl, ops.nce_loss): plain kernel launches only, so the step can live in a hipGraph
        self.nce_temperature = 1.0
        self.lam_pc = lam_pc
        self._bufsync = None
        if process_group is not None and torch.distributed.get_world_size(process_group) > 1:
            from .parallel import BufferSync
            self._bufsync = BufferSync(self.T, 0, process_group)   # DDP's per-forward broadcast of rank 0's BatchNorm statistics
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.bucket_elems = bucket_mb * (1 << 20) // 4
        self._graph = None

    _front = None          # data-parallel front graph (capture_front)
    _front_items = ()

    # -- optional adversarial branch (train_NAR.py:22-30,37-41,66-79; train_FAR.py:22-45,66-80): off in the reference scripts ----
    def _init_gan(self, disc, lam_gan, lr, gan_mode):
        self.disc, self.lam_gan = disc, lam_gan
        if disc is None:
            return
        if lam_gan is None:
            raise ValueError("a discriminator needs lam_gan (train_NAR.py:67)")
        for p in disc.parameters():
            p.requires_grad_(True)
        self.opt_D = FlatAdamW(list(disc.parameters()), lr=lr, betas=(0.5, 0.999), weight_decay=0.0)  # Adam, :204
        self.gan = GANLoss(gan_mode, target_real_label=1.0, target_fake_label=0.0).to(self.opt_D.flat.device)

    def _disc_update(self, fake, real):
        """cal_lossD + optimizer_D.step(), then freeze the discriminator for the generator pass"""
        if not self.disc.training:
            self.disc.train()
        for p in self.disc.parameters():
            p.requires_grad_(True)
        self.opt_D.zero_grad()
        l_fake = self.gan(self.disc(fake.detach().flatten(0, 1)), False)
        l_real = self.gan(self.disc(real.flatten(0, 1)), True)
        loss_D = (l_fake + l_real) * 0.5 * self.lam_gan
        loss_D.backward()
        if self.pg is not None and self.world > 1:   # the reference's DDP-wrapped discriminator averages its gradients too
            ops.flush_wgrads()
            from .parallel import allreduce_mean_
            allreduce_mean_(self.opt_D.grad, self.pg, self.bucket_elems)
        self.opt_D.step()
        for p in self.disc.parameters():
            p.requires_grad_(False)
        return {"Dtotal": loss_D.detach(), "Dfake": l_fake.detach(), "Dreal": l_real.detach()}

    # -- data-parallel gradient exchange: a few large RCCL all-reduces on the flat gradient slab ---------------------
    _grad_scale = 1.0   # factor the next optimizer step applies to the gradient slab (1 / world after a summed exchange)

    comm_stats = None   # set to a dict by bench.py: per-step exchange accounting (bytes, calls, device-side exposed time, host wait time)

    def _comm_begin(self):
        st = self.comm_stats
        if st is None:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()      # on the launch stream, behind the last kernel that produces gradients
        return (e0, time.perf_counter())

    def _comm_end(self, tok, nbytes, calls):
        """e0 -> e1 on the launch stream brackets nothing but the wait for the collectives: the time the stream idles there is the
        communication the step could not hide behind compute (0 when every all-reduce finished under the weight-gradient chunks)"""
        if tok is None:
            return
        st = self.comm_stats
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        st.setdefault("events", []).append((tok[0], e1))
        st["wait_host_s"] = st.get("wait_host_s", 0.0) + time.perf_counter() - tok[1]
        st["bytes"] = st.get("bytes", 0) + int(nbytes)
        st["calls"] = st.get("calls", 0) + int(calls)
        st["steps"] = st.get("steps", 0) + 1

    def _allreduce_grads(self):
        if self.pg is None or (self.world == 1 and os.environ.get("VPTR_DP_FORCE_EXCHANGE") != "1"):
            return
        from .parallel import allreduce_sum_
        tok = self._comm_begin()
        allreduce_sum_(self.opt.grad, self.pg, self.bucket_elems)
        self._comm_end(tok, self.opt.grad.numel() * 4, -(-self.opt.grad.numel() // self.bucket_elems))
        self._grad_scale = 1.0 / self.world

    def _backward_and_exchange(self, loss):
        """loss.backward(), the grouped weight-gradient GEMMs and the data-parallel gradient exchange.  With more than one
        rank the weight gradients are flushed in DP_CHUNKS grouped launches ordered by slab address; as soon as a chunk has
        been enqueued, the slab range below the next chunk's first destination is final and its all-reduce is issued
        asynchronously (RCCL runs it on its own stream), overlapping the next chunk's GEMM."""
        # VPTR_DP_FORCE_EXCHANGE=1: take the exchange path on a one-rank group too (a one-GPU box can then drive c10d's RCCL backend
        # -- its own stream, async Work objects, event ordering against the weight-gradient launches -- through the same code the
        # 8-GPU job runs: tests/test_22_rccl_gpu.py, bench.py --force-exchange)
        force = self.pg is not None and os.environ.get("VPTR_DP_FORCE_EXCHANGE") == "1"
        self._grad_scale = 1.0
        if self.pg is None or (self.world == 1 and not force) or os.environ.get("VPTR_DP_OVERLAP", "1") == "0":
            loss.backward()
            ops.flush_wgrads()  # no-op: the grouped weight-gradient launch already ran at the end of backward
            self._allreduce_grads()
            return
        with ops.hold_wgrads():
            loss.backward()
        self._exchange_held_wgrads()

    def _exchange_held_wgrads(self):
        """the recorded weight gradients as DP_CHUNKS grouped launches with the slab ranges they complete sent out in between (see
        `_backward_and_exchange`); also the eager tail of a front-graph step (`capture_front`)"""
        grad = self.opt.grad
        base, n = grad.data_ptr(), grad.numel()
        works, sent = [], [0]
        # bench.py's tuning table (comm_stats["timeline"] = True): HIP events on the launch stream at the start of the exchange, behind every
        # weight-gradient chunk (= the moment its all-reduces may start: c10d orders them behind this point of the stream) and behind every
        # Work.wait() (= the moment that all-reduce had finished, or the stream reached the wait -- whichever is later)
        tl = None
        if self.comm_stats is not None and self.comm_stats.get("timeline"):
            tl = {"t0": torch.cuda.Event(enable_timing=True), "chunk_end": [], "sent_bytes": [], "ar_done": [], "ar_bytes": []}
            tl["t0"].record()

        def send_upto(next_ptr):
            hi = n if next_ptr is None else max(sent[0], min(n, (next_ptr - base) // 4))
            off = sent[0]
            if tl is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                tl["chunk_end"].append(e)
                tl["sent_bytes"].append((hi - off) * 4)
            while off < hi:  # sub-ranges of at most bucket_elems, like the non-overlapped path
                end = min(hi, off + self.bucket_elems)
                works.append(torch.distributed.all_reduce(grad[off:end], op=torch.distributed.ReduceOp.SUM, group=self.pg,
                                                          async_op=True))
                if tl is not None:
                    tl["ar_bytes"].append((end - off) * 4)
                off = end
            sent[0] = hi

        ops.flush_wgrads(chunks=DP_CHUNKS, on_chunk=send_upto)
        if sent[0] < n:
            send_upto(None)
        tok = self._comm_begin()
        for w in works:
            w.wait()
            if tl is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                tl["ar_done"].append(e)
        if tl is not None:
            self.comm_stats["last_timeline"] = tl     # of the most recent step; bench.py resolves the events after its final synchronize
        self._comm_end(tok, n * 4, len(works))
        self._grad_scale = 1.0 / self.world   # the mean over ranks is folded into the optimizer kernel (FlatAdamW.step(grad_scale)): no extra pass over the slab

    def losses(self, pred_frames, future, pred_feats, future_feats):
        """cal_lossT (train_NAR.py:32-47, 81-91): MSE + GDL on the frames, lam_pc * BiPatchNCE on the L2-normalised NCE projections"""
        a = self.T.NCE_projector(pred_feats.permute(0, 1, 3, 4, 2))      # (N, T, h, w, C): token-major memory
        b = self.T.NCE_projector(future_feats.permute(0, 1, 3, 4, 2))
        N, T, h, w, C = a.shape
        l_mse, l_gdl = ops.mse_gdl(pred_frames, future)
        l_pc = ops.nce_loss(b.reshape(-1, C), a.reshape(-1, C), N * T, h * w, self.nce_temperature)
        return l_gdl + l_mse + self.lam_pc * l_pc, l_gdl, l_mse, l_pc

    def _step_impl(self, past, future, front=False):
        with torch.no_grad():
            # train_NAR.py:54-56 encodes past and future in two calls; the encoder is per-frame (eval-mode BN), so one call on
            # the concatenated clip gives the same features with twice the rows per conv GEMM (480 instead of 240 tiles)
            if past.shape[0] == future.shape[0] and past.shape[2:] == future.shape[2:]:
                feats = self.enc(torch.cat([past, future], dim=1))
                past_feats, future_feats = feats[:, :past.shape[1]], feats[:, past.shape[1]:]
            else:
                past_feats, future_feats = self.enc(past), self.enc(future)
        if not self.T.training:   # nn.Module.train() walks ~500 sub-modules: ~5 ms of host time per step
            self.T.train()
        self.opt.zero_grad()
        if self.dec_weight_grads:
            self.dec.zero_grad(set_to_none=True)  # train_NAR.py:61
        if self._bufsync and not front:      # a collective: the front-graph step issues it eagerly before the replay
            self._bufsync.sync()
        pred_feats = self.T(past_feats)
        pred_frames = self.dec(pred_feats)
        extra = {}
        if self.disc is not None:
            extra = self._disc_update(pred_frames, future)
        loss, l_gdl, l_mse, l_pc = self.losses(pred_frames, future, pred_feats, future_feats)
        if self.disc is not None:
            extra["T_gan"] = self.gan(self.disc(pred_frames.flatten(0, 1)), True)
            loss = loss + self.lam_gan * extra["T_gan"]
            extra["T_gan"] = extra["T_gan"].detach()
        return self._finish(loss, dict({"T_total": loss.detach(), "T_GDL": l_gdl.detach(), "T_MSE": l_mse.detach(), "T_bpc": l_pc.detach()},
                                       **extra), front)

    def step(self, past, future):
        if self._front is not None:
            # data-parallel step: forward + backward replayed as one hipGraph, then the part that talks to other ranks runs eagerly --
            # DP_CHUNKS grouped weight-gradient launches on the recorded operands (static addresses in the graph's pool), the all-reduces
            # between them, the optimizer (3 launches)
            if self.opt.planes is not None and self.opt.planes.stale():
                self.opt.planes.refresh()
            self._static_past.copy_(past)
            self._static_future.copy_(future)
            if self._bufsync:
                self._bufsync.sync()
            self._front.replay()
            ops.requeue_wgrads(self._front_items)
            self._grad_scale = 1.0
            if os.environ.get("VPTR_DP_OVERLAP", "1") == "0":
                ops.flush_wgrads()
                self._allreduce_grads()
            else:
                self._exchange_held_wgrads()
            self.opt.step(grad_scale=self._grad_scale)
            self._static_out["grad_norm"] = self.opt.grad_norm()
            return self._static_out
        if self._graph is not None:
            # replays are stream-ordered like any other launch; tests/test_20_graph_gpu.py compares them with eager steps (the
            # round-1 corruption was a captured table upload reading a recycled pinned buffer: ops._to_device_async)
            if self.opt.planes is not None and self.opt.planes.stale():
                self.opt.planes.refresh()   # parameters were changed from outside (load_state_dict) since the last replay
            self._static_past.copy_(past)
            self._static_future.copy_(future)
            self._graph.replay()
            return self._static_out
        return self._step_impl(past, future)

    @staticmethod
    def _warm_up_for_capture(warmup, step):
        """eager steps in front of a capture: `warmup` - 1 of them, then as many more as the tile-row tuning of the grouped weight-gradient
        launches still wants (ops.wgrad_tune_open: each setting is timed ops._TUNE_SAMPLES times, the first, cold sample not counted), then
        the geometry is FIXED (ops.wgrad_tune_settle) and one last step issues exactly the launches -- and table uploads, counted in
        ops._upload_stats -- the capture will issue."""
        for _ in range(max(warmup, 1) - 1):
            step()
        extra = 0
        while ops.wgrad_tune_open() and extra < 2 * ops._TUNE_SAMPLES:
            step()
            extra += 1
        ops.wgrad_tune_settle()
        ops._upload_stats.update(count=0, max_bytes=0)
        step()

    def capture(self, past, future, warmup=3):
        """Capture the whole step (forward, losses, backward, clip, AdamW) into one hipGraph.  `past`/`future` give the
        static shapes; real data is copied into the captured input buffers by `step`.

        The captured graph is inspected before it is instantiated: it must consist of kernel and memcpy nodes only.  A MEMSET
        node (hipMemsetAsync under capture -- ATen's multi-block reductions zero their semaphores that way) writes garbage from
        the second replay on with this HIP runtime (tools/memset_node_probe.py), which is what corrupted the round-2 graph
        (F.normalize's backward in the BiPatchNCE branch); the step's own losses are plain kernels now (csrc/losses.hip) and
        any memset node that sneaks back in raises here instead of silently training on garbage."""
        if self.pg is not None and self.world > 1:
            raise RuntimeError("graph capture of the data-parallel step is not enabled (RCCL calls stay eager)")
        self._static_past, self._static_future = past.clone(), future.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._warm_up_for_capture(warmup, lambda: self._step_impl(self._static_past, self._static_future))
        torch.cuda.current_stream().wait_stream(s)
        # the captured step uploads as many host-built tables as the last warm-up step did (grouped launches: 2 per group, more with
        # the GAN branch or non-P16 groups); each gets a pinned buffer of its own that lives as long as the graph
        ops.reserve_graph_staging(count=ops._upload_stats["count"] + 4, nbytes=max(1 << 16, 2 * ops._upload_stats["max_bytes"]))
        ops.wgrad_tune_settle()
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g):
            self._static_out = self._step_impl(self._static_past, self._static_future)
        self.graph_nodes = graph_node_census(g)
        bad = {k: v for k, v in self.graph_nodes.items() if k not in ("kernel", "memcpy", "empty")}
        if bad:
            g.reset()
            raise RuntimeError("captured step holds node types that are not replay-safe on this HIP runtime: %s (all nodes: %s)"
                               % (bad, self.graph_nodes))
        g.instantiate()
        self._graph = g
        return g

    def capture_front(self, past, future, warmup=3):
        """Data-parallel twin of `capture`: everything of the step that involves no other rank -- encoder, forward, losses, backward
        with the weight gradients only RECORDED (ops.hold_wgrads), the LayerNorm parameter-gradient reductions -- becomes one hipGraph;
        `step` replays it and then runs the exchange eagerly (grouped weight-gradient chunks, RCCL all-reduces, optimizer): ~1000
        launches per step leave the host's critical path, no collective is ever captured.  The recorded weight-gradient operands live
        in the graph's private pool, so their addresses are the same at every replay."""
        if self.disc is not None:
            raise RuntimeError("capture_front: the adversarial branch all-reduces inside the step; run it eagerly")
        if self.pg is None:
            raise RuntimeError("capture_front is the data-parallel capture; use capture() without a process group")
        self._static_past, self._static_future = past.clone(), future.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._warm_up_for_capture(warmup, lambda: self._step_impl(self._static_past, self._static_future))
        torch.cuda.current_stream().wait_stream(s)
        ops.reserve_graph_staging(count=ops._upload_stats["count"] + 4, nbytes=max(1 << 16, 2 * ops._upload_stats["max_bytes"]))
        ops.wgrad_tune_settle()
        # c10d's RCCL watchdog thread polls the events of earlier collectives (hipEventQuery): under the default GLOBAL capture mode
        # that call, made while THIS thread captures, aborts the process ("operation not permitted when stream is capturing").  Drain
        # the device first (no work left to poll) and capture in thread-local mode (other threads' calls stay legal).
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._static_out = self._front_impl(self._static_past, self._static_future)
            self._front_items = ops.take_wgrads()
        self.graph_nodes = graph_node_census(g)
        bad = {k: v for k, v in self.graph_nodes.items() if k not in ("kernel", "memcpy", "empty")}
        if bad:
            g.reset()
            raise RuntimeError("captured step holds node types that are not replay-safe on this HIP runtime: %s (all nodes: %s)"
                               % (bad, self.graph_nodes))
        g.instantiate()
        self._front = g
        return g

    def _front_impl(self, past, future):
        """`_step_impl` up to the end of the backward pass: weight gradients recorded but not launched, no collective, no optimizer"""
        return self._step_impl(past, future, front=True)

    def _finish(self, loss, terms, front):
        """backward + exchange + optimizer (the whole step), or -- front=True, under capture_front -- backward only, with the weight
        gradients held for the eager tail"""
        if front:
            with ops.hold_wgrads():
                loss.backward()
            ops.flush_partial_reduces()
            return terms
        self._backward_and_exchange(loss)
        self.opt.step(grad_scale=self._grad_scale)
        terms["grad_norm"] = self.opt.grad_norm()
        return terms

    # -- replay == eager, checked on the live model (bench.py runs this before it times a graph) -------------------------------------
    def _snapshot(self):
        dev = self.opt.flat.device
        snap = {"flat": self.opt.flat.clone(), "m": self.opt.m.clone(), "v": self.opt.v.clone(), "step": self.opt.step_dev.clone(),
                "buffers": [b.detach().clone() for b in self.T.buffers()], "seed": ops._master_seed(dev).clone()}
        if self.disc is not None:   # the adversarial branch steps a second slab and the discriminator's BatchNorm statistics
            o = self.opt_D
            snap["D"] = {"flat": o.flat.clone(), "m": o.m.clone(), "v": o.v.clone(), "step": o.step_dev.clone(),
                         "buffers": [b.detach().clone() for b in self.disc.buffers()]}
        return snap

    def _restore(self, snap):
        dev = self.opt.flat.device
        with torch.no_grad():
            self.opt.flat.copy_(snap["flat"]); self.opt.m.copy_(snap["m"]); self.opt.v.copy_(snap["v"]); self.opt.step_dev.copy_(snap["step"])
            for b, v in zip(self.T.buffers(), snap["buffers"]):
                b.copy_(v)
            ops._master_seed(dev).copy_(snap["seed"])
            if self.disc is not None:
                o, d = self.opt_D, snap["D"]
                o.flat.copy_(d["flat"]); o.m.copy_(d["m"]); o.v.copy_(d["v"]); o.step_dev.copy_(d["step"])
                for b, v in zip(self.disc.buffers(), d["buffers"]):
                    b.copy_(v)
                if o.planes is not None:
                    o.planes.refresh()
        ops._seed_scope.pop(ops._dev_key(dev), None)
        if self.opt.planes is not None:
            self.opt.planes.refresh()

    @staticmethod
    def _term_diff(a, b):
        return abs(a - b) / (abs(a) + 1e-6) if (b == b and abs(b) < 1e30) else float("inf")

    def verify_graph(self, past, future, steps=3, rtol=2e-3, traj_rtol=5e-2, param_rtol=3e-4):
        """Replay == eager, checked on the live model; the state is restored afterwards.  -> (ok, report)

        Two comparisons, because a train step is a noise amplifier (the order of fp32 atomics perturbs a
        gradient by ~1e-7, the first AdamW updates are ~lr * sign(g), and a random-filled model turns the resulting sign flips into
        1e-4 ... 1e-3 differences of the NEXT step's gradient norm):
          * lock-step (bound `rtol` per term, `param_rtol` on the post-step parameters): replay i and eager step i both start from
            the SAME state -- the state the previous replay left (parameters, Adam moments, step count, BatchNorm statistics, dropout
            seed; the discriminator's too on the GAN branch).  Differences are those of ONE forward / backward / update from identical
            inputs with identical masks, nothing accumulates, and replay i is still the i-th consecutive replay of the graph (the
            round-2 corruption began at the second replay).
          * trajectory (`traj_rtol`, loose, plus finiteness): `steps` back-to-back replays with no host read in between vs `steps`
            eager steps from the same initial state."""
        if self._graph is None and self._front is None:
            raise RuntimeError("verify_graph: capture() / capture_front() first")
        snap0 = self._snapshot()
        kind = "_graph" if self._graph is not None else "_front"
        g = getattr(self, kind)
        worst, where, prel = 0.0, None, 0.0
        lock = []
        try:
            for i in range(steps):
                pre = self._snapshot()
                setattr(self, kind, g)
                rg = {k: float(v) for k, v in self.step(past, future).items()}
                post_g = self._snapshot()
                self._restore(pre)
                setattr(self, kind, None)
                re_ = {k: float(v) for k, v in self.step(past, future).items()}
                pe, pg = self.opt.flat.double(), post_g["flat"].double()
                prel = max(prel, float((pe - pg).norm() / pe.norm()))
                for k in re_:
                    d = self._term_diff(re_[k], rg[k])
                    if d > worst:
                        worst, where = d, (i, k, re_[k], rg[k])
                lock.append((re_, rg))
                self._restore(post_g)          # the chain follows the graph: replay i + 1 sees what replay i left
            runs = {}
            for mode in ("eager", "graph"):
                self._restore(snap0)
                setattr(self, kind, g if mode == "graph" else None)
                outs = [self.step(past, future) for _ in range(steps)]           # no host read between the steps
                torch.cuda.synchronize()
                runs[mode] = [{k: float(v) for k, v in o.items()} for o in (outs if mode == "eager" else outs[-1:])]
            tworst, twhere = 0.0, None
            for k, a in runs["eager"][-1].items():
                d = self._term_diff(a, runs["graph"][-1][k])
                if d > tworst:
                    tworst, twhere = d, (steps - 1, k, a, runs["graph"][-1][k])
        finally:
            setattr(self, kind, g)
            self._restore(snap0)
        ok = worst <= rtol and prel <= param_rtol and tworst <= traj_rtol
        return ok, {"steps": steps, "worst_term_rel_diff": worst, "worst_term": where, "param_rel_l2": prel,
                    "trajectory_worst_rel_diff": tworst, "trajectory_worst": twhere,
                    "eager_last": runs["eager"][-1], "graph_last": runs["graph"][-1]}

    @torch.no_grad()
    def predict(self, past):
        """Inference: Enc -> NAR -> Dec (Test_VPTR.ipynb cell 5, one NAR round)."""
        self.T.eval()
        return self.dec(self.T(self.enc(past)))


class FARTrainer(NARTrainer):
    """One FAR training step without the GAN branch (`single_iter` of train_FAR.py:48-101 with VPTR_Disc = None, the
    script's default :186-192): Enc(cat(past, future[:, :-1])) under no_grad, VPTRFormerFAR (causal temporal attention as a
    kernel flag), Dec, MSE + GDL against cat(past[:, 1:], future), backward, clip_grad_norm_(max_norm), AdamW.  Shares the
    flat-slab optimizer, grouped weight gradients and data-parallel exchange with `NARTrainer`."""

    def __init__(self, enc, dec, transformer, lr=1e-4, max_grad_norm=1.0, process_group=None, bucket_mb=64, dec_weight_grads=True,
                 disc=None, lam_gan=None, gan_mode="vanilla"):
        self.enc, self.dec, self.T = enc.eval(), dec.eval(), transformer
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.bucket_elems = bucket_mb * (1 << 20) // 4
        self._init_gan(disc, lam_gan, lr, gan_mode)
        for p in enc.parameters():
            p.requires_grad_(False)
        for p in dec.parameters():
            p.requires_grad_(bool(dec_weight_grads))  # train_FAR.py:181-182 leaves the decoder trainable (never stepped)
        self.dec_weight_grads = bool(dec_weight_grads)
        for mod in list(self.enc.modules()) + list(self.dec.modules()):
            mod._vptr_frozen = True
        self.opt = FlatAdamW(self.T.parameters(), lr=lr, max_grad_norm=max_grad_norm, channel_last=_channel_last_ids(self.T))
        self._bufsync = None   # the FAR transformer has no BatchNorm (LayerNorm conv-FFNs): nothing to broadcast per forward
        self._graph = None

    def _step_impl(self, past, future, front=False):
        with torch.no_grad():
            gt_feats = self.enc(torch.cat([past, future[:, :-1]], dim=1))    # train_FAR.py:53-55
        if not self.T.training:   # nn.Module.train() walks ~500 sub-modules: ~5 ms of host time per step
            self.T.train()
        self.opt.zero_grad()
        if self.dec_weight_grads:
            self.dec.zero_grad(set_to_none=True)
        pred_frames = self.dec(self.T(gt_feats))
        extra = {}
        if self.disc is not None:                                            # cal_lossD(VPTR_Disc, pred_frames, future_frames) :72
            extra = self._disc_update(pred_frames, future)
        real = torch.cat([past[:, 1:], future], dim=1)                       # :80
        l_mse, l_gdl = ops.mse_gdl(pred_frames, real)                        # cal_lossT :32-46
        loss = l_gdl + l_mse
        if self.disc is not None:
            t_gan = self.gan(self.disc(pred_frames.flatten(0, 1)), True)
            loss = loss + self.lam_gan * t_gan
            extra["T_gan"] = t_gan.detach()
        return self._finish(loss, dict({"T_total": loss.detach(), "T_GDL": l_gdl.detach(), "T_MSE": l_mse.detach()}, **extra), front)

    @torch.no_grad()
    def predict(self, past, num_pred):
        """Autoregressive test-phase rollout (train_FAR.py:103-125): see vptr_amd.inference.far_rollout."""
        from .inference import far_rollout
        return far_rollout(self.enc, self.dec, self.T, past, num_pred)


class AETrainer:
    """One stage-1 step (`single_iter` of train_AutoEncoder.py:44-78): rec = Dec(Enc(cat(past, future))) with train-mode
    BatchNorm; discriminator update on (rec.detach(), x): 0.5 * lam_gan * (BCE(D(fake), 0) + BCE(D(real), 1)), Adam; generator
    update: lam_gan * BCE(D(rec), 1) + MSE + GDL, Adam on Enc + Dec.  torch.optim.Adam(lr 2e-4, betas (0.5, 0.999)) is
    AdamW with weight_decay 0, so both optimizers are `FlatAdamW` slabs (no clipping in this stage)."""

    def __init__(self, enc, dec, disc, lr=2e-4, lam_gan=0.01, betas=(0.5, 0.999), gan_mode="vanilla"):
        self.enc, self.dec, self.disc = enc, dec, disc
        for m in (enc, dec, disc):
            for p in m.parameters():
                p.requires_grad_(True)
        self.opt_G = FlatAdamW(list(enc.parameters()) + list(dec.parameters()), lr=lr, betas=betas, weight_decay=0.0)
        self.opt_D = FlatAdamW(list(disc.parameters()), lr=lr, betas=betas, weight_decay=0.0)
        self.gan = GANLoss(gan_mode, target_real_label=1.0, target_fake_label=0.0).to(self.opt_G.flat.device)
        self.lam_gan = lam_gan

    def step(self, past, future):
        x = torch.cat([past, future], dim=1)
        if not self.enc.training:
            self.enc.train()
        if not self.dec.training:
            self.dec.train()
        self.opt_G.zero_grad()
        rec = self.dec(self.enc(x))
        # ---- discriminator (cal_lossD, :20-29)
        if not self.disc.training:
            self.disc.train()
        for p in self.disc.parameters():
            p.requires_grad_(True)
        self.opt_D.zero_grad()
        l_fake = self.gan(self.disc(rec.detach().flatten(0, 1)), False)
        l_real = self.gan(self.disc(x.flatten(0, 1)), True)
        loss_D = (l_fake + l_real) * 0.5 * self.lam_gan
        loss_D.backward()
        self.opt_D.step()
        # ---- generator (cal_lossG, :31-42)
        for p in self.disc.parameters():
            p.requires_grad_(False)
        l_gan = self.gan(self.disc(rec.flatten(0, 1)), True)
        l_mse, l_gdl = ops.mse_gdl(rec, x)
        loss_G = self.lam_gan * l_gan + l_mse + l_gdl
        loss_G.backward()
        self.opt_G.step()
        return {"AEgan": l_gan.detach(), "AE_MSE": l_mse.detach(), "AE_GDL": l_gdl.detach(), "AE_total": loss_G.detach(),
                "Dtotal": loss_D.detach(), "Dfake": l_fake.detach(), "Dreal": l_real.detach()}


def script_style_nar_iter(enc, dec, transformer, optimizer, past, future, mse_loss, gdl_loss, bpnce, lam_pc=0.1, max_grad_norm=1.0):
    """One stage-2 iteration the way the reference's OWN script drives the `model` package (train_NAR.py:49-107 without the GAN
    branch): module calls, `zero_grad(set_to_none=True)`, the criterion classes + F.normalize, `loss.backward()`,
    `nn.utils.clip_grad_norm_`, a stock `torch.optim` optimizer -- no NARTrainer, no flat slab, no fused losses.  This is what
    "train_NAR.py drops in unchanged" executes on this package; tests/test_04_dropin_gpu.py pins it against the reference's step
    records and bench.py times it beside the NARTrainer step (`other_configs.drop_in_single_iter`)."""
    with torch.no_grad():
        past_feats = enc(past)
        future_feats = enc(future)
    transformer = transformer.train()
    transformer.zero_grad(set_to_none=True)
    dec.zero_grad(set_to_none=True)
    pred_feats = transformer(past_feats)
    pred_frames = dec(pred_feats)
    pf = transformer.NCE_projector(pred_feats.permute(0, 1, 3, 4, 2)).permute(0, 1, 4, 2, 3)
    gf = transformer.NCE_projector(future_feats.permute(0, 1, 3, 4, 2)).permute(0, 1, 4, 2, 3)
    l_mse = mse_loss(pred_frames, future)
    l_gdl = gdl_loss(future, pred_frames)
    l_pc = bpnce(F.normalize(gf, p=2.0, dim=2), F.normalize(pf, p=2.0, dim=2))
    loss = l_gdl + l_mse + lam_pc * l_pc
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(transformer.parameters(), max_norm=max_grad_norm, norm_type=2)
    optimizer.step()
    return {"T_total": loss.detach(), "T_GDL": l_gdl.detach(), "T_MSE": l_mse.detach(), "T_bpc": l_pc.detach(), "grad_norm": gn.detach()}
