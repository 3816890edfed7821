"""Attention cores (window + RPE, temporal, temporal-spatial) and the projection + attention nodes."""

import torch

from .. import _lib
from .._lib import check, lib, ptr, stream
from .core import A_P16, B_P16, _c, _direct_apply, gemm_raw, p16_ok, seed_tensor, to_p16
from .grads import flat_grad_for
from .planes import weight_planes_for
from .linear import _linear_param_grads


# ------------------------------------------------------------------------------------------------------------------
# attention cores
# ------------------------------------------------------------------------------------------------------------------
def _winattn_workspace(device, nh):
    """scratch of one window-attention backward call (the bias-table gradient's per-workgroup partial sums, vptr_winattn_bwd_ws):
    a fresh caching-allocator block per call -- stream-ordered like every other temporary, so calls may overlap nothing"""
    return torch.empty((lib.vptr_winattn_bwd_workspace(int(nh)),), device=device, dtype=torch.float32)


class _WinAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, table, rel_index, B, H, W, nh, ws, p, site):
        q, k, v = _c(q), _c(k), _c(v)
        C = q.shape[1]
        o = torch.empty_like(q)
        ctx.seed = seed_tensor(q.device) if p > 0 else None
        check(lib.vptr_winattn_fwd(ptr(q), ptr(k), ptr(v), ptr(table), ptr(rel_index), ptr(o), B, H, W, C, nh, ws, p,
                                   ptr(ctx.seed), site, 0, stream()), "vptr_winattn_fwd")
        ctx.save_for_backward(q, k, v, table, rel_index)
        ctx.cfg = (B, H, W, nh, ws, p, site)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, table, rel_index = ctx.saved_tensors
        B, H, W, nh, ws, p, site = ctx.cfg
        do = _c(do)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        slab = flat_grad_for(table)
        dtable = slab if slab is not None else (torch.zeros_like(table) if table is not None else None)
        wsp = _winattn_workspace(q.device, nh) if dtable is not None else None
        check(lib.vptr_winattn_bwd_ws(ptr(q), ptr(k), ptr(v), ptr(table), ptr(rel_index), ptr(do), ptr(dq), ptr(dk), ptr(dv),
                                      ptr(dtable), B, H, W, q.shape[1], nh, ws, p, ptr(ctx.seed), site, 1.0, 0, ptr(wsp),
                                      wsp.numel() if wsp is not None else 0, stream()), "vptr_winattn_bwd_ws")
        if slab is not None:
            dtable = None
        return dq, dk, dv, dtable, None, None, None, None, None, None, None, None


_WinAttnFn_apply = _direct_apply(_WinAttnFn)


def window_attention(q, k, v, table, rel_index, B, H, W, nh, ws, dropout_p=0.0, site=0):
    """q (pre-scaled), k, v: [B*H*W, C]; table [(2ws-1)^2, nh] or None; returns [B*H*W, C] (before out_proj)."""
    return _WinAttnFn_apply(q, k, v, table, rel_index, int(B), int(H), int(W), int(nh), int(ws), float(dropout_p), int(site))


class _TAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, Nb, Tq, Tk, HW, nh, causal, p, site):
        q, k, v = _c(q), _c(k), _c(v)
        C = q.shape[1]
        o = torch.empty_like(q)
        ctx.seed = seed_tensor(q.device) if p > 0 else None
        check(lib.vptr_tattn_fwd(ptr(q), ptr(k), ptr(v), ptr(o), Nb, Tq, Tk, HW, C, nh, causal, p, ptr(ctx.seed), site,
                                 0, stream()), "vptr_tattn_fwd")
        ctx.save_for_backward(q, k, v)
        ctx.cfg = (Nb, Tq, Tk, HW, nh, causal, p, site)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v = ctx.saved_tensors
        Nb, Tq, Tk, HW, nh, causal, p, site = ctx.cfg
        do = _c(do)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        check(lib.vptr_tattn_bwd(ptr(q), ptr(k), ptr(v), ptr(do), ptr(dq), ptr(dk), ptr(dv), Nb, Tq, Tk, HW, q.shape[1], nh,
                                 causal, p, ptr(ctx.seed), site, 1.0, 0, stream()), "vptr_tattn_bwd")
        return dq, dk, dv, None, None, None, None, None, None, None, None


_TAttnFn_apply = _direct_apply(_TAttnFn)


def temporal_attention(q, k, v, Nb, Tq, Tk, HW, nh, causal=False, dropout_p=0.0, site=0):
    """q [(n,tq,p), C] pre-scaled; k, v [(n,tk,p), C]; attends over time for every (n, pixel, head)."""
    return _TAttnFn_apply(q, k, v, int(Nb), int(Tq), int(Tk), int(HW), int(nh), int(bool(causal)), float(dropout_p), int(site))


class KVGradAccum:
    """Shared accumulators for the input gradients of ONE key / value source that several attentions read (the encoder memory of the
    8 decoder blocks): every `_ProjAttnFn` forward that is handed the object counts itself in; in backward the first one allocates the
    two [Mk, K] buffers, the following ones add into them inside their input-gradient GEMM (vptr_gemm_desc.batch_accum), and the LAST
    one hands the sums to autograd -- the others return None.  One object per forward pass."""

    def __init__(self, sources=()):
        """sources: the shared key / value tensors themselves -- their autograd nodes tell, per backward pass, whether anybody wants
        the gradient this object sums (a pruned pass -- torch.autograd.grad(loss, [one decoder weight]), backward(inputs=...) -- visits
        only some of the users and needs no memory gradient at all)"""
        self.uses, self.k, self.v = 0, None, None
        self.left, self.task = 0, -1      # users still to come in the running backward pass; its graph-task id
        self.needed = True
        self.nodes = []
        for t in sources:
            if t is None or not t.requires_grad:
                continue
            if t.grad_fn is not None:
                self.nodes.append(t.grad_fn)
            else:
                with torch.enable_grad():
                    self.nodes.append(t.view_as(t).grad_fn.next_functions[0][0])

    def _source_grad_needed(self):
        will = getattr(torch._C, "_will_engine_execute_node", None)
        if will is None or not self.nodes:
            return True
        try:
            return any(bool(will(n)) for n in self.nodes)
        except (RuntimeError, TypeError):
            return True

    def enter_backward(self):
        """called by every user's backward; True for the first user of a backward pass.  Participation is counted per BACKWARD
        pass (a second pass over a retained graph starts a fresh count).  A pass that ends with users missing although the running
        graph task wants the sources' gradient (a user that took another code path, a loss taken from an intermediate layer) raises
        instead of silently handing over an incomplete sum; a pruned pass that does not want that gradient just drops the sums."""
        task = torch._C._current_graph_task_id()
        if self.left == 0 or task != self.task:
            if self.left != 0:
                left, needed = self.left, self.needed
                self.k = self.v = None
                self.left = 0
                if needed:
                    raise RuntimeError("KVGradAccum: the previous backward pass ended with %d of %d users missing" % (left, self.uses))
            self.left, self.task = self.uses, task
            self.needed = self._source_grad_needed()
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self._check_done)
            except RuntimeError:
                pass
            return True
        return False

    def _check_done(self):
        if self.left != 0:
            left, self.left, self.k, self.v = self.left, 0, None, None
            if self.needed:
                raise RuntimeError("KVGradAccum: backward finished with %d of %d key / value users not visited: the gradient of the shared "
                                   "key / value source would be incomplete" % (left, self.uses))


class _ProjAttnFn(torch.autograd.Function):
    """o = attention(alpha * (xq Wq^T + bq), xk Wk^T + bk, xv Wv^T + bv), alpha = head_dim^-0.5: the q/k/v projections
    (MultiHeadAttentionRPE.py:543-545,586; nn.MultiheadAttention's in_proj, VidHRFormer_modules.py:79-84) and the attention
    core as one autograd node.

    A 528 x 528 projection of 10 240 tokens alone is 240 tiles on 256 CUs and spends half of its time in prologue and
    epilogue, so the three projections run as ONE batched launch (vptr_gemm_desc.batch: 720 tiles, two workgroups per CU),
    and the input gradients as one K-segmented GEMM dX = dQ.Wq + dK.Wk + dV.Wv (vptr_gemm_desc.ksegs) when q, k and v come
    from the same tensor, dXqk = dQ.Wq + dK.Wk plus dXv when only q and k do, a batched launch of three otherwise.  The
    attention backward kernels emit dQ already multiplied by alpha (dq_scale), i.e. w.r.t. the unscaled projection.

    kind 0: local-window attention with relative-position bias, geom = (B, H, W, ws);
    kind 1: temporal attention, geom = (Nb, Tq, Tk, HW, causal).
    same_qk / same_v: xk is xq / xv is xq.  merge_v: the CALLER guarantees that xv's gradient is only ever added to xq's
    (xq = xv + a constant table): the whole input gradient is then returned for xq and None for xv."""

    @staticmethod
    def forward(ctx, xq, xk, xv, Wq, bq, Wk, bk, Wv, bv, table, rel_index, kind, geom, nh, p, site, same_qk, same_v, merge_v, x_p16, o_p16,
                kv_acc=None):
        _lib.require_cuda(xq, xk, xv, Wq)
        ctx.kv_acc = kv_acc
        if kv_acc is not None:
            kv_acc.uses += 1
        xq, xk, xv = _c(xq), _c(xk), _c(xv)
        Wq, Wk, Wv = _c(Wq), _c(Wk), _c(Wv)
        Mq, K = xq.shape
        Mk = xk.shape[0]
        N = Wq.shape[0]
        alpha = float(N // nh) ** -0.5
        dev = xq.device
        use = p16_ok(K, N)
        if (x_p16 or o_p16) and not use:
            raise RuntimeError("attention: P16 operands need the embedding width to be a multiple of 16 (got %d)" % K)
        q = torch.empty((Mq, N), device=dev, dtype=torch.float32)
        k = torch.empty((Mk, N), device=dev, dtype=torch.float32)
        v = torch.empty((Mk, N), device=dev, dtype=torch.float32)
        if use:
            if not x_p16:   # one conversion pass per distinct input
                cq = to_p16(xq)
                ck = cq if same_qk else to_p16(xk)
                cv = cq if same_v else to_p16(xv)
                xq, xk, xv = cq, ck, cv
            (Pq, lq, _, _), (Pk, lk, _, _), (Pv, lv, _, _) = weight_planes_for(Wq), weight_planes_for(Wk), weight_planes_for(Wv)
            if not (lq == lk == lv):
                raise RuntimeError("attention: q/k/v weight planes with different pitches")
            if Mq == Mk:
                gemm_raw(xq, Pq, q, Mq, N, K, A_P16, B_P16, lda=K, ldb=lq, bias=bq, alpha=alpha,
                         batch_extra=[(xk, Pk, k, bk, 1.0), (xv, Pv, v, bv, 1.0)])
            else:
                gemm_raw(xq, Pq, q, Mq, N, K, A_P16, B_P16, lda=K, ldb=lq, bias=bq, alpha=alpha)
                gemm_raw(xk, Pk, k, Mk, N, K, A_P16, B_P16, lda=K, ldb=lk, bias=bk, batch_extra=[(xv, Pv, v, bv, 1.0)])
        elif Mq == Mk:
            gemm_raw(xq, Wq, q, Mq, N, K, 0, 0, bias=bq, alpha=alpha, batch_extra=[(xk, Wk, k, bk, 1.0), (xv, Wv, v, bv, 1.0)])
        else:
            gemm_raw(xq, Wq, q, Mq, N, K, 0, 0, bias=bq, alpha=alpha)
            gemm_raw(xk, Wk, k, Mk, N, K, 0, 0, bias=bk, batch_extra=[(xv, Wv, v, bv, 1.0)])
        o = torch.empty_like(q)
        ctx.seed = seed_tensor(dev) if p > 0 else None
        if kind == 0:
            B, H, W, ws = geom
            check(lib.vptr_winattn_fwd(ptr(q), ptr(k), ptr(v), ptr(table), ptr(rel_index), ptr(o), B, H, W, N, nh, ws, p,
                                       ptr(ctx.seed), site, int(o_p16), stream()), "vptr_winattn_fwd")
        else:
            Nb, Tq, Tk, HW, causal = geom
            check(lib.vptr_tattn_fwd(ptr(q), ptr(k), ptr(v), ptr(o), Nb, Tq, Tk, HW, N, nh, causal, p, ptr(ctx.seed), site,
                                     int(o_p16), stream()), "vptr_tattn_fwd")
        ctx.save_for_backward(xq, xk, xv, Wq, Wk, Wv, q, k, v, table, rel_index)
        ctx.bias_refs = (bq, bk, bv)   # the parameters (or views of them) themselves: gradient-destination lookup
        ctx.cfg = (kind, geom, nh, p, site, alpha, same_qk, same_v, merge_v, use)
        return o

    @staticmethod
    def backward(ctx, do):
        xq, xk, xv, Wq, Wk, Wv, q, k, v, table, rel_index = ctx.saved_tensors
        kind, geom, nh, p, site, alpha, same_qk, same_v, merge_v, use = ctx.cfg
        do = _c(do)
        Mq, K = xq.shape
        Mk = xk.shape[0]
        N = Wq.shape[0]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dtable = None
        if kind == 0:
            B, H, W, ws = geom
            slab = flat_grad_for(table) if table is not None else None
            dtable = slab if slab is not None else (torch.zeros_like(table) if table is not None else None)
            wsp = _winattn_workspace(q.device, nh) if dtable is not None else None
            check(lib.vptr_winattn_bwd_ws(ptr(q), ptr(k), ptr(v), ptr(table), ptr(rel_index), ptr(do), ptr(dq), ptr(dk), ptr(dv),
                                          ptr(dtable), B, H, W, N, nh, ws, p, ptr(ctx.seed), site, alpha, int(use), ptr(wsp),
                                          wsp.numel() if wsp is not None else 0, stream()), "vptr_winattn_bwd_ws")
            if slab is not None:
                dtable = None
        else:
            Nb, Tq, Tk, HW, causal = geom
            check(lib.vptr_tattn_bwd(ptr(q), ptr(k), ptr(v), ptr(do), ptr(dq), ptr(dk), ptr(dv), Nb, Tq, Tk, HW, N, nh, causal, p,
                                     ptr(ctx.seed), site, alpha, int(use), stream()), "vptr_tattn_bwd")
        need = ctx.needs_input_grad
        rq, rk, rv = ctx.bias_refs
        dWq, dbq = _linear_param_grads(dq, xq, Wq, rq, need[3], need[4], p16=use)
        dWk, dbk = _linear_param_grads(dk, xk, Wk, rk, need[5], need[6], p16=use)
        dWv, dbv = _linear_param_grads(dv, xv, Wv, rv, need[7], need[8], p16=use)

        def new(M):
            return torch.empty((M, K), device=do.device, dtype=torch.float32)
        if use:
            (_, _, Tq_, lt), (_, _, Tk_, _), (_, _, Tv_, _) = weight_planes_for(Wq), weight_planes_for(Wk), weight_planes_for(Wv)
            am, bm, lda, ldb = A_P16, B_P16, N, lt
        else:
            Tq_, Tk_, Tv_ = Wq, Wk, Wv
            am, bm, lda, ldb = 0, 1, None, None

        def dgrad(g, WT, out, M, **kw):
            return gemm_raw(g, WT, out, M, K, N, am, bm, lda=lda, ldb=ldb, **kw)
        dxq = dxk = dxv = None
        took_acc = False
        if same_qk and (same_v or merge_v):
            if need[0] or need[1] or need[2]:
                dxq = dgrad(dq, Tq_, new(Mq), Mq, kseg_extra=[(dk, Tk_), (dv, Tv_)])
        elif same_qk:
            if need[0] or need[1]:
                dxq = dgrad(dq, Tq_, new(Mq), Mq, kseg_extra=[(dk, Tk_)])
            if need[2]:
                dxv = dgrad(dv, Tv_, new(Mk), Mk)
        elif need[0] and need[1] and need[2] and Mq == Mk and use and ctx.kv_acc is not None:
            acc = ctx.kv_acc     # shared key / value source: sum the gradients inside the GEMMs (see KVGradAccum)
            took_acc = True
            acc.enter_backward()
            first = acc.k is None
            if first:
                acc.k, acc.v = new(Mk), new(Mk)
            dxq = new(Mq)
            dgrad(dq, Tq_, dxq, Mq, batch_extra=[(dk, Tk_, acc.k, None, 1.0), (dv, Tv_, acc.v, None, 1.0)], batch_accum=0 if first else 0b110)
            acc.left -= 1
            if acc.left == 0:
                dxk, dxv = acc.k, acc.v
                acc.k = acc.v = None
        elif need[0] and need[1] and need[2] and Mq == Mk:
            dxq, dxk, dxv = new(Mq), new(Mk), new(Mk)
            dgrad(dq, Tq_, dxq, Mq, batch_extra=[(dk, Tk_, dxk, None, 1.0), (dv, Tv_, dxv, None, 1.0)])
        else:
            if need[0]:
                dxq = dgrad(dq, Tq_, new(Mq), Mq)
            if need[1] and need[2]:
                dxk, dxv = new(Mk), new(Mk)
                dgrad(dk, Tk_, dxk, Mk, batch_extra=[(dv, Tv_, dxv, None, 1.0)])
            elif need[1]:
                dxk = dgrad(dk, Tk_, new(Mk), Mk)
            elif need[2]:
                dxv = dgrad(dv, Tv_, new(Mk), Mk)
        if ctx.kv_acc is not None and not took_acc:
            # a user of the shared source that could not take the accumulating branch still counts as visited; if it is the last one
            # of this backward pass it hands the sums over next to its own gradients
            acc = ctx.kv_acc
            acc.enter_backward()
            acc.left -= 1
            if acc.left == 0 and acc.k is not None:
                dxk = acc.k if dxk is None else dxk + acc.k
                dxv = acc.v if dxv is None else dxv + acc.v
                acc.k = acc.v = None
        return (dxq, dxk, dxv, dWq, dbq, dWk, dbk, dWv, dbv, dtable) + (None,) * 12


_ProjAttnFn_apply = _direct_apply(_ProjAttnFn)


def _proj_attention(xq, xk, xv, Wq, bq, Wk, bk, Wv, bv, table, rel_index, kind, geom, nh, p, site, merge_v_grad, x_p16=False, o_p16=False,
                    kv_acc=None):
    same_qk = xk is xq
    same_v = same_qk and xv is xq
    return _ProjAttnFn_apply(xq, xk, xv, Wq, bq, Wk, bk, Wv, bv, table, rel_index, kind, geom, int(nh), float(p), int(site),
                             same_qk, same_v, bool(merge_v_grad) and same_qk, bool(x_p16), bool(o_p16), kv_acc)


def proj_window_attention(xqk, xv, Wq, bq, Wk, bk, Wv, bv, table, rel_index, B, H, W, nh, ws, dropout_p=0.0, site=0,
                          merge_v_grad=False, x_p16=False, o_p16=False):
    """Window attention INCLUDING its q/k/v projections (q and k from xqk, v from xv; [B*H*W, C] tokens); returns the
    [B*H*W, C] heads before out_proj.  merge_v_grad: see _ProjAttnFn."""
    return _proj_attention(xqk, xqk, xv, Wq, bq, Wk, bk, Wv, bv, table, rel_index, 0, (int(B), int(H), int(W), int(ws)), nh,
                           dropout_p, site, merge_v_grad, x_p16, o_p16)


def proj_temporal_attention(q_in, k_in, v_in, Wq, bq, Wk, bk, Wv, bv, Nb, Tq, Tk, HW, nh, causal=False, dropout_p=0.0, site=0,
                            merge_v_grad=False, x_p16=False, o_p16=False, kv_acc=None):
    """Temporal attention INCLUDING its q/k/v projections; q_in [(n,tq,p), C], k_in, v_in [(n,tk,p), C].
    kv_acc: a KVGradAccum shared by every attention that reads the same k_in / v_in tensors (gradients summed inside the GEMMs)."""
    return _proj_attention(q_in, k_in, v_in, Wq, bq, Wk, bk, Wv, bv, None, None, 1,
                           (int(Nb), int(Tq), int(Tk), int(HW), int(bool(causal))), nh, dropout_p, site, merge_v_grad, x_p16, o_p16, kv_acc)


class _TSAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, Nb, Tq, Tk, H, W, ws, nh, p, site):
        q, k, v = _c(q), _c(k), _c(v)
        C = q.shape[1]
        o = torch.empty_like(q)
        ctx.seed = seed_tensor(q.device) if p > 0 else None
        check(lib.vptr_tsattn_fwd(ptr(q), ptr(k), ptr(v), ptr(o), Nb, Tq, Tk, H, W, ws, C, nh, p, ptr(ctx.seed), site, 0, stream()),
              "vptr_tsattn_fwd")
        ctx.save_for_backward(q, k, v)
        ctx.cfg = (Nb, Tq, Tk, H, W, ws, nh, p, site)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v = ctx.saved_tensors
        Nb, Tq, Tk, H, W, ws, nh, p, site = ctx.cfg
        do = _c(do)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        check(lib.vptr_tsattn_bwd(ptr(q), ptr(k), ptr(v), ptr(do), ptr(dq), ptr(dk), ptr(dv), Nb, Tq, Tk, H, W, ws, q.shape[1], nh,
                                  p, ptr(ctx.seed), site, 0, stream()), "vptr_tsattn_bwd")
        return dq, dk, dv, None, None, None, None, None, None, None, None, None


def temporal_spatial_window_attention(q, k, v, Nb, Tq, Tk, H, W, ws, nh, dropout_p=0.0, site=0):
    """q [(n,tq,h,w), C] pre-scaled; k, v [(n,tk,h,w), C]: every ws x ws window attends over (time x window) tokens."""
    return _TSAttnFn.apply(q, k, v, int(Nb), int(Tq), int(Tk), int(H), int(W), int(ws), int(nh), float(dropout_p), int(site))
