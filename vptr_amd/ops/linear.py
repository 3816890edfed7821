"""nn.Linear-shaped operators: linear (one GEMM with fused epilogue), the transformer MLP, frame-statistics buffers."""
import os

import torch

from .. import _lib
from .._lib import check, lib, ptr, stream
from .core import ACT_GELU, ACT_NONE, ACT_RELU, A_P16, B_P16, _c, _direct_apply, config, gemm_raw, p16_ok, seed_tensor, to_p16
from .wgrad import _launch_wgrad_group, _split_k_for, defer_wgrad
from .grads import _bw_zeros, flat_grad_for, grad_dest_for
from .planes import weight_planes_for


def _linear_param_grads(g, x, W, bias_ref, need_w, need_b, alpha=1.0, p16=False):
    """dW[N,K] (+)= alpha * g^T . x and db (+)= alpha * column sums of g for y = x W^T + b with g = dL/dy [M, N].  With a flat
    gradient slab -- or, for plain parameters, their own `.grad` (_loose_grad_for) -- the products are recorded for the grouped
    end-of-backward launch (which also takes the bias gradient from its A staging registers) and (None, None) is returned;
    otherwise fresh tensors are."""
    N, K = W.shape
    M = x.shape[0]
    dW = db = None
    bias_done = False
    if need_w:
        slab = grad_dest_for(W)          # accumulate straight into the flat gradient slab / the parameter's .grad
        want_b = bias_ref is not None and need_b
        bslab = grad_dest_for(bias_ref) if want_b else None
        # the bias gradient rides on the weight's launch: deferring needs an in-place destination for BOTH (a bias whose gradient must
        # go back through autograd -- backward(inputs=[weight]) without it, a hooked bias -- has to be complete when this node returns)
        if slab is not None and config.group_wgrads and (bslab is not None or not want_b or not (p16 or alpha != 1.0)):
            defer_wgrad(g, x, slab, N, K, M, db=bslab, alpha=alpha, p16=p16)   # grouped at the end of backward
            bias_done = bslab is not None
        elif p16:
            # P16 operands without in-place destinations (torch.autograd.grad, stand-alone modules in a torch.distributed job, tests):
            # the token-major kernel as a group of one, results handed to autograd
            dW = slab if slab is not None else _bw_zeros((N, K), g.device)
            if want_b:
                db = bslab if bslab is not None else _bw_zeros((N,), g.device)
                bias_done = True
            _launch_wgrad_group([(g, x, dW, N, K, M, 3, db, float(alpha), True)])
            if slab is not None:
                dW = None
            if bslab is not None:
                db = None
            return dW, db
        else:
            if alpha != 1.0:
                raise RuntimeError("an output scale is only folded into grouped weight gradients")
            dW = slab if slab is not None else _bw_zeros((N, K), g.device)
            tiles = ((N + 127) // 128) * ((K + 175) // 176)
            gemm_raw(g, x, dW, N, K, M, 1, 1, atomic=True, split_k=_split_k_for(tiles, M))
            if slab is not None:
                dW = None
    if bias_ref is not None and need_b and not bias_done:
        if alpha != 1.0 or p16:
            raise RuntimeError("a bias gradient without its weight gradient is not available for scaled / P16 gradients")
        slab = flat_grad_for(bias_ref)
        db = slab if slab is not None else _bw_zeros((N,), g.device)
        check(lib.vptr_colsum(ptr(g), ptr(db), M, N, stream()), "vptr_colsum")
        if slab is not None:
            db = None
    return dW, db


class _LinearFn(torch.autograd.Function):
    """y = dropout(rowscale * act((x W^T + b) * alpha)) + residual   -- one GEMM launch with a fused epilogue.

    Replaces F.linear call sites (MultiHeadAttentionRPE.py:543-545,687-688; VidHRFormer_modules.py:87-89,190-192)
    and the 1x1 convs of MlpDWBN (:430,:436).  Backward: epilogue-gradient kernel, input-gradient GEMM, weight (+ bias)
    gradient recorded for the grouped end-of-backward launch.

    P16 path (all of K, N multiples of 16, precision 3): operands are P16 tensors staged by DMA.  x_p16: x already is P16 (its
    producer wrote it); otherwise one conversion pass.  out_p16: y is written as P16 (it only feeds another GEMM).  dy_p16: the
    incoming gradient is P16 (its producer wrote it for this node alone; no epilogue terms may need a gradient pass then).
    """

    @staticmethod
    def forward(ctx, x, W, b, residual, rowscale, alpha, act, rs_div, rs_mod, dropout_p, site, x_p16, out_p16, dy_p16, frame_stats=None,
                frame_rows=0):
        _lib.require_cuda(x, W)
        if act == ACT_RELU and (residual is not None or rowscale is not None or dropout_p > 0):
            raise RuntimeError("linear: a ReLU epilogue cannot be combined with residual/rowscale/dropout")
        x, W = _c(x), _c(W)
        M, K = x.shape
        N = W.shape[0]
        use = p16_ok(K, N)
        if (x_p16 or out_p16 or dy_p16) and not use:
            raise RuntimeError("linear: P16 operands need K, N multiples of 16 and the split-bf16 precision (K %d, N %d)" % (K, N))
        y = torch.empty((M, N), device=x.device, dtype=torch.float32)
        pre = torch.empty_like(y) if act == ACT_GELU else None
        res = _c(residual) if residual is not None else None
        ctx.seed = seed_tensor(x.device) if dropout_p > 0 else None
        if use:
            xs = x if x_p16 else to_p16(x)
            Wp, ldw, _, _ = weight_planes_for(W)
            gemm_raw(xs, Wp, y, M, N, K, A_P16, B_P16, lda=K, ldb=ldw, bias=b, alpha=alpha, act=act, Dpre=pre, rowscale=rowscale,
                     rs_div=rs_div, rs_mod=rs_mod, dropout_p=dropout_p, site=site, residual=res, seed=ctx.seed, d_p16=out_p16,
                     frame_stats=frame_stats, frame_rows=frame_rows)
        else:
            if frame_stats is not None:
                raise RuntimeError("linear: frame_stats is an epilogue of the P16 GEMMs only")
            xs = x
            gemm_raw(x, W, y, M, N, K, 0, 0, bias=b, alpha=alpha, act=act, Dpre=pre, rowscale=rowscale, rs_div=rs_div,
                     rs_mod=rs_mod, dropout_p=dropout_p, site=site, residual=res, seed=ctx.seed)
        if act == ACT_RELU and out_p16:
            raise RuntimeError("linear: a ReLU epilogue saves its output for backward and cannot write it as P16")
        ctx.save_for_backward(xs, W, pre if act == ACT_GELU else (y if act == ACT_RELU else None), rowscale)
        ctx.cfg = (alpha, act, rs_div, rs_mod, dropout_p, site, b is not None, residual is not None, use, dy_p16)
        ctx.bias_ref = b   # the parameter (or a view of it) itself: gradient-destination lookup (flat slab by address, else its .grad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, h, rowscale = ctx.saved_tensors
        alpha, act, rs_div, rs_mod, p, site, has_b, has_res, use, dy_p16 = ctx.cfg
        dy = _c(dy)
        M, K = x.shape
        N = W.shape[0]
        # a bare output scale (the q projections' head_dim^-0.5) needs no pass of its own when the weight / bias gradients go
        # through the grouped launch: dx = alpha * (dy . W) and dW = alpha * (dy^T . x) take alpha in their GEMM epilogues
        wslab = grad_dest_for(W) if ctx.needs_input_grad[1] else None
        bslab0 = grad_dest_for(ctx.bias_ref) if (has_b and ctx.needs_input_grad[2]) else None
        fold_alpha = (alpha != 1.0 and act == ACT_NONE and rowscale is None and p == 0 and config.group_wgrads
                      and ctx.needs_input_grad[1] and wslab is not None and (not (has_b and ctx.needs_input_grad[2]) or bslab0 is not None))
        galpha = alpha if fold_alpha else 1.0
        if (act != ACT_NONE or alpha != 1.0 or rowscale is not None or p > 0) and not fold_alpha:
            if act == ACT_RELU and (p > 0 or rowscale is not None):
                raise RuntimeError("ReLU epilogue with dropout/rowscale is not differentiable from its output")
            if dy_p16:
                raise RuntimeError("linear: a P16 gradient cannot pass through an activation / dropout / scale epilogue")
            g = torch.empty_like(dy)
            check(lib.vptr_act_bwd(ptr(dy), ptr(h), ptr(g), M, N, act, alpha, ptr(rowscale), rs_div, rs_mod, p,
                                   ptr(ctx.seed), site, int(use), stream()), "vptr_act_bwd")
        elif use and not dy_p16:
            g = to_p16(dy)
        else:
            g = dy
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), device=dy.device, dtype=torch.float32)
            if use:
                _, _, WT, ldt = weight_planes_for(W)
                gemm_raw(g, WT, dx, M, K, N, A_P16, B_P16, lda=N, ldb=ldt, alpha=galpha)   # dx[M,K] = g[M,N] . W[N,K]
            else:
                gemm_raw(g, W, dx, M, K, N, 0, 1, alpha=galpha)
        dW, db = _linear_param_grads(g, x, W, ctx.bias_ref if has_b else None, ctx.needs_input_grad[1], ctx.needs_input_grad[2], galpha,
                                     p16=use)
        dres = None
        if has_res and ctx.needs_input_grad[3]:
            if dy_p16:
                raise RuntimeError("linear: the residual branch needs the fp32 gradient")
            dres = dy
        return (dx, dW, db, dres) + (None,) * 12


_LinearFn_apply = _direct_apply(_LinearFn)


def linear(x, W, b=None, residual=None, alpha=1.0, act=ACT_NONE, rowscale=None, rs_div=1, rs_mod=1, dropout_p=0.0, site=0,
           x_p16=False, out_p16=False, dy_p16=False, frame_stats=None, frame_rows=0):
    """frame_stats / frame_rows: a zeroed [rows / frame_rows, FRAME_STATS_STRIDE] buffer (frame_stats_buffer) that the GEMM epilogue fills with each
    frame's sum / sum of squares of y, for norm_act(..., raw_stats=...) -- the LayerNorm((F,H,W)) after a 1x1 convolution then needs no
    statistics pass of its own."""
    return _LinearFn_apply(x, W, b, residual, rowscale, float(alpha), int(act), int(rs_div), int(rs_mod), float(dropout_p),
                           int(site), bool(x_p16), bool(out_p16), bool(dy_p16), frame_stats, int(frame_rows))


def frame_stats_ok(rows, HW, F, W=None):
    """can the producers of a conv-FFN tensor [rows, F] (frames of HW rows) deliver its LayerNorm((F,H,W)) statistics themselves?
    (64-row epilogue halves inside one frame; the depthwise kernel's waves inside one frame)"""
    return (config.fused_frame_stats and config.use_p16 and config.gemm_precision == 3 and HW % 64 == 0 and rows % HW == 0 and F % 16 == 0
            and (W is None or (W % 2 == 0 and ((W // 2) * (F // 4)) % 64 == 0)))


_zero_arena = {"buf": None, "off": 0}


class zero_arena:
    """Scope of one model forward in which the small zero-initialised accumulators (frame_stats_buffer) are slices of ONE zero-filled
    tensor: one fill launch per forward instead of one per conv-FFN (16 in the K64 NAR model).  The arena tensor stays alive as long
    as any slice does (autograd saves them); requests beyond its size fall back to their own torch.zeros."""

    def __init__(self, nfloats, device):
        self.n, self.device = int(nfloats), device

    def __enter__(self):
        self.prev = dict(_zero_arena)
        _zero_arena["buf"] = torch.zeros(self.n, device=self.device, dtype=torch.float32) if self.n > 0 else None
        _zero_arena["off"] = 0
        return self

    def __exit__(self, *exc):
        _zero_arena.update(self.prev)
        return False


FRAME_STATS_STRIDE = 32   # include/vptr_hip.h VPTR_FRAME_STATS_STRIDE: one 128-byte line per frame (sum at [0], sum of squares at [1])


def frame_stats_buffer(frames, device):
    """a zeroed [frames, FRAME_STATS_STRIDE] fp32 buffer for one producer / consumer pair (a slice of the forward's zero_arena when one is
    open; inside a graph capture the arena's fill is re-run at every replay)"""
    n = FRAME_STATS_STRIDE * int(frames)
    buf = _zero_arena["buf"]
    if buf is not None and buf.device == torch.device(device) and _zero_arena["off"] + n <= buf.numel():
        off = _zero_arena["off"]
        _zero_arena["off"] = off + n
        return buf[off:off + n].view(frames, FRAME_STATS_STRIDE)
    return torch.zeros((frames, FRAME_STATS_STRIDE), device=device, dtype=torch.float32)


class _MlpFn(torch.autograd.Function):
    """y = dropout(linear2(dropout(GELU(linear1(x))))) + residual -- the transformer MLP (VidHRFormer_modules.py:87-89, 190-192) as ONE
    autograd node on P16 operands.  Forward: two GEMM launches (GELU + saved pre-activation + dropout + P16 output in the first
    epilogue; dropout + residual in the second).  Backward: g2 = dy * mask2 (one pass, P16); dh never exists: linear2's input-gradient
    GEMM applies GELU'(pre) and mask1 in its epilogue (desc.act_grad_src) and writes g1 as P16; dx = g1 . W1; both weight (+ bias)
    gradients go to the grouped end-of-backward launch."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, residual, p, site1, site2, x_p16):
        _lib.require_cuda(x, W1, W2)
        x, W1, W2 = _c(x), _c(W1), _c(W2)
        M, C = x.shape
        Fh, N = W1.shape[0], W2.shape[0]
        if not p16_ok(C, Fh, N):
            raise RuntimeError("mlp: P16 operands need every width to be a multiple of 16 and the split-bf16 precision")
        dev = x.device
        xs = x if x_p16 else to_p16(x)
        ctx.seed = seed_tensor(dev) if p > 0 else None
        h = torch.empty((M, Fh), device=dev, dtype=torch.float32)      # P16
        pre = torch.empty((M, Fh), device=dev, dtype=torch.float32)
        W1p, ld1, _, _ = weight_planes_for(W1)
        gemm_raw(xs, W1p, h, M, Fh, C, A_P16, B_P16, lda=C, ldb=ld1, bias=b1, act=ACT_GELU, Dpre=pre, dropout_p=p, site=site1, seed=ctx.seed,
                 d_p16=True)
        y = torch.empty((M, N), device=dev, dtype=torch.float32)
        res = _c(residual) if residual is not None else None
        W2p, ld2, _, _ = weight_planes_for(W2)
        gemm_raw(h, W2p, y, M, N, Fh, A_P16, B_P16, lda=Fh, ldb=ld2, bias=b2, dropout_p=p, site=site2, residual=res, seed=ctx.seed)
        ctx.save_for_backward(xs, W1, W2, h, pre)
        ctx.cfg = (p, site1, site2, residual is not None)
        ctx.b1_ref, ctx.b2_ref = b1, b2   # the parameters themselves: gradient-destination lookup (grad_dest_for)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, W1, W2, h, pre = ctx.saved_tensors
        p, site1, site2, has_res = ctx.cfg
        dy = _c(dy)
        M, C = xs.shape
        Fh, N = W1.shape[0], W2.shape[0]
        if p > 0:
            g2 = torch.empty_like(dy)
            check(lib.vptr_act_bwd(ptr(dy), None, ptr(g2), M, N, ACT_NONE, 1.0, None, 1, 1, p, ptr(ctx.seed), site2, 1, stream()), "vptr_act_bwd")
        else:
            g2 = to_p16(dy)
        g1 = torch.empty((M, Fh), device=dy.device, dtype=torch.float32)   # P16: dL/d(linear1 output), never materialised as fp32 dh
        _, _, W2T, ld2t = weight_planes_for(W2)
        gemm_raw(g2, W2T, g1, M, Fh, N, A_P16, B_P16, lda=N, ldb=ld2t, act=ACT_GELU, act_grad_src=pre, dropout_p=p, site=site1, seed=ctx.seed,
                 d_p16=True)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, C), device=dy.device, dtype=torch.float32)
            _, _, W1T, ld1t = weight_planes_for(W1)
            gemm_raw(g1, W1T, dx, M, C, Fh, A_P16, B_P16, lda=Fh, ldb=ld1t)
        dW2, db2 = _linear_param_grads(g2, h, W2, ctx.b2_ref, ctx.needs_input_grad[3], ctx.needs_input_grad[4], 1.0, p16=True)
        dW1, db1 = _linear_param_grads(g1, xs, W1, ctx.b1_ref, ctx.needs_input_grad[1], ctx.needs_input_grad[2], 1.0, p16=True)
        dres = dy if (has_res and ctx.needs_input_grad[5]) else None
        return dx, dW1, db1, dW2, db2, dres, None, None, None, None


_MlpFn_apply = _direct_apply(_MlpFn)


def mlp(x, W1, b1, W2, b2, residual=None, dropout_p=0.0, site1=0, site2=0, x_p16=False):
    """The transformer MLP linear2(dropout(GELU(linear1(x)))) (+ dropout, + residual) as one autograd node (see _MlpFn); needs
    P16-eligible widths -- callers fall back to two `linear` calls otherwise."""
    return _MlpFn_apply(x, W1, b1, W2, b2, residual, float(dropout_p), int(site1), int(site2), bool(x_p16))
