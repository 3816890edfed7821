"""Conv-FFN pieces: normalise + activation (BatchNorm2d / LayerNorm((F,H,W))), depthwise 3x3, and their fused form."""

import torch

from .._lib import check, lib, ptr, stream
from .core import ACT_GELU, _c, _direct_apply, config, seed_tensor
from .wgrad import defer_partial_reduce
from .grads import _bw_zeros, flat_grad_for
from .linear import frame_stats_ok


# ------------------------------------------------------------------------------------------------------------------
# conv-FFN pieces
# ------------------------------------------------------------------------------------------------------------------
def _norm_act_backward(dy, x, w, b, mean, rstd, rowscale, HW, per_col, act, const_stats, p, seed, site, rs_div, rs_mod, dx_p16):
    """backward of y = dropout(act(norm(x) * w + b)) (* rowscale): (dx, dw, db); dw / db are None when the affine gradients were accumulated
    in place (flat gradient slab) -- directly or through the backward pass's deferred partial-sum reduction"""
    rows, F = x.shape
    dx = torch.empty_like(x)
    sw, sb = flat_grad_for(w), flat_grad_for(b)   # accumulate straight into the flat gradient slab when both live there
    in_slab = sw is not None and sb is not None
    dw, db = (sw, sb) if in_slab else (_bw_zeros(w.shape, w.device), _bw_zeros(b.shape, b.device))
    frames = rows // HW
    scratch = torch.empty((max(2 * F, 2 * frames * (1 + 4 * ((HW * F // 4 + 255) // 256))),), device=x.device, dtype=torch.float32)
    ncoop = lib.vptr_norm_act_bwd_coop_partials(rows, F, HW) if (in_slab and config.defer_ln_param_grads and not per_col and config.norm_coop) else 0
    if ncoop > 0:
        # LayerNorm((F,H,W)): both phases in ONE cooperative pass over (dy, x) -- the workgroups of a 10-frame chunk exchange the frames' sums through
        # a zeroed workspace line per frame (csrc/norm.hip norm_act_bwd_coop_kernel); affine gradients as per-chunk partial sums like below
        part = torch.empty((ncoop, 2, HW * F), device=x.device, dtype=torch.float32)
        ws = _bw_zeros(((frames + 1) * 32,), x.device)
        check(lib.vptr_norm_act_bwd_coop(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(w), ptr(b), ptr(dx), ptr(ws), rows, F, HW, act, p, ptr(seed), site,
                                         ptr(rowscale), rs_div, rs_mod, int(dx_p16), ptr(part), stream()), "vptr_norm_act_bwd_coop")
        defer_partial_reduce(part, sw, sb, ncoop, HW * F)
        return dx, None, None
    nparts = lib.vptr_norm_act_bwd_partials(rows, F, HW, int(per_col)) if (in_slab and config.defer_ln_param_grads) else 0
    if nparts > 0:
        # affine gradients with an in-place destination: per-chunk partial sums, added by the backward pass's one reduction launch
        part = torch.empty((nparts, 2, HW * F), device=x.device, dtype=torch.float32)
        check(lib.vptr_norm_act_bwd_deferred(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(w), ptr(b), ptr(dx), ptr(scratch), rows, F, HW,
                                             act, int(const_stats), p, ptr(seed), site, ptr(rowscale), rs_div, rs_mod, int(dx_p16),
                                             ptr(part), stream()), "vptr_norm_act_bwd_deferred")
        defer_partial_reduce(part, sw, sb, nparts, HW * F)
    else:
        check(lib.vptr_norm_act_bwd(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(w), ptr(b), ptr(dx), ptr(dw), ptr(db),
                                    ptr(scratch), rows, F, HW, int(per_col), act, int(const_stats), p,
                                    ptr(seed), site, ptr(rowscale), rs_div, rs_mod, int(dx_p16),
                                    stream()), "vptr_norm_act_bwd")
    if in_slab:
        dw = db = None
    return dx, dw, db


class _NormActFn(torch.autograd.Function):
    """y = dropout(act(norm(x) * w + b)) on channel-last [rows, F].

    mode 'bn'   : BatchNorm2d semantics (VidHRFormer_modules.py:397-419 with AR_model=False); batch statistics when
                  `training`, running statistics otherwise; running stats updated in place (momentum 0.1, unbiased var).
    mode 'ln'   : LayerNorm((F,H,W)) per frame; w, b given channel-last as [HW, F].
    """

    @staticmethod
    def forward(ctx, x, w, b, running_mean, running_var, mode, HW, training, act, eps, p, site, momentum, rowscale, rs_div,
                rs_mod, residual, out_p16, dx_p16, num_batches_tracked, raw_stats=None):
        x, w, b = _c(x), _c(w), _c(b)
        residual = _c(residual) if residual is not None else None
        rows, F = x.shape
        dev = x.device
        per_col = mode == "bn"
        const_stats = False
        if per_col:
            if training:
                mean = torch.empty((F,), device=dev, dtype=torch.float32)
                var = torch.empty_like(mean)
                nchunk = (rows + 255) // 256
                scratch = torch.empty((2 * F * nchunk,), device=dev, dtype=torch.float32)
                rstd = torch.empty_like(mean)
                # batch statistics + BatchNorm2d's running-statistics / num_batches_tracked bookkeeping in one launch pair
                check(lib.vptr_colstats_running(ptr(x), ptr(mean), ptr(var), ptr(rstd), eps, ptr(scratch), rows, F, ptr(running_mean),
                                                ptr(running_var), momentum, ptr(num_batches_tracked), stream()), "vptr_colstats_running")
            else:
                mean, var = running_mean, running_var
                rstd = torch.rsqrt(var + eps)
                const_stats = True
        else:
            frames = rows // HW
            mean = torch.empty((frames,), device=dev, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            if raw_stats is None:
                var = torch.empty_like(mean)
                check(lib.vptr_groupstats(ptr(x), ptr(mean), ptr(var), ptr(rstd), eps, frames, HW * F, stream()), "vptr_groupstats")
            # else: x's producer accumulated the per-frame sums; the normalise kernel derives mean / rstd and writes them for backward
        if raw_stats is not None and per_col:
            raise RuntimeError("norm_act: raw_stats belong to the LayerNorm((F,H,W)) mode")
        y = torch.empty_like(x)
        ctx.seed = seed_tensor(dev) if p > 0 else None
        check(lib.vptr_norm_act_fwd(ptr(x), ptr(mean), ptr(rstd), ptr(w), ptr(b), ptr(y), rows, F, HW, int(per_col), act, p,
                                    ptr(ctx.seed), site, ptr(rowscale), rs_div, rs_mod,
                                    ptr(residual), int(out_p16), ptr(raw_stats), eps, stream()), "vptr_norm_act_fwd")
        ctx.save_for_backward(x, w, b, mean, rstd, rowscale)
        ctx.cfg = (HW, per_col, act, const_stats, p, site, rs_div, rs_mod, residual is not None, dx_p16)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, mean, rstd, rowscale = ctx.saved_tensors
        HW, per_col, act, const_stats, p, site, rs_div, rs_mod, has_res, dx_p16 = ctx.cfg
        dy = _c(dy)
        dx, dw, db = _norm_act_backward(dy, x, w, b, mean, rstd, rowscale, HW, per_col, act, const_stats, p, ctx.seed, site, rs_div, rs_mod, dx_p16)
        dres = dy if has_res else None
        return dx, dw, db, None, None, None, None, None, None, None, None, None, None, None, None, None, dres, None, None, None, None


_NormActFn_apply = _direct_apply(_NormActFn)


def norm_act(x, w, b, mode, HW, training, running_mean=None, running_var=None, act=ACT_GELU, eps=1e-5, dropout_p=0.0, site=0,
             momentum=0.1, rowscale=None, rs_div=1, rs_mod=1, residual=None, out_p16=False, dx_p16=False, num_batches_tracked=None,
             raw_stats=None):
    """y = rowscale * dropout(act(norm(x)*w + b)) + residual  (one elementwise pass; see _NormActFn).
    out_p16: y is written as a P16 tensor (it only feeds a GEMM); dx_p16: the gradient w.r.t. x is returned as a P16 tensor (x is
    the output of a linear(..., dy_p16=True) and nothing else)."""
    return _NormActFn_apply(x, w, b, running_mean, running_var, mode, int(HW), bool(training), int(act), float(eps),
                            float(dropout_p), int(site), float(momentum), rowscale, int(rs_div), int(rs_mod), residual,
                            bool(out_p16), bool(dx_p16), num_batches_tracked, raw_stats)


class _DWConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w9, b, frames, H, W, frame_stats=None):
        x, w9 = _c(x), _c(w9)
        F = x.shape[1]
        y = torch.empty_like(x)
        check(lib.vptr_dwconv3x3_fwd(ptr(x), ptr(w9), ptr(b), ptr(y), frames, H, W, F, ptr(frame_stats), stream()), "vptr_dwconv3x3_fwd")
        ctx.save_for_backward(x, w9)
        ctx.cfg = (frames, H, W)
        ctx.bias_ref = b.detach() if b is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w9 = ctx.saved_tensors
        frames, H, W = ctx.cfg
        dy = _c(dy)
        F = x.shape[1]
        dx = torch.empty_like(x)
        sw, sb = flat_grad_for(w9), flat_grad_for(ctx.bias_ref)
        in_slab = sw is not None and sb is not None
        dw9 = sw if in_slab else _bw_zeros(w9.shape, w9.device)
        db = sb if in_slab else _bw_zeros((F,), x.device)
        check(lib.vptr_dwconv3x3_bwd(ptr(dy), ptr(x), ptr(w9), ptr(dx), ptr(dw9), ptr(db), frames, H, W, F, stream()),
              "vptr_dwconv3x3_bwd")
        if in_slab:
            dw9 = db = None
        return dx, dw9, db, None, None, None, None


_DWConvFn_apply = _direct_apply(_DWConvFn)


def dwconv3x3(x, weight, bias, frames, H, W, frame_stats=None):
    """Depthwise 3x3 (pad 1) on channel-last x [frames*H*W, F]; weight is the PyTorch parameter [F,1,3,3].  frame_stats: see linear."""
    F = x.shape[1]
    w9 = weight.reshape(F, 9).t().contiguous()  # tap-major [9, F] for coalesced reads
    return _DWConvFn_apply(x, w9, bias, int(frames), int(H), int(W), frame_stats)


def norm_dwconv_ok(rows, HW, F, H, W):
    """can the conv-FFN's first normalisation + activation run inside the depthwise kernel's load path? (LayerNorm((F,H,W)) statistics
    delivered by fc1's epilogue, x pairs of a channel quad in adjacent lanes, whole waves inside one frame)"""
    W2 = W // 2
    # where it pays (profiles/r06_cfg4 / cfg5_kernel_stats.md): maps of at most 64 pixels (16 KB LDS slabs: four workgroups per CU) and hidden tensors
    # that stay in the 256 MB Infinity Cache -- on 16 x 16 maps (KTH 128 x 128: 164 us against 48 + 48) and on 251 MB tensors (BAIR, 29 frames per
    # clip: 175 against 65 + 86) the two-launch form is faster; VPTR_FUSED_NORM_DW=2 forces the fused launch wherever it is valid
    if config.fused_norm_dwconv_mode != 2 and (H * W > 64 or rows * F * 4 > (128 << 20)):
        return False
    return (config.fused_norm_dwconv and not config.deterministic and frame_stats_ok(rows, HW, F, W) and W % 2 == 0 and W2 >= 1 and 16 % W2 == 0
            and H * W == HW)


class _NormDWConvFn(torch.autograd.Function):
    """y = dw3x3(act(LayerNorm((F,H,W))(x))) with the normalisation + activation applied in the depthwise kernel's load path
    (VidHRFormer_modules.py:430-434).  x: fc1's raw output, raw_stats: its per-frame sums (fc1's epilogue).  The activated tensor is kept
    only as an fp16 side copy -- the x operand of the depthwise weight gradient; the backward pass runs the depthwise data / weight
    gradients and then the ordinary two-phase backward of the normalisation on (dy_act, x)."""

    @staticmethod
    def forward(ctx, x, aw, ab, w9, b9, frames, H, W, raw_stats, frame_stats, act, eps, dx_p16):
        x, aw, ab, w9 = _c(x), _c(aw), _c(ab), _c(w9)
        rows, F = x.shape
        y = torch.empty_like(x)
        ah = torch.empty((rows, F), device=x.device, dtype=torch.float16) if any(ctx.needs_input_grad) else None   # no-grad forward: not written
        mean = torch.empty((frames,), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        check(lib.vptr_dwconv3x3_norm_fwd(ptr(x), ptr(raw_stats), ptr(aw), ptr(ab), eps, act, ptr(w9), ptr(b9), ptr(y), ptr(ah), ptr(mean), ptr(rstd),
                                          frames, H, W, F, ptr(frame_stats), stream()), "vptr_dwconv3x3_norm_fwd")
        ctx.save_for_backward(x, aw, ab, w9, mean, rstd, ah)
        ctx.cfg = (frames, H, W, act, dx_p16)
        ctx.bias_ref = b9.detach() if b9 is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, aw, ab, w9, mean, rstd, ah = ctx.saved_tensors
        frames, H, W, act, dx_p16 = ctx.cfg
        dy = _c(dy)
        F = x.shape[1]
        da = torch.empty_like(x)
        sw, sb = flat_grad_for(w9), flat_grad_for(ctx.bias_ref)
        in_slab = sw is not None and sb is not None
        dw9 = sw if in_slab else _bw_zeros(w9.shape, w9.device)
        db9 = sb if in_slab else _bw_zeros((F,), x.device)
        check(lib.vptr_dwconv3x3_bwd_xh(ptr(dy), ptr(ah), ptr(w9), ptr(da), ptr(dw9), ptr(db9), frames, H, W, F, stream()), "vptr_dwconv3x3_bwd_xh")
        if in_slab:
            dw9 = db9 = None
        dx, daw, dab = _norm_act_backward(da, x, aw, ab, mean, rstd, None, H * W, False, act, False, 0.0, None, 0, 1, 1, dx_p16)
        return dx, daw, dab, dw9, db9, None, None, None, None, None, None, None, None


_NormDWConvFn_apply = _direct_apply(_NormDWConvFn)


def norm_dwconv3x3(x, aff_w, aff_b, weight, bias, frames, H, W, raw_stats, frame_stats=None, act=ACT_GELU, eps=1e-5, dx_p16=False):
    """dw3x3(act(LayerNorm((F,H,W))(x))) in one launch (norm_dwconv_ok() decides where): aff_w / aff_b channel-last [H*W, F], weight the
    PyTorch depthwise parameter [F,1,3,3], raw_stats the per-frame sums of x from its producer, frame_stats as in dwconv3x3."""
    F = x.shape[1]
    w9 = weight.reshape(F, 9).t().contiguous()
    return _NormDWConvFn_apply(x, aff_w, aff_b, w9, bias, int(frames), int(H), int(W), raw_stats, frame_stats, int(act), float(eps), bool(dx_p16))
