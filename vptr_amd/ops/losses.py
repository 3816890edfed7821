"""Fused loss kernels: MSE + GDL, BiPatchNCE."""

import torch

from .. import _lib
from .._lib import check, lib, ptr, stream
from .core import _c


# ------------------------------------------------------------------------------------------------------------------
# losses of the train steps as plain kernel launches (csrc/losses.hip): no ATen reductions (-> no memset nodes) in a captured step
# ------------------------------------------------------------------------------------------------------------------
class _MseGdlFn(torch.autograd.Function):
    """(MSELoss()(gt, pred), GDL(alpha=1)(gt, pred)) of model/criterion.py:105-132,134-204 in two launches (partials + fixed-order
    sum); backward: ONE elementwise launch that takes both upstream gradients as device scalars."""

    @staticmethod
    def forward(ctx, pred, gt):
        _lib.require_cuda(pred, gt)
        if pred.shape != gt.shape or pred.dim() < 3:
            raise RuntimeError("mse_gdl: pred %s and gt %s must share a [..., H, W] shape" % (tuple(pred.shape), tuple(gt.shape)))
        pred, gt = _c(pred), _c(gt)
        H, W = pred.shape[-2], pred.shape[-1]
        planes = pred.numel() // (H * W)
        scratch = torch.empty(planes * ((H + 15) // 16) * 3, device=pred.device, dtype=torch.float32)
        mse = torch.empty((), device=pred.device, dtype=torch.float32)
        gdl = torch.empty((), device=pred.device, dtype=torch.float32)
        check(lib.vptr_mse_gdl_fwd(ptr(pred), ptr(gt), ptr(scratch), ptr(mse), ptr(gdl), planes, H, W, stream()), "vptr_mse_gdl_fwd")
        ctx.save_for_backward(pred, gt)
        return mse, gdl

    @staticmethod
    def backward(ctx, g_mse, g_gdl):
        pred, gt = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            raise RuntimeError("mse_gdl: the target (gt) is data; it has no gradient path here")
        H, W = pred.shape[-2], pred.shape[-1]
        dpred = torch.empty_like(pred)
        check(lib.vptr_mse_gdl_bwd(ptr(pred), ptr(gt), ptr(_c(g_mse)), ptr(_c(g_gdl)), ptr(dpred), pred.numel() // (H * W), H, W, stream()),
              "vptr_mse_gdl_bwd")
        return dpred, None


def mse_gdl(pred, gt):
    """-> (mse, gdl) 0-dim device tensors; replaces MSELoss()(gt, pred) and GDL(alpha=1)(gt, pred) (no temporal weights, no norm_dim)"""
    return _MseGdlFn.apply(pred, gt)


class _NceFn(torch.autograd.Function):
    """BiPatchNCE(temperature)(F.normalize(g, dim=channel), F.normalize(p, dim=channel)) of train_NAR.py:81-84 / criterion.py:206-259
    on token-major projections g, p [frames * L, C] (g: ground-truth features, p: predicted features): 4 launches forward, 1 backward."""

    @staticmethod
    def forward(ctx, g, p, frames, L, temperature):
        _lib.require_cuda(g, p)
        g, p = _c(g), _c(p)
        R, C = g.shape
        if R != frames * L or p.shape != g.shape:
            raise RuntimeError("nce_loss: g %s / p %s do not hold %d frames of %d patches" % (tuple(g.shape), tuple(p.shape), frames, L))
        scratch = torch.empty(frames * L * (4 + L) + frames, device=g.device, dtype=torch.float32)
        loss = torch.empty((), device=g.device, dtype=torch.float32)
        check(lib.vptr_nce_fwd(ptr(g), ptr(p), ptr(scratch), ptr(loss), frames, L, C, temperature, stream()), "vptr_nce_fwd")
        ctx.save_for_backward(g, p, scratch)
        ctx.cfg = (frames, L, temperature)
        return loss

    @staticmethod
    def backward(ctx, gout):
        g, p, scratch = ctx.saved_tensors
        frames, L, temperature = ctx.cfg
        dg, dp = torch.empty_like(g), torch.empty_like(p)
        check(lib.vptr_nce_bwd(ptr(g), ptr(p), ptr(scratch), ptr(_c(gout)), ptr(dg), ptr(dp), frames, L, g.shape[1], temperature, stream()),
              "vptr_nce_bwd")
        return dg, dp, None, None, None


def nce_loss(g_tok, p_tok, frames, L, temperature=1.0):
    """bidirectional patch-wise contrastive loss of the un-normalised projector outputs (see _NceFn)"""
    return _NceFn.apply(g_tok, p_tok, int(frames), int(L), float(temperature))
