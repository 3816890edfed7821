"""Token-grid layout helpers: centre padding / cropping, NCHW <-> token-major transposes."""

import torch

from .. import _lib
from .._lib import check, lib, ptr, stream
from .core import _c, _direct_apply


# ------------------------------------------------------------------------------------------------------------------
# centre padding / cropping of token grids (PadBlock, VidHRFormer_modules.py:538-569)
# ------------------------------------------------------------------------------------------------------------------
class _WindowCopyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, frames, Hs, Ws, Hd, Wd, off_h, off_w):
        x = _c(x)
        C = x.shape[1]
        y = torch.empty((frames * Hd * Wd, C), device=x.device, dtype=torch.float32)
        check(lib.vptr_window_copy(ptr(x), ptr(y), frames, Hs, Ws, Hd, Wd, off_h, off_w, C, stream()), "vptr_window_copy")
        ctx.cfg = (frames, Hs, Ws, Hd, Wd, off_h, off_w)
        return y

    @staticmethod
    def backward(ctx, dy):
        frames, Hs, Ws, Hd, Wd, off_h, off_w = ctx.cfg
        dy = _c(dy)
        C = dy.shape[1]
        dx = torch.empty((frames * Hs * Ws, C), device=dy.device, dtype=torch.float32)
        check(lib.vptr_window_copy(ptr(dy), ptr(dx), frames, Hd, Wd, Hs, Ws, -off_h, -off_w, C, stream()), "vptr_window_copy")
        return dx, None, None, None, None, None, None, None


_WindowCopyFn_apply = _direct_apply(_WindowCopyFn)


def pad_tokens(x, frames, H, W, ws):
    """[frames*H*W, C] -> ([frames*Hp*Wp, C], Hp, Wp): zero centre padding up to multiples of the window size"""
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    return _WindowCopyFn_apply(x, frames, H, W, Hp, Wp, (Hp - H) // 2, (Wp - W) // 2), Hp, Wp


def crop_tokens(x, frames, Hp, Wp, H, W):
    """inverse selection of pad_tokens"""
    return _WindowCopyFn_apply(x, frames, Hp, Wp, H, W, -((Hp - H) // 2), -((Wp - W) // 2))


# ------------------------------------------------------------------------------------------------------------------
# layout
# ------------------------------------------------------------------------------------------------------------------
class _ToTokensFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):  # x [B, C, H, W] -> [B*H*W, C]
        _lib.require_cuda(x)
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty((B * H * W, C), device=x.device, dtype=torch.float32)
        check(lib.vptr_nchw_to_tokens(ptr(x), ptr(y), B, C, H * W, stream()), "vptr_nchw_to_tokens")
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = _c(dy)
        dx = torch.empty((B, C, H, W), device=dy.device, dtype=torch.float32)
        check(lib.vptr_tokens_to_nchw(ptr(dy), ptr(dx), B, C, H * W, 0, stream()), "vptr_tokens_to_nchw")
        return dx


_ToTokensFn_apply = _direct_apply(_ToTokensFn)


class _FromTokensFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, B, C, H, W, relu):  # [B*H*W, C] -> [B, C, H, W] (+ReLU)
        x = _c(x)
        y = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32)
        check(lib.vptr_tokens_to_nchw(ptr(x), ptr(y), B, C, H * W, int(relu), stream()), "vptr_tokens_to_nchw")
        ctx.cfg = (B, C, H, W, relu)
        if relu:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W, relu = ctx.cfg
        dy = _c(dy)
        dx = torch.empty((B * H * W, C), device=dy.device, dtype=torch.float32)
        if relu:
            (y,) = ctx.saved_tensors
            check(lib.vptr_nchw_to_tokens_masked(ptr(dy), ptr(y), ptr(dx), B, C, H * W, stream()), "vptr_nchw_to_tokens_masked")
        else:
            check(lib.vptr_nchw_to_tokens(ptr(dy), ptr(dx), B, C, H * W, stream()), "vptr_nchw_to_tokens")
        return dx, None, None, None, None, None


_FromTokensFn_apply = _direct_apply(_FromTokensFn)


def nchw_to_tokens(x):
    return _ToTokensFn_apply(x)


def tokens_to_nchw(x, B, C, H, W, relu=False):
    return _FromTokensFn_apply(x, int(B), int(C), int(H), int(W), bool(relu))
