"""LayerNorm(C) with its fused positional add."""

import torch

from .. import _lib
from .._lib import check, lib, ptr, stream
from .core import _c, _direct_apply, config
from .wgrad import defer_partial_reduce
from .grads import _bw_zeros, flat_grad_for, grad_dest_for


# ------------------------------------------------------------------------------------------------------------------
# LayerNorm(C) (+ fused positional add)
# ------------------------------------------------------------------------------------------------------------------
class _LayerNormFn(torch.autograd.Function):
    """y = LN(x) [, y2 = y + tab[...]] [, xr = x].  The pass-through output xr is x itself: a sub-layer that uses it as its
    residual sends the residual gradient back through THIS node, where it is added inside the LayerNorm-backward kernel
    (dx_add) instead of by an autograd accumulation pass."""

    @staticmethod
    def forward(ctx, x, gamma, beta, tab, tab_div, tab_mod, eps, passthrough, out_p16, tab_grad_to=None):
        _lib.require_cuda(x)
        ctx.set_materialize_grads(False)  # an unused output (e.g. y when only y + tab is consumed) arrives as None, not zeros
        x = _c(x)
        rows, C = x.shape
        y = torch.empty_like(x)
        y2 = torch.empty_like(x) if tab is not None else None
        mean = torch.empty((rows,), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        tab_c = _c(tab) if tab is not None else None
        check(lib.vptr_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(y2), ptr(tab_c), tab_div, tab_mod, ptr(mean),
                                     ptr(rstd), rows, C, eps, int(out_p16), stream()), "vptr_layernorm_fwd")
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.beta_ref = beta.detach()
        ctx.tab = (tab is not None, tab_div, tab_mod, tuple(tab.shape) if tab is not None else None)
        ctx.tab_ref = tab_grad_to if tab_grad_to is not None else tab   # whose gradient destination receives the table gradient
        ctx.passthrough = passthrough
        outs = (y,) if tab is None else (y, y2)
        if passthrough:
            outs = outs + (x,)   # an input returned as-is: autograd makes it an identity output of this node
        return outs[0] if len(outs) == 1 else outs

    @staticmethod
    def backward(ctx, *grads):
        x, gamma, mean, rstd = ctx.saved_tensors
        has_tab, tab_div, tab_mod, tab_shape = ctx.tab
        rows, C = x.shape
        grads = list(grads)
        dres = grads.pop() if ctx.passthrough else None
        dy = grads[0]
        dy2 = grads[1] if has_tab else None
        dy2 = _c(dy2) if dy2 is not None else None
        dres = _c(dres) if dres is not None else None
        if dy is None:  # only the position-added output was consumed
            if dy2 is None:
                return (dres,) + (None,) * 9
            k1, k2 = dy2, None
        else:
            k1, k2 = _c(dy), dy2
        dx = torch.empty_like(x)
        sg, sb = flat_grad_for(gamma), flat_grad_for(ctx.beta_ref)
        in_slab = sg is not None and sb is not None
        nparts = lib.vptr_layernorm_bwd_partials(rows, C) if (in_slab and config.defer_ln_param_grads) else 0
        if nparts > 0:
            # in-place destination: per-workgroup partial sums now, one reduction launch for all LayerNorms at the end of backward
            part = torch.empty((nparts, 2, C), device=x.device, dtype=torch.float32)
            check(lib.vptr_layernorm_bwd_deferred(ptr(k1), ptr(k2), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), rows, C,
                                                  ptr(dres), ptr(part), stream()), "vptr_layernorm_bwd_deferred")
            defer_partial_reduce(part, sg, sb, nparts, C)
            dgamma = dbeta = None
        else:
            dgamma = sg if in_slab else _bw_zeros(gamma.shape, gamma.device)
            dbeta = sb if in_slab else _bw_zeros(gamma.shape, gamma.device)
            check(lib.vptr_layernorm_bwd(ptr(k1), ptr(k2), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dgamma),
                                         ptr(dbeta), rows, C, ptr(dres), stream()), "vptr_layernorm_bwd")
            if in_slab:
                dgamma = dbeta = None
        dtab = None
        if has_tab and ctx.needs_input_grad[3] and dy2 is not None:
            dst = grad_dest_for(ctx.tab_ref) if ctx.tab_ref is not None else None
            if dst is not None and dst.numel() == tab_mod * C:
                # the table is (a view of, or an affine image of) a parameter with an in-place gradient destination: accumulate there
                # -- no zero-filled temporary, no autograd add per call site (8 decoder blocks x 2 share frame_queries)
                check(lib.vptr_rowmod_sum(ptr(dy2), ptr(dst), rows, C, tab_div, tab_mod, stream()), "vptr_rowmod_sum")
            else:
                dtab = torch.zeros((tab_mod, C), device=x.device, dtype=torch.float32)
                check(lib.vptr_rowmod_sum(ptr(dy2), ptr(dtab), rows, C, tab_div, tab_mod, stream()), "vptr_rowmod_sum")
                dtab = dtab.reshape(tab_shape)
        return dx, dgamma, dbeta, dtab, None, None, None, None, None, None


_LayerNormFn_apply = _direct_apply(_LayerNormFn)


def layernorm(x, gamma, beta, tab=None, tab_div=1, tab_mod=1, eps=1e-5, passthrough=False, out_p16=False, tab_grad_to=None):
    """y = LN(x) [, y2 = y + tab[(row // tab_div) % tab_mod]] [, xr]; x [rows, C]; tab [tab_mod, C].
    passthrough=True appends xr (= x, for use as the residual of the sub-layer this LayerNorm feeds; see _LayerNormFn).
    out_p16: y and y2 are written as P16 tensors (they only feed GEMMs; their gradients arrive as ordinary fp32).
    tab_grad_to: a tensor of tab's size whose gradient IS tab's gradient (tab = tab_grad_to + constants): when it has an in-place
    gradient destination (flat slab / .grad) the table gradient is accumulated there instead of being handed to autograd."""
    return _LayerNormFn_apply(x, gamma, beta, tab, int(tab_div), int(tab_mod), float(eps), bool(passthrough), bool(out_p16), tab_grad_to)


class _AddRowTabFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, tab, div, mod):
        x, tab = _c(x), _c(tab)
        rows, C = x.shape
        y = torch.empty_like(x)
        check(lib.vptr_add_rowtab(ptr(x), ptr(tab), ptr(y), rows, C, div, mod, stream()), "vptr_add_rowtab")
        ctx.cfg = (div, mod, tuple(tab.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        div, mod, tshape = ctx.cfg
        dtab = None
        if ctx.needs_input_grad[1]:
            dy = _c(dy)
            dtab = torch.zeros((mod, dy.shape[1]), device=dy.device, dtype=torch.float32)
            check(lib.vptr_rowmod_sum(ptr(dy), ptr(dtab), dy.shape[0], dy.shape[1], div, mod, stream()), "vptr_rowmod_sum")
            dtab = dtab.reshape(tshape)
        return dy, dtab, None, None


_AddRowTabFn_apply = _direct_apply(_AddRowTabFn)


def add_rowtab(x, tab, div, mod):
    return _AddRowTabFn_apply(x, tab, int(div), int(mod))
