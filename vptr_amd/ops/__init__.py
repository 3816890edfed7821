"""Host-side operators of the VPTR hot path: thin autograd wrappers over the C-ABI HIP kernels.

Every forward/backward here is a call into libvptr_hip.so (vptr_amd/_lib.py); torch is used for device memory,
streams and autograd bookkeeping only.  Activations are token-major, channel-last 2-D tensors [rows, C] with
rows = (n, t, h, w) flattened -- the reference's window partition and (T, N*HW, C) permutes never materialise.

Split by concern (round 6): core (configuration, seeds, raw GEMM, P16 format) / planes (weight-plane stores) / wgrad (launch planning of the
deferred grouped weight gradients) / grads (gradient destinations: slabs, arena, autograd hand-off) / linear / norm / attention / convffn /
layout / conv / losses (operator wrappers).  `vptr_amd.ops.<name>` keeps resolving every name the single module had.
"""
from .core import (  # noqa: F401
    ACT_NONE, ACT_GELU, ACT_RELU, ACT_LRELU, PAD_MODES, _Config, config, set_deterministic, _direct_apply, _seed_state, _seed_scope,
    _dev_key, _master_seed, new_seed_scope, seed_tensor, manual_seed, DROPPATH_SITE0, droppath_scales, _c, gemm_raw, _Profiling,
    profiling, gemm_nfn, A_P16, B_P16, A_P16T, B_P16T, p16_ok, to_p16, _AsP16Fn, _AsP16Fn_apply, as_p16, p16_decode,
)
from .wgrad import (  # noqa: F401
    _wgrad_q, defer_wgrad, take_wgrads, requeue_wgrads, discard_wgrads, _reduce_q, defer_partial_reduce, flush_partial_reduces,
    _wgrad_hold, hold_wgrads, _wgrad_side, _flush_wgrads_side, join_wgrad_stream, _auto_flush_wgrads, _pin_pool, _pin_pool_small,
    _wgrad_tune, _graph_keepalive, _graph_reserve, _upload_stats, reserve_graph_staging, _to_device_async, _wgrad_tune_book, _wgrad_tune_decide, _TUNE_SAMPLES, wgrad_tune_open,
    wgrad_tune_settle, plan_wgrad_launches, _launch_wgrad_group, convt_weight_grads, flush_wgrads, _split_k_for,
)
from .grads import (  # noqa: F401
    _flat_slabs, register_flat_slab, unregister_flat_slabs, unregister_flat_slab, flat_grad_for, _grad_arenas, _storage_refs, _GradArena,
    _register_grad_arena, _arm_grad_arena, _arena_grad_for, _engine_accumulates_into, _acc_nodes, _drop_acc_nodes, _loose_grad_for,
    grad_dest_for, _bw_blocks, _BW_BLOCK, _bw_zeros,
)
from .planes import (  # noqa: F401
    WeightPlanes, _wplane_stores, _wplane_cache, _WPLANE_CACHE_BYTES, register_weight_planes, ensure_module_planes,
    invalidate_weight_planes, weight_planes_for, linear_weights_of,
)
from .linear import (  # noqa: F401
    _linear_param_grads, _LinearFn, _LinearFn_apply, linear, frame_stats_ok, _zero_arena, zero_arena, FRAME_STATS_STRIDE, frame_stats_buffer, _MlpFn,
    _MlpFn_apply, mlp,
)
from .norm import (  # noqa: F401
    _LayerNormFn, _LayerNormFn_apply, layernorm, _AddRowTabFn, _AddRowTabFn_apply, add_rowtab,
)
from .attention import (  # noqa: F401
    _winattn_workspace, _WinAttnFn, _WinAttnFn_apply, window_attention, _TAttnFn, _TAttnFn_apply, temporal_attention, KVGradAccum,
    _ProjAttnFn, _ProjAttnFn_apply, _proj_attention, proj_window_attention, proj_temporal_attention, _TSAttnFn,
    temporal_spatial_window_attention,
)
from .convffn import (  # noqa: F401
    _norm_act_backward, _NormActFn, _NormActFn_apply, norm_act, _DWConvFn, _DWConvFn_apply, dwconv3x3, norm_dwconv_ok, _NormDWConvFn,
    _NormDWConvFn_apply, norm_dwconv3x3,
)
from .layout import (  # noqa: F401
    _WindowCopyFn, _WindowCopyFn_apply, pad_tokens, crop_tokens, _ToTokensFn, _ToTokensFn_apply, _FromTokensFn, _FromTokensFn_apply,
    nchw_to_tokens, tokens_to_nchw,
)
from .conv import (  # noqa: F401
    SubpixelWeights, conv_weight_as_gemm_b, in_flat_slab, weights_cacheable, frozen_weights, conv_nhwc, split_planes,
    conv_weight_as_planes, conv_nhwc_planes, wino_ok, wino_filter, wino_buffers, wino_in, wino_gemm, wino_out, wino_fused_ok, wino_out_in, wino_conv3x3,
    wino_resnet_blocks, _Conv2dNHWCFn, conv2d_nhwc, _Conv7InFn, conv7_in, _Conv7OutFn, conv7_out, bn_fold,
)
from .losses import (  # noqa: F401
    _MseGdlFn, mse_gdl, _NceFn, nce_loss,
)
from . import core, planes, wgrad, grads, linear, norm, attention, convffn, layout, conv, losses  # noqa: F401,E402
from .._lib import lib  # noqa: F401,E402  (tools patch ops.lib entry points)
