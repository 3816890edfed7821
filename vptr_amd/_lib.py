"""ctypes binding of libvptr_hip.so (the C ABI declared in include/vptr_hip.h).

The product path has no CPU or PyTorch fallback: if the shared library is missing or a symbol cannot be
resolved, importing this module raises, and every op raises RuntimeError with the library's own message when a
call returns non-zero.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VPTR_HIP_LIB") or os.path.join(_HERE, "libvptr_hip.so")   # VPTR_HIP_LIB: an instrumented build of the same sources (tools/)

c_void_p, c_int, c_float, c_int64, c_uint32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64, ctypes.c_uint32


class GemmDesc(ctypes.Structure):
    """Mirror of `vptr_gemm_desc` (include/vptr_hip.h)."""
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("D", c_void_p), ("Dpre", c_void_p),
        ("lda", c_int64), ("ldb", c_int64), ("ldd", c_int64),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("a_mode", c_int), ("b_mode", c_int),
        ("precision", c_int), ("split_k", c_int), ("atomic", c_int),
        ("colscale", c_void_p), ("bias", c_void_p),
        ("alpha", c_float), ("act", c_int),
        ("rowscale", c_void_p), ("rs_div", c_int), ("rs_mod", c_int),
        ("dropout_p", c_float), ("seed_dev", c_void_p), ("site", c_uint32),
        ("residual", c_void_p), ("ldr", c_int64),
        ("act_after", c_int),
        ("conv_IH", c_int), ("conv_IW", c_int), ("conv_Cin", c_int), ("conv_OH", c_int), ("conv_OW", c_int),
        ("conv_KH", c_int), ("conv_KW", c_int), ("conv_stride", c_int), ("conv_pad", c_int), ("conv_pad_mode", c_int),
        ("conv_transposed", c_int),
        ("a_rowsum", c_void_p),
        ("batch", c_int), ("ksegs", c_int),
        ("A_x1", c_void_p), ("A_x2", c_void_p), ("B_x1", c_void_p), ("B_x2", c_void_p), ("D_x1", c_void_p), ("D_x2", c_void_p),
        ("bias_x1", c_void_p), ("bias_x2", c_void_p), ("alpha_x1", c_float), ("alpha_x2", c_float),
        ("D_planes", c_void_p),
        ("d_p16", c_int),
        ("act_grad_src", c_void_p),
        ("frame_stats", c_void_p), ("frame_rows", c_int),
        ("d_transposed", c_int),
        ("d_row_w", c_int), ("d_row_off", c_int),
        ("batch_accum", c_int),
        ("batch_stride_a", c_int64), ("batch_stride_b", c_int64), ("batch_stride_d", c_int64),
    ]


class ReduceEntry(ctypes.Structure):
    """Mirror of `vptr_reduce_entry` (include/vptr_hip.h)."""
    _fields_ = [("part", c_void_p), ("dst0", c_void_p), ("dst1", c_void_p), ("nparts", c_int), ("C", c_int)]


class WPlaneEntry(ctypes.Structure):
    """Mirror of `vptr_wplane_entry` (include/vptr_hip.h)."""
    _fields_ = [("W", c_void_p), ("Wp", c_void_p), ("WT", c_void_p), ("ldw", c_int64), ("N", c_int), ("K", c_int)]


P, I, F, L, U = c_void_p, c_int, c_float, c_int64, c_uint32
# name -> argument types (every function returns int and takes the stream last)
SIGNATURES = {
    "vptr_gemm": [ctypes.POINTER(GemmDesc), P],
    "vptr_gemm_tile_cols": [I],
    "vptr_split_planes": [P, P, L, I, P],
    "vptr_wino_in": [P, P, I, I, I, I, L, I, P],
    "vptr_wino_out": [P, P, P, P, P, I, I, I, I, L, I, I, P],
    "vptr_wino_out_in": [P, P, P, P, P, P, I, I, I, I, L, I, I, I, P],
    "vptr_to_p16": [P, P, L, I, P],
    "vptr_weight_planes": [P, P, I, I, P],
    "vptr_gemm_grouped": [ctypes.POINTER(GemmDesc), P, P, I, I, P],
    "vptr_layernorm_fwd": [P, P, P, P, P, P, I, I, P, P, I, I, F, I, P],
    "vptr_layernorm_bwd": [P, P, P, P, P, P, P, P, P, I, I, P, P],
    "vptr_layernorm_bwd_deferred": [P, P, P, P, P, P, P, I, I, P, P, P],
    "vptr_layernorm_bwd_partials": [I, I],
    "vptr_partial_reduce": [P, I, I, I, P],
    "vptr_rowmod_sum": [P, P, I, I, I, I, P],
    "vptr_colsum": [P, P, I, I, P],
    "vptr_window_copy": [P, P, I, I, I, I, I, I, I, I, P],
    "vptr_add_rowtab": [P, P, P, I, I, I, I, P],
    "vptr_winattn_fwd": [P, P, P, P, P, P, I, I, I, I, I, I, F, P, U, I, P],
    "vptr_winattn_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, P, U, F, I, P],
    "vptr_winattn_bwd_ws": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, P, U, F, I, P, I, P],
    "vptr_winattn_bwd_workspace": [I],
    "vptr_tattn_fwd": [P, P, P, P, I, I, I, I, I, I, I, F, P, U, I, P],
    "vptr_tattn_bwd": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, F, P, U, F, I, P],
    "vptr_tsattn_fwd": [P, P, P, P, I, I, I, I, I, I, I, I, F, P, U, I, P],
    "vptr_tsattn_bwd": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, F, P, U, I, P],
    "vptr_colstats": [P, P, P, P, F, P, I, I, P],
    "vptr_colstats_running": [P, P, P, P, F, P, I, I, P, P, F, P, P],
    "vptr_groupstats": [P, P, P, P, F, I, I, P],
    "vptr_norm_act_fwd": [P, P, P, P, P, P, I, I, I, I, I, F, P, U, P, I, I, P, I, P, F, P],
    "vptr_norm_act_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, P, U, P, I, I, I, P],
    "vptr_norm_act_bwd_deferred": [P, P, P, P, P, P, P, P, I, I, I, I, I, F, P, U, P, I, I, I, P, P],
    "vptr_norm_act_bwd_coop_partials": [I, I, I],
    "vptr_norm_act_bwd_coop": [P, P, P, P, P, P, P, P, I, I, I, I, F, P, U, P, I, I, I, P, P],
    "vptr_norm_act_bwd_partials": [I, I, I, I],
    "vptr_dwconv3x3_fwd": [P, P, P, P, I, I, I, I, P, P],
    "vptr_dwconv3x3_bwd": [P, P, P, P, P, P, I, I, I, I, P],
    "vptr_dwconv3x3_norm_fwd": [P, P, P, P, F, I, P, P, P, P, P, P, I, I, I, I, P, P],
    "vptr_dwconv3x3_bwd_xh": [P, P, P, P, P, P, I, I, I, I, P],
    "vptr_nchw_to_tokens": [P, P, I, I, I, P],
    "vptr_tokens_to_nchw": [P, P, I, I, I, I, P],
    "vptr_nchw_to_tokens_masked": [P, P, P, I, I, I, P],
    "vptr_act_bwd": [P, P, P, I, I, I, F, P, I, I, F, P, U, I, P],
    "vptr_dropout": [P, P, L, F, P, U, P],
    "vptr_rowscale": [P, P, P, I, I, I, I, P],
    "vptr_conv7_in_fwd": [P, P, P, P, P, I, I, I, I, I, P],
    "vptr_conv7_in_fwd_planes": [P, P, P, P, P, I, I, I, I, I, P],
    "vptr_conv7_out_fwd": [P, P, P, P, I, I, I, I, I, I, P],
    "vptr_conv7_out_bwd_data": [P, P, P, P, I, I, I, I, I, I, P],
    "vptr_conv7_out_bwd_weight": [P, P, P, P, P, I, I, I, I, I, I, P],
    "vptr_conv7_out_bwd_weight_ws": [P, P, P, P, P, I, I, I, I, I, I, P, I, P],
    "vptr_conv7_out_bwd_weight_workspace": [I, I],
    "vptr_bnrelu_bwd": [P, P, P, P, L, I, P],
    "vptr_bnrelu_bwd_params": [P, P, P, P, P, P, L, I, P],
    "vptr_bnrelu_bwd_fused": [P, P, P, P, P, P, P, P, L, I, P],
    "vptr_im2col_nhwc": [P, P, I, I, I, I, I, I, I, I, I, I, I, P],
    "vptr_im2col_nhwc_p16": [P, P, I, I, I, I, I, I, I, I, I, I, I, P],
    "vptr_reflect_fold": [P, P, I, I, I, I, I, P],
    "vptr_conv7_in_bwd_weight": [P, P, P, I, I, I, I, I, P],
    "vptr_mse_gdl_fwd": [P, P, P, P, P, I, I, I, P],
    "vptr_mse_gdl_bwd": [P, P, P, P, P, I, I, I, P],
    "vptr_nce_fwd": [P, P, P, P, I, I, I, F, P],
    "vptr_nce_bwd": [P, P, P, P, P, P, I, I, I, F, P],
    "vptr_droppath_scales": [P, P, I, I, P, U, P],
    "vptr_sumsq": [P, L, P, P],
    "vptr_sumsq_ws": [P, L, P, P, I, P],
    "vptr_set_deterministic": [I],
    "vptr_get_deterministic": [],
    "vptr_wgrad_sync_stats": [P, P],
    "vptr_adamw": [P, P, P, P, L, F, F, F, F, F, P, P, F, F, P],
}
EXPORTS = sorted(list(SIGNATURES) + ["vptr_abi_version", "vptr_last_error"])


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "vptr_amd: %s is missing. Build it with `python -m vptr_amd.build` (needs hipcc, gfx950). "
            "There is no CPU / PyTorch fallback for the VPTR hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argt in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.argtypes = argt
        fn.restype = c_int
    lib.vptr_abi_version.restype = c_int
    lib.vptr_last_error.restype = ctypes.c_char_p
    if lib.vptr_abi_version() != 10:
        raise ImportError("vptr_amd: ABI version mismatch in %s" % LIB_PATH)
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, lib.vptr_last_error().decode()))


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream():
    # raw handle of the current stream (torch.cuda.current_stream() builds a Stream object: ~9 us per launch, ~1.1 k launches/step)
    return c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vptr_amd ops run on the MI355X only (got a %s tensor); there is no CPU fallback" % t.device)
