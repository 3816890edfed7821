"""Configuration, dropout seed scopes, the raw GEMM launch and the P16 operand format."""
import ctypes
import os

import torch

from .._lib import GemmDesc, check, lib, ptr, stream


ACT_NONE, ACT_GELU, ACT_RELU, ACT_LRELU = 0, 1, 2, 3
PAD_MODES = {"zero": 0, "reflect": 1, "replicate": 2}


class _Config:
    """Process-wide numeric settings.

    gemm_precision: 3 = split-bf16 MFMA (three passes, fp32-class accuracy; meets the 1e-3 rel-L2 parity bar)
                    1 = single-pass bf16 MFMA (fastest; ~1e-2 end-to-end deviation from the fp32 reference)
    group_wgrads:   True = weight gradients that land in a flat gradient slab are recorded during backward and run as one
                    grouped launch at its end (defer_wgrad / flush_wgrads); False = one split-K launch per layer
    """
    gemm_precision = 3
    group_wgrads = True
    # plain parameters (no FlatAdamW slab) join the grouped launch through their own .grad (ops._loose_grad_for); 0 = A/B switch
    group_loose_wgrads = os.environ.get("VPTR_LOOSE_WGRADS", "1") != "0"
    # LayerNorm(C) gamma / beta gradients with an in-place destination: per-workgroup partial sums now, ONE reduction launch at the end
    # of the backward pass (ops.defer_partial_reduce) instead of 2 C atomics per workgroup and call; 0 = A/B switch
    defer_ln_param_grads = os.environ.get("VPTR_DEFER_LN", "1") != "0"
    # grouped token-major weight gradients: transposed-store orientation for dW whose row count leaves eighth-full tiles; 0 = A/B switch
    wgrad_flip = os.environ.get("VPTR_WGRAD_FLIP", "1") != "0"
    # partly filled last row tiles as separate problems launched after all full tiles (equal-duration tiles stay in step); 0 = A/B switch
    wgrad_split = os.environ.get("VPTR_WGRAD_SPLIT", "0") != "0"   # measured: no change (7.03 vs 7.05 ms bare launch): off
    wgrad_token_split = os.environ.get("VPTR_WGRAD_TOKEN_SPLIT", "1") != "0"   # small weight-gradient groups cut into token ranges (stock-DDP / autograd.grad paths)
    # tile rows of the grouped weight-gradient launches: 128 (rounds 1 - 4), 256 (tall problems on 256 x 176 tiles, one workgroup per CU:
    # 352 vs 300 TFLOP/s on the 2112- / 1584-row problems) or 192 (three stages); profiles/r05_wgrad_rows_ab.log
    # "auto" (default): per problem set, whichever of 128 / 256 measured faster (see _launch_wgrad_group)
    wgrad_rows = (lambda v: v if v == "auto" else int(v))(os.environ.get("VPTR_WGRAD_ROWS", "auto"))
    # stride-2 3x3 transposed convolutions as four parity-class gathers (ops.SubpixelWeights) instead of one 9-tap gather form; 0 = A/B
    subpixel_convt = os.environ.get("VPTR_SUBPIXEL_CONVT", "1") != "0"
    weights_frozen = False  # set by the frozen_weights scope only
    # P16 ("convert once") operands for every nn.Linear-shaped GEMM whose dimensions are multiples of 16 (precision 3 only):
    # the GEMMs stage pre-split bf16 hi / lo granules with global_load_lds instead of splitting fp32 in their main loops
    use_p16 = os.environ.get("VPTR_P16", "1") != "0"
    # weight-gradient chunks on a side stream during backward (see _flush_wgrads_side); 0 = one grouped launch at the end
    wgrad_async = os.environ.get("VPTR_WGRAD_ASYNC", "0") == "1"
    wgrad_chunk_tiles = int(os.environ.get("VPTR_WGRAD_CHUNK", "600"))
    # the transformer MLP as one autograd node (ops.mlp) instead of two ops.linear nodes; 0 = A/B switch
    fused_mlp = os.environ.get("VPTR_FUSED_MLP", "1") != "0"
    # LayerNorm((F,H,W)) statistics accumulated by the epilogue of the producing GEMM / depthwise convolution; 0 = separate pass (A/B)
    fused_frame_stats = os.environ.get("VPTR_FUSED_STATS", "1") != "0"
    winograd = os.environ.get("VPTR_ENC_WINOGRAD", "1") != "0"   # frozen 3x3 stride-1 convolutions of VPTREnc as Winograd F(4x4, 3x3) (round 6)
    norm_coop = os.environ.get("VPTR_NORM_COOP", "0") == "1"   # LayerNorm((F,H,W)) backward as one cooperative pass (round 6: correct, 133 vs 69 us per call -- off)
    winograd_fuse = os.environ.get("VPTR_WINO_FUSE", "1") != "0"   # output transform + next input transform in one pass through LDS
    fused_norm_dwconv_mode = int(os.environ.get("VPTR_FUSED_NORM_DW", "1") or 1)   # 0 off, 1 where it pays (ops.norm_dwconv_ok), 2 wherever valid
    fused_norm_dwconv = os.environ.get("VPTR_FUSED_NORM_DW", "1") != "0"   # conv-FFN norm1 + act1 inside the depthwise kernel's load path (round 6)
    loose_grad_arena = os.environ.get("VPTR_GRAD_ARENA", "1") != "0"   # models without a trainer: `.grad` tensors are views of one buffer per model
    deterministic = False   # ops.set_deterministic / VPTR_DETERMINISTIC=1


config = _Config()


def set_deterministic(on=True):
    """Run-to-run reproducibility of the stage-2 train step (NAR / FAR transformers with <= 16-token attention problems -- every K64
    / BAIR-64 attention -- on one device): the launchers of the library stop letting workgroups meet in fp32 atomics
    (vptr_set_deterministic: one adder per destination for the BatchNorm-type norm-act column sums, the depthwise-convolution weight
    gradients, row-table and column sums), the conv-FFN frame statistics go back to their own fixed-order pass, and `FlatAdamW` takes
    the gradient norm through a fixed-order two-pass sum.  The default path keeps the atomics (they are faster; the reference's
    cuDNN / cuBLAS path is not bit-deterministic either).  Slower: the single-adder geometries serialise ~40 small reductions per step.
    Also: VPTR_DETERMINISTIC=1 in the environment.  tests/test_11_deterministic_gpu.py runs steps twice and compares bit for bit."""
    on = bool(on)
    if on and not config.deterministic:
        config._fused_before = config.fused_frame_stats
        config.fused_frame_stats = False
    elif not on and config.deterministic:
        config.fused_frame_stats = getattr(config, "_fused_before", True)
    config.deterministic = on
    lib.vptr_set_deterministic(int(on))


if os.environ.get("VPTR_DETERMINISTIC") == "1":
    set_deterministic(True)


def _direct_apply(fn_cls):
    """torch.autograd.Function.apply without its Python prologue (functorch dead-wrapper scan, setup_context binding: ~10 us of
    the ~20 us a call costs on the host; ~290 custom nodes per model forward).  These ops are never used under functorch transforms."""
    return super(torch.autograd.Function, fn_cls).apply

_seed_state = {}
_seed_scope = {}


def _dev_key(device):
    return torch.device(device).index or 0


def _master_seed(device):
    """Device-resident master seed of the dropout / DropPath masks.  Initialised from torch.initial_seed() (so torch.manual_seed /
    the reference's set_seed steer it) mixed with the process rank and the device index: data-parallel replicas draw different
    masks, as the reference's per-process RNG streams do."""
    key = _dev_key(device)
    if key not in _seed_state:
        # the rank of the initialised process group (mp.spawn workers carry no RANK variable), else the launcher's RANK
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        else:
            rank = int(os.environ.get("RANK", "0"))
        v = (torch.initial_seed() * 0x9E3779B97F4A7C15 + (rank + 1) * 0xBF58476D1CE4E5B9 + (key + 1) * 0x94D049BB133111EB) & 0x7FFFFFFFFFFFFFFF
        _seed_state[key] = torch.full((1,), v, dtype=torch.int64, device=device)
    return _seed_state[key]


def new_seed_scope(device):
    """Start a new dropout scope (one per model forward): advances the device-resident master seed and snapshots it.
    Every op of the scope -- and its backward, whenever that runs -- reads the snapshot, so forward and backward masks
    agree even if another forward starts in between.  Pure device work: safe under hipGraph capture/replay."""
    m = _master_seed(device)
    m.add_(0x9E3779B9)
    snap = m.clone()
    _seed_scope[_dev_key(device)] = snap
    return snap


def seed_tensor(device):
    """Seed tensor (device, 1 x int64 read as uint64) of the current dropout scope."""
    key = _dev_key(device)
    if key not in _seed_scope:
        return new_seed_scope(device)
    return _seed_scope[key]


def manual_seed(device, value):
    _master_seed(device).fill_(int(value))
    _seed_scope.pop(_dev_key(device), None)


DROPPATH_SITE0 = 0x44500000   # hash sites of the DropPath requests (dropout call sites are small integers: model._assign_sites)


def droppath_scales(keep, maxcount, device, site_offset=0):
    """[len(keep), maxcount] stochastic-depth scales floor(keep + U) / keep; request r hashes (seed of the current dropout scope,
    site DROPPATH_SITE0 + site_offset + r, index) -- one launch, no torch generator (vptr_droppath_scales)"""
    out = torch.empty((keep.shape[0], int(maxcount)), device=device, dtype=torch.float32)
    check(lib.vptr_droppath_scales(ptr(keep), ptr(out), keep.shape[0], int(maxcount), ptr(seed_tensor(device)),
                                   DROPPATH_SITE0 + int(site_offset), stream()), "vptr_droppath_scales")
    return out


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# (Measured and rejected this round: running the q/k/v projections on forked HIP streams -- no gain on the MI355X,
# 110.9 vs 108.7 ms/step, and the cross-stream gradient accumulation into the flat slab needs extra fencing.)


# ------------------------------------------------------------------------------------------------------------------
# raw GEMM
# ------------------------------------------------------------------------------------------------------------------
def gemm_raw(A, B, D, M, N, K, a_mode=0, b_mode=0, lda=None, ldb=None, bias=None, colscale=None, alpha=1.0, act=ACT_NONE,
             Dpre=None, rowscale=None, rs_div=1, rs_mod=1, dropout_p=0.0, site=0, residual=None, act_after=False,
             atomic=False, split_k=1, conv=None, precision=None, seed=None, a_rowsum=None, batch_extra=None, kseg_extra=None,
             planes_out=None, d_p16=False, act_grad_src=None, frame_stats=None, frame_rows=0, row_map=None, ldd=None, batch_accum=0,
             batch_strided=None):
    """One vptr_gemm launch.  batch_extra = [(A, B, D, bias, alpha), ...] adds up to two same-shaped independent problems to
    the grid; kseg_extra = [(A, B), ...] adds up to two K-segments accumulated into the same D (include/vptr_hip.h)."""
    d = GemmDesc()
    if batch_extra:
        d.batch = 1 + len(batch_extra)
        for i, (A2, B2, D2, bias2, alpha2) in enumerate(batch_extra, 1):
            setattr(d, "A_x%d" % i, A2.data_ptr()), setattr(d, "B_x%d" % i, B2.data_ptr()), setattr(d, "D_x%d" % i, D2.data_ptr())
            setattr(d, "bias_x%d" % i, bias2.data_ptr() if bias2 is not None else None)
            setattr(d, "alpha_x%d" % i, alpha2)
    if batch_strided:   # (members, stride_a, stride_b, stride_d) in fp32 elements: ABI 10 strided members of a P16 launch
        d.batch, d.batch_stride_a, d.batch_stride_b, d.batch_stride_d = (int(v) for v in batch_strided)
    if kseg_extra:
        d.ksegs = 1 + len(kseg_extra)
        for i, (A2, B2) in enumerate(kseg_extra, 1):
            setattr(d, "A_x%d" % i, A2.data_ptr()), setattr(d, "B_x%d" % i, B2.data_ptr())
    d.a_rowsum = ptr(a_rowsum)
    d.D_planes = ptr(planes_out)
    d.d_p16 = int(bool(d_p16))
    d.act_grad_src = ptr(act_grad_src)
    d.frame_stats, d.frame_rows = ptr(frame_stats), int(frame_rows)
    d.A, d.B, d.D, d.Dpre = ptr(A), ptr(B), ptr(D), ptr(Dpre)
    d.lda = lda if lda is not None else (A.stride(0) if a_mode != 2 else 0)
    d.ldb = ldb if ldb is not None else B.stride(0)
    d.ldd = ldd if ldd is not None else (D.stride(0) if D is not None else N)
    if row_map is not None:
        d.d_row_w, d.d_row_off = int(row_map[0]), int(row_map[1])
    d.batch_accum = int(batch_accum)
    d.M, d.N, d.K = M, N, K
    d.a_mode, d.b_mode = a_mode, b_mode
    d.precision = precision if precision is not None else config.gemm_precision
    d.split_k, d.atomic = split_k, int(atomic)
    d.colscale, d.bias = ptr(colscale), ptr(bias)
    d.alpha, d.act = alpha, act
    d.rowscale, d.rs_div, d.rs_mod = ptr(rowscale), rs_div, rs_mod
    d.dropout_p = dropout_p
    if dropout_p > 0 and seed is None:
        raise RuntimeError("gemm_raw: dropout needs the scope seed tensor")
    d.seed_dev = ptr(seed) if dropout_p > 0 else None
    d.site = site
    d.residual = ptr(residual)
    d.ldr = residual.stride(0) if residual is not None else 0
    d.act_after = int(act_after)
    if conv is not None:
        (d.conv_IH, d.conv_IW, d.conv_Cin, d.conv_OH, d.conv_OW, d.conv_KH, d.conv_KW, d.conv_stride, d.conv_pad,
         d.conv_pad_mode, d.conv_transposed) = conv
    prof = profiling.gemm
    if prof is not None:  # bench.py roofline pass: HIP events on the launch stream around every GEMM launch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.vptr_gemm(ctypes.byref(d), stream()), "vptr_gemm")
    if prof is not None:
        e1.record()
        key = (gemm_nfn(N), d.precision, a_mode, b_mode)
        if a_mode == A_P16:   # name the instantiation csrc/gemm_p16.hip picks (epilogue flavour, stages), as rocprof lists it
            lean = (colscale is None and Dpre is None and rowscale is None and act == ACT_NONE and dropout_p == 0 and not act_after and not atomic)
            lean3 = (not lean and colscale is None and Dpre is None and act == ACT_NONE and not act_after and not atomic and frame_stats is None)
            tiles = ((M + 127) // 128) * ((N + 175) // 176) * max(d.batch, 1)
            cus = torch.cuda.get_device_properties(A.device).multi_processor_count
            lean4 = (not lean and not lean3 and act_grad_src is None and colscale is None and rowscale is None and residual is None and act != ACT_NONE
                     and not act_after and not atomic and frame_stats is None and os.environ.get("VPTR_GEMM_NO_EPI4") is None)
            key = key + ("p16", 2 if act_grad_src is not None else (4 if lean4 else (3 if lean3 else int(lean))),
                         (3 if os.environ.get("VPTR_GEMM_LONE_STAGES") == "3" else 4) if tiles <= cus else 2)
        elif a_mode != 3:     # register-staged kernels: pipelined loop below 384 workgroups (csrc/gemm.hip launch_one), else single-image
            cols = 16 * gemm_nfn(N)
            wgs = ((M + 127) // 128) * ((N + cols - 1) // cols) * max(split_k, 1) * max(d.batch, 1)
            key = key + ("staged", "p" if (wgs < int(os.environ.get("VPTR_GEMM_V4_MIN_TILES", "384")) or a_rowsum is not None) else "s")
        prof.append((key, 2.0 * M * N * K * max(d.batch, d.ksegs, 1), e0, e1))
    return D


class _Profiling:
    """bench.py's instrumented passes: `gemm` = a list that every GEMM launch appends (key, flops, event, event) to, `opt` = a list
    FlatAdamW.step appends (slab elements, plane elements, 3 HIP events) to; None = off"""
    gemm = None
    opt = None


profiling = _Profiling()


def gemm_nfn(N):
    """Column-fragment count of the kernel instantiation vptr_gemm picks for an N-wide output (mirrors csrc/gemm.hip)."""
    if N % 176 == 0:
        return 11
    if N <= 64:
        return 4
    if N <= 128:
        return 8
    cands = [((N + 175) // 176 * 176, 11), ((N + 127) // 128 * 128, 8), ((N + 63) // 64 * 64, 4)]
    best = cands[0]
    for c in cands[1:]:
        if c[0] < best[0]:
            best = c
    return best[1]


# ---- P16 operands ("convert once") --------------------------------------------------------------------------------------
# A P16 tensor is an ordinary float32 torch tensor of the logical shape [rows, C] whose BYTES are 16-channel granules of
# 16 bf16 hi | 16 bf16 lo (include/vptr_hip.h).  Same shape, dtype and size as the fp32 tensor it replaces, so it travels through
# autograd unchanged; which tensors are P16 is static knowledge of the call sites (`*_p16` flags), never inferred.
A_P16, B_P16, A_P16T, B_P16T = 5, 3, 6, 4


def p16_ok(*dims):
    """True when GEMM dimensions qualify for the P16 kernels (multiples of 16, split-bf16 precision, feature enabled)"""
    return config.use_p16 and config.gemm_precision == 3 and all(d % 16 == 0 for d in dims)


def to_p16(x):
    """fp32 [rows, C] -> P16 (one HBM pass; producers that can write P16 themselves make this unnecessary)"""
    x = _c(x)
    out = torch.empty_like(x)
    rows = x.numel() // x.shape[-1]
    check(lib.vptr_to_p16(ptr(x), ptr(out), rows, x.shape[-1], stream()), "vptr_to_p16")
    return out


class _AsP16Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return to_p16(x)

    @staticmethod
    def backward(ctx, dy):
        return dy


_AsP16Fn_apply = _direct_apply(_AsP16Fn)


def as_p16(x):
    """autograd-aware fp32 -> P16 conversion: the gradient of the P16 tensor (an ordinary fp32 tensor) passes through"""
    return _AsP16Fn_apply(x)


def p16_decode(t):
    """P16 -> fp32 values with plain torch ops (tests / debugging only)"""
    C = t.shape[-1]
    b = t.contiguous().view(torch.bfloat16).reshape(-1, C // 16, 2, 16).float()
    return (b[:, :, 0] + b[:, :, 1]).reshape(t.shape)
