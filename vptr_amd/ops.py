"""Host-side operators of the VPTR hot path: thin autograd wrappers over the C-ABI HIP kernels.

Every forward/backward here is a call into libvptr_hip.so (vptr_amd/_lib.py); torch is used for device memory,
streams and autograd bookkeeping only.  Activations are token-major, channel-last 2-D tensors [rows, C] with
rows = (n, t, h, w) flattened -- the reference's window partition and (T, N*HW, C) permutes never materialise.
"""
import bisect
import ctypes
import os

import torch

from . import _lib
from ._lib import GemmDesc, WPlaneEntry, check, lib, ptr, stream

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
             planes_out=None, d_p16=False, act_grad_src=None, frame_stats=None, frame_rows=0, row_map=None, ldd=None, batch_accum=0):
    """One vptr_gemm launch.  batch_extra = [(A, B, D, bias, alpha), ...] adds up to two same-shaped independent problems to
    the grid; kseg_extra = [(A, B), ...] adds up to two K-segments accumulated into the same D (include/vptr_hip.h)."""
    d = GemmDesc()
    if batch_extra:
        d.batch = 1 + len(batch_extra)
        for i, (A2, B2, D2, bias2, alpha2) in enumerate(batch_extra, 1):
            setattr(d, "A_x%d" % i, A2.data_ptr()), setattr(d, "B_x%d" % i, B2.data_ptr()), setattr(d, "D_x%d" % i, D2.data_ptr())
            setattr(d, "bias_x%d" % i, bias2.data_ptr() if bias2 is not None else None)
            setattr(d, "alpha_x%d" % i, alpha2)
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
    prof = _gemm_prof
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


_gemm_prof = None
_opt_prof = None    # list -> FlatAdamW.step appends (slab elements, plane elements, 3 HIP events) per call (bench.py hbm roofline)


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


class WeightPlanes:
    """P16 images of a set of nn.Linear-shaped weights ([N, K] views with N, K multiples of 16), rebuilt by ONE launch
    (vptr_weight_planes) after every optimizer step: Wp [N, K] for the forward GEMMs and WT [K, N] for the input-gradient GEMMs."""

    def __init__(self, weights):
        self.weights = [w for w in weights]
        dev = self.weights[0].device
        total = sum(w.shape[0] * w.shape[1] for w in self.weights)
        self.wp = torch.empty(total, device=dev, dtype=torch.float32)
        self.wt = torch.empty(total, device=dev, dtype=torch.float32)
        ents = (WPlaneEntry * len(self.weights))()
        starts, off, tiles = [], 0, 0
        self.index = []   # (ptr, nbytes, offset, N, K, weight tensor)
        for i, w in enumerate(self.weights):
            N, K = w.shape
            if N % 16 or K % 16 or w.stride(1) != 1:
                raise RuntimeError("WeightPlanes: weight %d of shape %s is not P16-eligible" % (i, tuple(w.shape)))
            e = ents[i]
            e.W, e.ldw, e.N, e.K = w.data_ptr(), w.stride(0), N, K
            e.Wp = self.wp.data_ptr() + off * 4
            e.WT = self.wt.data_ptr() + off * 4
            self.index.append((w.data_ptr(), N * w.stride(0) * 4, off, N, K, w))
            starts.append(tiles)
            tiles += ((N + 31) // 32) * ((K + 31) // 32)
            off += N * K
        import struct
        self.table = torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8).to(dev)
        self.starts = torch.frombuffer(bytearray(struct.pack("%di" % len(starts), *starts)), dtype=torch.uint8).to(dev)
        self.tiles, self.versions, self._views = tiles, None, {}
        self.index.sort(key=lambda t: t[0])
        self.bases = [t[0] for t in self.index]
        self.refresh()

    def refresh(self):
        check(lib.vptr_weight_planes(ptr(self.table), ptr(self.starts), len(self.weights), self.tiles, stream()), "vptr_weight_planes")
        self.versions = [t[5]._version for t in self.index]
        self.dirty = False

    grad_arena = None
    dirty = False   # set by invalidate_weight_planes(): a write torch's version counters cannot see (.data, slab writes, c10d collectives)

    def stale(self):
        """True when a registered weight changed through torch (load_state_dict, an external optimizer) since the last refresh, or
        when somebody declared the planes invalid (invalidate_weight_planes)"""
        return self.dirty or any(v != t[5]._version for v, t in zip(self.versions, self.index))

    def lookup(self, W):
        """(Wp, ld, WT, ld) for W = a registered weight or a whole-row slice of one, or None; stale planes (the weight changed
        through torch since the last refresh) are rebuilt first"""
        p = W.data_ptr()
        i = bisect.bisect_right(self.bases, p) - 1
        if i < 0:
            return None
        base, nbytes, off, N, K, w = self.index[i]
        if not (base <= p < base + nbytes):
            return None
        if w.stride(0) != K or W.shape[1] != K or W.stride(0) != K or (p - base) % (K * 4):
            return None
        if self.dirty or self.versions[i] != w._version:
            self.refresh()
        r0, n = (p - base) // (K * 4), W.shape[0]
        if r0 % 16 or n % 16 or r0 + n > N:
            return None
        key = (i, r0, n)
        hit = self._views.get(key)
        if hit is None:   # view construction costs ~10 us of host time per call site and step otherwise
            wp = self.wp[off + r0 * K: off + (r0 + n) * K].view(n, K)
            wt = self.wt[off: off + N * K].view(K, N)[:, r0:r0 + n]
            hit = self._views[key] = (wp, K, wt, N)
        return hit


_wplane_stores = []      # weakrefs of WeightPlanes registered by the trainers (FlatAdamW slabs)
_wplane_cache = {}       # (ptr, version, N, K, ld) -> (WeightPlanes, bytes): weights outside any store (eval / tests), LRU by bytes
_WPLANE_CACHE_BYTES = 3 << 30


def register_weight_planes(store):
    import weakref
    _wplane_stores.append(weakref.ref(store))


def ensure_module_planes(module):
    """One P16 weight store per MODEL for modules used without a trainer (the reference's scripts: plain nn.Parameters stepped by
    torch.optim.AdamW): every Linear-shaped weight of `module` gets its planes from ONE vptr_weight_planes launch per optimizer step
    (WeightPlanes.lookup rebuilds the whole store when a version counter moved) instead of one launch + two table uploads per weight
    and forward (196 per K64 forward: ~15 ms of host time, tools/dropin_prof.py).  Called at the top of VPTRFormerNAR / FAR.forward;
    a no-op when a trainer's store (FlatAdamW) already covers the module's weights or when the store is current."""
    if not config.use_p16:
        return
    st = module.__dict__.get("_vptr_planes")
    if st is not None and st.grad_arena is not None and torch.is_grad_enabled():
        _arm_grad_arena(st.grad_arena)
    first = next((p for p in module.parameters() if p.dim() == 2 and p.shape[0] % 16 == 0 and p.shape[1] % 16 == 0 and p.is_contiguous()), None)
    if first is None or not first.is_cuda:
        return
    if st is not None:
        if st.sentinel == (first.data_ptr(), first.shape):
            return
        module.__dict__["_vptr_planes"] = None      # the parameters moved (.to(), a flat slab took them over): rebuild or defer
        if st.grad_arena is not None:
            for k in [k for k, e in _grad_arenas.items() if e[1]() is st.grad_arena]:
                del _grad_arenas[k]
        _wplane_stores[:] = [r for r in _wplane_stores if r() is not None and r() is not st]
    for ref in _wplane_stores:
        other = ref()
        if other is not None and other.lookup(first.detach()) is not None:
            return                                   # a trainer's store serves these weights
    lin = linear_weights_of(module.parameters())
    if not lin:
        return
    with torch.no_grad():
        st = WeightPlanes(lin)
    st.sentinel = (first.data_ptr(), first.shape)
    module.__dict__["_vptr_planes"] = st             # not a registered buffer / submodule: never in state_dict
    register_weight_planes(st)
    st.grad_arena = None
    if flat_grad_for(first.detach()) is None and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        st.grad_arena = _register_grad_arena(module)   # kept alive by the store; entries of dead parameters are replaced on re-registration
        if st.grad_arena is not None and torch.is_grad_enabled():
            _arm_grad_arena(st.grad_arena)


def invalidate_weight_planes():
    """Declare every cached P16 weight image stale.  The images are keyed on torch's tensor version counters, which miss writes
    through `param.data`, direct writes to an optimizer slab and the in-place c10d collectives (dist.broadcast bumps no version):
    vptr_amd.parallel.broadcast_module / _broadcast_any and FlatAdamW.load_state_dict call this; so must any other code that
    rewrites weights behind autograd's back.  The next GEMM that needs a weight's planes rebuilds them (one launch per store)."""
    for ref in list(_wplane_stores):
        st = ref()
        if st is None:
            _wplane_stores.remove(ref)
        else:
            st.dirty = True
    _wplane_cache.clear()


def weight_planes_for(W):
    """P16 planes (Wp [N,K], ld, WT [K,N] view, ld) of a Linear-shaped weight: from a trainer's store, else from a small cache"""
    for ref in list(_wplane_stores):
        st = ref()
        if st is None:
            _wplane_stores.remove(ref)
            continue
        hit = st.lookup(W)
        if hit is not None:
            return hit
    key = (W.data_ptr(), W._version, W.shape[0], W.shape[1], W.stride(0))
    hit = _wplane_cache.get(key)
    if hit is None:
        with torch.no_grad():
            st = WeightPlanes([W.detach()])
        nbytes = 8 * W.shape[0] * W.shape[1]
        tot = nbytes + sum(v[1] for v in _wplane_cache.values())
        for k in list(_wplane_cache):      # insertion order = least recently built first
            if tot <= _WPLANE_CACHE_BYTES:
                break
            tot -= _wplane_cache.pop(k)[1]
        hit = _wplane_cache[key] = (st, nbytes)
    N, K = W.shape
    return hit[0].wp.view(N, K), K, hit[0].wt.view(K, N), N


def linear_weights_of(params):
    """the nn.Linear-shaped members of a parameter list ([N, K] or 1x1-conv [N, K, 1, 1]; N, K multiples of 16) as [N, K] views"""
    out = []
    for p in params:
        if p.dim() == 4 and p.shape[2] == 1 and p.shape[3] == 1 and p.is_contiguous():
            w = p.detach().view(p.shape[0], p.shape[1])
        elif p.dim() == 2 and p.is_contiguous():
            w = p.detach()
        else:
            continue
        if w.shape[0] % 16 == 0 and w.shape[1] % 16 == 0:
            out.append(w)
    return out


# ---- deferred, grouped weight gradients ---------------------------------------------------------------------------------
# dW = dY^T . X of one nn.Linear is 12-60 output tiles with K = all tokens: alone it cannot fill 256 CUs without ~30
# K-splits (each paying a prologue and a 90 KB atomic epilogue; measured 80 TFLOP/s).  When the weight's gradient lives
# in a registered flat slab (so nothing has to be handed back to autograd), the call is only recorded here and the whole
# backward pass's weight gradients run as ONE vptr_gemm_grouped launch, queued on the autograd engine's end-of-backward
# callback: every tile then runs the full K loop and writes once.
_wgrad_q = []


def defer_wgrad(g, x, dW, N, K, M, db=None, alpha=1.0, p16=False):
    """record dW[N,K] += g[M,N]^T . x[M,K] (dW, and db if given, must be views of a flat gradient slab); with db the bias
    gradient db[N] += column sums of g rides on the same launch (vptr_gemm_desc::a_rowsum).  p16: g and x are P16 tensors."""
    _wgrad_q.append((g, x, dW, N, K, M, config.gemm_precision, db, float(alpha), bool(p16)))
    if config.wgrad_async and not _wgrad_hold[0]:
        _wgrad_side["tiles"] += ((N + 127) // 128) * ((K + 175) // 176)
        if _wgrad_side["tiles"] >= config.wgrad_chunk_tiles:
            _flush_wgrads_side()
    # one end-of-backward callback per recorded call: flush_wgrads is idempotent, and registering every time stays correct
    # when an earlier backward died before its callbacks ran (a "callback already queued" flag would then be stale)
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_auto_flush_wgrads)
    except RuntimeError:  # not inside a backward pass: the caller flushes explicitly
        pass


def take_wgrads():
    """hand the recorded (not yet launched) weight gradients to the caller and clear the queue (NARTrainer.capture_front keeps the
    records of the captured backward pass: their operands have static addresses in the graph's pool)"""
    items = list(_wgrad_q)
    del _wgrad_q[:]
    return items


def requeue_wgrads(items):
    """put records obtained from take_wgrads() back (after a replay of the graph that produces their operands)"""
    _wgrad_q.extend(items)


def discard_wgrads():
    """drop recorded weight gradients that were never launched (a backward pass that raised); called by FlatAdamW.zero_grad"""
    del _wgrad_q[:]
    del _reduce_q[:]


# ---- deferred partial-sum reductions (parameter gradients of the LayerNorms) -----------------------------------------------
_reduce_q = []


def defer_partial_reduce(part, dst0, dst1, nparts, C):
    """record dst0[C] += sum_p part[p][0][:], dst1[C] += sum_p part[p][1][:]; every record of a backward pass is served by one
    vptr_partial_reduce launch at its end (before the grouped weight gradients: a data-parallel step sends gradient ranges out as
    soon as their weight-gradient chunk is done)."""
    _reduce_q.append((part, dst0, dst1, int(nparts), int(C)))
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_auto_flush_wgrads)
    except RuntimeError:  # not inside a backward pass: the caller flushes explicitly (flush_wgrads)
        pass


def flush_partial_reduces():
    if not _reduce_q:
        return
    items = list(_reduce_q)
    del _reduce_q[:]
    # one table, one upload; entries sorted by width and launched per width class (the grid is sized for the widest entry of a launch:
    # the 528-wide LayerNorm rows must not ride on the grid of the 135 168-wide LayerNorm((F,H,W)) rows)
    items.sort(key=lambda it: it[4])
    tab = (_lib.ReduceEntry * len(items))()
    for i, (part, d0, d1, nparts, C) in enumerate(items):
        tab[i].part, tab[i].dst0, tab[i].dst1, tab[i].nparts, tab[i].C = ptr(part), ptr(d0), ptr(d1), nparts, C
    dev = items[0][0].device
    raw = _to_device_async(bytes(tab), dev)
    esz = ctypes.sizeof(_lib.ReduceEntry)
    lo = 0
    while lo < len(items):
        hi = lo
        while hi < len(items) and items[hi][4] <= 4 * items[lo][4]:
            hi += 1
        dsts = [it[k].data_ptr() for it in items[lo:hi] for k in (1, 2)]
        unique = len(set(dsts)) == len(dsts)   # a module applied twice in one forward: atomics
        check(lib.vptr_partial_reduce(ctypes.c_void_p(raw.data_ptr() + lo * esz), hi - lo, items[hi - 1][4], int(unique), stream()),
              "vptr_partial_reduce")
        lo = hi


_wgrad_hold = [False]  # set by hold_wgrads(): the end-of-backward callback leaves the queue to an explicit chunked flush


class hold_wgrads:
    """Scope in which the end-of-backward callback does NOT launch the recorded weight gradients: the data-parallel trainer
    flushes them itself in a few chunks (flush_wgrads(chunks=..., on_chunk=...)) so that the all-reduce of one chunk's
    gradient range overlaps the GEMM launch of the next."""

    def __enter__(self):
        self.prev = _wgrad_hold[0]
        _wgrad_hold[0] = True
        return self

    def __exit__(self, *exc):
        _wgrad_hold[0] = self.prev
        return False


# ---- weight gradients on a side stream, overlapped with the rest of the backward pass -------------------------------------------
# The grouped weight-gradient launch is MFMA-bound, about half of the backward pass's other kernels are HBM-bound (normalisation,
# attention cores, LayerNorm) or leave CUs idle (240-tile GEMMs): instead of one launch at the very end, the recorded problems are
# flushed in chunks of >= config.wgrad_chunk_tiles tiles onto a second HIP stream while backward keeps running on the main one.
# g and x stay alive through record_stream (the caching allocator defers their reuse until the side stream has passed them).
_wgrad_side = {"stream": None, "tiles": 0, "dirty": False}


def _flush_wgrads_side():
    items = list(_wgrad_q)
    del _wgrad_q[:]
    _wgrad_side["tiles"] = 0
    if not items:
        return
    cur = torch.cuda.current_stream()
    if _wgrad_side["stream"] is None:
        _wgrad_side["stream"] = torch.cuda.Stream()
    side = _wgrad_side["stream"]
    side.wait_stream(cur)          # every operand recorded so far has been produced on the main stream
    with torch.cuda.stream(side):
        # never the panel-synchronous persistent kernel here: it assumes all its workgroups resident and owns the device-wide barrier
        # words (include/vptr_hip.h), and a side-stream launch runs beside the main stream's backward kernels
        _launch_wgrad_group(items, allow_sync=False)
    for it in items:
        it[0].record_stream(side)
        it[1].record_stream(side)
    _wgrad_side["dirty"] = True


def join_wgrad_stream():
    """make the current stream wait for weight-gradient chunks still running on the side stream (before the optimizer reads them)"""
    if _wgrad_side["dirty"]:
        torch.cuda.current_stream().wait_stream(_wgrad_side["stream"])
        _wgrad_side["dirty"] = False


def _auto_flush_wgrads():
    if not _wgrad_hold[0]:
        flush_partial_reduces()
        if config.wgrad_async and _wgrad_q:
            _flush_wgrads_side()
        else:
            flush_wgrads()
        join_wgrad_stream()


_pin_pool = {"slots": [], "next": 0}
_pin_pool_small = {"slots": [], "next": 0}
_wgrad_tune = {}    # problem-set signature -> {"ms": {tile rows: best ms}, "pending": (rows, e0, e1) | None, "choice": rows | None}
_graph_keepalive = []   # pinned upload sources of captured launches (must outlive every replay)
_graph_reserve = []     # pinned buffers set aside for the next capture


_upload_stats = {"count": 0, "max_bytes": 0}   # table uploads since the last reset (NARTrainer.capture sizes its reserve from a warm-up step)


def reserve_graph_staging(count=8, nbytes=1 << 18):
    """set `count` pinned staging buffers of `nbytes` aside for the host-built tables of a whole-step graph capture (pinned memory
    cannot be allocated while capturing); buffers that are too small are replaced"""
    _graph_reserve[:] = [b for b in _graph_reserve if b.numel() >= nbytes]
    while len(_graph_reserve) < count:
        _graph_reserve.append(torch.empty(nbytes, dtype=torch.uint8).pin_memory())


def _to_device_async(host_bytes, dev):
    """bytes -> uint8 device tensor through a rotating pool of pinned staging buffers with a non-blocking copy: a pageable
    `.to(device)` would block the host until every kernel enqueued so far has finished (once per step, right where the host
    should be running ahead into the optimizer and the next forward pass)."""
    n = len(host_bytes)
    _upload_stats["count"] += 1
    _upload_stats["max_bytes"] = max(_upload_stats["max_bytes"], n)
    if os.environ.get("VPTR_SYNC_UPLOAD") == "1":
        return torch.frombuffer(bytearray(host_bytes), dtype=torch.uint8).to(dev)
    if torch.cuda.is_current_stream_capturing():
        # the copy becomes a memcpy node that reads the HOST buffer at every replay: it gets a pinned buffer of its own that is
        # never reused (the rotating pool below is rewritten by later eager launches -- replays would upload stale descriptors)
        # (pinned memory cannot be allocated while capturing: reserve_graph_staging() set buffers aside beforehand)
        for i, cand in enumerate(_graph_reserve):
            if cand.numel() >= n:
                buf = _graph_reserve.pop(i)
                break
        else:
            raise RuntimeError("graph capture: no reserved pinned staging buffer of %d bytes (ops.reserve_graph_staging)" % n)
        buf[:n].copy_(torch.frombuffer(bytearray(host_bytes), dtype=torch.uint8))
        _graph_keepalive.append(buf)
        return buf[:n].to(dev, non_blocking=True)
    # two rotating pools: 1024 small buffers (descriptor tables of one-layer launches: a torch.distributed job makes ~200 of those per
    # backward pass, and with 16 buffers the host had to wait for the device every 8 launches -- it could never run ahead) and 16 large ones
    small = n <= 8192
    pool = _pin_pool_small if small else _pin_pool
    i = pool["next"] % (1024 if small else 16)
    pool["next"] += 1
    while len(pool["slots"]) <= i:
        pool["slots"].append([torch.empty(8192 if small else (1 << 16), dtype=torch.uint8).pin_memory(), None])
    slot = pool["slots"][i]
    if slot[1] is not None:
        slot[1].synchronize()  # the copy that last used this staging buffer (16 transfers ago) must have completed
    if slot[0].numel() < n:
        slot[0] = torch.empty(2 * n, dtype=torch.uint8).pin_memory()
    slot[0][:n].copy_(torch.frombuffer(bytearray(host_bytes), dtype=torch.uint8))
    out = slot[0][:n].to(dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    slot[1] = ev
    return out


def _wgrad_tune_book(tune):
    r_, e0_, e1_ = tune["pending"]
    tune["ms"][r_] = min(tune["ms"].get(r_, 1e30), e0_.elapsed_time(e1_))
    tune["pending"] = None
    if len(tune["ms"]) == 2:
        tune["choice"] = min(tune["ms"], key=tune["ms"].get)


def wgrad_tune_settle():
    """book every timed weight-gradient flush that is still in flight (device synchronisation); the trainers call it between their
    eager warm-up steps and a graph capture, where events can no longer be queried"""
    if any(t["pending"] is not None for t in _wgrad_tune.values()):
        torch.cuda.synchronize()
        for t in _wgrad_tune.values():
            if t["pending"] is not None:
                _wgrad_tune_book(t)


def plan_wgrad_launches(probs, cols, p16, atomic, allow_sync, rows_mode, split_rem=False, token_split=True):
    """Pure planning step of the grouped weight-gradient flush (no tensors, no launches: tests/test_cpu.py drives it with made-up
    addresses).  probs: (A ptr, B ptr, D ptr, rowsum ptr, lda, ldb, ldd, rows, cols, tokens, alpha, transposed) per weight, pointers as
    integers, leading dimensions in floats.  Returns [(sub-problems, vouch)]: one entry per kernel launch, every sub-problem the same
    tuple + its tile rows as a 13th element where they are not 128; `vouch` = every sub-problem of the launch walks the same number of
    tokens (the panel-synchronous persistent kernel may serve it).  Rows of a problem are cut between a 256- (or 192-) row launch and
    the 128-row launch; a small group is cut into token ranges that accumulate into one destination; problems of different token
    counts go to different persistent launches."""
    subs = []
    for (ap, bp, dp, rp, lda, ldb, ldd, rows_, cols_, M, alpha, flip) in probs:
        small_group = p16 and atomic and token_split and len(probs) <= 3   # one layer's launch: token ranges on 128-row tiles (below)
        if p16 and atomic and rows_mode == 192 and not small_group and rows_ >= 384:
            # 192 x 176 tiles (three stages, one workgroup per CU): 2112 = 11 x 192 exactly, 528 = 2.75 (three tiles, the last 3/4 full,
            # against 4.125 128-row tiles); a remainder that pads a 128-row tile less than a 192-row one joins the 128-row launch
            rem = rows_ % 192
            to128 = 0      # trailing rows handed to the 128-row launch
            if flip and rp:   # the column sums of a flipped problem need a free 16-row fragment in the tile that holds its last rows
                if rem == 0:
                    to128 = 192
                elif 192 - rem < 16:
                    to128 = rem
            elif rem and (192 - rem) > ((rem + 127) // 128) * 128 - rem:
                to128 = rem
            if to128:
                full = rows_ - to128
                subs.append((ap, bp, dp, 0 if flip else rp, lda, ldb, ldd, full, cols_, M, alpha, flip, 192))
                subs.append((ap + full * 4, bp, dp + (full * 4 if flip else full * ldd * 4), (rp + (0 if flip else full * 4)) if rp else 0,
                             lda, ldb, ldd, to128, cols_, M, alpha, flip, 128))
            else:
                subs.append((ap, bp, dp, rp, lda, ldb, ldd, rows_, cols_, M, alpha, flip, 192))
            continue
        if p16 and atomic and rows_mode == 256 and not small_group and rows_ >= 1024 and (rows_ // 256) * 256 >= 0.85 * rows_:
            # tall problems on 256 x 176 tiles (1.47x the flops per staged byte, one workgroup per CU): the multiple-of-256 part
            # goes to the 256-row launch, the rest of the rows stays a 128-row problem (and keeps the bias gradient of a flipped one)
            full = (rows_ // 256) * 256
            rem256 = rows_ - full
            subs.append((ap, bp, dp, 0 if (flip and rem256) else rp, lda, ldb, ldd, full, cols_, M, alpha, flip, 256))
            if rem256:
                subs.append((ap + full * 4, bp, dp + (full * 4 if flip else full * ldd * 4), (rp + (0 if flip else full * 4)) if rp else 0,
                             lda, ldb, ldd, rem256, cols_, M, alpha, flip, 128))
            continue
        rem = rows_ % 128
        if p16 and split_rem and rem and rows_ > 128:
            # the partly filled last row tile of every problem becomes a problem of its own, launched after all full tiles: full
            # tiles then all take the same time, so the tiles that share an operand panel stay in step (and in one L2), instead
            # of being scattered by the short tiles that used to finish early between them
            full = rows_ - rem
            subs.append((ap, bp, dp, 0 if flip else rp, lda, ldb, ldd, full, cols_, M, alpha, flip))
            subs.append((ap + full * 4, bp, dp + (full * 4 if flip else full * ldd * 4), (rp + (0 if flip else full * 4)) if rp else 0,
                         lda, ldb, ldd, rem, cols_, M, alpha, flip))
        else:
            subs.append((ap, bp, dp, rp, lda, ldb, ldd, rows_, cols_, M, alpha, flip))

    if p16 and split_rem:
        subs.sort(key=lambda t: (0 if t[7] >= 128 else 1, -t[7] * t[8], t[1], t[2]))   # full-tile problems first (largest first), remainders last
    def trows(sub):
        return sub[12] if len(sub) > 12 else 128

    def tiles_of(sub):
        return ((sub[7] + trows(sub) - 1) // trows(sub)) * ((sub[8] + cols - 1) // cols)
    # the panel-synchronous persistent launch needs ONE token count per launch (its barrier counts K-blocks): problems are classed by
    # token count, every class of >= 1024 tiles gets a launch of its own, the rest share a plain launch.  K64 / BAIR: one class.  KTH128
    # 10 -> 40: encoder layers (10 frames of tokens) and decoder layers (40 frames) = two persistent launches instead of one plain launch
    # that re-fetched every operand panel 5x over the fabric (80 GB per launch, profiles/r05_cfg5_kernel_stats.md).
    if p16 and atomic and token_split:   # (several adders per destination: not bit-reproducible)
        # a SMALL group (one layer's weight: the launches a torch.distributed job / torch.autograd.grad make, where every gradient must be
        # complete when its autograd node returns) is 15 - 60 tiles with a K loop over every token: 6 - 25 % of the CUs for the whole
        # launch.  Its problems are cut into token ranges that accumulate into the same (zero-initialised) destination, enough of them
        # to put ~2 workgroups on every CU -- what ops.convt_weight_grads does for the decoder (stock-DDP step: see bench.py
        # other_configs.drop_in_ddp_single_iter)
        tot = sum(tiles_of(x) for x in subs)
        if 0 < tot < 384:
            want = (512 + tot - 1) // tot      # ~2 workgroups per CU; every range >= 1024 tokens (each range pays a full atomic epilogue)
            cut = []
            for sub in subs:
                Mtok = sub[9]
                S = max(1, min(want, Mtok // 1024))
                if S == 1:
                    cut.append(sub)
                    continue
                chunk = (((Mtok + S - 1) // S + 31) // 32) * 32
                t0 = 0
                while t0 < Mtok:
                    n_t = min(chunk, Mtok - t0)
                    cut.append((sub[0] + t0 * sub[4] * 4, sub[1] + t0 * sub[5] * 4) + tuple(sub[2:9]) + (n_t,) + tuple(sub[10:]))
                    t0 += n_t
            subs = cut
    tall = [x for x in subs if trows(x) != 128]
    subs = [x for x in subs if trows(x) == 128]
    launches = [(subs, False)] if subs else []
    if allow_sync and p16 and atomic:
        classes = {}
        for sub in subs:
            classes.setdefault(sub[9], []).append(sub)
        if len(classes) == 1:
            launches = [(subs, True)]
        elif classes:
            big = [(t, c) for t, c in classes.items() if sum(tiles_of(x) for x in c) >= 1024]
            rest = [x for t, c in classes.items() if sum(tiles_of(y) for y in c) < 1024 for x in c]
            launches = [(c, True) for _, c in sorted(big, key=lambda tc: -tc[0])] + ([(rest, False)] if rest else [])
    if tall:   # one 256-row launch per token count (panel-synchronous when allowed), ahead of the 128-row launches
        tclasses = {}
        for sub in tall:
            tclasses.setdefault(sub[9], []).append(sub)
        launches = [(c, bool(allow_sync)) for _, c in sorted(tclasses.items(), key=lambda tc: -tc[0])] + launches
    return launches


def _launch_wgrad_group(its, atomic=1, allow_sync=True):
    groups = {}
    for it in its:
        p16 = it[9]
        groups.setdefault((176 if p16 else int(lib.vptr_gemm_tile_cols(it[4])), it[6], p16), []).append(it)
    for (cols, prec, p16), grp in groups.items():
        # tile rows of this flush: a fixed setting, or -- VPTR_WGRAD_ROWS=auto, the default -- whichever of 128 / 256 ran faster on THIS set
        # of problems (measured once per problem set with HIP events on the launch stream, during the eager warm-up steps every caller
        # runs before it times or captures anything: the two settings trade a better tile for a second launch with a tail of its own, and
        # which side wins depends on the model -- K64 7.06 vs 7.25 ms, KTH128 11.0 vs 12.2, BAIR FAR 16.7 vs 15.0)
        rows_mode = config.wgrad_rows
        tune = None
        if rows_mode == "auto":
            rows_mode = 128
            if p16 and atomic and len(grp) > 3:
                sig = (allow_sync,) + tuple(sorted((it[3], it[4], it[5]) for it in grp))
                tune = _wgrad_tune.setdefault(sig, {"ms": {}, "pending": None, "choice": None})
                capturing = torch.cuda.is_current_stream_capturing()     # (no event queries under capture: wgrad_tune_settle ran before it)
                if not capturing and tune["pending"] is not None and tune["pending"][2].query():     # the timed flush has finished: book it
                    _wgrad_tune_book(tune)
                if tune["choice"] is not None:
                    rows_mode, tune = tune["choice"], None
                elif capturing or tune["pending"] is not None:
                    rows_mode, tune = (min(tune["ms"], key=tune["ms"].get) if tune["ms"] else 128), None   # no timing now: best known so far
                else:
                    rows_mode = 256 if 128 in tune["ms"] else 128
        # problems: (A ptr, B ptr, D ptr, rowsum ptr, lda, ldb, ldd, rows, cols, tokens, alpha, transposed); plan_wgrad_launches cuts them
        # into the sub-problems of one or more launches
        probs = []
        flops = 0.0
        for (g, x, dW, N, K, M, _, db, alpha, _p) in grp:
            flops += 2.0 * M * N * K
            # token-major P16 problems: put the 176-wide tile side on the dimension it divides.  dW[528][2112] as 128 x 176 tiles of
            # (rows of dW) x (columns) is 5 x 12 tiles with every fifth row tile one-eighth full; computed as X^T . dY and stored
            # transposed (vptr_gemm_desc.d_transposed) it is 17 x 3 tiles.  The bias gradient then needs >= 32 tile rows beyond K.
            flip = (p16 and config.wgrad_flip and N < K and N % 176 == 0 and K % 128 != 0 and 128 - K % 128 >= 32)
            if flip:
                a, b, rows_, cols_, lda, ldb = x, g, K, N, x.stride(0), g.stride(0)
            else:
                a, b, rows_, cols_, lda, ldb = g, x, N, K, g.stride(0), x.stride(0)
            ap, bp, dp, rp, ldd = a.data_ptr(), b.data_ptr(), dW.data_ptr(), (db.data_ptr() if db is not None else 0), dW.stride(0)
            probs.append((ap, bp, dp, rp, lda, ldb, ldd, rows_, cols_, M, alpha, flip))
        launches = plan_wgrad_launches(probs, cols, p16, atomic, allow_sync, rows_mode, split_rem=config.wgrad_split,
                                       token_split=config.wgrad_token_split and not config.deterministic)

        def trows(sub):
            return sub[12] if len(sub) > 12 else 128

        def tiles_of(sub):
            return ((sub[7] + trows(sub) - 1) // trows(sub)) * ((sub[8] + cols - 1) // cols)
        dev = grp[0][0].device
        import struct
        if tune is not None:
            t_e0, t_e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t_e0.record()
        for lsubs, vouch in launches:
            n = len(lsubs)
            descs = (GemmDesc * n)()
            starts = []
            total = 0
            lflops = 0.0
            tr = trows(lsubs[0])
            for i, sub in enumerate(lsubs):
                (ap, bp, dp, rp, lda, ldb, ldd, rows_, cols_, M, alpha, flip) = sub[:12]
                d = descs[i]
                d.precision, d.split_k, d.atomic, d.alpha = prec, 1, int(atomic), alpha
                d.A, d.B, d.D, d.a_rowsum = ap, bp, dp, (rp or None)
                d.lda, d.ldb, d.ldd = lda, ldb, ldd
                d.M, d.N, d.K = rows_, cols_, M
                d.d_transposed = int(flip)
                d.a_mode, d.b_mode = (A_P16T, B_P16T) if p16 else (1, 1)
                starts.append(total)
                total += tiles_of(lsubs[i])
                lflops += 2.0 * rows_ * cols_ * M
            if tr == 256:
                descs[0].split_k = -2 if vouch else -3   # 256-row tiles: panel-synchronous / plain (include/vptr_hip.h)
            elif tr == 192:
                descs[0].split_k = -4 if vouch else -5   # 192-row tiles, three stages
            elif vouch:
                descs[0].split_k = -1     # every problem walks the same number of tokens: the panel-synchronous launch may serve the group (VPTR_WGRAD_SYNC)
            raw = _to_device_async(bytes(descs), dev)
            st = _to_device_async(struct.pack("%di" % len(starts), *starts), dev)
            prof = _gemm_prof
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            check(lib.vptr_gemm_grouped(ctypes.byref(descs[0]), ptr(raw), ptr(st), n, total, stream()), "vptr_gemm_grouped")
            if prof is not None:
                e1.record()
                # (which kernel the launcher picks for this group: the panel-synchronous persistent one needs VPTR_WGRAD_SYNC != 0 (default 16), the
                # uniform-token vouch and >= 1024 tiles -- mirrored here so that bench.py names the kernel rocprofv3 will list)
                sync = p16 and atomic and vouch and total >= (512 if tr != 128 else 1024) and os.environ.get("VPTR_WGRAD_SYNC", "16") not in ("0", "")
                prof.append(((cols // 16, prec, 6 if p16 else 1, 4 if p16 else 1, (("grouped_sync%d" % tr if tr != 128 else "grouped_sync") if sync else "grouped") if atomic else "grouped_split"),
                             lflops, e0, e1))
        if tune is not None:
            t_e1.record()
            tune["pending"] = (rows_mode, t_e0, t_e1)


def convt_weight_grads(layers, tokens_per_split=2560):
    """Weight gradients of ConvTranspose2d(3x3, s2, p1, op1) layers, D[ci][(ky, kx, co)] = sum_pix x[pix][ci] * P[pix][(ky, kx, co)] with
    P = im2col (3x3, s2, p1) of the output gradient, on the token-major P16 kernel of the grouped weight-gradient launch.  Alone such
    a problem is 4 - 70 output tiles over up to 164 k tokens, so every layer is cut into token ranges of `tokens_per_split`; all ranges
    of all layers run as ONE grouped launch (plain stores into per-range buffers), one vptr_partial_reduce launch adds them up.
    layers: (x [pix, ci] fp32, g [B * oh * ow, co] fp32, B, ih, iw, ci, oh, ow, co); returns the D tensors [ci, 9 * co]."""
    items, keep, outs, red = [], [], [], []
    for (x, g, B, ih, iw, ci, oh, ow, co) in layers:
        pix = B * ih * iw
        P = torch.empty((pix, 9 * co), device=g.device, dtype=torch.float32)      # P16
        check(lib.vptr_im2col_nhwc_p16(ptr(g), ptr(P), B, oh, ow, co, ih, iw, 3, 3, 2, 1, 0, stream()), "vptr_im2col_nhwc_p16")
        xs = to_p16(x)
        S = max(1, pix // int(tokens_per_split))
        step = (pix + S - 1) // S
        step = (step + 31) // 32 * 32                                              # whole 32-token steps per range
        S = (pix + step - 1) // step
        part = torch.empty((S, ci, 9 * co), device=g.device, dtype=torch.float32)
        for k in range(S):
            r0, r1 = k * step, min(pix, (k + 1) * step)
            items.append((xs[r0:r1], P[r0:r1], part[k], ci, 9 * co, r1 - r0, config.gemm_precision, None, 1.0, True))
        D = torch.zeros((ci, 9 * co), device=g.device, dtype=torch.float32)
        red.append((part, D, S, ci * 9 * co))
        keep += [P, xs, part]
        outs.append(D)
    _launch_wgrad_group(items, atomic=0)
    tab = (_lib.ReduceEntry * len(red))()
    for i, (part, D, S, C) in enumerate(red):
        tab[i].part, tab[i].dst0, tab[i].dst1, tab[i].nparts, tab[i].C = ptr(part), ptr(D), None, S, C
    raw = _to_device_async(bytes(tab), outs[0].device)
    esz = ctypes.sizeof(_lib.ReduceEntry)
    for i, (part, D, S, C) in enumerate(red):   # one launch per layer: the widths differ by 4x
        check(lib.vptr_partial_reduce(ctypes.c_void_p(raw.data_ptr() + i * esz), 1, C, 1, stream()), "vptr_partial_reduce")
    return outs


def flush_wgrads(chunks=1, on_chunk=None):
    """Launch every recorded weight gradient (idempotent; runs automatically at the end of a backward pass).

    chunks > 1: the records are ordered by the address of their destination and launched as `chunks` grouped GEMMs of about
    equal work; after each launch `on_chunk(first_dW_ptr)` is called with the lowest destination address of the NEXT chunk
    (None after the last): everything below it is final, so its gradient range can go out to the other ranks while the next
    chunk computes."""
    flush_partial_reduces()
    if not _wgrad_q:
        if on_chunk is not None:
            on_chunk(None)
        return
    items = list(_wgrad_q)
    del _wgrad_q[:]
    if chunks <= 1 or len(items) < 2 * chunks:
        # largest problems first (their 51-tile waves fill the chip; the 15-tile problems then pack the tail), problems that read the
        # same X next to each other: 8.55 -> 8.42 ms on the K64 step's 196 problems (tools/wgrad_ab.sh)
        items.sort(key=lambda it: (-it[3] * it[4], it[1].data_ptr(), it[2].data_ptr()))
        _launch_wgrad_group(items)
        if on_chunk is not None:
            on_chunk(None)
        return
    items.sort(key=lambda it: it[2].data_ptr())
    work = [float(it[3]) * it[4] for it in items]
    per = sum(work) / chunks
    acc, lo = 0.0, 0
    bounds = []
    for i, w in enumerate(work):
        acc += w
        if acc >= per * (len(bounds) + 1) and len(bounds) < chunks - 1 and i + 1 < len(items):
            bounds.append(i + 1)
    bounds.append(len(items))
    for hi in bounds:
        # the chunk boundaries follow slab addresses (what makes a gradient range final); INSIDE a chunk the single-launch order applies:
        # largest problems first, problems that read the same X next to each other
        # plain launch for the chunks: the persistent panel-synchronous kernel assumes that ALL its 512 workgroups are resident at once (every
        # CU's whole LDS), and a chunk runs beside the all-reduce kernels of the previous one -- a displaced workgroup would cost the others
        # a bounded-spin time-out (VPTR_WGRAD_SYNC_CHUNKS=1 allows it anyway)
        _launch_wgrad_group(sorted(items[lo:hi], key=lambda it: (-it[3] * it[4], it[1].data_ptr(), it[2].data_ptr())),
                            allow_sync=os.environ.get("VPTR_WGRAD_SYNC_CHUNKS") == "1")
        if on_chunk is not None:
            on_chunk(items[hi][2].data_ptr() if hi < len(items) else None)
        lo = hi


def _split_k_for(tiles, K):
    """Enough K-splits to put >= ~512 workgroups on the 256 CUs, each split >= 256 deep."""
    if tiles >= 384:
        return 1
    return max(1, min((512 + tiles - 1) // tiles, K // 256))


_flat_slabs = []  # (param_base_ptr, nbytes, grad_slab) registered by vptr_amd.train.FlatAdamW


def register_flat_slab(param_slab, grad_slab):
    """Parameters that live inside `param_slab` have their gradient at the same offset of `grad_slab`: backward kernels
    then accumulate weight gradients straight into the slab (fp32 atomics) instead of materialising a zero-filled
    temporary that autograd adds to `.grad` (2 extra launches and 3 passes over every parameter per step)."""
    import weakref
    _flat_slabs.append((param_slab.data_ptr(), param_slab.numel() * 4, weakref.ref(param_slab), weakref.ref(grad_slab)))


def unregister_flat_slabs():
    del _flat_slabs[:]


def unregister_flat_slab(param_slab):
    """drop the registration of ONE slab (FlatAdamW.close / __del__) and of slabs that no longer exist -- by identity, not by address:
    a later optimizer's slab may have been given the address of a collected one"""
    _flat_slabs[:] = [e for e in _flat_slabs if e[2]() is not None and e[2]() is not param_slab]


def flat_grad_for(t):
    """Gradient-slab view for a parameter tensor (or a contiguous slice of one) that lives in a registered slab."""
    if t is None or not _flat_slabs or not t.is_contiguous():
        return None
    p = t.data_ptr()
    for base, nbytes, pref, gref in _flat_slabs:
        if base <= p < base + nbytes:
            pslab, gslab = pref(), gref()
            if pslab is None or gslab is None or pslab.data_ptr() != base:
                continue  # stale registration (the optimizer that owned the slab is gone)
            off = (p - base) // 4
            return gslab[off:off + t.numel()].view(t.shape)
    return None


# ---- gradient arena of a model used WITHOUT a trainer (the reference's scripts: zero_grad(set_to_none=True) every iteration) ----------
# After set_to_none every parameter's first gradient of the next backward pass needs a zero-filled `.grad` to accumulate into: as
# torch.zeros_like per parameter that is 664 allocations + 664 fill launches per K64 iteration (tools/dropin_prof.py).  A model that
# ran ensure_module_planes() owns ONE flat fp32 buffer instead: a forward pass that finds every `.grad` None zero-fills it with one
# launch, and in backward a parameter's `.grad` becomes a view of its (still zero) range -- contiguous, own shape: torch.optim and
# clip_grad_norm_ see ordinary tensors.  A range is handed out once per fill; anything else (a `.grad` set to None by hand between two
# backward passes, ...) falls back to a fresh zeros_like.  A `.grad` tensor (or a view of one) somebody KEPT from the previous iteration
# is never overwritten: the fill sees the extra reference on the buffer's storage and takes a new buffer for this iteration, the kept
# tensors keep the old one alive (stock-autograd semantics; config.loose_grad_arena = False restores per-parameter tensors).
_grad_arenas = {}     # id(param) -> (weakref(param), weakref(arena), offset); the arena itself is owned by the model's weight-plane store


def _storage_refs(t):
    """number of live tensors (views included) that share t's storage, + the temporary wrapper of this query; a very large number
    when the runtime cannot tell (the arena then always takes a fresh buffer: correct, merely slower)"""
    f = getattr(torch._C, "_storage_Use_Count", None)
    if f is None:
        return 1 << 30
    return int(f(t.untyped_storage()._cdata))


class _GradArena:
    """one flat gradient buffer per model; lives as long as the model's `_vptr_planes` store does (module.__dict__)"""

    def __init__(self, params):
        import weakref
        self.buf = torch.empty(sum(p.numel() for p in params), device=params[0].device, dtype=torch.float32)
        self.base_refs = _storage_refs(self.buf)     # the buffer alone: anything above it at arm time is a gradient somebody kept
        self.clean = False
        self.handed = set()
        self.params = [weakref.ref(p) for p in params]


def _register_grad_arena(module):
    import weakref
    params = [p for p in module.parameters() if p.requires_grad and p.dtype == torch.float32 and p.is_contiguous()]
    if not params or not config.loose_grad_arena:
        return None
    arena = _GradArena(params)
    aref = weakref.ref(arena)
    for k in [k for k, e in _grad_arenas.items() if e[0]() is None or e[1]() is None]:   # entries of models that are gone
        del _grad_arenas[k]
    off = 0
    for p in params:
        _grad_arenas[id(p)] = (weakref.ref(p), aref, off)
        off += p.numel()
    return arena


def _arm_grad_arena(arena):
    """forward pass: with every gradient None (the iteration began with zero_grad(set_to_none=True)) the arena is zero-filled"""
    for r in arena.params:
        p = r()
        if p is not None and p.grad is not None:
            return
    if _storage_refs(arena.buf) > arena.base_refs:
        # somebody KEPT a gradient of the previous iteration (a stashed `p.grad` or a view of it: per-task gradients, logging, manual
        # accumulation): stock autograd would never touch that tensor again, so it keeps the old buffer and this iteration gets a new one
        arena.buf = torch.empty_like(arena.buf)
        arena.base_refs = _storage_refs(arena.buf)
    arena.buf.zero_()
    arena.handed.clear()
    arena.clean = True


def _arena_grad_for(base):
    ent = _grad_arenas.get(id(base))
    arena = ent[1]() if ent is not None and ent[0]() is base else None
    if arena is None or not arena.clean or id(base) in arena.handed or arena.buf.device != base.device:
        return torch.zeros_like(base)
    arena.handed.add(id(base))
    return arena.buf[ent[2]:ent[2] + base.numel()].view(base.shape)


def _engine_accumulates_into(leaf):
    """True when the running backward pass is one that ACCUMULATES into `leaf.grad` (`loss.backward()`, or `backward(inputs=[...
    leaf ...])`), False under `torch.autograd.grad(...)` / `backward(inputs=<others>)`: there the engine captures or drops the
    gradient, and writing `.grad` behind its back would hand the caller None and pollute `.grad`.  The engine is
    asked through `torch._C._will_engine_execute_node` on the leaf's AccumulateGrad node: with no explicit inputs every node of the
    graph executes (True); with inputs it answers for this node, and raises for a leaf that `autograd.grad` captures."""
    will = getattr(torch._C, "_will_engine_execute_node", None)
    if will is None:      # a torch build without the query: take the conservative autograd hand-off
        return False
    # the AccumulateGrad node of a leaf is unique, but it OWNS its variable: a process-wide cache of nodes would pin every parameter
    # (with its .grad and the arena range it views) for the life of the process.  The cache therefore lives for ONE
    # backward pass: keyed by the engine's graph-task id, emptied by an end-of-backward callback (and by the next pass, should the
    # callback of a failed pass never have run).
    task = torch._C._current_graph_task_id()
    if _acc_nodes["task"] != task:
        _acc_nodes["nodes"].clear()
        _acc_nodes["task"] = task
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_drop_acc_nodes)
        except RuntimeError:   # not inside a backward pass
            pass
    acc = _acc_nodes["nodes"].get(id(leaf))
    if acc is None:
        with torch.enable_grad():
            acc = leaf.view_as(leaf).grad_fn.next_functions[0][0]
        _acc_nodes["nodes"][id(leaf)] = acc
    try:
        return bool(will(acc))
    except (RuntimeError, TypeError):
        return False


_acc_nodes = {"task": None, "nodes": {}}    # graph-task id -> {id(leaf): its AccumulateGrad node}, for the running backward pass only


def _drop_acc_nodes():
    _acc_nodes["nodes"].clear()
    _acc_nodes["task"] = None


def _loose_grad_for(t):
    """Gradient destination for a parameter OUTSIDE any flat slab (the reference's scripts: plain nn.Parameters, torch.optim.AdamW,
    zero_grad(set_to_none=True)): a view into the `.grad` of the leaf parameter that `t` is (or is a contiguous view of -- a row
    block of in_proj_weight, a 1x1 conv weight seen as [N, K]), created zero-filled if it is None.  The weight gradient can then
    join the grouped end-of-backward launch exactly like a slab-backed one, instead of running as a launch of its own (12-60 tiles
    with a 10 240-token K loop on 256 CUs: 196 of those made the script-style step 2.7x slower than NARTrainer's).  Returns None
    -- the caller then hands a fresh tensor to autograd -- outside a backward pass, under `torch.autograd.grad` / `backward(inputs=
    ...)` without this parameter / `create_graph=True` (the engine is not accumulating into `.grad` there), in a torch.distributed job
    (DDP's reducer must see gradients arrive through AccumulateGrad hooks) and for parameters with hooks."""
    if not (config.group_wgrads and config.group_loose_wgrads) or t is None or not t.is_contiguous():
        return None
    if torch._C._current_graph_task_id() < 0:
        return None
    if torch.is_grad_enabled():      # backward(create_graph=True): the gradient must stay a graph output, not an in-place sum
        return None
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return None
    base = t if t.is_leaf else t._base
    if base is None or not base.is_leaf or not base.requires_grad or not base.is_contiguous() or base.dtype != torch.float32:
        return None
    if base._backward_hooks or getattr(base, "_post_accumulate_grad_hooks", None):
        return None
    if not _engine_accumulates_into(base):
        return None
    off = (t.data_ptr() - base.data_ptr()) // 4
    if off < 0 or off + t.numel() > base.numel():
        return None
    if base.grad is None:
        base.grad = _arena_grad_for(base)
    elif not base.grad.is_contiguous() or base.grad.dtype != torch.float32:
        return None
    return base.grad.view(-1)[off:off + t.numel()].view(t.shape)


def grad_dest_for(t):
    """where a parameter's gradient is accumulated in place: its range of a registered flat gradient slab, else its own `.grad`"""
    d = flat_grad_for(t)
    return d if d is not None else _loose_grad_for(t)


_bw_blocks = {}     # device key -> [graph-task id, current zeroed block, floats used]
_BW_BLOCK = 16 << 20   # floats per block (64 MB)


def _bw_zeros(shape, device):
    """Zero-filled fp32 tensor for a parameter gradient that is handed to autograd (torch.distributed jobs -- DDP's reducer must see every
    gradient arrive through its AccumulateGrad hook --, torch.autograd.grad, parameters with hooks): a slice of a 64 MB block zeroed by ONE
    fill per block and backward pass instead of one allocation + fill launch per parameter (a stock-DDP K64 iteration spent 1100 launches /
    3.9 ms of GPU time on those fills).  Blocks are never reused or re-zeroed: a gradient somebody keeps keeps its block alive; the next
    backward pass (another graph-task id) starts new blocks."""
    n = 1
    for d in shape:
        n *= int(d)
    task = torch._C._current_graph_task_id()
    if task < 0 or n == 0 or n > _BW_BLOCK:
        return torch.zeros(tuple(shape), device=device, dtype=torch.float32)
    key = _dev_key(device)
    st = _bw_blocks.get(key)
    if st is None or st[0] != task:
        st = _bw_blocks[key] = [task, None, 0]
    n_al = (n + 63) // 64 * 64     # 256-byte aligned slices
    if st[1] is None or st[2] + n_al > st[1].numel():
        st[1] = torch.zeros(_BW_BLOCK, device=device, dtype=torch.float32)
        st[2] = 0
    v = st[1][st[2]:st[2] + n].view(tuple(shape))
    st[2] += n_al
    return v


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
    """frame_stats / frame_rows: a zeroed [rows / frame_rows, 2] buffer (frame_stats_buffer) that the GEMM epilogue fills with each
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


def frame_stats_buffer(frames, device):
    """a zeroed [frames, 2] fp32 buffer for one producer / consumer pair (a slice of the forward's zero_arena when one is open;
    inside a graph capture the arena's fill is re-run at every replay)"""
    n = 2 * int(frames)
    buf = _zero_arena["buf"]
    if buf is not None and buf.device == torch.device(device) and _zero_arena["off"] + n <= buf.numel():
        off = _zero_arena["off"]
        _zero_arena["off"] = off + n
        return buf[off:off + n].view(frames, 2)
    return torch.zeros((frames, 2), device=device, dtype=torch.float32)


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


# ------------------------------------------------------------------------------------------------------------------
# conv-FFN pieces
# ------------------------------------------------------------------------------------------------------------------
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
        rows, F = x.shape
        dx = torch.empty_like(x)
        sw, sb = flat_grad_for(w), flat_grad_for(b)   # accumulate straight into the flat gradient slab when both live there
        in_slab = sw is not None and sb is not None
        dw, db = (sw, sb) if in_slab else (_bw_zeros(w.shape, w.device), _bw_zeros(b.shape, b.device))
        frames = rows // HW
        scratch = torch.empty((max(2 * F, 2 * frames * (1 + 4 * ((HW * F // 4 + 255) // 256))),), device=x.device, dtype=torch.float32)
        nparts = lib.vptr_norm_act_bwd_partials(rows, F, HW, int(per_col)) if (in_slab and config.defer_ln_param_grads) else 0
        if nparts > 0:
            # affine gradients with an in-place destination: per-chunk partial sums, added by the backward pass's one reduction launch
            part = torch.empty((nparts, 2, HW * F), device=x.device, dtype=torch.float32)
            check(lib.vptr_norm_act_bwd_deferred(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(w), ptr(b), ptr(dx), ptr(scratch), rows, F, HW,
                                                 act, int(const_stats), p, ptr(ctx.seed), site, ptr(rowscale), rs_div, rs_mod, int(dx_p16),
                                                 ptr(part), stream()), "vptr_norm_act_bwd_deferred")
            defer_partial_reduce(part, sw, sb, nparts, HW * F)
        else:
            check(lib.vptr_norm_act_bwd(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(w), ptr(b), ptr(dx), ptr(dw), ptr(db),
                                        ptr(scratch), rows, F, HW, int(per_col), act, int(const_stats), p,
                                        ptr(ctx.seed), site, ptr(rowscale), rs_div, rs_mod, int(dx_p16),
                                        stream()), "vptr_norm_act_bwd")
        dres = dy if has_res else None
        if in_slab:
            dw = db = None
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


# ------------------------------------------------------------------------------------------------------------------
# auto-encoder convolutions (implicit GEMM on NHWC) -- see vptr_amd/model/autoencoder.py for the layer wiring
# ------------------------------------------------------------------------------------------------------------------
class SubpixelWeights:
    """ConvTranspose2d(3x3, stride 2, pad 1, output_padding 1) weight [Cin, Cout, 3, 3] split by output parity: class (py, px)
    keeps the taps that reach output pixels (2y + py, 2x + px): ky = 1 for py = 0; for py = 1 the taps ky = 2 (input row y) and
    ky = 0 (input row y + 1), in the order a stride-1, pad-0 gather with KH' = 2 walks them; likewise in x.
    classes = [(py, px, B[n = Cout][k = (ky', kx', ci)])]; `full` = the 9-tap gather-form matrix (residual epilogues, other geometries)."""

    def __init__(self, weight):
        self.full = weight.permute(1, 2, 3, 0).reshape(weight.shape[1], -1).contiguous()
        taps = {0: [1], 1: [2, 0]}
        self.classes = []
        for py in (0, 1):
            for px in (0, 1):
                w = weight[:, :, taps[py]][:, :, :, taps[px]]                      # [Cin, Cout, KH', KW']
                self.classes.append((py, px, w.permute(1, 2, 3, 0).reshape(weight.shape[1], -1).contiguous()))


def conv_weight_as_gemm_b(weight, transposed):
    """PyTorch conv weight -> B[n = Cout][k = (ky, kx, ci)] (k contiguous).

    Conv2d weight [Cout, Cin, KH, KW]; ConvTranspose2d weight [Cin, Cout, KH, KW] (3x3: a SubpixelWeights, see conv_nhwc).
    """
    def pack():
        if transposed:
            if config.subpixel_convt and weight.shape[2] == 3 and weight.shape[3] == 3:
                return SubpixelWeights(weight)
            return weight.permute(1, 2, 3, 0).reshape(weight.shape[1], -1).contiguous()
        return weight.permute(0, 2, 3, 1).reshape(weight.shape[0], -1).contiguous()

    if not config.weights_frozen:
        return pack()
    # frozen_weights scope (stage 2: the auto-encoder is never stepped): the packed copy is kept on the parameter object,
    # keyed by version counter and address; otherwise ~30 conv weights (up to 10 MB each) are re-packed 3x per step
    attr = "_vptr_packed_t" if transposed else "_vptr_packed"
    key = (weight._version, weight.data_ptr())
    hit = getattr(weight, attr, None)
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        B = pack()
    setattr(weight, attr, (key, B))
    return B


def in_flat_slab(t):
    """True when `t`'s storage lies inside a registered optimizer slab (FlatAdamW): such parameters are stepped by raw kernels that
    bump no version counter"""
    p = t.data_ptr()
    for base, nbytes, pref, _ in _flat_slabs:
        if base <= p < base + nbytes and pref() is not None:
            return True
    return False


def weights_cacheable(module):
    """May packed conv weights / eval-BN folds of `module` be cached under their tensors' version counters?  Yes when a trainer marked
    the module `_vptr_frozen`; else -- the reference's scripts, which drive plain modules (train_NAR.py:190-191: Enc / Dec in eval mode,
    never stepped) -- when the module is in eval mode and its parameters do not live in an optimizer slab: torch.optim, load_state_dict
    and the c10d broadcasts all write through versioned in-place ops, so a stale entry cannot be hit.  (Writes through `.data` are
    invisible to version counters: call ops.invalidate_weight_planes() / re-create the module after such a write.)"""
    flag = getattr(module, "_vptr_frozen", None)
    if flag is not None:
        return bool(flag)
    if module.training:
        return False
    p = next(module.parameters(), None)
    return p is not None and p.is_cuda and not in_flat_slab(p)


class frozen_weights:
    """Scope in which derived copies of module weights (packed conv weights, eval-BN folds) may be cached on the module /
    parameter objects.  Entered by the auto-encoder modules that `NARTrainer` marks `_vptr_frozen` (stage 2 never steps
    them); a cache entry is keyed by the tensors' version counters and addresses, so `load_state_dict` / optimizer steps
    invalidate it -- but writes through `.data` or raw kernels would not, hence the explicit opt-in."""

    def __init__(self, flag):
        self.flag = bool(flag)

    def __enter__(self):
        self.prev = config.weights_frozen
        config.weights_frozen = self.flag
        return self

    def __exit__(self, *exc):
        config.weights_frozen = self.prev
        return False


def conv_nhwc(x, Bmat, frames, IH, IW, Cin, OH, OW, KH, KW, stride, pad, pad_mode, transposed, Cout, colscale=None,
              bias=None, act=ACT_NONE, residual=None, act_after=False):
    """Implicit-GEMM convolution: x NHWC [frames*IH*IW, Cin] -> [frames*OH*OW, Cout] with fused folded-BN/ReLU/residual."""
    M = frames * OH * OW
    y = torch.empty((M, Cout), device=x.device, dtype=torch.float32)
    if (transposed and isinstance(Bmat, SubpixelWeights) and residual is None and KH == 3 and KW == 3 and stride == 2 and pad == 1
            and OH == 2 * IH and OW == 2 * IW and Cin % 4 == 0):
        # ConvTranspose2d(3x3, stride 2, pad 1, output_padding 1) as its four output-parity classes: class (py, px) is a stride-1
        # gather over the INPUT grid with (1 + py) x (1 + px) taps whose rows land on output pixels (2y + py, 2x + px) through the
        # GEMM's output row map -- 2.25 taps per output pixel on average instead of the 9-tap gather form's 6.75 zero products
        for (py, px, Bc) in Bmat.classes:
            gemm_raw(x, Bc, y[:, :] if px == 0 else y.view(-1)[px * Cout:], frames * IH * IW, Cout, (1 + py) * (1 + px) * Cin, 2, 0, lda=0,
                     colscale=colscale, bias=bias, act=act, act_after=act_after, ldd=2 * Cout, row_map=(IW, py * IW),
                     conv=(IH, IW, Cin, IH, IW, 1 + py, 1 + px, 1, 0, PAD_MODES["zero"], 0))
        return y
    if isinstance(Bmat, SubpixelWeights):
        Bmat = Bmat.full
    gemm_raw(x, Bmat, y, M, Cout, KH * KW * Cin, 2, 0, lda=0, colscale=colscale, bias=bias, act=act, residual=residual,
             act_after=act_after, conv=(IH, IW, Cin, OH, OW, KH, KW, stride, pad, PAD_MODES[pad_mode], int(transposed)))
    return y


# ---- "convert once" operands (bf16 hi / lo planes) for the frozen encoder's convolutions ------------------------------------
def split_planes(x, out=None):
    """x [rows, C] fp32 -> planes [(rows + 1), ceil(C / 32), 64] bf16 (hi 32 | lo 32 per block; last row and pad channels zero):
    the operand format of conv_nhwc_planes (include/vptr_hip.h, vptr_split_planes)."""
    x = _c(x)
    rows, C = x.shape
    if out is None:
        out = torch.empty((rows + 1, (C + 31) // 32, 64), device=x.device, dtype=torch.bfloat16)
    check(lib.vptr_split_planes(ptr(x), ptr(out), rows, C, stream()), "vptr_split_planes")
    return out


def conv_weight_as_planes(weight):
    """Conv2d weight [Cout, Cin, KH, KW] -> plane form B[n][tap][ceil(Cin / 32)][hi 32 | lo 32] bf16.  Only inside a
    frozen_weights scope (the copy is cached on the parameter, keyed by version and address)."""
    if not config.weights_frozen:
        raise RuntimeError("conv_weight_as_planes: plane weights are only kept for frozen modules")
    key = (weight._version, weight.data_ptr())
    hit = getattr(weight, "_vptr_planes", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        Cout, Cin, KH, KW = weight.shape
        CB = (Cin + 31) // 32
        w = weight.permute(0, 2, 3, 1).reshape(Cout, KH * KW, Cin).float()
        w = torch.nn.functional.pad(w, (0, CB * 32 - Cin)).reshape(Cout, KH * KW, CB, 32)
        hi = w.to(torch.bfloat16)
        lo = (w - hi.float()).to(torch.bfloat16)
        B = torch.stack([hi, lo], dim=3).contiguous()       # [Cout, taps, CB, 2, 32]
    setattr(weight, "_vptr_planes", (key, B))
    return B


def conv_nhwc_planes(x_planes, Bplanes, frames, IH, IW, Cin, OH, OW, KH, KW, stride, pad, pad_mode, Cout, colscale=None, bias=None,
                     act=ACT_NONE, residual=None, act_after=False, planes_out=None, fp32_out=True):
    """conv_nhwc with both operands in plane form (x_planes from split_planes, Bplanes from conv_weight_as_planes): the GEMM
    stages them with global_load_lds -- no fp32 -> bf16 split and no LDS stores in its main loop.  planes_out (a ZEROED
    [(M + 1), ceil(Cout / 32), 64] bf16 buffer, reusable) also receives the result in plane form for the next plane conv;
    fp32_out=False then skips the fp32 copy.  Returns the fp32 output (or None)."""
    M = frames * OH * OW
    y = torch.empty((M, Cout), device=x_planes.device, dtype=torch.float32) if fp32_out else None
    if y is None and planes_out is None:
        raise RuntimeError("conv_nhwc_planes: no output requested")
    gemm_raw(x_planes, Bplanes, y, M, Cout, KH * KW * Cin, 3, 2, lda=0, ldb=0, colscale=colscale, bias=bias, act=act, residual=residual,
             act_after=act_after, conv=(IH, IW, Cin, OH, OW, KH, KW, stride, pad, PAD_MODES[pad_mode], 0), precision=3, planes_out=planes_out)
    return y


# ---- trainable convolutions (stage-1 auto-encoder / PatchGAN training, train_AutoEncoder.py:44-86) -------------------------
class _Conv2dNHWCFn(torch.autograd.Function):
    """nn.Conv2d / nn.ConvTranspose2d on NHWC token grids with full autograd, every piece an MFMA GEMM:
    forward  = implicit-GEMM gather (vptr_gemm, a_mode = conv);
    dgrad    = the gather-form transposed convolution of dy (Conv2d) / the strided convolution of dy (ConvTranspose2d);
               reflection padding: gradient on the padded grid, then vptr_reflect_fold;
    wgrad    = vptr_im2col_nhwc + one k-strided x k-strided split-K GEMM."""

    @staticmethod
    def forward(ctx, x, weight, bias, frames, IH, IW, stride, pad, pad_mode, transposed, out_pad, act):
        x = _c(x)
        if transposed:
            Cin, Cout, KH, KW = weight.shape
            OH, OW = (IH - 1) * stride - 2 * pad + KH + out_pad, (IW - 1) * stride - 2 * pad + KW + out_pad
        else:
            Cout, Cin, KH, KW = weight.shape
            OH, OW = (IH + 2 * pad - KH) // stride + 1, (IW + 2 * pad - KW) // stride + 1
        if x.shape != (frames * IH * IW, Cin):
            raise RuntimeError("conv2d_nhwc: input %s does not match frames*IH*IW x Cin = %d x %d" % (tuple(x.shape), frames * IH * IW, Cin))
        if transposed and pad_mode != "zero":
            raise RuntimeError("conv2d_nhwc: ConvTranspose2d supports zero padding only")
        y = conv_nhwc(x, conv_weight_as_gemm_b(weight, transposed), frames, IH, IW, Cin, OH, OW, KH, KW, stride, pad, pad_mode,
                      transposed, Cout, bias=bias, act=act)
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        ctx.cfg = (frames, IH, IW, OH, OW, Cin, Cout, KH, KW, stride, pad, pad_mode, transposed, act, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        frames, IH, IW, OH, OW, Cin, Cout, KH, KW, stride, pad, pad_mode, transposed, act, has_b = ctx.cfg
        dy = _c(dy)
        pix_o, pix_i = frames * OH * OW, frames * IH * IW
        if act != ACT_NONE:  # ReLU / LeakyReLU epilogue: sign of the output decides
            if act == ACT_GELU:
                raise RuntimeError("conv2d_nhwc: GELU epilogue is not differentiable from its output")
            g = torch.empty_like(dy)
            check(lib.vptr_act_bwd(ptr(dy), ptr(y), ptr(g), pix_o, Cout, act, 1.0, None, 1, 1, 0.0, None, 0, 0, stream()), "vptr_act_bwd")
        else:
            g = dy
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            if transposed:   # dx = Conv2d(g, W) with the same stride / padding
                dx = conv_nhwc(g, weight.permute(0, 2, 3, 1).reshape(Cin, -1).contiguous(), frames, OH, OW, Cout, IH, IW, KH, KW,
                               stride, pad, "zero", False, Cin)
            else:
                Bt = weight.permute(1, 2, 3, 0).reshape(Cin, -1).contiguous()      # [ci][(ky, kx, co)]
                if pad_mode == "zero" or pad == 0:
                    dx = conv_nhwc(g, Bt, frames, OH, OW, Cout, IH, IW, KH, KW, stride, pad, "zero", True, Cin)
                elif pad_mode == "reflect" and stride == 1:
                    dxp = conv_nhwc(g, Bt, frames, OH, OW, Cout, IH + 2 * pad, IW + 2 * pad, KH, KW, 1, 0, "zero", True, Cin)
                    dx = torch.empty((pix_i, Cin), device=dy.device, dtype=torch.float32)
                    check(lib.vptr_reflect_fold(ptr(dxp), ptr(dx), frames, IH, IW, Cin, pad, stream()), "vptr_reflect_fold")
                else:
                    raise NotImplementedError("conv2d_nhwc backward: padding mode %r with stride %d" % (pad_mode, stride))
        if ctx.needs_input_grad[1]:
            if transposed:   # dW[ci][co][ky][kx] = sum_pix x[pix][ci] * patches(g)[pix][(ky, kx, co)]
                P = torch.empty((pix_i, KH * KW * Cout), device=dy.device, dtype=torch.float32)
                check(lib.vptr_im2col_nhwc(ptr(g), ptr(P), frames, OH, OW, Cout, IH, IW, KH, KW, stride, pad, 0, stream()), "vptr_im2col_nhwc")
                D = torch.zeros((Cin, KH * KW * Cout), device=dy.device, dtype=torch.float32)
                tiles = ((Cin + 127) // 128) * ((KH * KW * Cout + 175) // 176)
                gemm_raw(x, P, D, Cin, KH * KW * Cout, pix_i, 1, 1, atomic=True, split_k=_split_k_for(tiles, pix_i))
                dW = D.view(Cin, KH, KW, Cout).permute(0, 3, 1, 2).contiguous()
            else:            # dW[co][ci][ky][kx] = sum_pix g[pix][co] * patches(x)[pix][(ky, kx, ci)]
                P = torch.empty((pix_o, KH * KW * Cin), device=dy.device, dtype=torch.float32)
                check(lib.vptr_im2col_nhwc(ptr(x), ptr(P), frames, IH, IW, Cin, OH, OW, KH, KW, stride, pad, PAD_MODES[pad_mode] if pad else 0,
                                           stream()), "vptr_im2col_nhwc")
                D = torch.zeros((Cout, KH * KW * Cin), device=dy.device, dtype=torch.float32)
                tiles = ((Cout + 127) // 128) * ((KH * KW * Cin + 175) // 176)
                gemm_raw(g, P, D, Cout, KH * KW * Cin, pix_o, 1, 1, atomic=True, split_k=_split_k_for(tiles, pix_o))
                dW = D.view(Cout, KH, KW, Cin).permute(0, 3, 1, 2).contiguous()
        if has_b and ctx.needs_input_grad[2]:
            db = torch.zeros((Cout,), device=dy.device, dtype=torch.float32)
            check(lib.vptr_colsum(ptr(g), ptr(db), pix_o, Cout, stream()), "vptr_colsum")
        return dx, dW, db, None, None, None, None, None, None, None, None, None


def conv2d_nhwc(x, weight, bias, frames, IH, IW, stride=1, pad=0, pad_mode="zero", transposed=False, output_padding=0, act=ACT_NONE):
    """Trainable convolution on an NHWC token grid [frames*IH*IW, Cin] -> ([frames*OH*OW, Cout], OH, OW).  Channel counts must be
    multiples of 4 (callers zero-pad 1- and 3-channel ends)."""
    y = _Conv2dNHWCFn.apply(x, weight, bias, int(frames), int(IH), int(IW), int(stride), int(pad), pad_mode, bool(transposed),
                            int(output_padding), int(act))
    if transposed:
        KH, KW = weight.shape[2], weight.shape[3]
        OH, OW = (IH - 1) * stride - 2 * pad + KH + output_padding, (IW - 1) * stride - 2 * pad + KW + output_padding
    else:
        KH, KW = weight.shape[2], weight.shape[3]
        OH, OW = (IH + 2 * pad - KH) // stride + 1, (IW + 2 * pad - KW) // stride + 1
    return y, OH, OW


class _Conv7InFn(torch.autograd.Function):
    """ReflectionPad2d(3) + Conv7x7(Cimg -> 64) of the encoder's first layer, raw output (ResNetAutoEncoder.py:26-27);
    x NCHW -> y NHWC tokens.  Only the weight gradient exists (the input is the image)."""

    @staticmethod
    def forward(ctx, x, weight):
        x, weight = _c(x), _c(weight)
        B, Cimg, H, W = x.shape
        Cout = weight.shape[0]
        y = torch.empty((B * H * W, Cout), device=x.device, dtype=torch.float32)
        check(lib.vptr_conv7_in_fwd(ptr(x), ptr(weight), None, None, ptr(y), B, Cimg, H, W, Cout, stream()), "vptr_conv7_in_fwd")
        ctx.save_for_backward(x)
        ctx.wshape = tuple(weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, Cimg, H, W = x.shape
        dW = torch.zeros(ctx.wshape, device=dy.device, dtype=torch.float32)
        check(lib.vptr_conv7_in_bwd_weight(ptr(_c(dy)), ptr(x), ptr(dW), B, Cimg, H, W, ctx.wshape[0], stream()), "vptr_conv7_in_bwd_weight")
        return None, dW


def conv7_in(x, weight):
    return _Conv7InFn.apply(x, weight)


class _Conv7OutFn(torch.autograd.Function):
    """ReflectionPad2d(3) + Conv7x7(64 -> Cimg) + bias + Tanh / Sigmoid of the decoder's last layer
    (ResNetAutoEncoder.py:89-96); x NHWC tokens [B*H*W, 64] -> y NCHW."""

    @staticmethod
    def forward(ctx, x, weight, bias, B, H, W, out_act):
        x, weight, bias = _c(x), _c(weight), _c(bias)
        Cimg, Cin = weight.shape[0], weight.shape[1]
        y = torch.empty((B, Cimg, H, W), device=x.device, dtype=torch.float32)
        check(lib.vptr_conv7_out_fwd(ptr(x), ptr(weight), ptr(bias), ptr(y), B, Cin, H, W, Cimg, out_act, stream()), "vptr_conv7_out_fwd")
        ctx.save_for_backward(x, weight, y)
        ctx.cfg = (B, H, W, out_act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        B, H, W, out_act = ctx.cfg
        Cimg, Cin = weight.shape[0], weight.shape[1]
        dy = _c(dy)
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B * H * W, Cin), device=dy.device, dtype=torch.float32)
            check(lib.vptr_conv7_out_bwd_data(ptr(dy), ptr(y), ptr(weight), ptr(dx), B, Cin, H, W, Cimg, out_act, stream()),
                  "vptr_conv7_out_bwd_data")
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dW = torch.zeros_like(weight)
            db = torch.zeros((Cimg,), device=dy.device, dtype=torch.float32)
            wsp = torch.empty((lib.vptr_conv7_out_bwd_weight_workspace(B, Cimg),), device=dy.device, dtype=torch.float32)
            check(lib.vptr_conv7_out_bwd_weight_ws(ptr(dy), ptr(y), ptr(x), ptr(dW), ptr(db), B, Cin, H, W, Cimg, out_act, ptr(wsp),
                                                   wsp.numel(), stream()), "vptr_conv7_out_bwd_weight_ws")
        return dx, dW, db, None, None, None, None


def conv7_out(x, weight, bias, B, H, W, out_act):
    return _Conv7OutFn.apply(x, weight, bias, int(B), int(H), int(W), int(out_act))


def bn_fold(bn_weight, bn_bias, running_mean, running_var, eps=1e-5):
    """Eval-mode BatchNorm2d as per-channel (scale, shift)."""
    scale = bn_weight * torch.rsqrt(running_var + eps)
    return scale, bn_bias - running_mean * scale


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
