"""Convolutions of the auto-encoder as implicit GEMMs (frozen plane-operand path, trainable path, 7x7 ends, BatchNorm folding)."""

import torch

from .._lib import check, lib, ptr, stream
from .core import ACT_GELU, ACT_NONE, PAD_MODES, _c, config, gemm_raw
from .wgrad import _split_k_for
from .grads import _flat_slabs


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


# ---- Winograd F(4x4, 3x3) for frozen stride-1 3x3 convolutions (csrc/winograd.hip; ResNetAutoEncoder.py:127-151 under train_NAR.py:54-56) ------
_WINO_G = ((0.25, 0.0, 0.0), (-1.0 / 6, -1.0 / 6, -1.0 / 6), (-1.0 / 6, 1.0 / 6, -1.0 / 6), (1.0 / 24, 1.0 / 12, 1.0 / 6),
           (1.0 / 24, -1.0 / 12, 1.0 / 6), (0.0, 0.0, 1.0))


def wino_ok(H, W, Cin, Cout):
    """can a frozen stride-1, pad-1 3x3 convolution on H x W maps run as Winograd F(4x4, 3x3)?  (whole 4 x 4 output tiles, P16 operands)"""
    return (config.winograd and config.weights_frozen and config.use_p16 and config.gemm_precision == 3 and H % 4 == 0 and W % 4 == 0
            and H >= 4 and W >= 4 and Cin % 16 == 0 and Cout % 16 == 0)


def wino_filter(weight):
    """Conv2d weight [Cout, Cin, 3, 3] -> U [36, Cout, Cin] in the P16 operand format, U[xi nu] = (G g G^T)[xi][nu] (fp64 transform, rounded
    once).  Only inside a frozen_weights scope: the transformed filter is cached on the parameter, keyed by version and address."""
    if not config.weights_frozen:
        raise RuntimeError("wino_filter: transformed filters are only kept for frozen modules")
    key = (weight._version, weight.data_ptr())
    hit = getattr(weight, "_vptr_wino", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        Cout, Cin, KH, KW = weight.shape
        if (KH, KW) != (3, 3):
            raise RuntimeError("wino_filter: 3x3 kernels only")
        G = torch.tensor(_WINO_G, dtype=torch.float64, device=weight.device)
        U = torch.einsum("ia,kcab,jb->ijkc", G, weight.double(), G).reshape(36 * Cout, Cin).float().contiguous()
        Up = torch.empty_like(U)
        check(lib.vptr_to_p16(ptr(U), ptr(Up), U.shape[0], Cin, stream()), "vptr_to_p16")
        Up = Up.view(36, Cout, Cin)
    setattr(weight, "_vptr_wino", (key, Up))
    return Up


def wino_buffers(frames, H, W, C, device, holder):
    """(V, M36, Mpad): the persistent Winograd-domain operand (P16, zero-initialised: rows beyond the tile count are never written) and
    product buffers of one module, keyed on shape -- one forward at a time per module, like the plane buffers"""
    rows = frames * (H // 4) * (W // 4)
    Mpad = (rows + 127) // 128 * 128
    key = ("wino", Mpad, C, device)
    hit = getattr(holder, "_wino_bufs", None)
    if hit is None or hit[0] != key:
        hit = (key, torch.zeros((36, Mpad, C), device=device, dtype=torch.float32), torch.empty((36, Mpad, C), device=device, dtype=torch.float32))
        holder._wino_bufs = hit
    return hit[1], hit[2], Mpad


def wino_in(x, frames, H, W, bufs, pad_mode="zero"):
    """input transform: NHWC tokens x [frames * H * W, C] -> bufs' V (P16, [36][Mpad][C])"""
    V, _, Mpad = bufs
    x = _c(x)
    check(lib.vptr_wino_in(ptr(x), ptr(V), frames, H, W, x.shape[1], Mpad, PAD_MODES[pad_mode], stream()), "vptr_wino_in")


def wino_gemm(U, frames, H, W, bufs):
    """M36[xi nu] = V[xi nu] . U[xi nu]^T: ONE strided-batch launch of the P16 nt GEMM, 36 members"""
    V, M36, Mpad = bufs
    Cout, C = U.shape[1], U.shape[2]
    rows = frames * (H // 4) * (W // 4)
    gemm_raw(V, U, M36, rows, Cout, C, 5, 3, lda=C, ldb=C, ldd=Cout, precision=3, batch_strided=(36, Mpad * C, Cout * C, Mpad * Cout))


def wino_out(frames, H, W, C, bufs, colscale=None, bias=None, relu=False, residual=None, act_after=False, out=None):
    """output transform + folded BN / ReLU / skip: bufs' M36 -> NHWC tokens (out may be `residual` itself)"""
    _, M36, Mpad = bufs
    if out is None:
        out = torch.empty((frames * H * W, C), device=M36.device, dtype=torch.float32)
    check(lib.vptr_wino_out(ptr(M36), ptr(colscale), ptr(bias), ptr(residual), ptr(out), frames, H, W, C, Mpad, int(bool(relu)),
                            int(bool(act_after)), stream()), "vptr_wino_out")
    return out


def wino_fused_ok(H, W):
    """can the output transform of one convolution feed the input transform of the next through LDS? (vptr_wino_out_in: whole maps of at most
    16 tiles per 16-quad channel slab)"""
    T = (H // 4) * (W // 4)
    return config.winograd_fuse and 1 <= T <= 16 and 16 % T == 0


def wino_out_in(frames, H, W, C, bufs, pad_mode="zero", colscale=None, bias=None, relu=False, residual=None, act_after=False, out=None):
    """wino_out of this convolution + wino_in of the next in one pass (the map stays in LDS); `out` = None: the map itself is not kept"""
    V, M36, Mpad = bufs
    check(lib.vptr_wino_out_in(ptr(M36), ptr(colscale), ptr(bias), ptr(residual), ptr(out), ptr(V), frames, H, W, C, Mpad, int(bool(relu)),
                               int(bool(act_after)), PAD_MODES[pad_mode], stream()), "vptr_wino_out_in")
    return out


def wino_conv3x3(x, U, frames, H, W, bufs, pad_mode="zero", colscale=None, bias=None, relu=False, residual=None, act_after=False, out=None):
    """y = [relu](conv3x3(x) * colscale + bias) [+ residual] [relu] on NHWC tokens x [frames * H * W, C] (stride 1, one pixel of `pad_mode`
    padding) as Winograd F(4x4, 3x3): input transform -> ONE strided-batch P16 GEMM of 36 members -> output transform with the epilogue.
    U from wino_filter, bufs from wino_buffers.  out may be `residual` itself."""
    C = x.shape[1]
    if U.shape[1] != C or U.shape[2] != C:
        raise RuntimeError("wino_conv3x3: square convolutions only (C %d, filter %s)" % (C, tuple(U.shape)))
    wino_in(x, frames, H, W, bufs, pad_mode)
    wino_gemm(U, frames, H, W, bufs)
    return wino_out(frames, H, W, C, bufs, colscale, bias, relu, residual, act_after, out)


def wino_resnet_blocks(y, blocks, frames, H, W, bufs, pad_mode, last_relu=True):
    """The ResnetBlock chain of the frozen encoder (ResNetAutoEncoder.py:153-157: y <- y + BN(conv(pad(ReLU(BN(conv(pad(y)))))))) in the Winograd
    domain.  blocks = [(U1, scale1, shift1, U2, scale2, shift2), ...]; y [frames * H * W, C] fp32 is updated in place and returned.  With
    wino_fused_ok the map of every convolution goes to the next one's input transform through LDS: one launch between two GEMMs."""
    C = y.shape[1]
    fused = wino_fused_ok(H, W)
    tbuf = None if fused else torch.empty_like(y)
    wino_in(y, frames, H, W, bufs, pad_mode)
    n = len(blocks)
    for bi, (U1, s1, b1, U2, s2, b2) in enumerate(blocks):
        last = bi == n - 1
        wino_gemm(U1, frames, H, W, bufs)
        if fused:
            wino_out_in(frames, H, W, C, bufs, pad_mode, colscale=s1, bias=b1, relu=True)
        else:
            wino_in(wino_out(frames, H, W, C, bufs, s1, b1, relu=True, out=tbuf), frames, H, W, bufs, pad_mode)
        wino_gemm(U2, frames, H, W, bufs)
        if last:
            wino_out(frames, H, W, C, bufs, s2, b2, residual=y, act_after=last_relu, out=y)
        elif fused:
            wino_out_in(frames, H, W, C, bufs, pad_mode, colscale=s2, bias=b2, residual=y, out=y)
        else:
            wino_in(wino_out(frames, H, W, C, bufs, s2, b2, residual=y, out=y), frames, H, W, bufs, pad_mode)
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
