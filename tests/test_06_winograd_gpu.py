"""GPU parity of the Winograd F(4x4, 3x3) path of the frozen encoder (csrc/winograd.hip, ops.wino_conv3x3; ResNetAutoEncoder.py:127-157 under
train_NAR.py:54-56): the strided-batch form of the P16 nt GEMM, the transforms with every padding mode and epilogue option against fp64
torch on the CPU, and the nine-block encoder tail against the direct implicit-GEMM path of the same build.  Tolerance: 1e-4 rel-L2 per
convolution (fp32 transforms amplify the operand split's 2^-17 by the interpolation matrices' cancellation: measured 2 - 4e-5)."""
import pytest
import torch
import torch.nn.functional as F

from helpers import rel
from oracle import fill

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import vptr_amd.ops as ops
    return ops


def rn(shape, seed, scale=1.0):
    return fill.rand_normal(shape, seed, scale)


@pytest.mark.parametrize("members,M,N,K", [(36, 1280, 528, 528), (5, 200, 96, 80), (36, 48, 64, 64)])
def test_gemm_p16_strided_batch(ops, dev, members, M, N, K):
    """vptr_gemm_desc.batch_stride_*: member i = A + i * stride_a, B + i * stride_b -> D + i * stride_d, rows beyond M of a padded member untouched"""
    Mpad = (M + 127) // 128 * 128
    A = torch.zeros(members, Mpad, K)
    A[:, :M] = rn((members, M, K), 1)
    Bm = rn((members, N, K), 2, K ** -0.5)
    Ap, Bp = ops.to_p16(A.reshape(-1, K).to(dev)), ops.to_p16(Bm.reshape(-1, K).to(dev))
    D = torch.full((members, Mpad, N), 7.0, device=dev)
    ops.gemm_raw(Ap, Bp, D, M, N, K, 5, 3, lda=K, ldb=K, ldd=N, precision=3, batch_strided=(members, Mpad * K, N * K, Mpad * N))
    ref = torch.einsum("bmk,bnk->bmn", A[:, :M].double(), Bm.double())
    assert rel(D[:, :M].cpu(), ref) < 3e-5
    assert bool((D[:, M:] == 7.0).all())


def _ref_conv(x, w, frames, H, W, pad_mode, scale, shift, relu, residual, act_after):
    C = x.shape[1]
    xi = x.double().view(frames, H, W, C).permute(0, 3, 1, 2)
    if pad_mode == "zero":
        y = F.conv2d(xi, w.double(), padding=1)
    else:
        y = F.conv2d(F.pad(xi, (1, 1, 1, 1), mode=pad_mode), w.double())
    y = y.permute(0, 2, 3, 1).reshape(frames * H * W, -1)
    if scale is not None:
        y = y * scale.double() + shift.double()
    if relu:
        y = torch.relu(y)
    if residual is not None:
        y = y + residual.double()
    if act_after:
        y = torch.relu(y)
    return y


@pytest.mark.parametrize("frames,H,W,C", [(3, 8, 8, 64), (2, 4, 12, 48), (5, 16, 16, 32)])
@pytest.mark.parametrize("pad_mode", ["reflect", "zero", "replicate"])
def test_wino_conv3x3_matches_fp64(ops, dev, frames, H, W, C, pad_mode):
    x = rn((frames * H * W, C), 1)
    w = rn((C, C, 3, 3), 2, (9 * C) ** -0.5)
    scale, shift = rn((C,), 3).abs() + 0.5, rn((C,), 4, 0.3)
    res = rn((frames * H * W, C), 5)
    wd = w.to(dev)
    with ops.frozen_weights(True):
        assert ops.wino_ok(H, W, C, C)
        U = ops.wino_filter(wd)
        bufs = ops.wino_buffers(frames, H, W, C, dev, type("Holder", (), {})())
        y0 = ops.wino_conv3x3(x.to(dev), U, frames, H, W, bufs, pad_mode)
        y1 = ops.wino_conv3x3(x.to(dev), U, frames, H, W, bufs, pad_mode, colscale=scale.to(dev), bias=shift.to(dev), relu=True)
        r = res.to(dev).clone()
        y2 = ops.wino_conv3x3(x.to(dev), U, frames, H, W, bufs, pad_mode, colscale=scale.to(dev), bias=shift.to(dev), residual=r, act_after=True, out=r)
    assert y2.data_ptr() == r.data_ptr()
    assert rel(y0.cpu(), _ref_conv(x, w, frames, H, W, pad_mode, None, None, False, None, False)) < 1e-4
    assert rel(y1.cpu(), _ref_conv(x, w, frames, H, W, pad_mode, scale, shift, True, None, False)) < 1e-4
    assert rel(y2.cpu(), _ref_conv(x, w, frames, H, W, pad_mode, scale, shift, False, res, True)) < 1e-4


def test_wino_filter_cache_follows_the_weight(ops, dev):
    w = rn((32, 32, 3, 3), 7).to(dev)
    with ops.frozen_weights(True):
        U0 = ops.wino_filter(w)
        assert ops.wino_filter(w) is U0
        w.mul_(2.0)
        U1 = ops.wino_filter(w)
    assert U1 is not U0
    assert rel(ops.p16_decode(U1.reshape(-1, 32)).cpu(), 2.0 * ops.p16_decode(U0.reshape(-1, 32)).cpu()) < 1e-6


def test_encoder_winograd_matches_direct_path(dev):
    """VPTREnc (frozen, eval) on 64 x 64 clips: the Winograd ResnetBlocks against the direct plane-operand implicit GEMMs of the same build and
    against fp64 torch running the module tree the reference's way"""
    import vptr_amd.model as M
    import vptr_amd.ops as ops
    torch.manual_seed(5)
    enc = M.VPTREnc(1, 528, 3, "reflect").to(dev).eval()
    for mod in enc.modules():   # non-trivial eval statistics
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.1)
    x = torch.randn(2, 3, 1, 64, 64, device=dev).clamp(-1, 1)
    with torch.no_grad():
        assert ops.config.winograd
        a = enc(x)
        ops.config.winograd = False
        try:
            b = enc(x)
        finally:
            ops.config.winograd = True
    assert rel(a, b) < 2e-4
    assert float((a - b).abs().max()) < 2e-3 * float(b.abs().max())


@pytest.mark.parametrize("frames,H,W,C", [(5, 8, 8, 80), (3, 16, 16, 48), (7, 4, 8, 64), (9, 4, 4, 32), (2, 12, 8, 32)])
@pytest.mark.parametrize("pad_mode", ["reflect", "zero"])
def test_wino_resnet_blocks_fused_vs_fp64(ops, dev, frames, H, W, C, pad_mode):
    """two ResnetBlocks in the Winograd domain (ops.wino_resnet_blocks): the LDS-fused output -> input transform (vptr_wino_out_in; partial last
    channel slab, several units per workgroup, 1 - 16 tiles per map; 12 x 8 = 6 tiles falls back to the two-launch form) == the unfused chain
    == fp64 torch"""
    x = rn((frames * H * W, C), 1)
    ws = [rn((C, C, 3, 3), 10 + i, (9 * C) ** -0.5) for i in range(4)]
    scs = [rn((C,), 20 + i).abs() * 0.5 + 0.5 for i in range(4)]
    shs = [rn((C,), 30 + i, 0.2) for i in range(4)]
    ref = x.double()
    for b in range(2):
        t = _ref_conv(ref, ws[2 * b], frames, H, W, pad_mode, scs[2 * b], shs[2 * b], True, None, False)
        ref = _ref_conv(t, ws[2 * b + 1], frames, H, W, pad_mode, scs[2 * b + 1], shs[2 * b + 1], False, ref, b == 1)
    wd = [w.to(dev) for w in ws]
    outs = []
    for fuse in (True, False):
        ops.config.winograd_fuse = fuse
        try:
            with ops.frozen_weights(True):
                assert ops.wino_fused_ok(H, W) == (fuse and 16 % ((H // 4) * (W // 4)) == 0)
                blocks = [(ops.wino_filter(wd[2 * b]), scs[2 * b].to(dev), shs[2 * b].to(dev), ops.wino_filter(wd[2 * b + 1]), scs[2 * b + 1].to(dev),
                           shs[2 * b + 1].to(dev)) for b in range(2)]
                bufs = ops.wino_buffers(frames, H, W, C, dev, type("Holder", (), {})())
                y = x.to(dev).clone()
                out = ops.wino_resnet_blocks(y, blocks, frames, H, W, bufs, pad_mode, last_relu=True)
                assert out.data_ptr() == y.data_ptr()
                outs.append(out.cpu())
        finally:
            ops.config.winograd_fuse = True
    assert rel(outs[0], ref) < 1.5e-4 and rel(outs[1], ref) < 1.5e-4
    assert rel(outs[0], outs[1]) < 1e-6     # the same arithmetic either way
