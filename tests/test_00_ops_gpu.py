"""GPU parity of every C-ABI kernel against plain torch fp64/fp32 CPU math on the same seeded inputs.

Tolerances (rel-L2): split-bf16 GEMM (precision 3) 3e-5; single-pass bf16 (precision 1) 1.5e-2; fp32 vector kernels 2e-5.
"""

import pytest
import torch
import torch.nn.functional as F

from helpers import rel
from oracle import fill
from oracle import vptr_oracle as O

pytestmark = pytest.mark.gpu

TOL3, TOL1, TOLV = 3e-5, 1.5e-2, 2e-5


@pytest.fixture(scope="module")
def ops():
    import vptr_amd.ops as ops
    return ops


def rn(shape, seed, scale=1.0):
    return fill.rand_normal(shape, seed, scale)


# ---------------------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("prec,tol", [(3, TOL3), (1, TOL1)])
@pytest.mark.parametrize("M,N,K", [(300, 528, 528), (257, 100, 72), (128, 352, 2112), (64, 48, 48), (1000, 1584, 528),
                                   (200, 50, 48), (5000, 530, 64), (20000, 530, 64)])   # N % 4 != 0: fragment-layout epilogues (both loops)
def test_gemm_nt_epilogue(ops, dev, M, N, K, prec, tol):
    x, W, b, r = rn((M, K), 1), rn((N, K), 2, K ** -0.5), rn((N,), 3), rn((M, N), 4)
    ref = F.gelu((x.double() @ W.double().t() + b.double()) * 0.5) + r.double()
    y = torch.empty((M, N), device=dev)
    pre = torch.empty((M, N), device=dev)
    ops.gemm_raw(x.to(dev), W.to(dev), y, M, N, K, 0, 0, bias=b.to(dev), alpha=0.5, act=ops.ACT_GELU, Dpre=pre,
                 residual=r.to(dev), precision=prec)
    assert rel(y, ref) < tol
    assert rel(pre, (x.double() @ W.double().t() + b.double()) * 0.5) < tol
    # asymmetric check against a transposed result (would catch a swapped C/D fragment layout)
    if M != N:
        assert y.shape == (M, N)


@pytest.mark.parametrize("prec,tol", [(3, TOL3), (1, TOL1)])
def test_gemm_dgrad_wgrad_modes(ops, dev, prec, tol):
    M, N, K = 777, 528, 2112
    g, W, x = rn((M, N), 5), rn((N, K), 6, N ** -0.5), rn((M, K), 7)
    dx = torch.empty((M, K), device=dev)
    ops.gemm_raw(g.to(dev), W.to(dev), dx, M, K, N, 0, 1, precision=prec)          # g[M,N] @ W[N,K]
    assert rel(dx, g.double() @ W.double()) < tol
    dW = torch.zeros((N, K), device=dev)
    ops.gemm_raw(g.to(dev), x.to(dev), dW, N, K, M, 1, 1, atomic=True, split_k=3, precision=prec)  # g^T @ x
    assert rel(dW, g.double().t() @ x.double()) < tol
    dW2 = torch.zeros((N, K), device=dev)
    ops.gemm_raw(g.to(dev), x.to(dev), dW2, N, K, M, 1, 1, atomic=True, split_k=1, precision=prec)
    assert rel(dW2, g.double().t() @ x.double()) < tol
    # A k-strided x B k-contiguous
    Bk = rn((260, M + 3), 8)[:, :M].contiguous()  # [n, k] with k = M contiguous; M % 4 != 0 would be rejected
    if M % 4 == 0:
        out = torch.empty((N, 260), device=dev)
        ops.gemm_raw(g.to(dev), Bk.to(dev), out, N, 260, M, 1, 0, precision=prec)
        assert rel(out, g.double().t() @ Bk.double().t()) < tol


def test_gemm_grouped_wgrad_with_bias_rowsum(ops, dev):
    """vptr_gemm_grouped (deferred weight gradients of a backward pass, no split-K) + the a_rowsum bias-gradient fusion, and
    the same a_rowsum through the split-K single launch; accumulation into pre-filled destinations."""
    probs = [(1000, 528, 528), (640, 176, 2112), (512, 1056, 528), (260, 48, 48)]  # tokens, out features, in features
    keep, refs = [], []
    for i, (M, N, K) in enumerate(probs):
        g, x = rn((M, N), 10 + i), rn((M, K), 20 + i)
        dW0, db0 = rn((N, K), 30 + i), rn((N,), 40 + i)
        gd, xd, dW, db = g.to(dev), x.to(dev), dW0.to(dev), db0.to(dev)
        keep.append((gd, xd, dW, db))
        refs.append((dW0.double() + g.double().t() @ x.double(), db0.double() + g.double().sum(0)))
        ops.defer_wgrad(gd, xd, dW, N, K, M, db=db)
    ops.flush_wgrads()
    for (gd, xd, dW, db), (rW, rb) in zip(keep, refs):
        assert rel(dW, rW) < TOL3
        assert rel(db, rb) < TOL3
    # single launch, split-K + atomics, with the row sums
    M, N, K = 2048, 528, 352
    g, x = rn((M, N), 50), rn((M, K), 51)
    dW, db = torch.zeros((N, K), device=dev), torch.zeros((N,), device=dev)
    gd, xd = g.to(dev), x.to(dev)
    d = ops.gemm_raw(gd, xd, dW, N, K, M, 1, 1, atomic=True, split_k=4, a_rowsum=db)
    assert rel(dW, g.double().t() @ x.double()) < TOL3
    assert rel(db, g.double().sum(0)) < TOL3


@pytest.mark.parametrize("M,N,K", [(300, 528, 528), (5000, 528, 528), (130, 100, 72)])
def test_gemm_batched_members(ops, dev, M, N, K):
    """vptr_gemm_desc.batch: three same-shaped problems (own A, B, D, bias, alpha) in one launch == three launches."""
    xs = [rn((M, K), 200 + i).to(dev) for i in range(3)]
    Ws = [rn((N, K), 210 + i, K ** -0.5).to(dev) for i in range(3)]
    bs = [rn((N,), 220 + i).to(dev) for i in range(3)]
    al = [0.25, 1.0, 2.0]
    ys = [torch.empty((M, N), device=dev) for _ in range(3)]
    ops.gemm_raw(xs[0], Ws[0], ys[0], M, N, K, 0, 0, bias=bs[0], alpha=al[0],
                 batch_extra=[(xs[1], Ws[1], ys[1], bs[1], al[1]), (xs[2], Ws[2], ys[2], None, al[2])])
    for i in range(3):
        ref = (xs[i].double().cpu() @ Ws[i].double().cpu().t() + (bs[i].double().cpu() if i < 2 else 0.0)) * al[i]
        assert rel(ys[i], ref) < TOL3
    # two members, k-strided B (the input-gradient orientation)
    gs = [rn((M, N), 230 + i).to(dev) for i in range(2)]
    ds = [torch.empty((M, K), device=dev) for _ in range(2)]
    ops.gemm_raw(gs[0], Ws[0], ds[0], M, K, N, 0, 1, batch_extra=[(gs[1], Ws[1], ds[1], None, 1.0)])
    for i in range(2):
        assert rel(ds[i], gs[i].double().cpu() @ Ws[i].double().cpu()) < TOL3


@pytest.mark.parametrize("M,N,K,nseg", [(300, 528, 528, 3), (5000, 528, 528, 2), (130, 72, 100, 3), (1000, 528, 48, 2)])
def test_gemm_k_segments(ops, dev, M, N, K, nseg):
    """vptr_gemm_desc.ksegs: D = sum_s A_s[M,K] . B_s[K,N] (+ epilogue), every segment with its own K tail."""
    gs = [rn((M, K), 240 + i).to(dev) for i in range(nseg)]
    Ws = [rn((K, N), 250 + i, K ** -0.5).to(dev) for i in range(nseg)]      # k-strided B: [K, N]
    r = rn((M, N), 260).to(dev)
    y = torch.empty((M, N), device=dev)
    ops.gemm_raw(gs[0], Ws[0], y, M, N, K, 0, 1, alpha=0.5, residual=r, kseg_extra=[(gs[i], Ws[i]) for i in range(1, nseg)])
    ref = sum(g.double().cpu() @ W.double().cpu() for g, W in zip(gs, Ws)) * 0.5 + r.double().cpu()
    assert rel(y, ref) < TOL3
    # k-contiguous B as well
    Wt = [W.t().contiguous() for W in Ws]
    ops.gemm_raw(gs[0], Wt[0], y, M, N, K, 0, 0, kseg_extra=[(gs[i], Wt[i]) for i in range(1, nseg)])
    assert rel(y, (ref - r.double().cpu()) * 2.0) < TOL3


def test_gemm_epilogue_variants(ops, dev):
    M, N, K = 200, 176, 64
    x, W = rn((M, K), 9), rn((N, K), 10, K ** -0.5)
    cs, b, rs = rn((N,), 11).abs() + 0.5, rn((N,), 12), rn((10,), 13).abs()
    r = rn((M, N), 14)
    acc = x.double() @ W.double().t()
    rows = torch.arange(M)
    ref = torch.relu(torch.relu(acc * cs.double() + b.double()) * rs.double()[(rows // 4) % 10][:, None] + r.double())
    y = torch.empty((M, N), device=dev)
    ops.gemm_raw(x.to(dev), W.to(dev), y, M, N, K, 0, 0, colscale=cs.to(dev), bias=b.to(dev), act=ops.ACT_RELU,
                 rowscale=rs.to(dev), rs_div=4, rs_mod=10, residual=r.to(dev), act_after=True, precision=3)
    assert rel(y, ref) < TOL3


def test_linear_autograd(ops, dev):
    M, K, N = 384, 96, 192
    x, W, b, r = rn((M, K), 20), rn((N, K), 21, K ** -0.5), rn((N,), 22), rn((M, N), 23)
    gout = rn((M, N), 24)
    xs = [t.clone().requires_grad_(True) for t in (x.double(), W.double(), b.double(), r.double())]
    ref = F.gelu((xs[0] @ xs[1].t() + xs[2]) * 0.7) + xs[3]
    ref.backward(gout.double())
    ds = [t.to(dev).requires_grad_(True) for t in (x, W, b, r)]
    y = ops.linear(ds[0], ds[1], ds[2], residual=ds[3], alpha=0.7, act=ops.ACT_GELU)
    y.backward(gout.to(dev))
    assert rel(y, ref) < TOL3
    for a, c in zip(ds, xs):
        assert rel(a.grad, c.grad) < 5e-5


def test_linear_dropout_mask_consistency(ops, dev):
    M, K = 512, 64
    ops.manual_seed(dev, 1234)
    ops.new_seed_scope(dev)
    x = torch.ones((M, K), device=dev, requires_grad=True)
    W = torch.eye(K, device=dev)
    y = ops.linear(x, W, None, dropout_p=0.25, site=7)
    y.backward(torch.ones_like(y))
    yv = y.detach().cpu()
    kept = (yv != 0).float().mean().item()
    assert abs(kept - 0.75) < 0.02
    assert torch.allclose(yv[yv != 0], torch.full_like(yv[yv != 0], 1 / 0.75), rtol=1e-4)
    assert rel(x.grad, yv) < 1e-4      # backward re-creates the same mask
    ops.new_seed_scope(dev)
    y2 = ops.linear(x.detach(), W, None, dropout_p=0.25, site=7)
    assert (y2.cpu() != yv).float().mean().item() > 0.2  # fresh scope -> fresh mask


# ----------------------------------------------------------------------------------------------------------- conv GEMM
@pytest.mark.parametrize("pad_mode", ["zero", "reflect", "replicate"])
@pytest.mark.parametrize("stride", [1, 2])
def test_conv_gather(ops, dev, pad_mode, stride):
    B, Cin, Cout, H, W = 3, 16, 40, 12, 10
    x, w = rn((B, Cin, H, W), 30), rn((Cout, Cin, 3, 3), 31, 0.1)
    if pad_mode == "zero":
        ref = F.conv2d(x.double(), w.double(), stride=stride, padding=1)
    else:
        ref = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode=pad_mode), w.double(), stride=stride)
    OH, OW = ref.shape[2:]
    xt = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(dev)
    y = ops.conv_nhwc(xt, ops.conv_weight_as_gemm_b(w, False).to(dev), B, H, W, Cin, OH, OW, 3, 3, stride, 1, pad_mode, False, Cout)
    assert rel(y.reshape(B, OH, OW, Cout).permute(0, 3, 1, 2), ref) < TOL3


@pytest.mark.parametrize("pad_mode,stride,Cin,Cout", [("reflect", 1, 48, 528), ("zero", 1, 40, 100), ("zero", 2, 64, 176), ("replicate", 1, 528, 528)])
def test_conv_planes_matches_register_staged(ops, dev, pad_mode, stride, Cin, Cout):
    """a_mode = VPTR_A_CONV_PLANES (bf16 hi / lo plane operands staged by global_load_lds) == the fp32-staged implicit GEMM"""
    Bf, H, W = 5, 8, 8
    OH, OW = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    x = rn((Bf * H * W, Cin), 400).to(dev)
    wt = rn((Cout, Cin, 3, 3), 401, (9 * Cin) ** -0.5).to(dev)
    cs, bs = (rn((Cout,), 402).abs() + 0.5).to(dev), rn((Cout,), 403).to(dev)
    res = rn((Bf * OH * OW, Cout), 404).to(dev)
    ref = ops.conv_nhwc(x, ops.conv_weight_as_gemm_b(wt, False), Bf, H, W, Cin, OH, OW, 3, 3, stride, 1, pad_mode, False, Cout,
                        colscale=cs, bias=bs, act=ops.ACT_RELU)
    ref2 = ops.conv_nhwc(x, ops.conv_weight_as_gemm_b(wt, False), Bf, H, W, Cin, OH, OW, 3, 3, stride, 1, pad_mode, False, Cout,
                         colscale=cs, bias=bs, residual=res, act_after=True)
    with ops.frozen_weights(True):
        xp, wp = ops.split_planes(x), ops.conv_weight_as_planes(wt)
        assert xp.shape == (Bf * H * W + 1, (Cin + 31) // 32, 64) and float(xp[-1].abs().max()) == 0.0
        rec = xp[:-1, :, :32].float() + xp[:-1, :, 32:].float()                      # hi + lo reproduces x to 2^-17
        assert rel(rec.reshape(Bf * H * W, -1)[:, :Cin], x) < 1e-5
        y = ops.conv_nhwc_planes(xp, wp, Bf, H, W, Cin, OH, OW, 3, 3, stride, 1, pad_mode, Cout, colscale=cs, bias=bs, act=ops.ACT_RELU)
        y2 = ops.conv_nhwc_planes(xp, wp, Bf, H, W, Cin, OH, OW, 3, 3, stride, 1, pad_mode, Cout, colscale=cs, bias=bs, residual=res,
                                  act_after=True)
    assert rel(y, ref) < TOL3 and rel(y2, ref2) < TOL3
    if Cout % 4 == 0:  # plane-form output written by the epilogue == a split pass over the fp32 output
        with ops.frozen_weights(True):
            po = torch.zeros((Bf * OH * OW + 1, (Cout + 31) // 32, 64), device=dev, dtype=torch.bfloat16)
            y3 = ops.conv_nhwc_planes(xp, wp, Bf, H, W, Cin, OH, OW, 3, 3, stride, 1, pad_mode, Cout, colscale=cs, bias=bs, residual=res,
                                      act_after=True, planes_out=po)
            assert torch.equal(y3, y2) and torch.equal(po, ops.split_planes(y2))
            po2 = torch.zeros_like(po)
            none = ops.conv_nhwc_planes(xp, wp, Bf, H, W, Cin, OH, OW, 3, 3, stride, 1, pad_mode, Cout, colscale=cs, bias=bs, residual=res,
                                        act_after=True, planes_out=po2, fp32_out=False)
            assert none is None and torch.equal(po2, po)


@pytest.mark.parametrize("subpixel", [True, False])
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 24, 20, 6, 5), (3, 528, 256, 8, 8), (2, 128, 64, 32, 32)])
def test_conv_transposed_gather(ops, dev, monkeypatch, subpixel, B, Cin, Cout, H, W):
    """ConvTranspose2d(3x3, s2, p1, op1) + folded BN + ReLU: as four output-parity classes through the GEMM's output row map (default)
    and as the 9-tap gather form"""
    monkeypatch.setattr(ops.config, "subpixel_convt", subpixel)
    x, w = rn((B, Cin, H, W), 32), rn((Cin, Cout, 3, 3), 33, 0.1)
    sc, sh = rn((Cout,), 34).abs() + 0.5, rn((Cout,), 35)
    ref = torch.relu(F.conv_transpose2d(x.double(), w.double(), stride=2, padding=1, output_padding=1) * sc.double()[None, :, None, None]
                     + sh.double()[None, :, None, None])
    xt = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(dev)
    Bm = ops.conv_weight_as_gemm_b(w.to(dev), True)
    assert isinstance(Bm, ops.SubpixelWeights) == subpixel
    y = ops.conv_nhwc(xt, Bm, B, H, W, Cin, 2 * H, 2 * W, 3, 3, 2, 1, "zero", True, Cout, colscale=sc.to(dev), bias=sh.to(dev), act=ops.ACT_RELU)
    assert rel(y.reshape(B, 2 * H, 2 * W, Cout).permute(0, 3, 1, 2), ref) < TOL3


# ----------------------------------------------------------------------------------------------------------- layernorm
@pytest.mark.parametrize("nb,C,HW,T", [(2, 48, 16, 3), (32, 528, 48, 3), (40, 100, 40, 3)])   # small, model-sized and C % 16 != 0 rows
def test_layernorm_fwd_bwd(ops, dev, nb, C, HW, T):
    rows = nb * T * HW
    x, g, b, tab = rn((rows, C), 40), rn((C,), 41).abs() + 0.5, rn((C,), 42), rn((T, C), 43)
    go, go2 = rn((rows, C), 44), rn((rows, C), 45)
    xs = [t.double().clone().requires_grad_(True) for t in (x, g, b, tab)]
    y = F.layer_norm(xs[0], (C,), xs[1], xs[2], 1e-5)
    t_idx = (torch.arange(rows) // HW) % T
    y2 = y + xs[3][t_idx]
    (y * go.double()).sum().backward(retain_graph=True)
    (y2 * go2.double()).sum().backward()
    ds = [t.to(dev).requires_grad_(True) for t in (x, g, b, tab)]
    o, o2 = ops.layernorm(ds[0], ds[1], ds[2], tab=ds[3], tab_div=HW, tab_mod=T)
    ((o * go.to(dev)).sum() + (o2 * go2.to(dev)).sum()).backward()
    assert rel(o, y) < TOLV and rel(o2, y2) < TOLV
    for a, c in zip(ds, xs):
        assert rel(a.grad, c.grad) < 5e-5
    # single-output form
    d0 = x.to(dev).requires_grad_(True)
    o = ops.layernorm(d0, ds[1].detach(), ds[2].detach())
    assert rel(o, y) < TOLV
    # pass-through output: the residual gradient is added inside the backward kernel (dx_add); C = 48 (fused kernel) and 1028 (two-kernel path)
    for Cc in (48, 1028):
        xc, gc, bc = rn((rows, Cc), 46), rn((Cc,), 47).abs() + 0.5, rn((Cc,), 48)
        g1, g2, g3 = rn((rows, Cc), 49), rn((rows, Cc), 50), rn((rows, Cc), 51)
        tabc = rn((T, Cc), 52)
        xr_ = xc.double().clone().requires_grad_(True)
        yr = F.layer_norm(xr_, (Cc,), gc.double(), bc.double(), 1e-5)
        ((yr * g1.double()).sum() + ((yr + tabc.double()[t_idx]) * g2.double()).sum() + (xr_ * g3.double()).sum()).backward()
        xd = xc.to(dev).requires_grad_(True)
        o, o2, xp = ops.layernorm(xd, gc.to(dev), bc.to(dev), tab=tabc.to(dev), tab_div=HW, tab_mod=T, passthrough=True)
        assert torch.equal(xp, xd)
        ((o * g1.to(dev)).sum() + (o2 * g2.to(dev)).sum() + (xp * g3.to(dev)).sum()).backward()
        assert rel(xd.grad, xr_.grad) < 5e-5
        xd2 = xc.to(dev).requires_grad_(True)      # only the pass-through and the plain output consumed
        o, xp = ops.layernorm(xd2, gc.to(dev), bc.to(dev), passthrough=True)
        ((o * g1.to(dev)).sum() + (xp * g3.to(dev)).sum()).backward()
        xr2 = xc.double().clone().requires_grad_(True)
        ((F.layer_norm(xr2, (Cc,), gc.double(), bc.double(), 1e-5) * g1.double()).sum() + (xr2 * g3.double()).sum()).backward()
        assert rel(xd2.grad, xr2.grad) < 5e-5


def test_rowtab_colsum(ops, dev):
    rows, C = 7 * 5 * 4, 32
    x, tab = rn((rows, C), 46), rn((5, C), 47)
    xd, td = x.to(dev).requires_grad_(True), tab.to(dev).requires_grad_(True)
    y = ops.add_rowtab(xd, td, 4, 5)
    idx = (torch.arange(rows) // 4) % 5
    assert rel(y, x + tab[idx]) < 1e-6
    y.backward(torch.ones_like(y) * 2)
    ref = torch.zeros(5, C).index_add_(0, idx, torch.full((rows, C), 2.0))
    assert rel(td.grad, ref) < 1e-6


# ----------------------------------------------------------------------------------------------------------- attention
TOLA = 5e-5   # attention cores: split-bf16 MFMA products (forward; gradients 1e-4)


@pytest.fixture(params=["default", "attn16", "attn16fwd", "mfma", "vector"])
def attn_mode(request):
    """every attention geometry through all kernel families: the default (problems of at most 16 tokens on the MFMA kernels of
    attn16.hip -- forward without LDS, backward of the second generation when C % 4 == 0; larger ones on the LDS-staged MFMA kernels of
    attn_mfma.hip), VPTR_ATTN16=2 (attn16.hip forward + its first-generation backward), VPTR_ATTN16=4 (attn16.hip forward, fp32 vector
    backward), VPTR_ATTN_MFMA=2 (attn_mfma.hip wherever it covers the geometry) and VPTR_ATTN_MFMA=0 (the fp32 vector kernels of
    attn.hip everywhere)"""
    import os
    old = os.environ.get("VPTR_ATTN_MFMA")
    old16 = os.environ.get("VPTR_ATTN16")
    os.environ.pop("VPTR_ATTN16", None)
    if request.param in ("default", "attn16", "attn16fwd"):
        os.environ.pop("VPTR_ATTN_MFMA", None)
        if request.param != "default":
            os.environ["VPTR_ATTN16"] = "2" if request.param == "attn16" else "4"
    else:
        os.environ["VPTR_ATTN_MFMA"] = "2" if request.param == "mfma" else "0"
    yield request.param
    if old16 is None:
        os.environ.pop("VPTR_ATTN16", None)
    else:
        os.environ["VPTR_ATTN16"] = old16
    if old is None:
        os.environ.pop("VPTR_ATTN_MFMA", None)
    else:
        os.environ["VPTR_ATTN_MFMA"] = old


@pytest.mark.parametrize("ws,H,W,C", [(4, 8, 8, 48), (8, 16, 8, 48), (4, 8, 8, 528), (8, 16, 16, 528), (2, 4, 6, 64)])
def test_window_attention(ops, dev, attn_mode, ws, H, W, C):
    B, nh = 3, 8
    L = ws * ws
    q, k, v = rn((B * H * W, C), 50, 0.5), rn((B * H * W, C), 51, 0.5), rn((B * H * W, C), 52)
    table = rn(((2 * ws - 1) ** 2, nh), 53, 0.5)
    idx = O.rpe_index(ws)
    go = rn((B * H * W, C), 54)
    ins = [t.double().clone().requires_grad_(True) for t in (q, k, v, table)]

    def part(t):
        return O.win_partition(t.reshape(B, H, W, C), ws)
    bias = ins[3][idx.reshape(-1)].reshape(L, L, nh).permute(2, 0, 1)
    o = O._attend(O._heads(part(ins[0]), nh), O._heads(part(ins[1]), nh), O._heads(part(ins[2]), nh), bias)
    o = O.win_reverse(o, B, H, W, ws).reshape(B * H * W, C)
    (o * go.double()).sum().backward()
    ds = [t.to(dev).requires_grad_(True) for t in (q, k, v, table)]
    od = ops.window_attention(ds[0], ds[1], ds[2], ds[3], idx.to(dev), B, H, W, nh, ws)
    (od * go.to(dev)).sum().backward()
    assert rel(od, o) < TOLA
    for a, c in zip(ds, ins):
        assert rel(a.grad, c.grad) < 1e-4
    # without a bias table (rpe=False path)
    od2 = ops.window_attention(ds[0].detach(), ds[1].detach(), ds[2].detach(), None, None, B, H, W, nh, ws)
    o2 = O._attend(O._heads(part(q.double()), nh), O._heads(part(k.double()), nh), O._heads(part(v.double()), nh), None)
    assert rel(od2, O.win_reverse(o2, B, H, W, ws).reshape(B * H * W, C)) < TOLA


@pytest.mark.parametrize("Tq,Tk,causal,N,HW,C", [(5, 5, False, 2, 6, 48), (5, 5, True, 2, 6, 48), (3, 7, False, 2, 6, 48), (29, 29, True, 2, 6, 48),
                                                 # >= 256 pixel problems: the variant with 4 pixels per wave + register prefetch
                                                 (5, 5, True, 3, 90, 48), (3, 7, False, 3, 90, 48), (10, 10, False, 2, 129, 48),
                                                 # head dim 66 (the model's), the long sequences of BASELINE configs 4 / 5
                                                 (10, 10, False, 2, 64, 528), (29, 29, True, 1, 16, 528), (40, 10, False, 1, 20, 528),
                                                 (50, 50, False, 1, 9, 528), (64, 33, False, 1, 5, 128)])
def test_temporal_attention(ops, dev, attn_mode, Tq, Tk, causal, N, HW, C):
    nh = 8
    q, k, v = rn((N * Tq * HW, C), 60, 0.5), rn((N * Tk * HW, C), 61, 0.5), rn((N * Tk * HW, C), 62)
    go = rn((N * Tq * HW, C), 63)
    ins = [t.double().clone().requires_grad_(True) for t in (q, k, v)]

    def seq(t, T):  # (n,t,p) rows -> (n*p, T, C)
        return t.reshape(N, T, HW, C).permute(0, 2, 1, 3).reshape(N * HW, T, C)
    o = O._attend(O._heads(seq(ins[0], Tq), nh), O._heads(seq(ins[1], Tk), nh), O._heads(seq(ins[2], Tk), nh), None, causal)
    o = o.reshape(N, HW, Tq, C).permute(0, 2, 1, 3).reshape(N * Tq * HW, C)
    (o * go.double()).sum().backward()
    ds = [t.to(dev).requires_grad_(True) for t in (q, k, v)]
    od = ops.temporal_attention(ds[0], ds[1], ds[2], N, Tq, Tk, HW, nh, causal)
    (od * go.to(dev)).sum().backward()
    assert rel(od, o) < TOLA
    for a, c in zip(ds, ins):
        assert rel(a.grad, c.grad) < 1e-4


@pytest.mark.parametrize("nh,C", [(6, 48), (2, 132), (5, 80)])
def test_attention_partial_head_groups(ops, dev, nh, C):
    """head counts that are not a multiple of the 4 heads of an attn16 workgroup (idle waves must not store), head widths that put
    every other head on an unaligned quad (hd = 66, 16) or none (hd = 8), through the default kernels, both attention kinds"""
    B, H, W, ws = 2, 8, 8, 4
    L = ws * ws
    q, k, v = rn((B * H * W, C), 70, 0.5), rn((B * H * W, C), 71, 0.5), rn((B * H * W, C), 72)
    table, idx, go = rn(((2 * ws - 1) ** 2, nh), 73, 0.5), O.rpe_index(ws), rn((B * H * W, C), 74)
    ins = [t.double().clone().requires_grad_(True) for t in (q, k, v, table)]

    def part(t):
        return O.win_partition(t.reshape(B, H, W, C), ws)
    bias = ins[3][idx.reshape(-1)].reshape(L, L, nh).permute(2, 0, 1)
    o = O._attend(O._heads(part(ins[0]), nh), O._heads(part(ins[1]), nh), O._heads(part(ins[2]), nh), bias)
    o = O.win_reverse(o, B, H, W, ws).reshape(B * H * W, C)
    (o * go.double()).sum().backward()
    ds = [t.to(dev).requires_grad_(True) for t in (q, k, v, table)]
    od = ops.window_attention(ds[0], ds[1], ds[2], ds[3], idx.to(dev), B, H, W, nh, ws)
    (od * go.to(dev)).sum().backward()
    assert rel(od, o) < TOLA
    for a, c in zip(ds, ins):
        assert rel(a.grad, c.grad) < 1e-4
    N, T, HW = 2, 7, 9
    q, k, v, go = rn((N * T * HW, C), 75, 0.5), rn((N * T * HW, C), 76, 0.5), rn((N * T * HW, C), 77), rn((N * T * HW, C), 78)
    ins = [t.double().clone().requires_grad_(True) for t in (q, k, v)]

    def seq(t):
        return t.reshape(N, T, HW, C).permute(0, 2, 1, 3).reshape(N * HW, T, C)
    o = O._attend(O._heads(seq(ins[0]), nh), O._heads(seq(ins[1]), nh), O._heads(seq(ins[2]), nh), None, True)
    o = o.reshape(N, HW, T, C).permute(0, 2, 1, 3).reshape(N * T * HW, C)
    (o * go.double()).sum().backward()
    ds = [t.to(dev).requires_grad_(True) for t in (q, k, v)]
    od = ops.temporal_attention(ds[0], ds[1], ds[2], N, T, T, HW, nh, True)
    (od * go.to(dev)).sum().backward()
    assert rel(od, o) < TOLA
    for a, c in zip(ds, ins):
        assert rel(a.grad, c.grad) < 1e-4


def _proj_ref(ops, xq, xk, xv, Ws, bs, nh, attend):
    """composition of the separate nodes: three linears (alpha on q) + the attention core"""
    C = Ws[0].shape[0]
    q = ops.linear(xq, Ws[0], bs[0], alpha=float(C // nh) ** -0.5)
    k = ops.linear(xk, Ws[1], bs[1])
    v = ops.linear(xv, Ws[2], bs[2])
    return attend(q, k, v)


@pytest.mark.parametrize("variant", ["same_all", "same_qk", "merge_v"])
def test_proj_window_attention_matches_composition(ops, dev, variant):
    """fused q/k/v-projection + window-attention node (batched GEMM, dq_scale, K-segmented input gradient) == separate nodes"""
    B, H, W, C, nh, ws = 5, 8, 8, 48, 8, 4
    M = B * H * W
    idx = O.rpe_index(ws).to(dev)
    go = rn((M, C), 300).to(dev)
    tab = rn((H * W, C), 301, 0.3).to(dev)

    def run(fused):
        x = rn((M, C), 302).to(dev).requires_grad_(True)
        Ws = [rn((C, C), 303 + i, C ** -0.5).to(dev).requires_grad_(True) for i in range(3)]
        bs = [rn((C,), 306 + i).to(dev).requires_grad_(True) for i in range(3)]
        table = rn(((2 * ws - 1) ** 2, nh), 309, 0.5).to(dev).requires_grad_(True)
        xqk = x if variant == "same_all" else ops.add_rowtab(x, tab, 1, H * W)
        if fused:
            o = ops.proj_window_attention(xqk, x, Ws[0], bs[0], Ws[1], bs[1], Ws[2], bs[2], table, idx, B, H, W, nh, ws,
                                          merge_v_grad=(variant == "merge_v"))
        else:
            o = _proj_ref(ops, xqk, xqk, x, Ws, bs, nh, lambda q, k, v: ops.window_attention(q, k, v, table, idx, B, H, W, nh, ws))
        (o * go).sum().backward()
        return [o.detach(), x.grad, table.grad] + [w.grad for w in Ws] + [b.grad for b in bs]
    for a, c in zip(run(True), run(False)):
        assert rel(a, c.double().cpu()) < 2e-5


@pytest.mark.parametrize("variant,Tq,Tk", [("self_merge", 5, 5), ("self", 5, 5), ("cross", 5, 5), ("cross", 3, 7)])
def test_proj_temporal_attention_matches_composition(ops, dev, variant, Tq, Tk):
    N, HW, C, nh = 2, 40, 48, 8
    Mq, Mk = N * Tq * HW, N * Tk * HW
    go = rn((Mq, C), 320).to(dev)
    tab = rn((Tq, C), 321, 0.3).to(dev)

    def run(fused):
        xq = rn((Mq, C), 322).to(dev).requires_grad_(True)
        xk = rn((Mk, C), 323).to(dev).requires_grad_(True)
        xv = rn((Mk, C), 324).to(dev).requires_grad_(True)
        w = rn((3 * C, C), 325, C ** -0.5).to(dev).requires_grad_(True)      # packed in_proj of nn.MultiheadAttention
        b = rn((3 * C,), 326).to(dev).requires_grad_(True)
        Ws, bs = [w[:C], w[C:2 * C], w[2 * C:]], [b[:C], b[C:2 * C], b[2 * C:]]
        if variant == "cross":
            a_q, a_k, a_v = xq, xk, xv
        else:
            a_q = a_k = ops.add_rowtab(xq, tab, HW, Tq)
            a_v = xq
        if fused:
            o = ops.proj_temporal_attention(a_q, a_k, a_v, Ws[0], bs[0], Ws[1], bs[1], Ws[2], bs[2], N, Tq, Tk, HW, nh,
                                            merge_v_grad=(variant == "self_merge"))
        else:
            o = _proj_ref(ops, a_q, a_k, a_v, Ws, bs, nh, lambda q, k, v: ops.temporal_attention(q, k, v, N, Tq, Tk, HW, nh))
        (o * go).sum().backward()
        out = [o.detach(), xq.grad, w.grad, b.grad]
        if variant == "cross":
            out += [xk.grad, xv.grad]
        return out
    for a, c in zip(run(True), run(False)):
        assert rel(a, c.double().cpu()) < 2e-5


# ------------------------------------------------------------------------------------------------------ conv-FFN pieces
@pytest.mark.parametrize("mode", ["bn", "ln", "bn_eval"])
def test_norm_act(ops, dev, mode):
    frames, H, W, Fc = 6, 4, 4, 32
    HW, rows = H * W, frames * H * W
    x, go, res = rn((rows, Fc), 70, 2.0) + 0.3, rn((rows, Fc), 71), rn((rows, Fc), 72)
    rs = rn((3,), 73).abs() + 0.5
    rs_idx = (torch.arange(rows) // (2 * HW)) % 3
    xn = x.double().reshape(frames, H, W, Fc).permute(0, 3, 1, 2).clone().requires_grad_(True)
    if mode == "ln":
        w, b = rn((Fc, H, W), 74).abs() + 0.5, rn((Fc, H, W), 75)
        wd, bd = w.double().clone().requires_grad_(True), b.double().clone().requires_grad_(True)
        z = F.layer_norm(xn, (Fc, H, W), wd, bd, 1e-5)
    else:
        w, b = rn((Fc,), 74).abs() + 0.5, rn((Fc,), 75)
        rm, rv = rn((Fc,), 76, 0.1), rn((Fc,), 77).abs() + 0.5
        wd, bd = w.double().clone().requires_grad_(True), b.double().clone().requires_grad_(True)
        rmd, rvd = rm.double().clone(), rv.double().clone()
        z = F.batch_norm(xn, rmd, rvd, wd, bd, mode == "bn", 0.1, 1e-5)
    y = F.gelu(z).permute(0, 2, 3, 1).reshape(rows, Fc) * rs.double()[rs_idx][:, None] + res.double()
    (y * go.double()).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    resd = res.to(dev).requires_grad_(True)
    if mode == "ln":
        wp, bp = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        wcl = wp.reshape(Fc, HW).t().contiguous()
        bcl = bp.reshape(Fc, HW).t().contiguous()
        yd = ops.norm_act(xd, wcl, bcl, "ln", HW, True, rowscale=rs.to(dev), rs_div=2 * HW, rs_mod=3, residual=resd)
    else:
        wp, bp = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        rmg, rvg = rm.to(dev), rv.to(dev)
        yd = ops.norm_act(xd, wp, bp, "bn", HW, mode == "bn", rmg, rvg, rowscale=rs.to(dev), rs_div=2 * HW, rs_mod=3, residual=resd)
        if mode == "bn":
            assert rel(rmg, rmd) < 1e-5 and rel(rvg, rvd) < 1e-5
    (yd * go.to(dev)).sum().backward()
    assert rel(yd, y) < TOLV
    assert rel(xd.grad, xn.grad.permute(0, 2, 3, 1).reshape(rows, Fc)) < 1e-4
    assert rel(wp.grad, wd.grad) < 1e-4 and rel(bp.grad, bd.grad) < 1e-4
    assert rel(resd.grad, go) < 1e-6


@pytest.mark.parametrize("frames,H,W,Fc", [(5, 8, 6, 32), (32, 8, 8, 1024), (3, 5, 7, 16), (70, 4, 4, 64), (40, 4, 4, 2048), (24, 16, 16, 1024),
                                            (48, 8, 6, 1024)])   # 2nd, 5th, 6th: two-column forward kernel with DPP halo exchange (W / 2 = 4, 2, 8); 3rd: odd W; 7th: W / 2 = 3 (no DPP)
def test_dwconv(ops, dev, frames, H, W, Fc):
    x, w, b, go = rn((frames * H * W, Fc), 80), rn((Fc, 1, 3, 3), 81, 0.3), rn((Fc,), 82), rn((frames * H * W, Fc), 83)
    xn = x.double().reshape(frames, H, W, Fc).permute(0, 3, 1, 2).clone().requires_grad_(True)
    wd, bd = w.double().clone().requires_grad_(True), b.double().clone().requires_grad_(True)
    y = F.conv2d(xn, wd, bd, padding=1, groups=Fc).permute(0, 2, 3, 1).reshape(-1, Fc)
    (y * go.double()).sum().backward()
    xd, wp, bp = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    yd = ops.dwconv3x3(xd, wp, bp, frames, H, W)
    (yd * go.to(dev)).sum().backward()
    assert rel(yd, y) < TOLV
    assert rel(xd.grad, xn.grad.permute(0, 2, 3, 1).reshape(-1, Fc)) < 2e-5
    assert rel(wp.grad, wd.grad) < 5e-5 and rel(bp.grad, bd.grad) < 5e-5


def test_layout(ops, dev):
    B, C, H, W = 3, 50, 5, 7
    x, go = rn((B, C, H, W), 90), rn((B, C, H, W), 91)
    xd = x.to(dev).requires_grad_(True)
    t = ops.nchw_to_tokens(xd)
    assert torch.equal(t.cpu(), x.permute(0, 2, 3, 1).reshape(-1, C))
    back = ops.tokens_to_nchw(t, B, C, H, W, relu=True)
    assert torch.equal(back.detach().cpu(), torch.relu(x))
    back.backward(go.to(dev))
    assert torch.equal(xd.grad.cpu(), go * (x > 0))


# ------------------------------------------------------------------------------------------------------------ 7x7 convs
@pytest.mark.parametrize("cimg,geom", [(1, (2, 16, 24)), (3, (2, 16, 24)), (1, (3, 64, 64)), (3, (1, 64, 64)), (1, (2, 24, 64)), (1, (2, 8, 16))])
def test_conv7_ends(ops, dev, cimg, geom):
    """first / last 7x7 convolutions of the auto-encoder.  Single-channel images take the second-generation kernels (taps in
    registers, broadcast LDS windows; backward-data only for W = 64), 3-channel images and odd widths the first-generation ones"""
    from vptr_amd._lib import check, lib, ptr, stream
    B, H, W = geom
    x, w = rn((B, cimg, H, W), 100), rn((64, cimg, 7, 7), 101, 0.1)
    sc, sh = rn((64,), 102).abs() + 0.5, rn((64,), 103)
    ref = torch.relu(F.conv2d(F.pad(x.double(), (3, 3, 3, 3), mode="reflect"), w.double()) * sc.double()[None, :, None, None]
                     + sh.double()[None, :, None, None])
    y = torch.empty((B * H * W, 64), device=dev)
    xd, wd_, scd, shd = x.to(dev), w.to(dev), sc.to(dev), sh.to(dev)  # keep the device tensors alive across the raw call
    check(lib.vptr_conv7_in_fwd(ptr(xd), ptr(wd_), ptr(scd), ptr(shd), ptr(y), B, cimg, H, W, 64, stream()), "conv7_in")
    assert rel(y.reshape(B, H, W, 64).permute(0, 3, 1, 2), ref) < TOLV
    # output layer fwd + bwd-data
    for act, fn in ((1, torch.tanh), (2, torch.sigmoid)):
        xin = rn((B, 64, H, W), 104, 0.5).double().requires_grad_(True)
        w2, b2 = rn((cimg, 64, 7, 7), 105, 0.03), rn((cimg,), 106, 0.1)
        go = rn((B, cimg, H, W), 107)
        out = fn(F.conv2d(F.pad(xin, (3, 3, 3, 3), mode="reflect"), w2.double(), b2.double()))
        (out * go.double()).sum().backward()
        xt = xin.detach().float().permute(0, 2, 3, 1).reshape(-1, 64).contiguous().to(dev)
        yo = torch.empty((B, cimg, H, W), device=dev)
        w2g, b2g, gog = w2.to(dev), b2.to(dev), go.to(dev)
        check(lib.vptr_conv7_out_fwd(ptr(xt), ptr(w2g), ptr(b2g), ptr(yo), B, 64, H, W, cimg, act, stream()), "c7o")
        assert rel(yo, out) < TOLV
        dx = torch.empty((B * H * W, 64), device=dev)
        check(lib.vptr_conv7_out_bwd_data(ptr(gog), ptr(yo), ptr(w2g), ptr(dx), B, 64, H, W, cimg, act, stream()), "c7obd")
        assert rel(dx.reshape(B, H, W, 64).permute(0, 3, 1, 2), xin.grad) < 5e-5
        dw = torch.zeros((cimg, 64, 7, 7), device=dev)
        db = torch.zeros((cimg,), device=dev)
        w2d = w2.double().clone().requires_grad_(True)
        b2d = b2.double().clone().requires_grad_(True)
        out2 = fn(F.conv2d(F.pad(xin.detach(), (3, 3, 3, 3), mode="reflect"), w2d, b2d))
        (out2 * go.double()).sum().backward()
        check(lib.vptr_conv7_out_bwd_weight(ptr(gog), ptr(yo), ptr(xt), ptr(dw), ptr(db), B, 64, H, W, cimg, act, stream()),
              "c7obw")
        assert rel(dw, w2d.grad) < 5e-5 and rel(db, b2d.grad) < 5e-5
        # the workspace variant (second kernel: taps in registers, partial sums reduced in a fixed order); accumulates like the first
        wsp = torch.empty((lib.vptr_conv7_out_bwd_weight_workspace(B, cimg),), device=dev)
        dw2, db2 = torch.ones((cimg, 64, 7, 7), device=dev), torch.ones((cimg,), device=dev)
        check(lib.vptr_conv7_out_bwd_weight_ws(ptr(gog), ptr(yo), ptr(xt), ptr(dw2), ptr(db2), B, 64, H, W, cimg, act, ptr(wsp), wsp.numel(),
                                               stream()), "c7obw_ws")
        assert rel(dw2 - 1.0, w2d.grad) < 5e-5 and rel(db2 - 1.0, b2d.grad) < 5e-5


def test_bnrelu_bwd_fused(dev):
    """dx and the affine gradients of eval-BatchNorm + ReLU in one pass == the two separate kernels == autograd"""
    from vptr_amd._lib import check, lib, ptr, stream
    for rows, C in ((3000, 64), (1111, 128), (517, 256), (300, 24)):
        xh = rn((rows, C), 120)
        w, b, sc = rn((C,), 121).abs() + 0.5, rn((C,), 122, 0.3), rn((C,), 123).abs() + 0.2
        dy = rn((rows, C), 124)
        wd_, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
        y = torch.relu(xh.double() * wd_ + bd)
        (y * dy.double()).sum().backward()
        yd, dyd = y.detach().float().to(dev), dy.to(dev)
        dx, dw, db = torch.empty((rows, C), device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        wg, bg, sg = w.to(dev), b.to(dev), sc.to(dev)
        check(lib.vptr_bnrelu_bwd_fused(ptr(dyd), ptr(yd), ptr(sg), ptr(wg), ptr(bg), ptr(dx), ptr(dw), ptr(db), rows, C, stream()), "fused")
        assert rel(dx, (y.detach() > 0).double() * dy.double() * sc.double()) < 1e-6
        assert rel(dw, wd_.grad) < 2e-5 and rel(db, bd.grad) < 2e-5


# ------------------------------------------------------------------------------------------------------------ optimizer
def test_adamw_clip(ops, dev):
    from vptr_amd._lib import check, lib, ptr, stream
    n = 10007
    p0, g = rn((n,), 110), rn((n,), 111, 0.3)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3)
    pd, m, v = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    step = torch.zeros(1, device=dev)
    for it in range(3):
        gi = g * (it + 1)
        pr.grad = gi.clone()
        torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        ss = torch.zeros(1, device=dev)
        gd = gi.to(dev)
        check(lib.vptr_sumsq(ptr(gd), n, ptr(ss), stream()), "sumsq")
        step += 1
        check(lib.vptr_adamw(ptr(pd), ptr(gd), ptr(m), ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, ptr(step), ptr(ss), 1.0, 1.0,
                             stream()), "adamw")
    assert rel(pd, pr) < 1e-6


def test_window_copy_pad_and_crop(ops, dev):
    frames, H, W, C, ws = 3, 6, 5, 8, 4
    x = rn((frames * H * W, C), 90).to(dev).requires_grad_(True)
    y, Hp, Wp = ops.pad_tokens(x, frames, H, W, ws)
    assert (Hp, Wp) == (8, 8)
    ref = F.pad(x.detach().cpu().view(frames, H, W, C), (0, 0, (Wp - W) // 2, Wp - W - (Wp - W) // 2, (Hp - H) // 2, Hp - H - (Hp - H) // 2))
    assert torch.equal(y.detach().cpu().view(frames, Hp, Wp, C), ref)
    z = ops.crop_tokens(y, frames, Hp, Wp, H, W)
    assert torch.equal(z.detach().cpu(), x.detach().cpu())
    g = rn((frames * Hp * Wp, C), 91).to(dev)
    y.backward(g)
    gref = g.cpu().view(frames, Hp, Wp, C)[:, (Hp - H) // 2:(Hp - H) // 2 + H, (Wp - W) // 2:(Wp - W) // 2 + W].reshape(-1, C)
    assert torch.equal(x.grad.cpu(), gref)


# ---------------------------------------------------------------------------------------------------------------- losses
@pytest.mark.parametrize("shape", [(3, 2, 1, 64, 64), (2, 3, 3, 20, 36), (5, 1, 17, 9)])
def test_mse_gdl_fused(ops, dev, shape):
    """vptr_mse_gdl_fwd / bwd vs the oracle's MSELoss + GDL in fp64 (values 1e-6, gradient 2e-6), with unequal upstream weights"""
    gt = rn(shape, 901, 0.3)
    pr = rn(shape, 902, 0.3)
    pr[..., :3, :] = gt[..., :3, :]            # exact ties: sign(0) = 0 branches
    ref_p = pr.double().requires_grad_(True)
    lm, lg = O.mse_loss(gt.double(), ref_p), O.gdl_loss(gt.double(), ref_p)
    (0.7 * lm + 1.9 * lg).backward()
    x = pr.to(dev).requires_grad_(True)
    m, g = ops.mse_gdl(x, gt.to(dev))
    (0.7 * m + 1.9 * g).backward()
    assert abs(float(m) - float(lm)) < 1e-6 * abs(float(lm)) and abs(float(g) - float(lg)) < 1e-6 * abs(float(lg))
    assert rel(x.grad, ref_p.grad) < 2e-6


@pytest.mark.parametrize("frames,h,w,C,tau", [(6, 8, 8, 528, 1.0), (3, 16, 16, 48, 0.07), (2, 5, 7, 96, 0.5), (1, 4, 4, 16, 1.0)])
def test_bipatch_nce_fused(ops, dev, frames, h, w, C, tau):
    """vptr_nce_fwd / bwd (normalise + scores + both cross-entropies + stop-gradient structure) vs F.normalize + the oracle's
    BiPatchNCE in fp64: value 1e-6, both gradients 1e-5; one all-zero token exercises the eps clamp of the normalisation"""
    L = h * w
    g = rn((frames * L, C), 911, 1.0)
    p = rn((frames * L, C), 912, 1.0) + 0.5 * g
    p[1] = 0.0
    gd, pd = g.double().requires_grad_(True), p.double().requires_grad_(True)

    def as5(t):   # token-major [frames * L, C] -> (N = 1, T = frames, C, h, w)
        return t.reshape(1, frames, h, w, C).permute(0, 1, 4, 2, 3)
    ref = O.bipatch_nce(F.normalize(as5(gd), p=2.0, dim=2), F.normalize(as5(pd), p=2.0, dim=2), tau)
    (1.3 * ref).backward()
    gx, px = g.to(dev).requires_grad_(True), p.to(dev).requires_grad_(True)
    out = ops.nce_loss(gx, px, frames, L, tau)
    (1.3 * out).backward()
    assert abs(float(out) - float(ref)) < 2e-6 * abs(float(ref)), (float(out), float(ref))
    assert rel(gx.grad, gd.grad) < 1e-5, rel(gx.grad, gd.grad)
    keep = torch.ones(frames * L, dtype=torch.bool)
    keep[1] = False                               # the zero token: d/dx of x / max(|x|, 1e-12) is 1e12 * dxh -- compare the others
    assert rel(px.grad[keep.to(dev)], pd.grad[keep]) < 1e-5, rel(px.grad[keep.to(dev)], pd.grad[keep])


def test_droppath_scales(ops, dev):
    """vptr_droppath_scales: values in {0, 1/keep}, keep rates as requested, deterministic under the scope seed, different per site"""
    ops.manual_seed(dev, 4242)
    ops.new_seed_scope(dev)
    keep = torch.tensor([0.9, 0.5, 0.75], device=dev)
    a = ops.droppath_scales(keep, 20000, dev)
    b = ops.droppath_scales(keep, 20000, dev)
    assert torch.equal(a, b)
    for r, k in enumerate([0.9, 0.5, 0.75]):
        vals = torch.unique(a[r]).cpu()
        assert all(min(abs(float(v)), abs(float(v) - 1.0 / k)) < 1e-6 for v in vals), vals
        assert abs(float((a[r] > 0).float().mean()) - k) < 0.015
    c = ops.droppath_scales(keep, 20000, dev, site_offset=3)
    assert not torch.equal(a[0], c[0])
    ops.new_seed_scope(dev)
    assert not torch.equal(a, ops.droppath_scales(keep, 20000, dev))
