"""GPU parity of the P16 ("convert once") path: the plane format itself, the weight-plane builder, the DMA-staged nt GEMM with
its fused epilogues / batch members / K segments / K tail, the token-major grouped weight-gradient kernel (transposing LDS reads,
bias gradient by a ones-fragment, token tail), and the autograd wrappers that route nn.Linear-shaped work through them.
References are fp64 torch on the CPU; tolerance = the split-bf16 GEMM's 3e-5 rel-L2."""
import pytest
import torch
import torch.nn.functional as F

from helpers import rel
from oracle import fill

pytestmark = pytest.mark.gpu
TOL3 = 3e-5


@pytest.fixture(scope="module")
def ops():
    import vptr_amd.ops as ops
    return ops


def rn(shape, seed, scale=1.0):
    return fill.rand_normal(shape, seed, scale)


def test_p16_format_round_trip(ops, dev):
    x = rn((300, 528), 1) * torch.logspace(-6, 3, 528)[None, :]
    xp = ops.to_p16(x.to(dev))
    assert xp.shape == x.shape and xp.dtype == torch.float32
    back = ops.p16_decode(xp).cpu()
    assert float(((back - x).abs() / x.abs().clamp_min(1e-30)).max()) < 2.0 ** -16
    # layout: granule g of row r = 16 bf16 hi, then 16 bf16 lo, hi = bf16(x)
    raw = xp.cpu().view(torch.bfloat16).reshape(300, 33, 2, 16)
    assert torch.equal(raw[:, :, 0].reshape(300, 528), x.to(torch.bfloat16))


def test_weight_planes(ops, dev):
    Ws = [rn((528, 528), 2).to(dev), rn((2112, 528), 3).to(dev), rn((48, 192), 4).to(dev)]
    st = ops.WeightPlanes(Ws)
    for W in Ws:
        wp, ldw, wt, ldt = st.lookup(W)
        assert ldw == W.shape[1] and ldt == W.shape[0]
        assert float((ops.p16_decode(wp) - W).abs().max()) <= 2.0 ** -16 * float(W.abs().max())
        assert float((ops.p16_decode(wt.contiguous()) - W.t()).abs().max()) <= 2.0 ** -16 * float(W.abs().max())
    # row slices of a packed in_proj weight: forward planes are rows, transposed planes are a column block
    wp, ldw, wt, ldt = st.lookup(Ws[1][528:1056])
    assert torch.equal(ops.p16_decode(wp), ops.p16_decode(st.lookup(Ws[1])[0])[528:1056])
    assert wt.shape == (528, 528) and ldt == 2112
    # stale planes are rebuilt when the weight changed through torch
    Ws[0].mul_(2.0)
    assert float((ops.p16_decode(st.lookup(Ws[0])[0]) - Ws[0]).abs().max()) <= 2.0 ** -15 * float(Ws[0].abs().max())


@pytest.mark.parametrize("M,N,K", [(300, 528, 528), (1000, 1584, 528), (128, 352, 2112), (64, 48, 48), (257, 96, 80), (5000, 2112, 64),
                                   (10, 16, 16)])
def test_gemm_p16_nt_epilogue(ops, dev, M, N, K):
    x, W, b, r = rn((M, K), 1), rn((N, K), 2, K ** -0.5), rn((N,), 3), rn((M, N), 4)
    ref_pre = (x.double() @ W.double().t() + b.double()) * 0.5
    ref = F.gelu(ref_pre) + r.double()
    xp, Wp = ops.to_p16(x.to(dev)), ops.to_p16(W.to(dev))
    y, pre = torch.empty((M, N), device=dev), torch.empty((M, N), device=dev)
    ops.gemm_raw(xp, Wp, y, M, N, K, ops.A_P16, ops.B_P16, bias=b.to(dev), alpha=0.5, act=ops.ACT_GELU, Dpre=pre, residual=r.to(dev))
    assert rel(y, ref) < TOL3 and rel(pre, ref_pre) < TOL3
    # the same product written as P16 (the operand format of a following GEMM)
    yp = torch.empty((M, N), device=dev)
    ops.gemm_raw(xp, Wp, yp, M, N, K, ops.A_P16, ops.B_P16, bias=b.to(dev), alpha=0.5, act=ops.ACT_GELU, Dpre=pre, residual=r.to(dev), d_p16=True)
    assert rel(ops.p16_decode(yp), ref) < TOL3
    # row scale + ReLU-after-residual + plain
    rs = rn((7,), 5).abs().to(dev)
    ops.gemm_raw(xp, Wp, y, M, N, K, ops.A_P16, ops.B_P16, rowscale=rs, rs_div=3, rs_mod=7, residual=r.to(dev), act_after=True)
    idx = (torch.arange(M) // 3) % 7
    ref2 = torch.relu((x.double() @ W.double().t()) * rs.cpu().double()[idx][:, None] + r.double())
    assert rel(y, ref2) < TOL3


@pytest.mark.parametrize("M,N,K", [(300, 528, 528), (5000, 528, 528), (130, 96, 80)])
def test_gemm_p16_batch_and_segments(ops, dev, M, N, K):
    xs = [ops.to_p16(rn((M, K), 200 + i).to(dev)) for i in range(3)]
    Wf = [rn((N, K), 210 + i, K ** -0.5) for i in range(3)]
    Ws = [ops.to_p16(w.to(dev)) for w in Wf]
    bs = [rn((N,), 220 + i).to(dev) for i in range(3)]
    al = [0.25, 1.0, 2.0]
    ys = [torch.empty((M, N), device=dev) for _ in range(3)]
    ops.gemm_raw(xs[0], Ws[0], ys[0], M, N, K, ops.A_P16, ops.B_P16, bias=bs[0], alpha=al[0],
                 batch_extra=[(xs[1], Ws[1], ys[1], bs[1], al[1]), (xs[2], Ws[2], ys[2], None, al[2])])
    for i in range(3):
        ref = (ops.p16_decode(xs[i]).double().cpu() @ Wf[i].double().t() + (bs[i].double().cpu() if i < 2 else 0.0)) * al[i]
        assert rel(ys[i], ref) < TOL3
    # K segments: D = sum_s A_s . B_s^T with a residual; B_s here are the same [N, K] planes
    r = rn((M, N), 260).to(dev)
    y = torch.empty((M, N), device=dev)
    ops.gemm_raw(xs[0], Ws[0], y, M, N, K, ops.A_P16, ops.B_P16, alpha=0.5, residual=r, kseg_extra=[(xs[1], Ws[1]), (xs[2], Ws[2])])
    ref = sum(ops.p16_decode(a).double().cpu() @ w.double().t() for a, w in zip(xs, Wf)) * 0.5 + r.double().cpu()
    assert rel(y, ref) < TOL3


def test_wgrad_p16_grouped(ops, dev):
    """token-major P16 operands, several problems per launch, accumulation into pre-filled slabs, bias gradients, output scale,
    a token count that is not a multiple of 32 (masked tail) and row / column tiles that overhang the matrix"""
    probs = [(1024, 528, 528, 1.0), (640, 176, 2112, 1.0), (72, 48, 48, 0.5), (1000, 2112, 528, 1.0), (40, 16, 192, 2.0), (333, 528, 48, 1.0)]
    keep, refs = [], []
    for i, (T, N, K, alpha) in enumerate(probs):
        g, x = rn((T, N), 10 + i), rn((T, K), 20 + i)
        dW0, db0 = rn((N, K), 30 + i), rn((N,), 40 + i)
        gd, xd, dW, db = ops.to_p16(g.to(dev)), ops.to_p16(x.to(dev)), dW0.to(dev), db0.to(dev)
        keep.append((gd, xd, dW, db))
        refs.append((dW0.double() + alpha * (g.double().t() @ x.double()), db0.double() + alpha * g.double().sum(0)))
        ops.defer_wgrad(gd, xd, dW, N, K, T, db=db if i != 1 else None, alpha=alpha, p16=True)
    ops.flush_wgrads()
    for i, ((gd, xd, dW, db), (rW, rb)) in enumerate(zip(keep, refs)):
        assert rel(dW, rW) < TOL3, i
        if i != 1:
            assert rel(db, rb) < TOL3, i


def test_wgrad_p16_grouped_bench_size(ops, dev):
    """the grouped tn launch at the token count of the timed step (T = 10 240: 320 K-steps per tile) on the two weight shapes that
    dominate it, (2112, 528) and (528, 2112), plus a 528 x 528 problem with a bias gradient -- checked on GPU in fp64 against the
    P16-decoded operands (what the kernel is given), so the only error left is the kernel's own fp32 accumulation"""
    T = 10240
    probs = [(2112, 528), (528, 2112), (528, 528)]
    keep = []
    for i, (N, K) in enumerate(probs):
        gd = ops.to_p16(torch.randn((T, N), device=dev, generator=torch.Generator(device=dev).manual_seed(50 + i)) * 0.05)
        xd = ops.to_p16(torch.randn((T, K), device=dev, generator=torch.Generator(device=dev).manual_seed(60 + i)))
        dW, db = torch.zeros((N, K), device=dev), torch.zeros((N,), device=dev)
        keep.append((gd, xd, dW, db))
        ops.defer_wgrad(gd, xd, dW, N, K, T, db=db, alpha=1.0, p16=True)
    ops.flush_wgrads()
    for i, (gd, xd, dW, db) in enumerate(keep):
        g64, x64 = ops.p16_decode(gd).double(), ops.p16_decode(xd).double()
        assert rel(dW, g64.t() @ x64) < TOL3, i
        assert rel(db, g64.sum(0)) < TOL3, i


@pytest.mark.parametrize("M,N,K", [(10240, 2112, 528), (10240, 528, 2112), (10240, 528, 528), (2560, 2112, 528)])
def test_gemm_p16_nt_bench_size(ops, dev, M, N, K):
    """the nt instantiations the timed step launches: > 256-tile grids (two-stage loop) and the 240-tile N = 528 grids (four-stage
    loop), lean epilogue (bias + residual) and the GELU + saved pre-activation + P16-output epilogue of fc1 / linear1"""
    gen = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn((M, K), device=dev, generator=gen)
    W = torch.randn((N, K), device=dev, generator=gen) * K ** -0.5
    b, r = torch.randn((N,), device=dev, generator=gen), torch.randn((M, N), device=dev, generator=gen)
    xp, Wp = ops.to_p16(x), ops.to_p16(W)
    ref_pre = ops.p16_decode(xp).double() @ ops.p16_decode(Wp).double().t() + b.double()
    y = torch.empty((M, N), device=dev)
    ops.gemm_raw(xp, Wp, y, M, N, K, ops.A_P16, ops.B_P16, bias=b, residual=r)
    assert rel(y, ref_pre + r.double()) < TOL3
    yp, pre = torch.empty((M, N), device=dev), torch.empty((M, N), device=dev)
    ops.gemm_raw(xp, Wp, yp, M, N, K, ops.A_P16, ops.B_P16, bias=b, act=ops.ACT_GELU, Dpre=pre, d_p16=True)
    assert rel(pre, ref_pre) < TOL3 and rel(ops.p16_decode(yp), F.gelu(ref_pre)) < TOL3


def test_linear_autograd_p16(ops, dev):
    """ops.linear through the P16 kernels (inputs converted on the fly, weights through the plane cache) vs torch autograd,
    including a chain in which the first layer's output and the second layer's incoming gradient never exist as fp32"""
    M, K, F_, N = 640, 528, 2112, 528
    x = rn((M, K), 1).to(dev).requires_grad_(True)
    W1, b1 = rn((F_, K), 2, K ** -0.5).to(dev).requires_grad_(True), rn((F_,), 3).to(dev).requires_grad_(True)
    W2, b2 = rn((N, F_), 4, F_ ** -0.5).to(dev).requires_grad_(True), rn((N,), 5).to(dev).requires_grad_(True)
    r = rn((M, N), 6).to(dev).requires_grad_(True)
    g = rn((M, N), 7).to(dev)
    h = ops.linear(x, W1, b1, act=ops.ACT_GELU, out_p16=True)
    y = ops.linear(h, W2, b2, residual=r, x_p16=True)
    (y * g).sum().backward()
    xr, W1r, b1r, W2r, b2r, rr = [t.detach().double().cpu().requires_grad_(True) for t in (x, W1, b1, W2, b2, r)]
    yr = F.gelu(xr @ W1r.t() + b1r) @ W2r.t() + b2r + rr
    (yr * g.double().cpu()).sum().backward()
    assert rel(y, yr) < TOL3
    for got, ref in ((x, xr), (W1, W1r), (b1, b1r), (W2, W2r), (b2, b2r), (r, rr)):
        assert rel(got.grad, ref.grad) < 2 * TOL3


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_mlp_autograd_p16(ops, dev, p):
    """ops.mlp (one node: GELU' and the dropout mask applied by the epilogue of linear2's input-gradient GEMM, desc.act_grad_src) vs
    fp64 torch autograd without dropout, and vs the two-node ops.linear chain (same sites -> same masks) with dropout"""
    M, K, F_, N = 640, 528, 2112, 528
    vals = [rn((M, K), 1), rn((F_, K), 2, K ** -0.5), rn((F_,), 3), rn((N, F_), 4, F_ ** -0.5), rn((N,), 5), rn((M, N), 6)]
    g = rn((M, N), 7).to(dev)

    def leaves():
        return [t.clone().to(dev).requires_grad_(True) for t in vals]
    x, W1, b1, W2, b2, r = leaves()
    ops.manual_seed(dev, 77)
    y = ops.mlp(x, W1, b1, W2, b2, residual=r, dropout_p=p, site1=11, site2=12)
    (y * g).sum().backward()
    if p == 0.0:
        xr, W1r, b1r, W2r, b2r, rr = [t.double().requires_grad_(True) for t in vals]
        yr = F.gelu(xr @ W1r.t() + b1r) @ W2r.t() + b2r + rr
        (yr * g.double().cpu()).sum().backward()
        refs = (xr, W1r, b1r, W2r, b2r, rr)
    else:
        refs = leaves()
        ops.manual_seed(dev, 77)
        h = ops.linear(refs[0], refs[1], refs[2], act=ops.ACT_GELU, dropout_p=p, site=11, out_p16=True)
        yr = ops.linear(h, refs[3], refs[4], residual=refs[5], dropout_p=p, site=12, x_p16=True)
        (yr * g).sum().backward()
        assert float((y == 0).float().mean()) < 0.01 and rel(y, yr) < 1e-6       # identical masks, identical kernels in forward
    assert rel(y, yr) < TOL3
    for got, ref in zip((x, W1, b1, W2, b2, r), refs):
        assert rel(got.grad, ref.grad) < 2 * TOL3


def test_frame_stats_from_producers(ops, dev):
    """LayerNorm((F,H,W)) statistics accumulated by the producers (GEMM epilogue: desc.frame_stats; depthwise kernel) == the separate
    statistics pass: sums against fp64, and norm_act(raw_stats=...) against norm_act on its own statistics (forward and gradients)"""
    frames, HW, C, F_ = 12, 64, 48, 192
    rows = frames * HW
    x, W, b = rn((rows, C), 1).to(dev), rn((F_, C), 2, C ** -0.5).to(dev), rn((F_,), 3).to(dev)
    assert ops.frame_stats_ok(rows, HW, F_, 8)
    st = ops.frame_stats_buffer(frames, dev)
    y = ops.linear(x, W, b, frame_stats=st, frame_rows=HW)
    yd = y.double().view(frames, -1)
    assert float(((st[:, 1].double() - (yd ** 2).sum(1)).abs() / (yd ** 2).sum(1)).max()) < 1e-6
    assert float((st[:, 0].double() - yd.sum(1)).abs().max()) < 1e-6 * float(yd.abs().sum(1).max())
    st2 = ops.frame_stats_buffer(frames, dev)
    dw = rn((F_, 1, 3, 3), 4).to(dev)
    y2 = ops.dwconv3x3(y, dw, b, frames, 8, 8, frame_stats=st2)
    y2d = y2.double().view(frames, -1)
    assert float(((st2[:, 1].double() - (y2d ** 2).sum(1)).abs() / (y2d ** 2).sum(1)).max()) < 1e-6
    w, bb = rn((HW, F_), 5).to(dev).requires_grad_(True), rn((HW, F_), 6).to(dev).requires_grad_(True)
    outs = []
    for raw in (st2, None):
        xin = y2.detach().clone().requires_grad_(True)
        o = ops.norm_act(xin, w, bb, "ln", HW, True, raw_stats=raw)
        o.square().sum().backward()
        outs.append((o.detach(), xin.grad.clone(), w.grad.clone()))
        w.grad = None
        bb.grad = None
    for a, r in zip(outs[0], outs[1]):
        assert rel(a, r) < 1e-5


def test_frame_stats_large_mean_guard(ops, dev):
    """|mean| >> std in a LayerNorm((F,H,W)) input whose statistics come from its producer's single-pass sums: E[x^2] - mean^2 has lost
    its bits, the normalise kernel must notice (var < 1e-3 E[x^2]) and recompute the frame's variance around the mean"""
    frames, HW, C, F_ = 4, 64, 64, 528
    rows = frames * HW
    x, W = rn((rows, C), 30), rn((F_, C), 31, 0.002)
    b = torch.full((F_,), 40.0)                         # fc1 output = 40 +- ~0.02: E[x^2] / var ~ 4e6
    nw, nb = rn((HW, F_), 36).abs() + 0.5, rn((HW, F_), 37, 0.1)
    x, W, b, nw, nb = (t.to(dev) for t in (x, W, b, nw, nb))
    assert ops.frame_stats_ok(rows, HW, F_)
    st = ops.frame_stats_buffer(frames, dev)
    y = ops.linear(x, W, b, frame_stats=st, frame_rows=HW)
    a = ops.norm_act(y, nw, nb, "ln", HW, True, raw_stats=st)
    yd = y.double().reshape(frames, HW * F_)
    xh = (yd - yd.mean(1, keepdim=True)) / torch.sqrt(yd.var(1, unbiased=False, keepdim=True) + 1e-5)
    z = xh.reshape(rows, F_) * nw.double().repeat(frames, 1) + nb.double().repeat(frames, 1)
    ref = 0.5 * z * (1.0 + torch.erf(z / 2 ** 0.5))
    assert float((a.double() - ref).norm() / ref.norm()) < 1e-4


@pytest.mark.parametrize("geo", [(6, 8, 8, 48, 192), (2, 16, 16, 32, 96), (3, 16, 16, 32, 128)], ids=["lds8x8", "walk16x16_F96", "lds16x16"])
def test_norm_dwconv_fused_matches_chain_and_fp64(ops, dev, geo):
    """round 6: LayerNorm((F,H,W)) + GELU folded into the depthwise kernel's load path (ops.norm_dwconv3x3: VidHRFormer_modules.py:430-434)
    == norm_act followed by dwconv3x3 (forward, frame statistics of the output, every gradient) and == fp64 torch.  The depthwise WEIGHT
    gradient reads the activated tensor from an fp16 side copy: its bound is the one the model-level parity tests leave (2e-4)."""
    # F % 64 == 0 and H * W <= 256: the LDS-slab kernel (dwconv_norm_lds_kernel); F = 96 on 16 x 16 maps: the register-walk kernel
    frames, H, W, C, F_ = geo
    HW, rows = H * W, frames * H * W
    assert ops.norm_dwconv_ok(rows, HW, F_, H, W) == (HW <= 64)    # default: only where the fused launch pays (8 x 8 maps, cache-resident tensors)
    ops.config.fused_norm_dwconv_mode = 2
    try:
        assert ops.norm_dwconv_ok(rows, HW, F_, H, W)               # ... but valid on the larger maps too
    finally:
        ops.config.fused_norm_dwconv_mode = 1
    x0, W1, b1 = rn((rows, C), 1).to(dev), rn((F_, C), 2, C ** -0.5).to(dev), rn((F_,), 3).to(dev)
    aw, ab = (rn((HW, F_), 5).abs() + 0.5).to(dev), rn((HW, F_), 6, 0.3).to(dev)
    dwt, dbias = rn((F_, 1, 3, 3), 7, 0.3).to(dev), rn((F_,), 8, 0.1).to(dev)
    cot = rn((rows, F_), 9).to(dev)
    res = []
    for fused in (True, False):
        st, st2 = ops.frame_stats_buffer(frames, dev), ops.frame_stats_buffer(frames, dev)
        with torch.no_grad():
            y = ops.linear(x0, W1, b1, frame_stats=st, frame_rows=HW)
        xin = y.clone().requires_grad_(True)
        pw, pb, pd, pdb = (t.clone().requires_grad_(True) for t in (aw, ab, dwt, dbias))
        if fused:
            o = ops.norm_dwconv3x3(xin, pw, pb, pd, pdb, frames, H, W, st, frame_stats=st2)
        else:
            a = ops.norm_act(xin, pw, pb, "ln", HW, True, raw_stats=st)
            o = ops.dwconv3x3(a, pd, pdb, frames, H, W, frame_stats=st2)
        (o * cot).sum().backward()
        res.append((o.detach(), st2.clone(), xin.grad.clone(), pw.grad.clone(), pb.grad.clone(), pd.grad.clone(), pdb.grad.clone(), y))
    names = ("y", "stats", "dx", "daff_w", "daff_b", "ddw", "ddb")
    for n, a, r in zip(names, res[0], res[1]):
        assert rel(a, r) < (2e-4 if n == "ddw" else 2e-5), (n, rel(a, r))
    # fp64 reference on the CPU
    y = res[0][7].double().cpu().requires_grad_(True)
    pw, pb, pd, pdb = (t.double().cpu().requires_grad_(True) for t in (aw, ab, dwt, dbias))
    yf = y.view(frames, HW * F_)
    xh = ((yf - yf.mean(1, keepdim=True)) / torch.sqrt(yf.var(1, unbiased=False, keepdim=True) + 1e-5)).view(frames, HW, F_)
    a = F.gelu(xh * pw + pb)
    o = F.conv2d(a.view(frames, H, W, F_).permute(0, 3, 1, 2), pd, pdb, padding=1, groups=F_).permute(0, 2, 3, 1).reshape(rows, F_)
    (o * cot.double().cpu()).sum().backward()
    for n, a_, r in zip(names, res[0], (o.detach(), None, y.grad, pw.grad, pb.grad, pd.grad, pdb.grad)):
        if r is not None:
            assert rel(a_.double().cpu(), r) < (2e-4 if n == "ddw" else 3e-5), (n, rel(a_.double().cpu(), r))


def test_norm_dwconv_large_mean_guard(ops, dev):
    """the fused kernel's per-wave form of the large-mean guard (test_frame_stats_large_mean_guard): fc1 output 40 +- 0.02"""
    frames, H, W, C, F_ = 4, 8, 8, 64, 512
    HW, rows = H * W, frames * H * W
    assert ops.norm_dwconv_ok(rows, HW, F_, H, W)
    x, W1 = rn((rows, C), 30), rn((F_, C), 31, 0.002)
    b = torch.full((F_,), 40.0)
    nw, nb = rn((HW, F_), 36).abs() + 0.5, rn((HW, F_), 37, 0.1)
    dwt = rn((F_, 1, 3, 3), 38, 0.3)
    x, W1, b, nw, nb, dwt = (t.to(dev) for t in (x, W1, b, nw, nb, dwt))
    st = ops.frame_stats_buffer(frames, dev)
    y = ops.linear(x, W1, b, frame_stats=st, frame_rows=HW)
    with torch.no_grad():
        o = ops.norm_dwconv3x3(y, nw, nb, dwt, None, frames, H, W, st)
    yd = y.double().reshape(frames, HW * F_)
    xh = (yd - yd.mean(1, keepdim=True)) / torch.sqrt(yd.var(1, unbiased=False, keepdim=True) + 1e-5)
    z = xh.reshape(frames, HW, F_) * nw.double() + nb.double()
    a = 0.5 * z * (1.0 + torch.erf(z / 2 ** 0.5))
    ref = F.conv2d(a.view(frames, H, W, F_).permute(0, 3, 1, 2), dwt.double(), None, padding=1, groups=F_).permute(0, 2, 3, 1).reshape(rows, F_)
    assert float((o.double() - ref).norm() / ref.norm()) < 1e-4


@pytest.mark.parametrize("frames,HW,F_,p", [(32, 64, 192, 0.0), (23, 64, 528, 0.1), (160, 64, 2112, 0.1), (40, 256, 96, 0.0)])
def test_norm_act_bwd_coop_matches_two_phase(ops, dev, frames, HW, F_, p):
    """round 6: the cooperative ONE-pass LayerNorm((F,H,W)) backward (vptr_norm_act_bwd_coop: workgroups of a 10-frame chunk exchange the frames'
    sums through one workspace line per frame) == the two-phase call (frame sums + affine gradients, then dx): dx and the affine
    gradients (partial rows summed), with dropout + row scale, fp32 and P16 dx; the time-out flag of the workspace stays clear"""
    from vptr_amd._lib import check, lib, ptr, stream
    rows = frames * HW
    assert lib.vptr_norm_act_bwd_coop_partials(rows, F_, HW) == (frames + 9) // 10
    x, dy = rn((rows, F_), 1).to(dev), rn((rows, F_), 2).to(dev)
    w, b = (rn((HW, F_), 3).abs() + 0.5).to(dev), rn((HW, F_), 4, 0.3).to(dev)
    xf = x.view(frames, -1)
    mean = xf.mean(1).contiguous()
    rstd = torch.rsqrt(xf.var(1, unbiased=False) + 1e-5).contiguous()
    rowscale = (torch.rand(frames, device=dev) > 0.2).float() / 0.8
    seed = ops.seed_tensor(dev) if p > 0 else None
    for p16 in (0, 1):
        dx_a, dx_b = torch.empty_like(x), torch.empty_like(x)
        dw2, db2 = torch.zeros(HW * F_, device=dev), torch.zeros(HW * F_, device=dev)
        scratch = torch.empty((max(2 * F_, 2 * frames * (1 + 4 * ((HW * F_ // 4 + 255) // 256))),), device=dev)
        check(lib.vptr_norm_act_bwd(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(w), ptr(b), ptr(dx_a), ptr(dw2), ptr(db2), ptr(scratch), rows, F_, HW, 0, 1, 0,
                                    p, ptr(seed), 7, ptr(rowscale), HW, frames, p16, stream()), "two-phase")
        part2 = torch.stack([dw2, db2])
        n1 = (frames + 9) // 10
        part1 = torch.empty((n1, 2, HW * F_), device=dev)
        ws = torch.zeros((frames + 1) * 32, device=dev)
        check(lib.vptr_norm_act_bwd_coop(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(w), ptr(b), ptr(dx_b), ptr(ws), rows, F_, HW, 1, p, ptr(seed), 7,
                                         ptr(rowscale), HW, frames, p16, ptr(part1), stream()), "coop")
        torch.cuda.synchronize()
        assert int(ws.view(torch.int32)[frames * 32]) == 0, "a bounded wait of the cooperative kernel ran out"
        assert int(ws.view(torch.int32).view(frames + 1, 32)[:frames, 2].min()) == (HW * F_ // 4 + 255) // 256
        a = ops.p16_decode(dx_a) if p16 else dx_a
        bb = ops.p16_decode(dx_b) if p16 else dx_b
        assert rel(bb, a) < 2e-6, (p16, rel(bb, a))
        assert rel(part1.sum(0), part2) < 2e-6
