import torch, torch.nn.functional as F, math
torch.manual_seed(0)
# Winograd F(4x4,3x3) matrices (Lavin & Gray), points 0, +-1, +-2, inf
def mats(points=(1.0, 2.0), dtype=torch.float64):
    a, b = points
    # general construction via Vandermonde (Toom-Cook): polynomial points p = [0, a, -a, b, -b], inf
    import numpy as np
    p = [0.0, a, -a, b, -b]
    n, r = 4, 3
    alpha = n + r - 1  # 6
    # A^T (n x alpha): rows i: p_j^i ; last column: inf -> e_{n-1}
    AT = np.zeros((n, alpha)); 
    for i in range(n):
        for j in range(5): AT[i, j] = p[j] ** i
    AT[n - 1, 5] = 1.0
    # G (alpha x r): rows j: p_j^k / N_j ; N_j = prod_{m != j} (p_j - p_m); last row e_{r-1}
    G = np.zeros((alpha, r))
    for j in range(5):
        Nj = np.prod([p[j] - p[m] for m in range(5) if m != j])
        for k in range(r): G[j, k] = p[j] ** k / Nj
    G[5, r - 1] = 1.0
    # B^T (alpha x alpha): from the Lagrange basis: row j coefficients of prod_{m != j}(x - p_m) ; last row: coefficients of prod_m (x - p_m)
    BT = np.zeros((alpha, alpha))
    for j in range(5):
        c = np.poly([p[m] for m in range(5) if m != j])[::-1]  # ascending powers, degree 4
        BT[j, :5] = c
    c = np.poly(p)[::-1]
    BT[5, :6] = c
    return torch.tensor(AT, dtype=dtype), torch.tensor(G, dtype=dtype), torch.tensor(BT, dtype=dtype)

def check_mats(points):
    AT, G, BT = mats(points)
    d = torch.randn(6, dtype=torch.float64); g = torch.randn(3, dtype=torch.float64)
    y = AT @ ((G @ g) * (BT @ d))
    ref = torch.stack([(d[i:i+3] * g).sum() for i in range(4)])
    return float((y - ref).abs().max())
print("1D check", check_mats((1.0, 2.0)), check_mats((0.5, 1.0)), check_mats((1.0, 0.5)))

def split_bf16(x):
    hi = x.to(torch.bfloat16).to(torch.float32)
    lo = (x - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo
def mm3(a, b):   # emulated 3-pass split-bf16 product, fp32 accumulate;  a [.., M, K] b [.., K, N]
    ah, al = split_bf16(a); bh, bl = split_bf16(b)
    return ah @ bh + ah @ bl + al @ bh

def pad_reflect(x): return F.pad(x, (1, 1, 1, 1), mode="reflect")

def conv_direct3(x, w):   # x [B,C,8,8] fp32, w [K,C,3,3] fp32: emulated 3-pass implicit GEMM
    B, C, H, W = x.shape
    cols = F.unfold(pad_reflect(x), 3)              # [B, C*9, 64]
    A = cols.transpose(1, 2).reshape(B * 64, C * 9)
    out = mm3(A, w.reshape(w.shape[0], -1).t())
    return out.reshape(B, 64, -1).permute(0, 2, 1).reshape(B, -1, H, W)

def conv_wino(x, U, AT, BT, emulate=True):   # U [36, C, K] transformed filters (fp32)
    B, C, H, W = x.shape
    xp = pad_reflect(x)                                   # [B,C,10,10]
    tiles = xp.unfold(2, 6, 4).unfold(3, 6, 4)           # [B,C,2,2,6,6]
    V = torch.einsum("ij,bctujk,lk->bctuil", BT, tiles, BT)  # B^T d B
    V = V.permute(4, 5, 0, 2, 3, 1).reshape(36, B * 4, C)
    M = mm3(V, U) if emulate else V @ U                  # [36, B*4, K]
    M = M.reshape(6, 6, B, 2, 2, -1)
    Y = torch.einsum("ij,jkbtuc,lk->btuilc", AT, M, AT)  # [B,2,2,4,4,K]
    return Y.permute(0, 5, 1, 3, 2, 4).reshape(B, -1, 8, 8)

def filt(w, G):   # w [K,C,3,3] -> U [36, C, K]
    U = torch.einsum("ij,kcjl,ml->imck", G, w, G)
    return U.reshape(36, w.shape[1], w.shape[0])

C = 528; Bn = 8
for pts in ((1.0, 2.0), (0.5, 1.0), (1.0, 0.5), (0.5, 2.0)):
    AT, G, BT = mats(pts)
    ws = [torch.randn(C, C, 3, 3, dtype=torch.float64) * (2.0 / (9 * C)) ** 0.5 for _ in range(18)]
    bs = [torch.randn(C, dtype=torch.float64) * 0.1 for _ in range(18)]
    x0 = torch.relu(torch.randn(Bn, C, 8, 8, dtype=torch.float64))
    # fp64 direct reference through 9 residual blocks
    def run(conv):
        y = x0.clone() if conv == "ref" else x0.float()
        outs = []
        for bi in range(9):
            w1, w2 = ws[2 * bi], ws[2 * bi + 1]
            if conv == "ref":
                t = torch.relu(F.conv2d(pad_reflect(y), w1) + bs[2 * bi][None, :, None, None])
                y = y + F.conv2d(pad_reflect(t), w2) + bs[2 * bi + 1][None, :, None, None]
            elif conv == "direct3":
                t = torch.relu(conv_direct3(y, w1.float()) + bs[2 * bi].float()[None, :, None, None])
                y = y + conv_direct3(t, w2.float()) + bs[2 * bi + 1].float()[None, :, None, None]
            else:
                U1, U2 = filt(w1, G).float(), filt(w2, G).float()
                t = torch.relu(conv_wino(y, U1, AT.float(), BT.float()) + bs[2 * bi].float()[None, :, None, None])
                y = y + conv_wino(t, U2, AT.float(), BT.float()) + bs[2 * bi + 1].float()[None, :, None, None]
            outs.append(y)
        return outs
    ref = run("ref")
    for name in (("direct3", "wino") if pts == (1.0, 2.0) else ("wino",)):
        o = run(name)
        errs = [float((a.double() - r).norm() / r.norm()) for a, r in zip(o, ref)]
        mx = float((o[-1].double() - ref[-1]).abs().max() / ref[-1].abs().max())
        print(pts, name, "rel err after block 1, 5, 9: %.2e %.2e %.2e   max-abs/max %.2e" % (errs[0], errs[4], errs[8], mx))
