// Loss kernels of the VPTR train steps (gfx950): MSE + gradient-difference loss in one pass over the frames, the bidirectional
// patch-wise contrastive loss (BiPatchNCE) including the L2 normalisation in front of it, and the stochastic-depth scale vectors.
//
// Why these are not left to ATen any more (round 3): a whole-step hipGraph must not contain MEMSET nodes -- on this ROCm stack a
// captured hipMemsetAsync writes garbage from the second replay on (tools/memset_node_probe.py, profiles/r03_memset_node_probe.log),
// and ATen's multi-block reductions zero their semaphores with exactly that call (Reduce.cuh), so e.g. the backward of
// F.normalize(dim=2) left its output unwritten in every replay but the first.  Everything here is plain kernel nodes: partial sums
// go to a scratch row per workgroup and a single-workgroup kernel adds them in a fixed order (deterministic, no atomics, no memset).
#include "common.h"

// ------------------------------------------------------------------------------------------------------------------------------
// MSE + GDL (model/criterion.py:105-132, 134-204 with alpha = 1, no temporal weights): per image plane [H][W]
//   mse = mean (p - g)^2 ; gdl = mean | |g[y+1]-g[y]| - |p[y+1]-p[y]| |  +  mean | |g[x]-g[x+1]| - |p[x]-p[x+1]| |
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }   // torch.sign: sign(0) = 0

// one workgroup per (plane, 16-row band); partial[(blk) * 3 + {0,1,2}] = sum sq, sum dh, sum dw of the band
__global__ __launch_bounds__(256) void mse_gdl_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                          float* __restrict__ partial, int H, int W, int bands) {
  __shared__ float red[16];
  const int plane = blockIdx.x / bands, band = blockIdx.x % bands;
  const int y0 = band * 16, y1 = min(H, y0 + 16);
  const float* p = pred + (int64_t)plane * H * W;
  const float* g = gt + (int64_t)plane * H * W;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const int n = (y1 - y0) * W;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int y = y0 + i / W, x = i % W;
    const float pv = p[y * W + x], gv = g[y * W + x];
    const float d = pv - gv;
    s0 += d * d;
    if (y + 1 < H) s1 += fabsf(fabsf(g[(y + 1) * W + x] - gv) - fabsf(p[(y + 1) * W + x] - pv));
    if (x + 1 < W) s2 += fabsf(fabsf(gv - g[y * W + x + 1]) - fabsf(pv - p[y * W + x + 1]));
  }
  s0 = block_sum(s0, red);
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    partial[(int64_t)blockIdx.x * 3 + 0] = s0;
    partial[(int64_t)blockIdx.x * 3 + 1] = s1;
    partial[(int64_t)blockIdx.x * 3 + 2] = s2;
  }
}

// fixed-order sum of `n` rows of `k` partials each (double accumulators), out[j] = sum_j * scale[j] (+ combined terms, see callers)
__global__ __launch_bounds__(256) void mse_gdl_final_kernel(const float* __restrict__ partial, int n, double inv_mse, double inv_h, double inv_w,
                                                            float* __restrict__ mse_out, float* __restrict__ gdl_out) {
  __shared__ double red[3][256];
  double a = 0.0, b = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    a += (double)partial[(int64_t)i * 3 + 0];
    b += (double)partial[(int64_t)i * 3 + 1];
    c += (double)partial[(int64_t)i * 3 + 2];
  }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
      red[2][threadIdx.x] += red[2][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *mse_out = (float)(red[0][0] * inv_mse);
    *gdl_out = (float)(red[1][0] * inv_h + red[2][0] * inv_w);
  }
}

// dpred = g_mse * d mse / d pred + g_gdl * d gdl / d pred ; g_* are DEVICE scalars (upstream gradients; null = 0)
__global__ __launch_bounds__(256) void mse_gdl_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                          const float* __restrict__ g_mse, const float* __restrict__ g_gdl,
                                                          float* __restrict__ dpred, int64_t total, int H, int W, float inv_mse, float inv_h,
                                                          float inv_w) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float gm = g_mse ? *g_mse : 0.f, gg = g_gdl ? *g_gdl : 0.f;
  const int x = (int)(i % W), y = (int)((i / W) % H);
  const float pv = pred[i], gv = gt[i];
  float r = gm * 2.f * (pv - gv) * inv_mse;
  float th = 0.f, tw = 0.f;
  // pair (y, y+1): dh = | |gd| - |pd| |, pd = p[y+1] - p[y]:  d dh / d p[y+1] = -sign(|gd|-|pd|) sign(pd),  d dh / d p[y] = +...
  if (y + 1 < H) { const float pd = pred[i + W] - pv, gd = gt[i + W] - gv; th += sgn(fabsf(gd) - fabsf(pd)) * sgn(pd); }
  if (y > 0)     { const float pd = pv - pred[i - W], gd = gv - gt[i - W]; th -= sgn(fabsf(gd) - fabsf(pd)) * sgn(pd); }
  // pair (x, x+1): dw = | |gd| - |pd| |, pd = p[x] - p[x+1]:  d dw / d p[x] = -sign(|gd|-|pd|) sign(pd),  d dw / d p[x+1] = +...
  if (x + 1 < W) { const float pd = pv - pred[i + 1], gd = gv - gt[i + 1]; tw -= sgn(fabsf(gd) - fabsf(pd)) * sgn(pd); }
  if (x > 0)     { const float pd = pred[i - 1] - pv, gd = gt[i - 1] - gv; tw += sgn(fabsf(gd) - fabsf(pd)) * sgn(pd); }
  dpred[i] = r + gg * (th * inv_h + tw * inv_w);
}

extern "C" int vptr_mse_gdl_fwd(const float* pred, const float* gt, float* scratch, float* mse_out, float* gdl_out, int planes, int H, int W,
                                vptr_stream_t stream) {
  VPTR_CHECK(pred && gt && scratch && mse_out && gdl_out && planes > 0 && H > 1 && W > 1, "mse_gdl_fwd: bad arguments");
  const int bands = cdiv(H, 16);
  const int nblk = planes * bands;
  mse_gdl_fwd_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>(pred, gt, scratch, H, W, bands);
  VPTR_LAUNCH_CHECK();
  const double n = (double)planes;
  mse_gdl_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(scratch, nblk, 1.0 / (n * H * W), 1.0 / (n * (H - 1) * W), 1.0 / (n * H * (W - 1)), mse_out,
                                                           gdl_out);
  VPTR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vptr_mse_gdl_bwd(const float* pred, const float* gt, const float* g_mse, const float* g_gdl, float* dpred, int planes, int H, int W,
                                vptr_stream_t stream) {
  VPTR_CHECK(pred && gt && dpred && planes > 0 && H > 1 && W > 1, "mse_gdl_bwd: bad arguments");
  const int64_t total = (int64_t)planes * H * W;
  const double n = (double)planes;
  mse_gdl_bwd_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(pred, gt, g_mse, g_gdl, dpred, total, H, W, (float)(1.0 / (n * H * W)),
                                                                        (float)(1.0 / (n * (H - 1) * W)), (float)(1.0 / (n * H * (W - 1))));
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// BiPatchNCE (model/criterion.py:206-259) on token-major projections g = proj(gt features), p = proj(predicted features), both
// [frames * L, C] with L = h*w patches per frame, INCLUDING the F.normalize(p=2, dim=channel) of train_NAR.py:83-84:
//   gh = g / max(|g|, eps), ph likewise;  S = gh ph^T / tau per frame (L x L)
//   s1 = S with the gradient to p cut off the diagonal, s2 = S^T with the gradient to g cut off the diagonal (the .detach() terms)
//   loss = 0.5 * (CE(s1 rows, diagonal) + CE(s2 rows, diagonal)), mean over all frames * L rows
// forward: inverse norms -> S (HBM, L2-resident) -> per-frame row / column log-sum-exp + loss partials -> fixed-order sum
// backward: d gh_i = 1/tau [sum_j A_ij ph_j + Bc_ii ph_i], d ph_j = 1/tau [sum_i Bc_ij gh_i + A_jj gh_j] with
//   A_ij = c (exp(S_ij - rlse_i) - delta_ij), Bc_ij = c (exp(S_ij - clse_j) - delta_ij), c = 0.5 / rows; then the normalisation's
//   backward dx = (d xh - xh (xh . d xh)) / |x|  in the same kernel.
// ------------------------------------------------------------------------------------------------------------------------------
#define NCE_EPS 1e-12f

// a wave per row: inv[r] = 1 / max(|x_r|, eps) for the rows of both tensors (rows of g first, then rows of p)
__global__ __launch_bounds__(256) void nce_invnorm_kernel(const float* __restrict__ g, const float* __restrict__ p, float* __restrict__ inv, int R, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= 2 * R) return;
  const float* x = (row < R) ? g + (int64_t)row * C : p + (int64_t)(row - R) * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = x[c]; s += v * v; }
  s = wave_sum(s);
  if (lane == 0) inv[row] = 1.f / fmaxf(sqrtf(s), NCE_EPS);
}

// S[f][i][j] = (g_i . p_j) * inv_g[i] * inv_p[j] / tau ; one workgroup per (frame, 64 x 64 block); 16 x 16 threads, 4 x 4 outputs each
__global__ __launch_bounds__(256) void nce_scores_kernel(const float* __restrict__ g, const float* __restrict__ p, const float* __restrict__ inv,
                                                         float* __restrict__ S, int R, int L, int C, float inv_tau) {
  __shared__ float sG[16][68], sP[16][68];
  const int nb = (L + 63) / 64;
  const int f = blockIdx.x / (nb * nb), bi = (blockIdx.x / nb) % nb, bj = blockIdx.x % nb;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int lr = threadIdx.x >> 2, lk = (threadIdx.x & 3) * 4;   // staging: 64 rows x 4 float4 of a 16-channel chunk
  const int gi = bi * 64 + lr, pj = bj * 64 + lr;
  const float* grow = g + ((int64_t)f * L + min(gi, L - 1)) * C;
  const float* prow = p + ((int64_t)f * L + min(pj, L - 1)) * C;
  float acc[4][4] = {};
  for (int c0 = 0; c0 < C; c0 += 16) {
    float gv[4], pv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + lk + e;
      gv[e] = (c < C && gi < L) ? grow[c] : 0.f;
      pv[e] = (c < C && pj < L) ? prow[c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) { sG[lk + e][lr] = gv[e]; sP[lk + e][lr] = pv[e]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(&sG[k][ty * 4]);
      const float4 b = *reinterpret_cast<const float4*>(&sP[k][tx * 4]);
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[r][q] += av[r] * bv[q];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = bi * 64 + ty * 4 + r;
    if (i >= L) continue;
    const float ig = inv[(int64_t)f * L + i] * inv_tau;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = bj * 64 + tx * 4 + q;
      if (j < L) S[((int64_t)f * L + i) * L + j] = acc[r][q] * ig * inv[R + (int64_t)f * L + j];
    }
  }
}

// one workgroup per frame: rlse[i] = logsumexp_j S_ij, clse[j] = logsumexp_i S_ij, partial[f] = sum_i (rlse_i + clse_i - 2 S_ii)
__global__ __launch_bounds__(256) void nce_stats_kernel(const float* __restrict__ S, float* __restrict__ rlse, float* __restrict__ clse,
                                                        float* __restrict__ partial, int L) {
  __shared__ float red[16];
  const int f = blockIdx.x;
  const float* Sf = S + (int64_t)f * L * L;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float loss = 0.f;
  for (int i = w; i < L; i += 4) {   // rows: a wave per row
    float m = -INFINITY;
    for (int j = lane; j < L; j += 64) m = fmaxf(m, Sf[(int64_t)i * L + j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < L; j += 64) s += __expf(Sf[(int64_t)i * L + j] - m);
    s = wave_sum(s);
    const float l = m + __logf(s);
    if (lane == 0) { rlse[(int64_t)f * L + i] = l; loss += l - 2.f * Sf[(int64_t)i * L + i]; }
  }
  for (int j = threadIdx.x; j < L; j += 256) {   // columns: a thread per column (coalesced across the workgroup), online softmax
    float m = -INFINITY, s = 0.f;
    for (int i = 0; i < L; ++i) {
      const float v = Sf[(int64_t)i * L + j];
      if (v > m) { s = s * __expf(m - v) + 1.f; m = v; }
      else s += __expf(v - m);
    }
    const float l = m + __logf(s);
    clse[(int64_t)f * L + j] = l;
    loss += l;
  }
  loss = block_sum(loss, red);
  if (threadIdx.x == 0) partial[f] = loss;
}

__global__ __launch_bounds__(256) void nce_final_kernel(const float* __restrict__ partial, int n, double scale, float* __restrict__ out) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += (double)partial[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = (float)(red[0] * scale);
}

// one workgroup per (frame, 64-row block, direction): dir 0 -> d g rows, dir 1 -> d p rows.  4 waves x 16 rows; 4 lanes per row, each
// with every 4th float4 of the row's channels (NV float4 accumulators per lane; C <= 16 * NV).
template <int NV>
__global__ __launch_bounds__(256) void nce_bwd_kernel(const float* __restrict__ g, const float* __restrict__ p, const float* __restrict__ inv,
                                                      const float* __restrict__ S, const float* __restrict__ rlse, const float* __restrict__ clse,
                                                      const float* __restrict__ gout, float* __restrict__ dg, float* __restrict__ dp, int R, int L, int C,
                                                      float inv_tau, float cmean) {
  extern __shared__ float smem[];
  float* sM = smem;                 // [64][17] coefficients of this row block against 16 "other" tokens
  float* sY = smem + 64 * 17;       // [16][C4 * 4] normalised "other" tokens
  const int nb = (L + 63) / 64;
  const int dir = blockIdx.x & 1, bi = (blockIdx.x >> 1) % nb, f = (blockIdx.x >> 1) / nb;
  const int C4 = (C + 3) / 4;
  const float* X = dir ? p : g;     // the tensor whose gradient this workgroup produces
  const float* Y = dir ? g : p;     // the other one
  const float* invX = inv + (dir ? R : 0);
  const float* invY = inv + (dir ? 0 : R);
  const float* lseX = dir ? clse : rlse;   // log-sum-exp along this row's own softmax
  const float* lseD = dir ? rlse : clse;   // the other direction's, for the diagonal term
  float* dX = dir ? dp : dg;
  const float* Sf = S + (int64_t)f * L * L;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int rl = w * 16 + (lane >> 2), cg = lane & 3;
  const int row = bi * 64 + rl;     // token index inside the frame
  const float go = gout ? *gout : 1.f;
  float4 acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t0 = 0; t0 < L; t0 += 16) {
    __syncthreads();
    // coefficients M[r][t]: 64 x 16, one per 4 threads-iterations
    for (int e = threadIdx.x; e < 64 * 16; e += 256) {
      const int r = e >> 4, t = e & 15;
      const int i = bi * 64 + r, j = t0 + t;
      float m = 0.f;
      if (i < L && j < L) {
        const float s = dir ? Sf[(int64_t)j * L + i] : Sf[(int64_t)i * L + j];     // S is indexed [g token][p token]
        m = __expf(s - lseX[(int64_t)f * L + i]);
        if (i == j) m += __expf(s - lseD[(int64_t)f * L + i]) - 2.f;
        m *= cmean;
      }
      sM[r * 17 + t] = m;
    }
    // normalised other tokens: 16 rows x C
    for (int e = threadIdx.x; e < 16 * C4; e += 256) {
      const int t = e / C4, c4 = e % C4;
      const int j = t0 + t;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < L) {
        const float* yr = Y + ((int64_t)f * L + j) * C;
        const float s = invY[(int64_t)f * L + j];
        v = *reinterpret_cast<const float4*>(yr + c4 * 4);   // C % 4 == 0 (checked by the launcher)
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
      }
      *reinterpret_cast<float4*>(&sY[(t * C4 + c4) * 4]) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int t = 0; t < 16; ++t) {
      const float m = sM[rl * 17 + t];
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c4 = k * 4 + cg;
        if (c4 < C4) {
          const float4 y = *reinterpret_cast<const float4*>(&sY[(t * C4 + c4) * 4]);
          acc[k].x += m * y.x; acc[k].y += m * y.y; acc[k].z += m * y.z; acc[k].w += m * y.w;
        }
      }
    }
  }
  // d xh = acc * go / tau ; dx = (d xh - xh (xh . d xh)) * inv   (|x| < eps: the clamp cuts the norm's gradient: dx = d xh / eps)
  const bool live = row < L;
  const int64_t rr = (int64_t)f * L + (live ? row : 0);
  const float* xr = X + rr * C;
  const float iv = invX[rr];
  const float sc = go * inv_tau;
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c4 = k * 4 + cg;
    if (c4 < C4) {
      const float4 x = *reinterpret_cast<const float4*>(xr + c4 * 4);
      acc[k].x *= sc; acc[k].y *= sc; acc[k].z *= sc; acc[k].w *= sc;
      dot += x.x * acc[k].x + x.y * acc[k].y + x.z * acc[k].z + x.w * acc[k].w;
    }
  }
  dot += __shfl_xor(dot, 1, 64);
  dot += __shfl_xor(dot, 2, 64);
  dot *= iv;                               // xh . d xh
  if (iv >= 1.f / NCE_EPS) dot = 0.f;      // |x| clamped at eps
  if (live) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c4 = k * 4 + cg;
      if (c4 < C4) {
        const float4 x = *reinterpret_cast<const float4*>(xr + c4 * 4);
        const float xs = iv * dot;
        *reinterpret_cast<float4*>(dX + rr * C + c4 * 4) = make_float4((acc[k].x - x.x * xs) * iv, (acc[k].y - x.y * xs) * iv,
                                                                       (acc[k].z - x.z * xs) * iv, (acc[k].w - x.w * xs) * iv);
      }
    }
  }
}

// scratch of a forward (kept for its backward), in floats: inv [2R] | S [frames L L] | rlse [R] | clse [R] | partial [frames],  R = frames * L

extern "C" int vptr_nce_fwd(const float* g, const float* p, float* scratch, float* loss_out, int frames, int L, int C, float temperature,
                            vptr_stream_t stream) {
  VPTR_CHECK(g && p && scratch && loss_out && frames > 0 && L > 0 && C > 0 && temperature > 0.f, "nce_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int R = frames * L;
  float* inv = scratch;
  float* S = inv + 2 * (int64_t)R;
  float* rlse = S + (int64_t)frames * L * L;
  float* clse = rlse + R;
  float* partial = clse + R;
  nce_invnorm_kernel<<<cdiv(2 * R, 4), 256, 0, st>>>(g, p, inv, R, C);
  VPTR_LAUNCH_CHECK();
  const int nb = cdiv(L, 64);
  nce_scores_kernel<<<frames * nb * nb, 256, 0, st>>>(g, p, inv, S, R, L, C, 1.f / temperature);
  VPTR_LAUNCH_CHECK();
  nce_stats_kernel<<<frames, 256, 0, st>>>(S, rlse, clse, partial, L);
  VPTR_LAUNCH_CHECK();
  nce_final_kernel<<<1, 256, 0, st>>>(partial, frames, 0.5 / (double)R, loss_out);
  VPTR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vptr_nce_bwd(const float* g, const float* p, const float* scratch, const float* gout, float* dg, float* dp, int frames, int L, int C,
                            float temperature, vptr_stream_t stream) {
  VPTR_CHECK(g && p && scratch && dg && dp && frames > 0 && L > 0 && C > 0 && temperature > 0.f, "nce_bwd: bad arguments");
  VPTR_CHECK(C <= 16 * 40, "nce_bwd: at most 640 channels (got %d)", C);
  VPTR_CHECK((C % 4 == 0) && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(dg) | reinterpret_cast<uintptr_t>(dp)) & 15) == 0,
             "nce_bwd: C %% 4 == 0 and 16-byte aligned tensors expected");
  hipStream_t st = (hipStream_t)stream;
  const int R = frames * L;
  const float* inv = scratch;
  const float* S = inv + 2 * (int64_t)R;
  const float* rlse = S + (int64_t)frames * L * L;
  const float* clse = rlse + R;
  const int nb = cdiv(L, 64);
  const int C4 = cdiv(C, 4);
  const size_t lds = (size_t)(64 * 17 + 16 * C4 * 4) * sizeof(float);
  const int grid = frames * nb * 2;
  const float cmean = 0.5f / (float)R;
#define NCE_BWD(NV)                                                                                                                   \
  do {                                                                                                                                \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nce_bwd_kernel<NV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    nce_bwd_kernel<NV><<<grid, 256, lds, st>>>(g, p, inv, S, rlse, clse, gout, dg, dp, R, L, C, 1.f / temperature, cmean);           \
  } while (0)
  if (C4 <= 16) NCE_BWD(4);
  else if (C4 <= 64) NCE_BWD(16);
  else if (C4 <= 136) NCE_BWD(34);
  else NCE_BWD(40);
#undef NCE_BWD
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Stochastic-depth scale vectors (VidHRFormer_modules.py:563-575: floor(keep + U) / keep per sample): every request r of a model
// forward (count[r] indices, keep probability keep[r]) in ONE launch, from the counter-based hash of the step's dropout seed --
// no torch generator in the step, so eager steps and hipGraph replays draw identical vectors.
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void droppath_kernel(const float* __restrict__ keep, float* __restrict__ out, int nreq, int maxcount,
                                                       const uint64_t* __restrict__ seed_dev, uint32_t site0) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nreq * maxcount) return;
  const int r = i / maxcount, k = i % maxcount;
  const float kp = keep[r];
  const uint32_t h = vptr_hash3(*seed_dev, site0 + (uint32_t)r, (uint64_t)k);
  const float u = (float)(h >> 8) * (1.0f / 16777216.0f);   // [0, 1) on a 24-bit grid, like torch.rand's float32
  out[i] = floorf(kp + u) / kp;
}

extern "C" int vptr_droppath_scales(const float* keep, float* out, int nreq, int maxcount, const uint64_t* seed_dev, uint32_t site0,
                                    vptr_stream_t stream) {
  VPTR_CHECK(keep && out && seed_dev && nreq > 0 && maxcount > 0, "droppath_scales: bad arguments");
  droppath_kernel<<<cdiv((int64_t)nreq * maxcount, 256), 256, 0, (hipStream_t)stream>>>(keep, out, nreq, maxcount, seed_dev, site0);
  VPTR_LAUNCH_CHECK();
  return 0;
}
