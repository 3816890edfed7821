// LayerNorm(C) forward/backward, row/column reductions and the conv-FFN normalisation kernels (gfx950).
// All kernels here are HBM-bound: float4 accesses, one wave per row for row-wise ops, thread-per-column sweeps with
// coalesced row reads for column reductions (partials combined with fp32 atomics).
#include "common.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim: one wave per row, 4 rows per 256-thread block.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ y2, const float* __restrict__ tab, int tab_div,
                                                     int tab_mod, float* __restrict__ mean, float* __restrict__ rstd,
                                                     int rows, int C, float eps, int p16) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * C;
  const int C4 = C >> 2;
  float s = 0.f;
  for (int i = lane; i < C4; i += 64) {
    const float4 v = reinterpret_cast<const float4*>(xr)[i];
    s += (v.x + v.y) + (v.z + v.w);
  }
  for (int i = (C4 << 2) + lane; i < C; i += 64) s += xr[i];
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int i = lane; i < C4; i += 64) {
    const float4 v = reinterpret_cast<const float4*>(xr)[i];
    const float a = v.x - mu, b = v.y - mu, c = v.z - mu, d = v.w - mu;
    q += (a * a + b * b) + (c * c + d * d);
  }
  for (int i = (C4 << 2) + lane; i < C; i += 64) { const float a = xr[i] - mu; q += a * a; }
  const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  float* yr = y + (int64_t)row * C;
  float* y2r = y2 ? y2 + (int64_t)row * C : nullptr;
  const float* tr = tab ? tab + (int64_t)((row / tab_div) % tab_mod) * C : nullptr;
  for (int i = lane; i < C4; i += 64) {
    const float4 v = reinterpret_cast<const float4*>(xr)[i];
    const float4 g = reinterpret_cast<const float4*>(gamma)[i];
    const float4 b = reinterpret_cast<const float4*>(beta)[i];
    float4 o;
    o.x = (v.x - mu) * rs * g.x + b.x; o.y = (v.y - mu) * rs * g.y + b.y;
    o.z = (v.z - mu) * rs * g.z + b.z; o.w = (v.w - mu) * rs * g.w + b.w;
    vptr_store4_fmt(y, (int64_t)row * C + 4 * i, o, p16);   // p16: the outputs only feed GEMMs (C % 16 == 0)
    if (y2r) {
      const float4 t = reinterpret_cast<const float4*>(tr)[i];
      o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
      vptr_store4_fmt(y2, (int64_t)row * C + 4 * i, o, p16);
    }
  }
  for (int i = (C4 << 2) + lane; i < C; i += 64) {
    const float o = (xr[i] - mu) * rs * gamma[i] + beta[i];
    yr[i] = o;
    if (y2r) y2r[i] = o + tr[i];
  }
}

// The same with the row held in registers between the three sweeps (NC4 float4 per lane; C <= 256 * NC4): one global read of x
// instead of three dependent ones (a wave has nothing else to hide its round trips behind).
// NR rows per wave (round 6): the rows' load -> reduce -> reduce -> store chains are independent, so a wave overlaps their round trips
// instead of sitting through one chain per row (10 240 x 528: one row per wave = 10 240 one-chain waves, 14 us for 43 MB).
template <int NC4, int NR>
__global__ __launch_bounds__(256) void ln_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y,
                                                         float* __restrict__ y2, const float* __restrict__ tab, int tab_div,
                                                         int tab_mod, float* __restrict__ mean, float* __restrict__ rstd,
                                                         int rows, int C, float eps, int p16) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * NR;
  if (row0 >= rows) return;
  const int C4 = C >> 2;
  float4 v[NR][NC4];
  float s[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)min(row0 + r, rows - 1) * C);
    s[r] = 0.f;
#pragma unroll
    for (int j = 0; j < NC4; ++j) {
      const int i = lane + 64 * j;
      v[r][j] = i < C4 ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      s[r] += (v[r][j].x + v[r][j].y) + (v[r][j].z + v[r][j].w);
    }
  }
  float mu[NR], rs[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) mu[r] = wave_sum(s[r]) / (float)C;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NC4; ++j) {
      if (lane + 64 * j < C4) {
        const float a = v[r][j].x - mu[r], b = v[r][j].y - mu[r], c = v[r][j].z - mu[r], d = v[r][j].w - mu[r];
        q += (a * a + b * b) + (c * c + d * d);
      }
    }
    s[r] = q;
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) rs[r] = rsqrtf(wave_sum(s[r]) / (float)C + eps);
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int row = row0 + r;
    if (row >= rows) break;
    if (lane == 0) { mean[row] = mu[r]; rstd[row] = rs[r]; }
    const float4* tr = tab ? reinterpret_cast<const float4*>(tab + (int64_t)((row / tab_div) % tab_mod) * C) : nullptr;
#pragma unroll
    for (int j = 0; j < NC4; ++j) {
      const int i = lane + 64 * j;
      if (i < C4) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[i];
        const float4 b = reinterpret_cast<const float4*>(beta)[i];
        float4 o;
        o.x = (v[r][j].x - mu[r]) * rs[r] * g.x + b.x; o.y = (v[r][j].y - mu[r]) * rs[r] * g.y + b.y;
        o.z = (v[r][j].z - mu[r]) * rs[r] * g.z + b.z; o.w = (v[r][j].w - mu[r]) * rs[r] * g.w + b.w;
        vptr_store4_fmt(y, (int64_t)row * C + 4 * i, o, p16);
        if (y2) {
          const float4 t = tr[i];
          o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
          vptr_store4_fmt(y2, (int64_t)row * C + 4 * i, o, p16);
        }
      }
    }
  }
}

extern "C" int vptr_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* y2,
                                  const float* tab, int tab_div, int tab_mod, float* mean, float* rstd, int rows, int C,
                                  float eps, int p16, vptr_stream_t stream) {
  if (p16) VPTR_CHECK(C % 16 == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(y2)) & 63) == 0,
                      "layernorm_fwd: P16 outputs need C %% 16 == 0 and 64-byte aligned y, y2");
  VPTR_CHECK(rows > 0 && C > 0, "layernorm_fwd: empty input");
  VPTR_CHECK(C % 4 == 0, "layernorm_fwd: C must be a multiple of 4 (got %d)", C);
  if (y2) VPTR_CHECK(tab && tab_div >= 1 && tab_mod >= 1, "layernorm_fwd: y2 needs tab, tab_div, tab_mod");
  const int C4 = C >> 2;
  static int ln_rows = 0;   // VPTR_LN_ROWS = 1 | 2 | 4 rows per wave for the register-resident kernels (default 2 from 4096 rows on)
  if (!ln_rows) { const char* e = getenv("VPTR_LN_ROWS"); ln_rows = (e && (atoi(e) == 1 || atoi(e) == 2 || atoi(e) == 4)) ? atoi(e) : 2; }
  const int nr = rows >= 4096 ? ln_rows : 1;
#define LN_FWD_GO(NC, NR) ln_fwd_reg_kernel<NC, NR><<<cdiv(rows, 4 * NR), 256, 0, (hipStream_t)stream>>>(x, gamma, beta, y, y2, y2 ? tab : nullptr, tab_div, tab_mod, mean, rstd, rows, C, eps, p16)
  if (C4 <= 64) {
    if (nr == 4) LN_FWD_GO(1, 4); else if (nr == 2) LN_FWD_GO(1, 2); else LN_FWD_GO(1, 1);
  } else if (C4 <= 192) {
    if (nr == 4) LN_FWD_GO(3, 4); else if (nr == 2) LN_FWD_GO(3, 2); else LN_FWD_GO(3, 1);
  }
#undef LN_FWD_GO
  else
    ln_fwd_kernel<<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>(x, gamma, beta, y, y2, y2 ? tab : nullptr, tab_div, tab_mod,
                                                                  mean, rstd, rows, C, eps, p16);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// dx: one wave per row.  g = dy + dy2;  dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat))
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ dy2,
                                                        const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        float* __restrict__ dx, int rows, int C,
                                                        const float* __restrict__ dx_add) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t off = (int64_t)row * C;
  const float mu = mean[row], rs = rstd[row];
  float s1 = 0.f, s2 = 0.f;
  for (int i = lane; i < C; i += 64) {
    float g = dy[off + i];
    if (dy2) g += dy2[off + i];
    const float gg = g * gamma[i];
    s1 += gg;
    s2 += gg * (x[off + i] - mu) * rs;
  }
  s1 = wave_sum(s1) / (float)C;
  s2 = wave_sum(s2) / (float)C;
  for (int i = lane; i < C; i += 64) {
    float g = dy[off + i];
    if (dy2) g += dy2[off + i];
    const float xh = (x[off + i] - mu) * rs;
    dx[off + i] = rs * (g * gamma[i] - s1 - xh * s2) + (dx_add ? dx_add[off + i] : 0.f);
  }
}

// dgamma/dbeta: thread per column, block sweeps a chunk of rows; coalesced across threads.
__global__ __launch_bounds__(256) void ln_bwd_param_kernel(const float* __restrict__ dy, const float* __restrict__ dy2,
                                                           const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int rows, int C, int rows_per_block) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float ag = 0.f, ab = 0.f;
  for (int r = r0; r < r1; ++r) {
    float g = dy[(int64_t)r * C + c];
    if (dy2) g += dy2[(int64_t)r * C + c];
    ag += g * (x[(int64_t)r * C + c] - mean[r]) * rstd[r];
    ab += g;
  }
  unsafeAtomicAdd(dgamma + c, ag);
  unsafeAtomicAdd(dbeta + c, ab);
}

// dx + dgamma/dbeta in ONE pass: one wave per row, the row held in registers as NC4 float4 per lane between the two
// reductions; the parameter gradients are accumulated per lane over the wave's rows, summed over the block's 4 waves in LDS,
// and leave the block as one atomic per column.  C % 4 == 0 and C <= 256 * NC4.
template <int NC4, int NW>   // NW waves per workgroup, each walking every NW-th row of the workgroup's rpb rows
__global__ __launch_bounds__(64 * NW) void ln_bwd_fused_kernel(const float* __restrict__ dy_, const float* __restrict__ dy2_,
                                                           const float* __restrict__ x_, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float* __restrict__ dx_, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int rows, int C, int rpb,
                                                           const float* __restrict__ dx_add_, float* __restrict__ part) {
  __shared__ float4 red[NW][2][NC4 * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int C4 = C >> 2;
  const int r0 = blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gam[NC4], ag[NC4], ab[NC4];
#pragma unroll
  for (int k = 0; k < NC4; ++k) {
    const int i = lane + 64 * k;
    gam[k] = i < C4 ? reinterpret_cast<const float4*>(gamma)[i] : z;
    ag[k] = z;
    ab[k] = z;
  }
  const float inv_c = 1.f / (float)C;
  // the operands of a wave's NEXT row are requested before the two reductions of the current one (round 6: a wave walks rpb / NW rows one
  // dependent load -> reduce -> store chain after the other, and 2 560 such waves are all a 10 240-row launch has)
  float4 gN[NC4], xN[NC4];
  float muN = 0.f, rsN = 0.f;
  auto fetch = [&](const int row) {
    muN = mean[row];
    rsN = rstd[row];
#pragma unroll
    for (int k = 0; k < NC4; ++k) {
      const int ic = min(lane + 64 * k, C4 - 1);
      gN[k] = reinterpret_cast<const float4*>(dy_)[(int64_t)row * C4 + ic];
      if (dy2_) {
        const float4 g2 = reinterpret_cast<const float4*>(dy2_)[(int64_t)row * C4 + ic];
        gN[k].x += g2.x; gN[k].y += g2.y; gN[k].z += g2.z; gN[k].w += g2.w;
      }
      xN[k] = reinterpret_cast<const float4*>(x_)[(int64_t)row * C4 + ic];
    }
  };
  if (r0 + wv < r1) fetch(r0 + wv);
  for (int row = r0 + wv; row < r1; row += NW) {
    float4* dx = reinterpret_cast<float4*>(dx_) + (int64_t)row * C4;
    const float mu = muN, rs = rsN;
    float4 g[NC4], xh[NC4], ra[NC4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NC4; ++k) {
      const int i = lane + 64 * k;
      const float m = i < C4 ? 1.f : 0.f;
      const float4 gv = gN[k];
      const float4 xv = xN[k];
      ra[k] = z;
      if (dx_add_) ra[k] = reinterpret_cast<const float4*>(dx_add_)[(int64_t)row * C4 + min(i, C4 - 1)];
      g[k] = make_float4(gv.x * m, gv.y * m, gv.z * m, gv.w * m);
      xh[k] = make_float4((xv.x - mu) * rs * m, (xv.y - mu) * rs * m, (xv.z - mu) * rs * m, (xv.w - mu) * rs * m);
      const float4 gg = make_float4(g[k].x * gam[k].x, g[k].y * gam[k].y, g[k].z * gam[k].z, g[k].w * gam[k].w);
      s1 += (gg.x + gg.y) + (gg.z + gg.w);
      s2 += (gg.x * xh[k].x + gg.y * xh[k].y) + (gg.z * xh[k].z + gg.w * xh[k].w);
    }
    if (row + NW < r1) fetch(row + NW);   // in flight under the two reductions and the stores below
    s1 = wave_sum(s1) * inv_c;
    s2 = wave_sum(s2) * inv_c;
#pragma unroll
    for (int k = 0; k < NC4; ++k) {
      const int i = lane + 64 * k;
      if (i < C4)
        dx[i] = make_float4(rs * (g[k].x * gam[k].x - s1 - xh[k].x * s2) + ra[k].x, rs * (g[k].y * gam[k].y - s1 - xh[k].y * s2) + ra[k].y,
                            rs * (g[k].z * gam[k].z - s1 - xh[k].z * s2) + ra[k].z, rs * (g[k].w * gam[k].w - s1 - xh[k].w * s2) + ra[k].w);
      ag[k].x += g[k].x * xh[k].x; ag[k].y += g[k].y * xh[k].y; ag[k].z += g[k].z * xh[k].z; ag[k].w += g[k].w * xh[k].w;
      ab[k].x += g[k].x; ab[k].y += g[k].y; ab[k].z += g[k].z; ab[k].w += g[k].w;
    }
  }
#pragma unroll
  for (int k = 0; k < NC4; ++k) {
    red[wv][0][k * 64 + lane] = ag[k];
    red[wv][1][k * 64 + lane] = ab[k];
  }
  __syncthreads();
  const float* rf = reinterpret_cast<const float*>(&red[0][0][0]);
  constexpr int WS = 2 * NC4 * 64 * 4, PS = NC4 * 64 * 4;  // floats per wave / per plane
  for (int i = threadIdx.x; i < C; i += 64 * NW) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      sg += rf[w * WS + i];
      sb += rf[w * WS + PS + i];
    }
    if (part) {   // deferred: this workgroup's sums as one row of [gridDim.x][2][C]; vptr_partial_reduce adds the rows later
      part[((int64_t)blockIdx.x * 2) * C + i] = sg;
      part[((int64_t)blockIdx.x * 2 + 1) * C + i] = sb;
    } else {
      unsafeAtomicAdd(dgamma + i, sg);
      unsafeAtomicAdd(dbeta + i, sb);
    }
  }
}

// rows per workgroup of the deferred launch for 256 < C <= 768 (the step's LayerNorm(528)): 16 = 4 waves x 4 rows, 640 workgroups of 10 240
// rows, three of them per CU -- all resident at once (round 6; VPTR_LN_BWD_RPB=32 restores 8 waves x 4 rows: 320 workgroups, ONE per CU at
// 140 VGPRs, i.e. 1.25 rounds)
static int ln_bwd_rpb() {
  static int v = 0;
  if (!v) { const char* e = getenv("VPTR_LN_BWD_RPB"); v = (e && atoi(e) == 32) ? 32 : ((e && atoi(e) == 8) ? 8 : 16); }
  return v;
}
// rows of partial sums a deferred backward call writes (0: this geometry has no deferred variant)
extern "C" int vptr_layernorm_bwd_partials(int rows, int C) {
  if (g_vptr_deterministic) return (C % 4 == 0 && C <= 1024 && rows > 0) ? cdiv(rows, 32) : 0;   // every vectorised geometry: no atomics at all
  if (rows < 4096 || C % 4 != 0 || C <= 256 || C > 768) return 0;
  return cdiv(rows, ln_bwd_rpb());
}
static int layernorm_bwd_impl(const float* dy, const float* dy2, const float* x, const float* gamma, const float* mean,
                              const float* rstd, float* dx, float* dgamma, float* dbeta, int rows, int C,
                              const float* dx_add, float* partials, hipStream_t st) {
  VPTR_CHECK(rows > 0 && C > 0, "layernorm_bwd: empty input");
  if (partials) {
    // deferred parameter gradients: no atomics, so more and shorter workgroups (32 rows each instead of 64) cost nothing
    VPTR_CHECK(dx && vptr_layernorm_bwd_partials(rows, C) > 0, "layernorm_bwd: no deferred variant for rows %d, C %d", rows, C);
    if (C <= 256) ln_bwd_fused_kernel<1, 4><<<cdiv(rows, 32), 256, 0, st>>>(dy, dy2, x, gamma, mean, rstd, dx, nullptr, nullptr, rows, C, 32, dx_add, partials);
    else if (C <= 768 && !g_vptr_deterministic && ln_bwd_rpb() <= 16) ln_bwd_fused_kernel<3, 4><<<cdiv(rows, ln_bwd_rpb()), 256, 0, st>>>(dy, dy2, x, gamma, mean, rstd, dx, nullptr, nullptr, rows, C, ln_bwd_rpb(), dx_add, partials);
    else if (C <= 768) ln_bwd_fused_kernel<3, 8><<<cdiv(rows, 32), 512, 0, st>>>(dy, dy2, x, gamma, mean, rstd, dx, nullptr, nullptr, rows, C, 32, dx_add, partials);
    else ln_bwd_fused_kernel<4, 4><<<cdiv(rows, 32), 256, 0, st>>>(dy, dy2, x, gamma, mean, rstd, dx, nullptr, nullptr, rows, C, 32, dx_add, partials);
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  if (dx && dgamma && dbeta && C % 4 == 0 && C <= 1024) {
    // fewer, longer workgroups: the per-column atomics at the end contend across workgroups.  Big inputs: 8 waves x 8 rows each = 64 rows
    // per workgroup (half the atomics of 4 waves x 8 rows at the same number of waves in flight: -0.25 ms per step; 16 waves or fewer rows lose)
    const bool big = rows >= 4096;
    const bool det = g_vptr_deterministic != 0;   // callers without an in-place destination (no partial buffer): ONE workgroup, one adder per column
    const int rpb = det ? rows : (big ? 64 : 4);
    const int nb = cdiv(rows, rpb);
    const int rpb2 = det ? rows : (big ? 32 : 4);
    if (C <= 256) ln_bwd_fused_kernel<1, 4><<<cdiv(rows, rpb2), 256, 0, st>>>(dy, dy2, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, rpb2, dx_add, nullptr);
    else if (C <= 768 && big) ln_bwd_fused_kernel<3, 8><<<nb, 512, 0, st>>>(dy, dy2, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, rpb, dx_add, nullptr);
    else if (C <= 768) ln_bwd_fused_kernel<3, 4><<<nb, 256, 0, st>>>(dy, dy2, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, rpb, dx_add, nullptr);
    else ln_bwd_fused_kernel<4, 4><<<cdiv(rows, rpb2), 256, 0, st>>>(dy, dy2, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, rpb2, dx_add, nullptr);
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  if (dx) ln_bwd_dx_kernel<<<cdiv(rows, 4), 256, 0, st>>>(dy, dy2, x, gamma, mean, rstd, dx, rows, C, dx_add);
  if (dgamma && dbeta) {
    const int rpb = g_vptr_deterministic ? rows : 64;
    dim3 grid(cdiv(C, 256), cdiv(rows, rpb));
    ln_bwd_param_kernel<<<grid, 256, 0, st>>>(dy, dy2, x, mean, rstd, dgamma, dbeta, rows, C, rpb);
  }
  VPTR_LAUNCH_CHECK();
  return 0;
}
extern "C" int vptr_layernorm_bwd(const float* dy, const float* dy2, const float* x, const float* gamma, const float* mean,
                                  const float* rstd, float* dx, float* dgamma, float* dbeta, int rows, int C,
                                  const float* dx_add, vptr_stream_t stream) {
  return layernorm_bwd_impl(dy, dy2, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, dx_add, nullptr, (hipStream_t)stream);
}
extern "C" int vptr_layernorm_bwd_deferred(const float* dy, const float* dy2, const float* x, const float* gamma, const float* mean,
                                           const float* rstd, float* dx, int rows, int C, const float* dx_add, float* partials,
                                           vptr_stream_t stream) {
  VPTR_CHECK(partials && (reinterpret_cast<uintptr_t>(partials) & 15) == 0, "layernorm_bwd_deferred: needs a 16-byte aligned partial-sum buffer");
  return layernorm_bwd_impl(dy, dy2, x, gamma, mean, rstd, dx, nullptr, nullptr, rows, C, dx_add, partials, (hipStream_t)stream);
}
// Deferred parameter-gradient sums of a whole backward pass in ONE launch: entry e adds the nparts rows of part[nparts][2][C] into
// dst0[C] (row 0 of each pair) and dst1[C] (row 1).  The final add is an atomic: two entries may name the same destination (a module
// applied twice in one forward).  Workgroup = 64 float4 columns x 16 row lanes.
__global__ __launch_bounds__(1024) void partial_reduce_kernel(const vptr_reduce_entry* __restrict__ tab, const int unique_dst) {
  __shared__ float4 red[2][16][64];
  const vptr_reduce_entry e = tab[blockIdx.y];
  const int C4 = e.C >> 2;                      // C % 4 == 0 (checked on the host side of the table)
  const int l = threadIdx.x & 63, q = threadIdx.x >> 6, c4 = blockIdx.x * 64 + l;
  if (blockIdx.x * 64 >= C4) return;            // (workgroup-uniform: the grid is sized for the widest entry)
  const int rpp = e.dst1 ? 2 : 1;               // rows per part: [nparts][2][C] with two destinations, [nparts][1][C] with one
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  if (c4 < C4) {
    const float4* part = reinterpret_cast<const float4*>(e.part);
    int p = q;
    for (; p + 48 < e.nparts; p += 64) {        // 8 independent 16-byte loads in flight
      float4 t0[4], t1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        t0[u] = part[((int64_t)(p + 16 * u) * rpp) * C4 + c4];
        t1[u] = part[((int64_t)(p + 16 * u) * rpp + rpp - 1) * C4 + c4];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0.x += t0[u].x; a0.y += t0[u].y; a0.z += t0[u].z; a0.w += t0[u].w;
        a1.x += t1[u].x; a1.y += t1[u].y; a1.z += t1[u].z; a1.w += t1[u].w;
      }
    }
    for (; p < e.nparts; p += 16) {
      const float4 t0 = part[((int64_t)p * rpp) * C4 + c4], t1 = part[((int64_t)p * rpp + rpp - 1) * C4 + c4];
      a0.x += t0.x; a0.y += t0.y; a0.z += t0.z; a0.w += t0.w;
      a1.x += t1.x; a1.y += t1.y; a1.z += t1.z; a1.w += t1.w;
    }
  }
  red[0][q][l] = a0;
  red[1][q][l] = a1;
  __syncthreads();
  if (q < rpp && c4 < C4) {                     // row lane 0 finishes dst0, row lane 1 dst1
    float4 sum = red[q][0][l];
#pragma unroll
    for (int u = 1; u < 16; ++u) {
      const float4 t = red[q][u][l];
      sum.x += t.x; sum.y += t.y; sum.z += t.z; sum.w += t.w;
    }
    float* dst = (q ? e.dst1 : e.dst0) + (int64_t)c4 * 4;
    if (unique_dst) {   // no other entry of this launch (and nothing else in flight) writes this destination: plain read-add-write
      if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        float4 d = *reinterpret_cast<float4*>(dst);
        d.x += sum.x; d.y += sum.y; d.z += sum.z; d.w += sum.w;
        *reinterpret_cast<float4*>(dst) = d;
      } else {
        dst[0] += sum.x; dst[1] += sum.y; dst[2] += sum.z; dst[3] += sum.w;
      }
    } else {
      unsafeAtomicAdd(dst + 0, sum.x); unsafeAtomicAdd(dst + 1, sum.y);
      unsafeAtomicAdd(dst + 2, sum.z); unsafeAtomicAdd(dst + 3, sum.w);
    }
  }
}
extern "C" int vptr_partial_reduce(const vptr_reduce_entry* table_dev, int count, int max_C, int unique_dst, vptr_stream_t stream) {
  VPTR_CHECK(table_dev && count > 0 && max_C > 0 && max_C % 4 == 0, "partial_reduce: bad arguments (every C must be a multiple of 4)");
  partial_reduce_kernel<<<dim3(cdiv(max_C / 4, 64), count), 1024, 0, (hipStream_t)stream>>>(table_dev, unique_dst);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// small reductions / broadcasts
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowmod_sum_kernel(const float* __restrict__ src, float* __restrict__ out, int rows,
                                                         int C, int div, int mod, int groups_per_block) {
  // out row j = sum over all rows r with (r / div) % mod == j.  Rows come in runs of `div` rows with the same j,
  // repeating with period div*mod.  Thread per column; blockIdx.y = j; blockIdx.z = chunk of periods.
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int j = blockIdx.y;
  const int period = div * mod;
  const int nper = (rows + period - 1) / period;
  const int p0 = blockIdx.z * groups_per_block, p1 = min(nper, p0 + groups_per_block);
  float a = 0.f;
  int p = p0;
  if (div == 1) {   // one row per period: the periods are the independent loads
    for (; p + 3 < p1 && (p + 3) * period + j < rows; p += 4) {
      const float v0 = src[(int64_t)(p * period + j) * C + c], v1 = src[(int64_t)((p + 1) * period + j) * C + c];
      const float v2 = src[(int64_t)((p + 2) * period + j) * C + c], v3 = src[(int64_t)((p + 3) * period + j) * C + c];
      a += (v0 + v1) + (v2 + v3);
    }
  }
  for (; p < p1; ++p) {
    const int rbase = p * period + j * div;
    int d = 0;
    for (; d + 3 < div && rbase + d + 3 < rows; d += 4) {   // four independent loads in flight
      const float v0 = src[(int64_t)(rbase + d) * C + c], v1 = src[(int64_t)(rbase + d + 1) * C + c];
      const float v2 = src[(int64_t)(rbase + d + 2) * C + c], v3 = src[(int64_t)(rbase + d + 3) * C + c];
      a += (v0 + v1) + (v2 + v3);
    }
    for (; d < div; ++d) {
      const int r = rbase + d;
      if (r < rows) a += src[(int64_t)r * C + c];
    }
  }
  unsafeAtomicAdd(out + (int64_t)j * C + c, a);
}

extern "C" int vptr_rowmod_sum(const float* src, float* out, int rows, int C, int div, int mod, vptr_stream_t stream) {
  VPTR_CHECK(rows > 0 && C > 0 && div >= 1 && mod >= 1, "rowmod_sum: bad arguments");
  const int period = div * mod;
  const int nper = (rows + period - 1) / period;
  const int gpb = g_vptr_deterministic ? nper : 8;   // deterministic: one workgroup (one adder) per output element
  dim3 grid(cdiv(C, 256), mod, cdiv(nper, gpb));
  rowmod_sum_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, out, rows, C, div, mod, gpb);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// column sums: block = 32 float4 columns x 8 row lanes over a chunk of 256 rows; 32 independent float4 loads per thread,
// LDS reduction over the row lanes, one atomic per column per block.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ src, float* __restrict__ out, int rows, int C4) {
  __shared__ float4 red[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c4 = blockIdx.x * 32 + tx;
  const int r0 = blockIdx.y * 256, r1 = min(rows, r0 + 256);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < C4) {
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += 8) {
      const float4 v = reinterpret_cast<const float4*>(src)[(int64_t)r * C4 + c4];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && c4 < C4) {
#pragma unroll
    for (int k = 1; k < 8; ++k) { const float4 v = red[k][tx]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    unsafeAtomicAdd(out + c4 * 4 + 0, a.x);
    unsafeAtomicAdd(out + c4 * 4 + 1, a.y);
    unsafeAtomicAdd(out + c4 * 4 + 2, a.z);
    unsafeAtomicAdd(out + c4 * 4 + 3, a.w);
  }
}

extern "C" int vptr_colsum(const float* src, float* out, int rows, int C, vptr_stream_t stream) {
  VPTR_CHECK(rows > 0 && C > 0, "colsum: empty input");
  if (g_vptr_deterministic) {   // one thread walks a whole column: one adder per destination
    rowmod_sum_kernel<<<dim3(cdiv(C, 256), 1, 1), 256, 0, (hipStream_t)stream>>>(src, out, rows, C, rows, 1, 1);
  } else if (C % 4 == 0) {
    colsum_kernel<<<dim3(cdiv(C / 4, 32), cdiv(rows, 256)), 256, 0, (hipStream_t)stream>>>(src, out, rows, C / 4);
  } else {  // odd widths: one output row of rowmod_sum, runs of 64 rows per block
    dim3 grid(cdiv(C, 256), 1, cdiv(rows, 64));
    rowmod_sum_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, out, rows, C, 64, 1, 1);
  }
  VPTR_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void add_rowtab_kernel(const float* __restrict__ x, const float* __restrict__ tab,
                                                         float* __restrict__ y, int rows, int C4, int div, int mod) {
  const int64_t total = (int64_t)rows * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / C4), c4 = (int)(i - (int64_t)row * C4);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 t = reinterpret_cast<const float4*>(tab)[(int64_t)((row / div) % mod) * C4 + c4];
    reinterpret_cast<float4*>(y)[i] = make_float4(v.x + t.x, v.y + t.y, v.z + t.z, v.w + t.w);
  }
}

extern "C" int vptr_add_rowtab(const float* x, const float* tab, float* y, int rows, int C, int div, int mod,
                               vptr_stream_t stream) {
  VPTR_CHECK(rows > 0 && C > 0 && C % 4 == 0 && div >= 1 && mod >= 1, "add_rowtab: bad arguments");
  const int64_t total = (int64_t)rows * (C / 4);
  const int blocks = (int)hmin64((total + 255) / 256, 4096);
  add_rowtab_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(x, tab, y, rows, C / 4, div, mod);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// conv-FFN statistics.  colstats: per-channel mean / biased variance over all rows (BatchNorm2d batch stats).
// Pass 1: each block reduces 256 rows per column to (mean, M2); pass 2 merges the partials with Chan's formula.
// ---------------------------------------------------------------------------------------------------------------
// block = 32 float4 columns x 8 row lanes over a chunk of 256 rows, one pass: sums of (x - pivot) and (x - pivot)^2 with the
// chunk's first row as pivot (keeps the one-pass variance well conditioned), LDS reduction over the row lanes.
__global__ __launch_bounds__(256) void colstats_partial_kernel(const float* __restrict__ x, float* __restrict__ scratch,
                                                               int rows, int F4) {
  __shared__ float4 rs[8][32], rq[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c4 = blockIdx.x * 32 + tx;
  const int r0 = blockIdx.y * 256, r1 = min(rows, r0 + 256);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s, pv = s;
  if (c4 < F4) {
    pv = reinterpret_cast<const float4*>(x)[(int64_t)r0 * F4 + c4];
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += 8) {
      const float4 v = reinterpret_cast<const float4*>(x)[(int64_t)r * F4 + c4];
      const float a = v.x - pv.x, b = v.y - pv.y, c = v.z - pv.z, d = v.w - pv.w;
      s.x += a; s.y += b; s.z += c; s.w += d;
      q.x += a * a; q.y += b * b; q.z += c * c; q.w += d * d;
    }
  }
  rs[ty][tx] = s;
  rq[ty][tx] = q;
  __syncthreads();
  if (ty == 0 && c4 < F4) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const float4 u = rs[k][tx], w = rq[k][tx];
      s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
      q.x += w.x; q.y += w.y; q.z += w.z; q.w += w.w;
    }
    const float n = (float)(r1 - r0);
    float* o = scratch + ((int64_t)blockIdx.y * F4 * 4 + c4 * 4) * 2;
    o[0] = pv.x + s.x / n; o[1] = q.x - s.x * s.x / n;
    o[2] = pv.y + s.y / n; o[3] = q.y - s.y * s.y / n;
    o[4] = pv.z + s.z / n; o[5] = q.z - s.z * s.z / n;
    o[6] = pv.w + s.w / n; o[7] = q.w - s.w * s.w / n;
  }
}
__global__ __launch_bounds__(256) void colstats_final_kernel(const float* __restrict__ scratch, float* __restrict__ mean,
                                                             float* __restrict__ var, float* __restrict__ rstd, float eps,
                                                             int rows, int F, int nchunk, float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, float momentum,
                                                             long long* __restrict__ num_batches_tracked) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= F) return;
  float n = 0.f, mu = 0.f, m2 = 0.f;
  const float2* sc2 = reinterpret_cast<const float2*>(scratch);
  for (int k0 = 0; k0 < nchunk; k0 += 8) {   // 8 chunk records in flight, then the (sequential) Chan merges: the merge chain no longer
    float2 rec[8];                            // waits out a load round trip per chunk (40 chunks: 15 us -> a few)
#pragma unroll
    for (int u = 0; u < 8; ++u) rec[u] = sc2[(int64_t)min(k0 + u, nchunk - 1) * F + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u;
      if (k < nchunk) {
        const float nb = (float)min(256, rows - k * 256);
        const float d = rec[u].x - mu, nt = n + nb;
        mu += d * nb / nt;
        m2 += rec[u].y + d * d * n * nb / nt;
        n = nt;
      }
    }
  }
  mean[c] = mu;
  var[c] = m2 / n;
  if (rstd) rstd[c] = rsqrtf(m2 / n + eps);
  if (running_mean) {   // BatchNorm2d's train-mode bookkeeping (momentum update, unbiased variance) in the same launch
    running_mean[c] = running_mean[c] * (1.f - momentum) + mu * momentum;
    running_var[c] = running_var[c] * (1.f - momentum) + (m2 / n) * (momentum * n / fmaxf(n - 1.f, 1.f));
  }
  if (num_batches_tracked && c == 0) *num_batches_tracked += 1;
}

extern "C" int vptr_colstats(const float* x, float* mean, float* var, float* rstd, float eps, float* scratch, int rows, int F,
                             vptr_stream_t stream) {
  VPTR_CHECK(rows > 0 && F > 0 && F % 4 == 0 && scratch, "colstats: bad arguments (F must be a multiple of 4)");
  const int nchunk = cdiv(rows, 256);
  hipStream_t st = (hipStream_t)stream;
  colstats_partial_kernel<<<dim3(cdiv(F / 4, 32), nchunk), 256, 0, st>>>(x, scratch, rows, F / 4);
  colstats_final_kernel<<<cdiv(F, 256), 256, 0, st>>>(scratch, mean, var, rstd, eps, rows, F, nchunk, nullptr, nullptr, 0.f, nullptr);
  VPTR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vptr_colstats_running(const float* x, float* mean, float* var, float* rstd, float eps, float* scratch, int rows, int F,
                                     float* running_mean, float* running_var, float momentum, long long* num_batches_tracked,
                                     vptr_stream_t stream) {
  VPTR_CHECK(rows > 0 && F > 0 && F % 4 == 0 && scratch, "colstats: bad arguments (F must be a multiple of 4)");
  VPTR_CHECK((running_mean == nullptr) == (running_var == nullptr), "colstats_running: running_mean and running_var go together");
  const int nchunk = cdiv(rows, 256);
  hipStream_t st = (hipStream_t)stream;
  colstats_partial_kernel<<<dim3(cdiv(F / 4, 32), nchunk), 256, 0, st>>>(x, scratch, rows, F / 4);
  colstats_final_kernel<<<cdiv(F, 256), 256, 0, st>>>(scratch, mean, var, rstd, eps, rows, F, nchunk, running_mean, running_var, momentum,
                                                      num_batches_tracked);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// groupstats: mean / biased variance of each contiguous group of `group_elems` floats (LayerNorm((F,H,W)) per frame).
__global__ __launch_bounds__(1024) void groupstats_kernel(const float* __restrict__ x, float* __restrict__ mean,
                                                          float* __restrict__ var, float* __restrict__ rstd, float eps,
                                                          int group_elems) {
  // one pass: sums of (x - pivot) and (x - pivot)^2 with the group's first element as pivot
  __shared__ float red[16];
  const float* g = x + (int64_t)blockIdx.x * group_elems;
  const int n4 = group_elems >> 2;
  const float pv = g[0];
  float s = 0.f, q = 0.f;
#pragma unroll 4
  for (int i = threadIdx.x; i < n4; i += 1024) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    const float a = v.x - pv, b = v.y - pv, c = v.z - pv, d = v.w - pv;
    s += (a + b) + (c + d);
    q += (a * a + b * b) + (c * c + d * d);
  }
  for (int i = (n4 << 2) + threadIdx.x; i < group_elems; i += 1024) { const float a = g[i] - pv; s += a; q += a * a; }
  const float S = block_sum(s, red), Q = block_sum(q, red);
  if (threadIdx.x == 0) {
    const float n = (float)group_elems, ms = S / n;
    const float vv = fmaxf(Q / n - ms * ms, 0.f);
    mean[blockIdx.x] = pv + ms;
    var[blockIdx.x] = vv;
    if (rstd) rstd[blockIdx.x] = rsqrtf(vv + eps);
  }
}

extern "C" int vptr_groupstats(const float* x, float* mean, float* var, float* rstd, float eps, int groups, int group_elems,
                               vptr_stream_t stream) {
  VPTR_CHECK(groups > 0 && group_elems > 0 && group_elems % 4 == 0, "groupstats: bad arguments");
  groupstats_kernel<<<groups, 1024, 0, (hipStream_t)stream>>>(x, mean, var, rstd, eps, group_elems);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// y = act((x - mean) * rstd * w + b) [* dropout]; stats per column (BN) or per frame (LN over (F,H,W)); affine is
// [F] (per_col) or channel-last [HW, F].
// ---------------------------------------------------------------------------------------------------------------
template <bool PER_COL>
__global__ __launch_bounds__(256) void norm_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ y, int rows,
                                                           int F4, int HW, int act, float p, const uint64_t* seed_dev,
                                                           uint32_t site, const float* __restrict__ rowscale, int rs_div,
                                                           int rs_mod, const float* __restrict__ residual, int p16,
                                                           const float* __restrict__ raw_stats, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, float eps, int guard) {
  __shared__ float gred[16];
  const int64_t total = (int64_t)rows * F4;
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const float inv_n = 1.f / ((float)HW * (float)(F4 * 4));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / F4), c4 = (int)(i - (int64_t)row * F4);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 mu, rs, ww, bb;
    if (PER_COL) {
      mu = reinterpret_cast<const float4*>(mean)[c4];
      rs = reinterpret_cast<const float4*>(rstd)[c4];
      ww = reinterpret_cast<const float4*>(w)[c4];
      bb = reinterpret_cast<const float4*>(b)[c4];
    } else {
      const int f = row / HW, hw = row - f * HW;
      float m, r;
      if (raw_stats) {   // per-frame sum / sum of squares accumulated by the PRODUCER's epilogue (vptr_gemm frame_stats, vptr_dwconv3x3_fwd)
        m = raw_stats[VPTR_FRAME_STATS_STRIDE * f] * inv_n;
        const float e2 = raw_stats[VPTR_FRAME_STATS_STRIDE * f + 1] * inv_n;
        float var = fmaxf(e2 - m * m, 0.f);
        // E[x^2] - mean^2 from fp32 sums loses log2(E[x^2] / var) bits.  |mean| > ~30 std: recompute the frame's variance around its
        // mean (exact two-pass; this workgroup reads the whole frame -- every workgroup of the frame finds the same value).  guard:
        // a workgroup iteration lies inside ONE frame (HW * F4 % 256 == 0, checked by the launcher), so the branch is uniform.
        if (guard && var < 1e-3f * e2) {
          const float4* xf = reinterpret_cast<const float4*>(x) + (int64_t)f * HW * F4;
          float sq = 0.f, s1 = 0.f;   // around the approximate mean m: both sums are small, nothing cancels
          for (int j = threadIdx.x; j < HW * F4; j += 256) {
            const float4 t = xf[j];
            const float a = t.x - m, b2 = t.y - m, c = t.z - m, d = t.w - m;
            s1 += (a + b2) + (c + d);
            sq += (a * a + b2 * b2) + (c * c + d * d);
          }
          const float dm = block_sum(s1, gred) * inv_n;   // the mean of 135 k fp32 atomics is itself off by a fraction of such a std
          var = fmaxf(block_sum(sq, gred) * inv_n - dm * dm, 0.f);
          m += dm;
        }
        r = rsqrtf(var + eps);
        if (hw == 0 && c4 == 0) { mean_out[f] = m; rstd_out[f] = r; }   // kept for the backward pass
      } else {
        m = mean[f];
        r = rstd[f];
      }
      mu = make_float4(m, m, m, m);
      rs = make_float4(r, r, r, r);
      ww = reinterpret_cast<const float4*>(w)[(int64_t)hw * F4 + c4];
      bb = reinterpret_cast<const float4*>(b)[(int64_t)hw * F4 + c4];
    }
    float4 o;
    o.x = vptr_act((v.x - mu.x) * rs.x * ww.x + bb.x, act);
    o.y = vptr_act((v.y - mu.y) * rs.y * ww.y + bb.y, act);
    o.z = vptr_act((v.z - mu.z) * rs.z * ww.z + bb.z, act);
    o.w = vptr_act((v.w - mu.w) * rs.w * ww.w + bb.w, act);
    if (p > 0.f) {
      o.x *= vptr_drop_scale(seed, site, (uint64_t)i * 4 + 0, p);
      o.y *= vptr_drop_scale(seed, site, (uint64_t)i * 4 + 1, p);
      o.z *= vptr_drop_scale(seed, site, (uint64_t)i * 4 + 2, p);
      o.w *= vptr_drop_scale(seed, site, (uint64_t)i * 4 + 3, p);
    }
    if (rowscale) {
      const float r = rowscale[(row / rs_div) % rs_mod];
      o.x *= r; o.y *= r; o.z *= r; o.w *= r;
    }
    if (residual) {
      const float4 rv = reinterpret_cast<const float4*>(residual)[i];
      o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
    }
    vptr_store4_fmt(y, i * 4, o, p16);
  }
}

// LayerNorm((F,H,W)) mode, position-major: a thread owns ONE (h, w, channel quad) position and walks frames (blockIdx.y, stride gridDim.y),
// so its two affine float4s are loaded once instead of once per element (the row-major loop above re-reads the 2 x 0.54 MB tables for every
// frame: as many L2 requests again as the tensor itself; 39.8 us against 31.1 for the per-column mode at the same bytes).  Same arithmetic,
// same dropout sites (the flat element index), same variance guard (a workgroup still lies inside one frame).  VPTR_NORM_POS=0: row-major.
__global__ __launch_bounds__(256) void norm_act_fwd_pos_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ w,
                                                               const float* __restrict__ b, float* __restrict__ y, int frames,
                                                               int F4, int HW, int act, float p, const uint64_t* seed_dev,
                                                               uint32_t site, const float* __restrict__ rowscale, int rs_div,
                                                               int rs_mod, const float* __restrict__ residual, int p16,
                                                               const float* __restrict__ raw_stats, float* __restrict__ mean_out,
                                                               float* __restrict__ rstd_out, float eps, int guard) {
  __shared__ float gred[16];
  const int P = HW * F4;
  const int pos = blockIdx.x * 256 + threadIdx.x;
  const bool live = pos < P;
  const int posc = live ? pos : P - 1;
  const int hw = posc / F4;
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const float inv_n = 1.f / ((float)HW * (float)(F4 * 4));
  const float4 ww = reinterpret_cast<const float4*>(w)[posc], bb = reinterpret_cast<const float4*>(b)[posc];
  for (int f = blockIdx.y; f < frames; f += gridDim.y) {
    const int64_t i = (int64_t)f * P + posc;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float m, r;
    if (raw_stats) {
      m = raw_stats[VPTR_FRAME_STATS_STRIDE * f] * inv_n;
      const float e2 = raw_stats[VPTR_FRAME_STATS_STRIDE * f + 1] * inv_n;
      float var = fmaxf(e2 - m * m, 0.f);
      if (guard && var < 1e-3f * e2) {   // see norm_act_fwd_kernel; guard implies P % 256 == 0: every thread of the workgroup is live
        const float4* xf = reinterpret_cast<const float4*>(x) + (int64_t)f * P;
        float sq = 0.f, s1 = 0.f;
        for (int j = threadIdx.x; j < P; j += 256) {
          const float4 t = xf[j];
          const float a = t.x - m, b2 = t.y - m, c = t.z - m, d = t.w - m;
          s1 += (a + b2) + (c + d);
          sq += (a * a + b2 * b2) + (c * c + d * d);
        }
        const float dm = block_sum(s1, gred) * inv_n;
        var = fmaxf(block_sum(sq, gred) * inv_n - dm * dm, 0.f);
        m += dm;
      }
      r = rsqrtf(var + eps);
      if (pos == 0) { mean_out[f] = m; rstd_out[f] = r; }
    } else {
      m = mean[f];
      r = rstd[f];
    }
    if (!live) continue;
    float4 o;
    o.x = vptr_act((v.x - m) * r * ww.x + bb.x, act);
    o.y = vptr_act((v.y - m) * r * ww.y + bb.y, act);
    o.z = vptr_act((v.z - m) * r * ww.z + bb.z, act);
    o.w = vptr_act((v.w - m) * r * ww.w + bb.w, act);
    if (p > 0.f) {
      o.x *= vptr_drop_scale(seed, site, (uint64_t)i * 4 + 0, p);
      o.y *= vptr_drop_scale(seed, site, (uint64_t)i * 4 + 1, p);
      o.z *= vptr_drop_scale(seed, site, (uint64_t)i * 4 + 2, p);
      o.w *= vptr_drop_scale(seed, site, (uint64_t)i * 4 + 3, p);
    }
    if (rowscale) {
      const float rr = rowscale[((f * HW + hw) / rs_div) % rs_mod];
      o.x *= rr; o.y *= rr; o.z *= rr; o.w *= rr;
    }
    if (residual) {
      const float4 rv = reinterpret_cast<const float4*>(residual)[i];
      o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
    }
    vptr_store4_fmt(y, i * 4, o, p16);
  }
}
static int norm_pos_mode() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VPTR_NORM_POS");
    v = e ? atoi(e) : 1;
  }
  return v;
}

extern "C" int vptr_norm_act_fwd(const float* x, float* mean, float* rstd, const float* w, const float* b,
                                 float* y, int rows, int F, int HW, int per_col, int act, float dropout_p,
                                 const uint64_t* seed_dev, uint32_t site, const float* rowscale, int rs_div, int rs_mod,
                                 const float* residual, int p16, const float* raw_stats, float eps, vptr_stream_t stream) {
  if (raw_stats) VPTR_CHECK(!per_col && mean && rstd, "norm_act_fwd: raw_stats (per-frame sums) belong to the LayerNorm((F,H,W)) mode and need mean / rstd outputs");
  VPTR_CHECK(rows > 0 && F > 0 && F % 4 == 0 && HW >= 1, "norm_act_fwd: bad arguments");
  if (p16) VPTR_CHECK(F % 16 == 0 && (reinterpret_cast<uintptr_t>(y) & 63) == 0, "norm_act_fwd: a P16 output needs F %% 16 == 0 and a 64-byte aligned y");
  if (!per_col) VPTR_CHECK(rows % HW == 0, "norm_act_fwd: rows must be a multiple of HW");
  if (dropout_p > 0.f) VPTR_CHECK(seed_dev && dropout_p < 1.f, "norm_act_fwd: dropout needs seed_dev");
  const int64_t total = (int64_t)rows * (F / 4);
  const int blocks = (int)hmin64((total + 255) / 256, 8192);
  hipStream_t st = (hipStream_t)stream;
  if (rowscale) VPTR_CHECK(rs_div >= 1 && rs_mod >= 1, "norm_act_fwd: rowscale needs rs_div, rs_mod >= 1");
  if (per_col) norm_act_fwd_kernel<true><<<blocks, 256, 0, st>>>(x, mean, rstd, w, b, y, rows, F / 4, HW, act, dropout_p, seed_dev, site, rowscale, rs_div, rs_mod, residual, p16, nullptr, nullptr, nullptr, eps, 0);
  else if (norm_pos_mode() && rows / HW >= 16 && total >= (1 << 18)) {   // big inputs: position-major (affine tables read once per thread)
    const int frames = rows / HW, P = HW * (F / 4);
    norm_act_fwd_pos_kernel<<<dim3(cdiv(P, 256), (frames / 4 < 1 ? 1 : (frames / 4 > 65535 ? 65535 : frames / 4))), 256, 0, st>>>(
        x, raw_stats ? nullptr : mean, raw_stats ? nullptr : rstd, w, b, y, frames, F / 4, HW, act, dropout_p, seed_dev, site, rowscale, rs_div, rs_mod,
        residual, p16, raw_stats, mean, rstd, eps, (int)(raw_stats && P % 256 == 0));
  } else norm_act_fwd_kernel<false><<<blocks, 256, 0, st>>>(x, raw_stats ? nullptr : mean, raw_stats ? nullptr : rstd, w, b, y, rows, F / 4, HW, act, dropout_p, seed_dev, site, rowscale, rs_div, rs_mod, residual, p16, raw_stats, mean, rstd, eps,
                                                          (int)(raw_stats && ((int64_t)HW * (F / 4)) % 256 == 0 && total % 256 == 0));
  VPTR_LAUNCH_CHECK();
  return 0;
}

// backward helper: g = dy * drop * act'(z), z = xhat*w + b
__device__ __forceinline__ float norm_act_g(float dy, float xh, float w, float b, int act, float dscale) {
  const float z = xh * w + b;
  float g = dy * dscale;
  if (act == VPTR_ACT_GELU) g *= vptr_gelu_grad(z);
  else if (act == VPTR_ACT_RELU) g = z > 0.f ? g : 0.f;
  else if (act == VPTR_ACT_LRELU) g = z > 0.f ? g : 0.2f * g;
  return g;
}

// phase 1, per-column statistics (BN): dw[c] += sum g*xhat, db[c] += sum g.  (s1 = w*db, s2 = w*dw afterwards.)
__global__ __launch_bounds__(256) void norm_act_bwd_col_reduce(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ w, const float* __restrict__ b,
                                                               float* __restrict__ acc /* [2,F] */, int rows, int F, int act,
                                                               float p, const uint64_t* seed_dev, uint32_t site, int rpb,
                                                               const float* __restrict__ rowscale, int rs_div, int rs_mod) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= F) return;
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const int r0 = blockIdx.y * rpb, r1 = min(rows, r0 + rpb);
  const float mu = mean[c], rs = rstd[c], ww = w[c], bb = b[c];
  float aw = 0.f, ab = 0.f;
  for (int r = r0; r < r1; ++r) {
    const int64_t i = (int64_t)r * F + c;
    const float xh = (x[i] - mu) * rs;
    float ds = p > 0.f ? vptr_drop_scale(seed, site, (uint64_t)i, p) : 1.f;
    if (rowscale) ds *= rowscale[(r / rs_div) % rs_mod];
    const float g = norm_act_g(dy[i], xh, ww, bb, act, ds);
    aw += g * xh;
    ab += g;
  }
  unsafeAtomicAdd(acc + c, aw);
  unsafeAtomicAdd(acc + F + c, ab);
}
// the same for F % 4 == 0: 64 float4 columns x 4 row lanes per workgroup, 16-byte loads, four rows in flight per thread (the scalar
// version above walks 64 rows with two dependent 4-byte loads each: 52 us against 35 us of bytes at the step's shape)
__global__ __launch_bounds__(256) void norm_act_bwd_col_reduce4(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ w, const float* __restrict__ b,
                                                                float* __restrict__ acc /* [2,F] */, int rows, int F4, int act,
                                                                float p, const uint64_t* seed_dev, uint32_t site, int rpb,
                                                                const float* __restrict__ rowscale, int rs_div, int rs_mod) {
  __shared__ float4 red[2][3][64];
  const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c4 = blockIdx.x * 64 + l;
  const bool live = c4 < F4;
  const int cc = live ? c4 : F4 - 1;
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const int r0 = blockIdx.y * rpb, r1 = min(rows, r0 + rpb);
  const float4 mu4 = reinterpret_cast<const float4*>(mean)[cc], rs4 = reinterpret_cast<const float4*>(rstd)[cc];
  const float4 w4 = reinterpret_cast<const float4*>(w)[cc], b4 = reinterpret_cast<const float4*>(b)[cc];
  const float mus[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, rss[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
  const float wss[4] = {w4.x, w4.y, w4.z, w4.w}, bss[4] = {b4.x, b4.y, b4.z, b4.w};
  float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  auto one = [&](const int r, const float4 d, const float4 xv) {
    const int64_t i = ((int64_t)r * F4 + cc) * 4;
    const float rsc = rowscale ? rowscale[(r / rs_div) % rs_mod] : 1.f;
    const float dv[4] = {d.x, d.y, d.z, d.w}, xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float xh = (xs[u] - mus[u]) * rss[u];
      const float ds = (p > 0.f ? vptr_drop_scale(seed, site, (uint64_t)(i + u), p) : 1.f) * rsc;
      const float g = norm_act_g(dv[u], xh, wss[u], bss[u], act, ds);
      aw[u] += g * xh;
      ab[u] += g;
    }
  };
  int r = r0 + q;
  for (; r + 12 < r1; r += 16) {
    float4 d[4], xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      d[u] = reinterpret_cast<const float4*>(dy)[(int64_t)(r + 4 * u) * F4 + cc];
      xv[u] = reinterpret_cast<const float4*>(x)[(int64_t)(r + 4 * u) * F4 + cc];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(r + 4 * u, d[u], xv[u]);
  }
  for (; r < r1; r += 4) one(r, reinterpret_cast<const float4*>(dy)[(int64_t)r * F4 + cc], reinterpret_cast<const float4*>(x)[(int64_t)r * F4 + cc]);
  if (q > 0) {
    red[0][q - 1][l] = make_float4(aw[0], aw[1], aw[2], aw[3]);
    red[1][q - 1][l] = make_float4(ab[0], ab[1], ab[2], ab[3]);
  }
  __syncthreads();
  if (q == 0 && live) {
    const int F = F4 * 4;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float4 u0 = red[k][0][l], u1 = red[k][1][l], u2 = red[k][2][l];
      const float* mine = k ? ab : aw;
      float* dst = acc + (k ? F : 0) + c4 * 4;
      unsafeAtomicAdd(dst + 0, mine[0] + u0.x + u1.x + u2.x);
      unsafeAtomicAdd(dst + 1, mine[1] + u0.y + u1.y + u2.y);
      unsafeAtomicAdd(dst + 2, mine[2] + u0.z + u1.z + u2.z);
      unsafeAtomicAdd(dst + 3, mine[3] + u0.w + u1.w + u2.w);
    }
  }
}
// phase 1 (fused 1a + 1b, one pass over dy and x instead of two): affine gradients dw[e] += sum_f g*xhat, db[e] += sum_f g
// AND the frame sums s1[f] += sum_e g*w, s2[f] += sum_e g*w*xhat (wave reduction per frame, stored as per-wave partials).
// Workgroup = 64 float4 positions of the frame x 4 waves that take every fourth frame of the chunk: 8 waves per SIMD in
// flight instead of 2 (the first version -- thread per position, 40 frames in sequence -- ran its ~50 VALU ops per element
// and its two loads per frame back to back: 64 us against a 35 us HBM time), and the four waves' affine sums meet in LDS
// so that the atomic count does not grow with the parallelism.  Lanes past E4 keep running with zero weight so that every
// wave takes part in the shuffles.
__global__ __launch_bounds__(256) void norm_act_bwd_frame_affine(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ w, const float* __restrict__ b,
                                                                 float* __restrict__ dw, float* __restrict__ db,
                                                                 float* __restrict__ fsum /* [gridDim.x, frames, 2] partials */, int E4, int F,
                                                                 int HW, int act, float p, const uint64_t* seed_dev,
                                                                 uint32_t site, int frames, int fpb,
                                                                 const float* __restrict__ rowscale, int rs_div, int rs_mod,
                                                                 float* __restrict__ part /* [gridDim.y][2][4 * E4] or null */) {
  __shared__ float sred[3][64][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e_raw = blockIdx.x * 64 + lane;
  const bool live = e_raw < E4;
  const int e = live ? e_raw : E4 - 1;
  const float lv = live ? 1.f : 0.f;
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const int f0 = blockIdx.y * fpb, f1 = min(frames, f0 + fpb);
  const float4 wv = reinterpret_cast<const float4*>(w)[e], bv = reinterpret_cast<const float4*>(b)[e];
  const float ws[4] = {wv.x, wv.y, wv.z, wv.w}, bs[4] = {bv.x, bv.y, bv.z, bv.w};
  float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  const int hw = (e * 4) / F;
  // two frames per iteration: both frames' loads are issued before the first frame's reductions (a wave has nothing else in flight)
  auto one_frame = [&](const int f, const float4 d, const float4 xv, float& t1, float& t2) {
    const float mu = mean[f], rs = rstd[f];
    const int64_t i = ((int64_t)f * E4 + e) * 4;
    float rsc = 1.f;
    if (rowscale) rsc = rowscale[((f * HW + hw) / rs_div) % rs_mod];
    const float dv[4] = {d.x, d.y, d.z, d.w}, xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float xh = (xs[q] - mu) * rs;
      const float ds = (p > 0.f ? vptr_drop_scale(seed, site, (uint64_t)(i + q), p) : 1.f) * rsc;
      const float g = norm_act_g(dv[q], xh, ws[q], bs[q], act, ds) * lv;
      aw[q] += g * xh;
      ab[q] += g;
      t1 += g * ws[q];
      t2 += g * ws[q] * xh;
    }
  };
  for (int f = f0 + wave; f < f1; f += 8) {
    const int fb = f + 4;
    const bool two = fb < f1;
    const int fbc = two ? fb : f;
    const float4 d0 = reinterpret_cast<const float4*>(dy)[(int64_t)f * E4 + e];
    const float4 x0 = reinterpret_cast<const float4*>(x)[(int64_t)f * E4 + e];
    const float4 d1 = reinterpret_cast<const float4*>(dy)[(int64_t)fbc * E4 + e];
    const float4 x1 = reinterpret_cast<const float4*>(x)[(int64_t)fbc * E4 + e];
    float t1 = 0.f, t2 = 0.f, u1 = 0.f, u2 = 0.f;
    one_frame(f, d0, x0, t1, t2);
    if (two) one_frame(fb, d1, x1, u1, u2);   // wave-uniform
    if (fsum) {  // per-wave partials, no atomics: 500+ waves adding into the same 2*frames words serialise badly
      t1 = wave_sum(t1);
      t2 = wave_sum(t2);
      u1 = wave_sum(u1);
      u2 = wave_sum(u2);
      if (lane == 0) {
        float* dst = fsum + ((int64_t)blockIdx.x * frames + f) * 2;
        dst[0] = t1;
        dst[1] = t2;
        if (two) {
          float* dst2 = fsum + ((int64_t)blockIdx.x * frames + fb) * 2;
          dst2[0] = u1;
          dst2[1] = u2;
        }
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      sred[wave - 1][lane][q] = aw[q];
      sred[wave - 1][lane][4 + q] = ab[q];
    }
  }
  __syncthreads();
  if (wave > 0 || !live) return;
  if (part) {   // deferred: this frame chunk's sums as one row pair of [gridDim.y][2][E]; vptr_partial_reduce adds them later
    float4 ow, ob;
    ow.x = aw[0] + sred[0][lane][0] + sred[1][lane][0] + sred[2][lane][0];
    ow.y = aw[1] + sred[0][lane][1] + sred[1][lane][1] + sred[2][lane][1];
    ow.z = aw[2] + sred[0][lane][2] + sred[1][lane][2] + sred[2][lane][2];
    ow.w = aw[3] + sred[0][lane][3] + sred[1][lane][3] + sred[2][lane][3];
    ob.x = ab[0] + sred[0][lane][4] + sred[1][lane][4] + sred[2][lane][4];
    ob.y = ab[1] + sred[0][lane][5] + sred[1][lane][5] + sred[2][lane][5];
    ob.z = ab[2] + sred[0][lane][6] + sred[1][lane][6] + sred[2][lane][6];
    ob.w = ab[3] + sred[0][lane][7] + sred[1][lane][7] + sred[2][lane][7];
    float4* pw = reinterpret_cast<float4*>(part + ((int64_t)blockIdx.y * 2) * 4 * E4) + e;
    pw[0] = ow;
    pw[E4] = ob;
    return;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsafeAtomicAdd(dw + (int64_t)e * 4 + q, aw[q] + sred[0][lane][q] + sred[1][lane][q] + sred[2][lane][q]);
    unsafeAtomicAdd(db + (int64_t)e * 4 + q, ab[q] + sred[0][lane][4 + q] + sred[1][lane][4 + q] + sred[2][lane][4 + q]);
  }
}
// ---------------------------------------------------------------------------------------------------------------
// Round 6: the two phases in ONE pass over dy and x (LayerNorm((F,H,W)), deferred affine gradients).  A workgroup = 256 element quads
// (one 64-quad range per wave) x NF frames; every thread keeps g*w and xhat of its NF frames in registers (2 NF float4), the workgroup
// publishes its share of each frame's two sums -- the four waves' sums meet in LDS first, ONE thread per frame then adds them to the
// frame's own 128-byte line of the workspace (sum, sum, arrival counter) -- and every wave waits until all gridDim.x workgroups of its
// frame chunk have arrived before it computes dx from its registers: 260 MB per [10240 x 2112] call instead of 432, one launch instead
// of three.  Round 3 built this with 528 workgroups adding to 16-frames-per-line words and measured the whole step TWICE as slow; the
// elimination runs of round 6 (tools/dwn_probe.py) showed why -- same-line atomics serialise at the memory-side atomic unit -- and what
// it takes: one line per frame, as few atomics per frame as the register budget allows (here 132 x 3).
// Ordering: the two sums are added with RETURNING atomics whose results the thread waits for (they are performed at the memory side by
// then), only then the counter is bumped; a reader that has seen the counter complete reads the sums with device-scope atomic loads.
// Co-residency: the waiters of a chunk (blockIdx.y, the slow grid axis) only wait for workgroups dispatched before or together with them;
// the launcher takes this path only when a chunk is a small fraction of what the device holds.  A bounded spin turns a scheduling surprise
// into wrong numbers and a raised flag (sync_ws line `frames`), never into a hang.
// sync_ws: [frames + 1][32] floats, ZERO on entry (not restored: the caller hands a fresh zeroed slice per call).
// ---------------------------------------------------------------------------------------------------------------
template <int NF>
__global__ __launch_bounds__(256) void norm_act_bwd_coop_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ w, const float* __restrict__ b,
                                                                float* __restrict__ dx, float* __restrict__ part /* [gridDim.y][2][4 * E4] */,
                                                                float* __restrict__ sync_ws, int E4, int F, int HW, int act, float p,
                                                                const uint64_t* seed_dev, uint32_t site, int frames,
                                                                const float* __restrict__ rowscale, int rs_div, int rs_mod, int p16) {
  __shared__ float sred[NF][4][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e_raw = blockIdx.x * 256 + threadIdx.x;
  const bool live = e_raw < E4;
  const int e = live ? e_raw : E4 - 1;
  const float lv = live ? 1.f : 0.f;
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const int f0 = blockIdx.y * NF, f1 = min(frames, f0 + NF);
  const float4 wv = reinterpret_cast<const float4*>(w)[e], bv = reinterpret_cast<const float4*>(b)[e];
  const float ws[4] = {wv.x, wv.y, wv.z, wv.w}, bs[4] = {bv.x, bv.y, bv.z, bv.w};
  float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  const int hw = (e * 4) / F;
  float4 gw[NF], xh4[NF];
  // every frame's loads first (NF x 2 independent 16-byte loads in flight per thread)
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    const int f = min(f0 + k, frames - 1);
    gw[k] = reinterpret_cast<const float4*>(dy)[(int64_t)f * E4 + e];
    xh4[k] = reinterpret_cast<const float4*>(x)[(int64_t)f * E4 + e];
  }
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    const int f = f0 + k;
    float t1 = 0.f, t2 = 0.f;
    if (f < f1) {   // block-uniform
      const float mu = mean[f], rs = rstd[f];
      const int64_t i = ((int64_t)f * E4 + e) * 4;
      float rsc = 1.f;
      if (rowscale) rsc = rowscale[((f * HW + hw) / rs_div) % rs_mod];
      const float dv[4] = {gw[k].x, gw[k].y, gw[k].z, gw[k].w}, xs[4] = {xh4[k].x, xh4[k].y, xh4[k].z, xh4[k].w};
      float go[4], xo[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xh = (xs[q] - mu) * rs;
        const float ds = (p > 0.f ? vptr_drop_scale(seed, site, (uint64_t)(i + q), p) : 1.f) * rsc;
        const float g = norm_act_g(dv[q], xh, ws[q], bs[q], act, ds) * lv;
        aw[q] += g * xh;
        ab[q] += g;
        go[q] = g * ws[q];
        xo[q] = xh;
        t1 += go[q];
        t2 += go[q] * xh;
      }
      gw[k] = make_float4(go[0], go[1], go[2], go[3]);
      xh4[k] = make_float4(xo[0], xo[1], xo[2], xo[3]);
    }
    t1 = wave_sum(t1);
    t2 = wave_sum(t2);
    if (lane == 0) { sred[k][wave][0] = t1; sred[k][wave][1] = t2; }
  }
  __syncthreads();
  if (threadIdx.x < NF && f0 + (int)threadIdx.x < f1) {   // thread k publishes frame f0 + k
    const int f = f0 + threadIdx.x, k = threadIdx.x;
    float* line = sync_ws + (int64_t)f * 32;
    const float r1 = __hip_atomic_fetch_add(line, (sred[k][0][0] + sred[k][1][0]) + (sred[k][2][0] + sred[k][3][0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float r2 = __hip_atomic_fetch_add(line + 1, (sred[k][0][1] + sred[k][1][1]) + (sred[k][2][1] + sred[k][3][1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(r1), "v"(r2) : "memory");   // both additions have been performed (their old values are back) ...
    // ... before this workgroup counts as arrived.  RELAXED on purpose: device-scope atomics are performed at the memory side, which is all the
    // ordering this exchange needs; a release / acquire pair at agent scope writes back / invalidates the XCD's whole L2 on gfx950 (the L2s of
    // the eight XCDs are not coherent with each other) -- measured 728 us per launch instead of ~50.
    __hip_atomic_fetch_add(reinterpret_cast<int*>(line + 2), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // affine gradients of this frame chunk: one row pair of the partial-sum buffer (every thread owns its quad: no reduction needed)
  if (live) {
    float4* pw = reinterpret_cast<float4*>(part + ((int64_t)blockIdx.y * 2) * 4 * E4) + e;
    pw[0] = make_float4(aw[0], aw[1], aw[2], aw[3]);
    pw[E4] = make_float4(ab[0], ab[1], ab[2], ab[3]);
  }
  // phase 2 from registers
  const float inv_n = 1.f / ((float)HW * (float)F);
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    const int f = f0 + k;
    if (f < f1) {
      float s1 = 0.f, s2 = 0.f;
      if (lane == 0) {
        const float* line = sync_ws + (int64_t)f * 32;
        int spins = 0;
        while (__hip_atomic_load(reinterpret_cast<const int*>(line + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 22)) { reinterpret_cast<int*>(sync_ws + (int64_t)frames * 32)[0] = 1; break; }   // (about a second: never in a healthy launch)
        }
        asm volatile("" ::: "memory");   // the sums are requested only after the counter was seen complete
        s1 = __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s2 = __hip_atomic_load(line + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s1 = __shfl(s1, 0, 64) * inv_n;
      s2 = __shfl(s2, 0, 64) * inv_n;
      const float rs = rstd[f];
      if (live) {
        const float4 o = make_float4(rs * (gw[k].x - s1 - xh4[k].x * s2), rs * (gw[k].y - s1 - xh4[k].y * s2),
                                     rs * (gw[k].z - s1 - xh4[k].z * s2), rs * (gw[k].w - s1 - xh4[k].w * s2));
        vptr_store4_fmt(dx, ((int64_t)f * E4 + e) * 4, o, p16);
      }
    }
  }
}
constexpr int NORM_COOP_NF = 10;
// chunks of the one-pass launch (= rows of its partial-sum buffer); 0: this geometry / configuration takes the two-phase path
extern "C" int vptr_norm_act_bwd_coop_partials(int rows, int F, int HW) {
  static int on = -1, capacity = 0;
  if (on < 0) {
    const char* e = getenv("VPTR_NORM_COOP");
    on = (e && atoi(e) == 0) ? 0 : 1;
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (on && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, norm_act_bwd_coop_kernel<NORM_COOP_NF>, 256, 0) == hipSuccess)
      capacity = per_cu * prop.multiProcessorCount;
    if (capacity <= 0) on = 0;
  }
  if (!on || g_vptr_deterministic || HW < 1 || rows % HW != 0 || F % 4 != 0) return 0;
  const int frames = rows / HW;
  const int wgs = cdiv(HW * (F / 4), 256);   // mutually waiting workgroups of one chunk: at most a quarter of what the device holds at once
  if (frames < 16 || wgs * 4 > capacity) return 0;
  return cdiv(frames, NORM_COOP_NF);
}
extern "C" int vptr_norm_act_bwd_coop(const float* dy, const float* x, const float* mean, const float* rstd, const float* w, const float* b,
                                      float* dx, float* sync_ws, int rows, int F, int HW, int act, float dropout_p, const uint64_t* seed_dev,
                                      uint32_t site, const float* rowscale, int rs_div, int rs_mod, int p16, float* partials, vptr_stream_t stream) {
  const int chunks = vptr_norm_act_bwd_coop_partials(rows, F, HW);
  VPTR_CHECK(chunks > 0 && dy && x && mean && rstd && w && b && dx && sync_ws && partials, "norm_act_bwd_coop: no one-pass variant for rows %d, F %d, HW %d", rows, F, HW);
  VPTR_CHECK(((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(w) |
               reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(partials)) & 15) == 0 && (reinterpret_cast<uintptr_t>(sync_ws) & 127) == 0,
             "norm_act_bwd_coop: operands must be 16-byte aligned, sync_ws 128-byte aligned");
  if (p16) VPTR_CHECK(F % 16 == 0 && (reinterpret_cast<uintptr_t>(dx) & 63) == 0, "norm_act_bwd_coop: a P16 dx needs F %% 16 == 0 and a 64-byte aligned dx");
  if (dropout_p > 0.f) VPTR_CHECK(seed_dev && dropout_p < 1.f, "norm_act_bwd_coop: dropout needs seed_dev");
  const int E4 = HW * (F / 4);
  norm_act_bwd_coop_kernel<NORM_COOP_NF><<<dim3(cdiv(E4, 256), chunks), 256, 0, (hipStream_t)stream>>>(
      dy, x, mean, rstd, w, b, dx, partials, sync_ws, E4, F, HW, act, dropout_p, seed_dev, site, rows / HW, rowscale, rs_div, rs_mod, p16);
  VPTR_LAUNCH_CHECK();
  return 0;
}
// phase 2: dx = rstd * (g*w - S1/n - xhat*S2/n)
template <bool PER_COL>
__global__ __launch_bounds__(256) void norm_act_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ w, const float* __restrict__ b,
                                                              const float* __restrict__ acc, float* __restrict__ dx, int rows,
                                                              int F, int HW, int act, float p, const uint64_t* seed_dev,
                                                              uint32_t site, int nacc, int const_stats,
                                                              const float* __restrict__ rowscale, int rs_div, int rs_mod) {
  const int64_t total = (int64_t)rows * F;
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const float inv_n = PER_COL ? 1.f / (float)rows : 1.f / (float)((int64_t)HW * F);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / F), c = (int)(i - (int64_t)row * F);
    float mu, rs, ww, bb, s1, s2;
    if (PER_COL) {
      mu = mean[c]; rs = rstd[c]; ww = w[c]; bb = b[c];
      s1 = ww * acc[nacc + c];  // w * sum g
      s2 = ww * acc[c];         // w * sum g*xhat
    } else {
      const int f = row / HW, hw = row - f * HW;
      mu = mean[f]; rs = rstd[f];
      ww = w[(int64_t)hw * F + c]; bb = b[(int64_t)hw * F + c];
      s1 = acc[f]; s2 = acc[nacc + f];
    }
    const float xh = (x[i] - mu) * rs;
    float ds = p > 0.f ? vptr_drop_scale(seed, site, (uint64_t)i, p) : 1.f;
    if (rowscale) ds *= rowscale[(row / rs_div) % rs_mod];
    const float g = norm_act_g(dy[i], xh, ww, bb, act, ds);
    if (const_stats) { s1 = 0.f; s2 = 0.f; }
    dx[i] = rs * (g * ww - s1 * inv_n - xh * s2 * inv_n);
  }
}
// the same for F % 4 == 0: four channels per thread, 16-byte accesses, and optionally a P16 output (dx only feeds the input- and
// weight-gradient GEMMs of the 1x1 convolution in front of this normalisation)
template <bool PER_COL>
__global__ __launch_bounds__(256) void norm_act_bwd_dx4_kernel(const float4* __restrict__ dy, const float4* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ w, const float* __restrict__ b,
                                                               const float* __restrict__ acc, float* __restrict__ dx, int rows,
                                                               int F4, int HW, int act, float p, const uint64_t* seed_dev,
                                                               uint32_t site, int nacc, int const_stats,
                                                               const float* __restrict__ rowscale, int rs_div, int rs_mod, int p16) {
  const int64_t total = (int64_t)rows * F4;
  const int F = F4 * 4;
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const float inv_n = PER_COL ? 1.f / (float)rows : 1.f / (float)((int64_t)HW * F);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / F4), c4 = (int)(i - (int64_t)row * F4);
    const float4 xv = x[i], dv = dy[i];
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds4[4] = {dv.x, dv.y, dv.z, dv.w};
    float4 ww, bb;
    float mu[4], rs[4], s1[4], s2[4];
    if (PER_COL) {
      ww = reinterpret_cast<const float4*>(w)[c4];
      bb = reinterpret_cast<const float4*>(b)[c4];
      const float4 m4 = reinterpret_cast<const float4*>(mean)[c4], r4 = reinterpret_cast<const float4*>(rstd)[c4];
      const float4 a1 = reinterpret_cast<const float4*>(acc + nacc)[c4], a2 = reinterpret_cast<const float4*>(acc)[c4];
      mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
      rs[0] = r4.x; rs[1] = r4.y; rs[2] = r4.z; rs[3] = r4.w;
      s1[0] = ww.x * a1.x; s1[1] = ww.y * a1.y; s1[2] = ww.z * a1.z; s1[3] = ww.w * a1.w;   // w * sum g
      s2[0] = ww.x * a2.x; s2[1] = ww.y * a2.y; s2[2] = ww.z * a2.z; s2[3] = ww.w * a2.w;   // w * sum g*xhat
    } else {
      const int f = row / HW, hw = row - f * HW;
      ww = reinterpret_cast<const float4*>(w)[(int64_t)hw * F4 + c4];
      bb = reinterpret_cast<const float4*>(b)[(int64_t)hw * F4 + c4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { mu[u] = mean[f]; rs[u] = rstd[f]; s1[u] = acc[f]; s2[u] = acc[nacc + f]; }
    }
    const float wv[4] = {ww.x, ww.y, ww.z, ww.w}, bv[4] = {bb.x, bb.y, bb.z, bb.w};
    const float rsc = rowscale ? rowscale[(row / rs_div) % rs_mod] : 1.f;
    float o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float xh = (xs[u] - mu[u]) * rs[u];
      float dsc = p > 0.f ? vptr_drop_scale(seed, site, (uint64_t)i * 4 + u, p) : 1.f;
      dsc *= rsc;
      const float g = norm_act_g(ds4[u], xh, wv[u], bv[u], act, dsc);
      const float t1 = const_stats ? 0.f : s1[u], t2 = const_stats ? 0.f : s2[u];
      o[u] = rs[u] * (g * wv[u] - t1 * inv_n - xh * t2 * inv_n);
    }
    vptr_store4_fmt(dx, i * 4, make_float4(o[0], o[1], o[2], o[3]), p16);
  }
}
// position-major form of norm_act_bwd_dx4_kernel<false> (see norm_act_fwd_pos_kernel)
__global__ __launch_bounds__(256) void norm_act_bwd_dx4_pos_kernel(const float4* __restrict__ dy, const float4* __restrict__ x,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   const float* __restrict__ w, const float* __restrict__ b,
                                                                   const float* __restrict__ acc, float* __restrict__ dx, int frames,
                                                                   int F4, int HW, int act, float p, const uint64_t* seed_dev,
                                                                   uint32_t site, int nacc, int const_stats,
                                                                   const float* __restrict__ rowscale, int rs_div, int rs_mod, int p16) {
  const int P = HW * F4;
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= P) return;
  const int hw = pos / F4;
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const float inv_n = 1.f / (float)((int64_t)HW * F4 * 4);
  const float4 ww = reinterpret_cast<const float4*>(w)[pos], bb = reinterpret_cast<const float4*>(b)[pos];
  const float wv[4] = {ww.x, ww.y, ww.z, ww.w}, bv[4] = {bb.x, bb.y, bb.z, bb.w};
  for (int f = blockIdx.y; f < frames; f += gridDim.y) {
    const int64_t i = (int64_t)f * P + pos;
    const float4 xv = x[i], dv = dy[i];
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds4[4] = {dv.x, dv.y, dv.z, dv.w};
    const float mu = mean[f], rs = rstd[f];
    const float t1 = const_stats ? 0.f : acc[f], t2 = const_stats ? 0.f : acc[nacc + f];
    const float rsc = rowscale ? rowscale[((f * HW + hw) / rs_div) % rs_mod] : 1.f;
    float o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float xh = (xs[u] - mu) * rs;
      float dsc = p > 0.f ? vptr_drop_scale(seed, site, (uint64_t)i * 4 + u, p) : 1.f;
      dsc *= rsc;
      const float g = norm_act_g(ds4[u], xh, wv[u], bv[u], act, dsc);
      o[u] = rs * (g * wv[u] - t1 * inv_n - xh * t2 * inv_n);
    }
    vptr_store4_fmt(dx, i * 4, make_float4(o[0], o[1], o[2], o[3]), p16);
  }
}
// phase 1c: s1[f], s2[f] = sum of the per-wave partials of phase 1
// (256 threads per frame since round 6: one wave walked 33 dependent strides per frame on 16 x 16 maps -- 14.5 us for 170 KB)
__global__ __launch_bounds__(256) void norm_act_bwd_frame_final(const float* __restrict__ part, float* __restrict__ fsum, int nparts,
                                                               int frames) {
  __shared__ float red[8];
  const int f = blockIdx.x;
  float t1 = 0.f, t2 = 0.f;
  for (int q = threadIdx.x; q < nparts; q += 256) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((int64_t)q * frames + f) * 2);
    t1 += v.x;
    t2 += v.y;
  }
  t1 = wave_sum(t1);
  t2 = wave_sum(t2);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = t1; red[4 + (threadIdx.x >> 6)] = t2; }
  __syncthreads();
  if (threadIdx.x == 0) { fsum[f] = (red[0] + red[1]) + (red[2] + red[3]); fsum[frames + f] = (red[4] + red[5]) + (red[6] + red[7]); }
}
__global__ void accum2_kernel(const float* __restrict__ acc, float* __restrict__ dw, float* __restrict__ db, int F) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < F) { dw[c] += acc[c]; db[c] += acc[F + c]; }
}

// scratch buffers are zeroed by an ordinary kernel (a kernel node under stream capture) rather than hipMemsetAsync (a
// memset node that may be served by a different engine).
__global__ void zero_fill_kernel(float* __restrict__ p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

// frame chunks (= rows of partial sums [chunks][2][HW * F]) of a deferred LayerNorm((F,H,W)) backward call; 0: no deferred variant
extern "C" int vptr_norm_act_bwd_partials(int rows, int F, int HW, int per_col) {
  if (per_col || HW < 1 || rows % HW != 0 || F % 4 != 0) return 0;
  const int frames = rows / HW;
  if (frames < 64) return 0;
  static int big_split = -1, small_split = -1;
  if (big_split < 0) {
    const char* e = getenv("VPTR_NORM_SPLIT");   // "big,small" frame chunks (experiments)
    big_split = 4; small_split = 16;
    if (e) (void)sscanf(e, "%d,%d", &big_split, &small_split);
  }
  const int want = (int64_t)HW * F >= 65536 ? big_split : small_split;
  const int fpb = (frames + want - 1) / want;
  return (frames + fpb - 1) / fpb;   // chunks of fpb frames (<= want)
}
static int norm_act_bwd_impl(const float* dy, const float* x, const float* mean, const float* rstd, const float* w,
                             const float* b, float* dx, float* dw, float* db, float* scratch, int rows, int F, int HW,
                             int per_col, int act, int const_stats, float dropout_p, const uint64_t* seed_dev,
                             uint32_t site, const float* rowscale, int rs_div, int rs_mod, int p16, float* partials, vptr_stream_t stream) {
  VPTR_CHECK(rows > 0 && F > 0 && HW >= 1 && scratch && dx && (partials || (dw && db)), "norm_act_bwd: bad arguments");
  if (p16) VPTR_CHECK(F % 16 == 0 && (reinterpret_cast<uintptr_t>(dx) & 63) == 0, "norm_act_bwd: a P16 dx needs F %% 16 == 0 and a 64-byte aligned dx");
  const bool vec4 = F % 4 == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) |
                                     reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(mean) |
                                     reinterpret_cast<uintptr_t>(rstd) | reinterpret_cast<uintptr_t>(scratch)) & 15) == 0;
  const int blocks4 = (int)hmin64(((int64_t)rows * (F / 4) + 255) / 256, 8192);
  if (p16) VPTR_CHECK(vec4, "norm_act_bwd: a P16 dx needs 16-byte aligned operands");
  if (dropout_p > 0.f) VPTR_CHECK(seed_dev && dropout_p < 1.f, "norm_act_bwd: dropout needs seed_dev");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)rows * F;
  const int blocks = (int)hmin64((total + 255) / 256, 8192);
  if (per_col) {
    zero_fill_kernel<<<cdiv(2 * F, 256), 256, 0, st>>>(scratch, 2 * F);  // a kernel node, not a memset node (see below)
    const int rpb = g_vptr_deterministic ? rows : 64;   // (32 rows per chunk for the narrow tensors: 33.7 -> 42.2 us -- twice the atomics; deterministic: one adder per column)
    if (vec4)
      norm_act_bwd_col_reduce4<<<dim3(cdiv(F / 4, 64), cdiv(rows, rpb)), 256, 0, st>>>(dy, x, mean, rstd, w, b, scratch, rows, F / 4, act,
                                                                                      dropout_p, seed_dev, site, rpb, rowscale, rs_div, rs_mod);
    else
    norm_act_bwd_col_reduce<<<dim3(cdiv(F, 256), cdiv(rows, rpb)), 256, 0, st>>>(dy, x, mean, rstd, w, b, scratch, rows, F, act,
                                                                                 dropout_p, seed_dev, site, rpb, rowscale, rs_div, rs_mod);
    if (vec4)
      norm_act_bwd_dx4_kernel<true><<<blocks4, 256, 0, st>>>(reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(x), mean, rstd, w, b,
                                                             scratch, dx, rows, F / 4, HW, act, dropout_p, seed_dev, site, F, const_stats,
                                                             rowscale, rs_div, rs_mod, p16);
    else
    norm_act_bwd_dx_kernel<true><<<blocks, 256, 0, st>>>(dy, x, mean, rstd, w, b, scratch, dx, rows, F, HW, act, dropout_p,
                                                         seed_dev, site, F, const_stats, rowscale, rs_div, rs_mod);
    accum2_kernel<<<cdiv(F, 256), 256, 0, st>>>(scratch, dw, db, F);
  } else {
    VPTR_CHECK(rows % HW == 0, "norm_act_bwd: rows must be a multiple of HW");
    const int frames = rows / HW;
    VPTR_CHECK(F % 4 == 0, "norm_act_bwd: F must be a multiple of 4");
    const int E4 = HW * F / 4;
    const int ysplit = partials ? vptr_norm_act_bwd_partials(rows, F, HW, 0) : 0;
    if (partials) VPTR_CHECK(ysplit > 0 && (reinterpret_cast<uintptr_t>(partials) & 15) == 0, "norm_act_bwd: no deferred variant for this geometry");
    // deferred: no atomics, so the frames are cut into more chunks (more waves in flight, 2 - 5 frames per wave instead of 10)
    const int fpb = partials ? (frames + ysplit - 1) / ysplit : ((frames >= 64 && !g_vptr_deterministic) ? (frames + 3) / 4 : frames);   // no partial buffer + deterministic: one adder per element
    if (partials) VPTR_CHECK(cdiv(frames, fpb) == ysplit, "norm_act_bwd: frames %d do not split into %d chunks", frames, ysplit);   // (holds by construction)
    const int nparts = cdiv(E4, 64);  // scratch: [2*frames] sums followed by [nparts, frames, 2] per-wave partials
    float* part = scratch + 2 * frames;
    norm_act_bwd_frame_affine<<<dim3(nparts, cdiv(frames, fpb)), 256, 0, st>>>(dy, x, mean, rstd, w, b, dw, db, part, E4, F, HW, act,
                                                                                     dropout_p, seed_dev, site, frames, fpb, rowscale,
                                                                                     rs_div, rs_mod, partials);
    norm_act_bwd_frame_final<<<frames, 256, 0, st>>>(part, scratch, nparts, frames);
    if (vec4 && norm_pos_mode() && frames >= 16 && (int64_t)rows * (F / 4) >= (1 << 18))
      norm_act_bwd_dx4_pos_kernel<<<dim3(cdiv(HW * (F / 4), 256), (frames / 4 < 1 ? 1 : (frames / 4 > 65535 ? 65535 : frames / 4))), 256, 0, st>>>(
          reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(x), mean, rstd, w, b, scratch, dx, frames, F / 4, HW, act, dropout_p,
          seed_dev, site, frames, const_stats, rowscale, rs_div, rs_mod, p16);
    else if (vec4)
      norm_act_bwd_dx4_kernel<false><<<blocks4, 256, 0, st>>>(reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(x), mean, rstd, w, b,
                                                              scratch, dx, rows, F / 4, HW, act, dropout_p, seed_dev, site, frames, const_stats,
                                                              rowscale, rs_div, rs_mod, p16);
    else
    norm_act_bwd_dx_kernel<false><<<blocks, 256, 0, st>>>(dy, x, mean, rstd, w, b, scratch, dx, rows, F, HW, act, dropout_p,
                                                          seed_dev, site, frames, const_stats, rowscale, rs_div, rs_mod);
  }
  VPTR_LAUNCH_CHECK();
  return 0;
}
extern "C" int vptr_norm_act_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* w,
                                 const float* b, float* dx, float* dw, float* db, float* scratch, int rows, int F, int HW,
                                 int per_col, int act, int const_stats, float dropout_p, const uint64_t* seed_dev,
                                 uint32_t site, const float* rowscale, int rs_div, int rs_mod, int p16, vptr_stream_t stream) {
  return norm_act_bwd_impl(dy, x, mean, rstd, w, b, dx, dw, db, scratch, rows, F, HW, per_col, act, const_stats, dropout_p, seed_dev, site,
                           rowscale, rs_div, rs_mod, p16, nullptr, stream);
}
extern "C" int vptr_norm_act_bwd_deferred(const float* dy, const float* x, const float* mean, const float* rstd, const float* w,
                                          const float* b, float* dx, float* scratch, int rows, int F, int HW, int act, int const_stats,
                                          float dropout_p, const uint64_t* seed_dev, uint32_t site, const float* rowscale, int rs_div,
                                          int rs_mod, int p16, float* partials, vptr_stream_t stream) {
  VPTR_CHECK(partials != nullptr, "norm_act_bwd_deferred: null partial-sum buffer");
  return norm_act_bwd_impl(dy, x, mean, rstd, w, b, dx, nullptr, nullptr, scratch, rows, F, HW, 0, act, const_stats, dropout_p, seed_dev, site,
                           rowscale, rs_div, rs_mod, p16, partials, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// depthwise 3x3, padding 1, channel-last [frames, H, W, F]; weights tap-major [9, F].
// ---------------------------------------------------------------------------------------------------------------
// Forward (and, with flipped taps, the data gradient): thread = (frame, x column, 4 channels); it walks down the column
// with a rolling 3-row window in registers, so every output costs 3 new float4 loads instead of 9 inputs + 9 weights
// (the first version was bound by the CU's vector-memory issue rate, not by HBM).
struct DwRow { float4 l, m, r; };
// branch-free: addresses clamped into the image, out-of-range taps multiplied by zero
__device__ __forceinline__ float4 scale4(const float4 v, const float s) { return make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }
__device__ __forceinline__ DwRow dw_load_row(const float4* __restrict__ x, int64_t frame_row0, int row, int H, int xw, int W, int F4,
                                             int c4) {
  const float rok = (row >= 0 && row < H) ? 1.f : 0.f;
  const int64_t base = (frame_row0 + min(max(row, 0), H - 1)) * W;
  DwRow o;
  o.l = scale4(x[(base + max(xw - 1, 0)) * F4 + c4], xw > 0 ? rok : 0.f);
  o.m = scale4(x[(base + xw) * F4 + c4], rok);
  o.r = scale4(x[(base + min(xw + 1, W - 1)) * F4 + c4], xw + 1 < W ? rok : 0.f);
  return o;
}
__device__ __forceinline__ void fma4(float4& a, const float4 w, const float4 v) {
  a.x += w.x * v.x; a.y += w.y * v.y; a.z += w.z * v.z; a.w += w.w * v.w;
}
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float* __restrict__ x_, const float* __restrict__ w9,
                                                         const float* __restrict__ b, float* __restrict__ y_, int frames, int H,
                                                         int W, int F4, int flip) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)frames * W * F4) return;
  const int c4 = (int)(idx % F4);
  const int xw = (int)((idx / F4) % W);
  const int64_t f = idx / ((int64_t)F4 * W);
  const float4* __restrict__ x = reinterpret_cast<const float4*>(x_);
  float4* __restrict__ y = reinterpret_cast<float4*>(y_);
  float4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = reinterpret_cast<const float4*>(w9)[(int64_t)(flip ? 8 - t : t) * F4 + c4];
  const float4 bias = b ? reinterpret_cast<const float4*>(b)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  DwRow r0 = dw_load_row(x, f * H, -1, H, xw, W, F4, c4), r1 = dw_load_row(x, f * H, 0, H, xw, W, F4, c4);
  for (int yh = 0; yh < H; ++yh) {
    const DwRow r2 = dw_load_row(x, f * H, yh + 1, H, xw, W, F4, c4);
    float4 a = bias;
    fma4(a, w[0], r0.l); fma4(a, w[1], r0.m); fma4(a, w[2], r0.r);
    fma4(a, w[3], r1.l); fma4(a, w[4], r1.m); fma4(a, w[5], r1.r);
    fma4(a, w[6], r2.l); fma4(a, w[7], r2.m); fma4(a, w[8], r2.r);
    y[((f * H + yh) * W + xw) * F4 + c4] = a;
    r0 = r1;
    r1 = r2;
  }
}

// Two adjacent x columns per thread (W even): 4 column loads per row for 2 outputs instead of 6 -- the single-column version
// is bound by the CU's vector-memory issue rate (48 us against a 35 us HBM time at the step's shape).
struct DwRow2 { float4 c0, c1, c2, c3; };   // columns xw0-1, xw0, xw0+1, xw0+2 (out-of-range ones zeroed)
typedef __attribute__((ext_vector_type(4))) _Float16 half4_t;   // 4 channels of an fp16 side copy (dwconv_norm_fwd3_kernel)
__device__ __forceinline__ float4 dw_ld4(const float4* __restrict__ x, const int64_t i) { return x[i]; }
__device__ __forceinline__ float4 dw_ld4(const half4_t* __restrict__ x, const int64_t i) {
  const half4_t h = x[i];
  return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
}
template <typename XT>
__device__ __forceinline__ DwRow2 dw_load_row2(const XT* __restrict__ x, int64_t frame_row0, int row, int H, int xw0, int W, int F4,
                                               int c4) {
  const float rok = (row >= 0 && row < H) ? 1.f : 0.f;
  const int64_t base = (frame_row0 + min(max(row, 0), H - 1)) * W;
  DwRow2 o;
  o.c0 = scale4(dw_ld4(x, (base + max(xw0 - 1, 0)) * F4 + c4), xw0 > 0 ? rok : 0.f);
  o.c1 = scale4(dw_ld4(x, (base + xw0) * F4 + c4), rok);
  o.c2 = scale4(dw_ld4(x, (base + xw0 + 1) * F4 + c4), rok);
  o.c3 = scale4(dw_ld4(x, (base + min(xw0 + 2, W - 1)) * F4 + c4), xw0 + 2 < W ? rok : 0.f);
  return o;
}
__global__ __launch_bounds__(256) void dwconv_fwd2_kernel(const float* __restrict__ x_, const float* __restrict__ w9,
                                                          const float* __restrict__ b, float* __restrict__ y_, int frames, int H,
                                                          int W, int F4, int flip, float* __restrict__ stats) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int W2 = W >> 1;
  if (idx >= (int64_t)frames * W2 * F4) return;   // (with stats the launcher guarantees whole waves: W2 * F4 % 64 == 0)
  const int c4 = (int)(idx % F4);
  const int xw0 = (int)((idx / F4) % W2) * 2;
  const int64_t f = idx / ((int64_t)F4 * W2);
  const float4* __restrict__ x = reinterpret_cast<const float4*>(x_);
  float4* __restrict__ y = reinterpret_cast<float4*>(y_);
  float4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = reinterpret_cast<const float4*>(w9)[(int64_t)(flip ? 8 - t : t) * F4 + c4];
  const float4 bias = b ? reinterpret_cast<const float4*>(b)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  DwRow2 r0 = dw_load_row2(x, f * H, -1, H, xw0, W, F4, c4), r1 = dw_load_row2(x, f * H, 0, H, xw0, W, F4, c4);
  float ssum = 0.f, ssq = 0.f;
  for (int yh = 0; yh < H; ++yh) {
    const DwRow2 r2 = dw_load_row2(x, f * H, yh + 1, H, xw0, W, F4, c4);
    float4 a = bias, a2 = bias;
    fma4(a, w[0], r0.c0); fma4(a, w[1], r0.c1); fma4(a, w[2], r0.c2);
    fma4(a, w[3], r1.c0); fma4(a, w[4], r1.c1); fma4(a, w[5], r1.c2);
    fma4(a, w[6], r2.c0); fma4(a, w[7], r2.c1); fma4(a, w[8], r2.c2);
    fma4(a2, w[0], r0.c1); fma4(a2, w[1], r0.c2); fma4(a2, w[2], r0.c3);
    fma4(a2, w[3], r1.c1); fma4(a2, w[4], r1.c2); fma4(a2, w[5], r1.c3);
    fma4(a2, w[6], r2.c1); fma4(a2, w[7], r2.c2); fma4(a2, w[8], r2.c3);
    y[((f * H + yh) * W + xw0) * F4 + c4] = a;
    y[((f * H + yh) * W + xw0 + 1) * F4 + c4] = a2;
    if (stats) {
      ssum += ((a.x + a.y) + (a.z + a.w)) + ((a2.x + a2.y) + (a2.z + a2.w));
      ssq += ((a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w)) + ((a2.x * a2.x + a2.y * a2.y) + (a2.z * a2.z + a2.w * a2.w));
    }
    r0 = r1;
    r1 = r2;
  }
  if (stats) {   // the wave lies inside one frame (W2 * F4 % 64 == 0): per-frame sum / sum of squares for the LayerNorm((F,H,W)) that follows
    const float S = wave_sum(ssum), Q = wave_sum(ssq);
    if ((threadIdx.x & 63) == 0) {
      unsafeAtomicAdd(stats + VPTR_FRAME_STATS_STRIDE * f, S);
      unsafeAtomicAdd(stats + VPTR_FRAME_STATS_STRIDE * f + 1, Q);
    }
  }
}
// Third generation (W / 2 divides 16): the x pairs of one channel quad sit in ADJACENT lanes, every thread loads only its own two columns
// and takes the outer two from its neighbours with DPP row shifts -- 2 loads per row instead of 4 and every input element crosses the
// fabric once (dwconv_fwd2 re-fetches the shared columns from workgroups on other XCDs: 150 MB fetched for an 86.5 MB input at the
// step's shape, profiles/r04_pmc_traffic.json).
__device__ __forceinline__ float dpp_from_prev(float v) {   // lane i <- lane i - 1 (row of 16 lanes; 0 at the row start)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_next(float v) {   // lane i <- lane i + 1 (0 at the row end)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xf, 0xf, true));
}
__device__ __forceinline__ DwRow2 dw_load_row3(const float4* __restrict__ x, int64_t frame_row0, int row, int H, int xw0, int W, int F4,
                                               int c4, float lok, float rok_) {
  const float rok = (row >= 0 && row < H) ? 1.f : 0.f;
  const int64_t base = (frame_row0 + min(max(row, 0), H - 1)) * W;
  DwRow2 o;
  o.c1 = scale4(x[(base + xw0) * F4 + c4], rok);
  o.c2 = scale4(x[(base + xw0 + 1) * F4 + c4], rok);
  o.c0 = make_float4(dpp_from_prev(o.c2.x) * lok, dpp_from_prev(o.c2.y) * lok, dpp_from_prev(o.c2.z) * lok, dpp_from_prev(o.c2.w) * lok);
  o.c3 = make_float4(dpp_from_next(o.c1.x) * rok_, dpp_from_next(o.c1.y) * rok_, dpp_from_next(o.c1.z) * rok_, dpp_from_next(o.c1.w) * rok_);
  return o;
}
__global__ __launch_bounds__(256) void dwconv_fwd3_kernel(const float* __restrict__ x_, const float* __restrict__ w9,
                                                          const float* __restrict__ b, float* __restrict__ y_, int frames, int H,
                                                          int W, int F4, int flip, float* __restrict__ stats) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int W2 = W >> 1;
  if (idx >= (int64_t)frames * W2 * F4) return;   // whole groups of W2 lanes leave together (the total is a multiple of W2)
  const int xp = (int)(idx % W2), xw0 = xp * 2;
  const int c4 = (int)((idx / W2) % F4);
  const int64_t f = idx / ((int64_t)F4 * W2);
  const float lok = xp > 0 ? 1.f : 0.f, rok_ = xp + 1 < W2 ? 1.f : 0.f;
  const float4* __restrict__ x = reinterpret_cast<const float4*>(x_);
  float4* __restrict__ y = reinterpret_cast<float4*>(y_);
  float4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = reinterpret_cast<const float4*>(w9)[(int64_t)(flip ? 8 - t : t) * F4 + c4];
  const float4 bias = b ? reinterpret_cast<const float4*>(b)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  DwRow2 r0 = dw_load_row3(x, f * H, -1, H, xw0, W, F4, c4, lok, rok_), r1 = dw_load_row3(x, f * H, 0, H, xw0, W, F4, c4, lok, rok_);
  float ssum = 0.f, ssq = 0.f;
  for (int yh = 0; yh < H; ++yh) {
    const DwRow2 r2 = dw_load_row3(x, f * H, yh + 1, H, xw0, W, F4, c4, lok, rok_);
    float4 a = bias, a2 = bias;
    fma4(a, w[0], r0.c0); fma4(a, w[1], r0.c1); fma4(a, w[2], r0.c2);
    fma4(a, w[3], r1.c0); fma4(a, w[4], r1.c1); fma4(a, w[5], r1.c2);
    fma4(a, w[6], r2.c0); fma4(a, w[7], r2.c1); fma4(a, w[8], r2.c2);
    fma4(a2, w[0], r0.c1); fma4(a2, w[1], r0.c2); fma4(a2, w[2], r0.c3);
    fma4(a2, w[3], r1.c1); fma4(a2, w[4], r1.c2); fma4(a2, w[5], r1.c3);
    fma4(a2, w[6], r2.c1); fma4(a2, w[7], r2.c2); fma4(a2, w[8], r2.c3);
    y[((f * H + yh) * W + xw0) * F4 + c4] = a;
    y[((f * H + yh) * W + xw0 + 1) * F4 + c4] = a2;
    if (stats) {
      ssum += ((a.x + a.y) + (a.z + a.w)) + ((a2.x + a2.y) + (a2.z + a2.w));
      ssq += ((a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w)) + ((a2.x * a2.x + a2.y * a2.y) + (a2.z * a2.z + a2.w * a2.w));
    }
    r0 = r1;
    r1 = r2;
  }
  if (stats) {   // the wave lies inside one frame (W2 * F4 % 64 == 0)
    const float S = wave_sum(ssum), Q = wave_sum(ssq);
    if ((threadIdx.x & 63) == 0) {
      unsafeAtomicAdd(stats + VPTR_FRAME_STATS_STRIDE * f, S);
      unsafeAtomicAdd(stats + VPTR_FRAME_STATS_STRIDE * f + 1, Q);
    }
  }
}
// ---------------------------------------------------------------------------------------------------------------
// Round 6: LayerNorm((F,H,W)) + activation of the conv-FFN's first normalisation applied in the LOAD path of the depthwise kernel
// (VidHRFormer_modules.py:430-434: fc1 -> norm1 -> act1 -> dw3x3).  x is the RAW output of fc1, whose epilogue left the per-frame sum / sum of
// squares in raw_stats; every element is normalised, activated once by the thread that owns its column (the neighbours get it through the
// DPP shifts of dwconv_fwd3_kernel) and the activated tensor never exists in fp32: what the backward pass needs of it -- the x operand
// of the depthwise WEIGHT gradient -- is kept as fp16 (ah; half the bytes; |GELU| < 65504, relative rounding 2^-12 on one factor of a
// 10 240-term sum).  Per conv-FFN forward: 86.5 MB read + 86.5 MB written + 43 MB written instead of 2 x (86.5 + 86.5) MB in two launches.
// mean_out / rstd_out: the statistics the backward pass of the normalisation reads (same values in every wave of a frame: same code,
// same data; the rare large-mean guard of norm_act_fwd_kernel runs per wave here).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 dwn_act4(const float4 v, const float m, const float r, const float4 w, const float4 b, const int act, const float ok) {
  float4 o;
  o.x = vptr_act((v.x - m) * r * w.x + b.x, act) * ok;
  o.y = vptr_act((v.y - m) * r * w.y + b.y, act) * ok;
  o.z = vptr_act((v.z - m) * r * w.z + b.z, act) * ok;
  o.w = vptr_act((v.w - m) * r * w.w + b.w, act) * ok;
  return o;
}
__device__ __forceinline__ void dwn_store_half4(_Float16* __restrict__ ah, const int64_t e, const float4 v) {
  const half4_t h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  *reinterpret_cast<half4_t*>(ah + e) = h;
}
__global__ __launch_bounds__(256) void dwconv_norm_fwd3_kernel(const float* __restrict__ x_, const float* __restrict__ raw_stats,
                                                               const float* __restrict__ aw_, const float* __restrict__ ab_, float eps, int act,
                                                               const float* __restrict__ w9, const float* __restrict__ b, float* __restrict__ y_,
                                                               _Float16* __restrict__ ah, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                               int frames, int H, int W, int F4, float* __restrict__ stats) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int W2 = W >> 1;
  if (idx >= (int64_t)frames * W2 * F4) return;   // whole waves leave together (W2 * F4 % 64 == 0: a wave lies inside one frame)
  const int xp = (int)(idx % W2), xw0 = xp * 2;
  const int c4 = (int)((idx / W2) % F4);
  const int64_t f = idx / ((int64_t)F4 * W2);
  const float lok = xp > 0 ? 1.f : 0.f, rok_ = xp + 1 < W2 ? 1.f : 0.f;
  const float4* __restrict__ x = reinterpret_cast<const float4*>(x_);
  const float4* __restrict__ aw = reinterpret_cast<const float4*>(aw_);
  const float4* __restrict__ ab = reinterpret_cast<const float4*>(ab_);
  float4* __restrict__ y = reinterpret_cast<float4*>(y_);
  // the frame's statistics from its producer's sums (see norm_act_fwd_kernel)
  const int P = H * W * F4;
  const float inv_n = 1.f / ((float)P * 4.f);
  float m = raw_stats[VPTR_FRAME_STATS_STRIDE * f] * inv_n;
  const float e2 = raw_stats[VPTR_FRAME_STATS_STRIDE * f + 1] * inv_n;
  float var = fmaxf(e2 - m * m, 0.f);
  if (var < 1e-3f * e2) {   // wave-uniform (f is): |mean| > ~30 std -- exact second pass of this wave over its frame, around the approximate mean
    const float4* xf = x + f * P;
    float sq = 0.f, s1 = 0.f;
    for (int j = threadIdx.x & 63; j < P; j += 64) {
      const float4 t = xf[j];
      const float a = t.x - m, b2 = t.y - m, c = t.z - m, d = t.w - m;
      s1 += (a + b2) + (c + d);
      sq += (a * a + b2 * b2) + (c * c + d * d);
    }
    const float dm = wave_sum(s1) * inv_n;
    var = fmaxf(wave_sum(sq) * inv_n - dm * dm, 0.f);
    m += dm;
  }
  const float r = rsqrtf(var + eps);
  if (xp == 0 && c4 == 0) { mean_out[f] = m; rstd_out[f] = r; }
  float4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = reinterpret_cast<const float4*>(w9)[(int64_t)t * F4 + c4];
  const float4 bias = b ? reinterpret_cast<const float4*>(b)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  // raw operands of one input row (this thread's two columns): x and the two affine tables -- requested one row AHEAD of the row being
  // activated, so that the six loads of row y + 2 are in flight under the 8 GELUs and 72 FMAs of rows y + 1 / y
  struct Raw { float4 x1, x2, w1, w2, b1, b2; };
  auto fetch = [&](const int row) -> Raw {
    const int rc = min(max(row, 0), H - 1);
    const int64_t base = ((f * H + rc) * W + xw0) * F4 + c4;
    const int hw = (rc * W + xw0) * F4 + c4;
    Raw q;
    q.x1 = x[base]; q.x2 = x[base + F4];
#ifdef VPTR_DWN_NOAFF   // elimination build (WRONG results): no affine loads
    q.w1 = q.w2 = make_float4(1.f, 1.f, 1.f, 1.f); q.b1 = q.b2 = make_float4(0.f, 0.f, 0.f, 0.f);
#else
    q.w1 = aw[hw]; q.w2 = aw[hw + F4]; q.b1 = ab[hw]; q.b2 = ab[hw + F4];
#endif
    return q;
  };
  auto activate = [&](const Raw& q, const int row) -> DwRow2 {
    const bool in = row >= 0 && row < H;
    const float rok = in ? 1.f : 0.f;
    DwRow2 o;
    o.c1 = dwn_act4(q.x1, m, r, q.w1, q.b1, act, rok);
    o.c2 = dwn_act4(q.x2, m, r, q.w2, q.b2, act, rok);
#ifndef VPTR_DWN_NOHALF   // elimination build: no fp16 side copy
    if (ah && in) {   // every element is the OWN column of exactly one thread
      const int64_t base = ((f * H + row) * W + xw0) * F4 + c4;
      dwn_store_half4(ah, base * 4, o.c1);
      dwn_store_half4(ah, (base + F4) * 4, o.c2);
    }
#endif
    o.c0 = make_float4(dpp_from_prev(o.c2.x) * lok, dpp_from_prev(o.c2.y) * lok, dpp_from_prev(o.c2.z) * lok, dpp_from_prev(o.c2.w) * lok);
    o.c3 = make_float4(dpp_from_next(o.c1.x) * rok_, dpp_from_next(o.c1.y) * rok_, dpp_from_next(o.c1.z) * rok_, dpp_from_next(o.c1.w) * rok_);
    return o;
  };
  Raw q1 = fetch(0), q2 = fetch(1);
  DwRow2 r0, r1 = activate(q1, 0);
  r0.c0 = r0.c1 = r0.c2 = r0.c3 = make_float4(0.f, 0.f, 0.f, 0.f);   // row -1: padding of the ACTIVATED tensor
  float ssum = 0.f, ssq = 0.f;
  for (int yh = 0; yh < H; ++yh) {
    const Raw q3 = fetch(yh + 2);              // (clamped address; its values are only used while yh + 2 < H)
    const DwRow2 r2 = activate(q2, yh + 1);    // row H: all zeros (rok)
    float4 a = bias, a2 = bias;
    fma4(a, w[0], r0.c0); fma4(a, w[1], r0.c1); fma4(a, w[2], r0.c2);
    fma4(a, w[3], r1.c0); fma4(a, w[4], r1.c1); fma4(a, w[5], r1.c2);
    fma4(a, w[6], r2.c0); fma4(a, w[7], r2.c1); fma4(a, w[8], r2.c2);
    fma4(a2, w[0], r0.c1); fma4(a2, w[1], r0.c2); fma4(a2, w[2], r0.c3);
    fma4(a2, w[3], r1.c1); fma4(a2, w[4], r1.c2); fma4(a2, w[5], r1.c3);
    fma4(a2, w[6], r2.c1); fma4(a2, w[7], r2.c2); fma4(a2, w[8], r2.c3);
    y[((f * H + yh) * W + xw0) * F4 + c4] = a;
    y[((f * H + yh) * W + xw0 + 1) * F4 + c4] = a2;
    if (stats) {
      ssum += ((a.x + a.y) + (a.z + a.w)) + ((a2.x + a2.y) + (a2.z + a2.w));
      ssq += ((a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w)) + ((a2.x * a2.x + a2.y * a2.y) + (a2.z * a2.z + a2.w * a2.w));
    }
    r0 = r1;
    r1 = r2;
    q2 = q3;
  }
  if (stats) {
    const float S = wave_sum(ssum), Q = wave_sum(ssq);
    if ((threadIdx.x & 63) == 0) {
      unsafeAtomicAdd(stats + VPTR_FRAME_STATS_STRIDE * f, S);
      unsafeAtomicAdd(stats + VPTR_FRAME_STATS_STRIDE * f + 1, Q);
    }
  }
}
// The same operator on an LDS slab (second generation, round 6; VPTR_DWN_LDS=0 restores the register-walk kernel above, which measured 84 us
// per launch at the K64 step's shape -- 150 VGPRs, three waves per SIMD, an eight-row dependent chain per thread -- against 67 us for the two
// kernels it replaces).  One workgroup = one frame x 64 channels: phase 1 normalises + activates the slab's H*W x 16 channel quads ONCE (all
// loads of a thread issued before the first use), leaves them in LDS ([pixel][16 quads] float4: every wave access is 1 KB contiguous, no bank
// conflicts) and writes the fp16 side copy; phase 2 reads the nine taps of every output from LDS.  ~60 VGPRs, LDS H*W*256 B (16 KB on 8 x 8 maps).
__global__ __launch_bounds__(256) void dwconv_norm_lds_kernel(const float* __restrict__ x_, const float* __restrict__ raw_stats,
                                                              const float* __restrict__ aw_, const float* __restrict__ ab_, float eps, int act,
                                                              const float* __restrict__ w9, const float* __restrict__ b, float* __restrict__ y_,
                                                              _Float16* __restrict__ ah, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                              int H, int W, int F4, float* __restrict__ stats, const int dbg) {
  // dbg (VPTR_DWN_DBG, elimination runs of tools/dwn_probe.py; 0 in production): 1 no fp16 side copy, 2 no statistics atomics, 4 centre tap only,
  // 16 no affine-table loads, 32 no y store
  extern __shared__ float4 dwn_tile[];   // [H * W][16]
  const int tid = threadIdx.x, c4l = tid & 15, p0 = tid >> 4;
  const int c4 = blockIdx.x * 16 + c4l;
  const int64_t f = blockIdx.y;
  const int HW = H * W;
  const float4* __restrict__ x = reinterpret_cast<const float4*>(x_);
  const float4* __restrict__ aw = reinterpret_cast<const float4*>(aw_);
  const float4* __restrict__ ab = reinterpret_cast<const float4*>(ab_);
  float4* __restrict__ y = reinterpret_cast<float4*>(y_);
  // the frame's statistics from its producer's sums (see norm_act_fwd_kernel / dwconv_norm_fwd3_kernel)
  const int P = HW * F4;
  const float inv_n = 1.f / ((float)P * 4.f);
  float m = raw_stats[VPTR_FRAME_STATS_STRIDE * f] * inv_n;
  const float e2 = raw_stats[VPTR_FRAME_STATS_STRIDE * f + 1] * inv_n;
  float var = fmaxf(e2 - m * m, 0.f);
  if (var < 1e-3f * e2) {   // block-uniform: |mean| > ~30 std -- exact second pass of every wave over its frame, around the approximate mean
    const float4* xf = x + f * P;
    float sq = 0.f, s1 = 0.f;
    for (int j = tid & 63; j < P; j += 64) {
      const float4 t = xf[j];
      const float a = t.x - m, b2 = t.y - m, c = t.z - m, d = t.w - m;
      s1 += (a + b2) + (c + d);
      sq += (a * a + b2 * b2) + (c * c + d * d);
    }
    const float dm = wave_sum(s1) * inv_n;
    var = fmaxf(wave_sum(sq) * inv_n - dm * dm, 0.f);
    m += dm;
  }
  const float r = rsqrtf(var + eps);
  if (tid == 0 && blockIdx.x == 0) { mean_out[f] = m; rstd_out[f] = r; }
  float4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = reinterpret_cast<const float4*>(w9)[(int64_t)t * F4 + c4];
  const float4 bias = b ? reinterpret_cast<const float4*>(b)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  // ---- phase 1: four pixels per thread and trip (64 pixels per trip of the block): loads first, then the activations
  for (int pb = 0; pb < HW; pb += 64) {
    float4 xv[4], wv[4], bv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = min(pb + p0 + 16 * k, HW - 1);
      xv[k] = x[(f * HW + p) * F4 + c4];
      if (dbg & 16) { wv[k] = make_float4(1.f, 1.f, 1.f, 1.f); bv[k] = make_float4(0.f, 0.f, 0.f, 0.f); }
      else { wv[k] = aw[(int64_t)p * F4 + c4]; bv[k] = ab[(int64_t)p * F4 + c4]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = pb + p0 + 16 * k;
      if (p < HW) {
        const float4 a = dwn_act4(xv[k], m, r, wv[k], bv[k], act, 1.f);
        dwn_tile[p * 16 + c4l] = a;
        if (ah && !(dbg & 1)) dwn_store_half4(ah, ((f * HW + p) * F4 + c4) * 4, a);
      }
    }
  }
  __syncthreads();
  // ---- phase 2: nine taps from LDS (zero padding of the ACTIVATED tensor)
  float ssum = 0.f, ssq = 0.f;
  for (int p = p0; p < HW; p += 16) {
    const int py = p / W, px = p - py * W;
    float4 a = bias;
    if (dbg & 4) fma4(a, w[4], dwn_tile[p * 16 + c4l]);
    else {
      // branch-free taps: a clamped address and a 0 / 1 factor instead of divergent skips
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = py + ky - 1;
        const bool yok = yy >= 0 && yy < H;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = px + kx - 1;
          const bool ok = yok && xx >= 0 && xx < W;
          const float4 v = dwn_tile[(ok ? yy * W + xx : p) * 16 + c4l];
          fma4(a, w[ky * 3 + kx], scale4(v, ok ? 1.f : 0.f));
        }
      }
    }
    if (!(dbg & 32) || a.x == 12345.678f) y[(f * HW + p) * F4 + c4] = a;
    ssum += (a.x + a.y) + (a.z + a.w);
    ssq += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
  }
  if (stats && !(dbg & 2)) {   // one pair of atomics per workgroup: the four waves' sums meet in LDS first
    __shared__ float dwn_red[8];
    const float S = wave_sum(ssum), Q = wave_sum(ssq);
    if ((tid & 63) == 0) { dwn_red[tid >> 6] = S; dwn_red[4 + (tid >> 6)] = Q; }
    __syncthreads();
    if (tid == 0) {
      unsafeAtomicAdd(stats + VPTR_FRAME_STATS_STRIDE * f, (dwn_red[0] + dwn_red[1]) + (dwn_red[2] + dwn_red[3]));
      unsafeAtomicAdd(stats + VPTR_FRAME_STATS_STRIDE * f + 1, (dwn_red[4] + dwn_red[5]) + (dwn_red[6] + dwn_red[7]));
    }
  }
}
extern "C" int vptr_dwconv3x3_norm_fwd(const float* x, const float* raw_stats, const float* aff_w, const float* aff_b, float eps, int act,
                                       const float* w9, const float* b, float* y, void* a_half, float* mean_out, float* rstd_out,
                                       int frames, int H, int W, int F, float* frame_stats, vptr_stream_t stream) {
  VPTR_CHECK(x && raw_stats && aff_w && aff_b && w9 && y && mean_out && rstd_out && frames > 0 && H > 0 && W > 0 && F > 0 && F % 4 == 0,
             "dwconv3x3_norm_fwd: bad arguments");
  const int W2 = W / 2;
  VPTR_CHECK(W % 2 == 0 && W2 >= 1 && 16 % W2 == 0 && (W2 * (F / 4)) % 64 == 0,
             "dwconv3x3_norm_fwd: needs W even, W / 2 dividing 16 and (W/2)*(F/4) %% 64 == 0 (got W %d, F %d)", W, F);
  VPTR_CHECK(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(aff_w) | reinterpret_cast<uintptr_t>(aff_b) |
               reinterpret_cast<uintptr_t>(w9) | reinterpret_cast<uintptr_t>(b)) & 15) == 0 && (reinterpret_cast<uintptr_t>(a_half) & 7) == 0,
             "dwconv3x3_norm_fwd: operands must be 16-byte aligned");
  static int use_lds = -1, dbg = 0;
  if (use_lds < 0) {
    const char* e = getenv("VPTR_DWN_LDS");
    use_lds = (e && atoi(e) == 0) ? 0 : 1;
    const char* d = getenv("VPTR_DWN_DBG");
    dbg = d ? atoi(d) : 0;
  }
  if (use_lds && F % 64 == 0 && H * W <= 256 && frames <= 65535) {   // LDS slab: H * W * 256 bytes <= 64 KB
    dwconv_norm_lds_kernel<<<dim3(F / 64, frames), 256, (size_t)H * W * 256, (hipStream_t)stream>>>(
        x, raw_stats, aff_w, aff_b, eps, act, w9, b, y, reinterpret_cast<_Float16*>(a_half), mean_out, rstd_out, H, W, F / 4, frame_stats, dbg);
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  const int64_t total = (int64_t)frames * W2 * (F / 4);
  dwconv_norm_fwd3_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      x, raw_stats, aff_w, aff_b, eps, act, w9, b, y, reinterpret_cast<_Float16*>(a_half), mean_out, rstd_out, frames, H, W, F / 4, frame_stats);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// dw9[tap, c] += sum_{f,y,x} dy[f,y,x,c] * x[f,y+ky-1,x+kx-1,c];  db[c] += sum dy.
// Block = 32 channel quads x 8 x-lanes over a chunk of frames; every thread walks its columns with the same rolling window
// (1 + 3 float4 loads per pixel), the 8 x-lanes are summed through LDS and each block issues 40 atomics per channel quad.
// CQ channel quads x (256 / CQ) x-lanes per block: 32 x 8 (rounds 1 - 5) or 16 x 16 (round 6: twice the blocks for the same atomics -- the
// K64 step's launch is 340 blocks of the 32-quad form on 256 CUs, KTH 128 x 128's 170: latency-bound, 3.6x the time for 2x the data)
template <bool PAIR, bool XH = false, int CQ = 32>  // PAIR (W even): lane = (x pair, frame parity), 4 x + 2 dy loads per two pixels instead of 6 + 2; XH: x is the fp16 side copy of dwconv_norm_fwd3_kernel
__global__ __launch_bounds__(256) void dwconv_bwd_w_kernel(const float* __restrict__ dy_, const void* __restrict__ x_,
                                                           float* __restrict__ dw9, float* __restrict__ db, int frames, int H,
                                                           int W, int F4, int fpb) {
  typedef typename std::conditional<XH, half4_t, float4>::type XT;
  static_assert(PAIR || !XH, "the fp16 operand comes with the paired form");
  constexpr int DWB_C4 = CQ, DWB_XL = 256 / CQ;
  __shared__ float red[DWB_XL * DWB_C4 * 41];
  const int cl = threadIdx.x % DWB_C4, xl = threadIdx.x / DWB_C4;
  const int c4 = blockIdx.x * DWB_C4 + cl;
  const bool cok = c4 < F4;
  const int c4c = cok ? c4 : F4 - 1;
  const XT* __restrict__ x = reinterpret_cast<const XT*>(x_);
  const float4* __restrict__ dy = reinterpret_cast<const float4*>(dy_);
  const int f0 = blockIdx.y * fpb, f1 = min(frames, f0 + fpb);
  float4 acc[9], ab = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (PAIR) {
    for (int64_t f = f0 + (xl >> 2); f < f1; f += DWB_XL / 4)
      for (int xw0 = (xl & 3) * 2; xw0 < W; xw0 += 8) {
        DwRow2 r0 = dw_load_row2(x, f * H, -1, H, xw0, W, F4, c4c), r1 = dw_load_row2(x, f * H, 0, H, xw0, W, F4, c4c);
        for (int yh = 0; yh < H; ++yh) {
          const DwRow2 r2 = dw_load_row2(x, f * H, yh + 1, H, xw0, W, F4, c4c);
          const float4 g = dy[((f * H + yh) * W + xw0) * F4 + c4c], g2 = dy[((f * H + yh) * W + xw0 + 1) * F4 + c4c];
          ab.x += g.x + g2.x; ab.y += g.y + g2.y; ab.z += g.z + g2.z; ab.w += g.w + g2.w;
          fma4(acc[0], g, r0.c0); fma4(acc[1], g, r0.c1); fma4(acc[2], g, r0.c2);
          fma4(acc[3], g, r1.c0); fma4(acc[4], g, r1.c1); fma4(acc[5], g, r1.c2);
          fma4(acc[6], g, r2.c0); fma4(acc[7], g, r2.c1); fma4(acc[8], g, r2.c2);
          fma4(acc[0], g2, r0.c1); fma4(acc[1], g2, r0.c2); fma4(acc[2], g2, r0.c3);
          fma4(acc[3], g2, r1.c1); fma4(acc[4], g2, r1.c2); fma4(acc[5], g2, r1.c3);
          fma4(acc[6], g2, r2.c1); fma4(acc[7], g2, r2.c2); fma4(acc[8], g2, r2.c3);
          r0 = r1;
          r1 = r2;
        }
      }
  } else if constexpr (!XH)
  for (int64_t f = f0; f < f1; ++f)
    for (int xw = xl; xw < W; xw += DWB_XL) {
      DwRow r0 = dw_load_row(x, f * H, -1, H, xw, W, F4, c4c), r1 = dw_load_row(x, f * H, 0, H, xw, W, F4, c4c);
      for (int yh = 0; yh < H; ++yh) {
        const DwRow r2 = dw_load_row(x, f * H, yh + 1, H, xw, W, F4, c4c);
        const float4 g = dy[((f * H + yh) * W + xw) * F4 + c4c];
        ab.x += g.x; ab.y += g.y; ab.z += g.z; ab.w += g.w;
        fma4(acc[0], g, r0.l); fma4(acc[1], g, r0.m); fma4(acc[2], g, r0.r);
        fma4(acc[3], g, r1.l); fma4(acc[4], g, r1.m); fma4(acc[5], g, r1.r);
        fma4(acc[6], g, r2.l); fma4(acc[7], g, r2.m); fma4(acc[8], g, r2.r);
        r0 = r1;
        r1 = r2;
      }
    }
  float* mine = red + (xl * DWB_C4 + cl) * 41;  // 41-float pitch: conflict-free column sums below
#pragma unroll
  for (int t = 0; t < 9; ++t) { mine[t * 4 + 0] = acc[t].x; mine[t * 4 + 1] = acc[t].y; mine[t * 4 + 2] = acc[t].z; mine[t * 4 + 3] = acc[t].w; }
  mine[36] = ab.x; mine[37] = ab.y; mine[38] = ab.z; mine[39] = ab.w;
  __syncthreads();
  const int F = F4 * 4;
  for (int o = threadIdx.x; o < DWB_C4 * 40; o += 256) {
    const int ocl = o / 40, k = o - ocl * 40;
    const int oc4 = blockIdx.x * DWB_C4 + ocl;
    if (oc4 >= F4) continue;
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < DWB_XL; ++q) sum += red[(q * DWB_C4 + ocl) * 41 + k];
    const int comp = k & 3, tap = k >> 2;
    if (tap < 9) unsafeAtomicAdd(dw9 + (int64_t)tap * F + oc4 * 4 + comp, sum);
    else unsafeAtomicAdd(db + oc4 * 4 + comp, sum);
  }
}

static bool dw_v3(int W, int64_t total) {   // VPTR_DWCONV_GEN=2 restores dwconv_fwd2_kernel
  static int gen = -1;
  if (gen < 0) {
    const char* e = getenv("VPTR_DWCONV_GEN");
    gen = e ? atoi(e) : 3;
  }
  const int W2 = W / 2;
  return gen >= 3 && W % 2 == 0 && W2 >= 1 && 16 % W2 == 0 && total % 2 == 0;
}
extern "C" int vptr_dwconv3x3_fwd(const float* x, const float* w9, const float* b, float* y, int frames, int H, int W, int F,
                                  float* frame_stats, vptr_stream_t stream) {
  VPTR_CHECK(frames > 0 && H > 0 && W > 0 && F > 0 && F % 4 == 0, "dwconv3x3_fwd: bad arguments");
  const int64_t total = (int64_t)frames * W * (F / 4);
  if (frame_stats) {   // no silent fallback: the caller asks for statistics only where this kernel can give them (vptr_amd/ops.py)
    VPTR_CHECK(W % 2 == 0 && ((W / 2) * (F / 4)) % 64 == 0, "dwconv3x3_fwd: frame_stats needs W even and (W/2)*(F/4) %% 64 == 0");
    if (dw_v3(W, total)) dwconv_fwd3_kernel<<<(unsigned)((total / 2 + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, w9, b, y, frames, H, W, F / 4, 0, frame_stats);
    else dwconv_fwd2_kernel<<<(unsigned)((total / 2 + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, w9, b, y, frames, H, W, F / 4, 0, frame_stats);
  } else if (W % 2 == 0 && total >= (1 << 16)) {
    if (dw_v3(W, total)) dwconv_fwd3_kernel<<<(unsigned)((total / 2 + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, w9, b, y, frames, H, W, F / 4, 0, nullptr);
    else dwconv_fwd2_kernel<<<(unsigned)((total / 2 + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, w9, b, y, frames, H, W, F / 4, 0, nullptr);
  }
  else
    dwconv_fwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, w9, b, y, frames, H, W, F / 4, 0);
  VPTR_LAUNCH_CHECK();
  return 0;
}

static int dwconv3x3_bwd_impl(const float* dy, const void* x, int x_half, const float* w9, float* dx, float* dw9, float* db,
                              int frames, int H, int W, int F, vptr_stream_t stream);
extern "C" int vptr_dwconv3x3_bwd(const float* dy, const float* x, const float* w9, float* dx, float* dw9, float* db,
                                  int frames, int H, int W, int F, vptr_stream_t stream) {
  return dwconv3x3_bwd_impl(dy, x, 0, w9, dx, dw9, db, frames, H, W, F, stream);
}
// the same with the forward input given as the fp16 side copy vptr_dwconv3x3_norm_fwd wrote (W even)
extern "C" int vptr_dwconv3x3_bwd_xh(const float* dy, const void* x_half, const float* w9, float* dx, float* dw9, float* db,
                                     int frames, int H, int W, int F, vptr_stream_t stream) {
  VPTR_CHECK(W % 2 == 0 && (reinterpret_cast<uintptr_t>(x_half) & 7) == 0, "dwconv3x3_bwd_xh: needs W even and an 8-byte aligned fp16 operand");
  return dwconv3x3_bwd_impl(dy, x_half, 1, w9, dx, dw9, db, frames, H, W, F, stream);
}
static int dwconv3x3_bwd_impl(const float* dy, const void* x, int x_half, const float* w9, float* dx, float* dw9, float* db,
                              int frames, int H, int W, int F, vptr_stream_t stream) {
  VPTR_CHECK(frames > 0 && H > 0 && W > 0 && F > 0 && F % 4 == 0, "dwconv3x3_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)frames * W * (F / 4);
  if (dx) {
    if (W % 2 == 0 && total >= (1 << 16) && dw_v3(W, total))
      dwconv_fwd3_kernel<<<(unsigned)((total / 2 + 255) / 256), 256, 0, st>>>(dy, w9, nullptr, dx, frames, H, W, F / 4, 1, nullptr);
    else if (W % 2 == 0 && total >= (1 << 16))
      dwconv_fwd2_kernel<<<(unsigned)((total / 2 + 255) / 256), 256, 0, st>>>(dy, w9, nullptr, dx, frames, H, W, F / 4, 1, nullptr);
    else
      dwconv_fwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(dy, w9, nullptr, dx, frames, H, W, F / 4, 1);
  }
  if (dw9 && db) {
    static int fpb_env = -1;
    if (fpb_env < 0) { const char* e = getenv("VPTR_DWB_FPB"); fpb_env = (e && atoi(e) > 0) ? atoi(e) : 8; }
    const int fpb = g_vptr_deterministic ? frames : (frames >= 64 ? fpb_env : 1);   // deterministic: one adder per tap and channel
    static int cq16 = -1, cq8 = 0;
    if (cq16 < 0) { const char* e = getenv("VPTR_DWB_CQ"); cq16 = (e && atoi(e) == 32) ? 0 : 1; cq8 = (e && atoi(e) == 8) ? 1 : 0; }
    // 16-quad blocks (paired forms) while the 32-quad grid would leave CUs idle
    const bool narrow = cq16 && !g_vptr_deterministic && fpb >= 4 && cdiv(F / 4, 32) * cdiv(frames, fpb) < 1024;
    if (x_half && narrow && cq8)   // experiment (VPTR_DWB_CQ=8): 8 quads x 32 lanes
      dwconv_bwd_w_kernel<true, true, 8><<<dim3(cdiv(F / 4, 8), cdiv(frames, fpb)), 256, 0, st>>>(dy, x, dw9, db, frames, H, W, F / 4, fpb);
    else if (x_half && narrow)
      dwconv_bwd_w_kernel<true, true, 16><<<dim3(cdiv(F / 4, 16), cdiv(frames, fpb)), 256, 0, st>>>(dy, x, dw9, db, frames, H, W, F / 4, fpb);
    else if (x_half)
      dwconv_bwd_w_kernel<true, true><<<dim3(cdiv(F / 4, 32), cdiv(frames, fpb)), 256, 0, st>>>(dy, x, dw9, db, frames, H, W, F / 4, fpb);
    else if (W % 2 == 0 && narrow)
      dwconv_bwd_w_kernel<true, false, 16><<<dim3(cdiv(F / 4, 16), cdiv(frames, fpb)), 256, 0, st>>>(dy, x, dw9, db, frames, H, W, F / 4, fpb);
    else if (W % 2 == 0)
      dwconv_bwd_w_kernel<true><<<dim3(cdiv(F / 4, 32), cdiv(frames, fpb)), 256, 0, st>>>(dy, x, dw9, db, frames, H, W, F / 4, fpb);
    else
      dwconv_bwd_w_kernel<false><<<dim3(cdiv(F / 4, 32), cdiv(frames, fpb)), 256, 0, st>>>(dy, x, dw9, db, frames, H, W, F / 4, fpb);
  }
  VPTR_LAUNCH_CHECK();
  return 0;
}
