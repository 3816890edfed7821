// Attention cores of the VidHRFormer blocks (gfx950): local-window attention with relative-position bias and
// per-pixel temporal attention.  Problems are tiny (16x16 or 64x64 windows, T<=50 time steps, head_dim 66), i.e.
// 0.4 % of the model FLOPs, so they run in exact fp32 on the vector ALUs with every tile staged once through LDS;
// the window partition / (T, N*HW, C) permutes of the reference are pure index arithmetic here.
#include "attn_mfma.h"

#define ATT_MAXL 64   // max tokens per window (ws <= 8)
#define ATT_MAXT 64   // max time steps

// ------------------------------------------------------------------------------------------------------------
// window attention forward: block per (window, head)
// ------------------------------------------------------------------------------------------------------------
// scalar output element of an fp32 (p16 = 0) or P16 (p16 != 0: the tensor only feeds GEMMs) tensor
__device__ __forceinline__ void store1(float* __restrict__ dst, const int64_t e, const float v, const int p16) {
  if (p16) vptr_p16_store1(reinterpret_cast<unsigned char*>(dst), e, v);
  else dst[e] = v;
}

__device__ __forceinline__ int win_row(int win, int l, int H, int W, int ws, int nqh, int nqw) {
  const int b = win / (nqh * nqw), r = win - b * (nqh * nqw);
  const int qh = r / nqw, qw = r - qh * nqw;
  const int ph = l / ws, pw = l - ph * ws;
  return (b * H + qh * ws + ph) * W + qw * ws + pw;
}

__global__ __launch_bounds__(256) void winattn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, const float* __restrict__ table,
                                                          const int64_t* __restrict__ rel_index, float* __restrict__ o, int H,
                                                          int W, int C, int nh, int ws, float p, const uint64_t* seed_dev,
                                                          uint32_t site, int p16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = ws * ws, hd = C / nh, hp = hd + 1, Lp = L + 1;
  float* sq = smem;            // [L][hp]
  float* sk = sq + L * hp;     // [L][hp]
  float* sv = sk + L * hp;     // [L][hp]
  float* ss = sv + L * hp;     // [L][Lp]
  __shared__ int srow[ATT_MAXL];
  const int win = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int nqh = H / ws, nqw = W / ws;
  if (tid < L) srow[tid] = win_row(win, tid, H, W, ws, nqh, nqw);
  __syncthreads();
  for (int i = tid; i < L * hd; i += 256) {
    const int l = i / hd, d = i - l * hd;
    const int64_t g = (int64_t)srow[l] * C + h * hd + d;
    sq[l * hp + d] = q[g];
    sk[l * hp + d] = k[g];
    sv[l * hp + d] = v[g];
  }
  __syncthreads();
  for (int e = tid; e < L * L; e += 256) {
    const int i = e / L, j = e - i * L;
    float a = 0.f;
    for (int d = 0; d < hd; ++d) a += sq[i * hp + d] * sk[j * hp + d];
    if (table) a += table[rel_index[e] * nh + h];
    ss[i * Lp + j] = a;
  }
  __syncthreads();
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  if (tid < L) {
    float m = -INFINITY;
    for (int j = 0; j < L; ++j) m = fmaxf(m, ss[tid * Lp + j]);
    float s = 0.f;
    for (int j = 0; j < L; ++j) { const float e = __expf(ss[tid * Lp + j] - m); ss[tid * Lp + j] = e; s += e; }
    const float inv = 1.f / s;
    for (int j = 0; j < L; ++j) {
      float pr = ss[tid * Lp + j] * inv;
      if (p > 0.f) pr *= vptr_drop_scale(seed, site, ((uint64_t)(win * nh + h) * L + tid) * L + j, p);
      ss[tid * Lp + j] = pr;
    }
  }
  __syncthreads();
  for (int e = tid; e < L * hd; e += 256) {
    const int i = e / hd, d = e - i * hd;
    float a = 0.f;
    for (int j = 0; j < L; ++j) a += ss[i * Lp + j] * sv[j * hp + d];
    store1(o, (int64_t)srow[i] * C + h * hd + d, a, p16);
  }
}

// ------------------------------------------------------------------------------------------------------------
// fast paths for 16-token problems (4x4 windows; T <= 16 time steps further down).  The 16 x 16 score matrix is one
// element per thread (window) or per lane-round (time), so the softmax reductions (max, sum, and sum(dP * P) in the
// backward) are 16-lane shuffles instead of a serial loop on 16 threads; global loads are float2; the LDS tiles have a
// 16-byte-aligned pitch of an ODD number of float4 (68 floats for head dim 66, zero padded), which makes the 16 row reads
// of a score dot product conflict-free ds_read_b128 -- the first versions of these kernels were bound by ds_read_b32 count.
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int att_pitch(int hd) {
  int q = (hd + 3) / 4;
  if ((q & 1) == 0) ++q;
  return q * 4;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
  v = fmaxf(v, __shfl_xor(v, 8, 64));
  return v;
}
__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
__device__ __forceinline__ void axpy4(float4& y, const float a, const float4 x) { y.x += a * x.x; y.y += a * x.y; y.z += a * x.z; y.w += a * x.w; }
// rows[l] * C + hoff .. + hd of a token-major tensor -> LDS [nrows][hp] (columns hd .. hp-1 zeroed); hd even
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, float* dst, const int64_t* rowoff, int nrows, int hoff, int hd,
                                           int hp, int tid, int nthreads) {
  const int h2 = hp >> 1, v2 = hd >> 1;
  for (int e = tid; e < nrows * h2; e += nthreads) {
    const int l = e / h2, d2 = e - l * h2;
    float2 v = make_float2(0.f, 0.f);
    if (d2 < v2) v = *reinterpret_cast<const float2*>(src + rowoff[l] + hoff + 2 * d2);
    *reinterpret_cast<float2*>(dst + l * hp + 2 * d2) = v;
  }
}
// The same in two halves for a 256-thread workgroup and 16-row tiles: the global loads of the NEXT tile are issued into
// registers before the current tile is computed and written to LDS afterwards (the kernels spent ~70 % of their wave
// cycles waiting: load -> barrier -> compute -> store per window, 4-5 workgroups per CU deep).
struct Rows16 {
  float2 v[3];  // items tid, tid + 256, tid + 512 of the 16 x (hp / 2) float2 tile
};
__device__ __forceinline__ void rows16_load(Rows16& r, const float* __restrict__ src, int win, int H, int W, int C, int nqh, int nqw,
                                            int hoff, int hd, int hp, int tid) {
  const int h2 = hp >> 1, v2 = hd >> 1;
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int e = tid + it * 256;
    const int l = e / h2, d2 = e - l * h2;
    r.v[it] = make_float2(0.f, 0.f);
    if (l < 16 && d2 < v2) r.v[it] = *reinterpret_cast<const float2*>(src + (int64_t)win_row(win, l, H, W, 4, nqh, nqw) * C + hoff + 2 * d2);
  }
}
__device__ __forceinline__ void rows16_store(const Rows16& r, float* dst, int hp, int tid) {
  const int h2 = hp >> 1;
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int e = tid + it * 256;
    const int l = e / h2, d2 = e - l * h2;
    if (l < 16) *reinterpret_cast<float2*>(dst + l * hp + 2 * d2) = r.v[it];
  }
}
// out rows: dst[rowoff[r] + hoff + d] = acc (hd even; float2 stores)
// p16: dst is a P16 tensor (the output only feeds GEMMs): base + d is even, so a channel pair never straddles a 16-channel granule
__device__ __forceinline__ void store4(float* __restrict__ dst, int64_t base, int d, int hd, const float4 v, const int p16) {
  if (p16) {
    vptr_p16_store2(reinterpret_cast<unsigned char*>(dst), base + d, v.x, v.y);
    if (d + 2 < hd) vptr_p16_store2(reinterpret_cast<unsigned char*>(dst), base + d + 2, v.z, v.w);
    return;
  }
  *reinterpret_cast<float2*>(dst + base + d) = make_float2(v.x, v.y);
  if (d + 2 < hd) *reinterpret_cast<float2*>(dst + base + d + 2) = make_float2(v.z, v.w);
}


__global__ __launch_bounds__(256) void winattn16_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const float* __restrict__ table,
                                                            const int64_t* __restrict__ rel_index, float* __restrict__ o,
                                                            int nwin, int H, int W, int C, int nh, float p,
                                                            const uint64_t* seed_dev, uint32_t site, int wpb, int p16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hd = C / nh, hp = att_pitch(hd), n4 = hp >> 2;
  float* sq = smem;             // [16][hp]
  float* sk = sq + 16 * hp;
  float* sv = sk + 16 * hp;
  float* sp = sv + 16 * hp;     // [16][20]
  __shared__ int64_t srow[16];
  const int h = blockIdx.y, tid = threadIdx.x, i = tid >> 4, j = tid & 15;
  const int nqh = H / 4, nqw = W / 4;
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const float bias = table ? table[rel_index[tid] * nh + h] : 0.f;   // element (i, j) of the relative-position bias
  const int w0 = blockIdx.x * wpb, w1 = min(nwin, w0 + wpb);
  Rows16 rq, rk, rv;
  if (w0 < w1) {
    rows16_load(rq, q, w0, H, W, C, nqh, nqw, h * hd, hd, hp, tid);
    rows16_load(rk, k, w0, H, W, C, nqh, nqw, h * hd, hd, hp, tid);
    rows16_load(rv, v, w0, H, W, C, nqh, nqw, h * hd, hd, hp, tid);
  }
  for (int win = w0; win < w1; ++win) {
    __syncthreads();
    if (tid < 16) srow[tid] = (int64_t)win_row(win, tid, H, W, 4, nqh, nqw) * C;
    rows16_store(rq, sq, hp, tid);
    rows16_store(rk, sk, hp, tid);
    rows16_store(rv, sv, hp, tid);
    if (win + 1 < w1) {  // workgroup-uniform: next window's rows fly while this one is computed
      rows16_load(rq, q, win + 1, H, W, C, nqh, nqw, h * hd, hd, hp, tid);
      rows16_load(rk, k, win + 1, H, W, C, nqh, nqw, h * hd, hd, hp, tid);
      rows16_load(rv, v, win + 1, H, W, C, nqh, nqw, h * hd, hd, hp, tid);
    }
    __syncthreads();
    float a = bias;
    for (int d = 0; d < n4; ++d)
      a += dot4(reinterpret_cast<const float4*>(sq + i * hp)[d], reinterpret_cast<const float4*>(sk + j * hp)[d]);
    const float m = row16_max(a);
    const float e = __expf(a - m);
    float pr = e / row16_sum(e);
    if (p > 0.f) pr *= vptr_drop_scale(seed, site, ((uint64_t)(win * nh + h) * 16 + i) * 16 + j, p);
    sp[i * 20 + j] = pr;
    __syncthreads();
    for (int e2 = tid; e2 < 16 * n4; e2 += 256) {
      const int r = e2 / n4, d4 = e2 - r * n4;
      if (d4 * 4 >= hd) continue;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
      for (int c = 0; c < 16; ++c) axpy4(acc, sp[r * 20 + c], reinterpret_cast<const float4*>(sv + c * hp)[d4]);
      store4(o, srow[r] + h * hd, d4 * 4, hd, acc, p16);
    }
  }
}

__global__ __launch_bounds__(256) void winattn16_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const float* __restrict__ table,
                                                            const int64_t* __restrict__ rel_index, const float* __restrict__ dout,
                                                            float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                            float* __restrict__ dtable, int nwin, int H, int W, int C, int nh,
                                                            float p, const uint64_t* seed_dev, uint32_t site, int wpb,
                                                            float dq_scale, int p16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hd = C / nh, hp = att_pitch(hd), n4 = hp >> 2;
  float* sq = smem;             // [16][hp]
  float* sk = sq + 16 * hp;
  float* sv = sk + 16 * hp;
  float* sdo = sv + 16 * hp;
  float* sp = sdo + 16 * hp;    // [16][20]  dropped probabilities (for dV)
  float* sds = sp + 16 * 20;    // [16][20]  dS
  __shared__ int64_t srow[16];
  const int h = blockIdx.y, tid = threadIdx.x, i = tid >> 4, j = tid & 15;
  const int nqh = H / 4, nqw = W / 4;
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const int ridx = (table || dtable) ? (int)rel_index[tid] : 0;
  const float bias = table ? table[ridx * nh + h] : 0.f;
  float dbias = 0.f;  // this thread's (i, j) element of dS summed over the workgroup's windows
  const int w0 = blockIdx.x * wpb, w1 = min(nwin, w0 + wpb);
  Rows16 rq, rk, rv, rdo;
  if (w0 < w1) {
    rows16_load(rq, q, w0, H, W, C, nqh, nqw, h * hd, hd, hp, tid);
    rows16_load(rk, k, w0, H, W, C, nqh, nqw, h * hd, hd, hp, tid);
    rows16_load(rv, v, w0, H, W, C, nqh, nqw, h * hd, hd, hp, tid);
    rows16_load(rdo, dout, w0, H, W, C, nqh, nqw, h * hd, hd, hp, tid);
  }
  for (int win = w0; win < w1; ++win) {
    __syncthreads();
    if (tid < 16) srow[tid] = (int64_t)win_row(win, tid, H, W, 4, nqh, nqw) * C;
    rows16_store(rq, sq, hp, tid);
    rows16_store(rk, sk, hp, tid);
    rows16_store(rv, sv, hp, tid);
    rows16_store(rdo, sdo, hp, tid);
    if (win + 1 < w1) {  // workgroup-uniform: next window's rows fly while this one is computed
      rows16_load(rq, q, win + 1, H, W, C, nqh, nqw, h * hd, hd, hp, tid);
      rows16_load(rk, k, win + 1, H, W, C, nqh, nqw, h * hd, hd, hp, tid);
      rows16_load(rv, v, win + 1, H, W, C, nqh, nqw, h * hd, hd, hp, tid);
      rows16_load(rdo, dout, win + 1, H, W, C, nqh, nqw, h * hd, hd, hp, tid);
    }
    __syncthreads();
    float a = bias, b = 0.f;
    for (int d = 0; d < n4; ++d) {
      a += dot4(reinterpret_cast<const float4*>(sq + i * hp)[d], reinterpret_cast<const float4*>(sk + j * hp)[d]);
      b += dot4(reinterpret_cast<const float4*>(sdo + i * hp)[d], reinterpret_cast<const float4*>(sv + j * hp)[d]);
    }
    const float m = row16_max(a);
    const float e = __expf(a - m);
    const float pr = e / row16_sum(e);
    const float sc = p > 0.f ? vptr_drop_scale(seed, site, ((uint64_t)(win * nh + h) * 16 + i) * 16 + j, p) : 1.f;
    const float dpr = b * sc;                       // gradient w.r.t. the softmax probability
    const float ds = pr * (dpr - row16_sum(dpr * pr));
    dbias += ds;
    sp[i * 20 + j] = pr * sc;
    sds[i * 20 + j] = ds;
    __syncthreads();
    for (int e2 = tid; e2 < 16 * n4; e2 += 256) {
      const int r = e2 / n4, d4 = e2 - r * n4;
      if (d4 * 4 >= hd) continue;
      float4 aq = make_float4(0.f, 0.f, 0.f, 0.f), ak = aq, av = aq;
#pragma unroll 4
      for (int c = 0; c < 16; ++c) {
        axpy4(aq, sds[r * 20 + c], reinterpret_cast<const float4*>(sk + c * hp)[d4]);
        axpy4(ak, sds[c * 20 + r], reinterpret_cast<const float4*>(sq + c * hp)[d4]);
        axpy4(av, sp[c * 20 + r], reinterpret_cast<const float4*>(sdo + c * hp)[d4]);
      }
      const int64_t g = srow[r] + h * hd;
      store4(dq, g, d4 * 4, hd, make_float4(aq.x * dq_scale, aq.y * dq_scale, aq.z * dq_scale, aq.w * dq_scale), p16);
      store4(dk, g, d4 * 4, hd, ak, p16);
      store4(dv, g, d4 * 4, hd, av, p16);
    }
  }
  if (dtable) {  // 256 (i, j) elements -> 49 table entries: LDS atomics, then one global atomic per entry
    __syncthreads();
    float* stab = sp;  // 49 <= 16 * 20
    if (tid < 49) stab[tid] = 0.f;
    __syncthreads();
    atomicAdd(&stab[ridx], dbias);
    __syncthreads();
    if (tid < 49) unsafeAtomicAdd(dtable + (int64_t)tid * nh + h, stab[tid]);
  }
}

extern "C" int vptr_winattn_fwd(const float* q, const float* k, const float* v, const float* bias_table,
                                const int64_t* rel_index, float* o, int B, int H, int W, int C, int nh, int ws,
                                float dropout_p, const uint64_t* seed_dev, uint32_t site, int p16, vptr_stream_t stream) {
  if (p16) VPTR_CHECK(C % 16 == 0, "winattn_fwd: P16 outputs need C %% 16 == 0 (got %d)", C);
  VPTR_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && nh > 0 && ws > 0, "winattn_fwd: bad arguments");
  VPTR_CHECK(C % nh == 0, "winattn_fwd: embed_dim must be divisible by num_heads");
  VPTR_CHECK(H % ws == 0 && W % ws == 0, "winattn_fwd: H, W must be multiples of the window size (pad on the host)");
  VPTR_CHECK(ws * ws <= ATT_MAXL, "winattn_fwd: window too large (ws*ws <= %d)", ATT_MAXL);
  if (bias_table) VPTR_CHECK(rel_index != nullptr, "winattn_fwd: bias table needs rel_index");
  if (dropout_p > 0.f) VPTR_CHECK(seed_dev && dropout_p < 1.f, "winattn_fwd: dropout needs seed_dev");
  const int L = ws * ws, hd = C / nh;
  if (vptr_attn_mfma_ok(L, L, C, nh, 0, (int64_t)B * (H / ws) * (W / ws))) {
    AmGeom gm = {0, H, W, ws, 0, 0, 0, L, L, C, nh, hd, B * (H / ws) * (W / ws)};
    const int rc = vptr_attn_mfma_fwd(q, k, v, bias_table, rel_index, o, gm, 0, dropout_p, seed_dev, site, p16, (hipStream_t)stream);
    if (rc) return rc;
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  if (vptr_attn16_ok(0, L, L, C, nh, ws, 0)) {   // 4 x 4 windows: LDS-free MFMA kernels (attn16.hip)
    A16Geom g16 = {0, H, W, 0, 0, 0, C, nh, hd, 16, 16, B * (H / 4) * (W / 4), 0};
    const int rc = vptr_attn16_fwd(q, k, v, bias_table, rel_index, o, g16, dropout_p, seed_dev, site, p16, (hipStream_t)stream);
    if (rc) return rc;
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  if (ws == 4 && hd % 2 == 0 && C % 2 == 0) {
    const int nwin = B * (H / 4) * (W / 4);
    const int wpb = nwin >= 64 ? 2 : 1;   // two windows per workgroup: the second one's loads overlap the first one's math
    const size_t lds16 = sizeof(float) * (3 * 16 * att_pitch(hd) + 16 * 20);
    winattn16_fwd_kernel<<<dim3(cdiv(nwin, wpb), nh), 256, lds16, (hipStream_t)stream>>>(q, k, v, bias_table, rel_index, o, nwin, H, W, C,
                                                                                         nh, dropout_p, seed_dev, site, wpb, p16);
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  const size_t lds = sizeof(float) * (3 * L * (hd + 1) + L * (L + 1));
  VPTR_CHECK(lds <= 160 * 1024, "winattn_fwd: LDS budget exceeded (%zu B)", lds);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)winattn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid(B * (H / ws) * (W / ws), nh);
  winattn_fwd_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(q, k, v, bias_table, rel_index, o, H, W, C, nh, ws, dropout_p,
                                                             seed_dev, site, p16);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// window attention backward: block per (chunk of windows, head); bias-table gradient reduced in LDS first.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void winattn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, const float* __restrict__ table,
                                                          const int64_t* __restrict__ rel_index, const float* __restrict__ dout,
                                                          float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                          float* __restrict__ dtable, int nwin, int H, int W, int C, int nh,
                                                          int ws, float p, const uint64_t* seed_dev, uint32_t site, int wpb,
                                                          float dq_scale, int p16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = ws * ws, hd = C / nh, hp = hd + 1, Lp = L + 1;
  const int ntab = (2 * ws - 1) * (2 * ws - 1);
  float* sq = smem;
  float* sk = sq + L * hp;
  float* sv = sk + L * hp;
  float* sdo = sv + L * hp;   // [L][hp]
  float* sp = sdo + L * hp;   // [L][Lp]  probabilities (incl. dropout scale)
  float* sds = sp + L * Lp;   // [L][Lp]  dP then dS
  float* stab = sds + L * Lp; // [ntab]
  __shared__ int srow[ATT_MAXL];
  const int h = blockIdx.y, tid = threadIdx.x;
  const int nqh = H / ws, nqw = W / ws;
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  for (int i = tid; i < ntab; i += 256) stab[i] = 0.f;
  const int w0 = blockIdx.x * wpb, w1 = min(nwin, w0 + wpb);
  for (int win = w0; win < w1; ++win) {
    __syncthreads();
    if (tid < L) srow[tid] = win_row(win, tid, H, W, ws, nqh, nqw);
    __syncthreads();
    for (int i = tid; i < L * hd; i += 256) {
      const int l = i / hd, d = i - l * hd;
      const int64_t g = (int64_t)srow[l] * C + h * hd + d;
      sq[l * hp + d] = q[g];
      sk[l * hp + d] = k[g];
      sv[l * hp + d] = v[g];
      sdo[l * hp + d] = dout[g];
    }
    __syncthreads();
    for (int e = tid; e < L * L; e += 256) {
      const int i = e / L, j = e - i * L;
      float a = 0.f, b = 0.f;
      for (int d = 0; d < hd; ++d) {
        a += sq[i * hp + d] * sk[j * hp + d];
        b += sdo[i * hp + d] * sv[j * hp + d];
      }
      if (table) a += table[rel_index[e] * nh + h];
      sp[i * Lp + j] = a;
      sds[i * Lp + j] = b;  // dP (w.r.t. the dropped probabilities)
    }
    __syncthreads();
    if (tid < L) {
      float m = -INFINITY;
      for (int j = 0; j < L; ++j) m = fmaxf(m, sp[tid * Lp + j]);
      float s = 0.f;
      for (int j = 0; j < L; ++j) { const float e = __expf(sp[tid * Lp + j] - m); sp[tid * Lp + j] = e; s += e; }
      const float inv = 1.f / s;
      float dot = 0.f;
      for (int j = 0; j < L; ++j) {
        const float pr = sp[tid * Lp + j] * inv;  // softmax prob
        float sc = 1.f;
        if (p > 0.f) sc = vptr_drop_scale(seed, site, ((uint64_t)(win * nh + h) * L + tid) * L + j, p);
        const float dpr = sds[tid * Lp + j] * sc;  // grad wrt softmax prob
        dot += dpr * pr;
        sds[tid * Lp + j] = dpr;
        sp[tid * Lp + j] = pr;
      }
      for (int j = 0; j < L; ++j) {
        const float pr = sp[tid * Lp + j];
        float sc = 1.f;
        if (p > 0.f) sc = vptr_drop_scale(seed, site, ((uint64_t)(win * nh + h) * L + tid) * L + j, p);
        sds[tid * Lp + j] = pr * (sds[tid * Lp + j] - dot);  // dS
        sp[tid * Lp + j] = pr * sc;                          // dropped prob for dV
      }
    }
    __syncthreads();
    if (dtable)
      for (int e = tid; e < L * L; e += 256) atomicAdd(&stab[(int)rel_index[e]], sds[(e / L) * Lp + (e % L)]);
    for (int e = tid; e < L * hd; e += 256) {
      const int i = e / hd, d = e - i * hd;
      float aq = 0.f, ak = 0.f, av = 0.f;
      for (int j = 0; j < L; ++j) {
        aq += sds[i * Lp + j] * sk[j * hp + d];
        ak += sds[j * Lp + i] * sq[j * hp + d];
        av += sp[j * Lp + i] * sdo[j * hp + d];
      }
      const int64_t g = (int64_t)srow[i] * C + h * hd + d;
      store1(dq, g, aq * dq_scale, p16);
      store1(dk, g, ak, p16);
      store1(dv, g, av, p16);
    }
  }
  __syncthreads();
  if (dtable)
    for (int i = tid; i < ntab; i += 256) unsafeAtomicAdd(dtable + (int64_t)i * nh + h, stab[i]);
}

extern "C" int vptr_winattn_bwd_workspace(int nh) { return (int)vptr_attn16_ws_floats(nh); }
extern "C" int vptr_winattn_bwd(const float* q, const float* k, const float* v, const float* bias_table,
                                const int64_t* rel_index, const float* dout, float* dq, float* dk, float* dv,
                                float* dbias_table, int B, int H, int W, int C, int nh, int ws, float dropout_p,
                                const uint64_t* seed_dev, uint32_t site, float dq_scale, int p16, vptr_stream_t stream) {
  return vptr_winattn_bwd_ws(q, k, v, bias_table, rel_index, dout, dq, dk, dv, dbias_table, B, H, W, C, nh, ws, dropout_p, seed_dev, site, dq_scale, p16,
                             nullptr, 0, stream);
}
extern "C" int vptr_winattn_bwd_ws(const float* q, const float* k, const float* v, const float* bias_table,
                                   const int64_t* rel_index, const float* dout, float* dq, float* dk, float* dv,
                                   float* dbias_table, int B, int H, int W, int C, int nh, int ws, float dropout_p,
                                   const uint64_t* seed_dev, uint32_t site, float dq_scale, int p16, float* workspace,
                                   int workspace_floats, vptr_stream_t stream) {
  if (workspace) VPTR_CHECK((reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && workspace_floats > 0, "winattn_bwd: bad workspace");
  if (p16) VPTR_CHECK(C % 16 == 0, "winattn_bwd: P16 outputs need C %% 16 == 0 (got %d)", C);
  VPTR_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && nh > 0 && ws > 0, "winattn_bwd: bad arguments");
  VPTR_CHECK(C % nh == 0 && H % ws == 0 && W % ws == 0 && ws * ws <= ATT_MAXL, "winattn_bwd: unsupported geometry");
  if (bias_table || dbias_table) VPTR_CHECK(rel_index != nullptr, "winattn_bwd: bias table needs rel_index");
  if (dropout_p > 0.f) VPTR_CHECK(seed_dev && dropout_p < 1.f, "winattn_bwd: dropout needs seed_dev");
  const int L = ws * ws, hd = C / nh, ntab = (2 * ws - 1) * (2 * ws - 1);
  const int nwin = B * (H / ws) * (W / ws);
  if (vptr_attn_mfma_ok(L, L, C, nh, 0, (int64_t)B * (H / ws) * (W / ws))) {
    AmGeom gm = {0, H, W, ws, 0, 0, 0, L, L, C, nh, hd, nwin};
    const int rc = vptr_attn_mfma_bwd(q, k, v, bias_table, rel_index, dout, dq, dk, dv, dbias_table, gm, 0, dropout_p, seed_dev, site, dq_scale, p16,
                                      (hipStream_t)stream);
    if (rc) return rc;
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  if (vptr_attn16_ok(0, L, L, C, nh, ws, 1)) {
    A16Geom g16 = {0, H, W, 0, 0, 0, C, nh, hd, 16, 16, nwin, 0};
    const int rc = vptr_attn16_bwd(q, k, v, bias_table, rel_index, dout, dq, dk, dv, dbias_table, g16, dropout_p, seed_dev, site, dq_scale, p16,
                                   workspace, workspace_floats, (hipStream_t)stream);
    if (rc) return rc;
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  if (ws == 4 && hd % 2 == 0 && C % 2 == 0) {
    // many windows per workgroup: the 49 bias-table atomics per workgroup hit the same 49*nh addresses from every workgroup
    // with a bias-table gradient, fewer and longer workgroups (their 49 atomics per head all hit the same 392 words)
    const int wpb16 = nwin >= 512 ? (dbias_table ? 8 : 4) : (nwin >= 64 ? 2 : 1);
    const size_t lds16 = sizeof(float) * (4 * 16 * att_pitch(hd) + 2 * 16 * 20);
    winattn16_bwd_kernel<<<dim3(cdiv(nwin, wpb16), nh), 256, lds16, (hipStream_t)stream>>>(q, k, v, bias_table, rel_index, dout, dq, dk,
                                                                                          dv, dbias_table, nwin, H, W, C, nh,
                                                                                          dropout_p, seed_dev, site, wpb16, dq_scale, p16);
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  const size_t lds = sizeof(float) * (4 * L * (hd + 1) + 2 * L * (L + 1) + ntab);
  VPTR_CHECK(lds <= 160 * 1024, "winattn_bwd: LDS budget exceeded (%zu B)", lds);
  const int wpb = nwin >= 2048 ? 4 : 1;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)winattn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  winattn_bwd_kernel<<<dim3(cdiv(nwin, wpb), nh), 256, lds, (hipStream_t)stream>>>(q, k, v, bias_table, rel_index, dout, dq, dk, dv,
                                                                                 dbias_table, nwin, H, W, C, nh, ws, dropout_p,
                                                                                 seed_dev, site, wpb, dq_scale, p16);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// temporal attention: one wave per (n, pixel, head); rows of (n, t, pixel) are (n*T + t)*HW + pixel.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void tattn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ o, int Tq, int Tk,
                                                       int HW, int C, int nh, int causal, float p, const uint64_t* seed_dev,
                                                       uint32_t site, int p16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hd = C / nh, hp = hd + 1, Tp = Tk + 1;
  float* sq = smem;             // [Tq][hp]
  float* sk = sq + Tq * hp;     // [Tk][hp]
  float* sv = sk + Tk * hp;     // [Tk][hp]
  float* ss = sv + Tk * hp;     // [Tq][Tp]
  const int np = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const int n = np / HW, pix = np - n * HW;
  for (int i = lane; i < Tq * hd; i += 64) {
    const int t = i / hd, d = i - t * hd;
    sq[t * hp + d] = q[((int64_t)(n * Tq + t) * HW + pix) * C + h * hd + d];
  }
  for (int i = lane; i < Tk * hd; i += 64) {
    const int t = i / hd, d = i - t * hd;
    const int64_t g = ((int64_t)(n * Tk + t) * HW + pix) * C + h * hd + d;
    sk[t * hp + d] = k[g];
    sv[t * hp + d] = v[g];
  }
  __syncthreads();
  for (int e = lane; e < Tq * Tk; e += 64) {
    const int i = e / Tk, j = e - i * Tk;
    float a = 0.f;
    for (int d = 0; d < hd; ++d) a += sq[i * hp + d] * sk[j * hp + d];
    if (causal && j > i) a = -INFINITY;
    ss[i * Tp + j] = a;
  }
  __syncthreads();
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  for (int i = lane; i < Tq; i += 64) {
    float m = -INFINITY;
    for (int j = 0; j < Tk; ++j) m = fmaxf(m, ss[i * Tp + j]);
    float s = 0.f;
    for (int j = 0; j < Tk; ++j) { const float e = __expf(ss[i * Tp + j] - m); ss[i * Tp + j] = e; s += e; }
    const float inv = 1.f / s;
    for (int j = 0; j < Tk; ++j) {
      float pr = ss[i * Tp + j] * inv;
      if (p > 0.f) pr *= vptr_drop_scale(seed, site, (((uint64_t)np * nh + h) * Tq + i) * Tk + j, p);
      ss[i * Tp + j] = pr;
    }
  }
  __syncthreads();
  for (int e = lane; e < Tq * hd; e += 64) {
    const int i = e / hd, d = e - i * hd;
    float a = 0.f;
    for (int j = 0; j < Tk; ++j) a += ss[i * Tp + j] * sv[j * hp + d];
    store1(o, ((int64_t)(n * Tq + i) * HW + pix) * C + h * hd + d, a, p16);
  }
}

// ------------------------------------------------------------------------------------------------------------
// fast path for T <= 16 time steps (10 in every shipped configuration): one wave per (n, pixel, head); score element
// (i, j) sits on lane (i % 4) * 16 + j of round i / 4 (see the window fast path above for the LDS layout).
// ------------------------------------------------------------------------------------------------------------
// Temporal fast path, register-staged rows: T x (hp / 2) float2 items over one wave, at most TR_NIT per lane (T = 10,
// head dim 66: 340 items = 6 per lane).  Used to fetch the NEXT pixel's rows while the current pixel is computed.
#define TR_NIT 6
#define TR_PPW 4   // consecutive pixels per wave in the prefetching variant
struct TRows {
  float2 v[TR_NIT];
};
__device__ __forceinline__ void trows_load(TRows& r, const float* __restrict__ src, int n, int T, int HW, int pix, int C, int hoff, int hd,
                                           int hp, int lane) {
  const int h2 = hp >> 1, v2 = hd >> 1;
#pragma unroll
  for (int it = 0; it < TR_NIT; ++it) {
    const int e = lane + it * 64;
    const int l = e / h2, d2 = e - l * h2;
    r.v[it] = make_float2(0.f, 0.f);
    if (l < T && d2 < v2) r.v[it] = *reinterpret_cast<const float2*>(src + ((int64_t)(n * T + l) * HW + pix) * C + hoff + 2 * d2);
  }
}
__device__ __forceinline__ void trows_store(const TRows& r, float* dst, int T, int hp, int lane) {
  const int h2 = hp >> 1;
#pragma unroll
  for (int it = 0; it < TR_NIT; ++it) {
    const int e = lane + it * 64;
    const int l = e / h2, d2 = e - l * h2;
    if (l < T) *reinterpret_cast<float2*>(dst + l * hp + 2 * d2) = r.v[it];
  }
}
template <bool PF>  // PF: TR_PPW pixels per wave, the next pixel's rows prefetched into registers
__global__ __launch_bounds__(64) void tattn16_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, float* __restrict__ o, int Tq, int Tk,
                                                         int HW, int C, int nh, int causal, float p, const uint64_t* seed_dev,
                                                         uint32_t site, int NP, int p16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hd = C / nh, hp = att_pitch(hd), n4 = hp >> 2;
  float* sq = smem;              // [Tq][hp]
  float* sk = sq + Tq * hp;      // [Tk][hp]
  float* sv = sk + Tk * hp;      // [Tk][hp]
  float* ss = sv + Tk * hp;      // [16][20]
  __shared__ int64_t rq[16], rk[16];
  // workgroup b runs on XCD b % 8: the nh heads of a token (which share its cache lines) get the same XCD
  constexpr int PPW = PF ? TR_PPW : 1;
  const int h = (blockIdx.x >> 3) % nh, npg = (blockIdx.x / (8 * nh)) * 8 + (blockIdx.x & 7), lane = threadIdx.x;
  const int np0 = npg * PPW, np1 = min(NP, np0 + PPW);   // tail: the grid is rounded up to a multiple of 8 pixel groups
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const int j = lane & 15;
  TRows tq, tk, tv;
  if (PF && np0 < np1) {
    const int n = np0 / HW, pix = np0 - n * HW;
    trows_load(tq, q, n, Tq, HW, pix, C, h * hd, hd, hp, lane);
    trows_load(tk, k, n, Tk, HW, pix, C, h * hd, hd, hp, lane);
    trows_load(tv, v, n, Tk, HW, pix, C, h * hd, hd, hp, lane);
  }
  for (int np = np0; np < np1; ++np) {
    const int n = np / HW, pix = np - n * HW;
    __syncthreads();
    if (lane < Tq) rq[lane] = ((int64_t)(n * Tq + lane) * HW + pix) * C;
    if (lane < Tk) rk[lane] = ((int64_t)(n * Tk + lane) * HW + pix) * C;
    if (PF) {
      trows_store(tq, sq, Tq, hp, lane);
      trows_store(tk, sk, Tk, hp, lane);
      trows_store(tv, sv, Tk, hp, lane);
      if (np + 1 < np1) {  // wave-uniform: the next pixel's rows fly while this one is computed
        const int n2 = (np + 1) / HW, pix2 = np + 1 - n2 * HW;
        trows_load(tq, q, n2, Tq, HW, pix2, C, h * hd, hd, hp, lane);
        trows_load(tk, k, n2, Tk, HW, pix2, C, h * hd, hd, hp, lane);
        trows_load(tv, v, n2, Tk, HW, pix2, C, h * hd, hd, hp, lane);
      }
    } else {
      __syncthreads();
      stage_rows(q, sq, rq, Tq, h * hd, hd, hp, lane, 64);
      stage_rows(k, sk, rk, Tk, h * hd, hd, hp, lane, 64);
      stage_rows(v, sv, rk, Tk, h * hd, hd, hp, lane, 64);
    }
    __syncthreads();
    for (int i0 = 0; i0 < Tq; i0 += 4) {
      const int i = i0 + (lane >> 4);
      const bool ok = i < Tq && j < Tk && !(causal && j > i);
      const int ic = min(i, Tq - 1), jc = min(j, Tk - 1);
      float a = 0.f;
      for (int d = 0; d < n4; ++d)
        a += dot4(reinterpret_cast<const float4*>(sq + ic * hp)[d], reinterpret_cast<const float4*>(sk + jc * hp)[d]);
      a = ok ? a : -INFINITY;
      const float m = row16_max(a);
      const float e = ok ? __expf(a - m) : 0.f;
      float pr = e / row16_sum(e);
      if (p > 0.f) pr *= vptr_drop_scale(seed, site, (((uint64_t)np * nh + h) * Tq + ic) * Tk + jc, p);
      if (i < Tq) ss[i * 20 + j] = ok ? pr : 0.f;
    }
    __syncthreads();
    for (int e = lane; e < Tq * n4; e += 64) {
      const int i = e / n4, d4 = e - i * n4;
      if (d4 * 4 >= hd) continue;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int c = 0; c < Tk; ++c) axpy4(acc, ss[i * 20 + c], reinterpret_cast<const float4*>(sv + c * hp)[d4]);
      store4(o, rq[i] + h * hd, d4 * 4, hd, acc, p16);
    }
  }
}

template <bool PF>
__global__ __launch_bounds__(64) void tattn16_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ dout,
                                                         float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                         int Tq, int Tk, int HW, int C, int nh, int causal, float p,
                                                         const uint64_t* seed_dev, uint32_t site, int NP, float dq_scale, int p16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hd = C / nh, hp = att_pitch(hd), n4 = hp >> 2;
  float* sq = smem;               // [Tq][hp]
  float* sdo = sq + Tq * hp;      // [Tq][hp]
  float* sk = sdo + Tq * hp;      // [Tk][hp]
  float* sv = sk + Tk * hp;       // [Tk][hp]
  float* sp = sv + Tk * hp;       // [16][20]  dropped probabilities
  float* sds = sp + 16 * 20;      // [16][20]  dS
  __shared__ int64_t rq[16], rk[16];
  constexpr int PPW = PF ? TR_PPW : 1;
  const int h = (blockIdx.x >> 3) % nh, npg = (blockIdx.x / (8 * nh)) * 8 + (blockIdx.x & 7), lane = threadIdx.x;
  const int np0 = npg * PPW, np1 = min(NP, np0 + PPW);
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const int j = lane & 15;
  TRows tq, tdo, tk, tv;
  if (PF && np0 < np1) {
    const int n = np0 / HW, pix = np0 - n * HW;
    trows_load(tq, q, n, Tq, HW, pix, C, h * hd, hd, hp, lane);
    trows_load(tdo, dout, n, Tq, HW, pix, C, h * hd, hd, hp, lane);
    trows_load(tk, k, n, Tk, HW, pix, C, h * hd, hd, hp, lane);
    trows_load(tv, v, n, Tk, HW, pix, C, h * hd, hd, hp, lane);
  }
  for (int np = np0; np < np1; ++np) {
  const int n = np / HW, pix = np - n * HW;
  __syncthreads();
  if (lane < Tq) rq[lane] = ((int64_t)(n * Tq + lane) * HW + pix) * C;
  if (lane < Tk) rk[lane] = ((int64_t)(n * Tk + lane) * HW + pix) * C;
  if (PF) {
    trows_store(tq, sq, Tq, hp, lane);
    trows_store(tdo, sdo, Tq, hp, lane);
    trows_store(tk, sk, Tk, hp, lane);
    trows_store(tv, sv, Tk, hp, lane);
    if (np + 1 < np1) {  // wave-uniform: the next pixel's rows fly while this one is computed
      const int n2 = (np + 1) / HW, pix2 = np + 1 - n2 * HW;
      trows_load(tq, q, n2, Tq, HW, pix2, C, h * hd, hd, hp, lane);
      trows_load(tdo, dout, n2, Tq, HW, pix2, C, h * hd, hd, hp, lane);
      trows_load(tk, k, n2, Tk, HW, pix2, C, h * hd, hd, hp, lane);
      trows_load(tv, v, n2, Tk, HW, pix2, C, h * hd, hd, hp, lane);
    }
  } else {
    __syncthreads();
    stage_rows(q, sq, rq, Tq, h * hd, hd, hp, lane, 64);
    stage_rows(dout, sdo, rq, Tq, h * hd, hd, hp, lane, 64);
    stage_rows(k, sk, rk, Tk, h * hd, hd, hp, lane, 64);
    stage_rows(v, sv, rk, Tk, h * hd, hd, hp, lane, 64);
  }
  __syncthreads();
  for (int i0 = 0; i0 < Tq; i0 += 4) {
    const int i = i0 + (lane >> 4);
    const bool ok = i < Tq && j < Tk && !(causal && j > i);
    const int ic = min(i, Tq - 1), jc = min(j, Tk - 1);
    float a = 0.f, b = 0.f;
    for (int d = 0; d < n4; ++d) {
      a += dot4(reinterpret_cast<const float4*>(sq + ic * hp)[d], reinterpret_cast<const float4*>(sk + jc * hp)[d]);
      b += dot4(reinterpret_cast<const float4*>(sdo + ic * hp)[d], reinterpret_cast<const float4*>(sv + jc * hp)[d]);
    }
    a = ok ? a : -INFINITY;
    const float m = row16_max(a);
    const float e = ok ? __expf(a - m) : 0.f;
    const float pr = e / row16_sum(e);
    const float sc = p > 0.f ? vptr_drop_scale(seed, site, (((uint64_t)np * nh + h) * Tq + ic) * Tk + jc, p) : 1.f;
    const float dpr = ok ? b * sc : 0.f;
    const float ds = pr * (dpr - row16_sum(dpr * pr));
    if (i < Tq) {
      sp[i * 20 + j] = ok ? pr * sc : 0.f;
      sds[i * 20 + j] = ok ? ds : 0.f;
    }
  }
  __syncthreads();
  for (int e = lane; e < Tq * n4; e += 64) {
    const int i = e / n4, d4 = e - i * n4;
    if (d4 * 4 >= hd) continue;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < Tk; ++c) axpy4(acc, sds[i * 20 + c], reinterpret_cast<const float4*>(sk + c * hp)[d4]);
    store4(dq, rq[i] + h * hd, d4 * 4, hd, make_float4(acc.x * dq_scale, acc.y * dq_scale, acc.z * dq_scale, acc.w * dq_scale), p16);
  }
  for (int e = lane; e < Tk * n4; e += 64) {
    const int c = e / n4, d4 = e - c * n4;
    if (d4 * 4 >= hd) continue;
    float4 ak = make_float4(0.f, 0.f, 0.f, 0.f), av = ak;
    for (int i = 0; i < Tq; ++i) {
      axpy4(ak, sds[i * 20 + c], reinterpret_cast<const float4*>(sq + i * hp)[d4]);
      axpy4(av, sp[i * 20 + c], reinterpret_cast<const float4*>(sdo + i * hp)[d4]);
    }
    store4(dk, rk[c] + h * hd, d4 * 4, hd, ak, p16);
    store4(dv, rk[c] + h * hd, d4 * 4, hd, av, p16);
  }
  }
}

extern "C" int vptr_tattn_fwd(const float* q, const float* k, const float* v, float* o, int Nb, int Tq, int Tk, int HW, int C,
                              int nh, int causal, float dropout_p, const uint64_t* seed_dev, uint32_t site,
                              int p16, vptr_stream_t stream) {
  if (p16) VPTR_CHECK(C % 16 == 0, "tattn_fwd: P16 outputs need C %% 16 == 0 (got %d)", C);
  VPTR_CHECK(Nb > 0 && Tq > 0 && Tk > 0 && HW > 0 && C > 0 && nh > 0, "tattn_fwd: bad arguments");
  VPTR_CHECK(C % nh == 0, "tattn_fwd: embed_dim must be divisible by num_heads");
  VPTR_CHECK(Tq <= ATT_MAXT && Tk <= ATT_MAXT, "tattn_fwd: T must be <= %d", ATT_MAXT);
  if (causal) VPTR_CHECK(Tq == Tk, "tattn_fwd: causal mask needs Tq == Tk");
  if (dropout_p > 0.f) VPTR_CHECK(seed_dev && dropout_p < 1.f, "tattn_fwd: dropout needs seed_dev");
  const int hd = C / nh;
  if (vptr_attn_mfma_ok(Tq, Tk, C, nh, causal, (int64_t)Nb * HW)) {
    AmGeom gm = {1, 0, 0, 0, Tq, Tk, HW, Tq, Tk, C, nh, hd, Nb * HW};
    const int rc = vptr_attn_mfma_fwd(q, k, v, nullptr, nullptr, o, gm, causal, dropout_p, seed_dev, site, p16, (hipStream_t)stream);
    if (rc) return rc;
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  if (vptr_attn16_ok(1, Tq, Tk, C, nh, 0, 0)) {   // T <= 16: LDS-free MFMA kernels (attn16.hip)
    A16Geom g16 = {1, 0, 0, Tq, Tk, HW, C, nh, hd, Tq, Tk, Nb * HW, causal};
    const int rc = vptr_attn16_fwd(q, k, v, nullptr, nullptr, o, g16, dropout_p, seed_dev, site, p16, (hipStream_t)stream);
    if (rc) return rc;
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  if (Tq <= 16 && Tk <= 16 && hd % 2 == 0 && C % 2 == 0) {
    const size_t lds16 = sizeof(float) * ((Tq + 2 * Tk) * att_pitch(hd) + 16 * 20);
    const int items = (Tq > Tk ? Tq : Tk) * (att_pitch(hd) / 2);
    if (items <= 64 * TR_NIT && Nb * HW >= 64 * TR_PPW)
      tattn16_fwd_kernel<true><<<dim3(cdiv(cdiv(Nb * HW, TR_PPW), 8) * 8 * nh), 64, lds16, (hipStream_t)stream>>>(
          q, k, v, o, Tq, Tk, HW, C, nh, causal, dropout_p, seed_dev, site, Nb * HW, p16);
    else
      tattn16_fwd_kernel<false><<<dim3(cdiv(Nb * HW, 8) * 8 * nh), 64, lds16, (hipStream_t)stream>>>(q, k, v, o, Tq, Tk, HW, C, nh, causal,
                                                                                                    dropout_p, seed_dev, site, Nb * HW, p16);
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  const size_t lds = sizeof(float) * ((Tq + 2 * Tk) * (hd + 1) + Tq * (Tk + 1));
  VPTR_CHECK(lds <= 64 * 1024, "tattn_fwd: LDS budget exceeded (%zu B)", lds);
  tattn_fwd_kernel<<<dim3(Nb * HW, nh), 64, lds, (hipStream_t)stream>>>(q, k, v, o, Tq, Tk, HW, C, nh, causal, dropout_p, seed_dev,
                                                                       site, p16);
  VPTR_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(64) void tattn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, const float* __restrict__ dout,
                                                       float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                       int Tq, int Tk, int HW, int C, int nh, int causal, float p,
                                                       const uint64_t* seed_dev, uint32_t site, float dq_scale, int p16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hd = C / nh, hp = hd + 1, Tp = Tk + 1;
  float* sq = smem;              // [Tq][hp]
  float* sdo = sq + Tq * hp;     // [Tq][hp]
  float* sk = sdo + Tq * hp;     // [Tk][hp]
  float* sv = sk + Tk * hp;      // [Tk][hp]
  float* sp = sv + Tk * hp;      // [Tq][Tp]
  float* sds = sp + Tq * Tp;     // [Tq][Tp]
  const int np = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const int n = np / HW, pix = np - n * HW;
  for (int i = lane; i < Tq * hd; i += 64) {
    const int t = i / hd, d = i - t * hd;
    const int64_t g = ((int64_t)(n * Tq + t) * HW + pix) * C + h * hd + d;
    sq[t * hp + d] = q[g];
    sdo[t * hp + d] = dout[g];
  }
  for (int i = lane; i < Tk * hd; i += 64) {
    const int t = i / hd, d = i - t * hd;
    const int64_t g = ((int64_t)(n * Tk + t) * HW + pix) * C + h * hd + d;
    sk[t * hp + d] = k[g];
    sv[t * hp + d] = v[g];
  }
  __syncthreads();
  for (int e = lane; e < Tq * Tk; e += 64) {
    const int i = e / Tk, j = e - i * Tk;
    float a = 0.f, b = 0.f;
    for (int d = 0; d < hd; ++d) {
      a += sq[i * hp + d] * sk[j * hp + d];
      b += sdo[i * hp + d] * sv[j * hp + d];
    }
    if (causal && j > i) a = -INFINITY;
    sp[i * Tp + j] = a;
    sds[i * Tp + j] = b;
  }
  __syncthreads();
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  for (int i = lane; i < Tq; i += 64) {
    float m = -INFINITY;
    for (int j = 0; j < Tk; ++j) m = fmaxf(m, sp[i * Tp + j]);
    float s = 0.f;
    for (int j = 0; j < Tk; ++j) { const float e = __expf(sp[i * Tp + j] - m); sp[i * Tp + j] = e; s += e; }
    const float inv = 1.f / s;
    float dot = 0.f;
    for (int j = 0; j < Tk; ++j) {
      const float pr = sp[i * Tp + j] * inv;
      float sc = 1.f;
      if (p > 0.f) sc = vptr_drop_scale(seed, site, (((uint64_t)np * nh + h) * Tq + i) * Tk + j, p);
      const float dpr = sds[i * Tp + j] * sc;
      dot += dpr * pr;
      sds[i * Tp + j] = dpr;
      sp[i * Tp + j] = pr;
    }
    for (int j = 0; j < Tk; ++j) {
      const float pr = sp[i * Tp + j];
      float sc = 1.f;
      if (p > 0.f) sc = vptr_drop_scale(seed, site, (((uint64_t)np * nh + h) * Tq + i) * Tk + j, p);
      sds[i * Tp + j] = pr * (sds[i * Tp + j] - dot);
      sp[i * Tp + j] = pr * sc;
    }
  }
  __syncthreads();
  for (int e = lane; e < Tq * hd; e += 64) {
    const int i = e / hd, d = e - i * hd;
    float a = 0.f;
    for (int j = 0; j < Tk; ++j) a += sds[i * Tp + j] * sk[j * hp + d];
    store1(dq, ((int64_t)(n * Tq + i) * HW + pix) * C + h * hd + d, a * dq_scale, p16);
  }
  for (int e = lane; e < Tk * hd; e += 64) {
    const int j = e / hd, d = e - j * hd;
    float ak = 0.f, av = 0.f;
    for (int i = 0; i < Tq; ++i) {
      ak += sds[i * Tp + j] * sq[i * hp + d];
      av += sp[i * Tp + j] * sdo[i * hp + d];
    }
    const int64_t g = ((int64_t)(n * Tk + j) * HW + pix) * C + h * hd + d;
    store1(dk, g, ak, p16);
    store1(dv, g, av, p16);
  }
}

extern "C" int vptr_tattn_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq, float* dk,
                              float* dv, int Nb, int Tq, int Tk, int HW, int C, int nh, int causal, float dropout_p,
                              const uint64_t* seed_dev, uint32_t site, float dq_scale, int p16, vptr_stream_t stream) {
  if (p16) VPTR_CHECK(C % 16 == 0, "tattn_bwd: P16 outputs need C %% 16 == 0 (got %d)", C);
  VPTR_CHECK(Nb > 0 && Tq > 0 && Tk > 0 && HW > 0 && C > 0 && nh > 0, "tattn_bwd: bad arguments");
  VPTR_CHECK(C % nh == 0 && Tq <= ATT_MAXT && Tk <= ATT_MAXT, "tattn_bwd: unsupported geometry");
  if (causal) VPTR_CHECK(Tq == Tk, "tattn_bwd: causal mask needs Tq == Tk");
  if (dropout_p > 0.f) VPTR_CHECK(seed_dev && dropout_p < 1.f, "tattn_bwd: dropout needs seed_dev");
  const int hd = C / nh;
  if (vptr_attn_mfma_ok(Tq, Tk, C, nh, causal, (int64_t)Nb * HW)) {
    AmGeom gm = {1, 0, 0, 0, Tq, Tk, HW, Tq, Tk, C, nh, hd, Nb * HW};
    const int rc = vptr_attn_mfma_bwd(q, k, v, nullptr, nullptr, dout, dq, dk, dv, nullptr, gm, causal, dropout_p, seed_dev, site, dq_scale, p16,
                                      (hipStream_t)stream);
    if (rc) return rc;
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  if (vptr_attn16_ok(1, Tq, Tk, C, nh, 0, 1)) {
    A16Geom g16 = {1, 0, 0, Tq, Tk, HW, C, nh, hd, Tq, Tk, Nb * HW, causal};
    const int rc = vptr_attn16_bwd(q, k, v, nullptr, nullptr, dout, dq, dk, dv, nullptr, g16, dropout_p, seed_dev, site, dq_scale, p16, nullptr, 0,
                                   (hipStream_t)stream);
    if (rc) return rc;
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  if (Tq <= 16 && Tk <= 16 && hd % 2 == 0 && C % 2 == 0) {
    const size_t lds16 = sizeof(float) * (2 * (Tq + Tk) * att_pitch(hd) + 2 * 16 * 20);
    const int items = (Tq > Tk ? Tq : Tk) * (att_pitch(hd) / 2);
    if (items <= 64 * TR_NIT && Nb * HW >= 64 * TR_PPW)
      tattn16_bwd_kernel<true><<<dim3(cdiv(cdiv(Nb * HW, TR_PPW), 8) * 8 * nh), 64, lds16, (hipStream_t)stream>>>(
          q, k, v, dout, dq, dk, dv, Tq, Tk, HW, C, nh, causal, dropout_p, seed_dev, site, Nb * HW, dq_scale, p16);
    else
      tattn16_bwd_kernel<false><<<dim3(cdiv(Nb * HW, 8) * 8 * nh), 64, lds16, (hipStream_t)stream>>>(
          q, k, v, dout, dq, dk, dv, Tq, Tk, HW, C, nh, causal, dropout_p, seed_dev, site, Nb * HW, dq_scale, p16);
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  const size_t lds = sizeof(float) * (2 * (Tq + Tk) * (hd + 1) + 2 * Tq * (Tk + 1));
  VPTR_CHECK(lds <= 160 * 1024, "tattn_bwd: LDS budget exceeded (%zu B)", lds);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)tattn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  tattn_bwd_kernel<<<dim3(Nb * HW, nh), 64, lds, (hipStream_t)stream>>>(q, k, v, dout, dq, dk, dv, Tq, Tk, HW, C, nh, causal,
                                                                       dropout_p, seed_dev, site, dq_scale, p16);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// temporal-spatial window attention (TemporalSpatialLocalMultiheadAttention, VidHRFormer_modules.py:219-284,444-484):
// for every window (n, qh, qw) and head, the Tq*ws*ws query tokens attend to the Tk*ws*ws memory tokens of the same
// window.  Sequence element s = (t, ph, pw) lives at token row ((n*T + t)*H + qh*ws + ph)*W + qw*ws + pw, so the
// reference's pad / permute / reverse-permute copies are index arithmetic here.  Exact fp32, K and V of the window
// staged in LDS, queries in chunks of TS_QC.
// ------------------------------------------------------------------------------------------------------------
#define TS_QC 16
__device__ __forceinline__ int64_t ts_row(int n, int T, int H, int W, int ws, int qh, int qw, int s) {
  const int w2 = ws * ws;
  const int t = s / w2, r = s - t * w2;
  const int ph = r / ws, pw = r - ph * ws;
  return ((int64_t)(n * T + t) * H + qh * ws + ph) * W + qw * ws + pw;
}

__global__ __launch_bounds__(256) void tsattn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, float* __restrict__ o, int Tq, int Tk,
                                                         int H, int W, int ws, int C, int nh, float p,
                                                         const uint64_t* seed_dev, uint32_t site, int p16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hd = C / nh, hp = hd + 1, w2 = ws * ws;
  const int Lq = Tq * w2, Lk = Tk * w2, Lp = Lk + 1;
  float* sk = smem;               // [Lk][hp]
  float* sv = sk + Lk * hp;       // [Lk][hp]
  float* sq = sv + Lk * hp;       // [TS_QC][hp]
  float* ss = sq + TS_QC * hp;    // [TS_QC][Lp]
  const int nwx = W / ws, nwy = H / ws;
  const int b = blockIdx.x, h = blockIdx.y, i0 = blockIdx.z * TS_QC;
  const int n = b / (nwy * nwx), qh = (b / nwx) % nwy, qw = b % nwx;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nq = min(TS_QC, Lq - i0);
  for (int e = tid; e < Lk * hd; e += 256) {
    const int j = e / hd, d = e - j * hd;
    const int64_t g = ts_row(n, Tk, H, W, ws, qh, qw, j) * C + h * hd + d;
    sk[j * hp + d] = k[g];
    sv[j * hp + d] = v[g];
  }
  for (int e = tid; e < nq * hd; e += 256) {
    const int i = e / hd, d = e - i * hd;
    sq[i * hp + d] = q[ts_row(n, Tq, H, W, ws, qh, qw, i0 + i) * C + h * hd + d];
  }
  __syncthreads();
  for (int e = tid; e < nq * Lk; e += 256) {
    const int i = e / Lk, j = e - i * Lk;
    float a = 0.f;
    for (int d = 0; d < hd; ++d) a += sq[i * hp + d] * sk[j * hp + d];
    ss[i * Lp + j] = a;
  }
  __syncthreads();
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  for (int i = wv; i < nq; i += 4) {  // one wave per query row
    float m = -INFINITY;
    for (int j = lane; j < Lk; j += 64) m = fmaxf(m, ss[i * Lp + j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < Lk; j += 64) { const float e = __expf(ss[i * Lp + j] - m); ss[i * Lp + j] = e; s += e; }
    const float inv = 1.f / wave_sum(s);
    for (int j = lane; j < Lk; j += 64) {
      float pr = ss[i * Lp + j] * inv;
      if (p > 0.f) pr *= vptr_drop_scale(seed, site, (((uint64_t)b * nh + h) * Lq + (i0 + i)) * Lk + j, p);
      ss[i * Lp + j] = pr;
    }
  }
  __syncthreads();
  for (int e = tid; e < nq * hd; e += 256) {
    const int i = e / hd, d = e - i * hd;
    float a = 0.f;
    for (int j = 0; j < Lk; ++j) a += ss[i * Lp + j] * sv[j * hp + d];
    store1(o, ts_row(n, Tq, H, W, ws, qh, qw, i0 + i) * C + h * hd + d, a, p16);
  }
}

// backward: one workgroup of 512 threads per (window, head) walks the query chunks; dK / dV are accumulated in registers
// (thread (jg, d) owns channel d of keys jg, jg + 512/hd, ...: at most TS_NACC each) and written once, dQ chunk by chunk.
#define TS_NACC 24
__global__ __launch_bounds__(512) void tsattn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ dout,
                                                         float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                         int Tq, int Tk, int H, int W, int ws, int C, int nh, float p,
                                                         const uint64_t* seed_dev, uint32_t site, int p16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hd = C / nh, hp = hd + 1, w2 = ws * ws;
  const int Lq = Tq * w2, Lk = Tk * w2, Lp = Lk + 1;
  float* sk = smem;                 // [Lk][hp]
  float* sv = sk + Lk * hp;         // [Lk][hp]
  float* sq = sv + Lk * hp;         // [TS_QC][hp]
  float* sdo = sq + TS_QC * hp;     // [TS_QC][hp]
  float* sp = sdo + TS_QC * hp;     // [TS_QC][Lp]   P * dropout scale
  float* sds = sp + TS_QC * Lp;     // [TS_QC][Lp]   dS
  const int nwx = W / ws, nwy = H / ws;
  const int b = blockIdx.x, h = blockIdx.y;
  const int n = b / (nwy * nwx), qh = (b / nwx) % nwy, qw = b % nwx;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int e = tid; e < Lk * hd; e += 512) {
    const int j = e / hd, d = e - j * hd;
    const int64_t g = ts_row(n, Tk, H, W, ws, qh, qw, j) * C + h * hd + d;
    sk[j * hp + d] = k[g];
    sv[j * hp + d] = v[g];
  }
  float ak[TS_NACC], av[TS_NACC];
#pragma unroll
  for (int u = 0; u < TS_NACC; ++u) { ak[u] = 0.f; av[u] = 0.f; }
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  // dK / dV ownership: thread = (key group jg, channel od); it owns keys jg, jg + jgroups, ... (nu <= TS_NACC of them)
  const int jgroups = 512 / hd, jg = tid / hd, od = tid - jg * hd;
  const bool owner = jg < jgroups;
  const int nu = (Lk + jgroups - 1) / jgroups;
  for (int i0 = 0; i0 < Lq; i0 += TS_QC) {
    const int nq = min(TS_QC, Lq - i0);
    __syncthreads();  // previous chunk's tiles are no longer read (also orders the K/V stores before the first use)
    for (int e = tid; e < TS_QC * hd; e += 512) {
      const int i = e / hd, d = e - i * hd;
      float qv = 0.f, dv_ = 0.f;
      if (i < nq) {
        const int64_t g = ts_row(n, Tq, H, W, ws, qh, qw, i0 + i) * C + h * hd + d;
        qv = q[g];
        dv_ = dout[g];
      }
      sq[i * hp + d] = qv;      // rows beyond nq are zero: they add nothing to dK / dV
      sdo[i * hp + d] = dv_;
    }
    __syncthreads();
    for (int e = tid; e < TS_QC * Lk; e += 512) {
      const int i = e / Lk, j = e - i * Lk;
      float a = 0.f, c = 0.f;
      for (int d = 0; d < hd; ++d) {
        a += sq[i * hp + d] * sk[j * hp + d];
        c += sdo[i * hp + d] * sv[j * hp + d];
      }
      sp[i * Lp + j] = a;
      sds[i * Lp + j] = c;
    }
    __syncthreads();
    for (int i = wv; i < TS_QC; i += 8) {  // one wave per query row: softmax, dropout, dS = P * (dP - sum(dP * P))
      float m = -INFINITY;
      for (int j = lane; j < Lk; j += 64) m = fmaxf(m, sp[i * Lp + j]);
      m = wave_max(m);
      float s = 0.f;
      for (int j = lane; j < Lk; j += 64) { const float e = __expf(sp[i * Lp + j] - m); sp[i * Lp + j] = e; s += e; }
      const float inv = 1.f / wave_sum(s);
      float dot = 0.f;
      for (int j = lane; j < Lk; j += 64) {
        const float pr = sp[i * Lp + j] * inv;
        float sc = 1.f;
        if (p > 0.f) sc = vptr_drop_scale(seed, site, (((uint64_t)b * nh + h) * Lq + (i0 + i)) * Lk + j, p);
        const float dpr = sds[i * Lp + j] * sc;
        dot += dpr * pr;
        sds[i * Lp + j] = dpr;
        sp[i * Lp + j] = pr;
      }
      dot = wave_sum(dot);
      for (int j = lane; j < Lk; j += 64) {
        const float pr = sp[i * Lp + j];
        float sc = 1.f;
        if (p > 0.f) sc = vptr_drop_scale(seed, site, (((uint64_t)b * nh + h) * Lq + (i0 + i)) * Lk + j, p);
        sds[i * Lp + j] = (i < nq) ? pr * (sds[i * Lp + j] - dot) : 0.f;
        sp[i * Lp + j] = (i < nq) ? pr * sc : 0.f;
      }
    }
    __syncthreads();
    for (int e = tid; e < nq * hd; e += 512) {
      const int i = e / hd, d = e - i * hd;
      float a = 0.f;
      for (int j = 0; j < Lk; ++j) a += sds[i * Lp + j] * sk[j * hp + d];
      store1(dq, ts_row(n, Tq, H, W, ws, qh, qw, i0 + i) * C + h * hd + d, a, p16);
    }
    if (owner) {
#pragma unroll 1
      for (int i = 0; i < TS_QC; ++i) {
        const float qv = sq[i * hp + od], dov = sdo[i * hp + od];
#pragma unroll
        for (int u = 0; u < TS_NACC; ++u) {
          if (u < nu) {  // workgroup-uniform
            const int j = min(jg + u * jgroups, Lk - 1);  // the clamped duplicates of the last group are never written
            ak[u] += sds[i * Lp + j] * qv;
            av[u] += sp[i * Lp + j] * dov;
          }
        }
      }
    }
  }
  if (owner) {
#pragma unroll
    for (int u = 0; u < TS_NACC; ++u) {
      const int j = jg + u * jgroups;
      if (u < nu && j < Lk) {
        const int64_t g = ts_row(n, Tk, H, W, ws, qh, qw, j) * C + h * hd + od;
        store1(dk, g, ak[u], p16);
        store1(dv, g, av[u], p16);
      }
    }
  }
}

static int tsattn_check(const char* who, int Nb, int Tq, int Tk, int H, int W, int ws, int C, int nh, float p,
                        const uint64_t* seed_dev) {
  VPTR_CHECK(Nb > 0 && Tq > 0 && Tk > 0 && H > 0 && W > 0 && ws > 0 && C > 0 && nh > 0, "%s: bad arguments", who);
  VPTR_CHECK(C % nh == 0 && H % ws == 0 && W % ws == 0, "%s: C %% heads and H, W %% window must be 0 (pad first)", who);
  if (p > 0.f) VPTR_CHECK(seed_dev && p < 1.f, "%s: dropout needs seed_dev", who);
  return 0;
}

extern "C" int vptr_tsattn_fwd(const float* q, const float* k, const float* v, float* o, int Nb, int Tq, int Tk, int H, int W,
                               int ws, int C, int nh, float dropout_p, const uint64_t* seed_dev, uint32_t site,
                               int p16, vptr_stream_t stream) {
  if (p16) VPTR_CHECK(C % 16 == 0, "tsattn_fwd: P16 outputs need C %% 16 == 0 (got %d)", C);
  if (tsattn_check("tsattn_fwd", Nb, Tq, Tk, H, W, ws, C, nh, dropout_p, seed_dev)) return -1;
  const int hd = C / nh, Lq = Tq * ws * ws, Lk = Tk * ws * ws;
  const size_t lds = sizeof(float) * ((size_t)(2 * Lk + TS_QC) * (hd + 1) + (size_t)TS_QC * (Lk + 1));
  VPTR_CHECK(lds <= 160 * 1024, "tsattn_fwd: window sequence too long for LDS (Tk*ws*ws = %d, %zu B)", Lk, lds);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)tsattn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int windows = Nb * (H / ws) * (W / ws);
  tsattn_fwd_kernel<<<dim3(windows, nh, cdiv(Lq, TS_QC)), 256, lds, (hipStream_t)stream>>>(q, k, v, o, Tq, Tk, H, W, ws, C, nh,
                                                                                          dropout_p, seed_dev, site, p16);
  VPTR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vptr_tsattn_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq, float* dk,
                               float* dv, int Nb, int Tq, int Tk, int H, int W, int ws, int C, int nh, float dropout_p,
                               const uint64_t* seed_dev, uint32_t site, int p16, vptr_stream_t stream) {
  if (p16) VPTR_CHECK(C % 16 == 0, "tsattn_bwd: P16 outputs need C %% 16 == 0 (got %d)", C);
  if (tsattn_check("tsattn_bwd", Nb, Tq, Tk, H, W, ws, C, nh, dropout_p, seed_dev)) return -1;
  const int hd = C / nh, Lk = Tk * ws * ws;
  VPTR_CHECK(hd <= 512 && Lk <= TS_NACC * (512 / hd), "tsattn_bwd: Tk*ws*ws = %d exceeds %d keys per window", Lk, TS_NACC * (512 / hd));
  const size_t lds = sizeof(float) * ((size_t)(2 * Lk + 2 * TS_QC) * (hd + 1) + (size_t)2 * TS_QC * (Lk + 1));
  VPTR_CHECK(lds <= 160 * 1024, "tsattn_bwd: window sequence too long for LDS (Tk*ws*ws = %d, %zu B)", Lk, lds);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)tsattn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int windows = Nb * (H / ws) * (W / ws);
  tsattn_bwd_kernel<<<dim3(windows, nh), 512, lds, (hipStream_t)stream>>>(q, k, v, dout, dq, dk, dv, Tq, Tk, H, W, ws, C, nh,
                                                                         dropout_p, seed_dev, site, p16);
  VPTR_LAUNCH_CHECK();
  return 0;
}
