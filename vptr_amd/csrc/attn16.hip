// Attention cores for problems of at most 16 x 16 tokens on the matrix units, WITHOUT LDS (gfx950): every K64 attention of the
// VidHRFormer blocks -- 4 x 4 local windows (MultiHeadAttentionRPE.py:623-682) and per-pixel temporal attention over T <= 16 steps
// (nn.MultiheadAttention slow path, VidHRFormer_modules.py:74-84,183-206), head dim <= 96 (66 in every shipped configuration).
//
// One wave per (problem, head); a workgroup is four consecutive heads of one problem (its waves read 1 KB runs of the token rows).
// The trick that removes every transpose: the C/D layout of v_mfma_f32_16x16x{16,32} (lane (c, g): column c, rows 4g .. 4g+3) IS the
// B-operand layout of v_mfma_f32_16x16x16 (lane (c, g): column c, k = 4g .. 4g+3).  So
//   forward   S^T = K Q^T (16x16x32, operands straight from global memory: 8 consecutive head channels of row `lane & 15`)
//             softmax over the keys of a query = 4 registers + 2 cross-lane steps; P^T stays in registers
//             O^T = V^T P^T (16x16x16: A = V read "down a column" by 4-byte loads, B = P^T as it sits) -> lane (i, g) owns 4 consecutive
//             channels of output row i: 8 / 16-byte stores
//   backward  both orientations of the score-shaped tiles come from the SAME operand registers with A and B swapped
//             (S^T = K Q^T and S = Q K^T; dP^T = V dO^T and dP = dO V^T), so dQ^T = K^T dS^T, dK^T = Q^T dS and dV^T = dO^T P all find
//             their B operand already in place.
// Every product is split-bf16 (x = hi + lo; lo*hi + hi*lo + hi*hi, fp32 accumulate): fp32-class accuracy like the GEMMs.  Dropout
// masks hash the same element indices as the fp32 vector kernels of attn.hip (mask-exact parity test), the relative-position bias is
// 4 registers per lane, its gradient is summed in registers over a wave's problems and leaves through 49 LDS words per workgroup.
#include "attn_mfma.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;


// Addressing: row (problem, l) = base(problem) + lane-constant offset(l); the offsets are computed once per wave, the bases are
// workgroup-uniform (scalar registers), so a load is saddr + 32-bit lane offset.
__device__ __forceinline__ int a16_loff(const A16Geom& g, const int l) {   // float offset of sequence element l inside its problem
  return g.kind == 0 ? ((l >> 2) * g.W + (l & 3)) * g.C : l * g.HW * g.C;
}
__device__ __forceinline__ int64_t a16_base(const A16Geom& g, const int prob, const bool key) {   // float offset of element 0
  if (g.kind == 0) {
    const int nqw = g.W >> 2, nqh = g.H >> 2;
    const int b = prob / (nqh * nqw), rem = prob - b * (nqh * nqw);
    const int qh = rem / nqw, qw = rem - qh * nqw;
    return (((int64_t)b * g.H + qh * 4) * g.W + qw * 4) * g.C;
  }
  const int n = prob / g.HW, pix = prob - n * g.HW;
  return ((int64_t)n * (key ? g.Tk : g.Tq) * g.HW + pix) * g.C;
}
__device__ __forceinline__ void a16_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) vptr_split2(v[2 * p], v[2 * p + 1], h[p], l[p]);
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  const u32x4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
  hi = __builtin_bit_cast(bf16x8, hv);
  lo = __builtin_bit_cast(bf16x8, lv);
}
__device__ __forceinline__ void a16_split4(const float a, const float b, const float c, const float d, s16x4& hi, s16x4& lo) {
  uint32_t h0, l0, h1, l1;
  vptr_split2(a, b, h0, l0);
  vptr_split2(c, d, h1, l1);
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
  const u32x2 hv = {h0, h1}, lv = {l0, l1};
  hi = __builtin_bit_cast(s16x4, hv);
  lo = __builtin_bit_cast(s16x4, lv);
}
// split-bf16 products
__device__ __forceinline__ f32x4 a16_mma32(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 a16_mma16(const s16x4 ah, const s16x4 al, const s16x4 bh, const s16x4 bl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, c, 0, 0, 0);
}
// reductions over the 4 lane groups that share `lane & 15` (keys of one query in the transposed layout) ...
__device__ __forceinline__ float a16_gmax(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float a16_gsum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }
// ... and over the 16 lanes of one lane group (keys of one query in the plain layout)
__device__ __forceinline__ float a16_rmax(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float a16_rsum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Lane-constant part of a wave's work: offsets, validity masks, output byte offsets.  FULL = 16 x 16 problems without a causal mask
// (every K64 launch): no row masks anywhere.  Channel masks only exist for the LAST 32- / 16-channel block (the others are full).
template <int NB32, int NB16, bool FULL>
struct A16Lane {
  int lr, lq, h, hoff;
  bool hlive;
  int qoff8, koff8;        // row lr as a query / key row (8-channel operand loads), + 8 lq: float offsets incl. the head's first channel
  int qoff4[4], koff4[4];  // rows 4 lq + e as query / key rows (strided operand loads), + lr
  bool qok, kok, qrok[4], krok[4];
  bool c32[4];             // last 32-block: pair p of this lane's 8 channels exists
  bool c16, s16a, s16b;    // last 16-block: channel lr exists; output pairs (4 lq, +1), (4 lq + 2, +3) exist
  int oofs;                // fp32 output: float offset of channel 4 lq inside the head
  int p16a, p16b;          // P16 output: byte offsets (inside a row) of the hi halves of the pairs at channels 4 lq and 4 lq + 2 of block 0
  __device__ __forceinline__ void init(const A16Geom& g, const int hgroup) {
    const int lane = threadIdx.x & 63;
    lr = lane & 15; lq = lane >> 4;
    const int hraw = hgroup * 4 + (threadIdx.x >> 6);
    hlive = hraw < g.nh;
    h = hlive ? hraw : g.nh - 1;
    hoff = h * g.hd;
    qok = FULL || lr < g.Lq; kok = FULL || lr < g.Lk;
    qoff8 = a16_loff(g, qok ? lr : 0) + hoff + 8 * lq;
    koff8 = a16_loff(g, kok ? lr : 0) + hoff + 8 * lq;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qrok[e] = FULL || 4 * lq + e < g.Lq;
      krok[e] = FULL || 4 * lq + e < g.Lk;
      qoff4[e] = a16_loff(g, qrok[e] ? 4 * lq + e : 0) + hoff + lr;
      koff4[e] = a16_loff(g, krok[e] ? 4 * lq + e : 0) + hoff + lr;
    }
    const int r32 = g.hd - 32 * (NB32 - 1), r16 = g.hd - 16 * (NB16 - 1);
#pragma unroll
    for (int p = 0; p < 4; ++p) c32[p] = 8 * lq + 2 * p < r32;
    c16 = lr < r16;
    s16a = 4 * lq < r16; s16b = 4 * lq + 2 < r16;
    oofs = hoff + 4 * lq;
    const int ea = hoff + 4 * lq, eb = ea + 2;
    p16a = (ea >> 4) * 64 + (ea & 15) * 2;
    p16b = (eb >> 4) * 64 + (eb & 15) * 2;
  }
};
// 8 channels of block b of an operand row: unconditional loads (clamped rows), zeroed where the row / channel does not exist
template <int NB32, int NB16, bool FULL>
__device__ __forceinline__ void a16_ld8(const float* __restrict__ base, const int off, const int b, const bool rowok,
                                        const A16Lane<NB32, NB16, FULL>& L, bf16x8& hi, bf16x8& lo) {
  float v[8];
  const bool last = b == NB32 - 1;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float2 t = make_float2(0.f, 0.f);
    if (!last || L.c32[p]) t = *reinterpret_cast<const float2*>(base + off + b * 32 + 2 * p);
    if (!FULL && !rowok) t = make_float2(0.f, 0.f);
    v[2 * p] = t.x; v[2 * p + 1] = t.y;
  }
  a16_split8(v, hi, lo);
}
// rows 4 lq + e, channel 16 b + lr of an operand (the A fragment of a 16x16x16 product "down the columns")
template <int NB32, int NB16, bool FULL>
__device__ __forceinline__ void a16_ld4(const float* __restrict__ base, const int (&off)[4], const bool (&rok)[4], const int b,
                                        const A16Lane<NB32, NB16, FULL>& L, s16x4& hi, s16x4& lo) {
  float a[4];
  const bool cok = b < NB16 - 1 || L.c16;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = cok ? base[off[e] + b * 16] : 0.f;
    if (!FULL && !rok[e]) a[e] = 0.f;
  }
  a16_split4(a[0], a[1], a[2], a[3], hi, lo);
}
// 4 consecutive channels (16 b + 4 lq ..) of output row `rowf` (float offset of the row inside the tensor, lane-varying)
template <int NB32, int NB16, bool FULL>
__device__ __forceinline__ void a16_st4(float* __restrict__ dst, const int64_t rowf, const int b, const f32x4 v, const float scale,
                                        const A16Lane<NB32, NB16, FULL>& L, const int p16) {
  const bool last = b == NB16 - 1;
  const bool sa = !last || L.s16a, sb = !last || L.s16b;
  if (p16) {
    unsigned char* rb = reinterpret_cast<unsigned char*>(dst) + rowf * 4 + b * 64;
    uint32_t h, l;
    if (sa) { vptr_split2(v[0] * scale, v[1] * scale, h, l); *reinterpret_cast<uint32_t*>(rb + L.p16a) = h; *reinterpret_cast<uint32_t*>(rb + L.p16a + 32) = l; }
    if (sb) { vptr_split2(v[2] * scale, v[3] * scale, h, l); *reinterpret_cast<uint32_t*>(rb + L.p16b) = h; *reinterpret_cast<uint32_t*>(rb + L.p16b + 32) = l; }
  } else {
    float* o = dst + rowf + L.oofs + b * 16;
    if (sa) *reinterpret_cast<float2*>(o) = make_float2(v[0] * scale, v[1] * scale);
    if (sb) *reinterpret_cast<float2*>(o + 2) = make_float2(v[2] * scale, v[3] * scale);
  }
}
// dropout index of score element (query i, key j) of (problem, head): the flat index of the tensor the reference drops
__device__ __forceinline__ uint64_t a16_pidx(const A16Geom& g, const int prob, const int h, const int i, const int j) {
  return (((uint64_t)prob * g.nh + h) * g.Lq + i) * g.Lk + j;
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------------
template <int NB32, int NB16, bool FULL>   // 32- / 16-channel blocks that cover the head dim
__global__ __launch_bounds__(256, 4) void attn16_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                         const float* __restrict__ table, const int64_t* __restrict__ rel_index, float* __restrict__ o,
                                                         const A16Geom g, const float p, const uint64_t* __restrict__ seed_dev, const uint32_t site,
                                                         const int p16) {
  // work item = (problem, group of 4 heads); blockIdx.y = head group, so a workgroup keeps its heads (and bias registers) for all its problems
  A16Lane<NB32, NB16, FULL> L;
  L.init(g, blockIdx.y);
  const int lr = L.lr, lq = L.lq, h = L.h;
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  float bias[4] = {0.f, 0.f, 0.f, 0.f};   // S^T element (key j = 4 lq + r, query i = lr)
  if (table) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = table[rel_index[lr * 16 + 4 * lq + r] * g.nh + h];
  }
  const int64_t orow = a16_loff(g, L.qok ? lr : 0);   // output row offset of query lr (floats, relative to the problem's base)
  for (int prob = blockIdx.x; prob < g.nprob; prob += gridDim.x) {
    const int64_t qb0 = a16_base(g, prob, false), kb0 = a16_base(g, prob, true);   // workgroup-uniform
    const float* qb = q + qb0;
    const float* kb = k + kb0;
    const float* vb = v + kb0;
    // ---- S^T[j][i] = sum_d K[j][d] Q[i][d]
    f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NB32; ++b) {
      bf16x8 kh, kl, qh, ql;
      a16_ld8(kb, L.koff8, b, L.kok, L, kh, kl);
      a16_ld8(qb, L.qoff8, b, L.qok, L, qh, ql);
      st = a16_mma32(kh, kl, qh, ql, st);
    }
    // ---- softmax over the keys of query lr: 4 registers x 4 lane groups
    float e[4], m = -INFINITY;
    bool ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 4 * lq + r;
      ok[r] = FULL || (j < g.Lk && (!g.causal || j <= lr));
      e[r] = ok[r] ? st[r] + bias[r] : -INFINITY;
      m = fmaxf(m, e[r]);
    }
    m = a16_gmax(m);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { e[r] = ok[r] ? __expf(e[r] - m) : 0.f; sum += e[r]; }
    sum = a16_gsum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      e[r] *= inv;
      if (p > 0.f) e[r] *= vptr_drop_scale(seed, site, a16_pidx(g, prob, h, lr, 4 * lq + r), p);
    }
    s16x4 ph, pl;
    a16_split4(e[0], e[1], e[2], e[3], ph, pl);
    // ---- O^T[d][i] = sum_j V[j][d] P^T[j][i]; lane (i = lr, lq) ends up with channels 16 b + 4 lq .. + 3 of output row i
    const bool st_ok = L.qok && L.hlive;
#pragma unroll
    for (int b = 0; b < NB16; ++b) {
      s16x4 ah, al;
      a16_ld4(vb, L.koff4, L.krok, b, L, ah, al);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = a16_mma16(ah, al, ph, pl, acc);
      if (st_ok) a16_st4(o, qb0 + orow, b, acc, 1.f, L, p16);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------------
template <int NB32, int NB16, bool FULL>
__global__ __launch_bounds__(256, 3) void attn16_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                         const float* __restrict__ table, const int64_t* __restrict__ rel_index,
                                                         const float* __restrict__ dout, float* __restrict__ dq, float* __restrict__ dk,
                                                         float* __restrict__ dv, float* __restrict__ dtable, const A16Geom g, const float p,
                                                         const uint64_t* __restrict__ seed_dev, const uint32_t site, const float dq_scale, const int p16) {
  __shared__ float stab[4][52];   // per-head partial sums of the bias-table gradient (49 entries for 4 x 4 windows)
  A16Lane<NB32, NB16, FULL> L;
  L.init(g, blockIdx.y);
  const int lr = L.lr, lq = L.lq, h = L.h, wv = threadIdx.x >> 6;
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  float biasT[4] = {0.f, 0.f, 0.f, 0.f}, biasN[4] = {0.f, 0.f, 0.f, 0.f};   // transposed (j = 4 lq + r, i = lr) / plain (i = 4 lq + r, j = lr)
  int ridxT[4] = {0, 0, 0, 0};
  if (table || dtable) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ridxT[r] = (int)rel_index[lr * 16 + 4 * lq + r];
      if (table) {
        biasT[r] = table[ridxT[r] * g.nh + h];
        biasN[r] = table[rel_index[(4 * lq + r) * 16 + lr] * g.nh + h];
      }
    }
  }
  float dbias[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t qrow = a16_loff(g, L.qok ? lr : 0), krow = a16_loff(g, L.kok ? lr : 0);
  for (int prob = blockIdx.x; prob < g.nprob; prob += gridDim.x) {
    const int64_t qb0 = a16_base(g, prob, false), kb0 = a16_base(g, prob, true);   // workgroup-uniform
    const float* qb = q + qb0;
    const float* gb = dout + qb0;
    const float* kb = k + kb0;
    const float* vb = v + kb0;
    // ---- score-shaped tiles, both orientations from one set of operand registers
    f32x4 sT = {0.f, 0.f, 0.f, 0.f}, sN = sT, dpT = sT, dpN = sT;
#pragma unroll
    for (int b = 0; b < NB32; ++b) {
      bf16x8 kh, kl, qh, ql, vh, vl, gh, gl;
      a16_ld8(kb, L.koff8, b, L.kok, L, kh, kl);
      a16_ld8(qb, L.qoff8, b, L.qok, L, qh, ql);
      sT = a16_mma32(kh, kl, qh, ql, sT);     // S^T[j][i]
      sN = a16_mma32(qh, ql, kh, kl, sN);     // S[i][j]
      a16_ld8(vb, L.koff8, b, L.kok, L, vh, vl);
      a16_ld8(gb, L.qoff8, b, L.qok, L, gh, gl);
      dpT = a16_mma32(vh, vl, gh, gl, dpT);   // dP^T[j][i] = sum_d V[j][d] dO[i][d]
      dpN = a16_mma32(gh, gl, vh, vl, dpN);   // dP[i][j]
    }
    // ---- transposed layout: lane (i = lr, lq) holds keys j = 4 lq + r of query i
    s16x4 dsTh, dsTl, dsNh, dsNl, pNh, pNl;
    {
      bool ok[4];
      float m = -INFINITY, pT[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 4 * lq + r;
        ok[r] = FULL || (j < g.Lk && (!g.causal || j <= lr));
        pT[r] = ok[r] ? sT[r] + biasT[r] : -INFINITY;
        m = fmaxf(m, pT[r]);
      }
      m = a16_gmax(m);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { pT[r] = ok[r] ? __expf(pT[r] - m) : 0.f; sum += pT[r]; }
      sum = a16_gsum(sum);
      const float inv = 1.f / sum;
      float dot = 0.f, dpr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pT[r] *= inv;
        const float sc = p > 0.f ? vptr_drop_scale(seed, site, a16_pidx(g, prob, h, lr, 4 * lq + r), p) : 1.f;
        dpr[r] = dpT[r] * sc;
        dot += dpr[r] * pT[r];
      }
      dot = a16_gsum(dot);
      float dsT[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dsT[r] = (L.qok && ok[r]) ? pT[r] * (dpr[r] - dot) : 0.f;
        dbias[r] += dsT[r];
      }
      a16_split4(dsT[0], dsT[1], dsT[2], dsT[3], dsTh, dsTl);
    }
    // ---- plain layout: lane (j = lr, lq) holds queries i = 4 lq + r of key j
    {
      float pdN[4], dsN[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * lq + r;
        const bool ok = FULL || (L.kok && (!g.causal || lr <= i));
        const float s = ok ? sN[r] + biasN[r] : -INFINITY;
        const float m = a16_rmax(s);
        const float e = ok ? __expf(s - m) : 0.f;
        const float pr = e / a16_rsum(e);
        const float sc = p > 0.f ? vptr_drop_scale(seed, site, a16_pidx(g, prob, h, i, lr), p) : 1.f;
        const float dpr = dpN[r] * sc;
        const float dot = a16_rsum(dpr * pr);
        const bool iok = FULL || i < g.Lq;
        dsN[r] = (iok && ok) ? pr * (dpr - dot) : 0.f;
        pdN[r] = iok ? pr * sc : 0.f;
      }
      a16_split4(dsN[0], dsN[1], dsN[2], dsN[3], dsNh, dsNl);
      a16_split4(pdN[0], pdN[1], pdN[2], pdN[3], pNh, pNl);
    }
    // ---- dQ^T = K^T dS^T, dK^T = Q^T dS, dV^T = dO^T P: A operands read down the columns (rows 4 lq + e of the key / query side)
    const bool qst = L.qok && L.hlive, kst = L.kok && L.hlive;
#pragma unroll
    for (int b = 0; b < NB16; ++b) {
      s16x4 ah, al;
      f32x4 acc;
      a16_ld4(kb, L.koff4, L.krok, b, L, ah, al);            // dQ[i = lr][16 b + 4 lq + r] = sum_j K[j][d] dS^T[j][i]
      acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = a16_mma16(ah, al, dsTh, dsTl, acc);
      if (qst) a16_st4(dq, qb0 + qrow, b, acc, dq_scale, L, p16);
      a16_ld4(qb, L.qoff4, L.qrok, b, L, ah, al);            // dK[j = lr][...] = sum_i Q[i][d] dS[i][j]
      acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = a16_mma16(ah, al, dsNh, dsNl, acc);
      if (kst) a16_st4(dk, kb0 + krow, b, acc, 1.f, L, p16);
      a16_ld4(gb, L.qoff4, L.qrok, b, L, ah, al);            // dV[j = lr][...] = sum_i dO[i][d] P_dropped[i][j]
      acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = a16_mma16(ah, al, pNh, pNl, acc);
      if (kst) a16_st4(dv, kb0 + krow, b, acc, 1.f, L, p16);
    }
  }
  if (dtable) {   // 4 x 4 windows only (49 table entries): per-head LDS sums, then one global atomic per entry and head
    for (int e2 = threadIdx.x & 63; e2 < 52; e2 += 64) stab[wv][e2] = 0.f;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(&stab[wv][ridxT[r]], dbias[r]);
    __syncthreads();
    if ((threadIdx.x & 63) < 49 && L.hlive) unsafeAtomicAdd(dtable + (int64_t)(threadIdx.x & 63) * g.nh + h, stab[wv][threadIdx.x & 63]);
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// backward, second generation (the default for C % 4 == 0).  What bounded the kernel above was its memory instructions, not its
// arithmetic: per (problem, head) 48 8-byte operand loads, 60 4-byte loads "down the columns" for the second set of products and
// 60 4-byte stores (temporal, K64: 75 us; 47 without the stores, 34 without stores and column loads).  Here
//   * every head starts at a 16-byte aligned channel: an odd head of width 66 begins 2 channels early (cb = hoff & ~3) and masks the
//     two foreign channels out of the contractions, so operand loads are 16 bytes and a lane's 4 output channels are one aligned quad
//     (one 16-byte fp32 store, or 8 + 8 bytes of a P16 granule); the pairs at a head's edges are stored as pairs;
//   * the split operands of K, Q and dO are left in a wave-private LDS stash ([row][channel] bf16, hi and lo planes) and come back
//     as the "down the column" A fragments through ds_read_b64_tr_b16 -- no second trip to memory, no barriers (one wave owns it);
//   * the next problem's 24 operand loads are in flight while the current problem's products run (2 - 3 waves per SIMD, 256 registers);
//   * the bias-table gradient leaves through a workspace ([workgroup][head][52] partial sums + a small reduction kernel) when the
//     caller provides one, instead of 49 device-scope atomics per workgroup and head onto the same 392 words.
// ---------------------------------------------------------------------------------------------------------------------------
template <int NB32, int NB16, bool FULL>
struct A16LaneB {
  int lr, lq, h, cb, s0, span;
  bool hlive, qok, kok;
  int qoff, koff;      // float offset (inside a problem) of row lr's channel cb + 8 lq, as a query / key row
  unsigned ldm;        // bit 2 b + j: float4 j of 32-block b is loaded (inside the head's span and the row)
  unsigned kvm;        // bit 4 b + t: pair t of 32-block b belongs to this head (the K / V side of a contraction is zeroed elsewhere)
  unsigned vam, vbm;   // bit b: pairs (4 lq, +1) / (4 lq + 2, +3) of 16-block b belong to this head (stores)
  int oofs, p16q;      // fp32: float offset of channel cb + 4 lq inside a row; P16: byte offset of that quad's hi half inside a row
  int wroff, troff;    // stash: byte offset of this lane's 16-byte write (row lr, channel 8 lq) / 8-byte transposed read
  __device__ __forceinline__ void init(const A16Geom& g, const int hgroup, const int pitch) {
    const int lane = threadIdx.x & 63;
    lr = lane & 15; lq = lane >> 4;
    const int hraw = hgroup * 4 + (threadIdx.x >> 6);
    hlive = hraw < g.nh;
    h = hlive ? hraw : g.nh - 1;
    const int hoff = h * g.hd;
    cb = hoff & ~3; s0 = hoff - cb; span = s0 + g.hd;
    qok = FULL || lr < g.Lq; kok = FULL || lr < g.Lk;
    qoff = a16_loff(g, qok ? lr : 0) + cb + 8 * lq;
    koff = a16_loff(g, kok ? lr : 0) + cb + 8 * lq;
    const int lim = min((span + 3) & ~3, g.C - cb);
    ldm = kvm = vam = vbm = 0u;
#pragma unroll
    for (int b = 0; b < NB32; ++b) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (32 * b + 8 * lq + 4 * j + 4 <= lim) ldm |= 1u << (2 * b + j);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = 32 * b + 8 * lq + 2 * t;
        if (c >= s0 && c < span) kvm |= 1u << (4 * b + t);
      }
    }
#pragma unroll
    for (int b = 0; b < NB16; ++b) {
      const int c = 16 * b + 4 * lq;
      if (c >= s0 && c < span) vam |= 1u << b;
      if (c + 2 < span) vbm |= 1u << b;
    }
    oofs = cb + 4 * lq;
    p16q = (oofs >> 4) * 64 + (oofs & 15) * 2;
    wroff = lr * pitch + 16 * lq;
    troff = (4 * lq + (lr >> 2)) * pitch + 8 * (lr & 3);   // (the kernel redirects lane groups beyond the stash's rows)
  }
};
template <int NB32>
struct A16Raw { float4 v[NB32][2]; };
template <int NB32, int NB16, bool FULL>
__device__ __forceinline__ void a16b_load(const float* __restrict__ base, const int off, const A16LaneB<NB32, NB16, FULL>& L, A16Raw<NB32>& r) {
#pragma unroll
  for (int b = 0; b < NB32; ++b)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      r.v[b][j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((L.ldm >> (2 * b + j)) & 1u) r.v[b][j] = *reinterpret_cast<const float4*>(base + off + 32 * b + 4 * j);
    }
}
__device__ __forceinline__ void a16b_split(const float4 (&r)[2], const bool rowok, const unsigned pm, bf16x8& hi, bf16x8& lo) {
  float v[8] = {r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, r[1].z, r[1].w};
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (!rowok || !((pm >> t) & 1u)) v[2 * t] = v[2 * t + 1] = 0.f;
  a16_split8(v, hi, lo);
}
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ void a16b_stash(unsigned char* plane_hi, const int plane_bytes, const int wroff, const int b, const bf16x8 hi, const bf16x8 lo) {
  *reinterpret_cast<bf16x8*>(plane_hi + wroff + 64 * b) = hi;
  *reinterpret_cast<bf16x8*>(plane_hi + plane_bytes + wroff + 64 * b) = lo;
}
__device__ __forceinline__ void a16b_tr(const unsigned char* plane_hi, const int plane_bytes, const int troff, const int b, s16x4& hi, s16x4& lo) {
  hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plane_hi + troff + 32 * b));
  lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plane_hi + plane_bytes + troff + 32 * b));
}
template <int NB32, int NB16, bool FULL>
__device__ __forceinline__ void a16b_st4(float* __restrict__ dst, const int64_t rowf, const int b, const f32x4 v, const float scale,
                                         const A16LaneB<NB32, NB16, FULL>& L, const int p16) {
  const bool va = (L.vam >> b) & 1u, vb = (L.vbm >> b) & 1u;
  if (p16) {
    unsigned char* rb = reinterpret_cast<unsigned char*>(dst) + rowf * 4 + L.p16q + b * 64;
    uint32_t h0, l0, h1, l1;
    vptr_split2(v[0] * scale, v[1] * scale, h0, l0);
    vptr_split2(v[2] * scale, v[3] * scale, h1, l1);
    if (va && vb) {
      *reinterpret_cast<uint2*>(rb) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(rb + 32) = make_uint2(l0, l1);
    } else if (va) {
      *reinterpret_cast<uint32_t*>(rb) = h0; *reinterpret_cast<uint32_t*>(rb + 32) = l0;
    } else if (vb) {
      *reinterpret_cast<uint32_t*>(rb + 4) = h1; *reinterpret_cast<uint32_t*>(rb + 36) = l1;
    }
  } else {
    float* o = dst + rowf + L.oofs + b * 16;
    if (va && vb) *reinterpret_cast<float4*>(o) = make_float4(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale);
    else if (va) *reinterpret_cast<float2*>(o) = make_float2(v[0] * scale, v[1] * scale);
    else if (vb) *reinterpret_cast<float2*>(o + 2) = make_float2(v[2] * scale, v[3] * scale);
  }
}

template <int NB32, int NB16, bool FULL>
__global__ __launch_bounds__(256, 2) void attn16_bwd2_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                          const float* __restrict__ table, const int64_t* __restrict__ rel_index,
                                                          const float* __restrict__ dout, float* __restrict__ dq, float* __restrict__ dk,
                                                          float* __restrict__ dv, float* __restrict__ dtable, float* __restrict__ dtable_ws,
                                                          const A16Geom g, const float p, const uint64_t* __restrict__ seed_dev, const uint32_t site,
                                                          const float dq_scale, const int p16, const int rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char a16_stash[];
  __shared__ float stab[4][52];
  constexpr int PITCH = NB16 * 32;                 // bytes of a stash row: NB16 blocks of 16 bf16
  const int plane = rows * PITCH;                  // one plane (hi or lo) of one tensor
  A16LaneB<NB32, NB16, FULL> L;
  L.init(g, blockIdx.y, PITCH);
  const int lr = L.lr, lq = L.lq, h = L.h, wv = threadIdx.x >> 6;
  unsigned char* const sK = a16_stash + wv * (6 * plane);
  unsigned char* const sQ = sK + 2 * plane;
  unsigned char* const sG = sQ + 2 * plane;
  const bool wr_row = lr < rows;
  // rows = 12: lane group 3 would read rows 12 .. 15 (all-zero operands, not in the stash).  Every lane still executes the transposed
  // read (on rows 0 .. 3) and drops the result: with part of the wave masked off the instruction returned garbage now and then
  const bool rd_grp = 4 * lq < rows;
  const int troff = rd_grp ? L.troff : (lr >> 2) * PITCH + 8 * (lr & 3);
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  float biasT[4] = {0.f, 0.f, 0.f, 0.f}, biasN[4] = {0.f, 0.f, 0.f, 0.f};
  int ridxT[4] = {0, 0, 0, 0};
  if (table || dtable || dtable_ws) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ridxT[r] = (int)rel_index[lr * 16 + 4 * lq + r];
      if (table) {
        biasT[r] = table[ridxT[r] * g.nh + h];
        biasN[r] = table[rel_index[(4 * lq + r) * 16 + lr] * g.nh + h];
      }
    }
  }
  float dbias[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t qrow = a16_loff(g, L.qok ? lr : 0), krow = a16_loff(g, L.kok ? lr : 0);
  A16Raw<NB32> rk, rq, rv, rg;
  int prob = blockIdx.x;
  if (prob < g.nprob) {
    const int64_t qb0 = a16_base(g, prob, false), kb0 = a16_base(g, prob, true);
    a16b_load(k + kb0, L.koff, L, rk);
    a16b_load(q + qb0, L.qoff, L, rq);
    a16b_load(v + kb0, L.koff, L, rv);
    a16b_load(dout + qb0, L.qoff, L, rg);
  }
  for (; prob < g.nprob; prob += gridDim.x) {
    const int64_t qb0 = a16_base(g, prob, false), kb0 = a16_base(g, prob, true);   // workgroup-uniform
    // ---- score-shaped tiles, both orientations from one set of operand registers; K, Q, dO stay behind in the stash
    f32x4 sT = {0.f, 0.f, 0.f, 0.f}, sN = sT, dpT = sT, dpN = sT;
#pragma unroll
    for (int b = 0; b < NB32; ++b) {
      bf16x8 kh, kl, qh, ql, vh, vl, gh, gl;
      const bool wr = wr_row && 32 * b + 8 * lq < 16 * NB16;
      a16b_split(rk.v[b], L.kok, (L.kvm >> (4 * b)) & 15u, kh, kl);
      a16b_split(rq.v[b], L.qok, 15u, qh, ql);
      sT = a16_mma32(kh, kl, qh, ql, sT);     // S^T[j][i]
      sN = a16_mma32(qh, ql, kh, kl, sN);     // S[i][j]
      if (wr) { a16b_stash(sK, plane, L.wroff, b, kh, kl); a16b_stash(sQ, plane, L.wroff, b, qh, ql); }
      a16b_split(rv.v[b], L.kok, (L.kvm >> (4 * b)) & 15u, vh, vl);
      a16b_split(rg.v[b], L.qok, 15u, gh, gl);
      dpT = a16_mma32(vh, vl, gh, gl, dpT);   // dP^T[j][i]
      dpN = a16_mma32(gh, gl, vh, vl, dpN);   // dP[i][j]
      if (wr) a16b_stash(sG, plane, L.wroff, b, gh, gl);
    }
    {   // the next problem's operands: in flight during everything below
      const int nx = prob + gridDim.x;
      if (nx < g.nprob) {
        const int64_t qn = a16_base(g, nx, false), kn = a16_base(g, nx, true);
        a16b_load(k + kn, L.koff, L, rk);
        a16b_load(q + qn, L.qoff, L, rq);
        a16b_load(v + kn, L.koff, L, rv);
        a16b_load(dout + qn, L.qoff, L, rg);
      }
    }
    // ---- transposed layout: lane (i = lr, lq) holds keys j = 4 lq + r of query i
    s16x4 dsTh, dsTl, dsNh, dsNl, pNh, pNl;
    {
      bool ok[4];
      float m = -INFINITY, pT[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 4 * lq + r;
        ok[r] = FULL || (j < g.Lk && (!g.causal || j <= lr));
        pT[r] = ok[r] ? sT[r] + biasT[r] : -INFINITY;
        m = fmaxf(m, pT[r]);
      }
      m = a16_gmax(m);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { pT[r] = ok[r] ? __expf(pT[r] - m) : 0.f; sum += pT[r]; }
      sum = a16_gsum(sum);
      const float inv = 1.f / sum;
      float dot = 0.f, dpr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pT[r] *= inv;
        const float sc = p > 0.f ? vptr_drop_scale(seed, site, a16_pidx(g, prob, h, lr, 4 * lq + r), p) : 1.f;
        dpr[r] = dpT[r] * sc;
        dot += dpr[r] * pT[r];
      }
      dot = a16_gsum(dot);
      float dsT[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dsT[r] = (L.qok && ok[r]) ? pT[r] * (dpr[r] - dot) : 0.f;
        dbias[r] += dsT[r];
      }
      a16_split4(dsT[0], dsT[1], dsT[2], dsT[3], dsTh, dsTl);
    }
    // ---- plain layout: lane (j = lr, lq) holds queries i = 4 lq + r of key j
    {
      float pdN[4], dsN[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * lq + r;
        const bool ok = FULL || (L.kok && (!g.causal || lr <= i));
        const float s = ok ? sN[r] + biasN[r] : -INFINITY;
        const float m = a16_rmax(s);
        const float e = ok ? __expf(s - m) : 0.f;
        const float pr = e / a16_rsum(e);
        const float sc = p > 0.f ? vptr_drop_scale(seed, site, a16_pidx(g, prob, h, i, lr), p) : 1.f;
        const float dpr = dpN[r] * sc;
        const float dot = a16_rsum(dpr * pr);
        const bool iok = FULL || i < g.Lq;
        dsN[r] = (iok && ok) ? pr * (dpr - dot) : 0.f;
        pdN[r] = iok ? pr * sc : 0.f;
      }
      a16_split4(dsN[0], dsN[1], dsN[2], dsN[3], dsNh, dsNl);
      a16_split4(pdN[0], pdN[1], pdN[2], pdN[3], pNh, pNl);
    }
    asm volatile("" ::: "memory");   // the stash is written (by this wave, in order) before it is read back transposed
    // ---- dQ^T = K^T dS^T, dK^T = Q^T dS, dV^T = dO^T P: A fragments (channel lr of 16-block b, rows 4 lq ..) from the stash
    const bool qst = L.qok && L.hlive, kst = L.kok && L.hlive;
#pragma unroll
    for (int b = 0; b < NB16; ++b) {
      const s16x4 z4 = {0, 0, 0, 0};
      s16x4 ah, al;
      f32x4 acc;
      a16b_tr(sK, plane, troff, b, ah, al);
      if (!rd_grp) ah = al = z4;
      acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = a16_mma16(ah, al, dsTh, dsTl, acc);
      if (qst) a16b_st4(dq, qb0 + qrow, b, acc, dq_scale, L, p16);
      a16b_tr(sQ, plane, troff, b, ah, al);
      if (!rd_grp) ah = al = z4;
      acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = a16_mma16(ah, al, dsNh, dsNl, acc);
      if (kst) a16b_st4(dk, kb0 + krow, b, acc, 1.f, L, p16);
      a16b_tr(sG, plane, troff, b, ah, al);
      if (!rd_grp) ah = al = z4;
      acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = a16_mma16(ah, al, pNh, pNl, acc);
      if (kst) a16b_st4(dv, kb0 + krow, b, acc, 1.f, L, p16);
    }
    asm volatile("" ::: "memory");   // ... and read before the next problem overwrites it
  }
  if (dtable || dtable_ws) {   // 4 x 4 windows only (49 table entries): per-head LDS sums, then the workspace (or one atomic per entry)
    for (int e2 = threadIdx.x & 63; e2 < 52; e2 += 64) stab[wv][e2] = 0.f;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(&stab[wv][ridxT[r]], dbias[r]);
    __syncthreads();
    const int e2 = threadIdx.x & 63;
    if (e2 < 52 && dtable_ws) dtable_ws[((int64_t)blockIdx.x * (gridDim.y * 4) + blockIdx.y * 4 + wv) * 52 + e2] = L.hlive ? stab[wv][e2] : 0.f;
    else if (e2 < 49 && L.hlive && !dtable_ws) unsafeAtomicAdd(dtable + (int64_t)e2 * g.nh + h, stab[wv][e2]);
  }
}
// Forward with the same layout ideas (C % 4 == 0): 16-byte operand loads from the aligned head base, V left in a wave-private LDS
// stash and read back transposed for O^T = V^T P^T (no 4-byte "down the column" loads), outputs as aligned quads, the next
// problem's 18 loads in flight during the products.
template <int NB32, int NB16, bool FULL>
__global__ __launch_bounds__(256, 3) void attn16_fwd2_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                          const float* __restrict__ table, const int64_t* __restrict__ rel_index, float* __restrict__ o,
                                                          const A16Geom g, const float p, const uint64_t* __restrict__ seed_dev, const uint32_t site,
                                                          const int p16, const int rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char a16_stash[];
  constexpr int PITCH = NB16 * 32;
  const int plane = rows * PITCH;
  A16LaneB<NB32, NB16, FULL> L;
  L.init(g, blockIdx.y, PITCH);
  const int lr = L.lr, lq = L.lq, h = L.h, wv = threadIdx.x >> 6;
  unsigned char* const sV = a16_stash + wv * (2 * plane);
  const bool wr_row = lr < rows;
  const bool rd_grp = 4 * lq < rows;
  const int troff = rd_grp ? L.troff : (lr >> 2) * PITCH + 8 * (lr & 3);   // every lane executes the transposed reads (see the backward)
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  float bias[4] = {0.f, 0.f, 0.f, 0.f};   // S^T element (key j = 4 lq + r, query i = lr)
  if (table) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = table[rel_index[lr * 16 + 4 * lq + r] * g.nh + h];
  }
  const int64_t orow = a16_loff(g, L.qok ? lr : 0);
  A16Raw<NB32> rk, rq, rv;
  int prob = blockIdx.x;
  if (prob < g.nprob) {
    const int64_t qb0 = a16_base(g, prob, false), kb0 = a16_base(g, prob, true);
    a16b_load(k + kb0, L.koff, L, rk);
    a16b_load(q + qb0, L.qoff, L, rq);
    a16b_load(v + kb0, L.koff, L, rv);
  }
  for (; prob < g.nprob; prob += gridDim.x) {
    const int64_t qb0 = a16_base(g, prob, false);
    f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NB32; ++b) {
      bf16x8 kh, kl, qh, ql, vh, vl;
      a16b_split(rk.v[b], L.kok, (L.kvm >> (4 * b)) & 15u, kh, kl);
      a16b_split(rq.v[b], L.qok, 15u, qh, ql);
      st = a16_mma32(kh, kl, qh, ql, st);     // S^T[j][i]
      a16b_split(rv.v[b], L.kok, 15u, vh, vl);
      if (wr_row && 32 * b + 8 * lq < 16 * NB16) a16b_stash(sV, plane, L.wroff, b, vh, vl);
    }
    {
      const int nx = prob + gridDim.x;
      if (nx < g.nprob) {
        const int64_t qn = a16_base(g, nx, false), kn = a16_base(g, nx, true);
        a16b_load(k + kn, L.koff, L, rk);
        a16b_load(q + qn, L.qoff, L, rq);
        a16b_load(v + kn, L.koff, L, rv);
      }
    }
    float e[4], m = -INFINITY;
    bool ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 4 * lq + r;
      ok[r] = FULL || (j < g.Lk && (!g.causal || j <= lr));
      e[r] = ok[r] ? st[r] + bias[r] : -INFINITY;
      m = fmaxf(m, e[r]);
    }
    m = a16_gmax(m);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { e[r] = ok[r] ? __expf(e[r] - m) : 0.f; sum += e[r]; }
    sum = a16_gsum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      e[r] *= inv;
      if (p > 0.f) e[r] *= vptr_drop_scale(seed, site, a16_pidx(g, prob, h, lr, 4 * lq + r), p);
    }
    s16x4 ph, pl;
    a16_split4(e[0], e[1], e[2], e[3], ph, pl);
    asm volatile("" ::: "memory");
    const bool st_ok = L.qok && L.hlive;
#pragma unroll
    for (int b = 0; b < NB16; ++b) {
      const s16x4 z4 = {0, 0, 0, 0};
      s16x4 ah, al;
      a16b_tr(sV, plane, troff, b, ah, al);
      if (!rd_grp) ah = al = z4;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = a16_mma16(ah, al, ph, pl, acc);
      if (st_ok) a16b_st4(o, qb0 + orow, b, acc, 1.f, L, p16);
    }
    asm volatile("" ::: "memory");
  }
}

// dtable[e][h] += sum over workgroups of the workspace partials, in a fixed order: 16 thread groups take every 16th workgroup
// (8 independent loads in flight per thread), then one thread per entry adds the 16 sums
__global__ __launch_bounds__(1024) void attn16_dtable_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dtable, const int nwg, const int nhp, const int nh) {
  __shared__ float part[16][64];
  const int h = blockIdx.x, e = threadIdx.x & 63, qd = threadIdx.x >> 6;
  float s = 0.f;
  if (e < 49) {
    int x = qd;
    for (; x + 7 * 16 < nwg; x += 8 * 16) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = ws[((int64_t)(x + 16 * u) * nhp + h) * 52 + e];
      s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    for (; x < nwg; x += 16) s += ws[((int64_t)x * nhp + h) * 52 + e];
  }
  part[qd][e] = s;
  __syncthreads();
  if (qd == 0 && e < 49) {
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) a += part[u][e];
    dtable[(int64_t)e * nh + h] += a;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// host side (called by the entry points of attn.hip)
// ---------------------------------------------------------------------------------------------------------------------------
static int a16_mode() {   // read per call (tests switch modes inside one process).  VPTR_ATTN16: 0 = off; 1 (default) = second-generation
  const char* m = getenv("VPTR_ATTN_MFMA");   // forward and backward (C % 4 == 0 and 16-byte aligned tensors, else the first-generation
  if (m && atoi(m) != 1) return 0;             // forward + the fp32 vector backward); 2 = first-generation forward and backward; 4 = first-
  const char* e = getenv("VPTR_ATTN16");       // generation forward, vector backward.  VPTR_ATTN16_FWD1=1: mode 1 with the first-generation
  return e ? atoi(e) : 1;                      // forward.  An explicit VPTR_ATTN_MFMA=0 / 2 (vector / LDS-staged MFMA kernels everywhere)
}                                              // turns these kernels off
// Measured at the K64 shapes (tools/attn_bench.py, P16 outputs, dropout 0.1; copying the bytes of a forward call: 19 us):
//   forward,  first generation    24-25 us   (fp32 vector kernels 35-37)
//   forward,  second generation   20 / 23 us (window / temporal)
//   backward, first generation    80 / 73 us (window with bias-table gradient / temporal; vector kernels 95 / 66): bound by its 168
//                                 narrow memory instructions per (problem, head), see above
//   backward, second generation   49 / 47 us (window without a table gradient 42; with the table gradient through atomics instead of
//                                 the workspace 81)
bool vptr_attn16_ok(int kind, int Lq, int Lk, int C, int nh, int ws, int backward) {
  const int hd = C / nh;
  const int mode = a16_mode();
  if (mode == 0 || (backward && mode == 4)) return false;
  if (backward && mode == 1 && C % 4 != 0) return false;   // default: the second-generation backward, which needs 16-byte aligned rows
  if (kind == 0 && ws != 4) return false;
  return Lq >= 1 && Lk >= 1 && Lq <= 16 && Lk <= 16 && hd % 2 == 0 && C % 2 == 0 && hd <= 96;
}
static dim3 a16_grid(const A16Geom& g, const bool table_grad) {
  // 4-wave workgroups, 3 - 4 resident per CU, each looping over its share of the problems.  A few rounds of workgroups keep the tail
  // short; with a bias-table gradient fewer: every workgroup ends with 49 device-scope atomics per head onto the same 392 words, which
  // all land at the end of the launch (512 workgroups: +30 us, 768: +46, 2560: +90 on an 80 us kernel)
  const int groups = (g.nh + 3) / 4;
  const int want = (table_grad ? 256 * 2 : 256 * 4 * 3) / groups;
  return dim3(g.nprob < want ? g.nprob : want, groups);
}
static bool a16_full(const A16Geom& g) { return g.Lq == 16 && g.Lk == 16 && !g.causal; }
int vptr_attn16_fwd(const float* q, const float* k, const float* v, const float* table, const int64_t* rel_index, float* o, const A16Geom& g, float p,
                    const uint64_t* seed_dev, uint32_t site, int p16, hipStream_t st) {
  const bool full = a16_full(g);
  const uintptr_t fbits = reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(o);
  if (a16_mode() == 1 && g.C % 4 == 0 && (fbits & 15) == 0 && !getenv("VPTR_ATTN16_FWD1")) {   // second generation
    const int span = g.hd + ((g.hd & 3) ? 2 : 0);
    const int nb32 = (span + 31) / 32, nb16 = (span + 15) / 16;
    const int rows = ((g.Lq > g.Lk ? g.Lq : g.Lk) + 3) & ~3;
    const int groups = (g.nh + 3) / 4;
    const size_t lds = (size_t)4 * 2 * rows * nb16 * 32;
    int gx = 256 * 3 / groups;
    if (gx < 1) gx = 1;
    if (g.nprob < gx) gx = g.nprob;
    const dim3 grid(gx, groups), block(256);
#define A16F2(A, B)                                                                                                                  \
  do {                                                                                                                               \
    if (full) attn16_fwd2_kernel<A, B, true><<<grid, block, lds, st>>>(q, k, v, table, rel_index, o, g, p, seed_dev, site, p16, rows);   \
    else attn16_fwd2_kernel<A, B, false><<<grid, block, lds, st>>>(q, k, v, table, rel_index, o, g, p, seed_dev, site, p16, rows);       \
  } while (0)
    if (nb32 == 1 && nb16 == 1) A16F2(1, 1);
    else if (nb32 == 1) A16F2(1, 2);
    else if (nb32 == 2 && nb16 == 3) A16F2(2, 3);
    else if (nb32 == 2) A16F2(2, 4);
    else if (nb16 == 5) A16F2(3, 5);
    else if (nb32 == 3) A16F2(3, 6);
    else A16F2(4, 7);
#undef A16F2
    return 0;
  }
  const int nb32 = (g.hd + 31) / 32, nb16 = (g.hd + 15) / 16;
  const dim3 grid = a16_grid(g, false), block(256);
#define A16F(A, B)                                                                                                        \
  do {                                                                                                                    \
    if (full) attn16_fwd_kernel<A, B, true><<<grid, block, 0, st>>>(q, k, v, table, rel_index, o, g, p, seed_dev, site, p16);   \
    else attn16_fwd_kernel<A, B, false><<<grid, block, 0, st>>>(q, k, v, table, rel_index, o, g, p, seed_dev, site, p16);       \
  } while (0)
  if (nb32 == 1 && nb16 == 1) A16F(1, 1);
  else if (nb32 == 1) A16F(1, 2);
  else if (nb32 == 2 && nb16 == 3) A16F(2, 3);
  else if (nb32 == 2) A16F(2, 4);
  else if (nb16 == 5) A16F(3, 5);
  else A16F(3, 6);
#undef A16F
  return 0;
}
int vptr_attn16_bwd(const float* q, const float* k, const float* v, const float* table, const int64_t* rel_index, const float* dout, float* dq, float* dk,
                    float* dv, float* dtable, const A16Geom& g, float p, const uint64_t* seed_dev, uint32_t site, float dq_scale, int p16,
                    float* dtable_ws, int64_t ws_floats, hipStream_t st) {
  const bool full = a16_full(g);
  const uintptr_t bits = reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(dout) |
                         reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(dk) | reinterpret_cast<uintptr_t>(dv);
  if (a16_mode() != 2 && g.C % 4 == 0 && (bits & 15) == 0) {   // second generation: aligned heads, LDS stash, prefetch
    const int span = g.hd + ((g.hd & 3) ? 2 : 0);                // a head that starts on an odd pair begins 2 channels early
    const int nb32 = (span + 31) / 32, nb16 = (span + 15) / 16;
    const int rows = ((g.Lq > g.Lk ? g.Lq : g.Lk) + 3) & ~3;
    const int groups = (g.nh + 3) / 4;
    const size_t lds = (size_t)4 * 6 * rows * nb16 * 32;
    const int per_cu = 2;   // 219 - 224 registers: two workgroups (8 waves) per CU, each with the next problem's loads in flight
    int gx = 256 * per_cu / groups;
    if (gx < 1) gx = 1;
    if (g.nprob < gx) gx = g.nprob;
    float* ws = nullptr;
    if (dtable && dtable_ws && ws_floats >= (int64_t)gx * groups * 4 * 52) ws = dtable_ws;
    const dim3 grid(gx, groups), block(256);
#define A16B2(A, B)                                                                                                                                \
  do {                                                                                                                                             \
    if (full) attn16_bwd2_kernel<A, B, true><<<grid, block, lds, st>>>(q, k, v, table, rel_index, dout, dq, dk, dv, dtable, ws, g, p, seed_dev, site, dq_scale, p16, rows);  \
    else attn16_bwd2_kernel<A, B, false><<<grid, block, lds, st>>>(q, k, v, table, rel_index, dout, dq, dk, dv, dtable, ws, g, p, seed_dev, site, dq_scale, p16, rows);      \
  } while (0)
    if (nb32 == 1 && nb16 == 1) A16B2(1, 1);
    else if (nb32 == 1) A16B2(1, 2);
    else if (nb32 == 2 && nb16 == 3) A16B2(2, 3);
    else if (nb32 == 2) A16B2(2, 4);
    else if (nb16 == 5) A16B2(3, 5);
    else if (nb32 == 3) A16B2(3, 6);
    else A16B2(4, 7);
#undef A16B2
    if (ws) attn16_dtable_reduce_kernel<<<g.nh, 1024, 0, st>>>(ws, dtable, gx, groups * 4, g.nh);
    return 0;
  }
  const int nb32 = (g.hd + 31) / 32, nb16 = (g.hd + 15) / 16;
  const dim3 grid = a16_grid(g, dtable != nullptr), block(256);
#define A16B(A, B)                                                                                                                                                   \
  do {                                                                                                                                                               \
    if (full) attn16_bwd_kernel<A, B, true><<<grid, block, 0, st>>>(q, k, v, table, rel_index, dout, dq, dk, dv, dtable, g, p, seed_dev, site, dq_scale, p16);      \
    else attn16_bwd_kernel<A, B, false><<<grid, block, 0, st>>>(q, k, v, table, rel_index, dout, dq, dk, dv, dtable, g, p, seed_dev, site, dq_scale, p16);          \
  } while (0)
  if (nb32 == 1 && nb16 == 1) A16B(1, 1);
  else if (nb32 == 1) A16B(1, 2);
  else if (nb32 == 2 && nb16 == 3) A16B(2, 3);
  else if (nb32 == 2) A16B(2, 4);
  else if (nb16 == 5) A16B(3, 5);
  else A16B(3, 6);
#undef A16B
  return 0;
}
// floats of workspace that make the bias-table gradient of a backward call atomics-free (at most 3 workgroups per CU and head group)
int64_t vptr_attn16_ws_floats(int nh) { (void)nh; return (int64_t)256 * 2 * 4 * 52; }
