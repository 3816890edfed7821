// Attention cores on the matrix units (gfx950): softmax(Q K^T + bias) V for many small independent problems -- the local-window
// attention of MultiHeadAttentionRPE (16- or 64-token windows, relative-position bias; MultiHeadAttentionRPE.py:586-682 with the
// window partition of VidHRFormer_modules.py:497-525 as index arithmetic) and the temporal / encoder-decoder nn.MultiheadAttention
// cores (up to 64 time steps, causal flag; VidHRFormer_modules.py:74-84,183-187,199-206).
//
// One problem = (group g, head h): Lq query rows and Lk key rows of the token-major [rows, C] tensors (window: the ws x ws tokens
// of window g; temporal: the Tq / Tk time steps of pixel g).  A wave owns one 16-row block of queries of one problem; the waves of
// a workgroup that serve the same problem share the LDS image of V (forward) / K (backward).
//
// Arithmetic: every product runs as split-bf16 (x = hi + lo; lo.hi + hi.lo + hi.hi, fp32 accumulate) on
// v_mfma_f32_16x16x32_bf16 -- the same fp32-class scheme as the GEMMs.
//   S = Q K^T    A and B fragments want 8 consecutive head channels of one row per lane: contiguous in HBM, so Q and K go
//                straight from global memory into registers (float2 loads, split in registers), no LDS.
//   O = P V      P leaves the MFMA in the C/D layout (one column, four rows per lane) and re-enters as an A operand (one row, eight
//                columns) through a small per-wave LDS tile; V must be read "down a column" (8 keys of one channel per lane): its
//                bf16 hi / lo image [keys][16 channels] is read with ds_read_b64_tr_b16.
// The K index of both PV operands uses the same permuted map (element e of lane group q <-> key 32 kj + 16 (e >> 2) + 4 q + (e & 3))
// so that the transposing reads of a wave cover 512 contiguous bytes (conflict-free).
#include "attn_mfma.h"

typedef __attribute__((ext_vector_type(4))) short am_s16x4;
typedef __attribute__((ext_vector_type(8))) short am_s16x8;
typedef __attribute__((ext_vector_type(2))) uint32_t am_u32x2;

#define AM_MAXL 64

// window row of token l of window `win` (ws x ws windows, row-major over (frame, qh, qw))
// Row of sequence element l of problem g = base(g) + off(l): the base is workgroup-uniform (scalar unit), the per-lane part is a shift and a
// mask for power-of-two windows (round 5: the divisions by ws / nqw / HW per call were a sixth of the kernels' 2 - 4 k VALU instructions
// per wave).  All tensors the launchers admit have fewer than 2^31 elements, so row * C stays in 32 bits.
__device__ __forceinline__ int am_win_row(int win, int l, int H, int W, int ws, int nqh, int nqw) {
  const int b = win / (nqh * nqw), r = win - b * (nqh * nqw);
  const int qh = r / nqw, qw = r - qh * nqw;
  const int ph = l / ws, pw = l - ph * ws;
  return (b * H + qh * ws + ph) * W + qw * ws + pw;
}
__device__ __forceinline__ int am_row_off(const AmGeom& gm, const int l, const bool key) {
  if (gm.mode == 0) {
    if (gm.ws_shift >= 0) return (l >> gm.ws_shift) * gm.W + (l & (gm.ws - 1));
    const int ph = l / gm.ws;
    return ph * gm.W + (l - ph * gm.ws);
  }
  return l * gm.HW;
}
__device__ __forceinline__ int am_row_base(const AmGeom& gm, const int g, const bool key) {   // uniform
  if (gm.mode == 0) return am_win_row(g, 0, gm.H, gm.W, gm.ws, gm.H / gm.ws, gm.W / gm.ws);
  const int n = g / gm.HW, pix = g - n * gm.HW;
  return n * (key ? gm.Tk : gm.Tq) * gm.HW + pix;
}
__device__ __forceinline__ int am_qrow(const AmGeom& gm, int g, int i) { return am_row_base(gm, g, false) + am_row_off(gm, i, false); }
__device__ __forceinline__ int am_krow(const AmGeom& gm, int g, int j) { return am_row_base(gm, g, true) + am_row_off(gm, j, true); }
// dropout scale of score element `idx` (flat index of the tensor the reference drops): the 32-bit form of vptr_drop_scale for indices below
// 2^32 (vptr_hash3 adds (idx >> 32) * const = 0 there: the SAME hash, without the 64-bit index arithmetic)
__device__ __forceinline__ float am_drop_scale(const AmGeom& gm, const uint64_t seed, const uint32_t site, const int prob, const int i, const int j, const float p) {
  if (gm.idx32) {
    const uint32_t idx = ((uint32_t)prob * (uint32_t)gm.Lq + (uint32_t)i) * (uint32_t)gm.Lk + (uint32_t)j;
    const uint32_t key = vptr_mix32((uint32_t)seed ^ vptr_mix32((uint32_t)(seed >> 32) + (site + 1u) * 0x9E3779B9u));
    const uint32_t hsh = vptr_mix32(idx ^ key);
    const uint32_t thr = (uint32_t)((double)p * 4294967296.0);
    return (hsh >= thr) ? 1.0f / (1.0f - p) : 0.0f;
  }
  return vptr_drop_scale(seed, site, ((uint64_t)prob * gm.Lq + i) * gm.Lk + j, p);
}

// 8 consecutive channels d0 .. d0+7 of one row -> bf16 hi / lo fragments; channels >= hd and rows that do not exist read as zero
// (all loads unconditional from clamped addresses; hd even, so channel pairs are valid or invalid as a whole)
__device__ __forceinline__ void am_load_frag(const float* __restrict__ row, const int d0, const int hd, const bool rowok, bf16x8& hi, bf16x8& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int d = d0 + 2 * e;
    const float2 v = *reinterpret_cast<const float2*>(row + min(d, hd - 2));
    const bool ok = rowok && d < hd;
    vptr_split2(ok ? v.x : 0.f, ok ? v.y : 0.f, h[e], l[e]);
  }
  typedef __attribute__((ext_vector_type(4))) uint32_t u4;
  const u4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
  hi = __builtin_bit_cast(bf16x8, hv);
  lo = __builtin_bit_cast(bf16x8, lv);
}

__device__ __forceinline__ f32x4 am_mfma3(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}

// sum / max over the 16 lanes that share lane >> 4 (one row group of the C/D layout)
__device__ __forceinline__ float am_row16_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}
__device__ __forceinline__ float am_row16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64)); v = fmaxf(v, __shfl_xor(v, 8, 64));
  return v;
}

// LDS image of a [rows][hd] operand for transposing reads: per 16-channel block df and plane, [LKP rows][16 channels] bf16
// (32-byte rows).  Offsets in bytes inside one problem slot's region.
__device__ __forceinline__ int am_img_off(const int LKP, const int df, const int plane, const int j, const int dlo) {
  return ((df * 2 + plane) * LKP + j) * 32 + dlo * 2;
}
// B fragment (8 keys of channel 16 df + lr per lane; keys 32 kj + 16 (e >> 2) + 4 lq + (e & 3)) of one plane
__device__ __forceinline__ bf16x8 am_tr_frag(const unsigned char* img, const int LKP, const int df, const int plane, const int kj, const int lr, const int lq) {
  const unsigned char* p0 = img + am_img_off(LKP, df, plane, 32 * kj + 4 * lq + (lr >> 2), (lr & 3) * 4);
  const am_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) am_s16x4*)(p0));
  const am_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) am_s16x4*)(p0 + 16 * 32));
  const am_s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}
// A fragment of a [16][LKP] bf16 tile (pitch bytes) with the same key map
__device__ __forceinline__ bf16x8 am_tile_frag(const unsigned char* tile, const int pitch, const int kj, const int lr, const int lq) {
  const unsigned char* p0 = tile + lr * pitch + (32 * kj + 4 * lq) * 2;
  const am_u32x2 a = *reinterpret_cast<const am_u32x2*>(p0);
  const am_u32x2 b = *reinterpret_cast<const am_u32x2*>(p0 + 32);
  typedef __attribute__((ext_vector_type(4))) uint32_t u4;
  const u4 c = {a[0], a[1], b[0], b[1]};
  return __builtin_bit_cast(bf16x8, c);
}

// stage rows of `src` (keys of problem g, head h) as the transposable image; `nthr` threads of the slot cooperate (tid 0 .. nthr-1).
// Rows >= Lk and channels >= hd are zero-filled without loads (P is zero there, but 0 x stale-NaN would poison the product).
__device__ __forceinline__ void am_stage_img(unsigned char* img, const float* __restrict__ src, const AmGeom& gm, const int g, const int h, const int LKP,
                                             const int NDF, const int tid, const int nthr, const bool query_rows = false) {
  const int NP = NDF * 8;   // channel pairs per row including the zero padding up to 16 NDF
  const int L = query_rows ? gm.Lq : gm.Lk;
  const int total = L * NP;
  // BATCHES of loads (round 5): the loop used to be load -> wait -> split -> write per item, i.e. up to 10 - 40 serialised memory round
  // trips per thread and staged tensor (the trip count is not a compile-time constant, so nothing was unrolled); now up to AM_STAGE_U
  // unconditional loads (clamped index) are in flight before the first one is consumed
  constexpr int U = 10;
  const float* const hbase = src + h * gm.hd;
  for (int base = tid; base < total; base += nthr * U) {
    float2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (base - tid + u * nthr < total) {   // slot-uniform: batches beyond the tensor issue nothing (a 10-row key tensor is 2 items per thread)
        const int idx = min(base + u * nthr, total - 1);
        const int j = idx / NP, d = 2 * (idx - j * NP);
        v[u] = *reinterpret_cast<const float2*>(hbase + (query_rows ? am_qrow(gm, g, j) : am_krow(gm, g, j)) * gm.C + min(d, gm.hd - 2));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * nthr;
      if (idx < total) {
        const int j = idx / NP, d = 2 * (idx - j * NP);
        const bool ok = d < gm.hd;
        uint32_t hi, lo;
        vptr_split2(ok ? v[u].x : 0.f, ok ? v[u].y : 0.f, hi, lo);
        *reinterpret_cast<uint32_t*>(img + am_img_off(LKP, d >> 4, 0, j, d & 15)) = hi;
        *reinterpret_cast<uint32_t*>(img + am_img_off(LKP, d >> 4, 1, j, d & 15)) = lo;
      }
    }
  }
  const int ppb = (LKP - L) * 2;   // 16-byte pieces of the pad rows of one (df, plane) block (32 B per row)
  for (int idx = tid; idx < 2 * NDF * ppb; idx += nthr) {
    const int blk = idx / ppb, rem = idx - blk * ppb;
    *reinterpret_cast<uint4*>(img + (blk * LKP + L) * 32 + rem * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
}

// (Measured and dropped, same-box A/B in profiles/r05_attn64_ab.log: requesting the dO / Q rows of the later key-block phases at kernel start
// (registers) and staging K and V as one batch -- 285 / 248 us against 265 / 237 us for the 64-token window / T = 29 backward: the 40 - 80
// extra live registers cost more than the hidden round trips buy.)

// A / B operand fragment (8 consecutive channels 32 ks + 8 lq .. of row `row`) of one plane, read from a transposable image instead of
// global memory: the channels of a row are contiguous inside their 16-channel block, so it is one 16-byte read.  Second generation of
// the score products (round 5): the four query-block waves of a problem used to load and split the SAME K rows from global memory (48
// 8-byte loads + 288 VALU per wave); now the problem's K (and V, dO) rows are staged once and every wave reads fragments from LDS.
template <int NDF>
__device__ __forceinline__ bf16x8 am_img_frag(const unsigned char* img, const int LKP, const int ks, const int plane, const int row, const int lq) {
  const int df = 2 * ks + (lq >> 1);
  typedef __attribute__((ext_vector_type(4))) uint32_t u4;
  u4 v = {0u, 0u, 0u, 0u};
  if (df < NDF) v = *reinterpret_cast<const u4*>(img + am_img_off(LKP, df, plane, row, (lq & 1) * 8));
  return __builtin_bit_cast(bf16x8, v);
}

// S block of one wave: acc[jf] = Q[16 x hd] . K[16 jf .. +15][hd]^T for jf < njf, then bias, masks and the row softmax -> probabilities
// pr[jf][r] of element (i = 16 qb + 4 lq + r, j = 16 jf + lr) (0 for keys that do not exist); returns nothing else.
template <int NKS, int NDF = 0>   // NDF > 0: K fragments come from the transposable image `kimg` (LKP rows per block) instead of global memory
__device__ __forceinline__ void am_scores(const AmGeom& gm, const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ table,
                                          const int64_t* __restrict__ rel_index, const int g, const int h, const int qb, const int njf, const int causal,
                                          const int lr, const int lq, float (&pr)[4][4], const unsigned char* kimg = nullptr, const int LKP = 0) {
  bf16x8 qh[NKS], ql[NKS];
  {
    const int i = qb * 16 + lr;
    const float* row = q + am_qrow(gm, g, min(i, gm.Lq - 1)) * gm.C + h * gm.hd;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) am_load_frag(row, 32 * ks + 8 * lq, gm.hd, i < gm.Lq, qh[ks], ql[ks]);
  }
  f32x4 acc[4];
#pragma unroll
  for (int jf = 0; jf < 4; ++jf) {
    acc[jf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (jf < njf) {
      const int j = jf * 16 + lr;
      if (NDF > 0) {   // rows >= Lk and channels >= hd of the image are zero
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const bf16x8 kh = am_img_frag<NDF>(kimg, LKP, ks, 0, j, lq), kl = am_img_frag<NDF>(kimg, LKP, ks, 1, j, lq);
          acc[jf] = am_mfma3(qh[ks], ql[ks], kh, kl, acc[jf]);
        }
      } else {
        const float* row = k + am_krow(gm, g, min(j, gm.Lk - 1)) * gm.C + h * gm.hd;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          bf16x8 kh, kl;
          am_load_frag(row, 32 * ks + 8 * lq, gm.hd, j < gm.Lk, kh, kl);
          acc[jf] = am_mfma3(qh[ks], ql[ks], kh, kl, acc[jf]);
        }
      }
    }
  }
  // C/D layout: lane (lr, lq) holds rows i = 16 qb + 4 lq + r, column j = 16 jf + lr
  float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int jf = 0; jf < 4; ++jf) {
    const int j = jf * 16 + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = qb * 16 + 4 * lq + r;
      float s = -INFINITY;
      if (jf < njf && j < gm.Lk && !(causal && j > i)) {
        s = acc[jf][r];
        if (table) s += table[rel_index[min(i, gm.Lq - 1) * gm.Lk + j] * gm.nh + h];
      }
      pr[jf][r] = s;
      m[r] = fmaxf(m[r], s);
    }
  }
  float sum[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    m[r] = am_row16_max(m[r]);
    sum[r] = 0.f;
  }
#pragma unroll
  for (int jf = 0; jf < 4; ++jf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = pr[jf][r] == -INFINITY ? 0.f : __expf(pr[jf][r] - m[r]);
      pr[jf][r] = e;
      sum[r] += e;
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) sum[r] = 1.f / am_row16_sum(sum[r]);
#pragma unroll
  for (int jf = 0; jf < 4; ++jf)
#pragma unroll
    for (int r = 0; r < 4; ++r) pr[jf][r] *= sum[r];
}

// store of a TRANSPOSED output block (operands swapped in the MFMA: C^T[d][i], lane (lr, lq) owns channels 16 df + 4 lq .. + 3 of row
// lr): two 8-byte pair stores per lane (fp32) or 4 + 4 bytes per P16 plane -- a head of width 66 starts at an even channel, so pairs are
// aligned and never straddle a P16 granule -- instead of four 4-byte (P16: eight 2-byte) stores in the untransposed layout
__device__ __forceinline__ void am_store_quad(float* __restrict__ dst, const int64_t rowe, const int d0, const int hd, const f32x4 v, const float scale,
                                              const int p16) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int d = d0 + 2 * c;
    if (d < hd) {   // hd even: a pair is valid or invalid as a whole
      const float a = v[2 * c] * scale, b = v[2 * c + 1] * scale;
      if (p16) vptr_p16_store2(reinterpret_cast<unsigned char*>(dst), rowe + d, a, b);
      else *reinterpret_cast<float2*>(dst + rowe + d) = make_float2(a, b);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward.  Workgroup = 4 waves; NQBR (1, 2 or 4) consecutive waves serve the query blocks of one problem and share its V image.
// LDS: [4 / NQBR problem slots][V image: 2 NDF planes x LKP x 32 B]  then  [4 waves][P tile: 2 planes x 16 x pitch]
// ---------------------------------------------------------------------------------------------------------------------
template <int NKS, int NDF>
__global__ __launch_bounds__(256) void attn_mfma_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                            const float* __restrict__ table, const int64_t* __restrict__ rel_index, float* __restrict__ o,
                                                            const AmGeom gm, const int nqbr, const int causal, const float p, const uint64_t* seed_dev,
                                                            const uint32_t site, const int p16) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char am_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  const int NQB = (gm.Lq + 15) >> 4, njf = (gm.Lk + 15) >> 4, nkj = (gm.Lk + 31) >> 5, LKP = nkj * 32;
  const int slots = 4 / nqbr, slot = wave / nqbr, qb = wave - slot * nqbr;
  const int prob = blockIdx.x * slots + slot;                 // problem = g * nh + h
  const bool pvalid = prob < gm.groups * gm.nh;
  const int g = pvalid ? prob / gm.nh : 0, h = pvalid ? prob - g * gm.nh : 0;
  const bool active = pvalid && qb < NQB;
  const int img_bytes = 2 * NDF * LKP * 32, pitch = LKP * 2 + 16;
  unsigned char* img = am_smem + slot * 2 * img_bytes;          // V image, then the K image
  unsigned char* kimg = img + img_bytes;
  // the P tiles of a problem's waves take over its K image once every wave has its scores (one more barrier, a third less LDS: three
  // workgroups per CU at 64 keys x 66 channels) when they fit there; otherwise they have a region of their own behind the images
  const bool tiles_in_kimg = nqbr * 2 * 16 * pitch <= img_bytes;
  unsigned char* ptile = tiles_in_kimg ? kimg + qb * (2 * 16 * pitch) : am_smem + slots * 2 * img_bytes + wave * (2 * 16 * pitch);

  if (pvalid) {   // both operands of the problem staged once (all loads of a thread in flight together), shared by its query-block waves
    const int stid = tid - slot * nqbr * 64, snthr = nqbr * 64;
    am_stage_img(kimg, k, gm, g, h, LKP, NDF, stid, snthr);
    am_stage_img(img, v, gm, g, h, LKP, NDF, stid, snthr);
  }
  __syncthreads();
  float pr[4][4];
  if (active) am_scores<NKS, NDF>(gm, q, k, table, rel_index, g, h, qb, njf, causal, lr, lq, pr, kimg, LKP);
  if (tiles_in_kimg) __syncthreads();   // workgroup-uniform: nobody reads the K image any more
  if (active) {
    uint64_t seed = 0;
    if (p > 0.f) seed = *seed_dev;
#pragma unroll
    for (int jf = 0; jf < 4; ++jf) {
      if (jf * 16 < LKP) {
        const int j = jf * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int il = 4 * lq + r, i = qb * 16 + il;
          float x = pr[jf][r];
          if (p > 0.f && i < gm.Lq && j < gm.Lk) x *= am_drop_scale(gm, seed, site, prob, i, j, p);
          uint32_t hi, lo;
          vptr_split2(x, 0.f, hi, lo);
          *reinterpret_cast<uint16_t*>(ptile + il * pitch + j * 2) = (uint16_t)hi;
          *reinterpret_cast<uint16_t*>(ptile + 16 * pitch + il * pitch + j * 2) = (uint16_t)lo;
        }
      }
    }
  }
  if (!active) return;
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the P tile is wave-private -- its own writes have landed, no barrier needed
  f32x4 oacc[NDF];                      // O^T blocks (operands swapped: A = V^T fragment, B = P fragment): see am_store_quad
#pragma unroll
  for (int df = 0; df < NDF; ++df) oacc[df] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kj = 0; kj < 2; ++kj) {
    if (kj < nkj) {
      const bf16x8 ph = am_tile_frag(ptile, pitch, kj, lr, lq), pl = am_tile_frag(ptile + 16 * pitch, pitch, kj, lr, lq);
#pragma unroll
      for (int df = 0; df < NDF; ++df) {
        const bf16x8 vh = am_tr_frag(img, LKP, df, 0, kj, lr, lq), vl = am_tr_frag(img, LKP, df, 1, kj, lr, lq);
        oacc[df] = am_mfma3(vh, vl, ph, pl, oacc[df]);
      }
    }
  }
  const int i = qb * 16 + lr;
  if (i < gm.Lq) {
    const int64_t rowe = (int64_t)(am_qrow(gm, g, i) * gm.C + h * gm.hd);
#pragma unroll
    for (int df = 0; df < NDF; ++df) am_store_quad(o, rowe, df * 16 + 4 * lq, gm.hd, oacc[df], 1.f, p16);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward.  Per wave (problem, 16-query block): recompute P; dP = dO V^T (both operands straight from HBM); dS = P o (dP - rowsum);
//   dQ block = dS . K        dS re-enters as an A operand through the per-wave LDS tile, K through the slot's transposable image;
//   dV      += P_drop^T . dO  } contraction over this wave's 16 query rows on v_mfma_f32_16x16x16_bf16: the C/D-layout registers of
//   dK      += dS^T . Q       } P / dS ARE the A operand of the transposed product (row = key = lane & 15, k = 4 (lane >> 4) + r);
//                               B = 4 query rows of one channel per lane: scalar loads of dO / Q.
// With several query blocks per problem the partial dV / dK of the waves meet in an LDS accumulator (ds_add_f32), one after the
// other; with one block per problem they are stored from the registers.  The bias-table gradient is summed per workgroup in LDS.
// LDS: [slots][K image | accumulator LKP x 16 NDF fp32 | bias table gradient]  then  [4 waves][dS tile]
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short am_b16x4;

__device__ __forceinline__ f32x4 am_mfma3_k16(const am_b16x4 ah, const am_b16x4 al, const am_b16x4 bh, const am_b16x4 bl, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc, 0, 0, 0);
}
__device__ __forceinline__ void am_split4(const float (&x)[4], am_b16x4& hi, am_b16x4& lo) {
  uint32_t h0, l0, h1, l1;
  vptr_split2(x[0], x[1], h0, l0);
  vptr_split2(x[2], x[3], h1, l1);
  const am_u32x2 hv = {h0, h1}, lv = {l0, l1};
  hi = __builtin_bit_cast(am_b16x4, hv);
  lo = __builtin_bit_cast(am_b16x4, lv);
}

template <int NKS, int NDF>
__global__ __launch_bounds__(256) void attn_mfma_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                            const float* __restrict__ table, const int64_t* __restrict__ rel_index,
                                                            const float* __restrict__ dout, float* __restrict__ dq, float* __restrict__ dk,
                                                            float* __restrict__ dv, float* __restrict__ dtable, const AmGeom gm, const int nqbr,
                                                            const int causal, const float p, const uint64_t* seed_dev, const uint32_t site,
                                                            const float dq_scale, const int p16, const int ntab) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char am_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  const int NQB = (gm.Lq + 15) >> 4, njf = (gm.Lk + 15) >> 4, nkj = (gm.Lk + 31) >> 5, LKP = nkj * 32;
  const int slots = 4 / nqbr, slot = wave / nqbr, qb = wave - slot * nqbr;
  const int prob = blockIdx.x * slots + slot;
  const bool pvalid = prob < gm.groups * gm.nh;
  const int g = pvalid ? prob / gm.nh : 0, h = pvalid ? prob - g * gm.nh : 0;
  const bool active = pvalid && qb < NQB;
  const bool shared_acc = nqbr > 1;
  const int img_bytes = 2 * NDF * LKP * 32, acc_bytes = shared_acc ? LKP * NDF * 64 : 0, tab_bytes = dtable ? ((ntab * 4 + 15) & ~15) : 0;
  const int slot_bytes = img_bytes + acc_bytes + tab_bytes, pitch = LKP * 2 + 16;
  unsigned char* img = am_smem + slot * slot_bytes;
  float* accb = reinterpret_cast<float*>(img + img_bytes);
  float* stab = reinterpret_cast<float*>(img + img_bytes + acc_bytes);
  unsigned char* stile = am_smem + slots * slot_bytes + wave * (2 * 16 * pitch);
  const int stid = tid - slot * nqbr * 64, snthr = nqbr * 64;   // thread index inside the slot

  if (pvalid) am_stage_img(img, k, gm, g, h, LKP, NDF, stid, snthr);
  for (int e = stid; e < (acc_bytes + tab_bytes) / 4; e += snthr) accb[e] = 0.f;   // accumulator and table gradient are contiguous

  float pr[4][4], ds[4][4], pd[4][4];   // P, dS and the dropped P of this wave's block (C/D layout)
#pragma unroll
  for (int jf = 0; jf < 4; ++jf)
#pragma unroll
    for (int r = 0; r < 4; ++r) pr[jf][r] = ds[jf][r] = pd[jf][r] = 0.f;
  if (active) {
    am_scores<NKS>(gm, q, k, table, rel_index, g, h, qb, njf, causal, lr, lq, pr);
    // dP_drop = dO V^T
    bf16x8 gh[NKS], gl[NKS];
    {
      const int i = qb * 16 + lr;
      const float* row = dout + am_qrow(gm, g, min(i, gm.Lq - 1)) * gm.C + h * gm.hd;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) am_load_frag(row, 32 * ks + 8 * lq, gm.hd, i < gm.Lq, gh[ks], gl[ks]);
    }
    uint64_t seed = 0;
    if (p > 0.f) seed = *seed_dev;
    float delta[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jf = 0; jf < 4; ++jf) {
      if (jf < njf) {
        const int j = jf * 16 + lr;
        const float* row = v + am_krow(gm, g, min(j, gm.Lk - 1)) * gm.C + h * gm.hd;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          bf16x8 vh, vl;
          am_load_frag(row, 32 * ks + 8 * lq, gm.hd, j < gm.Lk, vh, vl);
          acc = am_mfma3(gh[ks], gl[ks], vh, vl, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = qb * 16 + 4 * lq + r;
          float sc = 1.f;
          if (p > 0.f && i < gm.Lq && j < gm.Lk) sc = am_drop_scale(gm, seed, site, prob, i, j, p);
          pd[jf][r] = pr[jf][r] * sc;
          ds[jf][r] = acc[r] * sc;          // dP
          delta[r] += pr[jf][r] * ds[jf][r];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) delta[r] = am_row16_sum(delta[r]);
#pragma unroll
    for (int jf = 0; jf < 4; ++jf)
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[jf][r] = pr[jf][r] * (ds[jf][r] - delta[r]);
    // dS -> per-wave tile (A operand of dQ = dS . K)
#pragma unroll
    for (int jf = 0; jf < 4; ++jf) {
      if (jf * 16 < LKP) {
        const int j = jf * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int il = 4 * lq + r;
          uint32_t hi, lo;
          vptr_split2(ds[jf][r], 0.f, hi, lo);
          *reinterpret_cast<uint16_t*>(stile + il * pitch + j * 2) = (uint16_t)hi;
          *reinterpret_cast<uint16_t*>(stile + 16 * pitch + il * pitch + j * 2) = (uint16_t)lo;
        }
      }
    }
  }
  __syncthreads();   // K image, zeroed accumulators, dS tile
  if (active && dtable) {   // bias-table gradient: dS summed by relative position (LDS atomics, flushed once per workgroup and head)
#pragma unroll
    for (int jf = 0; jf < 4; ++jf) {
      const int j = jf * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = qb * 16 + 4 * lq + r;
        if (jf < njf && j < gm.Lk && i < gm.Lq) atomicAdd(&stab[rel_index[i * gm.Lk + j]], ds[jf][r]);
      }
    }
  }
  if (active) {   // dQ block
    f32x4 qacc[NDF];
#pragma unroll
    for (int df = 0; df < NDF; ++df) qacc[df] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kj = 0; kj < 2; ++kj) {
      if (kj < nkj) {
        const bf16x8 sh = am_tile_frag(stile, pitch, kj, lr, lq), sl = am_tile_frag(stile + 16 * pitch, pitch, kj, lr, lq);
#pragma unroll
        for (int df = 0; df < NDF; ++df) {
          const bf16x8 kh = am_tr_frag(img, LKP, df, 0, kj, lr, lq), kl = am_tr_frag(img, LKP, df, 1, kj, lr, lq);
          qacc[df] = am_mfma3(sh, sl, kh, kl, qacc[df]);
        }
      }
    }
#pragma unroll
    for (int df = 0; df < NDF; ++df) {
      const int d = df * 16 + lr;
      if (d < gm.hd) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = qb * 16 + 4 * lq + r;
          if (i < gm.Lq) {
            const int64_t e = am_qrow(gm, g, i) * gm.C + h * gm.hd + d;
            if (p16) vptr_p16_store1(reinterpret_cast<unsigned char*>(dq), e, qacc[df][r] * dq_scale);
            else dq[e] = qacc[df][r] * dq_scale;
          }
        }
      }
    }
  }
  // dV (pass 0: A = dropped P, B = dO) and dK (pass 1: A = dS, B = Q), contraction over this wave's 16 query rows
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const float* bsrc = pass == 0 ? dout : q;
    float* dst = pass == 0 ? dv : dk;
    if (active) {
      am_b16x4 bh[NDF], bl[NDF];
#pragma unroll
      for (int df = 0; df < NDF; ++df) {
        const int d = df * 16 + lr;
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = qb * 16 + 4 * lq + e;
          const float t = bsrc[am_qrow(gm, g, min(i, gm.Lq - 1)) * gm.C + h * gm.hd + min(d, gm.hd - 1)];
          x[e] = (i < gm.Lq && d < gm.hd) ? t : 0.f;
        }
        am_split4(x, bh[df], bl[df]);
      }
#pragma unroll
      for (int jf = 0; jf < 4; ++jf) {
        if (jf < njf) {
          am_b16x4 ah, al;
          am_split4(pass == 0 ? pd[jf] : ds[jf], ah, al);
#pragma unroll
          for (int df = 0; df < NDF; ++df) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = am_mfma3_k16(ah, al, bh[df], bl[df], acc);
            const int d = df * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = jf * 16 + 4 * lq + r;
              if (shared_acc) {
                atomicAdd(&accb[j * (NDF * 16) + d], acc[r]);
              } else if (j < gm.Lk && d < gm.hd) {
                const int64_t e = am_krow(gm, g, j) * gm.C + h * gm.hd + d;
                if (p16) vptr_p16_store1(reinterpret_cast<unsigned char*>(dst), e, acc[r]);
                else dst[e] = acc[r];
              }
            }
          }
        }
      }
    }
    if (shared_acc) {   // workgroup-uniform
      __syncthreads();
      if (pvalid) {
        for (int e = stid; e < gm.Lk * (NDF * 16); e += snthr) {
          const int j = e / (NDF * 16), d = e - j * (NDF * 16);
          const float val = accb[e];
          accb[e] = 0.f;
          if (d < gm.hd) {
            const int64_t ge = am_krow(gm, g, j) * gm.C + h * gm.hd + d;
            if (p16) vptr_p16_store1(reinterpret_cast<unsigned char*>(dst), ge, val);
            else dst[ge] = val;
          }
        }
      }
      __syncthreads();
    }
  }
  if (dtable && pvalid) {   // every wave passed the barrier after the table atomics (shared_acc) or is alone in its slot
    if (!shared_acc) __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): own LDS atomics done (single wave per slot)
    for (int t = stid; t < ntab; t += snthr) {
      const float val = stab[t];
      if (val != 0.f) unsafeAtomicAdd(dtable + (int64_t)t * gm.nh + h, val);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward for problems with more than 16 query rows (64-token windows, T = 29 ... 50): the waves of a problem first act as
// QUERY-block owners (P, dP, dS of 16 query rows; dQ block), leave dS and the dropped P as bf16 hi / lo tiles in LDS, and then as
// KEY-block owners: dV[16 keys] = P_drop^T . dO and dK[16 keys] = dS^T . Q over ALL query rows -- A = a transposing read down the
// tiles of every query block, B = the transposable image of dO / Q, full-depth v_mfma_f32_16x16x32_bf16, results stored once.
// No scalar operand loads, no LDS atomics, no accumulator round trip (the single-block kernel above keeps those for 16-row
// problems, where one wave owns everything).  One image region is reused for K (dQ), dO (dV) and Q (dK).
// LDS: [slots][image: 2 NDF planes x max(LKP, LQ32) x 32 B | bias-table gradient]  then  [4 waves][dS hi, dS lo, Pd hi, Pd lo tiles]
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 am_tileT_frag(const unsigned char* tiles, const int tile_stride, const int plane_off, const int pitch, const int jb,
                                                const int kq, const int NQB, const int lr, const int lq) {
  const int rowoff = plane_off + (4 * lq + (lr >> 2)) * pitch + (jb * 16 + (lr & 3) * 4) * 2;
  am_s16x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  if (2 * kq < NQB) a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) am_s16x4*)(tiles + (2 * kq) * tile_stride + rowoff));
  if (2 * kq + 1 < NQB) b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) am_s16x4*)(tiles + (2 * kq + 1) * tile_stride + rowoff));
  const am_s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}

template <int NKS, int NDF>
__global__ __launch_bounds__(256) void attn_mfma_bwd_shared_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                                   const float* __restrict__ table, const int64_t* __restrict__ rel_index,
                                                                   const float* __restrict__ dout, float* __restrict__ dq, float* __restrict__ dk,
                                                                   float* __restrict__ dv, float* __restrict__ dtable, const AmGeom gm, const int nqbr,
                                                                   const int causal, const float p, const uint64_t* seed_dev, const uint32_t site,
                                                                   const float dq_scale, const int p16, const int ntab) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char am_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  const int NQB = (gm.Lq + 15) >> 4, njf = (gm.Lk + 15) >> 4, nkj = (gm.Lk + 31) >> 5, LKP = nkj * 32;
  const int nkq = (gm.Lq + 31) >> 5, LQ32 = nkq * 32, IMR = LKP > LQ32 ? LKP : LQ32;
  const int slots = 4 / nqbr, slot = wave / nqbr, qb = wave - slot * nqbr;
  const int prob = blockIdx.x * slots + slot;
  const bool pvalid = prob < gm.groups * gm.nh;
  const int g = pvalid ? prob / gm.nh : 0, h = pvalid ? prob - g * gm.nh : 0;
  const bool active = pvalid && qb < NQB;
  const int img_bytes = 2 * NDF * IMR * 32, vimg_bytes = 2 * NDF * LKP * 32, tab_bytes = dtable ? ((ntab * 4 + 15) & ~15) : 0;
  const int slot_bytes = img_bytes + vimg_bytes + tab_bytes, pitch = LKP * 2 + 16, tile_stride = 4 * 16 * pitch;
  unsigned char* img = am_smem + slot * slot_bytes;
  unsigned char* vimg = img + img_bytes;                      // V rows for dP = dO V^T (second generation: fragments from LDS, see am_img_frag)
  float* stab = reinterpret_cast<float*>(vimg + vimg_bytes);
  unsigned char* tiles = am_smem + slots * slot_bytes + slot * nqbr * tile_stride;   // tiles of this problem's query blocks
  unsigned char* mytile = tiles + qb * tile_stride;                                  // [dS hi][dS lo][Pd hi][Pd lo]
  const int stid = tid - slot * nqbr * 64, snthr = nqbr * 64;

  if (pvalid) {
    am_stage_img(img, k, gm, g, h, IMR, NDF, stid, snthr);   // the image region has IMR rows in every phase
    am_stage_img(vimg, v, gm, g, h, LKP, NDF, stid, snthr);
  }
  for (int e = stid; e < tab_bytes / 4; e += snthr) stab[e] = 0.f;
  __syncthreads();   // K and V images complete
  float ds[4][4];
#pragma unroll
  for (int jf = 0; jf < 4; ++jf)
#pragma unroll
    for (int r = 0; r < 4; ++r) ds[jf][r] = 0.f;
  if (active) {
    float pr[4][4];
    am_scores<NKS, NDF>(gm, q, k, table, rel_index, g, h, qb, njf, causal, lr, lq, pr, img, IMR);
    bf16x8 gh[NKS], gl[NKS];
    {
      const int i = qb * 16 + lr;
      const float* row = dout + am_qrow(gm, g, min(i, gm.Lq - 1)) * gm.C + h * gm.hd;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) am_load_frag(row, 32 * ks + 8 * lq, gm.hd, i < gm.Lq, gh[ks], gl[ks]);
    }
    uint64_t seed = 0;
    if (p > 0.f) seed = *seed_dev;
    float delta[4] = {0.f, 0.f, 0.f, 0.f}, pd[4][4];
#pragma unroll
    for (int jf = 0; jf < 4; ++jf) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pd[jf][r] = 0.f;
      if (jf < njf) {
        const int j = jf * 16 + lr;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const bf16x8 vh = am_img_frag<NDF>(vimg, LKP, ks, 0, j, lq), vl = am_img_frag<NDF>(vimg, LKP, ks, 1, j, lq);
          acc = am_mfma3(gh[ks], gl[ks], vh, vl, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = qb * 16 + 4 * lq + r;
          float sc = 1.f;
          if (p > 0.f && i < gm.Lq && j < gm.Lk) sc = am_drop_scale(gm, seed, site, prob, i, j, p);
          pd[jf][r] = i < gm.Lq ? pr[jf][r] * sc : 0.f;
          ds[jf][r] = acc[r] * sc;
          delta[r] += pr[jf][r] * ds[jf][r];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) delta[r] = am_row16_sum(delta[r]);
#pragma unroll
    for (int jf = 0; jf < 4; ++jf) {
      const int j = jf * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ds[jf][r] = pr[jf][r] * (ds[jf][r] - delta[r]);
        if (jf * 16 < LKP) {
          const int il = 4 * lq + r;
          uint32_t hi, lo;
          vptr_split2(ds[jf][r], pd[jf][r], hi, lo);   // low halves: dS, high halves: dropped P
          *reinterpret_cast<uint16_t*>(mytile + il * pitch + j * 2) = (uint16_t)hi;
          *reinterpret_cast<uint16_t*>(mytile + 16 * pitch + il * pitch + j * 2) = (uint16_t)lo;
          *reinterpret_cast<uint16_t*>(mytile + 32 * pitch + il * pitch + j * 2) = (uint16_t)(hi >> 16);
          *reinterpret_cast<uint16_t*>(mytile + 48 * pitch + il * pitch + j * 2) = (uint16_t)(lo >> 16);
        }
      }
    }
  }
  __syncthreads();   // tiles of every query block
  if (active && dtable) {
#pragma unroll
    for (int jf = 0; jf < 4; ++jf) {
      const int j = jf * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = qb * 16 + 4 * lq + r;
        if (jf < njf && j < gm.Lk && i < gm.Lq) atomicAdd(&stab[rel_index[i * gm.Lk + j]], ds[jf][r]);
      }
    }
  }
  if (active) {   // dQ block = dS . K
    f32x4 qacc[NDF];
#pragma unroll
    for (int df = 0; df < NDF; ++df) qacc[df] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kj = 0; kj < 2; ++kj) {
      if (kj < nkj) {
        const bf16x8 sh = am_tile_frag(mytile, pitch, kj, lr, lq), sl = am_tile_frag(mytile + 16 * pitch, pitch, kj, lr, lq);
#pragma unroll
        for (int df = 0; df < NDF; ++df) {
          const bf16x8 kh = am_tr_frag(img, IMR, df, 0, kj, lr, lq), kl = am_tr_frag(img, IMR, df, 1, kj, lr, lq);
          qacc[df] = am_mfma3(kh, kl, sh, sl, qacc[df]);   // dQ^T block (operands swapped: see am_store_quad)
        }
      }
    }
    const int i = qb * 16 + lr;
    if (i < gm.Lq) {
      const int64_t rowe = (int64_t)(am_qrow(gm, g, i) * gm.C + h * gm.hd);
#pragma unroll
      for (int df = 0; df < NDF; ++df) am_store_quad(dq, rowe, df * 16 + 4 * lq, gm.hd, qacc[df], dq_scale, p16);
    }
  }
  // key-block ownership: dV (A = dropped P tiles, B = dO image) then dK (A = dS tiles, B = Q image)
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();   // everyone is done with the previous image
    if (pvalid) am_stage_img(img, pass == 0 ? dout : q, gm, g, h, IMR, NDF, stid, snthr, true);
    __syncthreads();
    float* dst = pass == 0 ? dv : dk;
    const int plane_off = pass == 0 ? 32 * pitch : 0;
    if (pvalid) {
      for (int jb = qb; jb < njf; jb += nqbr) {   // wave-uniform
        f32x4 acc[NDF];
#pragma unroll
        for (int df = 0; df < NDF; ++df) acc[df] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
          if (kq < nkq) {
            const bf16x8 ah = am_tileT_frag(tiles, tile_stride, plane_off, pitch, jb, kq, NQB, lr, lq);
            const bf16x8 al = am_tileT_frag(tiles, tile_stride, plane_off + 16 * pitch, pitch, jb, kq, NQB, lr, lq);
#pragma unroll
            for (int df = 0; df < NDF; ++df) {
              const bf16x8 bh = am_tr_frag(img, IMR, df, 0, kq, lr, lq), bl = am_tr_frag(img, IMR, df, 1, kq, lr, lq);
              acc[df] = am_mfma3(bh, bl, ah, al, acc[df]);   // transposed block: lane (lr, lq) = key row lr, channels 16 df + 4 lq ..
            }
          }
        }
        const int j = jb * 16 + lr;
        if (j < gm.Lk) {
          const int64_t rowe = (int64_t)(am_krow(gm, g, j) * gm.C + h * gm.hd);
#pragma unroll
          for (int df = 0; df < NDF; ++df) am_store_quad(dst, rowe, df * 16 + 4 * lq, gm.hd, acc[df], 1.f, p16);
        }
      }
    }
  }
  if (dtable && pvalid) {   // the table atomics precede two barriers
    for (int t = stid; t < ntab; t += snthr) {
      const float val = stab[t];
      if (val != 0.f) unsafeAtomicAdd(dtable + (int64_t)t * gm.nh + h, val);
    }
  }
}

static int am_enabled() {   // read per call: tests switch the mode inside one process
  const char* e = getenv("VPTR_ATTN_MFMA");
  return e ? atoi(e) : 1;
}

// true when the MFMA kernels cover the geometry (otherwise the fp32 vector kernels of attn.hip run)
bool vptr_attn_mfma_ok(int Lq, int Lk, int C, int nh, int causal, int64_t groups) {
  const int hd = C / nh;
  // VPTR_ATTN_MFMA: 0 = never, 1 (default) = problems with more than 16 rows (the 16-token fp32 vector kernels of attn.hip are
  // faster on 16 x 16 problems: K64 step 56.9 vs 61.0 ms), 2 = every covered geometry
  const int mode = am_enabled();
  if (mode == 0 || (mode == 1 && Lq <= 16 && Lk <= 16)) return false;
  // row offsets are 32-bit inside these kernels (am_derive): tensors of 2^31 or more elements take the fp32 vector kernels of attn.hip
  if (groups * (int64_t)(Lq > Lk ? Lq : Lk) * C >= ((int64_t)1 << 31)) return false;
  return Lq >= 1 && Lk >= 1 && Lq <= AM_MAXL && Lk <= AM_MAXL && hd % 2 == 0 && C % 2 == 0 && hd <= 96 && (!causal || Lq == Lk);
}

template <int NKS, int NDF>
static int am_launch_fwd(const float* q, const float* k, const float* v, const float* table, const int64_t* rel_index, float* o, const AmGeom& gm, int causal,
                         float p, const uint64_t* seed_dev, uint32_t site, int p16, hipStream_t st) {
  const int NQB = (gm.Lq + 15) / 16, nqbr = NQB == 1 ? 1 : (NQB == 2 ? 2 : 4), slots = 4 / nqbr;
  const int LKP = (gm.Lk + 31) / 32 * 32, pitch = LKP * 2 + 16;
  const size_t img_bytes = (size_t)2 * NDF * LKP * 32;
  const size_t lds = slots * 2 * img_bytes + ((size_t)nqbr * 2 * 16 * pitch <= img_bytes ? 0 : 4 * 2 * 16 * pitch);   // K and V images per slot (+ P tiles)
  const int nprob = gm.groups * gm.nh;
  auto kern = attn_mfma_fwd_kernel<NKS, NDF>;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kern<<<(nprob + slots - 1) / slots, 256, lds, st>>>(q, k, v, table, rel_index, o, gm, nqbr, causal, p, seed_dev, site, p16);
  return 0;
}

// fields the kernels derive their index arithmetic from; refuses tensors whose element offsets leave 31 bits
static int am_derive(const AmGeom& in, AmGeom& gm) {
  gm = in;
  gm.ws_shift = -1;
  if (gm.mode == 0 && gm.ws > 0 && (gm.ws & (gm.ws - 1)) == 0) {
    gm.ws_shift = 0;
    while ((1 << gm.ws_shift) < gm.ws) ++gm.ws_shift;
  }
  const int64_t rows_q = (int64_t)gm.groups * (gm.mode == 0 ? gm.Lq : gm.Tq), rows_k = (int64_t)gm.groups * (gm.mode == 0 ? gm.Lk : gm.Tk);
  const int64_t rows = rows_q > rows_k ? rows_q : rows_k;
  VPTR_CHECK(rows * gm.C < ((int64_t)1 << 31), "attn_mfma: tensors of 2^31 or more elements are not supported (%lld rows x %d)", (long long)rows, gm.C);
  gm.idx32 = ((int64_t)gm.groups * gm.nh * gm.Lq * gm.Lk < ((int64_t)1 << 32)) ? 1 : 0;
  return 0;
}

int vptr_attn_mfma_fwd(const float* q, const float* k, const float* v, const float* table, const int64_t* rel_index, float* o, const AmGeom& gm_in, int causal,
                       float p, const uint64_t* seed_dev, uint32_t site, int p16, hipStream_t st) {
  AmGeom gm;
  if (am_derive(gm_in, gm)) return -1;
  const int nks = (gm.hd + 31) / 32, ndf = (gm.hd + 15) / 16;
#define AM_CASE(NKS, NDF) if (nks == NKS && ndf == NDF) return am_launch_fwd<NKS, NDF>(q, k, v, table, rel_index, o, gm, causal, p, seed_dev, site, p16, st);
  AM_CASE(1, 1) AM_CASE(1, 2) AM_CASE(2, 3) AM_CASE(2, 4) AM_CASE(3, 5) AM_CASE(3, 6)
#undef AM_CASE
  vptr_set_error("attn_mfma: unsupported head dim %d", gm.hd);
  return -1;
}

template <int NKS, int NDF>
static int am_launch_bwd(const float* q, const float* k, const float* v, const float* table, const int64_t* rel_index, const float* dout, float* dq, float* dk,
                         float* dv, float* dtable, const AmGeom& gm, int causal, float p, const uint64_t* seed_dev, uint32_t site, float dq_scale, int p16,
                         hipStream_t st) {
  const int NQB = (gm.Lq + 15) / 16, nqbr = NQB == 1 ? 1 : (NQB == 2 ? 2 : 4), slots = 4 / nqbr;
  const int LKP = (gm.Lk + 31) / 32 * 32, pitch = LKP * 2 + 16;
  const int ntab = dtable ? (2 * gm.ws - 1) * (2 * gm.ws - 1) : 0;
  const size_t slot_bytes = (size_t)2 * NDF * LKP * 32 + (nqbr > 1 ? (size_t)LKP * NDF * 64 : 0) + (dtable ? ((ntab * 4 + 15) & ~15) : 0);
  const size_t lds = slots * slot_bytes + 4 * 2 * 16 * pitch;
  const int nprob = gm.groups * gm.nh;
  if (nqbr > 1) {   // several query blocks per problem: the tile-sharing kernel
    const int LQ32 = (gm.Lq + 31) / 32 * 32, IMR = LKP > LQ32 ? LKP : LQ32;
    const size_t sb = (size_t)2 * NDF * IMR * 32 + (size_t)2 * NDF * LKP * 32 + (dtable ? ((ntab * 4 + 15) & ~15) : 0);   // K / dO / Q image, V image, table gradient
    const size_t lds2 = slots * sb + 4 * 4 * 16 * pitch;
    auto kern2 = attn_mfma_bwd_shared_kernel<NKS, NDF>;
    if (lds2 > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    kern2<<<(nprob + slots - 1) / slots, 256, lds2, st>>>(q, k, v, table, rel_index, dout, dq, dk, dv, dtable, gm, nqbr, causal, p, seed_dev, site, dq_scale, p16,
                                                          ntab);
    return 0;
  }
  auto kern = attn_mfma_bwd_kernel<NKS, NDF>;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kern<<<(nprob + slots - 1) / slots, 256, lds, st>>>(q, k, v, table, rel_index, dout, dq, dk, dv, dtable, gm, nqbr, causal, p, seed_dev, site, dq_scale, p16, ntab);
  return 0;
}

int vptr_attn_mfma_bwd(const float* q, const float* k, const float* v, const float* table, const int64_t* rel_index, const float* dout, float* dq, float* dk,
                       float* dv, float* dtable, const AmGeom& gm_in, int causal, float p, const uint64_t* seed_dev, uint32_t site, float dq_scale, int p16,
                       hipStream_t st) {
  AmGeom gm;
  if (am_derive(gm_in, gm)) return -1;
  const int nks = (gm.hd + 31) / 32, ndf = (gm.hd + 15) / 16;
#define AM_CASE(NKS, NDF) if (nks == NKS && ndf == NDF) return am_launch_bwd<NKS, NDF>(q, k, v, table, rel_index, dout, dq, dk, dv, dtable, gm, causal, p, seed_dev, site, dq_scale, p16, st);
  AM_CASE(1, 1) AM_CASE(1, 2) AM_CASE(2, 3) AM_CASE(2, 4) AM_CASE(3, 5) AM_CASE(3, 6)
#undef AM_CASE
  vptr_set_error("attn_mfma: unsupported head dim %d", gm.hd);
  return -1;
}
