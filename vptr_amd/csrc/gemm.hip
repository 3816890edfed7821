// MFMA GEMM with fused epilogue for the VPTR hot path (gfx950).
//
//   D[M,N] = epilogue( op(A)[M,K] * op(B)[K,N] ),  fp32 in HBM, bf16 (1 pass) or split-bf16 (3 passes) on the
//   matrix cores, fp32 accumulate.  The fp32 -> bf16 hi/lo split happens in the global->LDS staging path, so no
//   pre-converted copies of activations or weights exist in HBM (fp32 is as compact as hi+lo).
//
// Tiling: workgroup = 256 threads = 4 waves, block tile 128 x (16*NFN) x 32; wave w owns rows [32w, 32w+32) and all
// NFN column fragments (v_mfma_f32_16x16x32_bf16, 2 x NFN accumulators of 4 VGPRs).  NFN = 11 gives BN = 176, which
// divides every channel count of the model (528 = 3*176, 1056, 1584, 2112 = 12*176) with no tail waste.
// LDS image: [row][k] bf16 with a 40-element (80 B) pitch -> ds_read_b128 fragment reads and ds_write_b64 staging
// writes are at worst 2-way conflicted for both operand orientations.
// Operand orientations: k-contiguous (nn.Linear forward), k-strided (dgrad's W, wgrad's dY and X; transposed in the
// staging path), and an implicit-GEMM gather of NHWC images for Conv2d / ConvTranspose2d.
#include "common.h"

#define GBM 128
#define GBK 32
#define GLP 40

struct ConvRow {  // per staged A row of the implicit-GEMM gather
  int64_t base;   // frame offset in floats, -1 if the row is out of range
  int oy, ox;
};

__device__ __forceinline__ uint2 pack_hi4(const float4 v) {
  bf16x4 h;
  h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
  return *reinterpret_cast<uint2*>(&h);
}
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  bf16x4 h, l;
  h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
  l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
  l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
  hi = *reinterpret_cast<uint2*>(&h);
  lo = *reinterpret_cast<uint2*>(&l);
}

template <int NPASS>
__device__ __forceinline__ void lds_put4(__bf16* s_hi, __bf16* s_lo, int row, int kc, const float4 v) {
  if constexpr (NPASS == 3) {
    uint2 hi, lo;
    split4(v, hi, lo);
    *reinterpret_cast<uint2*>(&s_hi[row * GLP + kc]) = hi;
    *reinterpret_cast<uint2*>(&s_lo[row * GLP + kc]) = lo;
  } else {
    *reinterpret_cast<uint2*>(&s_hi[row * GLP + kc]) = pack_hi4(v);
  }
}

// k-contiguous operand: ROWS x 32 fp32 tile, thread t loads float4 slots s = t + 256 i, row = s>>3, kc = (s&7)*4
template <int ROWS>
struct StageKC {
  static constexpr int NIT = (ROWS * 8 + 255) / 256;
  float4 r[NIT];
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int row0, int nrows, int k0, int kend, int tid) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int s = tid + 256 * i;
      const int row = s >> 3, kc = (s & 7) << 2;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < ROWS && row0 + row < nrows && k0 + kc < kend)
        v = *reinterpret_cast<const float4*>(P + (int64_t)(row0 + row) * ld + k0 + kc);
      r[i] = v;
    }
  }
  template <int NPASS>
  __device__ __forceinline__ void store(__bf16* s_hi, __bf16* s_lo, int tid) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int s = tid + 256 * i;
      const int row = s >> 3, kc = (s & 7) << 2;
      if (row < ROWS) lds_put4<NPASS>(s_hi, s_lo, row, kc, r[i]);
    }
  }
};

// k-strided operand stored [K, ld] with the output dim contiguous: 4(k) x 4(out) micro-tiles, slot s -> kb = s&7
// (k block of 4), ob = s>>3 (out block of 4); transposed in registers, 4 ds_write_b64 per slot.
template <int ROWS>
struct StageKS {
  static constexpr int NSLOT = ROWS * 2;
  static constexpr int NIT = (NSLOT + 255) / 256;
  float4 r[NIT][4];
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int row0, int nrows, int k0, int kend, int tid) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int s = tid + 256 * i;
      const int kb = s & 7, ob = s >> 3;
      const int gm = row0 + ob * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gk = k0 + kb * 4 + j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < NSLOT && gk < kend) {
          const float* src = P + (int64_t)gk * ld + gm;
          if (gm + 3 < nrows) {
            v = *reinterpret_cast<const float4*>(src);
          } else {
            if (gm < nrows) v.x = src[0];
            if (gm + 1 < nrows) v.y = src[1];
            if (gm + 2 < nrows) v.z = src[2];
          }
        }
        r[i][j] = v;
      }
    }
  }
  template <int NPASS>
  __device__ __forceinline__ void store(__bf16* s_hi, __bf16* s_lo, int tid) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int s = tid + 256 * i;
      const int kb = s & 7, ob = s >> 3;
      if (s < NSLOT) {
        lds_put4<NPASS>(s_hi, s_lo, ob * 4 + 0, kb * 4, make_float4(r[i][0].x, r[i][1].x, r[i][2].x, r[i][3].x));
        lds_put4<NPASS>(s_hi, s_lo, ob * 4 + 1, kb * 4, make_float4(r[i][0].y, r[i][1].y, r[i][2].y, r[i][3].y));
        lds_put4<NPASS>(s_hi, s_lo, ob * 4 + 2, kb * 4, make_float4(r[i][0].z, r[i][1].z, r[i][2].z, r[i][3].z));
        lds_put4<NPASS>(s_hi, s_lo, ob * 4 + 3, kb * 4, make_float4(r[i][0].w, r[i][1].w, r[i][2].w, r[i][3].w));
      }
    }
  }
};

// implicit-GEMM gather of an NHWC image (Conv2d / gather-form ConvTranspose2d); same slot map as StageKC
struct StageConv {
  static constexpr int NIT = 4;  // GBM * 8 / 256
  float4 r[NIT];
  ConvRow cr[NIT];
  __device__ __forceinline__ void init(const vptr_gemm_desc& p, int row0, int tid) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int row = (tid + 256 * i) >> 3;
      const int gm = row0 + row;
      if (gm < p.M) {
        const int per = p.conv_OH * p.conv_OW;
        const int f = gm / per, rem = gm - f * per;
        cr[i].oy = rem / p.conv_OW;
        cr[i].ox = rem - cr[i].oy * p.conv_OW;
        cr[i].base = (int64_t)f * p.conv_IH * p.conv_IW * p.conv_Cin;
      } else {
        cr[i].base = -1; cr[i].oy = 0; cr[i].ox = 0;
      }
    }
  }
  __device__ __forceinline__ int map_coord(int o, int kk, int I, const vptr_gemm_desc& p) const {
    if (p.conv_transposed) {
      const int num = o + p.conv_pad - kk;
      if (num < 0) return -1;
      const int q = num / p.conv_stride;
      if (q * p.conv_stride != num || q >= I) return -1;
      return q;
    }
    int c = o * p.conv_stride - p.conv_pad + kk;
    if (c < 0 || c >= I) {
      if (p.conv_pad_mode == VPTR_PAD_ZERO) return -1;
      if (p.conv_pad_mode == VPTR_PAD_REFLECT) c = c < 0 ? -c : 2 * I - 2 - c;
      else c = c < 0 ? 0 : I - 1;
    }
    return c;
  }
  __device__ __forceinline__ void load(const vptr_gemm_desc& p, int k0, int kend, int tid) {
    const int kc = (tid & 7) << 2;
    const int gk = k0 + kc;
    const int tap = gk / p.conv_Cin, ci = gk - tap * p.conv_Cin;
    const int ky = tap / p.conv_KW, kx = tap - ky * p.conv_KW;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cr[i].base >= 0 && gk < kend) {
        const int iy = map_coord(cr[i].oy, ky, p.conv_IH, p);
        const int ix = map_coord(cr[i].ox, kx, p.conv_IW, p);
        if (iy >= 0 && ix >= 0)
          v = *reinterpret_cast<const float4*>(p.A + cr[i].base + ((int64_t)iy * p.conv_IW + ix) * p.conv_Cin + ci);
      }
      r[i] = v;
    }
  }
  template <int NPASS>
  __device__ __forceinline__ void store(__bf16* s_hi, __bf16* s_lo, int tid) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int s = tid + 256 * i;
      lds_put4<NPASS>(s_hi, s_lo, s >> 3, (s & 7) << 2, r[i]);
    }
  }
};

template <int NFN, int NPASS, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void vptr_gemm_kernel(const vptr_gemm_desc p, const int k_chunk) {
  constexpr int BN = 16 * NFN;
  constexpr int NPL = (NPASS == 3) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) __bf16 sA[NPL][GBM * GLP];
  __shared__ __attribute__((aligned(16))) __bf16 sB[NPL][BN * GLP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
  const int m0 = tm * GBM, n0 = tn * BN;
  const int kbeg = blockIdx.z * k_chunk;
  const int kend = min(p.K, kbeg + k_chunk);
  const int nkt = (kend - kbeg + GBK - 1) / GBK;

  f32x4 acc[2][NFN];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NFN; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typename std::conditional<AMODE == VPTR_A_KCONTIG, StageKC<GBM>,
                            typename std::conditional<AMODE == VPTR_A_KSTRIDED, StageKS<GBM>, StageConv>::type>::type stA;
  typename std::conditional<BMODE == VPTR_B_KCONTIG, StageKC<BN>, StageKS<BN>>::type stB;

  if constexpr (AMODE == VPTR_A_CONV) stA.init(p, m0, tid);

  auto loadA = [&](int k0) {
    if constexpr (AMODE == VPTR_A_CONV) stA.load(p, k0, kend, tid);
    else stA.load(p.A, p.lda, m0, p.M, k0, kend, tid);
  };
  auto loadB = [&](int k0) { stB.load(p.B, p.ldb, n0, p.N, k0, kend, tid); };

  if (nkt > 0) { loadA(kbeg); loadB(kbeg); }

  for (int kt = 0; kt < nkt; ++kt) {
    stA.template store<NPASS>(sA[0], sA[NPL - 1], tid);
    stB.template store<NPASS>(sB[0], sB[NPL - 1], tid);
    __syncthreads();
    if (kt + 1 < nkt) { loadA(kbeg + (kt + 1) * GBK); loadB(kbeg + (kt + 1) * GBK); }

    bf16x8 ah[2], al[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int off = (wave * 32 + mi * 16 + lr) * GLP + lq * 8;
      ah[mi] = *reinterpret_cast<const bf16x8*>(&sA[0][off]);
      if constexpr (NPASS == 3) al[mi] = *reinterpret_cast<const bf16x8*>(&sA[NPL - 1][off]);
    }
#pragma unroll
    for (int ni = 0; ni < NFN; ++ni) {
      const int off = (ni * 16 + lr) * GLP + lq * 8;
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&sB[0][off]);
      bf16x8 bl;
      if constexpr (NPASS == 3) bl = *reinterpret_cast<const bf16x8*>(&sB[NPL - 1][off]);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        if constexpr (NPASS == 3) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl, acc[mi][ni], 0, 0, 0);
        }
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh, acc[mi][ni], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: C/D fragment layout of v_mfma_f32_16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg
  const bool first_split = (blockIdx.z == 0);
  const bool use_atomic = p.atomic || gridDim.z > 1;
  uint64_t seed = 0;
  if (p.dropout_p > 0.f) seed = *p.seed_dev;
#pragma unroll
  for (int ni = 0; ni < NFN; ++ni) {
    const int col = n0 + ni * 16 + lr;
    if (col >= p.N) continue;
    const float cs = p.colscale ? p.colscale[col] : 1.f;
    const float bs = (p.bias && first_split) ? p.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wave * 32 + mi * 16 + lq * 4 + r;
        if (row >= p.M) continue;
        float v = acc[mi][ni][r];
        v = (v * cs + bs) * p.alpha;
        if (p.Dpre) p.Dpre[(int64_t)row * p.ldd + col] = v;
        v = vptr_act(v, p.act);
        if (p.rowscale) v *= p.rowscale[(row / p.rs_div) % p.rs_mod];
        if (p.dropout_p > 0.f) v *= vptr_drop_scale(seed, p.site, (uint64_t)row * (uint64_t)p.N + col, p.dropout_p);
        if (p.residual && first_split) v += p.residual[(int64_t)row * p.ldr + col];
        if (p.act_after) v = v > 0.f ? v : 0.f;
        float* dst = p.D + (int64_t)row * p.ldd + col;
        if (use_atomic) unsafeAtomicAdd(dst, v);
        else *dst = v;
      }
    }
  }
}

template <int NFN, int NPASS>
static int launch_modes(const vptr_gemm_desc& d, dim3 grid, int k_chunk, hipStream_t st) {
  if (d.a_mode == VPTR_A_KCONTIG && d.b_mode == VPTR_B_KCONTIG)
    vptr_gemm_kernel<NFN, NPASS, VPTR_A_KCONTIG, VPTR_B_KCONTIG><<<grid, 256, 0, st>>>(d, k_chunk);
  else if (d.a_mode == VPTR_A_KCONTIG && d.b_mode == VPTR_B_KSTRIDED)
    vptr_gemm_kernel<NFN, NPASS, VPTR_A_KCONTIG, VPTR_B_KSTRIDED><<<grid, 256, 0, st>>>(d, k_chunk);
  else if (d.a_mode == VPTR_A_KSTRIDED && d.b_mode == VPTR_B_KSTRIDED)
    vptr_gemm_kernel<NFN, NPASS, VPTR_A_KSTRIDED, VPTR_B_KSTRIDED><<<grid, 256, 0, st>>>(d, k_chunk);
  else if (d.a_mode == VPTR_A_KSTRIDED && d.b_mode == VPTR_B_KCONTIG)
    vptr_gemm_kernel<NFN, NPASS, VPTR_A_KSTRIDED, VPTR_B_KCONTIG><<<grid, 256, 0, st>>>(d, k_chunk);
  else if (d.a_mode == VPTR_A_CONV && d.b_mode == VPTR_B_KCONTIG)
    vptr_gemm_kernel<NFN, NPASS, VPTR_A_CONV, VPTR_B_KCONTIG><<<grid, 256, 0, st>>>(d, k_chunk);
  else if (d.a_mode == VPTR_A_CONV && d.b_mode == VPTR_B_KSTRIDED)
    vptr_gemm_kernel<NFN, NPASS, VPTR_A_CONV, VPTR_B_KSTRIDED><<<grid, 256, 0, st>>>(d, k_chunk);
  else {
    vptr_set_error("vptr_gemm: unsupported operand modes a=%d b=%d", d.a_mode, d.b_mode);
    return -1;
  }
  return 0;
}

template <int NFN>
static int launch_prec(const vptr_gemm_desc& d, dim3 grid, int k_chunk, hipStream_t st) {
  if (d.precision == 3) return launch_modes<NFN, 3>(d, grid, k_chunk, st);
  return launch_modes<NFN, 1>(d, grid, k_chunk, st);
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int vptr_gemm(const vptr_gemm_desc* desc, vptr_stream_t stream) {
  VPTR_CHECK(desc != nullptr, "vptr_gemm: null descriptor");
  vptr_gemm_desc d = *desc;
  VPTR_CHECK(d.M > 0 && d.N > 0 && d.K > 0, "vptr_gemm: empty problem M=%d N=%d K=%d", d.M, d.N, d.K);
  VPTR_CHECK(d.A && d.B && d.D, "vptr_gemm: null operand");
  VPTR_CHECK(d.precision == 1 || d.precision == 3, "vptr_gemm: precision must be 1 or 3 (got %d)", d.precision);
  VPTR_CHECK(al16(d.A) && al16(d.B), "vptr_gemm: A and B must be 16-byte aligned");
  if (d.a_mode == VPTR_A_KCONTIG) VPTR_CHECK(d.lda % 4 == 0 && d.K % 4 == 0, "vptr_gemm: k-contiguous A needs lda%%4==0 and K%%4==0");
  if (d.a_mode == VPTR_A_KSTRIDED) VPTR_CHECK(d.lda % 4 == 0, "vptr_gemm: k-strided A needs lda%%4==0");
  if (d.b_mode == VPTR_B_KCONTIG) VPTR_CHECK(d.ldb % 4 == 0 && d.K % 4 == 0, "vptr_gemm: k-contiguous B needs ldb%%4==0 and K%%4==0");
  if (d.b_mode == VPTR_B_KSTRIDED) VPTR_CHECK(d.ldb % 4 == 0, "vptr_gemm: k-strided B needs ldb%%4==0");
  if (d.a_mode == VPTR_A_CONV) {
    VPTR_CHECK(d.conv_Cin % 4 == 0, "vptr_gemm(conv): Cin must be a multiple of 4 (got %d)", d.conv_Cin);
    VPTR_CHECK(d.K == d.conv_KH * d.conv_KW * d.conv_Cin, "vptr_gemm(conv): K != KH*KW*Cin");
    VPTR_CHECK(d.conv_stride >= 1 && d.conv_OH > 0 && d.conv_OW > 0, "vptr_gemm(conv): bad geometry");
  }
  if (d.split_k < 1) d.split_k = 1;
  if (d.split_k > 1) VPTR_CHECK(d.act == VPTR_ACT_NONE && !d.act_after && d.dropout_p == 0.f && !d.rowscale && !d.Dpre,
                                "vptr_gemm: split_k > 1 supports only linear epilogues");
  if (d.dropout_p > 0.f) VPTR_CHECK(d.seed_dev != nullptr && d.dropout_p < 1.f, "vptr_gemm: dropout needs seed_dev and p < 1");
  if (d.rowscale) VPTR_CHECK(d.rs_div >= 1 && d.rs_mod >= 1, "vptr_gemm: rowscale needs rs_div, rs_mod >= 1");
  if (d.alpha == 0.f) d.alpha = 1.f;

  // k range per split, multiple of the K tile
  int k_chunk = ((d.K + d.split_k - 1) / d.split_k + GBK - 1) / GBK * GBK;
  const int splits = (d.K + k_chunk - 1) / k_chunk;

  // column-fragment count: exact 176-wide tiles when N is a multiple of 176, otherwise least padding
  int nfn;
  if (d.N % 176 == 0) nfn = 11;
  else if (d.N <= 64) nfn = 4;
  else if (d.N <= 128) nfn = 8;
  else {
    const int w11 = (d.N + 175) / 176 * 176, w8 = (d.N + 127) / 128 * 128, w4 = (d.N + 63) / 64 * 64;
    nfn = 11;
    int best = w11;
    if (w8 < best) { best = w8; nfn = 8; }
    if (w4 < best) { best = w4; nfn = 4; }
  }
  const int bn = 16 * nfn;
  const int tiles_m = (d.M + GBM - 1) / GBM, tiles_n = (d.N + bn - 1) / bn;
  dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)splits);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc;
  if (nfn == 11) rc = launch_prec<11>(d, grid, k_chunk, st);
  else if (nfn == 8) rc = launch_prec<8>(d, grid, k_chunk, st);
  else rc = launch_prec<4>(d, grid, k_chunk, st);
  if (rc) return rc;
  VPTR_LAUNCH_CHECK();
  return 0;
}
