// MFMA GEMM with fused epilogue for the VPTR hot path (gfx950).
//
//   D[M,N] = epilogue( op(A)[M,K] * op(B)[K,N] ),  fp32 in HBM, bf16 (1 pass) or split-bf16 (3 passes) on the
//   matrix cores, fp32 accumulate.  The fp32 -> bf16 hi/lo split happens in the global->LDS staging path, so no
//   pre-converted copies of activations or weights exist in HBM (fp32 is as compact as hi+lo).
//
// Tiling: workgroup = 512 threads = 8 waves in a 4 (M) x 2 (N) grid, block tile 128 x (16*NFN) x 32.  Wave (wm, wn) owns
// rows [32 wm, 32 wm + 32) and column fragments wn*NFW .. wn*NFW + NFW - 1 (NFW = ceil(NFN/2)) of
// v_mfma_f32_16x16x32_bf16: 2 x NFW accumulators of 4 VGPRs.  NFN = 11 gives BN = 176, which divides every channel
// count of the model (528 = 3*176, 1056, 1584, 2112 = 12*176) with no tail waste in HBM traffic (the 12th fragment
// slot of the odd wave column is computed on padding and never stored).  Two waves per SIMD (and a register budget
// of <= 128 VGPRs, i.e. two co-resident workgroups when the grid is large enough) cover the LDS and HBM latencies
// that a single wave per SIMD exposes.
// LDS image: [row][k] bf16 with a 40-element (80 B) pitch -> ds_read_b128 fragment reads and ds_write_b64 staging
// writes are at worst 2-way conflicted for both operand orientations.
// Operand orientations: k-contiguous (nn.Linear forward), k-strided (dgrad's W, wgrad's dY and X; transposed in the
// staging path), and an implicit-GEMM gather of NHWC images for Conv2d / ConvTranspose2d.
// The K loop is straight-line code: every global load is unconditional (addresses clamped into the matrix), the K
// tail is zeroed by an AND mask on the A operand only, and partial staging iterations are executed redundantly by the
// otherwise idle threads.  (A guarded load, or a select/branch behind one, makes hipcc wait for every load
// individually; a dynamically indexed accumulator array is demoted to scratch memory.)
#include "common.h"

#define GBM 128
#define GBK 32
#define GLP 40
#define GNT 512

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ float4 mask4(const float4 v, const unsigned m) {
  float4 o;
  o.x = __uint_as_float(__float_as_uint(v.x) & m);
  o.y = __uint_as_float(__float_as_uint(v.y) & m);
  o.z = __uint_as_float(__float_as_uint(v.z) & m);
  o.w = __uint_as_float(__float_as_uint(v.w) & m);
  return o;
}

// two fp32 -> one dword of two bf16 (round-to-nearest-even): a single v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pk_bf16(const float a, const float b) {
  const f32x2 f = {a, b};
  const bf16x2 h = __builtin_convertvector(f, bf16x2);
  return *reinterpret_cast<const uint32_t*>(&h);
}
// split-bf16: x = hi + lo + O(2^-17 |x|); hi = bf16(x), lo = bf16(x - hi).  6 VALU per pair.
__device__ __forceinline__ void split2(const float a, const float b, uint32_t& hi, uint32_t& lo) {
  hi = pk_bf16(a, b);
  const float fa = __uint_as_float(hi << 16), fb = __uint_as_float(hi & 0xffff0000u);
  lo = pk_bf16(a - fa, b - fb);
}

template <int NPASS>
__device__ __forceinline__ void lds_put4(__bf16* s_hi, __bf16* s_lo, int row, int kc, const float4 v) {
  if constexpr (NPASS == 3) {
    uint2 hi, lo;
    split2(v.x, v.y, hi.x, lo.x);
    split2(v.z, v.w, hi.y, lo.y);
    *reinterpret_cast<uint2*>(&s_hi[row * GLP + kc]) = hi;
    *reinterpret_cast<uint2*>(&s_lo[row * GLP + kc]) = lo;
  } else {
    *reinterpret_cast<uint2*>(&s_hi[row * GLP + kc]) = make_uint2(pk_bf16(v.x, v.y), pk_bf16(v.z, v.w));
  }
}

// thread -> slot of staging iteration i for a tile of NSLOT slots: full iterations use i*512 + tid; in a partial last
// iteration (V valid slots) the surplus threads repeat slots of the first ones (same data, same LDS address).
template <int NSLOT>
__device__ __forceinline__ int slot_of(const int i, const int tid) {
  constexpr int NIT = (NSLOT + GNT - 1) / GNT;
  constexpr int V = NSLOT - GNT * (NIT - 1);
  if (i < NIT - 1 || V == GNT) return GNT * i + tid;
  if constexpr ((V & (V - 1)) == 0) return GNT * i + (tid & (V - 1));
  static_assert((V & (V - 1)) == 0 || 2 * V >= GNT, "unsupported partial staging iteration");
  return GNT * i + (tid >= V ? tid - V : tid);
}

// k-contiguous operand: ROWS x 32 fp32 tile, float4 slots, slot s -> row = s>>3, kc = (s&7)*4.
// Masking policy: rows/columns beyond the matrix edge are never stored by the epilogue, so their (clamped, finite)
// garbage needs no zeroing; only the K tail must contribute zero, and zeroing it in ONE operand (A: KMASK) is enough.
template <int ROWS, int LROWS, bool KMASK>
struct StageKC {
  static constexpr int LDS_ROWS = LROWS;
  static constexpr int NSLOT = ROWS * 8;
  static constexpr int NIT = (NSLOT + GNT - 1) / GNT;
  float4 r[NIT];
  unsigned kmask;
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int row0, int nrows, int k0, int kend, int tid) {
    const int kc = (tid & 7) << 2;  // the same in every iteration (slot bases and folds are multiples of 8)
    const int kk = min(k0 + kc, kend - 4);
    kmask = (k0 + kc < kend) ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int row = slot_of<NSLOT>(i, tid) >> 3;
      r[i] = *reinterpret_cast<const float4*>(P + (int64_t)min(row0 + row, nrows - 1) * ld + kk);
    }
  }
  template <int NPASS>
  __device__ __forceinline__ void store(__bf16* s_hi, __bf16* s_lo, int tid) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int s = slot_of<NSLOT>(i, tid);
      lds_put4<NPASS>(s_hi, s_lo, s >> 3, (s & 7) << 2, KMASK ? mask4(r[i], kmask) : r[i]);
    }
  }
};

// k-strided operand stored [K, ld] with the output dim contiguous: 4(k) x 4(out) micro-tiles, slot s -> kb = s&7
// (k block of 4), ob = s>>3 (out block of 4); transposed in registers, 4 ds_write_b64 per slot.
template <int LROWS, bool KMASK>
struct StageKS {
  static constexpr int LDS_ROWS = LROWS;
  static constexpr int NSLOT = LROWS * 2;
  static constexpr int NIT = (NSLOT + GNT - 1) / GNT;
  float4 r[NIT][4];
  int nvalid;  // number of k rows of this thread's 4-row k block that lie inside the K range (<= 0: none, >= 4: all)
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int row0, int nrows, int k0, int kend, int tid) {
    const int kb = tid & 7;
    nvalid = kend - (k0 + kb * 4);
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int ob = slot_of<NSLOT>(i, tid) >> 3;
      const int mm = min(row0 + ob * 4, nrows - 4);  // nrows % 4 == 0 (checked on the host)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        r[i][j] = *reinterpret_cast<const float4*>(P + (int64_t)min(k0 + kb * 4 + j, kend - 1) * ld + mm);
    }
  }
  template <int NPASS>
  __device__ __forceinline__ void store(__bf16* s_hi, __bf16* s_lo, int tid) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int s = slot_of<NSLOT>(i, tid);
      const int kb = s & 7, ob = s >> 3;
      const unsigned m0 = nvalid > 0 ? ~0u : 0u, m1 = nvalid > 1 ? ~0u : 0u, m2 = nvalid > 2 ? ~0u : 0u, m3 = nvalid > 3 ? ~0u : 0u;
      const float4 v0 = KMASK ? mask4(r[i][0], m0) : r[i][0], v1 = KMASK ? mask4(r[i][1], m1) : r[i][1];
      const float4 v2 = KMASK ? mask4(r[i][2], m2) : r[i][2], v3 = KMASK ? mask4(r[i][3], m3) : r[i][3];
      lds_put4<NPASS>(s_hi, s_lo, ob * 4 + 0, kb * 4, make_float4(v0.x, v1.x, v2.x, v3.x));
      lds_put4<NPASS>(s_hi, s_lo, ob * 4 + 1, kb * 4, make_float4(v0.y, v1.y, v2.y, v3.y));
      lds_put4<NPASS>(s_hi, s_lo, ob * 4 + 2, kb * 4, make_float4(v0.z, v1.z, v2.z, v3.z));
      lds_put4<NPASS>(s_hi, s_lo, ob * 4 + 3, kb * 4, make_float4(v0.w, v1.w, v2.w, v3.w));
    }
  }
};

// implicit-GEMM gather of an NHWC image (Conv2d / gather-form ConvTranspose2d); slot map of StageKC<128>
struct StageConv {
  static constexpr int LDS_ROWS = GBM;
  static constexpr int NIT = GBM * 8 / GNT;  // 2
  float4 r[NIT];
  unsigned okbits;
  // source coordinate, or -1 when the tap contributes zero; selects only (no divergent branches)
  static __device__ __forceinline__ int map_coord(int o, int kk, int I, const vptr_gemm_desc& p) {
    if (p.conv_transposed) {  // kernel-uniform
      const int num = o + p.conv_pad - kk;
      const int q = (p.conv_stride == 2) ? (num >> 1) : (num / p.conv_stride);
      const bool ok = (num >= 0) & (q * p.conv_stride == num) & (q < I);
      return ok ? q : -1;
    }
    const int c = o * p.conv_stride - p.conv_pad + kk;
    const bool inside = (c >= 0) & (c < I);
    const int refl = c < 0 ? -c : 2 * I - 2 - c;
    const int repl = c < 0 ? 0 : I - 1;
    const int outv = p.conv_pad_mode == VPTR_PAD_ZERO ? -1 : (p.conv_pad_mode == VPTR_PAD_REFLECT ? refl : repl);
    return inside ? c : outv;
  }
  __device__ __forceinline__ void load(const vptr_gemm_desc& p, int row0, int k0, int kend, int tid) {
    const int kc = (tid & 7) << 2;
    const int gk = min(k0 + kc, kend - 4);
    const bool okk = (k0 + kc) < kend;
    const int tap = gk / p.conv_Cin, ci = gk - tap * p.conv_Cin;
    const int ky = tap / p.conv_KW, kx = tap - ky * p.conv_KW;
    const int per = p.conv_OH * p.conv_OW;
    okbits = 0;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int gm = row0 + ((tid + GNT * i) >> 3);
      const int gmc = min(gm, p.M - 1);
      const int f = gmc / per, rem = gmc - f * per;
      const int oy = rem / p.conv_OW, ox = rem - oy * p.conv_OW;
      const int iy = map_coord(oy, ky, p.conv_IH, p);
      const int ix = map_coord(ox, kx, p.conv_IW, p);
      const bool ok = okk & (iy >= 0) & (ix >= 0);
      const int64_t off = ((int64_t)(f * p.conv_IH + max(iy, 0)) * p.conv_IW + max(ix, 0)) * p.conv_Cin + ci;
      r[i] = *reinterpret_cast<const float4*>(p.A + off);
      okbits |= (ok ? 1u : 0u) << i;
    }
  }
  template <int NPASS>
  __device__ __forceinline__ void store(__bf16* s_hi, __bf16* s_lo, int tid) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int s = tid + GNT * i;
      lds_put4<NPASS>(s_hi, s_lo, s >> 3, (s & 7) << 2, mask4(r[i], 0u - ((okbits >> i) & 1u)));
    }
  }
};

template <int NFN, int NPASS, int AMODE, int BMODE>
__global__ __launch_bounds__(GNT, 4) void vptr_gemm_kernel(const vptr_gemm_desc p, const int k_chunk) {
  constexpr int BN = 16 * NFN;
  constexpr int NFW = (NFN + 1) / 2;     // column fragments per wave
  constexpr int BROWS = 2 * NFW * 16;    // LDS rows of the B image (>= BN; the surplus rows feed never-stored fragments)
  constexpr int NPL = (NPASS == 3) ? 2 : 1;
  using StA = typename std::conditional<AMODE == VPTR_A_KCONTIG, StageKC<GBM, GBM, true>,
                                        typename std::conditional<AMODE == VPTR_A_KSTRIDED, StageKS<GBM, true>, StageConv>::type>::type;
  using StB = typename std::conditional<BMODE == VPTR_B_KCONTIG, StageKC<BN, BROWS, false>, StageKS<BROWS, false>>::type;
  __shared__ __attribute__((aligned(16))) __bf16 sA[NPL][GBM * GLP];
  __shared__ __attribute__((aligned(16))) __bf16 sB[NPL][BROWS * GLP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
  const int m0 = tm * GBM, n0 = tn * BN;
  const int kbeg = blockIdx.z * k_chunk;
  const int kend = min(p.K, kbeg + k_chunk);
  const int nkt = (kend - kbeg + GBK - 1) / GBK;

  f32x4 acc[2][NFW];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NFW; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  StA stA;
  StB stB;
  // K-step j+1 is fetched into registers while step j is multiplied out of LDS (single LDS image, two barriers per step);
  // the other resident waves of the SIMD cover what is left of the HBM / LDS latencies.
  if constexpr (AMODE == VPTR_A_CONV) stA.load(p, m0, kbeg, kend, tid);
  else stA.load(p.A, p.lda, m0, p.M, kbeg, kend, tid);
  stB.load(p.B, p.ldb, n0, p.N, kbeg, kend, tid);

  for (int kt = 0; kt < nkt; ++kt) {
    stA.template store<NPASS>(sA[0], sA[NPL - 1], tid);
    stB.template store<NPASS>(sB[0], sB[NPL - 1], tid);
    __syncthreads();
    {  // unconditional prefetch of the next step (clamped + masked beyond the K range)
      const int k1 = kbeg + (kt + 1) * GBK;
      if constexpr (AMODE == VPTR_A_CONV) stA.load(p, m0, k1, kend, tid);
      else stA.load(p.A, p.lda, m0, p.M, k1, kend, tid);
      stB.load(p.B, p.ldb, n0, p.N, k1, kend, tid);
    }
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int off = (wm * 32 + mi * 16 + lr) * GLP + lq * 8;
      ah[mi] = *reinterpret_cast<const bf16x8*>(&sA[0][off]);
      if constexpr (NPASS == 3) al[mi] = *reinterpret_cast<const bf16x8*>(&sA[NPL - 1][off]);
    }
#pragma unroll
    for (int ni = 0; ni < NFW; ++ni) {
      const int off = ((wn * NFW + ni) * 16 + lr) * GLP + lq * 8;
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

  // ---- epilogue: C/D fragment layout of v_mfma_f32_16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg.
  // Every acc index is a compile-time constant (fully unrolled, no `continue`).
  const bool first_split = (blockIdx.z == 0);
  const bool use_atomic = p.atomic || gridDim.z > 1;
  const bool plain = !p.colscale && !p.Dpre && p.act == VPTR_ACT_NONE && !p.rowscale && p.dropout_p == 0.f && !p.act_after &&
                     p.alpha == 1.f;
  const int row_base = m0 + wm * 32 + lq * 4;
  if (plain) {  // kernel-uniform fast path: bias (+ residual), store or atomic accumulate
#pragma unroll
    for (int ni = 0; ni < NFW; ++ni) {
      const int nf = wn * NFW + ni;
      const int col = n0 + nf * 16 + lr;
      const bool colok = (nf < NFN) & (col < p.N);
      const float bs = (p.bias && first_split && colok) ? p.bias[col] : 0.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row_base + mi * 16 + r;
          if (colok && row < p.M) {
            float v = acc[mi][ni][r] + bs;
            if (p.residual && first_split) v += p.residual[(int64_t)row * p.ldr + col];
            float* dst = p.D + (int64_t)row * p.ldd + col;
            if (use_atomic) unsafeAtomicAdd(dst, v);
            else *dst = v;
          }
        }
      }
    }
  } else {
    uint64_t seed = 0;
    if (p.dropout_p > 0.f) seed = *p.seed_dev;
#pragma unroll
    for (int ni = 0; ni < NFW; ++ni) {
      const int nf = wn * NFW + ni;
      const int col = n0 + nf * 16 + lr;
      const bool colok = (nf < NFN) & (col < p.N);
      const float bs = (p.bias && first_split && colok) ? p.bias[col] : 0.f;
      const float cs = (p.colscale && colok) ? p.colscale[col] : 1.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row_base + mi * 16 + r;
          if (colok && row < p.M) {
            float v = (acc[mi][ni][r] * cs + bs) * p.alpha;
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
  }
}

template <int NFN, int NPASS>
static int launch_modes(const vptr_gemm_desc& d, dim3 grid, int k_chunk, hipStream_t st) {
  if (d.a_mode == VPTR_A_KCONTIG && d.b_mode == VPTR_B_KCONTIG)
    vptr_gemm_kernel<NFN, NPASS, VPTR_A_KCONTIG, VPTR_B_KCONTIG><<<grid, GNT, 0, st>>>(d, k_chunk);
  else if (d.a_mode == VPTR_A_KCONTIG && d.b_mode == VPTR_B_KSTRIDED)
    vptr_gemm_kernel<NFN, NPASS, VPTR_A_KCONTIG, VPTR_B_KSTRIDED><<<grid, GNT, 0, st>>>(d, k_chunk);
  else if (d.a_mode == VPTR_A_KSTRIDED && d.b_mode == VPTR_B_KSTRIDED)
    vptr_gemm_kernel<NFN, NPASS, VPTR_A_KSTRIDED, VPTR_B_KSTRIDED><<<grid, GNT, 0, st>>>(d, k_chunk);
  else if (d.a_mode == VPTR_A_KSTRIDED && d.b_mode == VPTR_B_KCONTIG)
    vptr_gemm_kernel<NFN, NPASS, VPTR_A_KSTRIDED, VPTR_B_KCONTIG><<<grid, GNT, 0, st>>>(d, k_chunk);
  else if (d.a_mode == VPTR_A_CONV && d.b_mode == VPTR_B_KCONTIG)
    vptr_gemm_kernel<NFN, NPASS, VPTR_A_CONV, VPTR_B_KCONTIG><<<grid, GNT, 0, st>>>(d, k_chunk);
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
  if (d.a_mode != VPTR_A_KSTRIDED || d.b_mode != VPTR_B_KSTRIDED)
    VPTR_CHECK(d.K % 4 == 0, "vptr_gemm: K must be a multiple of 4 for k-contiguous operands (got %d)", d.K);
  if (d.a_mode == VPTR_A_KCONTIG) VPTR_CHECK(d.lda % 4 == 0, "vptr_gemm: k-contiguous A needs lda%%4==0");
  if (d.a_mode == VPTR_A_KSTRIDED) VPTR_CHECK(d.lda % 4 == 0 && d.M % 4 == 0, "vptr_gemm: k-strided A needs lda%%4==0 and M%%4==0");
  if (d.b_mode == VPTR_B_KCONTIG) VPTR_CHECK(d.ldb % 4 == 0, "vptr_gemm: k-contiguous B needs ldb%%4==0");
  if (d.b_mode == VPTR_B_KSTRIDED) VPTR_CHECK(d.ldb % 4 == 0 && d.N % 4 == 0, "vptr_gemm: k-strided B needs ldb%%4==0 and N%%4==0");
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
