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
// LDS image: [row][32 k] bf16, 64-byte rows, XOR-swizzled (lds_off): the 16-byte k chunk kq of row r sits at chunk
// kq ^ ((-(r >> 2)) & 3) of physical row r ^ ((r >> 2) & 1).  With the lane groups gfx950 uses for ds_read_b128
// ({0-3,12-15,20-27}, ...) and ds_write_b64 (16 contiguous lanes) this is conflict-free for the fragment reads and
// for the staging writes of both operand orientations (an 80-byte padded pitch was 2-way conflicted on every read).
// Operand orientations: k-contiguous (nn.Linear forward), k-strided (dgrad's W, wgrad's dY and X; transposed in the
// staging path), and an implicit-GEMM gather of NHWC images for Conv2d / ConvTranspose2d.
// The K loop is straight-line code: every global load is unconditional (addresses clamped into the matrix), the K
// tail is zeroed by an AND mask on the A operand only, and partial staging iterations are executed redundantly by the
// otherwise idle threads.  (A guarded load, or a select/branch behind one, makes hipcc wait for every load
// individually; a dynamically indexed accumulator array is demoted to scratch memory.)
#include "gemm_shared.h"

#ifdef VPTR_GEMM_TIMING  // tools/gemm_probe.hip: per-phase shader-clock stamps of wave 0 of every workgroup
__device__ long long* vptr_gemm_timing_buf = nullptr;
#define TS_DECL long long ts_acc[6] = {0, 0, 0, 0, 0, 0}; long long ts_t = clock64(); const long long ts_begin = ts_t;
#define TS(i) { const long long ts_n = clock64(); ts_acc[i] += ts_n - ts_t; ts_t = ts_n; }
#define TS_FLUSH { if (threadIdx.x == 0 && vptr_gemm_timing_buf) { for (int q = 0; q < 6; ++q) vptr_gemm_timing_buf[blockIdx.x * 8 + q] = ts_acc[q]; \
                   vptr_gemm_timing_buf[blockIdx.x * 8 + 6] = ts_begin; vptr_gemm_timing_buf[blockIdx.x * 8 + 7] = clock64(); } }
#else
#define TS_DECL
#define TS(i)
#define TS_FLUSH
#endif

// bf16 element offset of k chunk k4 (4 elements = 8 bytes, k4 = 0..7) of tile row `row` in the swizzled LDS image
__device__ __forceinline__ int lds_off(const int row, const int k4) {
  const int j = row >> 2;
  return ((row ^ (j & 1)) << 5) + ((((k4 >> 1) ^ (0 - j)) & 3) << 3) + ((k4 & 1) << 2);
}

template <int NPASS>
__device__ __forceinline__ void lds_put4(__bf16* s_hi, __bf16* s_lo, int row, int kc, const float4 v) {
  if constexpr (NPASS == 3) {
    uint2 hi, lo;
    split2(v.x, v.y, hi.x, lo.x);
    split2(v.z, v.w, hi.y, lo.y);
    const int o = lds_off(row, kc >> 2);
    *reinterpret_cast<uint2*>(&s_hi[o]) = hi;
    *reinterpret_cast<uint2*>(&s_lo[o]) = lo;
  } else {
    *reinterpret_cast<uint2*>(&s_hi[lds_off(row, kc >> 2)]) = make_uint2(pk_bf16(v.x, v.y), pk_bf16(v.z, v.w));
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
  static constexpr int NUNIT = NIT;  // store units (one ds_write_b64 per plane each), for interleaving with MFMAs
  template <int NPASS>
  __device__ __forceinline__ void store_unit(__bf16* s_hi, __bf16* s_lo, int tid, const int i) {
    const int s = slot_of<NSLOT>(i, tid);
    lds_put4<NPASS>(s_hi, s_lo, s >> 3, (s & 7) << 2, KMASK ? mask4(r[i], kmask) : r[i]);
  }
  template <int NPASS>
  __device__ __forceinline__ void store(__bf16* s_hi, __bf16* s_lo, int tid) {
#pragma unroll
    for (int i = 0; i < NUNIT; ++i) store_unit<NPASS>(s_hi, s_lo, tid, i);
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
  static constexpr int NUNIT = NIT * 4;
  static __device__ __forceinline__ float comp(const float4 v, const int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }
  template <int NPASS>
  __device__ __forceinline__ void store_unit(__bf16* s_hi, __bf16* s_lo, int tid, const int u) {
    const int i = u >> 2, j = u & 3;  // compile-time constants after unrolling
    const int s = slot_of<NSLOT>(i, tid);
    const int kb = s & 7, ob = s >> 3;
    float4 v = make_float4(comp(r[i][0], j), comp(r[i][1], j), comp(r[i][2], j), comp(r[i][3], j));
    if constexpr (KMASK) {
      v.x = __uint_as_float(__float_as_uint(v.x) & (nvalid > 0 ? ~0u : 0u));
      v.y = __uint_as_float(__float_as_uint(v.y) & (nvalid > 1 ? ~0u : 0u));
      v.z = __uint_as_float(__float_as_uint(v.z) & (nvalid > 2 ? ~0u : 0u));
      v.w = __uint_as_float(__float_as_uint(v.w) & (nvalid > 3 ? ~0u : 0u));
    }
    lds_put4<NPASS>(s_hi, s_lo, ob * 4 + j, kb * 4, v);
  }
  template <int NPASS>
  __device__ __forceinline__ void store(__bf16* s_hi, __bf16* s_lo, int tid) {
#pragma unroll
    for (int u = 0; u < NUNIT; ++u) store_unit<NPASS>(s_hi, s_lo, tid, u);
  }
};

// k-strided A operand of exactly GBM = 128 rows: 2(k) x 4(out) micro-tiles -> 512 slots, one per thread, no surplus
// (the 4 x 4 tiling above has only 256 slots for 128 rows, so every thread would load and convert a duplicate: +21 % on
// the weight-gradient loop).  kb2 = k pair (0..15), ob = out block (0..31); a wave covers 8 k pairs x 8 out blocks =
// 128-byte row segments; 4 ds_write_b32 per plane, conflict-free with the swizzle above.
template <bool KMASK>
struct StageKS2 {
  static constexpr int LDS_ROWS = GBM;
  float4 r[2];
  int nvalid;
  float rsum[4] = {0.f, 0.f, 0.f, 0.f};  // running sum over k of this thread's 4 output rows (vptr_gemm_desc::a_rowsum)
  static __device__ __forceinline__ int kb2_of(const int tid) { return (tid & 7) | (((tid >> 6) & 1) << 3); }
  static __device__ __forceinline__ int ob_of(const int tid) { return ((tid >> 3) & 7) | ((tid >> 7) << 3); }
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int row0, int nrows, int k0, int kend, int tid) {
    const int kb2 = kb2_of(tid), ob = ob_of(tid);
    nvalid = kend - (k0 + kb2 * 2);
    const int mm = min(row0 + ob * 4, nrows - 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) r[j] = *reinterpret_cast<const float4*>(P + (int64_t)min(k0 + kb2 * 2 + j, kend - 1) * ld + mm);
  }
  static constexpr int NUNIT = 4;
  static __device__ __forceinline__ float comp(const float4 v, const int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }
  template <int NPASS>
  __device__ __forceinline__ void store_unit(__bf16* s_hi, __bf16* s_lo, int tid, const int j) {
    const int kb2 = kb2_of(tid), ob = ob_of(tid);
    float a = comp(r[0], j), b = comp(r[1], j);
    if constexpr (KMASK) {
      a = __uint_as_float(__float_as_uint(a) & (nvalid > 0 ? ~0u : 0u));
      b = __uint_as_float(__float_as_uint(b) & (nvalid > 1 ? ~0u : 0u));
    }
    rsum[j] += a + b;
    const int o = lds_off(ob * 4 + j, kb2 >> 1) + ((kb2 & 1) << 1);
    if constexpr (NPASS == 3) {
      uint32_t hi, lo;
      split2(a, b, hi, lo);
      *reinterpret_cast<uint32_t*>(&s_hi[o]) = hi;
      *reinterpret_cast<uint32_t*>(&s_lo[o]) = lo;
    } else {
      *reinterpret_cast<uint32_t*>(&s_hi[o]) = pk_bf16(a, b);
    }
  }
  template <int NPASS>
  __device__ __forceinline__ void store(__bf16* s_hi, __bf16* s_lo, int tid) {
#pragma unroll
    for (int u = 0; u < NUNIT; ++u) store_unit<NPASS>(s_hi, s_lo, tid, u);
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
  static constexpr int NUNIT = NIT;
  template <int NPASS>
  __device__ __forceinline__ void store_unit(__bf16* s_hi, __bf16* s_lo, int tid, const int i) {
    const int s = tid + GNT * i;
    lds_put4<NPASS>(s_hi, s_lo, s >> 3, (s & 7) << 2, mask4(r[i], 0u - ((okbits >> i) & 1u)));
  }
  template <int NPASS>
  __device__ __forceinline__ void store(__bf16* s_hi, __bf16* s_lo, int tid) {
#pragma unroll
    for (int i = 0; i < NUNIT; ++i) store_unit<NPASS>(s_hi, s_lo, tid, i);
  }
};

// K-segment cursor (desc.ksegs > 1: D = sum_s op(A_s) op(B_s), every segment K long): K-step kt of the virtual K range
// -> operand pointers and the k offset inside the segment.  Steps beyond the last segment (prefetch) land at k >= K of the
// last one, i.e. fully masked.  Two compares instead of a division: ksegs <= 3.
struct KSeg {
  int spk, last;  // K-steps per segment; index of the last segment
  __device__ __forceinline__ KSeg(const vptr_gemm_desc& p) {
    last = p.ksegs > 1 ? p.ksegs - 1 : 0;
    spk = last ? (p.K + GBK - 1) / GBK : (1 << 26);
  }
  __device__ __forceinline__ int steps(const int kbeg, const int kend) const {
    return last ? (last + 1) * spk : (kend - kbeg + GBK - 1) / GBK;
  }
  // segment index and k offset of K-step kt
  __device__ __forceinline__ int seg(const int kbeg, const int kt, int& k0) const {
    const int s = min((int)(kt >= spk) + (int)(kt >= 2 * spk), last);
    k0 = kbeg + (kt - s * spk) * GBK;
    return s;
  }
};
// operand of segment s as base + element offset (offsets are plain integers: selecting between POINTERS read from the
// descriptor made hipcc keep the descriptor in scratch)
#define KSEG_OFFSETS(p)                                                                  \
  const int64_t ksA1 = (p).ksegs > 1 ? (p).A_x1 - (p).A : 0, ksA2 = (p).ksegs > 2 ? (p).A_x2 - (p).A : 0; \
  const int64_t ksB1 = (p).ksegs > 1 ? (p).B_x1 - (p).B : 0, ksB2 = (p).ksegs > 2 ? (p).B_x2 - (p).B : 0;
#define KSEG_PTRS(mb, s, Ak, Bk)                                           \
  const float* Ak = (mb).A + ((s) == 0 ? (int64_t)0 : ((s) == 1 ? ksA1 : ksA2)); \
  const float* Bk = (mb).B + ((s) == 0 ? (int64_t)0 : ((s) == 1 ? ksB1 : ksB2));

template <int NFN, int NPASS, int AMODE, int BMODE>
__global__ __launch_bounds__(GNT, 4) void vptr_gemm_kernel(const vptr_gemm_desc p, const int k_chunk, const int epi_rows) {
  constexpr int BN = 16 * NFN;
  constexpr int NFW = (NFN + 1) / 2;     // column fragments per wave
  constexpr int BROWS = 2 * NFW * 16;    // LDS rows of the B image (>= BN; the surplus rows feed never-stored fragments)
  constexpr int NPL = (NPASS == 3) ? 2 : 1;
  using StA = typename std::conditional<AMODE == VPTR_A_KCONTIG, StageKC<GBM, GBM, true>,
                                        typename std::conditional<AMODE == VPTR_A_KSTRIDED, StageKS2<true>, StageConv>::type>::type;
  using StB = typename std::conditional<BMODE == VPTR_B_KCONTIG, StageKC<BN, BROWS, false>, StageKS<BROWS, false>>::type;
  constexpr int LOOP_BYTES = NPL * (GBM + BROWS) * GLP * (int)sizeof(__bf16);
  constexpr int LDS_BYTES = LOOP_BYTES > epi_half_lds_bytes<NFN>() ? LOOP_BYTES : epi_half_lds_bytes<NFN>();
  __shared__ __attribute__((aligned(16))) unsigned char sraw[LDS_BYTES];   // [A hi, A lo, B hi, B lo] / epilogue half tile
  __bf16* const sA0 = reinterpret_cast<__bf16*>(sraw);
  __bf16* const sA1 = sA0 + (NPL - 1) * GBM * GLP;
  __bf16* const sB0 = sA0 + NPL * GBM * GLP;
  __bf16* const sB1 = sB0 + (NPL - 1) * BROWS * GLP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lq = lane >> 4;
  // XCD-aware tile order: workgroup b runs on XCD b % 8, and every XCD has its own L2.  XCD x is given a contiguous
  // range of the (split, tile_m, tile_n) order, so the A row panel of a tile_m is fetched by one L2 instead of all eight.
  const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
  const int logical = xcd * xq + min(xcd, xr) + (blockIdx.x >> 3);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles = tiles_n * ((p.M + GBM - 1) / GBM);
  const int grp = logical / tiles, tile = logical - grp * tiles;
  const bool batched = p.batch > 1;                 // the grid's outer index is the batch member or the K split
  const int split = batched ? 0 : grp;
  const Member mb = member_of(p, batched ? grp : 0);
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const int m0 = tm * GBM, n0 = tn * BN;
  const int kbeg = split * k_chunk;
  const int kend = min(p.K, kbeg + k_chunk);
  const KSeg ks(p);
  KSEG_OFFSETS(p)
  const int nkt = ks.steps(kbeg, kend);

  f32x4 acc[2][NFW];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NFW; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  StA stA;
  StB stB;
  TS_DECL
  // K-step j+1 is fetched into registers while step j is multiplied out of LDS (single LDS image, two barriers per step);
  // the other resident waves of the SIMD cover what is left of the HBM / LDS latencies.
  auto load_step = [&](const int kt) {
    int k0;
    const int sg = ks.seg(kbeg, kt, k0);
    KSEG_PTRS(mb, sg, Ak, Bk)
    if constexpr (AMODE == VPTR_A_CONV) stA.load(p, m0, k0, kend, tid);
    else stA.load(Ak, p.lda, m0, p.M, k0, kend, tid);
    stB.load(Bk, p.ldb, n0, p.N, k0, kend, tid);
  };
  load_step(0);

  TS(0)
  for (int kt = 0; kt < nkt; ++kt) {
    stA.template store<NPASS>(sA0, sA1, tid);
    stB.template store<NPASS>(sB0, sB1, tid);
    TS(1)
    __syncthreads();
    TS(2)
    load_step(kt + 1);  // unconditional prefetch of the next step (clamped + masked beyond the K range)
    TS(3)
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int off = lds_off(wm * 32 + mi * 16 + lr, lq * 2);
      ah[mi] = *reinterpret_cast<const bf16x8*>(&sA0[off]);
      if constexpr (NPASS == 3) al[mi] = *reinterpret_cast<const bf16x8*>(&sA1[off]);
    }
#pragma unroll
    for (int ni = 0; ni < NFW; ++ni) {
      const int off = lds_off((wn * NFW + ni) * 16 + lr, lq * 2);
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&sB0[off]);
      bf16x8 bl;
      if constexpr (NPASS == 3) bl = *reinterpret_cast<const bf16x8*>(&sB1[off]);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        if constexpr (NPASS == 3) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl, acc[mi][ni], 0, 0, 0);
        }
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh, acc[mi][ni], 0, 0, 0);
      }
    }
    TS(4)
    __syncthreads();
    TS(2)
  }

  const bool use_atomic = p.atomic || (!batched && nblk > tiles);
  // (176-wide tiles only: with 64 / 128-wide tiles -- the auto-encoder's convs, M up to 655 360 -- the LDS round trip and its
  // barriers cost more than the short rows gain: stage-1 step 32.8 -> 35.3 ms; and no atomic accumulation: float atomics
  // stay scalar, so the round trip buys nothing)
  if (NFN == 11 && epi_rows && !use_atomic && epi_vec_ok(p) && p.d_row_w == 0) gemm_epilogue_rows_halves<NFN>(p, mb, acc, reinterpret_cast<float*>(sraw), m0, n0, wm, wn, lr, lq, tid, split == 0, use_atomic);
  else gemm_epilogue_serial<NFN>(p, mb, acc, m0, n0, wm, wn, lr, lq, split == 0, use_atomic);
  TS(5)
  TS_FLUSH
}

// ---- software-pipelined main loop ("P" variant) -------------------------------------------------------------------------
// The phase probe (tools/gemm_probe.hip) showed the single-image loop above spending ~25 % of a workgroup's cycles in the
// convert+store phase, ~25 % at the two barriers and ~20 % issuing MFMAs: all 8 waves are in the same phase, so the VALU
// split and the matrix cores never overlap unless a second workgroup shares the CU.  Here the LDS image is double
// buffered (80 KB), global loads run two K-steps ahead in two register sets, and the fp32 -> bf16 hi/lo conversion of step
// j+1 is interleaved, unit by unit, between the MFMA groups of step j: one barrier per step, one workgroup per CU.
template <int NFN, int NPASS, int AMODE, int BMODE>
__device__ __forceinline__ void gemm_tile_p(const vptr_gemm_desc& p, const Member& mb, __bf16* smem, const int m0, const int n0, const int kbeg,
                                            const int kend, const bool first_split, const bool use_atomic, const bool epi_rows = true) {
  constexpr int BN = 16 * NFN;
  constexpr int NFW = (NFN + 1) / 2;
  constexpr int BROWS = 2 * NFW * 16;
  constexpr int NPL = (NPASS == 3) ? 2 : 1;
  using StA = typename std::conditional<AMODE == VPTR_A_KCONTIG, StageKC<GBM, GBM, true>,
                                        typename std::conditional<AMODE == VPTR_A_KSTRIDED, StageKS2<true>, StageConv>::type>::type;
  using StB = typename std::conditional<BMODE == VPTR_B_KCONTIG, StageKC<BN, BROWS, false>, StageKS<BROWS, false>>::type;
  constexpr int A_EL = GBM * GLP, B_EL = BROWS * GLP, BUF_EL = NPL * (A_EL + B_EL);  // smem: [2 buffers][A hi, A lo, B hi, B lo]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const KSeg ks(p);
  KSEG_OFFSETS(p)
  const int nkt = ks.steps(kbeg, kend);

  f32x4 acc[2][NFW];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NFW; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  StA stA0, stA1;
  StB stB0, stB1;
  TS_DECL
  // VPTR_EXP_NOLOAD / VPTR_EXP_NOCVT: elimination experiments of tools/gemm_probe.hip (results in DESIGN.md section 4)
  auto loadAB = [&](StA& sa, StB& sb, const int kt) {  // operands of K-step kt into a register set
#ifdef VPTR_EXP_NOLOAD
    if (kt > 2) return;
#endif
    int k0;
    const int sg = ks.seg(kbeg, kt, k0);
    KSEG_PTRS(mb, sg, Ak, Bk)
    if constexpr (AMODE == VPTR_A_CONV) sa.load(p, m0, k0, kend, tid);
    else sa.load(Ak, p.lda, m0, p.M, k0, kend, tid);
    sb.load(Bk, p.ldb, n0, p.N, k0, kend, tid);
  };
  // fragment read offsets (bf16 elements inside one plane)
  int offA[2], offB[NFW];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) offA[mi] = lds_off(wm * 32 + mi * 16 + lr, lq * 2);
#pragma unroll
  for (int ni = 0; ni < NFW; ++ni) offB[ni] = lds_off((wn * NFW + ni) * 16 + lr, lq * 2);

  // one K-step: multiply out of buffer CUR while converting register set `cv` (step kt+1) into the other buffer
  auto step = [&](const int cur, StA& cvA, StB& cvB) {
    __bf16* cA = smem + cur * BUF_EL;
    __bf16* cB = cA + NPL * A_EL;
    __bf16* nA = smem + (cur ^ 1) * BUF_EL;
    __bf16* nB = nA + NPL * A_EL;
    constexpr int UA = StA::NUNIT, UB = StB::NUNIT, UT = UA + UB;
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      ah[mi] = *reinterpret_cast<const bf16x8*>(&cA[offA[mi]]);
      if constexpr (NPASS == 3) al[mi] = *reinterpret_cast<const bf16x8*>(&cA[A_EL + offA[mi]]);
    }
#pragma unroll
    for (int ni = 0; ni < NFW; ++ni) {
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&cB[offB[ni]]);
      bf16x8 bl;
      if constexpr (NPASS == 3) bl = *reinterpret_cast<const bf16x8*>(&cB[B_EL + offB[ni]]);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        if constexpr (NPASS == 3) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl, acc[mi][ni], 0, 0, 0);
        }
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh, acc[mi][ni], 0, 0, 0);
      }
      // this group's share of the conversion work for the next step
#ifndef VPTR_EXP_NOCVT
#pragma unroll
      for (int u = (ni * UT) / NFW; u < ((ni + 1) * UT) / NFW; ++u) {
        if (u < UA) cvA.template store_unit<NPASS>(nA, nA + (NPL - 1) * A_EL, tid, u);
#ifndef VPTR_EXP_NOCVT_B
        else cvB.template store_unit<NPASS>(nB, nB + (NPL - 1) * B_EL, tid, u - UA);
#endif
      }
#endif
    }
  };

  loadAB(stA0, stB0, 0);
  loadAB(stA1, stB1, 1);
  TS(0)
  stA0.template store<NPASS>(smem, smem + (NPL - 1) * A_EL, tid);
  stB0.template store<NPASS>(smem + NPL * A_EL, smem + NPL * A_EL + (NPL - 1) * B_EL, tid);
  TS(1)
  __syncthreads();
  TS(2)
  for (int kt = 0; kt < nkt; kt += 2) {
    // even step: buffer 0 holds step kt, set 1 holds step kt+1, set 0 is free -> fetch step kt+2 into it
    loadAB(stA0, stB0, kt + 2);
    __builtin_amdgcn_sched_barrier(0);  // keep the loads up here: hipcc otherwise sinks them to the end of the step
    TS(3)
    step(0, stA1, stB1);
    TS(4)
    __syncthreads();
    TS(2)
    if (kt + 1 < nkt) {  // workgroup-uniform
      loadAB(stA1, stB1, kt + 3);
      __builtin_amdgcn_sched_barrier(0);
      TS(3)
      step(1, stA0, stB0);
      TS(4)
      __syncthreads();
      TS(2)
    }
  }
  if (epi_rows && !use_atomic && epi_vec_ok(p) && p.d_row_w == 0) gemm_epilogue_rows<NFN>(p, mb, acc, reinterpret_cast<float*>(smem), m0, n0, wm, wn, lr, lq, tid, first_split, use_atomic);
  else gemm_epilogue<NFN, 1>(p, mb, acc, m0, n0, wm, wn, lr, lq, first_split, use_atomic);
  if constexpr (AMODE == VPTR_A_KSTRIDED) {
    if (p.a_rowsum && n0 == 0) {  // workgroup-uniform: column tile 0 owns the row sums of its A panel
      // thread (kb2, ob) summed k = kb2*2 + {0,1} (mod 32) of rows ob*4 + j: fold the 8 kb2 lanes, then the wave pair
      float* sred = reinterpret_cast<float*>(smem);  // the K loop's last barrier has passed: LDS is free
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = stA0.rsum[j] + stA1.rsum[j];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        if ((lane & 7) == 0) sred[(wave * 8 + (lane >> 3)) * 4 + j] = v;
      }
      __syncthreads();
      if (tid < GBM) {
        const int ob = tid >> 2, j = tid & 3;                  // ob = (wave pair q) * 8 + (lane >> 3)
        const int q = ob >> 3, lo8 = ob & 7;
        const float v = sred[((2 * q) * 8 + lo8) * 4 + j] + sred[((2 * q + 1) * 8 + lo8) * 4 + j];
        // rows of a clamped out block (beyond M - 4) hold duplicates of other rows: only in-range rows are written
        if (m0 + tid < p.M && m0 + ob * 4 <= p.M - 4) unsafeAtomicAdd(p.a_rowsum + m0 + tid, v * p.alpha);
      }
    }
  }
  TS(5)
  TS_FLUSH
}

template <int NFN, int NPASS, int AMODE, int BMODE>
__global__ __launch_bounds__(GNT, 2) void vptr_gemm_kernel_p(const vptr_gemm_desc p, const int k_chunk, const int epi_rows) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  const int logical = xcd_logical_block();
  const int tiles_n = (p.N + 16 * NFN - 1) / (16 * NFN);
  const int tiles = tiles_n * ((p.M + GBM - 1) / GBM);
  const int grp = logical / tiles, tile = logical - grp * tiles;
  const bool batched = p.batch > 1;
  const int split = batched ? 0 : grp;
  const Member mb = member_of(p, batched ? grp : 0);
  const int kbeg = split * k_chunk;
  gemm_tile_p<NFN, NPASS, AMODE, BMODE>(p, mb, smem, (tile / tiles_n) * GBM, (tile % tiles_n) * 16 * NFN, kbeg, min(p.K, kbeg + k_chunk),
                                        split == 0, p.atomic || (!batched && (int)gridDim.x > tiles), epi_rows != 0);
}

// Grouped launch: `count` independent problems (same operand modes / precision / NFN class) in one grid, no split-K.
// tile_start[g] = first logical tile of problem g (prefix sums, tile_start[count] = grid size).  Used for the weight
// gradients of a whole backward pass (vptr_gemm_grouped): K = all tokens, so every tile runs a long K loop and writes its
// output once, instead of ~30 K-splits each paying a prologue and a 90 KB atomic epilogue.
template <int NFN, int NPASS, int AMODE, int BMODE>
__global__ __launch_bounds__(GNT, 2) void vptr_gemm_grouped_kernel(const vptr_gemm_desc* __restrict__ descs,
                                                                   const int* __restrict__ tile_start, const int count) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  const int logical = xcd_logical_block();
  int lo = 0, hi = count - 1;  // last g with tile_start[g] <= logical (workgroup-uniform scalar search)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_start[mid] <= logical) lo = mid;
    else hi = mid - 1;
  }
  const vptr_gemm_desc p = descs[lo];
  const int tile = logical - tile_start[lo];
  const int tiles_n = (p.N + 16 * NFN - 1) / (16 * NFN);
  gemm_tile_p<NFN, NPASS, AMODE, BMODE>(p, member_of(p, 0), smem, (tile / tiles_n) * GBM, (tile % tiles_n) * 16 * NFN, 0, p.K, true,
                                        p.atomic != 0);
}


// Main-loop choice.  The pipelined loop keeps one workgroup per CU busy (80 KB LDS, 150-190 VGPRs); the single-image loop
// fits two workgroups per CU, whose phases interleave on their own.  Measured (tools/gemm_bench.py, M = 10240): with >= 2
// workgroups per CU in the grid the single-image loop wins (N = 2112: 228 vs 179 TFLOP/s); with ~1 per CU the pipelined
// one does (N = 528, K = 2112: 237 vs 186).  2 = choose by grid size (default); 0 / 1 force one (tools/gemm_probe.hip).
static int g_gemm_variant = 2;
// VPTR_GEMM_EPI_ROWS: bit 0 = row-major epilogue in the pipelined kernels, bit 1 = in the single-image kernel (default 3;
// 0 = fragment-layout epilogues everywhere; tools/ab_epi.sh A/B runs)
static int epi_rows_flag() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VPTR_GEMM_EPI_ROWS");
    v = e ? atoi(e) : 3;
  }
  return v;
}
static int v4_min_tiles() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VPTR_GEMM_V4_MIN_TILES");
    v = e ? atoi(e) : 384;
  }
  return v;
}

template <int NFN, int NPASS, int AM, int BM>
static int launch_one(const vptr_gemm_desc& d, dim3 grid, int k_chunk, hipStream_t st) {
  const bool pipelined = g_gemm_variant == 1 || (g_gemm_variant == 2 && ((int)grid.x < v4_min_tiles() || d.a_rowsum != nullptr));
  if (pipelined) {
    constexpr int NFW = (NFN + 1) / 2, BROWS = 2 * NFW * 16, NPL = (NPASS == 3) ? 2 : 1;
    constexpr int LOOP_BYTES = 2 * NPL * (GBM + BROWS) * GLP * (int)sizeof(__bf16);
    constexpr int LDS_BYTES = LOOP_BYTES > epi_lds_bytes<NFN>() ? LOOP_BYTES : epi_lds_bytes<NFN>();
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_kernel_p<NFN, NPASS, AM, BM>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
        vptr_set_error("vptr_gemm: cannot reserve %d bytes of LDS", LDS_BYTES);
        return -1;
      }
      attr_set = true;
    }
    vptr_gemm_kernel_p<NFN, NPASS, AM, BM><<<grid, GNT, LDS_BYTES, st>>>(d, k_chunk, epi_rows_flag() & 1);
  } else {
    vptr_gemm_kernel<NFN, NPASS, AM, BM><<<grid, GNT, 0, st>>>(d, k_chunk, epi_rows_flag() & 2);
  }
  return 0;
}

template <int NFN, int NPASS>
static int launch_modes(const vptr_gemm_desc& d, dim3 grid, int k_chunk, hipStream_t st) {
  if (d.a_mode == VPTR_A_KCONTIG && d.b_mode == VPTR_B_KCONTIG)
    return launch_one<NFN, NPASS, VPTR_A_KCONTIG, VPTR_B_KCONTIG>(d, grid, k_chunk, st);
  if (d.a_mode == VPTR_A_KCONTIG && d.b_mode == VPTR_B_KSTRIDED)
    return launch_one<NFN, NPASS, VPTR_A_KCONTIG, VPTR_B_KSTRIDED>(d, grid, k_chunk, st);
  if (d.a_mode == VPTR_A_KSTRIDED && d.b_mode == VPTR_B_KSTRIDED)
    return launch_one<NFN, NPASS, VPTR_A_KSTRIDED, VPTR_B_KSTRIDED>(d, grid, k_chunk, st);
  if (d.a_mode == VPTR_A_KSTRIDED && d.b_mode == VPTR_B_KCONTIG)
    return launch_one<NFN, NPASS, VPTR_A_KSTRIDED, VPTR_B_KCONTIG>(d, grid, k_chunk, st);
  if (d.a_mode == VPTR_A_CONV && d.b_mode == VPTR_B_KCONTIG)
    return launch_one<NFN, NPASS, VPTR_A_CONV, VPTR_B_KCONTIG>(d, grid, k_chunk, st);
  vptr_set_error("vptr_gemm: unsupported operand modes a=%d b=%d", d.a_mode, d.b_mode);
  return -1;
}

template <int NFN>
static int launch_prec(const vptr_gemm_desc& d, dim3 grid, int k_chunk, hipStream_t st) {
  if (d.precision == 3) return launch_modes<NFN, 3>(d, grid, k_chunk, st);
  return launch_modes<NFN, 1>(d, grid, k_chunk, st);
}

// column-fragment count: exact 176-wide tiles when N is a multiple of 176, otherwise least padding
static int nfn_for(const int N) {
  if (N % 176 == 0) return 11;
  if (N <= 64) return 4;
  if (N <= 128) return 8;
  const int w11 = (N + 175) / 176 * 176, w8 = (N + 127) / 128 * 128, w4 = (N + 63) / 64 * 64;
  int nfn = 11, best = w11;
  if (w8 < best) { best = w8; nfn = 8; }
  if (w4 < best) { best = w4; nfn = 4; }
  return nfn;
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- convert-once path (first slice: the frozen encoder's 3x3 convolutions) ---------------------------------------------
// Operands arrive as interleaved bf16 hi / lo planes, X[row][c / 32][hi: 32 | lo: 32] (one 128-byte line per row and 32
// channels; channels padded to a multiple of 32 with zeros, and -- for activations -- one extra all-zero row at the end that
// zero-padded taps point at).  They are staged with global_load_lds_dwordx4: every lane fetches the 16-byte chunk that
// belongs at its linear LDS position (rows of 128 bytes, chunk c of row r at c ^ ((r >> 1) & 7): conflict-free fragment
// reads), so the main loop has no VALU split and no ds_write.  Two 40 KB stages and ~110 VGPRs: two workgroups per CU.
// tools/gemm_planes_probe.hip is the stand-alone study of this loop (1.4-1.45x the register-staged kernels).
constexpr int PL_STAGE = (GBM + 192) * 128;          // A 16 KB + B 24 KB (192 B rows: 176 used)
constexpr int PL_UPW = PL_STAGE / 1024 / 8;          // 1 KB DMA pieces per wave and stage: 5 (2 of A, 3 of B)

__global__ void split_planes_kernel(const float* __restrict__ x, __bf16* __restrict__ out, int64_t rows, int C, int CB) {
  // thread = (row, 32-channel block, group of 4 channels); row == rows is the all-zero row
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (rows + 1) * CB * 8) return;
  const int j = (int)(i & 7);
  const int64_t rb = i >> 3;
  const int64_t r = rb / CB;
  const int cb = (int)(rb - r * CB), c = cb * 32 + j * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (r < rows) {
    if (c + 3 < C) {
      const float4 t = *reinterpret_cast<const float4*>(x + r * C + c);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
      for (int e = 0; e < 4; ++e) if (c + e < C) v[e] = x[r * C + c + e];
    }
  }
  uint32_t hi[2], lo[2];
  split2(v[0], v[1], hi[0], lo[0]);
  split2(v[2], v[3], hi[1], lo[1]);
  __bf16* o = out + rb * 64 + j * 4;
  *reinterpret_cast<uint2*>(o) = make_uint2(hi[0], hi[1]);
  *reinterpret_cast<uint2*>(o + 32) = make_uint2(lo[0], lo[1]);
}

// CONV = false: plain k-contiguous operands A[M][K/32][64], B[N][K/32][64] (K % 32 == 0), up to three batch members.
template <bool CONV>
__global__ __launch_bounds__(GNT, 4) void vptr_conv_planes_kernel(const vptr_gemm_desc p, const int epi_rows) {
  constexpr int NFN = 11, BN = 176;
  extern __shared__ __attribute__((aligned(1024))) unsigned char pl_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2, lr = lane & 15, lq = lane >> 4;   // SIMD-balanced column halves, see gemm_p16.hip
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles = tiles_n * ((p.M + GBM - 1) / GBM);
  const int grp = xcd_logical_block() / tiles, logical = xcd_logical_block() - grp * tiles;
  const Member mb = member_of(p, (!CONV && p.batch > 1) ? grp : 0);
  const int m0 = (logical / tiles_n) * GBM, n0 = (logical % tiles_n) * BN;
  const int CB = CONV ? (p.conv_Cin + 31) >> 5 : p.K >> 5;
  const int nk = CONV ? p.conv_KH * p.conv_KW * CB : CB;
  const __bf16* Ail = reinterpret_cast<const __bf16*>(mb.A);
  const __bf16* Bil = reinterpret_cast<const __bf16*>(mb.B);
  const int64_t zero_pix = CONV ? (int64_t)(p.M / (p.conv_OH * p.conv_OW)) * p.conv_IH * p.conv_IW : 0;  // the appended all-zero row

  // A pieces (2 per wave): 8 output pixels x 128 B each; the source pixel depends on the tap
  int a_f[2], a_oy[2], a_ox[2], a_c[2], a_dst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = wave + 8 * i, prow = u * 8 + (lane >> 3);
    a_c[i] = (lane & 7) ^ ((prow >> 1) & 7);
    a_dst[i] = u * 1024;
    a_f[i] = a_oy[i] = a_ox[i] = 0;
    if constexpr (CONV) {
      const int gm = min(m0 + prow, p.M - 1), per = p.conv_OH * p.conv_OW;
      a_f[i] = gm / per;
      const int rem = gm - a_f[i] * per;
      a_oy[i] = rem / p.conv_OW;
      a_ox[i] = rem - a_oy[i] * p.conv_OW;
    }
  }
  // B pieces (3 per wave): 8 weight rows x 128 B
  const __bf16* b_src[3];
  int b_dst[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int u = wave + 8 * i, prow = u * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((prow >> 1) & 7);
    b_src[i] = Bil + (int64_t)min(n0 + prow, p.N - 1) * nk * 64 + c * 8;
    b_dst[i] = GBM * 128 + u * 1024;
  }
  const __bf16* a_src[2];   // source line of the current tap, block 0
  auto set_tap = [&](const int tap) {
    if constexpr (!CONV) {
#pragma unroll
      for (int i = 0; i < 2; ++i) a_src[i] = Ail + (int64_t)min(m0 + (wave + 8 * i) * 8 + (lane >> 3), p.M - 1) * CB * 64 + a_c[i] * 8;
      return;
    }
    const int ky = tap / p.conv_KW, kx = tap - ky * p.conv_KW;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int iy = StageConv::map_coord(a_oy[i], ky, p.conv_IH, p), ix = StageConv::map_coord(a_ox[i], kx, p.conv_IW, p);
      const int64_t pix = (iy >= 0 && ix >= 0) ? ((int64_t)(a_f[i] * p.conv_IH + iy) * p.conv_IW + ix) : zero_pix;
      a_src[i] = Ail + pix * CB * 64 + a_c[i] * 8;
    }
  };
  auto issue = [&](const int kt, const int cb, const int stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const __bf16* g = a_src[i] + cb * 64;
      const uint32_t laddr = (uint32_t)(stage * PL_STAGE + a_dst[i]);
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(laddr)), "v"(g) : "memory");
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const __bf16* g = b_src[i] + (int64_t)kt * 64;
      const uint32_t laddr = (uint32_t)(stage * PL_STAGE + b_dst[i]);
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(laddr)), "v"(g) : "memory");
    }
  };

  f32x4 acc[2][6];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int offAh[2], offAl[2], offBh[6], offBl[6];   // byte offsets of the fragments inside a stage
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int r = wm * 32 + mi * 16 + lr, f = (r >> 1) & 7;
    offAh[mi] = r * 128 + ((lq ^ f) << 4);
    offAl[mi] = r * 128 + (((4 + lq) ^ f) << 4);
  }
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int r = (wn * 6 + ni) * 16 + lr, f = (r >> 1) & 7;
    offBh[ni] = GBM * 128 + r * 128 + ((lq ^ f) << 4);
    offBl[ni] = GBM * 128 + r * 128 + (((4 + lq) ^ f) << 4);
  }

  int tap = 0, cb = 0;   // position of the NEXT step to issue
  set_tap(0);
  issue(0, 0, 0);
  cb = 1;
  const int ntap = CONV ? p.conv_KH * p.conv_KW : 1;
  if (cb == CB) { cb = 0; tap = 1; if (tap < ntap) set_tap(tap); }
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): step kt has landed (the only DMA in flight)
    __syncthreads();                      // ... for every wave, and everyone is done reading the other stage
    if (kt + 1 < nk) {
      issue(kt + 1, cb, (kt + 1) & 1);
      if (++cb == CB) { cb = 0; ++tap; if (tap < ntap) set_tap(tap); }
    }
    const unsigned char* st = pl_smem + (kt & 1) * PL_STAGE;
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      ah[mi] = *reinterpret_cast<const bf16x8*>(st + offAh[mi]);
      al[mi] = *reinterpret_cast<const bf16x8*>(st + offAl[mi]);
    }
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
      if (ni == 5 && wn == 1) break;   // wave-uniform: the 12th fragment of a 176-wide tile is padding
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(st + offBh[ni]);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(st + offBl[ni]);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh, acc[mi][ni], 0, 0, 0);
      }
    }
  }
  if ((epi_rows || p.D_planes) && !p.atomic && epi_vec_ok(p)) {
    __syncthreads();  // the last stage is still being read by slower waves
    gemm_epilogue_rows_halves<NFN>(p, mb, acc, reinterpret_cast<float*>(pl_smem), m0, n0, wm, wn, lr, lq, tid, true, false);
  } else {
    gemm_epilogue_serial<NFN>(p, mb, acc, m0, n0, wm, wn, lr, lq, true, p.atomic != 0);
  }
}

extern "C" int vptr_split_planes(const float* x, void* planes, int64_t rows, int C, vptr_stream_t stream) {
  VPTR_CHECK(x && planes && rows > 0 && C > 0 && C % 4 == 0, "split_planes: bad arguments (C must be a multiple of 4)");
  const int CB = (C + 31) / 32;
  const int64_t n = (rows + 1) * CB * 8;
  split_planes_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, reinterpret_cast<__bf16*>(planes), rows, C, CB);
  VPTR_LAUNCH_CHECK();
  return 0;
}

static int launch_conv_planes(const vptr_gemm_desc& d, hipStream_t st) {
  const bool conv = true;
  VPTR_CHECK(d.b_mode == VPTR_B_PLANES && d.split_k <= 1 && d.precision == 3 && d.ksegs <= 1,
             "vptr_gemm(planes): needs plane weights, split_k = 1, precision 3, no K segments");
  if (conv) VPTR_CHECK(!d.conv_transposed && !d.Dpre && d.batch <= 1 && d.M % (d.conv_OH * d.conv_OW) == 0,
                       "vptr_gemm(conv planes): forward convolution only, M = frames * OH * OW, no Dpre / batch");
  else VPTR_CHECK(d.K % 32 == 0, "vptr_gemm(planes): K must be a multiple of 32 (got %d)", d.K);
  uintptr_t bits = reinterpret_cast<uintptr_t>(d.A) | reinterpret_cast<uintptr_t>(d.B);
  if (d.batch > 1) bits |= reinterpret_cast<uintptr_t>(d.A_x1) | reinterpret_cast<uintptr_t>(d.B_x1) |
                           (d.batch > 2 ? reinterpret_cast<uintptr_t>(d.A_x2) | reinterpret_cast<uintptr_t>(d.B_x2) : 0);
  VPTR_CHECK((bits & 127) == 0, "vptr_gemm(planes): plane buffers must be 128-byte aligned");
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_conv_planes_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * PL_STAGE) != hipSuccess ||
        false) {
      vptr_set_error("vptr_gemm(planes): cannot reserve %d bytes of LDS", 2 * PL_STAGE);
      return -1;
    }
    attr_set = true;
  }
  const int tiles = ((d.M + GBM - 1) / GBM) * ((d.N + 175) / 176) * (conv ? 1 : d.batch);
  vptr_conv_planes_kernel<true><<<tiles, GNT, 2 * PL_STAGE, st>>>(d, epi_rows_flag() & 2);
  return 0;
}

extern "C" int vptr_gemm(const vptr_gemm_desc* desc, vptr_stream_t stream) {
  VPTR_CHECK(desc != nullptr, "vptr_gemm: null descriptor");
  vptr_gemm_desc d = *desc;
  VPTR_CHECK(d.M > 0 && d.N > 0 && d.K > 0, "vptr_gemm: empty problem M=%d N=%d K=%d", d.M, d.N, d.K);
  VPTR_CHECK(d.A && d.B && (d.D || d.D_planes), "vptr_gemm: null operand");
  if (d.D_planes) VPTR_CHECK(d.a_mode == VPTR_A_CONV_PLANES, "vptr_gemm: D_planes is an output of the conv plane kernel only");
  VPTR_CHECK(d.precision == 1 || d.precision == 3, "vptr_gemm: precision must be 1 or 3 (got %d)", d.precision);
  if (d.a_mode != VPTR_A_P16) VPTR_CHECK(d.batch_accum == 0 && d.batch_stride_d == 0, "vptr_gemm: batch_accum / strided batches are options of the P16 kernels only");
  if (d.a_mode == VPTR_A_P16 || d.b_mode == VPTR_B_P16) {
    VPTR_CHECK(d.d_row_w == 0, "vptr_gemm(p16): no output row map");
    const int rc = vptr_gemm_p16_launch(d, reinterpret_cast<hipStream_t>(stream));
    if (rc) return rc;
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  VPTR_CHECK(!d.d_p16, "vptr_gemm: d_p16 is an output format of the P16 kernels only");
  if (d.d_row_w > 0 || d.d_row_off != 0)
    VPTR_CHECK(d.d_row_w > 0 && d.a_mode <= VPTR_A_CONV && !d.residual && !d.Dpre && d.batch <= 1 && !d.atomic && d.split_k <= 1,
               "vptr_gemm: the output row map (d_row_w / d_row_off) is an option of the fp32-staged kernels without residual / Dpre / batch / atomics");
  VPTR_CHECK(!d.act_grad_src && !d.frame_stats, "vptr_gemm: act_grad_src / frame_stats are epilogues of the P16 kernels only");
  VPTR_CHECK(d.a_mode != 4, "vptr_gemm: a_mode 4 (k-contiguous 32-wide planes) was replaced by VPTR_A_P16");
  if (d.a_mode == VPTR_A_CONV_PLANES) {
    VPTR_CHECK(d.d_row_w == 0, "vptr_gemm(planes): no output row map");
    if (d.alpha == 0.f) d.alpha = 1.f;
    if (d.batch < 1) d.batch = 1;
    if (d.batch > 1) {
      VPTR_CHECK(d.batch <= 3 && d.A_x1 && d.B_x1 && d.D_x1 && (d.batch < 3 || (d.A_x2 && d.B_x2 && d.D_x2)) && !d.Dpre && !d.residual && !d.atomic,
                 "vptr_gemm(planes): bad batch members");
      if (d.alpha_x1 == 0.f) d.alpha_x1 = 1.f;
      if (d.alpha_x2 == 0.f) d.alpha_x2 = 1.f;
    }
    if (d.rowscale) VPTR_CHECK(d.rs_div >= 1 && d.rs_mod >= 1, "vptr_gemm: rowscale needs rs_div, rs_mod >= 1");
    if (d.D_planes) {
      vptr_gemm_desc t = d;
      if (!t.D) t.D = reinterpret_cast<float*>(d.D_planes);   // alignment test only
      VPTR_CHECK(!d.atomic && d.batch == 1 && (d.N & 3) == 0 && (d.ldd & 3) == 0 && (d.ldr & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(d.D_planes) | reinterpret_cast<uintptr_t>(t.D) | reinterpret_cast<uintptr_t>(d.residual) |
                       reinterpret_cast<uintptr_t>(d.bias) | reinterpret_cast<uintptr_t>(d.colscale) | reinterpret_cast<uintptr_t>(d.Dpre)) & 15) == 0,
                 "vptr_gemm(planes): D_planes needs the row-major epilogue (N, ldd, ldr multiples of 4, 16-byte aligned pointers, no atomics / batch)");
    } else {
      VPTR_CHECK(d.D != nullptr, "vptr_gemm(planes): null output");
    }
    if (d.a_mode == VPTR_A_CONV_PLANES)
      VPTR_CHECK(d.K == d.conv_KH * d.conv_KW * d.conv_Cin && d.conv_stride >= 1 && d.conv_OH > 0 && d.conv_OW > 0, "vptr_gemm(conv planes): bad geometry");
    if (d.dropout_p > 0.f) VPTR_CHECK(d.seed_dev != nullptr && d.dropout_p < 1.f, "vptr_gemm: dropout needs seed_dev and p < 1");
    const int rc = launch_conv_planes(d, reinterpret_cast<hipStream_t>(stream));
    if (rc) return rc;
    VPTR_LAUNCH_CHECK();
    return 0;
  }
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
  if (d.batch < 1) d.batch = 1;
  if (d.ksegs < 1) d.ksegs = 1;
  VPTR_CHECK(d.batch <= 3 && d.ksegs <= 3, "vptr_gemm: at most 3 batch members / K segments");
  if (d.batch > 1 || d.ksegs > 1) {
    VPTR_CHECK(d.batch == 1 || d.ksegs == 1, "vptr_gemm: batch and ksegs are mutually exclusive");
    VPTR_CHECK(d.split_k == 1 && d.a_mode != VPTR_A_CONV && !d.a_rowsum, "vptr_gemm: batched / K-segmented launches need split_k = 1, no conv operand, no a_rowsum");
    const int extra = (d.batch > 1 ? d.batch : d.ksegs) - 1;
    VPTR_CHECK(d.A_x1 && d.B_x1 && al16(d.A_x1) && al16(d.B_x1), "vptr_gemm: member/segment 1 needs 16-byte aligned A_x1, B_x1");
    if (extra > 1) VPTR_CHECK(d.A_x2 && d.B_x2 && al16(d.A_x2) && al16(d.B_x2), "vptr_gemm: member/segment 2 needs 16-byte aligned A_x2, B_x2");
    if (d.batch > 1) {
      VPTR_CHECK(d.D_x1 && (extra < 2 || d.D_x2), "vptr_gemm: every batch member needs its D_x");
      if (d.alpha_x1 == 0.f) d.alpha_x1 = 1.f;
      if (d.alpha_x2 == 0.f) d.alpha_x2 = 1.f;
    }
    if (d.batch > 1) VPTR_CHECK(!d.Dpre && !d.residual && !d.atomic, "vptr_gemm: batched launches support bias / alpha / act / dropout epilogues only");
  }
  if (d.a_rowsum) VPTR_CHECK(d.a_mode == VPTR_A_KSTRIDED && g_gemm_variant != 0, "vptr_gemm: a_rowsum needs a k-strided A operand");

  // k range per split, multiple of the K tile
  int k_chunk = ((d.K + d.split_k - 1) / d.split_k + GBK - 1) / GBK * GBK;
  const int splits = (d.K + k_chunk - 1) / k_chunk;

  const int nfn = nfn_for(d.N);
  const int bn = 16 * nfn;
  const int tiles_m = (d.M + GBM - 1) / GBM, tiles_n = (d.N + bn - 1) / bn;
  dim3 grid((unsigned)(tiles_m * tiles_n * splits * d.batch), 1, 1);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc;
  if (nfn == 11) rc = launch_prec<11>(d, grid, k_chunk, st);
  else if (nfn == 8) rc = launch_prec<8>(d, grid, k_chunk, st);
  else rc = launch_prec<4>(d, grid, k_chunk, st);
  if (rc) return rc;
  VPTR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vptr_gemm_tile_cols(int N) { return 16 * nfn_for(N); }

template <int NFN, int NPASS>
static int launch_grouped(const vptr_gemm_desc* descs, const int* tile_start, int count, int total_tiles, hipStream_t st) {
  constexpr int NFW = (NFN + 1) / 2, BROWS = 2 * NFW * 16, NPL = (NPASS == 3) ? 2 : 1;
  constexpr int LOOP_BYTES = 2 * NPL * (GBM + BROWS) * GLP * (int)sizeof(__bf16);
  constexpr int LDS_BYTES = LOOP_BYTES > epi_lds_bytes<NFN>() ? LOOP_BYTES : epi_lds_bytes<NFN>();
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_grouped_kernel<NFN, NPASS, VPTR_A_KSTRIDED, VPTR_B_KSTRIDED>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
      vptr_set_error("vptr_gemm_grouped: cannot reserve %d bytes of LDS", LDS_BYTES);
      return -1;
    }
    attr_set = true;
  }
  vptr_gemm_grouped_kernel<NFN, NPASS, VPTR_A_KSTRIDED, VPTR_B_KSTRIDED><<<dim3((unsigned)total_tiles), GNT, LDS_BYTES, st>>>(descs, tile_start, count);
  return 0;
}

extern "C" int vptr_gemm_grouped(const vptr_gemm_desc* proto, const vptr_gemm_desc* descs_dev, const int* tile_start_dev, int count,
                                 int total_tiles, vptr_stream_t stream) {
  VPTR_CHECK(proto && descs_dev && tile_start_dev, "vptr_gemm_grouped: null argument");
  VPTR_CHECK(count > 0 && total_tiles > 0, "vptr_gemm_grouped: empty group");
  if (proto->a_mode == VPTR_A_P16T) {
    const int rc = vptr_wgrad_p16_launch(proto, descs_dev, tile_start_dev, count, total_tiles, reinterpret_cast<hipStream_t>(stream));
    if (rc) return rc;
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  VPTR_CHECK(proto->a_mode == VPTR_A_KSTRIDED && proto->b_mode == VPTR_B_KSTRIDED,
             "vptr_gemm_grouped: only k-strided x k-strided problems (weight gradients) are grouped");
  VPTR_CHECK(proto->precision == 1 || proto->precision == 3, "vptr_gemm_grouped: precision must be 1 or 3");
  const int nfn = nfn_for(proto->N);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc;
  if (proto->precision == 3) rc = nfn == 11 ? launch_grouped<11, 3>(descs_dev, tile_start_dev, count, total_tiles, st)
                                 : nfn == 8 ? launch_grouped<8, 3>(descs_dev, tile_start_dev, count, total_tiles, st)
                                            : launch_grouped<4, 3>(descs_dev, tile_start_dev, count, total_tiles, st);
  else rc = nfn == 11 ? launch_grouped<11, 1>(descs_dev, tile_start_dev, count, total_tiles, st)
          : nfn == 8 ? launch_grouped<8, 1>(descs_dev, tile_start_dev, count, total_tiles, st)
                     : launch_grouped<4, 1>(descs_dev, tile_start_dev, count, total_tiles, st);
  if (rc) return rc;
  VPTR_LAUNCH_CHECK();
  return 0;
}
