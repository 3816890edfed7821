// Split-bf16 MFMA GEMMs on "P16" operands (gfx950): the convert-once path of every nn.Linear forward, input gradient and
// weight gradient of the VPTR transformers.
//
// P16 is the operand format: a [rows][C] matrix with the bytes, pitch and shape of its fp32 original (C % 16 == 0) in which
// every 16-channel granule (64 bytes) holds 16 bf16 `hi` followed by 16 bf16 `lo`, x = hi + lo + O(2^-17 |x|).  The producer
// of a tensor (LayerNorm, attention core, normalise + GELU, a GEMM epilogue, the optimizer for the weights) writes it once;
// the GEMMs stage it with global_load_lds_dwordx4 -- no fp32 -> bf16 split and no ds_write in any main loop, which is what
// bounded the register-staged kernels of gemm.hip.
//
//   nt  (vptr_gemm, a_mode = VPTR_A_P16, b_mode = VPTR_B_P16):  D[M,N] = epi( A[M,K] . B[N,K]^T ), both k-contiguous:
//        forward (B = W planes) and input gradients (B = W^T planes); batch members and K segments as in gemm.hip.
//   tn  (vptr_gemm_grouped, a_mode = VPTR_A_P16T, b_mode = VPTR_B_P16T):  dW[NG,KX] += alpha * G[T,NG]^T . X[T,KX], both
//        operands TOKEN-major: the MFMA fragments (8 consecutive tokens per lane) come out of a [tokens][16 channels] LDS
//        image through ds_read_b64_tr_b16; the bias gradient (column sums of G) rides in the otherwise idle 12th column
//        fragment of the odd wave column as a product with a vector of ones.
//
// Tile 128 x 176 x 32, 8 waves (4 x 2) of 32 x 96, two 40 KB stages of 40 DMA pieces (1 KB = 8 rows x 128 B each: full
// 128-byte lines on the global side), <= 128 VGPRs: two workgroups per CU.  tools/gemm_p16_probe.hip is the stand-alone
// study (nt 290-315 TFLOP/s, tn 230-265 TFLOP/s at the model's shapes vs 200-237 / 159 for the register-staged kernels).
#include "gemm_shared.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int P16_STAGE = 40 * 1024;  // 16 pieces of A + 24 pieces of B

#define P16_GLDS(laddr, gptr) \
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(laddr)), "v"(gptr) : "memory")

// ---------------------------------------------------------------------------------------------------------------------
// fp32 -> P16 (one pass; used where no producer kernel can emit the format itself)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void to_p16_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    vptr_p16_store4(out, i * 4, v);   // C % 16 == 0: granules never straddle rows, so the flat element index addresses them
  }
}
extern "C" int vptr_to_p16(const float* x, void* out, int64_t rows, int C, vptr_stream_t stream) {
  VPTR_CHECK(x && out && rows > 0 && C > 0 && C % 16 == 0, "to_p16: C must be a positive multiple of 16 (got %d)", C);
  VPTR_CHECK(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "to_p16: pointers must be 16-byte aligned");
  const int64_t n4 = rows * C / 4;
  to_p16_kernel<<<(unsigned)hmin64((n4 + 255) / 256, 16384), 256, 0, (hipStream_t)stream>>>(x, reinterpret_cast<unsigned char*>(out), n4);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight planes: W[N][K] fp32 -> Wp[N][K] P16 (forward operand) and WT[K][N] P16 (input-gradient operand), for a whole table of
// weights in one launch (once per optimizer step: 2 x 473 MB written for the K64 transformer, ~0.3 ms).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weight_planes_kernel(const vptr_wplane_entry* __restrict__ tab, const int* __restrict__ tile_start,
                                                            int count) {
  __shared__ float tile[32][33];
  const int b = blockIdx.x;
  int lo = 0, hi = count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_start[mid] <= b) lo = mid;
    else hi = mid - 1;
  }
  const vptr_wplane_entry e = tab[lo];
  const int t = b - tile_start[lo], tk = (e.K + 31) >> 5;
  const int n0 = (t / tk) * 32, k0 = (t % tk) * 32;
  const int r = threadIdx.x >> 3, c = (threadIdx.x & 7) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool ok = n0 + r < e.N && k0 + c < e.K;   // K % 16 == 0: a float4 is inside or outside as a whole
  if (ok) {
    v = *reinterpret_cast<const float4*>(e.W + (int64_t)(n0 + r) * e.ldw + k0 + c);
    vptr_p16_store4(reinterpret_cast<unsigned char*>(e.Wp), (int64_t)(n0 + r) * e.K + k0 + c, v);
  }
  tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
  __syncthreads();
  if (k0 + r < e.K && n0 + c < e.N) {
    const float4 w = make_float4(tile[c][r], tile[c + 1][r], tile[c + 2][r], tile[c + 3][r]);
    vptr_p16_store4(reinterpret_cast<unsigned char*>(e.WT), (int64_t)(k0 + r) * e.N + n0 + c, w);
  }
}
extern "C" int vptr_weight_planes(const vptr_wplane_entry* table_dev, const int* tile_start_dev, int count, int total_tiles,
                                  vptr_stream_t stream) {
  VPTR_CHECK(table_dev && tile_start_dev && count > 0 && total_tiles > 0, "weight_planes: bad arguments");
  weight_planes_kernel<<<total_tiles, 256, 0, (hipStream_t)stream>>>(table_dev, tile_start_dev, count);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// nt kernel.  Stage: piece u < 16 = rows 8u .. 8u+7 of the A tile, piece 16 + v = rows 8v .. of the B tile; a piece row is the
// 128 bytes of one K-step (two granules: hi16 | lo16 | hi16 | lo16), chunk c of row r at physical chunk c ^ ((r >> 1) & 7):
// the 16 rows of a ds_read_b128 fragment read hit 16 different 16-byte bank groups.  The fragment of lane (lr, lq) is
// k = 8 lq .. 8 lq + 7 of row lr: hi chunk (lq >> 1) * 4 + (lq & 1), lo chunk = hi chunk + 2.
// K % 32 == 16: the last step's second granule does not exist; its DMA lanes re-fetch the first one (always valid memory)
// and the A fragments of lanes lq >= 2 are zeroed.
// ---------------------------------------------------------------------------------------------------------------------
// LEAN: plain epilogue only (gemm_shared.h).  NST >= 3: the instantiations for grids of at most one workgroup per CU (nothing else on
// the CU hides a stall): NST stages with the DMA NST - 1 K-steps ahead, its pieces issued between the MFMA groups instead of in a
// burst after the barrier (cache-cold 10 240 x 528 x 2112: 79.7 -> 73.1 us in tools/gemm_p16_probe with three stages; four stages = the
// CU's whole 160 KB: 77.9 -> 74.0 us inside the step, VPTR_GEMM_LONE_STAGES=3 restores three).  Chosen by the launcher.
template <int EPI, int NST>   // EPI: 0 every epilogue option, 1 lean, 2 activation gradient, 3 lean + row scale + dropout, 4 activation + Dpre + dropout (gemm_shared.h)
__global__ __launch_bounds__(GNT, NST >= 3 ? 2 : 4) void vptr_gemm_p16_kernel(const vptr_gemm_desc p, const int epi_rows_) {
  constexpr int NFN = 11, BN = 176;
  extern __shared__ __attribute__((aligned(1024))) unsigned char p16_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int epi_rows = epi_rows_ & 0xff;
  // experiment (VPTR_GEMM_PRIO=1): static priority for the later-dispatched half of the workgroup -- on every SIMD wave w + 4 is the
  // arbitration loser against wave w (MI355X_MICROARCH.md, "two waves per SIMD", item 4)
  if ((epi_rows_ & 0x100) && wave >= 4) __builtin_amdgcn_s_setprio(1);
#ifdef VPTR_P16_TIMING   // debug build: p.Dpre is a [tiles][4] int64 buffer of wall-clock stamps (100 MHz) -- tools/nt_timing.py
  const long long tm0 = wall_clock64();
  long long tm1 = 0;
#endif
  // waves w and w + 4 of a workgroup share a SIMD: with wn = wave >> 2 every SIMD hosts one wave of each column half, so skipping
  // the padding fragment of the odd half (176 = 11 fragments = 6 + 5) takes 1/12 off every SIMD's MFMA time (+4-5 % measured)
  const int wm = wave & 3, wn = wave >> 2, lr = lane & 15, lq = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles = tiles_n * ((p.M + GBM - 1) / GBM);
  const int lg = xcd_logical_block();
  const int grp = lg / tiles, tile = lg - grp * tiles;
  const Member mb = member_of(p, p.batch > 1 ? grp : 0);
  const int m0 = (tile / tiles_n) * GBM, n0 = (tile % tiles_n) * BN;
  const int nk = (p.K + 31) >> 5;
  const bool ktail = (p.K & 16) != 0;
  const int nseg = p.ksegs > 1 ? p.ksegs : 1;
  const int64_t pa = p.lda * 4, pb = p.ldb * 4;
  const unsigned char* Ab = reinterpret_cast<const unsigned char*>(mb.A);
  const unsigned char* Bb = reinterpret_cast<const unsigned char*>(mb.B);
  // K segments: byte offsets of segment s relative to segment 0 (plain integers, see KSEG_OFFSETS in gemm.hip)
  const int64_t sA1 = nseg > 1 ? (p.A_x1 - p.A) * 4 : 0, sA2 = nseg > 2 ? (p.A_x2 - p.A) * 4 : 0;
  const int64_t sB1 = nseg > 1 ? (p.B_x1 - p.B) * 4 : 0, sB2 = nseg > 2 ? (p.B_x2 - p.B) * 4 : 0;

  const unsigned char* srcA[2];
  const unsigned char* srcB[3];
  int tadj;   // this lane's chunk is in the second granule of a K-step: -64 in a tail step (the same for all its pieces)
  {
    const int pch = lane & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int prow = (wave + 8 * i) * 8 + (lane >> 3);
      const int c = pch ^ ((prow >> 1) & 7);
      srcA[i] = Ab + (int64_t)min(m0 + prow, p.M - 1) * pa + c * 16;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int prow = (wave + 8 * i) * 8 + (lane >> 3);
      const int c = pch ^ ((prow >> 1) & 7);
      srcB[i] = Bb + (int64_t)min(n0 + prow, p.N - 1) * pb + c * 16;
    }
    // rows 8u + (lane >> 3) with u = wave + 8 i: (prow >> 1) & 7 = ((lane >> 4) + 4 * wave) & 7 for every piece of this lane
    const int c = pch ^ (((lane >> 4) + 4 * wave) & 7);
    tadj = c >= 4 ? -64 : 0;
  }
#ifdef VPTR_NT_ELIM   // elimination build (WRONG results): only the first VPTR_NT_ELIM of a wave's 5 pieces per K-step are staged
  constexpr int NPIECE = VPTR_NT_ELIM;
#else
  constexpr int NPIECE = 5;
#endif
  auto issue1 = [&](const int kt, const int stage, const int i) {   // piece i of this wave: 0, 1 = A, 2 .. 4 = B
    if (i >= NPIECE) return;
    const int sg = (int)(kt >= nk) + (int)(kt >= 2 * nk);
    const int kk = kt - sg * nk;
    const int64_t off = (int64_t)kk * 128 + ((ktail && kk == nk - 1) ? tadj : 0);
    if (i < 2) P16_GLDS((uint32_t)(stage * P16_STAGE + (wave + 8 * i) * 1024), srcA[i] + ((sg == 0 ? (int64_t)0 : (sg == 1 ? sA1 : sA2)) + off));
    else P16_GLDS((uint32_t)(stage * P16_STAGE + 16384 + (wave + 8 * (i - 2)) * 1024), srcB[i - 2] + ((sg == 0 ? (int64_t)0 : (sg == 1 ? sB1 : sB2)) + off));
  };
  auto issue = [&](const int kt, const int stage) {
#pragma unroll
    for (int i = 0; i < 5; ++i) issue1(kt, stage, i);
  };

  f32x4 acc[2][6];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int offAh[2], offBh[6];   // byte offsets of the hi fragments inside a stage; lo = chunk + 2
  const int ch = (lq >> 1) * 4 + (lq & 1);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int r = wm * 32 + mi * 16 + lr, f = (r >> 1) & 7;
    offAh[mi] = r * 128 + ((ch ^ f) << 4);
  }
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int r = (wn * 6 + ni) * 16 + lr, f = (r >> 1) & 7;
    offBh[ni] = 16384 + r * 128 + ((ch ^ f) << 4);
  }
  // lo chunk = hi chunk + 2 under the XOR swizzle: (ch + 2) ^ f = (ch ^ f) ^ 2 because bit 1 of ch is clear
  const int nkt = nk * nseg;
  issue(0, 0);
  if (NST >= 3 && nkt > 1) issue(1, 1);
  if (NST >= 4 && nkt > 2) issue(2, 2);
  int sc = 0, sn = NST - 1;   // NST >= 3: stage of step kt, stage that step kt + NST - 1 goes to
#ifdef VPTR_P16_TIMING
  long long cyc_wait = 0, cyc_issue = 0, cyc_t = 0;   // shader-clock cycles this wave spent waiting for the step / issuing its DMA
#endif
  for (int kt = 0; kt < nkt; ++kt) {
#ifdef VPTR_P16_TIMING
    cyc_t = clock64();
#endif
    // step kt has landed: with three stages step kt + 1 (5 pieces per wave) may still be in flight
    if (NST >= 4 && kt + 2 < nkt) __builtin_amdgcn_s_waitcnt(0x0f70 | (2 * NPIECE));   // four stages: steps kt + 1 and kt + 2 may be in flight
    else if (NST >= 3 && kt + 1 < nkt) __builtin_amdgcn_s_waitcnt(0x0f70 | NPIECE);
    else __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();                      // ... for every wave, and everyone is done reading the stage the next DMA overwrites
#ifdef VPTR_P16_TIMING
    if (kt == 0) tm1 = wall_clock64();
    { const long long c = clock64(); cyc_wait += c - cyc_t; cyc_t = c; }
#endif
    if (NST == 2 && kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);
#ifdef VPTR_P16_TIMING
    cyc_issue += clock64() - cyc_t;
#endif
    const unsigned char* st = p16_smem + (NST >= 3 ? sc : (kt & 1)) * P16_STAGE;
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      ah[mi] = *reinterpret_cast<const bf16x8*>(st + offAh[mi]);
      al[mi] = *reinterpret_cast<const bf16x8*>(st + (offAh[mi] ^ 32));
    }
    if (ktail) {
      const int sg = (int)(kt >= nk) + (int)(kt >= 2 * nk);
      if (kt - sg * nk == nk - 1 && lq >= 2) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          ah[mi] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
          al[mi] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
    }
    // the next B fragment pair is requested before the MFMAs of the current one (an in-order wave otherwise waits out every
    // LDS round trip with the matrix pipe idle)
    bf16x8 bh[2], bl[2];
    bh[0] = *reinterpret_cast<const bf16x8*>(st + offBh[0]);
    bl[0] = *reinterpret_cast<const bf16x8*>(st + (offBh[0] ^ 32));
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
      if (ni == 5 && wn == 1) break;   // wave-uniform: fragment 11 of the tile does not exist
      if (ni + 1 < 6 && !(ni + 1 == 5 && wn == 1)) {
        bh[(ni + 1) & 1] = *reinterpret_cast<const bf16x8*>(st + offBh[ni + 1]);
        bl[(ni + 1) & 1] = *reinterpret_cast<const bf16x8*>(st + (offBh[ni + 1] ^ 32));
      }
      if (NST >= 3 && ni < 5 && kt + NST - 1 < nkt) issue1(kt + NST - 1, sn, ni);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh[ni & 1], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl[ni & 1], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh[ni & 1], acc[mi][ni], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NST >= 3) {
      sc = sc == NST - 1 ? 0 : sc + 1;
      sn = sn == NST - 1 ? 0 : sn + 1;
    }
  }
#ifdef VPTR_P16_TIMING
  const long long tm2 = wall_clock64();
  const long long cyc_end = clock64();
  long long* const tbuf = reinterpret_cast<long long*>(p.Dpre);
  long long tme[5] = {0, 0, 0, 0, 0};
#endif
  constexpr bool LEAN = EPI != 0;
  if (LEAN) {
    __syncthreads();  // the last stage is still being read by slower waves
#ifdef VPTR_P16_TIMING
    gemm_epilogue_rows_halves_batched<NFN, EPI>(p, mb, acc, reinterpret_cast<float*>(p16_smem), m0, n0, wm, wn, lr, lq, tid, true, false, tme);
#else
    gemm_epilogue_rows_halves_batched<NFN, EPI>(p, mb, acc, reinterpret_cast<float*>(p16_smem), m0, n0, wm, wn, lr, lq, tid, true, false);
#endif
  } else if (NST >= 3 && (epi_rows || p.d_p16) && !p.atomic && epi_vec_ok(p)) {
    // the full epilogue with its operand loads batched: affordable under this instantiation's 256-register budget
    __syncthreads();
    gemm_epilogue_rows_halves_batched<NFN, 0>(p, mb, acc, reinterpret_cast<float*>(p16_smem), m0, n0, wm, wn, lr, lq, tid, true, false);
  } else if ((epi_rows || p.d_p16) && !p.atomic && epi_vec_ok(p)) {
    __syncthreads();
    gemm_epilogue_rows_halves<NFN>(p, mb, acc, reinterpret_cast<float*>(p16_smem), m0, n0, wm, wn, lr, lq, tid, true, false);
  } else {
    gemm_epilogue_serial<NFN>(p, mb, acc, m0, n0, wm, wn, lr, lq, true, p.atomic != 0);
  }
#ifdef VPTR_P16_TIMING
  if (EPI == 1 && tbuf) {
    __syncthreads();
    if (tid == 0) {
      tbuf[blockIdx.x * 16 + 0] = tm0; tbuf[blockIdx.x * 16 + 1] = tm1; tbuf[blockIdx.x * 16 + 2] = tm2; tbuf[blockIdx.x * 16 + 3] = wall_clock64();
      for (int i = 0; i < 5; ++i) tbuf[blockIdx.x * 16 + 4 + i] = tme[i];
      tbuf[blockIdx.x * 16 + 9] = cyc_wait; tbuf[blockIdx.x * 16 + 10] = cyc_issue; tbuf[blockIdx.x * 16 + 11] = clock64() - cyc_end;
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// nt kernel, REGISTER-staged (round 6; VPTR_GEMM_RS).  Same tile, stage image, fragment reads, MFMA order and epilogues as
// vptr_gemm_p16_kernel; only the operand path differs: every lane fetches its five 16-byte chunks of a K-step with plain
// global_load_dwordx4 (the chunk that belongs at its linear LDS position under the swizzle -- the address map of the DMA pieces) NRS
// K-steps ahead into registers and stores them with ds_write_b128, interleaved piece by piece between the MFMA groups.  Why: rounds 4 - 5
// located the bound of the DMA-staged kernels in the global_load_lds path itself (the DMA-only build takes 89 % of the full launch, ~15 B
// per cycle and CU; two thirds of that cost is per DMA INSTRUCTION, independent of bytes and of where they come from), while ordinary
// vector loads from L2 are priced at ~56 B per cycle and CU and ds_write_b128 at ~79 (MI355X_MICROARCH.md).  P16 operands need no
// conversion, so the register path costs 4 VGPRs per piece and set, no VALU.  Two LDS stages, ONE barrier per K-step: the stores of step
// kt + 1 go to the stage whose last readers (step kt - 1) all passed this step's barrier; they are read after the next one.
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int EPI, int NRS, int WGS>   // NRS: register sets (K-steps the global loads run ahead of their LDS stores); WGS: workgroups per CU built for
__global__ __launch_bounds__(GNT, WGS == 1 ? 2 : 4) void vptr_gemm_p16_rs_kernel(const vptr_gemm_desc p, const int epi_rows_) {
  static_assert(NRS == 1 || NRS == 2, "one or two register sets");
  constexpr int NFN = 11, BN = 176;
  extern __shared__ __attribute__((aligned(1024))) unsigned char p16_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int epi_rows = epi_rows_ & 0xff;
  const int wm = wave & 3, wn = wave >> 2, lr = lane & 15, lq = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles = tiles_n * ((p.M + GBM - 1) / GBM);
  const int lg = xcd_logical_block();
  const int grp = lg / tiles, tile = lg - grp * tiles;
  const Member mb = member_of(p, p.batch > 1 ? grp : 0);
  const int m0 = (tile / tiles_n) * GBM, n0 = (tile % tiles_n) * BN;
  const int nk = (p.K + 31) >> 5;
  const bool ktail = (p.K & 16) != 0;
  const int nseg = p.ksegs > 1 ? p.ksegs : 1;
  const int nkt = nk * nseg;
  const int64_t pa = p.lda * 4, pb = p.ldb * 4;
  const unsigned char* Ab = reinterpret_cast<const unsigned char*>(mb.A);
  const unsigned char* Bb = reinterpret_cast<const unsigned char*>(mb.B);
  const int64_t sA1 = nseg > 1 ? (p.A_x1 - p.A) * 4 : 0, sA2 = nseg > 2 ? (p.A_x2 - p.A) * 4 : 0;
  const int64_t sB1 = nseg > 1 ? (p.B_x1 - p.B) * 4 : 0, sB2 = nseg > 2 ? (p.B_x2 - p.B) * 4 : 0;

  const unsigned char* src[5];   // pieces 0, 1 = A rows 8 (wave + 8 i) + (lane >> 3); 2 .. 4 = B rows likewise
  int tadj;
  {
    const int pch = lane & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int prow = (wave + 8 * i) * 8 + (lane >> 3);
      src[i] = Ab + (int64_t)min(m0 + prow, p.M - 1) * pa + (pch ^ ((prow >> 1) & 7)) * 16;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int prow = (wave + 8 * i) * 8 + (lane >> 3);
      src[2 + i] = Bb + (int64_t)min(n0 + prow, p.N - 1) * pb + (pch ^ ((prow >> 1) & 7)) * 16;
    }
    const int c = pch ^ (((lane >> 4) + 4 * wave) & 7);
    tadj = c >= 4 ? -64 : 0;
  }
  // LDS position of this lane's chunk of piece i inside a stage (the DMA's linear order: piece * 1024 + lane * 16)
  const int dstA = wave * 1024 + lane * 16, dstB = 16384 + wave * 1024 + lane * 16;
  // byte offsets of K-step kt relative to src[]: wave-uniform (A, B) + this lane's tail adjustment.  Steps beyond the last are CLAMPED to
  // it (a redundant, L2-resident fetch): every step then issues exactly five loads, so the compiler's vmcnt bookkeeping is static
  // (conditional loads made it drain vmcnt(0) before a step's first LDS store)
  auto step_off = [&](const int kt_, int64_t& oa, int64_t& ob) {
    const int kt = min(kt_, nkt - 1);
    const int sg = (int)(kt >= nk) + (int)(kt >= 2 * nk);
    const int kk = kt - sg * nk;
    const int64_t off = (int64_t)kk * 128 + ((ktail && kk == nk - 1) ? tadj : 0);
    oa = (sg == 0 ? (int64_t)0 : (sg == 1 ? sA1 : sA2)) + off;
    ob = (sg == 0 ? (int64_t)0 : (sg == 1 ? sB1 : sB2)) + off;
  };
  auto fetch = [&](const int i, const int64_t oa, const int64_t ob) -> u32x4 {
    return *reinterpret_cast<const u32x4*>(src[i] + (i < 2 ? oa : ob));
  };
  auto put = [&](const int stage, const int i, const u32x4 v) {
    *reinterpret_cast<u32x4*>(p16_smem + stage * P16_STAGE + (i < 2 ? dstA + i * 8192 : dstB + (i - 2) * 8192)) = v;
  };

  f32x4 acc[2][6];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int offAh[2], offBh[6];
  const int ch = (lq >> 1) * 4 + (lq & 1);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int r = wm * 32 + mi * 16 + lr, f = (r >> 1) & 7;
    offAh[mi] = r * 128 + ((ch ^ f) << 4);
  }
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int r = (wn * 6 + ni) * 16 + lr, f = (r >> 1) & 7;
    offBh[ni] = 16384 + r * 128 + ((ch ^ f) << 4);
  }
  u32x4 rg[NRS][5];   // rg[s]: the chunks of step kt + 1 + s (s = (kt + 1 + s') & (NRS - 1) with the loop unrolled by NRS: compile-time indices)
  // prologue: step 0 straight into stage 0, steps 1 .. NRS into the register sets
  {
    u32x4 t0[5];
    int64_t oa, ob;
    step_off(0, oa, ob);
#pragma unroll
    for (int i = 0; i < 5; ++i) t0[i] = fetch(i, oa, ob);
#pragma unroll
    for (int s = 0; s < NRS; ++s) {
      step_off(1 + s, oa, ob);
#pragma unroll
      for (int i = 0; i < 5; ++i) rg[(1 + s) % NRS][i] = fetch(i, oa, ob);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) put(0, i, t0[i]);
  }
  auto step = [&](const int kt, auto SET_) {
    constexpr int SET = decltype(SET_)::value;     // register set that holds step kt + 1 (= (kt + 1) % NRS)
    __syncthreads();   // stage kt & 1 is complete (every wave's stores of step kt), stage (kt + 1) & 1 is free (every wave's reads of step kt - 1)
    const unsigned char* st = p16_smem + (kt & 1) * P16_STAGE;
    const int sn = (kt + 1) & 1;
    int64_t oa, ob;
    step_off(kt + 1 + NRS, oa, ob);
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      ah[mi] = *reinterpret_cast<const bf16x8*>(st + offAh[mi]);
      al[mi] = *reinterpret_cast<const bf16x8*>(st + (offAh[mi] ^ 32));
    }
    if (ktail) {
      const int sg = (int)(kt >= nk) + (int)(kt >= 2 * nk);
      if (kt - sg * nk == nk - 1 && lq >= 2) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          ah[mi] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
          al[mi] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
    }
    bf16x8 bh[2], bl[2];
    bh[0] = *reinterpret_cast<const bf16x8*>(st + offBh[0]);
    bl[0] = *reinterpret_cast<const bf16x8*>(st + (offBh[0] ^ 32));
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
      if (ni == 5 && wn == 1) break;   // wave-uniform: fragment 11 of the tile does not exist
      if (ni + 1 < 6 && !(ni + 1 == 5 && wn == 1)) {
        bh[(ni + 1) & 1] = *reinterpret_cast<const bf16x8*>(st + offBh[ni + 1]);
        bl[(ni + 1) & 1] = *reinterpret_cast<const bf16x8*>(st + (offBh[ni + 1] ^ 32));
      }
#ifndef VPTR_RS_NOSTAGE   // elimination build: no operand movement after the prologue
      if (ni < 5) {                    // piece ni of step kt + 1: registers -> the other stage; its register then takes step kt + 1 + NRS
        put(sn, ni, rg[SET][ni]);      // (in the last step: a clamped copy of itself into the stage nobody reads any more)
        rg[SET][ni] = fetch(ni, oa, ob);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
#ifndef VPTR_RS_NOMFMA    // elimination build: operand movement, fragment reads and barriers only
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh[ni & 1], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl[ni & 1], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh[ni & 1], acc[mi][ni], 0, 0, 0);
      }
#else
      asm volatile("" ::"v"(ah[0]), "v"(al[0]), "v"(ah[1]), "v"(al[1]), "v"(bh[ni & 1]), "v"(bl[ni & 1]));
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int kt = 0;
  for (; kt + NRS <= nkt; kt += NRS) {
    step(kt, std::integral_constant<int, 1 % NRS>());
    if (NRS >= 2) step(kt + 1, std::integral_constant<int, 2 % NRS>());
  }
  if (NRS >= 2 && kt < nkt) step(kt, std::integral_constant<int, 1 % NRS>());

  constexpr bool LEAN = EPI != 0;
  if (LEAN) {
    __syncthreads();  // the last stage is still being read by slower waves
    gemm_epilogue_rows_halves_batched<NFN, EPI>(p, mb, acc, reinterpret_cast<float*>(p16_smem), m0, n0, wm, wn, lr, lq, tid, true, false);
  } else if (WGS == 1 && (epi_rows || p.d_p16) && !p.atomic && epi_vec_ok(p)) {
    __syncthreads();
    gemm_epilogue_rows_halves_batched<NFN, 0>(p, mb, acc, reinterpret_cast<float*>(p16_smem), m0, n0, wm, wn, lr, lq, tid, true, false);
  } else if ((epi_rows || p.d_p16) && !p.atomic && epi_vec_ok(p)) {
    __syncthreads();
    gemm_epilogue_rows_halves<NFN>(p, mb, acc, reinterpret_cast<float*>(p16_smem), m0, n0, wm, wn, lr, lq, tid, true, false);
  } else {
    gemm_epilogue_serial<NFN>(p, mb, acc, m0, n0, wm, wn, lr, lq, true, p.atomic != 0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// nt kernel on 256 x 176 x 32 tiles (round 5).  The P16 GEMMs are bound by the L2 -> CU operand stream: a tile of
// TM x TN stages (TM + TN) x 128 B per K-step for 2 TM TN 32 flop, so 256 rows give 1.47x the flops per staged byte of 128.  Same stage
// layout and fragment reads as vptr_gemm_p16_kernel with a 32 KB A region (32 pieces; 4 A + 3 B pieces per wave), two 56 KB stages, ONE
// workgroup per CU (256 registers per lane).  Wave (wm, wn) owns rows s * 128 + wm * 32 + (mi & 1) * 16, s = mi >> 1: the tile is two
// stacked 128-row sub-tiles with the wave layout of the 128-row kernel, so every epilogue of gemm_shared.h runs unchanged, once per sub-tile.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int P16_STAGE256 = 56 * 1024;
template <int EPI>   // EPI as in vptr_gemm_p16_kernel
__global__ __launch_bounds__(GNT, 2) void vptr_gemm_p16_kernel256(const vptr_gemm_desc p, const int epi_rows_) {
  constexpr int NFN = 11, BN = 176, TR = 256, AREG = TR * 128;
  extern __shared__ __attribute__((aligned(1024))) unsigned char p16_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int epi_rows = epi_rows_ & 0xff;
  const int wm = wave & 3, wn = wave >> 2, lr = lane & 15, lq = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles = tiles_n * ((p.M + TR - 1) / TR);
  const int lg = xcd_logical_block();
  const int grp = lg / tiles, tile = lg - grp * tiles;
  const Member mb = member_of(p, p.batch > 1 ? grp : 0);
  const int m0 = (tile / tiles_n) * TR, n0 = (tile % tiles_n) * BN;
  const int nk = (p.K + 31) >> 5;
  const bool ktail = (p.K & 16) != 0;
  const int nseg = p.ksegs > 1 ? p.ksegs : 1;
  const int64_t pa = p.lda * 4, pb = p.ldb * 4;
  const unsigned char* Ab = reinterpret_cast<const unsigned char*>(mb.A);
  const unsigned char* Bb = reinterpret_cast<const unsigned char*>(mb.B);
  const int64_t sA1 = nseg > 1 ? (p.A_x1 - p.A) * 4 : 0, sA2 = nseg > 2 ? (p.A_x2 - p.A) * 4 : 0;
  const int64_t sB1 = nseg > 1 ? (p.B_x1 - p.B) * 4 : 0, sB2 = nseg > 2 ? (p.B_x2 - p.B) * 4 : 0;
  const unsigned char* srcA[4];
  const unsigned char* srcB[3];
  int tadj;
  {
    const int pch = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int prow = (wave + 8 * i) * 8 + (lane >> 3);
      const int c = pch ^ ((prow >> 1) & 7);
      srcA[i] = Ab + (int64_t)min(m0 + prow, p.M - 1) * pa + c * 16;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int prow = (wave + 8 * i) * 8 + (lane >> 3);
      const int c = pch ^ ((prow >> 1) & 7);
      srcB[i] = Bb + (int64_t)min(n0 + prow, p.N - 1) * pb + c * 16;
    }
    const int c = pch ^ (((lane >> 4) + 4 * wave) & 7);
    tadj = c >= 4 ? -64 : 0;
  }
  auto issue = [&](const int kt, const int stage) {
    const int sg = (int)(kt >= nk) + (int)(kt >= 2 * nk);
    const int kk = kt - sg * nk;
    const int64_t off = (int64_t)kk * 128 + ((ktail && kk == nk - 1) ? tadj : 0);
    const int64_t oa = (sg == 0 ? (int64_t)0 : (sg == 1 ? sA1 : sA2)) + off, ob = (sg == 0 ? (int64_t)0 : (sg == 1 ? sB1 : sB2)) + off;
#pragma unroll
    for (int i = 0; i < 4; ++i) P16_GLDS((uint32_t)(stage * P16_STAGE256 + (wave + 8 * i) * 1024), srcA[i] + oa);
#pragma unroll
    for (int i = 0; i < 3; ++i) P16_GLDS((uint32_t)(stage * P16_STAGE256 + AREG + (wave + 8 * i) * 1024), srcB[i] + ob);
  };
  f32x4 acc[2][2][6];   // [sub-tile][row fragment][column fragment]
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 6; ++ni) acc[s2][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int offAh[4], offBh[6];
  const int ch = (lq >> 1) * 4 + (lq & 1);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int r = (mi >> 1) * 128 + wm * 32 + (mi & 1) * 16 + lr, f = (r >> 1) & 7;
    offAh[mi] = r * 128 + ((ch ^ f) << 4);
  }
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int r = (wn * 6 + ni) * 16 + lr, f = (r >> 1) & 7;
    offBh[ni] = AREG + r * 128 + ((ch ^ f) << 4);
  }
  const int nkt = nk * nseg;
  issue(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);
    const unsigned char* st = p16_smem + (kt & 1) * P16_STAGE256;
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      ah[mi] = *reinterpret_cast<const bf16x8*>(st + offAh[mi]);
      al[mi] = *reinterpret_cast<const bf16x8*>(st + (offAh[mi] ^ 32));
    }
    if (ktail) {
      const int sg = (int)(kt >= nk) + (int)(kt >= 2 * nk);
      if (kt - sg * nk == nk - 1 && lq >= 2) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          ah[mi] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
          al[mi] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
    }
    bf16x8 bh[2], bl[2];
    bh[0] = *reinterpret_cast<const bf16x8*>(st + offBh[0]);
    bl[0] = *reinterpret_cast<const bf16x8*>(st + (offBh[0] ^ 32));
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
      if (ni == 5 && wn == 1) break;   // wave-uniform: fragment 11 of the tile does not exist
      if (ni + 1 < 6 && !(ni + 1 == 5 && wn == 1)) {
        bh[(ni + 1) & 1] = *reinterpret_cast<const bf16x8*>(st + offBh[ni + 1]);
        bl[(ni + 1) & 1] = *reinterpret_cast<const bf16x8*>(st + (offBh[ni + 1] ^ 32));
      }
      __builtin_amdgcn_sched_barrier(0);
      // two waves per SIMD: the three passes go round the four row fragments (no back-to-back MFMAs on one accumulator)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[mi >> 1][mi & 1][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh[ni & 1], acc[mi >> 1][mi & 1][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[mi >> 1][mi & 1][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl[ni & 1], acc[mi >> 1][mi & 1][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[mi >> 1][mi & 1][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh[ni & 1], acc[mi >> 1][mi & 1][ni], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  constexpr bool LEAN = EPI != 0;
  float* const sE = reinterpret_cast<float*>(p16_smem);
  const bool vec = LEAN || ((epi_rows || p.d_p16) && !p.atomic && epi_vec_ok(p));   // kernel-uniform
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    if (m0 + s2 * 128 >= p.M) break;   // workgroup-uniform: the second sub-tile of the last row tile may not exist
    __syncthreads();                  // the last stage / the previous sub-tile's LDS tile is still being read by slower waves
    if (vec) gemm_epilogue_rows_halves_batched<NFN, EPI>(p, mb, acc[s2], sE, m0 + s2 * 128, n0, wm, wn, lr, lq, tid, true, false);
    else gemm_epilogue_serial<NFN>(p, mb, acc[s2], m0 + s2 * 128, n0, wm, wn, lr, lq, true, p.atomic != 0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// tn kernel (grouped weight gradients).  Per K-step (32 tokens) and operand the stage holds, for every PAIR of granules of the
// tile, 4 pieces of [8 tokens][128 B]; a piece is laid out as 4 mini-subtiles [8 tokens][16 channels] (g0 hi, g0 lo, g1 hi,
// g1 lo; 256 B each, 32-byte channel rows): DMA lane L fetches chunk (L >> 4) * 2 + (L & 1) of token row (L & 15) >> 1 -- whole
// 128-byte lines on the global side.  ds_read_b64_tr_b16 hands lane (i, q) of a 16-lane group the 4 values of channel i from
// the 4 token rows whose addresses lanes 4j .. 4j+3 of the group supply; read j of lane group q takes token block j ^ (q & 1)
// of piece q (so that the two groups served in one LDS cycle sit in different halves of the banks).  Both operands use the same
// token <-> (lane group, element) map, which is all the MFMA's K index needs.
// T % 32 != 0: the last step's DMA rows are clamped to the last token and the A fragments of tokens >= T are zeroed.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 p16_tr_frag(const unsigned char* st, const int off, const int rb0) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(st + off + rb0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(st + off + (128 - rb0)));
  const s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}

// TAG only names the launch for the profiler: 0 = the end-of-backward launch into the gradient slab (atomic adds), 1 = plain-store launches
// of token-range sub-problems (ops.convt_weight_grads) -- same code, separate rows in rocprofv3's kernel table
// Panel-synchronous scheduling (VPTR_WGRAD_SYNC=S, default 16): the co-resident tiles of one XCD keep within 1.5 blocks of S K-steps of
// each other, so that tiles which share an operand panel find it in the XCD's 4 MB L2 instead of re-fetching it over the fabric (default
// launch: L2 hit rate 35 %, 38 - 48 GB per launch against 8.9 GB of distinct bytes).  A counting barrier in split phases on one 32-bit word
// per XCD: a workgroup ARRIVES (fire-and-forget L2 atomic) when it has finished block b and WAITS half a block later until all n
// participants have arrived for block b -- it cannot arrive for block b + 1 before that, so the cumulative count is exact.  Every wait is a
// BOUNDED spin (a workgroup that times out once stops waiting for the rest of the launch but keeps arriving), so a participant that is
// not resident costs time, never a hang.
struct WgSync {
  int* cnt;        // this XCD's arrival counter (cumulative over the launch; reset by the last workgroup of the XCD to leave)
  int base;        // arrivals of all earlier rounds: round * slots * arrivals_per_tile
  int n;           // participants of this round
  bool live;       // false after a timeout
};

// NW: waves per workgroup.  8 (4 x 2 waves of 32 x 96) is the round 1 - 4 geometry.  4 (2 x 2 waves of 64 x 96, round 5): the same tile and
// stage, but every A fragment a wave reads from LDS feeds 6 column fragments and every B fragment 4 row fragments -- 40 transposing
// reads per 72 MFMAs instead of 32 per 36, i.e. LDS read bytes per MFMA down 1.6x (at 8 waves the reads + the DMA writes of a K-step
// need 1344 LDS cycles per workgroup against 1224 MFMA cycles per SIMD: the LDS, not the matrix pipe, was the binding unit).
// MI: 16-row fragments per wave (tile rows TR = 16 MI NW / 2).  (NW, MI) = (8, 4): 256 x 176 tiles, 56 KB stages, ONE workgroup per CU --
// 1.47x the flops per staged byte of the 128-row tile (the elimination builds and the 4-wave A/B both say the launch is bound by what
// the CU can ingest through the vector-memory path, not by LDS reads or the matrix pipe).
// RS = 1 (round 6, VPTR_WGRAD_RS): the operand pieces travel global -> registers -> LDS (global_load_dwordx4 one K-step ahead of their
// ds_write_b128, same lane <-> chunk map as the DMA pieces, interleaved between the MFMA groups) instead of global_load_lds: see
// vptr_gemm_p16_rs_kernel.  Two LDS stages, one barrier per K-step.
template <int NSTAGE, int SYNC, int NW = 8, int MI = 16 / NW, int RS = 0>   // SYNC: 0 = none, else the block length S (a power of two) of the panel-synchronous schedule
__device__ __forceinline__ void wgrad_p16_tile(const vptr_gemm_desc& p, const int tile, unsigned char* p16_smem, WgSync& sy) {
  static_assert(RS == 0 || NSTAGE == 2, "the register-staged loop has two LDS stages");
  constexpr int BN = 176, WM = NW / 2, TR = 16 * MI * WM;   // MI row fragments per wave, WM wave rows, TR tile rows
  constexpr int PA = TR / 8 / NW, PB = 24 / NW;             // DMA pieces per wave and K-step: A, B
  constexpr int AREG = TR * 128, STG = AREG + 24 * 1024;    // bytes of the A region of a stage / of a stage
  const int NG = p.M, KX = p.N, T = p.K;   // D[NG][KX] += alpha * G[T][NG]^T . X[T][KX]
  const int tiles_n = (KX + BN - 1) / BN;
  const int m0 = (tile / tiles_n) * TR, n0 = (tile % tiles_n) * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM, lr = lane & 15, lq = lane >> 4;   // see vptr_gemm_p16_kernel
  const int nk = (T + 31) >> 5;
  const int64_t pg = p.lda * 4, px = p.ldb * 4;
  const unsigned char* Gb = reinterpret_cast<const unsigned char*>(p.A);
  const unsigned char* Xb = reinterpret_cast<const unsigned char*>(p.B);

  // DMA pieces: u = wave + NW i; A pieces u < 16: granule pair u >> 2, token block u & 3; B pieces v = u - 16 likewise
  int colA[PA], colB[PB], trow[PA + PB];
  {
    const int ms = lane >> 4, half = lane & 1, t = (lane & 15) >> 1;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int u = wave + NW * i;
      const int gran = min((m0 >> 4) + (u >> 2) * 2 + (ms >> 1), (NG >> 4) - 1);
      colA[i] = gran * 64 + (ms & 1) * 32 + half * 16;
      trow[i] = (u & 3) * 8 + t;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int v = wave + NW * i;
      const int gran = min((n0 >> 4) + (v >> 2) * 2 + (ms >> 1), (KX >> 4) - 1);
      colB[i] = gran * 64 + (ms & 1) * 32 + half * 16;
      trow[PA + i] = (v & 3) * 8 + t;
    }
  }
  auto issue = [&](const int kt, const int stage) {
    const int t0 = kt * 32;
#pragma unroll
    for (int i = 0; i < PA; ++i)
      P16_GLDS((uint32_t)(stage * STG + (wave + NW * i) * 1024), Gb + (int64_t)min(t0 + trow[i], T - 1) * pg + colA[i]);
#pragma unroll
    for (int i = 0; i < PB; ++i)
      P16_GLDS((uint32_t)(stage * STG + AREG + (wave + NW * i) * 1024), Xb + (int64_t)min(t0 + trow[PA + i], T - 1) * px + colB[i]);
  };
  // register-staged variant: this lane's chunk of piece j (j < PA: A, else B) of K-step kt (clamped to the last: a redundant L2-resident
  // fetch keeps the number of loads per step constant, so the compiler's vmcnt bookkeeping stays static)
  auto rs_load = [&](const int kt_, const int j) -> u32x4 {
    const int t0 = min(kt_, ((T + 31) >> 5) - 1) * 32;
    const unsigned char* a = j < PA ? Gb + (int64_t)min(t0 + trow[j], T - 1) * pg + colA[j < PA ? j : 0]
                                    : Xb + (int64_t)min(t0 + trow[j], T - 1) * px + colB[j < PA ? 0 : j - PA];
    return *reinterpret_cast<const u32x4*>(a);
  };
  auto rs_put = [&](const int stage, const int j, const u32x4 v) {
    const int off = j < PA ? (wave + NW * j) * 1024 : AREG + (wave + NW * (j - PA)) * 1024;
    *reinterpret_cast<u32x4*>(p16_smem + stage * STG + off + lane * 16) = v;
  };
  u32x4 rg[RS ? PA + PB : 1];

  f32x4 acc[MI][6];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // fragment f (granule f of the operand's tile), plane pl: piece (f >> 1) * 4 + lq, mini-subtile (f & 1) * 2 + pl
  const int lane_off = lq * 1024 + (lr >> 2) * 32 + (lr & 3) * 8;
  const int rb0 = (lq & 1) * 128;
  int offA[MI], offB[6];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int f = wm * MI + mi;
    offA[mi] = (f >> 1) * 4096 + (f & 1) * 512 + lane_off;
  }
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int f = wn * 6 + ni;
    offB[ni] = AREG + (f >> 1) * 4096 + (f & 1) * 512 + lane_off;
  }
  // bias gradient: column tile 0 only, odd wave column, its 6th (otherwise idle) fragment multiplies by ones: acc[mi][5][r] =
  // sum_t G[t][row] for every column of the fragment
  const bool flip = p.d_transposed != 0;   // D stored transposed; a_rowsum = column sums of B, taken by a wave row beyond M (see vptr_hip.h)
  const bool want_rowsum = !flip && p.a_rowsum != nullptr && n0 == 0 && wn == 1;   // wave-uniform
  // column sums of B (flipped problems): the first 16-row FRAGMENT of the tile that lies entirely beyond M multiplies by ones instead
  const int f0 = (NG - m0 + 15) >> 4;   // (the host admits a flipped problem with a bias only when at least 32 tile rows are free)
  const bool colsum_wave = flip && p.a_rowsum != nullptr && f0 < TR / 16 && wm == f0 / MI;   // wave-uniform
  const int cmi = f0 - wm * MI;          // that fragment's index among this wave's
  const __bf16 one = (__bf16)1.0f, zero = (__bf16)0.0f;
  const bf16x8 ones = {one, one, one, one, one, one, one, one};
  const bf16x8 zeros = {zero, zero, zero, zero, zero, zero, zero, zero};
  const bool ttail = (T & 31) != 0;
  const bool rows_live = m0 + wm * (16 * MI) < NG;

  if (RS) {
    u32x4 t0[PA + PB];
#pragma unroll
    for (int j = 0; j < PA + PB; ++j) t0[j] = rs_load(0, j);
#pragma unroll
    for (int j = 0; j < PA + PB; ++j) rg[j] = rs_load(1, j);
#pragma unroll
    for (int j = 0; j < PA + PB; ++j) rs_put(0, j, t0[j]);
  } else {
    issue(0, 0);
    if (NSTAGE >= 3 && nk > 1) issue(1, 1);
    if (NSTAGE >= 4 && nk > 2) issue(2, 2);
  }
  for (int kt = 0; kt < nk; ++kt) {
    if (SYNC > 0 && kt > 0 && (kt & (SYNC / 2 - 1)) == 0 && threadIdx.x == 0) {   // wave 0 reaches this step's barrier late if it has to wait: the other waves wait there
      const int ph = kt & (SYNC - 1);
      if (ph == 0) __hip_atomic_fetch_add(sy.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // finished block kt / S - 1
      else if (kt > SYNC && sy.live) {
        const int target = sy.base + (kt / SYNC) * sy.n;
        int spins = 0;
        while (__hip_atomic_load(sy.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (++spins > 3000) {   // ~1.5 ms: somebody is not resident -- go on unsynchronised (counted: tools/wgrad_sync_probe.py reads the word)
            sy.live = false;
            __hip_atomic_fetch_add(sy.cnt + 48, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
      }
    }
    if (RS) {
      // stage kt & 1 is complete once every wave's stores of step kt have landed (lgkmcnt(0) precedes the barrier); stage (kt + 1) & 1
      // was last read in step kt - 1, i.e. before this barrier
    } else if (NSTAGE >= 4 && kt + 2 < nk) __builtin_amdgcn_s_waitcnt(0x0f70 | (2 * (PA + PB)));
    else if (NSTAGE >= 3 && kt + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0f70 | (PA + PB));   // vmcnt(pieces of one step): step kt landed, step kt + 1 may still fly
    else __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (RS) {
    } else if (NSTAGE >= 3) {
      if (kt + NSTAGE - 1 < nk) issue(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
    } else if (kt + 1 < nk) {
      issue(kt + 1, (kt + 1) & 1);
    }
#ifdef VPTR_TN_DMA_ONLY   // elimination build (WRONG results): staging, waits and barriers only -- what the launch costs when the CUs do nothing
    if (!RS) continue;     // but ingest their operand tiles (tools/build_variant.sh dmaonly -DVPTR_TN_DMA_ONLY; profiles/r05_ingest_roofline.log)
#endif
    // wave-uniform: this wave's 32 rows lie beyond NG (the last row tile of a 528-row problem keeps 16 of 128).  Not under RS: the wave has
    // its share of the operand pieces to move, and a second code path with loads of its own costs the compiler its static vmcnt bookkeeping
    // (it then drains vmcnt(0) before every step's first LDS store); the dead rows' products are masked by the epilogue
    if (!RS && !rows_live && !colsum_wave) continue;
    const unsigned char* st = p16_smem + (NSTAGE >= 3 ? kt % NSTAGE : (kt & 1)) * STG;
    bf16x8 ah[MI], al[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      ah[mi] = p16_tr_frag(st, offA[mi], rb0);
      al[mi] = p16_tr_frag(st, offA[mi] + 256, rb0);
    }
    if (colsum_wave) {   // a fragment of rows beyond M: ONES instead -- its accumulator rows all become sum_t B[t][n]
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        if (mi == cmi) { ah[mi] = ones; al[mi] = zeros; }
    }
    if (ttail && kt == nk - 1) {   // workgroup-uniform: zero the A values of tokens beyond T
      const int tv = T - kt * 32;  // valid tokens of this step
      // element e of this lane: token 8 lq + 4 (j ^ (lq & 1)) + (e & 3), j = e >> 2
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int tok = 8 * lq + 4 * ((e >> 2) ^ (lq & 1)) + (e & 3);
        if (tok >= tv) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) { ah[mi][e] = zero; al[mi][e] = zero; }
        }
      }
    }
    // the next B fragment pair is requested before the MFMAs of the current one (see vptr_gemm_p16_kernel)
    bf16x8 bh[2], bl[2];
    bh[0] = p16_tr_frag(st, offB[0], rb0);
    bl[0] = p16_tr_frag(st, offB[0] + 256, rb0);
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
      if (ni == 5 && wn == 1 && !want_rowsum) break;   // wave-uniform: the padding fragment only works for the bias gradient
      if (ni + 1 < 6) {
        if (ni + 1 == 5 && wn == 1) {   // wave-uniform: fragment 11 does not exist; ones for the bias gradient (unused otherwise)
          bh[(ni + 1) & 1] = ones;
          bl[(ni + 1) & 1] = zeros;
        } else {
          bh[(ni + 1) & 1] = p16_tr_frag(st, offB[ni + 1], rb0);
          bl[(ni + 1) & 1] = p16_tr_frag(st, offB[ni + 1] + 256, rb0);
        }
      }
      if (RS && ni < 5) {   // pieces ni, ni + 5, ... of step kt + 1: registers -> the other stage; the registers then take step kt + 2
#pragma unroll
        for (int j = ni; j < PA + PB; j += 5) {
          rs_put((kt + 1) & 1, j, rg[j]);
          rg[j] = rs_load(kt + 2, j);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (NW == 8) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh[ni & 1], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl[ni & 1], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh[ni & 1], acc[mi][ni], 0, 0, 0);
        }
      } else {   // two waves per SIMD: nobody else fills the pipe behind a dependent accumulate, so the passes go round the row fragments
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh[ni & 1], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl[ni & 1], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh[ni & 1], acc[mi][ni], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // epilogue: D += alpha * acc (fp32 atomics into the gradient slab: the same weight may receive several contributions).  Cost, measured
  // in round 4 by returning here instead (bare launch of the K64 step's 196 problems): 7.10 -> 6.85 ms, i.e. 3.6 % of the launch for 124 M
  // scalar atomics; a row-major / float4 read-add-write for single-writer destinations could recover part of that
  const float alpha = p.alpha;
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int nf = wn * 6 + ni, col = n0 + nf * 16 + lr;
    if (nf < 11 && col < KX) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + (wm * MI + mi) * 16 + lq * 4 + r;
          if (row < NG) {
            float* dst = flip ? p.D + (int64_t)col * p.ldd + row : p.D + (int64_t)row * p.ldd + col;
            if (p.atomic) unsafeAtomicAdd(dst, acc[mi][ni][r] * alpha);
            else *dst = acc[mi][ni][r] * alpha;
          }
        }
    }
  }
  if (colsum_wave && lq == 0) {   // row 0 of the ones-fragment product: lane lr holds the column sum of column lr of every fragment
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
      const int nf = wn * 6 + ni, col = n0 + nf * 16 + lr;
      if (nf < 11 && col < KX) {
        float cs = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          if (mi == cmi) cs = acc[mi][ni][0];
        unsafeAtomicAdd(p.a_rowsum + col, cs * alpha);
      }
    }
  }
  if (want_rowsum && lr == 0) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + (wm * MI + mi) * 16 + lq * 4 + r;
        if (row < NG) unsafeAtomicAdd(p.a_rowsum + row, acc[mi][5][r] * alpha);
      }
  }
}


template <int NSTAGE, int TAG = 0, int NW = 8, int MI = 16 / NW, int RS = 0>   // 2: two workgroups per CU; 3: one workgroup per CU with the DMA two K-steps ahead (experiment, VPTR_WGRAD_STAGES=3)
__global__ __launch_bounds__(64 * NW, NSTAGE == 2 ? (NW * MI == 16 && NW == 8 && !RS ? 4 : 2) : 2) void vptr_wgrad_p16_kernel(const vptr_gemm_desc* __restrict__ descs, const int* __restrict__ tile_start,
                                                                const int count, const int xmode, const int tile_base) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char p16_smem[];
  const int lg = tile_base + ((xmode & 0xff) == 1 ? (int)blockIdx.x : xcd_logical_block());
  if ((xmode & 0x100) && (threadIdx.x >> 6) >= 4) __builtin_amdgcn_s_setprio(1);   // experiment: see vptr_gemm_p16_kernel
  int lo = 0, hi = count - 1;  // last g with tile_start[g] <= lg (workgroup-uniform scalar search)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_start[mid] <= lg) lo = mid;
    else hi = mid - 1;
  }
  WgSync none = {nullptr, 0, 0, false};
  wgrad_p16_tile<NSTAGE, 0, NW, MI, RS>(descs[lo], lg - tile_start[lo], p16_smem, none);
}

// Persistent form for the panel-synchronous schedule: gridDim.x = 8 * slots workgroups (two per CU), workgroup b serves XCD b & 7 as its
// slot b >> 3; XCD x owns the same contiguous range of the logical tile order as in the plain launch and walks it in ROUNDS of `slots`
// tiles.  Requires every problem of the launch to have the same token count (the caller vouches: vptr_gemm_desc.split_k = -S on the
// prototype).  g_wgrad_sync_ws: 64 ints per XCD (counter at [x * 64], leave counter at [x * 64 + 32]); the kernel leaves them zero.
__device__ int g_wgrad_sync_ws[8 * 64];   // module-scope, zero at load; one launch of the kernel at a time (launches on ONE stream serialise)
template <int S, int NW = 8, int MI = 16 / NW, int NST = 2, int RS = 0>
__global__ __launch_bounds__(64 * NW, NW * MI == 16 && NW == 8 && !RS ? 4 : 2) void vptr_wgrad_p16_sync_kernel(const vptr_gemm_desc* __restrict__ descs, const int* __restrict__ tile_start,
                                                                    const int count, const int total_tiles) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char p16_smem[];
  int* const ws = g_wgrad_sync_ws;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int xq = total_tiles >> 3, xr = total_tiles & 7;
  const int first = xcd * xq + min(xcd, xr), mine = xq + (xcd < xr ? 1 : 0);   // this XCD's tiles: [first, first + mine)
  const int nk = (descs[0].K + 31) >> 5;
  const int per_tile = (nk - 1) / S;   // arrivals per tile (steps S, 2 S, ... < nk)
  WgSync sy = {ws + xcd * 64, 0, 0, true};
  for (int r = 0; r * slots + slot < mine; ++r) {
    const int lg = first + r * slots + slot;
    int lo = 0, hi = count - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tile_start[mid] <= lg) lo = mid;
      else hi = mid - 1;
    }
    sy.base = r * slots * per_tile;
    sy.n = min(slots, mine - r * slots);
    if (r > 0) __syncthreads();   // the previous tile's last stage is still being read by slower waves
    wgrad_p16_tile<NST, S, NW, MI, RS>(descs[lo], lg - tile_start[lo], p16_smem, sy);
  }
  if (threadIdx.x == 0) {   // the last workgroup of this XCD to leave puts the two words back to zero for the next launch
    int* done = ws + xcd * 64 + 32;
    if (__hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == slots - 1) {
      __hip_atomic_store(ws + xcd * 64, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ void wgrad_sync_stats_kernel(int* __restrict__ out) {
  if (threadIdx.x < 8) out[threadIdx.x] = g_wgrad_sync_ws[threadIdx.x * 64 + 48];
}
// telemetry: out_dev[8] (device ints) = how often a workgroup of XCD x gave up waiting in the panel-synchronous weight-gradient launches
// since the library was loaded (0 everywhere = every participant was always resident)
extern "C" int vptr_wgrad_sync_stats(int* out_dev, vptr_stream_t stream) {
  VPTR_CHECK(out_dev != nullptr, "wgrad_sync_stats: null output");
  wgrad_sync_stats_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out_dev);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static int vptr_cu_count() {   // compute units of the current device (256 on MI355X); 0 if the query fails (then no grid counts as "lone")
  static int n = -1;
  if (n < 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) n = v;
    else n = 0;
  }
  return n;
}

static int p16_prio_flag() {   // VPTR_GEMM_PRIO=1: bit 8 of the kernels' mode argument
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VPTR_GEMM_PRIO");
    v = (e && atoi(e) != 0) ? 0x100 : 0;
  }
  return v;
}

static bool p16_no_epi3() {   // VPTR_GEMM_NO_EPI3 (A/B switch), read once
  static int v = -1;
  if (v < 0) v = getenv("VPTR_GEMM_NO_EPI3") != nullptr;
  return v != 0;
}

static bool p16_no_epi4() {   // VPTR_GEMM_NO_EPI4 (A/B switch), read once
  static int v = -1;
  if (v < 0) v = getenv("VPTR_GEMM_NO_EPI4") != nullptr;
  return v != 0;
}

static int p16_epi_rows_flag() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VPTR_GEMM_EPI_ROWS");
    v = e ? atoi(e) : 3;
  }
  return v;
}

int vptr_gemm_p16_launch(vptr_gemm_desc& d, hipStream_t st) {
  VPTR_CHECK(d.a_mode == VPTR_A_P16 && d.b_mode == VPTR_B_P16, "vptr_gemm(p16): both operands must be P16 (a_mode %d, b_mode %d)", d.a_mode, d.b_mode);
  VPTR_CHECK(d.K % 16 == 0 && d.lda % 16 == 0 && d.ldb % 16 == 0, "vptr_gemm(p16): K, lda, ldb must be multiples of 16 (K %d)", d.K);
  VPTR_CHECK(d.split_k <= 1 && d.precision == 3 && !d.a_rowsum && !d.D_planes, "vptr_gemm(p16): split_k = 1, precision 3, no a_rowsum / D_planes");
  if (d.alpha == 0.f) d.alpha = 1.f;
  if (d.batch < 1) d.batch = 1;
  if (d.ksegs < 1) d.ksegs = 1;
  const bool strided = d.batch_stride_d != 0;
  if (strided)   // ABI 10: any number of members at constant strides, plain epilogue (shared bias / alpha)
    VPTR_CHECK(d.ksegs == 1 && d.batch <= 4096 && !d.Dpre && !d.residual && !d.atomic && !d.batch_accum && !d.frame_stats && !d.act_grad_src && !d.rowscale &&
                   !d.colscale && d.dropout_p == 0.f && d.act == VPTR_ACT_NONE && !d.act_after &&
                   ((d.batch_stride_a | d.batch_stride_b | d.batch_stride_d) & 15) == 0 && d.batch_stride_a >= 0 && d.batch_stride_b >= 0 && d.batch_stride_d > 0,
               "vptr_gemm(p16): a strided batch takes bias / alpha only and strides that are non-negative multiples of 16");
  VPTR_CHECK((strided || d.batch <= 3) && d.ksegs <= 3 && (d.batch == 1 || d.ksegs == 1), "vptr_gemm(p16): at most 3 batch members or 3 K segments");
  uintptr_t bits = reinterpret_cast<uintptr_t>(d.A) | reinterpret_cast<uintptr_t>(d.B);
  const int extra = strided ? 0 : (d.batch > 1 ? d.batch : d.ksegs) - 1;
  if (extra >= 1) {
    VPTR_CHECK(d.A_x1 && d.B_x1, "vptr_gemm(p16): member / segment 1 needs A_x1, B_x1");
    bits |= reinterpret_cast<uintptr_t>(d.A_x1) | reinterpret_cast<uintptr_t>(d.B_x1);
  }
  if (extra >= 2) {
    VPTR_CHECK(d.A_x2 && d.B_x2, "vptr_gemm(p16): member / segment 2 needs A_x2, B_x2");
    bits |= reinterpret_cast<uintptr_t>(d.A_x2) | reinterpret_cast<uintptr_t>(d.B_x2);
  }
  VPTR_CHECK((bits & 63) == 0, "vptr_gemm(p16): operands must be 64-byte aligned (whole granules)");
  if (d.batch > 1 && !strided) {
    VPTR_CHECK(d.D_x1 && (d.batch < 3 || d.D_x2) && !d.Dpre && !d.residual && !d.atomic, "vptr_gemm(p16): bad batch members");
    if (d.alpha_x1 == 0.f) d.alpha_x1 = 1.f;
    if (d.alpha_x2 == 0.f) d.alpha_x2 = 1.f;
  }
  if (d.rowscale) VPTR_CHECK(d.rs_div >= 1 && d.rs_mod >= 1, "vptr_gemm: rowscale needs rs_div, rs_mod >= 1");
  if (d.dropout_p > 0.f) VPTR_CHECK(d.seed_dev != nullptr && d.dropout_p < 1.f, "vptr_gemm: dropout needs seed_dev and p < 1");
  if (d.d_p16)
    VPTR_CHECK(!d.atomic && d.N % 16 == 0 && d.ldd % 16 == 0 && (d.ldr & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(d.D) | reinterpret_cast<uintptr_t>(d.D_x1) | reinterpret_cast<uintptr_t>(d.D_x2)) & 63) == 0 &&
                   ((reinterpret_cast<uintptr_t>(d.residual) | reinterpret_cast<uintptr_t>(d.bias) | reinterpret_cast<uintptr_t>(d.colscale) |
                     reinterpret_cast<uintptr_t>(d.Dpre) | reinterpret_cast<uintptr_t>(d.bias_x1) | reinterpret_cast<uintptr_t>(d.bias_x2)) & 15) == 0,
               "vptr_gemm(p16): a P16 output needs N, ldd multiples of 16, 64-byte aligned D and 16-byte aligned epilogue operands, no atomics");
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel<0, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel<1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel<3, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel<0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel<3, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * P16_STAGE) != hipSuccess) {
      vptr_set_error("vptr_gemm(p16): cannot reserve %d bytes of LDS", 2 * P16_STAGE);
      return -1;
    }
    attr_set = true;
  }
  const int tiles = ((d.M + GBM - 1) / GBM) * ((d.N + 175) / 176) * d.batch;
  // the plain launches (bias / alpha / residual, fp32 or P16 output, vector-aligned) take the lean instantiation
  uintptr_t ebits = reinterpret_cast<uintptr_t>(d.D) | reinterpret_cast<uintptr_t>(d.residual) | reinterpret_cast<uintptr_t>(d.bias);
  if (d.batch > 1 && !strided) ebits |= reinterpret_cast<uintptr_t>(d.D_x1) | reinterpret_cast<uintptr_t>(d.D_x2) | reinterpret_cast<uintptr_t>(d.bias_x1) | reinterpret_cast<uintptr_t>(d.bias_x2);
  #ifdef VPTR_P16_TIMING
  const bool dpre_ok = true;
#else
  const bool dpre_ok = !d.Dpre;
#endif
  const bool lean = (p16_epi_rows_flag() & 4) == 0 && !d.colscale && dpre_ok && !d.rowscale && d.act == VPTR_ACT_NONE && d.dropout_p == 0.f && !d.act_after && !d.atomic &&
                    (ebits & 15) == 0 && (d.N & 3) == 0 && (d.ldd & 3) == 0 && (d.ldr & 3) == 0;
  // the same plus a DropPath row scale and / or dropout (out-projections and linear2 of every block: 46 launches of the K64 step)
  const bool lean3 = !lean && (p16_epi_rows_flag() & 4) == 0 && !d.colscale && dpre_ok && (d.rowscale || d.dropout_p > 0.f) && d.act == VPTR_ACT_NONE &&
                     !d.act_after && !d.atomic && !d.frame_stats && (ebits & 15) == 0 && (d.N & 3) == 0 && (d.ldd & 3) == 0 && (d.ldr & 3) == 0 &&
                     !p16_no_epi3();
  // activation (+ saved pre-activation, dropout, P16 output) and nothing else: linear1 of the MLP blocks -- the full epilogue's ~20 k
  // instructions of skipped branches cost these launches a quarter of their time (213 vs 290 TFLOP/s at 29 696 x 2112 x 528)
  const bool lean4 = !lean && !lean3 && (p16_epi_rows_flag() & 4) == 0 && !d.colscale && !d.rowscale && !d.residual && !d.act_after && !d.atomic &&
                     d.act != VPTR_ACT_NONE && !d.act_grad_src && !d.frame_stats && d.batch == 1 && !d.batch_accum &&
                     ((ebits | reinterpret_cast<uintptr_t>(d.Dpre)) & 15) == 0 && (d.N & 3) == 0 && (d.ldd & 3) == 0 && !p16_no_epi4();
  static int lone_stages = 0, force_lone = 0;
  if (!lone_stages) {
    const char* e = getenv("VPTR_GEMM_LONE_STAGES");
    lone_stages = (e && atoi(e) == 3) ? 3 : 4;   // default since round 4: four stages (all 160 KB), the DMA three K-steps ahead
    const char* f = getenv("VPTR_GEMM_FORCE_LONE");   // experiment: the one-workgroup-per-CU instantiation for every grid
    force_lone = f ? atoi(f) : 0;
  }
  const bool lone = (tiles <= vptr_cu_count() || force_lone) && (p16_epi_rows_flag() & 16) == 0;   // at most one workgroup per CU
  const bool lone4 = lone && lone_stages == 4;
  const int rows = lean ? 1 : (p16_epi_rows_flag() & 2);
  if (d.frame_stats)   // served by the lean epilogue only: no fallback
    VPTR_CHECK(d.frame_rows >= 64 && d.frame_rows % 64 == 0 && d.M % 64 == 0 && !d.act_grad_src && lean && d.batch == 1,
               "vptr_gemm(p16): frame_stats needs frame_rows %% 64 == 0, M %% 64 == 0 and a plain launch (bias / alpha / residual only)");
  // 256 x 176 tiles (vptr_gemm_p16_kernel256: 1.47x the flops per staged byte, one workgroup per CU) where they fill the chip at least as
  // well as 128-row tiles fill it with two workgroups per CU: whole-round efficiency x 1.17 (the gain measured on the tn side).
  // VPTR_GEMM_ROWS=model enables that rule, =256 forces them for every grid of more than one round; default: off (round-5 A/B: with K loops of
  // 17 - 66 steps the lone workgroup's prologue and four half-tile epilogue passes cost more than the operand stream saves).
  bool use256 = false;
  {
    static int rows_mode = -1;
    if (rows_mode < 0) {
      const char* e = getenv("VPTR_GEMM_ROWS");
      rows_mode = e ? (atoi(e) == 256 ? 2 : (e[0] == 'm' ? 1 : 0)) : 0;   // default OFF: measured slower at the model's K (17 / 66 K-steps), tools/rejected/README.md
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel256<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE256) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel256<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE256) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_kernel256<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE256) != hipSuccess)
        rows_mode = 0;
    }
    const int cus = vptr_cu_count();
    const int t256 = ((d.M + 255) / 256) * ((d.N + 175) / 176) * d.batch;
    // (lean / lean3 / activation-gradient epilogues only: the full epilogue next to 96 accumulator registers spills)
    if (rows_mode > 0 && cus > 0 && d.M >= 512 && t256 > cus && (lean || lean3 || d.act_grad_src)) {
      const double e256 = 1.17 * t256 / (double)(((t256 + cus - 1) / cus) * cus);
      const double e128 = tiles / (double)(((tiles + 2 * cus - 1) / (2 * cus)) * 2 * cus);
      use256 = rows_mode == 2 || e256 > 1.03 * e128;
    }
    // strided batches (the 36 Winograd-domain products of a frozen 3 x 3 convolution): VPTR_WINO_ROWS=256 puts them on the 256-row tiles
    static int wino_rows = -1;
    if (wino_rows < 0) { const char* e = getenv("VPTR_WINO_ROWS"); wino_rows = (e && atoi(e) == 256 && rows_mode >= 0) ? 256 : 128; }
    if (strided && wino_rows == 256 && lean && d.M >= 256) use256 = true;
  }
  if (d.act_grad_src) {   // activation-gradient epilogue: its own instantiation, no fallback
    VPTR_CHECK(!d.colscale && !d.Dpre && !d.rowscale && !d.residual && !d.bias && !d.act_after && !d.atomic && d.batch == 1 && d.ksegs == 1 &&
                   d.act != VPTR_ACT_NONE && ((ebits | reinterpret_cast<uintptr_t>(d.act_grad_src)) & 15) == 0 && (d.N & 3) == 0 && (d.ldd & 3) == 0,
               "vptr_gemm(p16): act_grad_src combines with alpha / dropout / P16 output only and needs 16-byte aligned operands, N, ldd multiples of 4");
    if (use256) vptr_gemm_p16_kernel256<2><<<((d.M + 255) / 256) * ((d.N + 175) / 176), GNT, 2 * P16_STAGE256, st>>>(d, 1);
    else if (lone4) vptr_gemm_p16_kernel<2, 4><<<tiles, GNT, 4 * P16_STAGE, st>>>(d, 1 | p16_prio_flag());
    else if (lone) vptr_gemm_p16_kernel<2, 3><<<tiles, GNT, 3 * P16_STAGE, st>>>(d, 1 | p16_prio_flag());
    else vptr_gemm_p16_kernel<2, 2><<<tiles, GNT, 2 * P16_STAGE, st>>>(d, 1 | p16_prio_flag());
    return 0;
  }
  if (d.batch_accum)
    VPTR_CHECK(lean && d.batch > 1 && !d.d_p16 && (d.batch_accum >> d.batch) == 0, "vptr_gemm(p16): batch_accum is an option of plain fp32-output batch launches");
  if (use256) {
    const int t256 = ((d.M + 255) / 256) * ((d.N + 175) / 176) * d.batch;
    if (lean3) vptr_gemm_p16_kernel256<3><<<t256, GNT, 2 * P16_STAGE256, st>>>(d, 1);
    else vptr_gemm_p16_kernel256<1><<<t256, GNT, 2 * P16_STAGE256, st>>>(d, rows);
    return 0;
  }
  // register-staged operand path (round 6): VPTR_GEMM_RS = 0 off, 1 grids of at most one workgroup per CU, 2 every grid;
  // VPTR_GEMM_RS_SETS = 1 | 2 register sets for the lone grids
  static int rs_mode = -1, rs_sets = 2;
  if (rs_mode < 0) {
    const char* e = getenv("VPTR_GEMM_RS");
    rs_mode = e ? atoi(e) : 0;
    const char* f = getenv("VPTR_GEMM_RS_SETS");
    rs_sets = (f && atoi(f) == 1) ? 1 : 2;
    bool ok = true;
#define RS_ATTR(E, R, W) ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_gemm_p16_rs_kernel<E, R, W>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) == hipSuccess
    RS_ATTR(0, 1, 1); RS_ATTR(1, 1, 1); RS_ATTR(3, 1, 1); RS_ATTR(4, 1, 1);
    RS_ATTR(0, 2, 1); RS_ATTR(1, 2, 1); RS_ATTR(3, 2, 1); RS_ATTR(4, 2, 1);
    RS_ATTR(0, 1, 2); RS_ATTR(1, 1, 2); RS_ATTR(3, 1, 2); RS_ATTR(4, 1, 2);
#undef RS_ATTR
    if (!ok) rs_mode = 0;
  }
  if ((rs_mode == 2 || (rs_mode == 1 && lone)) && !(lean4 && lone && !lone4)) {
    const int e = lean4 ? 4 : (lean3 ? 3 : (lean ? 1 : 0));
    const int arg = (e == 1 || e == 0 ? rows : 1) | p16_prio_flag();
#define RS_GO(E, R, W) vptr_gemm_p16_rs_kernel<E, R, W><<<tiles, GNT, 2 * P16_STAGE, st>>>(d, arg)
    if (lone && rs_sets == 2) { if (e == 4) RS_GO(4, 2, 1); else if (e == 3) RS_GO(3, 2, 1); else if (e == 1) RS_GO(1, 2, 1); else RS_GO(0, 2, 1); }
    else if (lone)            { if (e == 4) RS_GO(4, 1, 1); else if (e == 3) RS_GO(3, 1, 1); else if (e == 1) RS_GO(1, 1, 1); else RS_GO(0, 1, 1); }
    else                      { if (e == 4) RS_GO(4, 1, 2); else if (e == 3) RS_GO(3, 1, 2); else if (e == 1) RS_GO(1, 1, 2); else RS_GO(0, 1, 2); }
#undef RS_GO
    return 0;
  }
  if (lean4 && lone4) vptr_gemm_p16_kernel<4, 4><<<tiles, GNT, 4 * P16_STAGE, st>>>(d, 1 | p16_prio_flag());
  else if (lean4 && !lone) vptr_gemm_p16_kernel<4, 2><<<tiles, GNT, 2 * P16_STAGE, st>>>(d, 1 | p16_prio_flag());
  else if (lean3 && lone4) vptr_gemm_p16_kernel<3, 4><<<tiles, GNT, 4 * P16_STAGE, st>>>(d, 1 | p16_prio_flag());
  else if (lean && lone4) vptr_gemm_p16_kernel<1, 4><<<tiles, GNT, 4 * P16_STAGE, st>>>(d, rows | p16_prio_flag());
  else if (!lean3 && !lean && lone4) vptr_gemm_p16_kernel<0, 4><<<tiles, GNT, 4 * P16_STAGE, st>>>(d, rows | p16_prio_flag());
  else if (lean3 && lone) vptr_gemm_p16_kernel<3, 3><<<tiles, GNT, 3 * P16_STAGE, st>>>(d, 1 | p16_prio_flag());
  else if (lean3) vptr_gemm_p16_kernel<3, 2><<<tiles, GNT, 2 * P16_STAGE, st>>>(d, 1 | p16_prio_flag());
  else if (lean && lone) vptr_gemm_p16_kernel<1, 3><<<tiles, GNT, 3 * P16_STAGE, st>>>(d, rows | p16_prio_flag());
  else if (lean) vptr_gemm_p16_kernel<1, 2><<<tiles, GNT, 2 * P16_STAGE, st>>>(d, rows | p16_prio_flag());
  else if (lone) vptr_gemm_p16_kernel<0, 3><<<tiles, GNT, 3 * P16_STAGE, st>>>(d, rows | p16_prio_flag());
  else vptr_gemm_p16_kernel<0, 2><<<tiles, GNT, 2 * P16_STAGE, st>>>(d, rows | p16_prio_flag());
  return 0;
}

int vptr_wgrad_p16_launch(const vptr_gemm_desc* proto, const vptr_gemm_desc* descs_dev, const int* tile_start_dev, int count, int total_tiles,
                          hipStream_t st) {
  VPTR_CHECK(proto->b_mode == VPTR_B_P16T && proto->precision == 3, "vptr_gemm_grouped(p16): both operands token-major P16, precision 3");
  static int stages = -1, xmode = 0, gen = 0;
  if (stages < 0) {
    const char* xm = getenv("VPTR_WGRAD_XCD");
    xmode = xm ? atoi(xm) : 0;
    const char* ge = getenv("VPTR_WGRAD_GEN");
    gen = ge ? atoi(ge) : 0;
    const char* e = getenv("VPTR_WGRAD_STAGES");
    stages = (e && (atoi(e) == 3 || atoi(e) == 4)) ? atoi(e) : 2;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * P16_STAGE) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * P16_STAGE) != hipSuccess) {
      vptr_set_error("vptr_gemm_grouped(p16): cannot reserve LDS");
      stages = -1;
      return -1;
    }
  }
  // panel-synchronous persistent launch (default since round 4; VPTR_WGRAD_SYNC=0 restores the plain one) for prototypes whose split_k is -1
  // (the host vouches that all problems share one token count)
  static int sync_s = -1;
  if (sync_s < 0) {
    const char* e = getenv("VPTR_WGRAD_SYNC");   // block length S in K-steps (8, 16 or 32); 0 = plain launch.  Default 16: same time as
    sync_s = e ? atoi(e) : 16;                    // the plain launch, a third of its fabric traffic (profiles/r04_wgrad_standalone_pmc.txt)
    if (sync_s != 8 && sync_s != 16 && sync_s != 32) sync_s = 0;   // instantiated block lengths
    int per_cu = 0;   // the schedule assumes that 2 workgroups per CU are resident at once: ask the runtime (a wrong answer costs time, not a hang)
    if (sync_s && (hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_sync_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess ||
                   hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_sync_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess ||
                   hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_sync_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess ||
                   hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vptr_wgrad_p16_sync_kernel<16>, GNT, 2 * P16_STAGE) != hipSuccess || per_cu < 2))
      sync_s = 0;
  }
  // VPTR_WGRAD_WAVES=4: the four-wave geometry (64 x 96 wave tiles; see wgrad_p16_tile) for the two-stage launches; default 8
  static int waves = -1;
  if (waves < 0) {
    const char* e = getenv("VPTR_WGRAD_WAVES");
    waves = (e && atoi(e) == 4) ? 4 : 8;
    if (waves == 4 && (hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_sync_kernel<16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess ||
                       hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_kernel<2, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess ||
                       hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_kernel<2, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess))
      waves = 8;
  }
  // register-staged operand path (round 6): VPTR_WGRAD_RS bit 0 = the 256-row launches, bit 1 = the 128-row launches (on the four-wave geometry)
  static int rs = -1;
  if (rs < 0) {
    const char* e = getenv("VPTR_WGRAD_RS");
    rs = e ? atoi(e) : 0;
    constexpr int S256 = 256 * 128 + 24 * 1024;
    if (rs && (hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_sync_kernel<16, 8, 4, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * S256) != hipSuccess ||
               hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_kernel<2, 0, 8, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * S256) != hipSuccess ||
               hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_sync_kernel<16, 4, 4, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess ||
               hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_kernel<2, 0, 4, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess ||
               hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_kernel<2, 1, 4, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P16_STAGE) != hipSuccess))
      rs = 0;
  }
  // 256-row tiles (split_k -2: panel-synchronous, -3: plain; the host counted this launch's tiles with 256 rows): one workgroup per CU
  if ((proto->split_k == -2 || proto->split_k == -3) && (rs & 1)) {
    constexpr int STG256 = 256 * 128 + 24 * 1024;
    VPTR_CHECK(proto->atomic, "vptr_gemm_grouped(p16): 256-row tiles accumulate with atomics only");
    if (proto->split_k == -2 && sync_s && total_tiles >= 512 && vptr_cu_count() > 0 && vptr_cu_count() % 8 == 0)
      vptr_wgrad_p16_sync_kernel<16, 8, 4, 2, 1><<<vptr_cu_count(), GNT, 2 * STG256, st>>>(descs_dev, tile_start_dev, count, total_tiles);
    else
      vptr_wgrad_p16_kernel<2, 0, 8, 4, 1><<<total_tiles, GNT, 2 * STG256, st>>>(descs_dev, tile_start_dev, count, xmode | p16_prio_flag(), 0);
    return 0;
  }
  if (proto->split_k == -2 || proto->split_k == -3) {
    constexpr int STG256 = 256 * 128 + 24 * 1024;
    static bool attr256 = false;
    if (!attr256) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_sync_kernel<16, 8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STG256) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_kernel<2, 0, 8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STG256) != hipSuccess) {
        vptr_set_error("vptr_gemm_grouped(p16): cannot reserve LDS for 256-row tiles");
        return -1;
      }
      attr256 = true;
    }
    VPTR_CHECK(proto->atomic, "vptr_gemm_grouped(p16): 256-row tiles accumulate with atomics only");
    if (proto->split_k == -2 && sync_s && total_tiles >= 512 && vptr_cu_count() > 0 && vptr_cu_count() % 8 == 0)
      vptr_wgrad_p16_sync_kernel<16, 8, 4><<<vptr_cu_count(), GNT, 2 * STG256, st>>>(descs_dev, tile_start_dev, count, total_tiles);
    else
      vptr_wgrad_p16_kernel<2, 0, 8, 4><<<total_tiles, GNT, 2 * STG256, st>>>(descs_dev, tile_start_dev, count, xmode | p16_prio_flag(), 0);
    return 0;
  }
  // 192-row tiles, three 46 KB stages (split_k -4: panel-synchronous, -5: plain): one workgroup per CU with TWO K-steps of operands in
  // flight (92 KB, more than the two 40 KB workgroups of the 128-row geometry keep) and 1.25x the flops per staged byte; 2112 = 11 x 192
  if (proto->split_k == -4 || proto->split_k == -5) {
    constexpr int STG192 = 192 * 128 + 24 * 1024;
    static bool attr192 = false;
    if (!attr192) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_sync_kernel<16, 8, 3, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STG192) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&vptr_wgrad_p16_kernel<3, 0, 8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STG192) != hipSuccess) {
        vptr_set_error("vptr_gemm_grouped(p16): cannot reserve LDS for 192-row tiles");
        return -1;
      }
      attr192 = true;
    }
    VPTR_CHECK(proto->atomic, "vptr_gemm_grouped(p16): 192-row tiles accumulate with atomics only");
    if (proto->split_k == -4 && sync_s && total_tiles >= 512 && vptr_cu_count() > 0 && vptr_cu_count() % 8 == 0)
      vptr_wgrad_p16_sync_kernel<16, 8, 3, 3><<<vptr_cu_count(), GNT, 3 * STG192, st>>>(descs_dev, tile_start_dev, count, total_tiles);
    else
      vptr_wgrad_p16_kernel<3, 0, 8, 3><<<total_tiles, GNT, 3 * STG192, st>>>(descs_dev, tile_start_dev, count, xmode | p16_prio_flag(), 0);
    return 0;
  }
  if ((rs & 2) && proto->split_k >= -1) {   // 128-row launches, register-staged, four waves of 64 x 96
    if (sync_s && proto->split_k == -1 && proto->atomic && total_tiles >= 1024 && vptr_cu_count() > 0 && vptr_cu_count() % 4 == 0)
      vptr_wgrad_p16_sync_kernel<16, 4, 4, 2, 1><<<2 * vptr_cu_count(), 256, 2 * P16_STAGE, st>>>(descs_dev, tile_start_dev, count, total_tiles);
    else if (!proto->atomic) vptr_wgrad_p16_kernel<2, 1, 4, 4, 1><<<total_tiles, 256, 2 * P16_STAGE, st>>>(descs_dev, tile_start_dev, count, xmode | p16_prio_flag(), 0);
    else vptr_wgrad_p16_kernel<2, 0, 4, 4, 1><<<total_tiles, 256, 2 * P16_STAGE, st>>>(descs_dev, tile_start_dev, count, xmode | p16_prio_flag(), 0);
    return 0;
  }
  if (sync_s && proto->split_k == -1 && proto->atomic && total_tiles >= 1024 && vptr_cu_count() > 0 && vptr_cu_count() % 4 == 0) {
    const int grid = 2 * vptr_cu_count();   // two workgroups per CU (80 KB of LDS each), a multiple of 8
    if (waves == 4) {
      vptr_wgrad_p16_sync_kernel<16, 4><<<grid, 256, 2 * P16_STAGE, st>>>(descs_dev, tile_start_dev, count, total_tiles);
      return 0;
    }
    if (sync_s == 8) vptr_wgrad_p16_sync_kernel<8><<<grid, GNT, 2 * P16_STAGE, st>>>(descs_dev, tile_start_dev, count, total_tiles);
    else if (sync_s == 32) vptr_wgrad_p16_sync_kernel<32><<<grid, GNT, 2 * P16_STAGE, st>>>(descs_dev, tile_start_dev, count, total_tiles);
    else vptr_wgrad_p16_sync_kernel<16><<<grid, GNT, 2 * P16_STAGE, st>>>(descs_dev, tile_start_dev, count, total_tiles);
    return 0;
  }
  // gen > 0: the tile list as consecutive launches of `gen` tiles (one "generation" of resident workgroups each): every launch starts its
  // tiles together, so tiles that share operand panels begin in step instead of inheriting the finishing skew of their predecessors
  const int per = gen > 0 ? gen : total_tiles;
  for (int base = 0; base < total_tiles; base += per) {
    const int nt = total_tiles - base < per ? total_tiles - base : per;
    if (stages == 4) vptr_wgrad_p16_kernel<4><<<nt, GNT, 4 * P16_STAGE, st>>>(descs_dev, tile_start_dev, count, xmode | p16_prio_flag(), base);
    else if (stages == 3) vptr_wgrad_p16_kernel<3><<<nt, GNT, 3 * P16_STAGE, st>>>(descs_dev, tile_start_dev, count, xmode | p16_prio_flag(), base);
    else if (waves == 4 && !proto->atomic) vptr_wgrad_p16_kernel<2, 1, 4><<<nt, 256, 2 * P16_STAGE, st>>>(descs_dev, tile_start_dev, count, xmode | p16_prio_flag(), base);
    else if (waves == 4) vptr_wgrad_p16_kernel<2, 0, 4><<<nt, 256, 2 * P16_STAGE, st>>>(descs_dev, tile_start_dev, count, xmode | p16_prio_flag(), base);
    else if (!proto->atomic) vptr_wgrad_p16_kernel<2, 1><<<nt, GNT, 2 * P16_STAGE, st>>>(descs_dev, tile_start_dev, count, xmode | p16_prio_flag(), base);
    else vptr_wgrad_p16_kernel<2><<<nt, GNT, 2 * P16_STAGE, st>>>(descs_dev, tile_start_dev, count, xmode | p16_prio_flag(), base);
  }
  return 0;
}
