// Next-round probe (NOT part of the product path): what does the 128 x 176 x 32 split-bf16 tile reach when BOTH operands
// arrive as pre-split bf16 hi / lo planes and are staged with global_load_lds (no VALU split, no ds_write)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_planes_probe.hip -o gpurun_out/gemm_planes_probe
//   gemm_planes_probe M N K      D[M,N] = A[M,K] . B[N,K]^T, A and B k-contiguous fp32, split on the device first
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 128, BN = 176, BROWS = 192, BK = 32, NT = 512, NSTAGE = 3;
constexpr int A_B = BM * 64, B_B = BROWS * 64;              // bytes of one plane image (rows of 32 bf16 = 64 B)
constexpr int STAGE_B = 2 * A_B + 2 * B_B;                  // [A hi][A lo][B hi][B lo] = 40 960 B
constexpr int UNITS = STAGE_B / 1024;                       // 1 KB = 16 rows x 64 B per wave instruction: 40
constexpr int UPW = UNITS / 8;                              // 5 per wave

__device__ __forceinline__ uint32_t pk(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  bf2 v = {(__bf16)a, (__bf16)b};
  return *reinterpret_cast<uint32_t*>(&v);
}
__global__ void split_kernel(const float* __restrict__ x, __bf16* __restrict__ hi, __bf16* __restrict__ lo, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const __bf16 h = (__bf16)v;
  hi[i] = h;
  lo[i] = (__bf16)(v - (float)h);
}

// element offset (bf16) of logical (row, 16-byte chunk kq) inside a plane image -- the product kernels' swizzle
__device__ __forceinline__ int lds_off(const int row, const int kq) {
  const int prow = row ^ ((row >> 2) & 1), pch = kq ^ ((-(row >> 2)) & 3);
  return prow * 32 + pch * 8;
}

template <bool ASM>
__global__ __launch_bounds__(NT, 2) void gemm_planes(const __bf16* __restrict__ Ah, const __bf16* __restrict__ Al,
                                                     const __bf16* __restrict__ Bh, const __bf16* __restrict__ Bl,
                                                     float* __restrict__ D, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lq = lane >> 4;
  const int tiles_n = (N + BN - 1) / BN;
  const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
  const int logical = xcd * xq + min(xcd, xr) + (blockIdx.x >> 3);
  const int m0 = (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BN;
  const int nk = K / BK;

  // per-lane global source of each of this wave's UPW units (element offsets into the planes, k0 = 0)
  const __bf16* src[UPW];
  int dst[UPW];  // byte offset of the unit inside a stage
#pragma unroll
  for (int i = 0; i < UPW; ++i) {
    const int u = wave + 8 * i;
    const __bf16* plane;
    int r0, rows, base_row, ub;
    if (u < 8) { plane = Ah; r0 = u * 16; base_row = m0; rows = M; ub = 0; }
    else if (u < 16) { plane = Al; r0 = (u - 8) * 16; base_row = m0; rows = M; ub = A_B; }
    else if (u < 28) { plane = Bh; r0 = (u - 16) * 16; base_row = n0; rows = N; ub = 2 * A_B; }
    else { plane = Bl; r0 = (u - 28) * 16; base_row = n0; rows = N; ub = 2 * A_B + B_B; }
    const int prow = r0 + (lane >> 2), pch = lane & 3;
    const int row = prow ^ ((prow >> 2) & 1), kq = pch ^ ((-(row >> 2)) & 3);   // inverse of lds_off (an involution)
    const int grow = min(base_row + row, rows - 1);
    src[i] = plane + (int64_t)grow * K + kq * 8;
    dst[i] = ub + r0 * 64;
  }
  auto issue = [&](const int kt, const int stage) {
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
      const __bf16* g = src[i] + (int64_t)kt * BK;
      if constexpr (ASM) {
        const uint32_t laddr = (uint32_t)(stage * STAGE_B + dst[i]);   // dynamic LDS starts at 0 (no static LDS in this kernel)
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(laddr)), "v"(g) : "memory");
      } else {
        __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(smem + stage * STAGE_B + dst[i]), 16, 0, 0);
      }
    }
  };

  f32x4 acc[2][6];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int offA[2], offB[6];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) offA[mi] = lds_off(wm * 32 + mi * 16 + lr, lq);
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) offB[ni] = lds_off((wn * 6 + ni) * 16 + lr, lq);

  issue(0, 0);
  if (nk > 1) issue(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0f70 | UPW);  // vmcnt(UPW): step kt has landed, step kt+1 may still fly
    else __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0)
    __syncthreads();
    if (kt + 2 < nk) issue(kt + 2, (kt + 2) % NSTAGE);
    const __bf16* st = reinterpret_cast<const __bf16*>(smem + (kt % NSTAGE) * STAGE_B);
    const __bf16 *sAh = st, *sAl = st + A_B / 2, *sBh = st + A_B, *sBl = st + A_B + B_B / 2;
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      ah[mi] = *reinterpret_cast<const bf16x8*>(&sAh[offA[mi]]);
      al[mi] = *reinterpret_cast<const bf16x8*>(&sAl[offA[mi]]);
    }
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&sBh[offB[ni]]);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&sBl[offB[ni]]);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh, acc[mi][ni], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int nf = wn * 6 + ni, col = n0 + nf * 16 + lr;
    if (nf >= 11 || col >= N) continue;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 32 + mi * 16 + lq * 4 + r;
        if (row < M) D[(int64_t)row * N + col] = acc[mi][ni][r];
      }
  }
}

// Variant 3: two stages (80 KB), 104 VGPRs -> TWO workgroups per CU (4 waves per SIMD) covering each other's barrier / DMA phases
template <int ELIM>  // 0 full, 1 no DMA after the first step, 2 no MFMA, 3 fragments read once
__global__ __launch_bounds__(NT, 4) void gemm_planes_2cu(const __bf16* __restrict__ Ah, const __bf16* __restrict__ Al,
                                                     const __bf16* __restrict__ Bh, const __bf16* __restrict__ Bl,
                                                     float* __restrict__ D, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lq = lane >> 4;
  const int tiles_n = (N + BN - 1) / BN;
  const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
  const int logical = xcd * xq + min(xcd, xr) + (blockIdx.x >> 3);
  const int m0 = (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BN;
  const int nk = K / BK;

  // per-lane global source of each of this wave's UPW units (element offsets into the planes, k0 = 0)
  const __bf16* src[UPW];
  int dst[UPW];  // byte offset of the unit inside a stage
#pragma unroll
  for (int i = 0; i < UPW; ++i) {
    const int u = wave + 8 * i;
    const __bf16* plane;
    int r0, rows, base_row, ub;
    if (u < 8) { plane = Ah; r0 = u * 16; base_row = m0; rows = M; ub = 0; }
    else if (u < 16) { plane = Al; r0 = (u - 8) * 16; base_row = m0; rows = M; ub = A_B; }
    else if (u < 28) { plane = Bh; r0 = (u - 16) * 16; base_row = n0; rows = N; ub = 2 * A_B; }
    else { plane = Bl; r0 = (u - 28) * 16; base_row = n0; rows = N; ub = 2 * A_B + B_B; }
    const int prow = r0 + (lane >> 2), pch = lane & 3;
    const int row = prow ^ ((prow >> 2) & 1), kq = pch ^ ((-(row >> 2)) & 3);   // inverse of lds_off (an involution)
    const int grow = min(base_row + row, rows - 1);
    src[i] = plane + (int64_t)grow * K + kq * 8;
    dst[i] = ub + r0 * 64;
  }
  auto issue = [&](const int kt, const int stage) {
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
      const __bf16* g = src[i] + (int64_t)kt * BK;
      if constexpr (true) {
        const uint32_t laddr = (uint32_t)(stage * STAGE_B + dst[i]);   // dynamic LDS starts at 0 (no static LDS in this kernel)
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(laddr)), "v"(g) : "memory");
      } else {
        __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(smem + stage * STAGE_B + dst[i]), 16, 0, 0);
      }
    }
  };

  f32x4 acc[2][6];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int offA[2], offB[6];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) offA[mi] = lds_off(wm * 32 + mi * 16 + lr, lq);
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) offB[ni] = lds_off((wn * 6 + ni) * 16 + lr, lq);

  bf16x8 kah[2], kal[2], kbh[6], kbl[6];
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_waitcnt(0x0f70);                         // vmcnt(0): step kt has landed
    __syncthreads();
    if (kt + 1 < nk && ((ELIM != 1 && ELIM != 4 && ELIM != 5) || kt < 1)) issue(kt + 1, (kt + 1) & 1);
    const __bf16* st = reinterpret_cast<const __bf16*>(smem + (kt & 1) * STAGE_B);
    const __bf16 *sAh = st, *sAl = st + A_B / 2, *sBh = st + A_B, *sBl = st + A_B + B_B / 2;
    if ((ELIM == 3 || ELIM == 4 || ELIM == 5) && kt > 0) {  // fragments stay from step 0: MFMA + DMA + barrier only
#pragma unroll
      for (int ni = 0; ni < 6; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kal[mi], kbh[ni], acc[mi][ni], 0, 0, 0);
          if (ELIM != 5) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kah[mi], kbl[ni], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kah[mi], kbh[ni], acc[mi][ni], 0, 0, 0);
          }
        }
      if (ELIM == 5) {  // pass-major order: 12 independent MFMAs between two that touch the same accumulator
#pragma unroll
        for (int ni = 0; ni < 6; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kah[mi], kbl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 6; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kah[mi], kbh[ni], acc[mi][ni], 0, 0, 0);
      }
      continue;
    }
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      ah[mi] = *reinterpret_cast<const bf16x8*>(&sAh[offA[mi]]);
      al[mi] = *reinterpret_cast<const bf16x8*>(&sAl[offA[mi]]);
      if (ELIM >= 3) { kah[mi] = ah[mi]; kal[mi] = al[mi]; }
    }
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&sBh[offB[ni]]);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&sBl[offB[ni]]);
      if (ELIM >= 3) { kbh[ni] = bh; kbl[ni] = bl; }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        if (ELIM == 2) {  // no MFMA: fold the fragments into the accumulators with one cheap VALU op each
          acc[mi][ni][0] += (float)al[mi][0] + (float)bh[0] + (float)ah[mi][1] + (float)bl[1];
        } else {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh, acc[mi][ni], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int nf = wn * 6 + ni, col = n0 + nf * 16 + lr;
    if (nf >= 11 || col >= N) continue;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 32 + mi * 16 + lq * 4 + r;
        if (row < M) D[(int64_t)row * N + col] = acc[mi][ni][r];
      }
  }
}


// Variant 2: the fragments of step kt+1 are read from LDS into a second register set while the MFMAs of step kt run
// (one barrier per step; the LDS copy of a step is dead once its fragments sit in registers, so its stage takes step kt+3).
struct Frags {
  bf16x8 ah[2], al[2], bh[6], bl[6];
};
__global__ __launch_bounds__(NT, 2) void gemm_planes_fp(const __bf16* __restrict__ Ah, const __bf16* __restrict__ Al,
                                                        const __bf16* __restrict__ Bh, const __bf16* __restrict__ Bl,
                                                        float* __restrict__ D, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lq = lane >> 4;
  const int tiles_n = (N + BN - 1) / BN;
  const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
  const int logical = xcd * xq + min(xcd, xr) + (blockIdx.x >> 3);
  const int m0 = (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BN;
  const int nk = K / BK;
  const __bf16* src[UPW];
  int dst[UPW];
#pragma unroll
  for (int i = 0; i < UPW; ++i) {
    const int u = wave + 8 * i;
    const __bf16* plane;
    int r0, rows, base_row, ub;
    if (u < 8) { plane = Ah; r0 = u * 16; base_row = m0; rows = M; ub = 0; }
    else if (u < 16) { plane = Al; r0 = (u - 8) * 16; base_row = m0; rows = M; ub = A_B; }
    else if (u < 28) { plane = Bh; r0 = (u - 16) * 16; base_row = n0; rows = N; ub = 2 * A_B; }
    else { plane = Bl; r0 = (u - 28) * 16; base_row = n0; rows = N; ub = 2 * A_B + B_B; }
    const int prow = r0 + (lane >> 2), pch = lane & 3;
    const int row = prow ^ ((prow >> 2) & 1), kq = pch ^ ((-(row >> 2)) & 3);
    const int grow = min(base_row + row, rows - 1);
    src[i] = plane + (int64_t)grow * K + kq * 8;
    dst[i] = ub + r0 * 64;
  }
  auto issue = [&](const int kt) {
    const int stage = kt % NSTAGE;
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
      const __bf16* g = src[i] + (int64_t)kt * BK;
      const uint32_t laddr = (uint32_t)(stage * STAGE_B + dst[i]);
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(laddr)), "v"(g) : "memory");
    }
  };
  f32x4 acc[2][6];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int offA[2], offB[6];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) offA[mi] = lds_off(wm * 32 + mi * 16 + lr, lq);
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) offB[ni] = lds_off((wn * 6 + ni) * 16 + lr, lq);
  auto read = [&](Frags& f, const int kt) {
    const __bf16* st = reinterpret_cast<const __bf16*>(smem + (kt % NSTAGE) * STAGE_B);
    const __bf16 *sAh = st, *sAl = st + A_B / 2, *sBh = st + A_B, *sBl = st + A_B + B_B / 2;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      f.ah[mi] = *reinterpret_cast<const bf16x8*>(&sAh[offA[mi]]);
      f.al[mi] = *reinterpret_cast<const bf16x8*>(&sAl[offA[mi]]);
    }
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
      f.bh[ni] = *reinterpret_cast<const bf16x8*>(&sBh[offB[ni]]);
      f.bl[ni] = *reinterpret_cast<const bf16x8*>(&sBl[offB[ni]]);
    }
  };
  auto mma = [&](const Frags& f) {
#pragma unroll
    for (int ni = 0; ni < 6; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.al[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah[mi], f.bl[ni], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah[mi], f.bh[ni], acc[mi][ni], 0, 0, 0);
      }
  };
  // one pipeline step: step kt's fragments are in `cur`; fetch step kt+1's into `nxt` while multiplying
  auto step = [&](const int kt, Frags& cur, Frags& nxt) {
    if (kt + 1 < nk) {
      if (kt + 2 < nk) __builtin_amdgcn_s_waitcnt(0x0f70 | UPW);   // vmcnt(UPW): step kt+1 landed, kt+2 may fly
      else __builtin_amdgcn_s_waitcnt(0x0f70);
      __syncthreads();                                            // everyone's DMA of kt+1 visible; everyone has read stage kt
      if (kt + 3 < nk) issue(kt + 3);                             // into the stage step kt just vacated
      read(nxt, kt + 1);
    }
    mma(cur);
  };
  issue(0);
  if (nk > 1) issue(1);
  if (nk > 2) issue(2);
  if (nk > 2) __builtin_amdgcn_s_waitcnt(0x0f70 | (2 * UPW));
  else if (nk > 1) __builtin_amdgcn_s_waitcnt(0x0f70 | UPW);
  else __builtin_amdgcn_s_waitcnt(0x0f70);
  __syncthreads();
  Frags f0, f1;
  read(f0, 0);
  for (int kt = 0; kt < nk; kt += 2) {
    step(kt, f0, f1);
    if (kt + 1 < nk) step(kt + 1, f1, f0);
  }
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int nf = wn * 6 + ni, col = n0 + nf * 16 + lr;
    if (nf >= 11 || col >= N) continue;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 32 + mi * 16 + lq * 4 + r;
        if (row < M) D[(int64_t)row * N + col] = acc[mi][ni][r];
      }
  }
}

// Variant 4: hi and lo interleaved per row and 32-k block -- X_il[row][k / 32][hi: 32 bf16 | lo: 32 bf16] -- so that every DMA
// lane group fetches one FULL 128-byte line per row and K-step.  LDS image: rows of 128 bytes, 16-byte chunk c (0-3 hi, 4-7 lo)
// of row r at physical chunk c ^ ((r >> 1) & 7): the 16 rows of a fragment read hit 16 different 16-byte bank groups.
__global__ void split_il_kernel(const float* __restrict__ x, __bf16* __restrict__ out, int64_t rows, int K) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * K) return;
  const int64_t r = i / K;
  const int k = (int)(i - r * K), kb = k >> 5, kk = k & 31;
  const float v = x[i];
  const __bf16 h = (__bf16)v;
  __bf16* o = out + (r * (K >> 5) + kb) * 64;
  o[kk] = h;
  o[32 + kk] = (__bf16)(v - (float)h);
}
constexpr int ILA_B = BM * 128, ILB_B = BROWS * 128, ILSTAGE_B = ILA_B + ILB_B;  // 16 KB + 24 KB
template <int ELIM>
__global__ __launch_bounds__(NT, 4) void gemm_planes_il(const __bf16* __restrict__ A, const __bf16* __restrict__ /*unused*/, const __bf16* __restrict__ Bm,
                                                        const __bf16* __restrict__ /*unused*/, float* __restrict__ D, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lq = lane >> 4;
  const int tiles_n = (N + BN - 1) / BN;
  const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
  const int logical = xcd * xq + min(xcd, xr) + (blockIdx.x >> 3);
  const int m0 = (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BN;
  const int nk = K / BK;
  const int64_t pitch = (int64_t)(K >> 5) * 64;   // bf16 elements per row
  const __bf16* src[UPW];
  int dst[UPW];
#pragma unroll
  for (int i = 0; i < UPW; ++i) {
    const int u = wave + 8 * i;                    // 40 units of 8 rows x 128 B: 16 of A, 24 of B
    const bool isA = u < 16;
    const int r0 = (isA ? u : u - 16) * 8;
    const int prow = r0 + (lane >> 3), pch = lane & 7;
    const int c = pch ^ ((prow >> 1) & 7);
    const int grow = isA ? min(m0 + prow, M - 1) : min(n0 + prow, N - 1);
    src[i] = (isA ? A : Bm) + (int64_t)grow * pitch + c * 8;
    dst[i] = (isA ? 0 : ILA_B) + r0 * 128;
  }
  auto issue = [&](const int kt, const int stage) {
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
      const __bf16* g = src[i] + (int64_t)kt * 64;
      const uint32_t laddr = (uint32_t)(stage * ILSTAGE_B + dst[i]);
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(laddr)), "v"(g) : "memory");
    }
  };
  f32x4 acc[2][6];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int offAh[2], offAl[2], offBh[6], offBl[6];   // byte offsets inside a stage
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int r = wm * 32 + mi * 16 + lr, f = (r >> 1) & 7;
    offAh[mi] = r * 128 + ((lq ^ f) << 4);
    offAl[mi] = r * 128 + (((4 + lq) ^ f) << 4);
  }
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int r = (wn * 6 + ni) * 16 + lr, f = (r >> 1) & 7;
    offBh[ni] = ILA_B + r * 128 + ((lq ^ f) << 4);
    offBl[ni] = ILA_B + r * 128 + (((4 + lq) ^ f) << 4);
  }
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (kt + 1 < nk && (ELIM != 1 || kt < 1)) issue(kt + 1, (kt + 1) & 1);
    const unsigned char* st = smem + (kt & 1) * ILSTAGE_B;
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      ah[mi] = *reinterpret_cast<const bf16x8*>(st + offAh[mi]);
      al[mi] = *reinterpret_cast<const bf16x8*>(st + offAl[mi]);
    }
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(st + offBh[ni]);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(st + offBl[ni]);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        if (ELIM == 2) {
          acc[mi][ni][0] += (float)al[mi][0] + (float)bh[0] + (float)ah[mi][1] + (float)bl[1];
        } else {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh, acc[mi][ni], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int nf = wn * 6 + ni, col = n0 + nf * 16 + lr;
    if (nf >= 11 || col >= N) continue;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 32 + mi * 16 + lq * 4 + r;
        if (row < M) D[(int64_t)row * N + col] = acc[mi][ni][r];
      }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef void (*kern_t)(const __bf16*, const __bf16*, const __bf16*, const __bf16*, float*, int, int, int);
static int run(const char* name, kern_t kern, int nstage, const __bf16* Ah, const __bf16* Al, const __bf16* Bh, const __bf16* Bl, float* D, int M, int N, int K,
               const std::vector<float>& hA, const std::vector<float>& hB) {
  const int lds = nstage * STAGE_B;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  CK(hipMemset(D, 0, sizeof(float) * (size_t)M * N));
  kern<<<tiles, NT, lds>>>(Ah, Al, Bh, Bl, D, M, N, K);
  CK(hipDeviceSynchronize());
  std::vector<float> hD((size_t)M * N);
  CK(hipMemcpy(hD.data(), D, sizeof(float) * hD.size(), hipMemcpyDeviceToHost));
  double num = 0, den = 0;
  for (int s = 0; s < 4000; ++s) {
    const int i = (int)(((uint64_t)s * 2654435761u) % M), j = (int)(((uint64_t)s * 40503u + 17) % N);
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k];
    num += (hD[(size_t)i * N + j] - ref) * (hD[(size_t)i * N + j] - ref);
    den += ref * ref;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 20;
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) kern<<<tiles, NT, lds>>>(Ah, Al, Bh, Bl, D, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("%-8s M %d N %d K %d  tiles %d  %8.1f us  %7.1f TFLOP/s  sampled rel-L2 %.2e\n", name, M, N, K, tiles, us, 2.0 * M * N * K / us / 1e6,
         sqrt(num / den));
  return 0;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 10240, N = argc > 2 ? atoi(argv[2]) : 528, K = argc > 3 ? atoi(argv[3]) : 2112;
  if (K % BK) { printf("K must be a multiple of %d\n", BK); return 1; }
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hA) v = rnd();
  for (auto& v : hB) v = rnd() * 0.05f;
  float *dA, *dB, *D;
  __bf16 *Ah, *Al, *Bh, *Bl;
  CK(hipMalloc(&dA, sizeof(float) * hA.size())); CK(hipMalloc(&dB, sizeof(float) * hB.size())); CK(hipMalloc(&D, sizeof(float) * (size_t)M * N));
  CK(hipMalloc(&Ah, 2 * hA.size())); CK(hipMalloc(&Al, 2 * hA.size())); CK(hipMalloc(&Bh, 2 * hB.size())); CK(hipMalloc(&Bl, 2 * hB.size()));
  CK(hipMemcpy(dA, hA.data(), sizeof(float) * hA.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), sizeof(float) * hB.size(), hipMemcpyHostToDevice));
  split_kernel<<<(unsigned)((hA.size() + 255) / 256), 256>>>(dA, Ah, Al, (int64_t)hA.size());
  split_kernel<<<(unsigned)((hB.size() + 255) / 256), 256>>>(dB, Bh, Bl, (int64_t)hB.size());
  CK(hipDeviceSynchronize());
  if (run("builtin", gemm_planes<false>, 3, Ah, Al, Bh, Bl, D, M, N, K, hA, hB)) return 1;
  if (run("asm", gemm_planes<true>, 3, Ah, Al, Bh, Bl, D, M, N, K, hA, hB)) return 1;
  if (run("asm+fp", gemm_planes_fp, 3, Ah, Al, Bh, Bl, D, M, N, K, hA, hB)) return 1;
  if (run("2 wg/CU", gemm_planes_2cu<0>, 2, Ah, Al, Bh, Bl, D, M, N, K, hA, hB)) return 1;
  {
    __bf16 *Ail, *Bil;
    CK(hipMalloc(&Ail, 4 * hA.size())); CK(hipMalloc(&Bil, 4 * hB.size()));
    split_il_kernel<<<(unsigned)((hA.size() + 255) / 256), 256>>>(dA, Ail, M, K);
    split_il_kernel<<<(unsigned)((hB.size() + 255) / 256), 256>>>(dB, Bil, N, K);
    CK(hipDeviceSynchronize());
    if (run("il 2wg", gemm_planes_il<0>, 2, Ail, Ail, Bil, Bil, D, M, N, K, hA, hB)) return 1;
    if (getenv("ELIM")) {
      if (run("  il no DMA", gemm_planes_il<1>, 2, Ail, Ail, Bil, Bil, D, M, N, K, hA, hB)) return 1;
      if (run("  il no MFMA", gemm_planes_il<2>, 2, Ail, Ail, Bil, Bil, D, M, N, K, hA, hB)) return 1;
    }
  }
  if (getenv("ELIM")) {  // elimination runs (results are wrong by construction)
    if (run("  no DMA", gemm_planes_2cu<1>, 2, Ah, Al, Bh, Bl, D, M, N, K, hA, hB)) return 1;
    if (run("  no MFMA", gemm_planes_2cu<2>, 2, Ah, Al, Bh, Bl, D, M, N, K, hA, hB)) return 1;
    if (run("  no reads", gemm_planes_2cu<3>, 2, Ah, Al, Bh, Bl, D, M, N, K, hA, hB)) return 1;
    if (run("  MFMA only", gemm_planes_2cu<4>, 2, Ah, Al, Bh, Bl, D, M, N, K, hA, hB)) return 1;
    if (run("  MFMA only, pass-major", gemm_planes_2cu<5>, 2, Ah, Al, Bh, Bl, D, M, N, K, hA, hB)) return 1;
  }
  return 0;
}
