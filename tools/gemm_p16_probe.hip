// Round-2 probe (NOT part of the product path): split-bf16 GEMMs whose operands arrive in the "P16" plane format and are staged
// with global_load_lds only -- no VALU split, no ds_write in the main loop.
//
//   P16: a [rows][C] fp32-sized matrix (C % 16 == 0); every 16-channel granule is 64 bytes: 16 bf16 hi | 16 bf16 lo with
//        x = hi + lo + O(2^-17 |x|).  Same bytes, same pitch and same shape as the fp32 tensor it replaces.
//
//   nt : D[M,N] = A[M,K] . B[N,K]^T         both operands k-contiguous (nn.Linear forward, input gradients with W^T planes)
//   tn : D[NG,KX] = G[T,NG]^T . X[T,KX]     both operands token-major (weight gradients): fragments by ds_read_b64_tr_b16
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_p16_probe.hip -o gpurun_out/gemm_p16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int BM = 128, BN = 176, BK = 32;
constexpr int STAGE_B = 40 * 1024;  // 16 pieces of A + 24 pieces of B, 1 KB each

__device__ __forceinline__ uint32_t pk_bf16(const float a, const float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  typedef __attribute__((ext_vector_type(2))) float f2;
  const f2 f = {a, b};
  const bf2 h = __builtin_convertvector(f, bf2);
  return *reinterpret_cast<const uint32_t*>(&h);
}
__device__ __forceinline__ void split2(const float a, const float b, uint32_t& hi, uint32_t& lo) {
  hi = pk_bf16(a, b);
  const float fa = __uint_as_float(hi << 16), fb = __uint_as_float(hi & 0xffff0000u);
  lo = pk_bf16(a - fa, b - fb);
}
// fp32 [rows][C] -> P16 (thread = 4 channels)
__global__ void to_p16_kernel(const float* __restrict__ x, void* __restrict__ out, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  uint32_t h0, l0, h1, l1;
  split2(v.x, v.y, h0, l0);
  split2(v.z, v.w, h1, l1);
  // element index 4i: granule = (4i) / 16, position in granule = (4i) % 16  (C % 16 == 0: granules never straddle rows)
  unsigned char* o = reinterpret_cast<unsigned char*>(out) + (i >> 2) * 64 + (i & 3) * 8;
  *reinterpret_cast<uint2*>(o) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(o + 32) = make_uint2(l0, l1);
}

#define GLDS(laddr, gptr) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(laddr)), "v"(gptr) : "memory")

__device__ __forceinline__ int xcd_logical() {
  const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
  return xcd * xq + min(xcd, xr) + (blockIdx.x >> 3);
}

// ------------------------------------------------------------------------------------------------------------------------
// tr16 semantics test: LDS holds its own element index; lane l passes the address of element addr_of[l]
__global__ void tr_test(const int* __restrict__ addr_el, short* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr_el[threadIdx.x]));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}

// ------------------------------------------------------------------------------------------------------------------------
template <int NW, int PROB, int NST = 2>  // NW waves: 8 -> 4x2 waves of 32x96, 4 -> 2x2 waves of 64x96; NST stages (3: DMA two K-steps ahead, one workgroup per CU)
__global__ __launch_bounds__(NW * 64, NST >= 3 ? 2 : (NW == 4 ? 2 : 4)) void gemm_nt(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B,
                                                                     float* __restrict__ D, int M, int N, int K, int64_t strideA, int64_t strideB,
                                                                     int64_t strideD, int elim) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int MI = 128 / (NW / 2) / 16, PPW = 40 / NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // PROB = wave map variant: 0: wm = wave >> 1, wn = wave & 1;  1, 2: wm = wave & (NW/2 - 1), wn = wave / (NW/2) (waves w and w + 4 share
  // a SIMD, so every SIMD then hosts one wave of each column half); 2: the 12th (padding) fragment of the odd half is skipped
  const int wm = PROB ? (wave & (NW / 2 - 1)) : (wave >> 1), wn = PROB ? (wave / (NW / 2)) : (wave & 1), lr = lane & 15, lq = lane >> 4;   // PROB 3 = 2 + B fragment prefetch
  const int tiles_n = (N + BN - 1) / BN, tiles = tiles_n * ((M + BM - 1) / BM);
  const int lg = xcd_logical();
  const int prob = lg / tiles, tile = lg - prob * tiles;
  A += prob * strideA; B += prob * strideB; D += prob * strideD;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int nk = (K + BK - 1) / BK;
  const bool ktail = (K & 31) != 0;   // K % 32 == 16: the last step holds one granule
  const int64_t pitch = (int64_t)K * 4;

  const unsigned char* src[PPW];
  int tail_adj[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int u = wave + NW * i;
    const bool isA = u < 16;
    const int prow = (isA ? u : u - 16) * 8 + (lane >> 3), pch = lane & 7;
    const int c = pch ^ ((prow >> 1) & 7);
    const int grow = isA ? min(m0 + prow, M - 1) : min(n0 + prow, N - 1);
    src[i] = (isA ? A : B) + grow * pitch + c * 16;
    tail_adj[i] = c >= 4 ? -64 : 0;
  }
  // PROB 5: scalar-base addressing: per piece one constant 32-bit lane offset; the K-step advances an SGPR base (no VALU per piece)
  const int swave = __builtin_amdgcn_readfirstlane(wave);
  uint32_t voff[PPW], voff_tail[PPW];
  const unsigned char* const tileA = A + (int64_t)m0 * pitch;
  const unsigned char* const tileB = B + (int64_t)n0 * pitch;
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int u = wave + NW * i;
    const bool isA = u < 16;
    const int prow = (isA ? u : u - 16) * 8 + (lane >> 3), pch = lane & 7;
    const int c = pch ^ ((prow >> 1) & 7);
    const int lrow = isA ? min(prow, M - 1 - m0) : min(prow, N - 1 - n0);
    voff[i] = (uint32_t)(lrow * (int)pitch + c * 16);
    voff_tail[i] = voff[i] + (c >= 4 ? -64 : 0);
  }
  auto issue1 = [&](const int kt, const int stage, const int i) {
    if (PROB == 5) {
      const bool lastk = ktail && kt == nk - 1;
      const bool isA = NW == 8 ? i < 2 : (swave + NW * i) < 16;
      const unsigned char* base = (isA ? tileA : tileB) + (int64_t)kt * 128;
      const uint32_t laddr = (uint32_t)(stage * STAGE_B + (swave + NW * i) * 1024);
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(laddr), "v"(lastk ? voff_tail[i] : voff[i]), "s"(base) : "memory");
      return;
    }
    const bool last = ktail && kt == nk - 1;
    // elim 2 / 3: only the A / B pieces after step 1; elim 4: every piece re-reads K-step 0 (cache-resident source)
    if (kt > 1 && ((elim == 2 && wave + NW * i >= 16) || (elim == 3 && wave + NW * i < 16))) return;
    const unsigned char* g = src[i] + (int64_t)(elim == 4 ? 0 : kt) * 128 + (last ? tail_adj[i] : 0);
    const uint32_t laddr = (uint32_t)(stage * STAGE_B + (wave + NW * i) * 1024);
    // elim 5 / 6: every piece is issued, but (after step 1) with 1 / 16 active lanes: instruction count kept, bytes cut
    if (kt > 1 && elim == 7) {   // M0 written for piece 0 only: the other pieces land on top of it (timing experiment)
      if (i == 0) GLDS(laddr, g);
      else asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(g) : "memory");
    } else if (kt > 1 && elim == 5) { if (lane == 0) GLDS(laddr, g); }
    else if (kt > 1 && elim == 6) { if (lane < 16) GLDS(laddr, g); }
    else GLDS(laddr, g);
  };
  auto issue = [&](const int kt, const int stage) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) issue1(kt, stage, i);
  };
  f32x4 acc[MI][6];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int offAh[MI], offAl[MI], offBh[6], offBl[6];
  const int ch = (lq >> 1) * 4 + (lq & 1);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int r = wm * MI * 16 + mi * 16 + lr, f = (r >> 1) & 7;
    offAh[mi] = r * 128 + ((ch ^ f) << 4);
    offAl[mi] = r * 128 + (((ch + 2) ^ f) << 4);
  }
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int r = (wn * 6 + ni) * 16 + lr, f = (r >> 1) & 7;
    offBh[ni] = 16384 + r * 128 + ((ch ^ f) << 4);
    offBl[ni] = 16384 + r * 128 + (((ch + 2) ^ f) << 4);
  }
  // PROB 6: L2 warm-up -- after the DMA of step kt + 1 every wave touches one dword of the lines step kt + 2 will fetch (38 of the
  // tile's 304 rows per wave); the load is never waited for (vmcnt(1) at the next barrier leaves it in flight)
  const unsigned char* pf_ptr = nullptr;
  int pf_sink = 0;
  if (PROB == 6) {
    const int r = wave * 38 + lane;
    pf_ptr = r < 128 ? A + (int64_t)min(m0 + r, M - 1) * pitch : B + (int64_t)min(n0 + r - 128, N - 1) * pitch;
  }
  issue(0, 0);
  if (NST >= 3 && nk > 1) issue(1, 1);
  if (NST >= 4 && nk > 2) issue(2, 2);
  for (int kt = 0; kt < nk; ++kt) {
    if (NST == 4 && kt + 2 < nk) __builtin_amdgcn_s_waitcnt(0x0f70 | (2 * PPW));   // steps kt + 1, kt + 2 may still fly
    else if (NST >= 3 && kt + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0f70 | PPW);   // vmcnt(PPW): step kt landed, step kt + 1 may still fly
    else if (PROB == 6 && kt >= 1 && kt + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0f70 | 1);   // the warm-up load issued after this step's DMA may still fly
    else __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (NST >= 3) {
      if (PROB != 4 && kt + NST - 1 < nk) issue(kt + NST - 1, (kt + NST - 1) % NST);
    } else if (PROB != 4 && kt + 1 < nk && !(elim == 1 && kt > 0)) issue(kt + 1, (kt + 1) & 1);   // elim 1: no DMA after the second step
    if (PROB == 6 && kt + 2 < nk) {
      if (lane < 38) asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(pf_ptr + (int64_t)(kt + 2) * 128) : "memory");
    }
    const unsigned char* st = smem + (NST >= 3 ? kt % NST : (kt & 1)) * STAGE_B;
    bf16x8 ah[MI], al[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      ah[mi] = *reinterpret_cast<const bf16x8*>(st + offAh[mi]);
      al[mi] = *reinterpret_cast<const bf16x8*>(st + offAl[mi]);
    }
    if (ktail && kt == nk - 1 && lq >= 2) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        ah[mi] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        al[mi] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    if (PROB >= 3) {   // map 2 + the next B fragment pair requested before the MFMAs of the current one; 4: DMA pieces between the MFMA groups
      bf16x8 bh[2], bl[2];
      bh[0] = *reinterpret_cast<const bf16x8*>(st + offBh[0]);
      bl[0] = *reinterpret_cast<const bf16x8*>(st + offBl[0]);
#pragma unroll
      for (int ni = 0; ni < 6; ++ni) {
        if (ni == 5 && wn == 1) break;
        if (ni + 1 < 6 && !(ni + 1 == 5 && wn == 1)) {
          bh[(ni + 1) & 1] = *reinterpret_cast<const bf16x8*>(st + offBh[ni + 1]);
          bl[(ni + 1) & 1] = *reinterpret_cast<const bf16x8*>(st + offBl[ni + 1]);
        }
        if (PROB == 4 && NW == 8 && ni < PPW && kt + NST - 1 < nk) issue1(kt + NST - 1, (kt + NST - 1) % NST, ni);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh[ni & 1], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl[ni & 1], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh[ni & 1], acc[mi][ni], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
      if ((PROB == 2) && ni == 5 && wn == 1) break;   // wave-uniform
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(st + offBh[ni]);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(st + offBl[ni]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh, acc[mi][ni], 0, 0, 0);
      }
    }
    }
  }
  if (PROB == 6) { __builtin_amdgcn_s_waitcnt(0x0f70); asm volatile("" ::"v"(pf_sink)); }
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int nf = wn * 6 + ni, col = n0 + nf * 16 + lr;
    if (nf >= 11 || col >= N) continue;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * MI * 16 + mi * 16 + lq * 4 + r;
        if (row < M) D[(int64_t)row * N + col] = acc[mi][ni][r];
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// nt64: 64 x 176 tiles, 4 waves (2 x 2 of 32 x 96), 32 KB stages: twice the tiles of the 128-row kernel for the N = 528 outputs
// (240 tiles on 256 CUs = one lone workgroup per CU) so that two workgroups share a CU and cover each other's barrier phases
__global__ __launch_bounds__(256, 2) void gemm_nt64(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B, float* __restrict__ D, int M, int N,
                                                    int K, int64_t strideA, int64_t strideB, int64_t strideD, int elim) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int ST = 32 * 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1, lr = lane & 15, lq = lane >> 4;
  const int tiles_n = (N + BN - 1) / BN, tiles = tiles_n * ((M + 63) / 64);
  const int lg = xcd_logical();
  const int prob = lg / tiles, tile = lg - prob * tiles;
  A += prob * strideA; B += prob * strideB; D += prob * strideD;
  const int m0 = (tile / tiles_n) * 64, n0 = (tile % tiles_n) * BN;
  const int nk = (K + BK - 1) / BK;
  const bool ktail = (K & 31) != 0;
  const int64_t pitch = (int64_t)K * 4;
  const unsigned char* src[8];
  int tail_adj[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int u = wave + 4 * i;
    const bool isA = u < 8;
    const int prow = (isA ? u : u - 8) * 8 + (lane >> 3), pch = lane & 7;
    const int c = pch ^ ((prow >> 1) & 7);
    const int grow = isA ? min(m0 + prow, M - 1) : min(n0 + prow, N - 1);
    src[i] = (isA ? A : B) + grow * pitch + c * 16;
    tail_adj[i] = c >= 4 ? -64 : 0;
  }
  auto issue = [&](const int kt, const int stage) {
    const bool last = ktail && kt == nk - 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned char* g = src[i] + (int64_t)kt * 128 + (last ? tail_adj[i] : 0);
      const uint32_t laddr = (uint32_t)(stage * ST + (wave + 4 * i) * 1024);
      GLDS(laddr, g);
    }
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
    offBh[ni] = 8192 + r * 128 + ((ch ^ f) << 4);
  }
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    const unsigned char* st = smem + (kt & 1) * ST;
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      ah[mi] = *reinterpret_cast<const bf16x8*>(st + offAh[mi]);
      al[mi] = *reinterpret_cast<const bf16x8*>(st + (offAh[mi] ^ 32));
    }
    if (ktail && kt == nk - 1 && lq >= 2) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        ah[mi] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        al[mi] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
      if (ni == 5 && wn == 1) break;
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(st + offBh[ni]);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(st + (offBh[ni] ^ 32));
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

// ------------------------------------------------------------------------------------------------------------------------
// Clean nt kernels for the tile-size question (no experiment switches): TM = 128: 8 waves (4 x 2), the product's loop (wave map
// 2, B-fragment prefetch); TM = 256: 16 waves (8 x 2) share ONE B panel per stage -- 56 DMA pieces per 256 x 176 x 32 step instead
// of 2 x 40, one workgroup per CU (2 stages of 56 KB).
template <int TM>
__global__ __launch_bounds__(TM * 4, TM == 256 ? 1 : 4) void gemm_nt_t(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B,
                                                                     float* __restrict__ D, int M, int N, int K, int64_t strideA, int64_t strideB,
                                                                     int64_t strideD, int elim) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int NWV = TM / 16, APC = TM / 8, NPC = APC + 24, PPW = (NPC + NWV - 1) / NWV, STG = NPC * 1024, BOFF = APC * 1024, WMN = NWV / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & (WMN - 1), wn = wave / WMN, lr = lane & 15, lq = lane >> 4;
  const int tiles_n = (N + BN - 1) / BN, tiles = tiles_n * ((M + TM - 1) / TM);
  const int lg = xcd_logical();
  const int prob = lg / tiles, tile = lg - prob * tiles;
  A += prob * strideA; B += prob * strideB; D += prob * strideD;
  const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * BN;
  const int nk = (K + BK - 1) / BK;
  const bool ktail = (K & 31) != 0;
  const int64_t pitch = (int64_t)K * 4;
  const unsigned char* src[PPW];
  int tail_adj[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int u = min(wave + NWV * i, NPC - 1);
    const bool isA = u < APC;
    const int prow = (isA ? u : u - APC) * 8 + (lane >> 3), pch = lane & 7;
    const int c = pch ^ ((prow >> 1) & 7);
    const int grow = isA ? min(m0 + prow, M - 1) : min(n0 + prow, N - 1);
    src[i] = (isA ? A : B) + grow * pitch + c * 16;
    tail_adj[i] = c >= 4 ? -64 : 0;
  }
  auto issue = [&](const int kt, const int stage) {
    const bool last = ktail && kt == nk - 1;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      if (wave + NWV * i < NPC) {   // wave-uniform
        const unsigned char* g = src[i] + (int64_t)kt * 128 + (last ? tail_adj[i] : 0);
        GLDS((uint32_t)(stage * STG + (wave + NWV * i) * 1024), g);
      }
    }
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
    offBh[ni] = BOFF + r * 128 + ((ch ^ f) << 4);
  }
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (kt + 1 < nk && !(elim == 1 && kt > 0)) issue(kt + 1, (kt + 1) & 1);
    const unsigned char* st = smem + (kt & 1) * STG;
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      ah[mi] = *reinterpret_cast<const bf16x8*>(st + offAh[mi]);
      al[mi] = *reinterpret_cast<const bf16x8*>(st + (offAh[mi] ^ 32));
    }
    if (ktail && kt == nk - 1 && lq >= 2) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        ah[mi] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        al[mi] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    bf16x8 bh[2], bl[2];
    bh[0] = *reinterpret_cast<const bf16x8*>(st + offBh[0]);
    bl[0] = *reinterpret_cast<const bf16x8*>(st + (offBh[0] ^ 32));
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
      if (ni == 5 && wn == 1) break;
      if (ni + 1 < 6 && !(ni + 1 == 5 && wn == 1)) {
        bh[(ni + 1) & 1] = *reinterpret_cast<const bf16x8*>(st + offBh[ni + 1]);
        bl[(ni + 1) & 1] = *reinterpret_cast<const bf16x8*>(st + (offBh[ni + 1] ^ 32));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh[ni & 1], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl[ni & 1], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh[ni & 1], acc[mi][ni], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
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

// 128 x 176 tile on 16 waves (4 x 4 of 32 x 48: 3 column fragments per wave, the last column group 2), NS stages, one workgroup per CU:
// the occupancy of two 8-wave workgroups on ONE tile, for grids of <= one tile per CU
template <int NS>
__global__ __launch_bounds__(1024, 1) void gemm_nt_w16(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B,
                                                                     float* __restrict__ D, int M, int N, int K, int64_t strideA, int64_t strideB,
                                                                     int64_t strideD, int elim) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int TM = 128, NWV = 16, APC = TM / 8, NPC = APC + 24, PPW = (NPC + NWV - 1) / NWV, STG = NPC * 1024, BOFF = APC * 1024, WMN = 4, NF = 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & (WMN - 1), wn = wave / WMN, lr = lane & 15, lq = lane >> 4;
  const int tiles_n = (N + BN - 1) / BN, tiles = tiles_n * ((M + TM - 1) / TM);
  const int lg = xcd_logical();
  const int prob = lg / tiles, tile = lg - prob * tiles;
  A += prob * strideA; B += prob * strideB; D += prob * strideD;
  const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * BN;
  const int nk = (K + BK - 1) / BK;
  const bool ktail = (K & 31) != 0;
  const int64_t pitch = (int64_t)K * 4;
  const unsigned char* src[PPW];
  int tail_adj[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int u = min(wave + NWV * i, NPC - 1);
    const bool isA = u < APC;
    const int prow = (isA ? u : u - APC) * 8 + (lane >> 3), pch = lane & 7;
    const int c = pch ^ ((prow >> 1) & 7);
    const int grow = isA ? min(m0 + prow, M - 1) : min(n0 + prow, N - 1);
    src[i] = (isA ? A : B) + grow * pitch + c * 16;
    tail_adj[i] = c >= 4 ? -64 : 0;
  }
  auto issue = [&](const int kt, const int stage) {
    const bool last = ktail && kt == nk - 1;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      if (wave + NWV * i < NPC) {   // wave-uniform
        const unsigned char* g = src[i] + (int64_t)kt * 128 + (last ? tail_adj[i] : 0);
        GLDS((uint32_t)(stage * STG + (wave + NWV * i) * 1024), g);
      }
    }
  };
  f32x4 acc[2][NF];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NF; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int offAh[2], offBh[NF];
  const int ch = (lq >> 1) * 4 + (lq & 1);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int r = wm * 32 + mi * 16 + lr, f = (r >> 1) & 7;
    offAh[mi] = r * 128 + ((ch ^ f) << 4);
  }
#pragma unroll
  for (int ni = 0; ni < NF; ++ni) {
    const int r = (wn * NF + ni) * 16 + lr, f = (r >> 1) & 7;
    offBh[ni] = BOFF + r * 128 + ((ch ^ f) << 4);
  }
  issue(0, 0);
  if (NS == 3 && nk > 1) issue(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    // pieces per wave: waves 0-7 own 3, waves 8-15 own 2 (40 pieces): vmcnt of the NEXT step's pieces may stay in flight
    if (NS == 3 && kt + 1 < nk) { if (wave < 8) __builtin_amdgcn_s_waitcnt(0x0f70 | 3); else __builtin_amdgcn_s_waitcnt(0x0f70 | 2); }
    else __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (NS == 3) { if (kt + 2 < nk && !(elim == 1 && kt > 0)) issue(kt + 2, (kt + 2) % 3); }
    else if (kt + 1 < nk && !(elim == 1 && kt > 0)) issue(kt + 1, (kt + 1) & 1);
    const unsigned char* st = smem + (NS == 3 ? kt % 3 : (kt & 1)) * STG;
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      ah[mi] = *reinterpret_cast<const bf16x8*>(st + offAh[mi]);
      al[mi] = *reinterpret_cast<const bf16x8*>(st + (offAh[mi] ^ 32));
    }
    if (ktail && kt == nk - 1 && lq >= 2) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        ah[mi] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        al[mi] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    bf16x8 bh[2], bl[2];
    bh[0] = *reinterpret_cast<const bf16x8*>(st + offBh[0]);
    bl[0] = *reinterpret_cast<const bf16x8*>(st + (offBh[0] ^ 32));
#pragma unroll
    for (int ni = 0; ni < NF; ++ni) {
      if (ni == 2 && wn == 3) break;
      if (ni + 1 < NF && !(ni + 1 == 2 && wn == 3)) {
        bh[(ni + 1) & 1] = *reinterpret_cast<const bf16x8*>(st + offBh[ni + 1]);
        bl[(ni + 1) & 1] = *reinterpret_cast<const bf16x8*>(st + (offBh[ni + 1] ^ 32));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh[ni & 1], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl[ni & 1], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh[ni & 1], acc[mi][ni], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int ni = 0; ni < NF; ++ni) {
    const int nf = wn * NF + ni, col = n0 + nf * 16 + lr;
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


// ------------------------------------------------------------------------------------------------------------------------
// tn: D[NG][KX] = sum_t G[t][ng] X[t][kx];  T % 32 == 0 in the probe.
// Stage = 40 pieces of [8 t][128 B]; a piece holds, for one pair of granules, 4 mini-subtiles [8 t][16 ch] (g0 hi, g0 lo, g1 hi,
// g1 lo) of 256 bytes each: lane L of the DMA fetches chunk (L >> 4) * 2 + (L & 1) of row (L & 15) >> 1 -- whole 128-byte lines
// per row on the global side, 32-byte channel rows on the LDS side, which is what ds_read_b64_tr_b16 wants.
template <int NW, int LAY>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 4) void gemm_tn(const unsigned char* __restrict__ G, const unsigned char* __restrict__ X,
                                                                     float* __restrict__ D, int T, int NG, int KX, int64_t strideG, int64_t strideX,
                                                                     int64_t strideD, int elim) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int MI = 128 / (NW / 2) / 16, PPW = 40 / NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lq = lane >> 4;
  const int tiles_n = (KX + BN - 1) / BN, tiles = tiles_n * ((NG + BM - 1) / BM);
  const int lg = xcd_logical();
  const int prob = lg / tiles, tile = lg - prob * tiles;
  G += prob * strideG; X += prob * strideX; D += prob * strideD;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int nk = T / BK;
  const int64_t pg = (int64_t)NG * 4, px = (int64_t)KX * 4;

  const unsigned char* src[PPW];
  int64_t step_b[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int u = wave + NW * i;
    const bool isA = u < 16;
    const int v = isA ? u : u - 16;
    const int gp = v >> 2, tp = v & 3;
    // LAY 1: 8 consecutive lanes fetch one token row's 128 bytes (2 granules x 2 planes), quarters XOR-swizzled by (row >> 1) & 3
    const int ms = LAY ? (((lane & 7) >> 1) ^ ((lane >> 4) & 3)) : lane >> 4, t = LAY ? lane >> 3 : (lane & 15) >> 1, half = lane & 1;
    int gran = ((isA ? m0 : n0) >> 4) + gp * 2 + (ms >> 1);
    gran = min(gran, ((isA ? NG : KX) >> 4) - 1);
    src[i] = (isA ? G : X) + (int64_t)(tp * 8 + t) * (isA ? pg : px) + gran * 64 + (ms & 1) * 32 + half * 16;
    step_b[i] = 32 * (isA ? pg : px);
  }
  auto issue = [&](const int kt, const int stage) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      // elim >= 32: the token index wraps at `elim` tokens -- every step re-reads the same few rows (L2-resident): the ceiling of a
      // design whose tiles walk the tokens in lock-step
      const unsigned char* g = src[i] + (elim >= 32 ? kt % (elim >> 5) : kt) * step_b[i];
      const uint32_t laddr = (uint32_t)(stage * STAGE_B + (wave + NW * i) * 1024);
      GLDS(laddr, g);
    }
  };
  f32x4 acc[MI][6];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // tr-read addresses: fragment f (granule f of the operand's tile), plane p, read j: piece (f >> 1) * 4 + lq, mini-subtile
  // (f & 1) * 2 + p, row block j ^ (lq & 1)
  // LAY 1: row r = 4 rb + (lr >> 2) of piece lq at r * 128, logical quarter Q = (f & 1) * 2 + plane at position Q ^ ((r >> 1) & 3);
  // read 0 takes row block rb = lq & 1, read 1 the other: address ^ (512 | 64); lo plane = address ^ 32
  const int r0 = 4 * (lq & 1) + (lr >> 2);
  const int lane_off = LAY ? lq * 1024 + r0 * 128 + (((r0 >> 1) & 3) * 32) + (lr & 3) * 8 : lq * 1024 + ((lr >> 2) * 32) + (lr & 3) * 8;
  int offA[MI], offB[6];   // hi plane, read 0; lo = + 256; read 1 = row block flipped
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int f = wm * MI + mi;
    offA[mi] = LAY ? ((f >> 1) * 4096 + lane_off) ^ ((f & 1) * 64) : (f >> 1) * 4096 + (f & 1) * 512 + lane_off;
  }
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int f = wn * 6 + ni;
    offB[ni] = LAY ? (16384 + (f >> 1) * 4096 + lane_off) ^ ((f & 1) * 64) : 16384 + (f >> 1) * 4096 + (f & 1) * 512 + lane_off;
  }
  const int rb0 = (lq & 1) * 128, rb1 = 128 - rb0;
  constexpr int LO = LAY ? 32 : 256;
  auto frag = [&](const unsigned char* st, const int off, const int lo) -> bf16x8 {
    const int o0 = LAY ? off ^ lo : off + lo + rb0, o1 = LAY ? off ^ lo ^ 576 : off + lo + rb1;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(st + o0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(st + o1));
    const s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, c);
  };
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (kt + 1 < nk && !(elim == 1 && kt > 0)) issue(kt + 1, (kt + 1) & 1);
    const unsigned char* st = smem + (kt & 1) * STAGE_B;
    bf16x8 ah[MI], al[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      ah[mi] = frag(st, offA[mi], 0);
      al[mi] = frag(st, offA[mi], LO);
    }
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
      const bf16x8 bh = frag(st, offB[ni], 0);
      const bf16x8 bl = frag(st, offB[ni], LO);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh, acc[mi][ni], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int ni = 0; ni < 6; ++ni) {
    const int nf = wn * 6 + ni, col = n0 + nf * 16 + lr;
    if (nf >= 11 || col >= KX) continue;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * MI * 16 + mi * 16 + lq * 4 + r;
        if (row < NG) D[(int64_t)row * KX + col] = acc[mi][ni][r];
      }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static std::vector<float> rnd_vec(size_t n, float scale, uint32_t seed) {
  std::vector<float> v(n);
  uint32_t s = seed;
  for (auto& x : v) { s = s * 1664525u + 1013904223u; x = (((s >> 8) & 0xffff) / 32768.0f - 1.0f) * scale; }
  return v;
}

template <typename F>
static float time_us(F f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

static int run_nt(int M, int N, int K, int P) {
  std::vector<float> hA = rnd_vec((size_t)M * K, 1.f, 1), hB = rnd_vec((size_t)N * K, 0.05f, 2);
  float *dA, *dB, *D;
  unsigned char *pA, *pB;
  const size_t sa = (size_t)M * K * 4, sb = (size_t)N * K * 4, sd = (size_t)M * N * 4;
  // ROT > 1: successive launches rotate through ROT operand / output sets (cache-cold operands, as inside a train step)
  const int ROT = getenv("ROT") ? atoi(getenv("ROT")) : 1, P1 = P;
  P *= ROT;
  CK(hipMalloc(&dA, sa)); CK(hipMalloc(&dB, sb)); CK(hipMalloc(&D, sd * P)); CK(hipMalloc(&pA, sa * P + 256)); CK(hipMalloc(&pB, sb * P + 256));
  CK(hipMemcpy(dA, hA.data(), sa, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), sb, hipMemcpyHostToDevice));
  for (int p = 0; p < P; ++p) {
    to_p16_kernel<<<(unsigned)((sa / 16 + 255) / 256), 256>>>(dA, pA + p * sa, sa / 16);
    to_p16_kernel<<<(unsigned)((sb / 16 + 255) / 256), 256>>>(dB, pB + p * sb, sb / 16);
  }
  CK(hipDeviceSynchronize());
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  std::vector<float> hD((size_t)M * N);
  for (int var = 1; var < 8; ++var) {
    const int nw = var == 0 || var == 4 ? 4 : 8;
    auto kern = var == 0 ? gemm_nt<4, 0> : (var == 1 ? gemm_nt<8, 0> : (var == 2 ? gemm_nt<8, 1> : (var == 3 ? gemm_nt<8, 2> : (var == 4 ? gemm_nt64 : (var == 5 ? gemm_nt<8, 3> :
                (var == 6 ? gemm_nt<8, 4, 3> : gemm_nt<8, 4, 4>))))));   // map5: interleaved DMA, 3 stages; map6: 4 stages   // map5 = map2 with 3 stages, map6 = map4 (prefetch) + DMA pieces between the MFMA groups
    const int lds = var == 4 ? 64 * 1024 : (var == 6 ? 3 : (var == 7 ? 4 : 2)) * STAGE_B;   // map5: interleaved DMA + 3 stages
    const int ntile = var == 4 ? ((M + 63) / 64) * ((N + BN - 1) / BN) : tiles;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipMemset(D, 0, sd * P));
    int rot = ROT - 1;
    auto launch = [&]() {
      kern<<<ntile * P1, nw * 64, lds>>>(pA + (size_t)rot * P1 * sa, pB + (size_t)rot * P1 * sb, D + (size_t)rot * P1 * M * N, M, N, K, (int64_t)sa, (int64_t)sb, (int64_t)M * N, getenv("ELIM") ? atoi(getenv("ELIM")) : 0);
      rot = (rot + 1) % ROT;
    };
    launch();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hD.data(), D + (size_t)(P - 1) * M * N, sd, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (int s = 0; s < 4000; ++s) {
      const int i = (int)(((uint64_t)s * 2654435761u) % M), j = (int)(((uint64_t)s * 40503u + 17) % N);
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k];
      num += (hD[(size_t)i * N + j] - ref) * (hD[(size_t)i * N + j] - ref);
      den += ref * ref;
    }
    const float us = time_us(launch, 20);
    printf("nt %dw map%d  M %d N %d K %d x%d rot%d  tiles %d  %8.1f us  %7.1f TFLOP/s  rel-L2 %.2e\n", nw, var > 1 ? var - 1 : 0, M, N, K, P1, ROT, ntile * P1, us,
           2.0 * M * N * K * P1 / us / 1e6, sqrt(num / den));
  }
  hipFree(dA); hipFree(dB); hipFree(D); hipFree(pA); hipFree(pB);
  return 0;
}

static int run_tn(int T, int NG, int KX, int P, int elim) {
  std::vector<float> hG = rnd_vec((size_t)T * NG, 1.f, 3), hX = rnd_vec((size_t)T * KX, 0.5f, 4);
  float *dG, *dX, *D;
  unsigned char *pG, *pX;
  const size_t sg = (size_t)T * NG * 4, sx = (size_t)T * KX * 4, sd = (size_t)NG * KX * 4;
  CK(hipMalloc(&dG, sg)); CK(hipMalloc(&dX, sx)); CK(hipMalloc(&D, sd * P)); CK(hipMalloc(&pG, sg * P + 256)); CK(hipMalloc(&pX, sx * P + 256));
  CK(hipMemcpy(dG, hG.data(), sg, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, hX.data(), sx, hipMemcpyHostToDevice));
  for (int p = 0; p < P; ++p) {
    to_p16_kernel<<<(unsigned)((sg / 16 + 255) / 256), 256>>>(dG, pG + p * sg, sg / 16);
    to_p16_kernel<<<(unsigned)((sx / 16 + 255) / 256), 256>>>(dX, pX + p * sx, sx / 16);
  }
  CK(hipDeviceSynchronize());
  const int tiles = ((NG + BM - 1) / BM) * ((KX + BN - 1) / BN);
  std::vector<float> hD((size_t)NG * KX);
  for (int lay = 0; lay < 2; ++lay)
  for (int nw = 4; nw <= 8; nw += 4) {
    auto kern = lay ? (nw == 4 ? gemm_tn<4, 1> : gemm_tn<8, 1>) : (nw == 4 ? gemm_tn<4, 0> : gemm_tn<8, 0>);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_B));
    CK(hipMemset(D, 0, sd * P));
    auto launch = [&]() { kern<<<tiles * P, nw * 64, 2 * STAGE_B>>>(pG, pX, D, T, NG, KX, (int64_t)sg, (int64_t)sx, (int64_t)NG * KX, elim); };
    launch();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hD.data(), D + (size_t)(P - 1) * NG * KX, sd, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (int s = 0; s < 2000; ++s) {
      const int i = (int)(((uint64_t)s * 2654435761u) % NG), j = (int)(((uint64_t)s * 40503u + 17) % KX);
      double ref = 0;
      for (int t = 0; t < T; ++t) ref += (double)hG[(size_t)t * NG + i] * hX[(size_t)t * KX + j];
      num += (hD[(size_t)i * KX + j] - ref) * (hD[(size_t)i * KX + j] - ref);
      den += ref * ref;
    }
    const float us = time_us(launch, 10);
    printf("tn lay%d %dw  T %d NG %d KX %d x%d  tiles %d  %8.1f us  %7.1f TFLOP/s  rel-L2 %.2e%s\n", lay, nw, T, NG, KX, P, tiles * P, us,
           2.0 * T * NG * KX * P / us / 1e6, sqrt(num / den), elim ? " (elim)" : "");
  }
  hipFree(dG); hipFree(dX); hipFree(D); hipFree(pG); hipFree(pX);
  return 0;
}

static int run_tile(int M, int N, int K) {
  std::vector<float> hA = rnd_vec((size_t)M * K, 1.f, 1), hB = rnd_vec((size_t)N * K, 0.05f, 2);
  const int ROT = getenv("ROT") ? atoi(getenv("ROT")) : 1, elim = getenv("ELIM") ? atoi(getenv("ELIM")) : 0;
  float *dA, *dB, *D;
  unsigned char *pA, *pB;
  const size_t sa = (size_t)M * K * 4, sb = (size_t)N * K * 4, sd = (size_t)M * N * 4;
  CK(hipMalloc(&dA, sa)); CK(hipMalloc(&dB, sb)); CK(hipMalloc(&D, sd * ROT)); CK(hipMalloc(&pA, sa * ROT + 256)); CK(hipMalloc(&pB, sb * ROT + 256));
  CK(hipMemcpy(dA, hA.data(), sa, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), sb, hipMemcpyHostToDevice));
  for (int p = 0; p < ROT; ++p) {
    to_p16_kernel<<<(unsigned)((sa / 16 + 255) / 256), 256>>>(dA, pA + p * sa, sa / 16);
    to_p16_kernel<<<(unsigned)((sb / 16 + 255) / 256), 256>>>(dB, pB + p * sb, sb / 16);
  }
  CK(hipDeviceSynchronize());
  std::vector<float> hD((size_t)M * N);
  for (int tm = 128; tm <= 256; tm += 128) {
    auto kern = tm == 128 ? gemm_nt_t<128> : gemm_nt_t<256>;
    const int lds = tm == 128 ? 2 * 40 * 1024 : 2 * 56 * 1024;
    const int ntile = ((M + tm - 1) / tm) * ((N + BN - 1) / BN);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipMemset(D, 0, sd * ROT));
    int rot = ROT - 1;
    auto launch = [&]() {
      kern<<<ntile, tm * 4, lds>>>(pA + (size_t)rot * sa, pB + (size_t)rot * sb, D + (size_t)rot * M * N, M, N, K, 0, 0, 0, elim);
      rot = (rot + 1) % ROT;
    };
    launch();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hD.data(), D + (size_t)(ROT - 1) * M * N, sd, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (int s = 0; s < 4000; ++s) {
      const int i = (int)(((uint64_t)s * 2654435761u) % M), j = (int)(((uint64_t)s * 40503u + 17) % N);
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k];
      num += (hD[(size_t)i * N + j] - ref) * (hD[(size_t)i * N + j] - ref);
      den += ref * ref;
    }
    const float us = time_us(launch, 20);
    printf("tile %dx176 (%d waves)  M %d N %d K %d rot%d  tiles %d  %8.1f us  %7.1f TFLOP/s  rel-L2 %.2e%s\n", tm, tm / 16, M, N, K, ROT, ntile, us,
           2.0 * M * N * K / us / 1e6, sqrt(num / den), elim ? " (no DMA after step 1)" : "");
  }
  for (int ns = 2; ns <= 3; ++ns) {
    auto kern = ns == 2 ? gemm_nt_w16<2> : gemm_nt_w16<3>;
    const int lds = ns * 40 * 1024;
    const int ntile = ((M + 127) / 128) * ((N + BN - 1) / BN);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipMemset(D, 0, sd * ROT));
    int rot = ROT - 1;
    auto launch = [&]() {
      kern<<<ntile, 1024, lds>>>(pA + (size_t)rot * sa, pB + (size_t)rot * sb, D + (size_t)rot * M * N, M, N, K, 0, 0, 0, elim);
      rot = (rot + 1) % ROT;
    };
    launch();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hD.data(), D + (size_t)(ROT - 1) * M * N, sd, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (int s = 0; s < 4000; ++s) {
      const int i = (int)(((uint64_t)s * 2654435761u) % M), j = (int)(((uint64_t)s * 40503u + 17) % N);
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k];
      num += (hD[(size_t)i * N + j] - ref) * (hD[(size_t)i * N + j] - ref);
      den += ref * ref;
    }
    const float us = time_us(launch, 20);
    printf("tile 128x176 on 16 waves, %d stages  M %d N %d K %d rot%d  tiles %d  %8.1f us  %7.1f TFLOP/s  rel-L2 %.2e\n", ns, M, N, K, ROT, ntile, us,
           2.0 * M * N * K / us / 1e6, sqrt(num / den));
  }
  hipFree(dA); hipFree(dB); hipFree(D); hipFree(pA); hipFree(pB);
  return 0;
}

static int run_tr_test() {
  int h_addr[64];
  short h_out[256];
  int* d_addr; short* d_out;
  CK(hipMalloc(&d_addr, sizeof(h_addr))); CK(hipMalloc(&d_out, sizeof(h_out)));
  // variant 0: lane l passes element 4 l (a row-major [..][16] matrix, 4 lanes per row)
  // variant 1: 16-lane group g at base 1000 g, rows of 32 elements (stride test), lane i -> row i/4, cols 4 (i%4)
  for (int var = 0; var < 2; ++var) {
    for (int l = 0; l < 64; ++l) h_addr[l] = var == 0 ? 4 * l : 1000 * (l >> 4) + ((l & 15) >> 2) * 32 + (l & 3) * 4;
    CK(hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice));
    tr_test<<<1, 64>>>(d_addr, d_out);
    CK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 4; ++e) {
        // hypothesis: result(lane i of group, elem e) = element (i % 4) of the address passed by lane 4 e + i / 4 of the group
        const int i = l & 15, srcl = (l & ~15) + 4 * e + (i >> 2);
        const int expect = h_addr[srcl] + (i & 3);
        if (h_out[l * 4 + e] != expect) ++bad;
      }
    printf("tr16 variant %d: %d mismatches vs hypothesis; lane 0: %d %d %d %d  lane 1: %d %d %d %d  lane 5: %d %d %d %d  lane 17: %d %d %d %d\n", var, bad,
           h_out[0], h_out[1], h_out[2], h_out[3], h_out[4], h_out[5], h_out[6], h_out[7], h_out[20], h_out[21], h_out[22], h_out[23], h_out[68],
           h_out[69], h_out[70], h_out[71]);
  }
  return 0;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "all";
  if (!strcmp(mode, "tr") || !strcmp(mode, "all")) if (run_tr_test()) return 1;
  if (!strcmp(mode, "nt") || !strcmp(mode, "all")) {
    if (run_nt(10240, 528, 2112, 1)) return 1;
    if (run_nt(10240, 2112, 528, 1)) return 1;
    if (run_nt(10240, 528, 528, 1)) return 1;
    if (run_nt(10240, 528, 528, 3)) return 1;
    if (run_nt(20480, 2112, 2112, 1)) return 1;
    if (run_nt(1000, 528, 528, 1)) return 1;
  }
  if (!strcmp(mode, "tile")) {
    if (run_tile(10240, 2112, 528)) return 1;
    if (run_tile(20480, 2112, 2112)) return 1;
    if (run_tile(10240, 528, 2112)) return 1;
    if (run_tile(10240, 528, 528)) return 1;
  }
  if (!strcmp(mode, "tn") || !strcmp(mode, "all")) {
    const int elim = getenv("ELIM") ? atoi(getenv("ELIM")) : 0;
    if (run_tn(10240, 528, 2112, 8, elim)) return 1;
    if (run_tn(10240, 2112, 528, 8, elim)) return 1;
    if (run_tn(10240, 528, 528, 16, elim)) return 1;
    if (run_tn(10240, 2112, 2112, 2, elim)) return 1;
  }
  return 0;
}
