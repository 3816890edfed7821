// Pieces shared by the MFMA GEMM translation units (gemm.hip: fp32 operands split in the staging path; gemm_p16.hip: operands
// that arrive pre-split in the P16 plane format): tile constants, the fp32 -> bf16 hi / lo split, batch members and the fused
// epilogues.  Header-only (static / inline device code).
#pragma once
#include "common.h"

#define GBM 128
#define GBK 32
#define GLP 32
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


// Batched launches (desc.batch > 1): member b > 0 of the grid reads A_x<b>/B_x<b> and writes D_x<b> with bias_x<b> / alpha_x<b>.
// (Five scalars handed to the loaders and the epilogue: a patched COPY of the descriptor ended up partly in scratch.)
struct Member {
  const float* A;
  const float* B;
  float* D;
  const float* bias;
  float alpha;
  const float* res;   // residual of this member (row pitch ldr): desc.residual for member 0; the member's own D when it accumulates
  int64_t ldr;        // (desc.batch_accum) -- honoured by gemm_epilogue_rows_halves_batched only (the host admits nothing else)
};
__device__ __forceinline__ Member member_of(const vptr_gemm_desc& p, const int member) {
  Member m = {p.A, p.B, p.D, p.bias, p.alpha, p.residual, p.ldr};
  if (p.batch_stride_d != 0) {   // strided members (ABI 10): shared bias / alpha, no residual
    m.A = p.A + (int64_t)member * p.batch_stride_a;
    m.B = p.B + (int64_t)member * p.batch_stride_b;
    m.D = p.D + (int64_t)member * p.batch_stride_d;
    return m;
  }
  if (member > 0) {  // workgroup-uniform
    const bool one = member == 1;
    m.A = one ? p.A_x1 : p.A_x2;
    m.B = one ? p.B_x1 : p.B_x2;
    m.D = one ? p.D_x1 : p.D_x2;
    m.bias = one ? p.bias_x1 : p.bias_x2;
    m.alpha = one ? p.alpha_x1 : p.alpha_x2;
    m.res = nullptr;
  }
  if ((p.batch_accum >> member) & 1) { m.res = m.D; m.ldr = p.ldd; }
  return m;
}

// output row of GEMM row m (vptr_gemm_desc.d_row_w: parity-class interleave of a stride-2 transposed convolution)
__device__ __forceinline__ int64_t epi_drow(const vptr_gemm_desc& p, const int row) {
  return p.d_row_w > 0 ? (int64_t)row + (int64_t)p.d_row_w * (row / p.d_row_w) + p.d_row_off : (int64_t)row;
}

// ---- fragment-layout epilogue of the pipelined loop, used when the row-major one below cannot be (N % 4 != 0, unaligned
// pointers).  One workgroup per CU: nothing else hides its latencies, so all loads are issued before the first store.
template <int NFN, int NGRP>  // NGRP: column-fragment groups, each = all its loads, then its stores
__device__ __forceinline__ void gemm_epilogue(const vptr_gemm_desc& p, const Member& mb, f32x4 (&acc)[2][(NFN + 1) / 2], const int m0, const int n0,
                                              const int wm, const int wn, const int lr, const int lq, const bool first_split,
                                              const bool use_atomic) {
  constexpr int NFW = (NFN + 1) / 2;
  // ---- epilogue: C/D fragment layout of v_mfma_f32_16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg.
  // Every acc index is a compile-time constant (fully unrolled, no `continue`).
  // Phase 0 issues EVERY load of the epilogue (bias, column / row scales, the residual tile) before the first store: D may
  // alias the residual, so hipcc keeps loads behind earlier stores, and the first version's load -> add -> store chain per
  // element cost 48 dependent HBM round trips per lane (+40 us on a 34 us K = 528 GEMM with a residual).
  const bool plain = !p.colscale && !p.Dpre && p.act == VPTR_ACT_NONE && !p.rowscale && p.dropout_p == 0.f && !p.act_after &&
                     mb.alpha == 1.f;
  const int row_base = m0 + wm * 32 + lq * 4;
  const bool has_res = p.residual && first_split;
  int col[NFW];
  bool colok[NFW];
  constexpr int GW = (NFW + NGRP - 1) / NGRP;
  float bs[NFW], cs[NFW], rsv[2][4], res[GW][2][4];
#pragma unroll
  for (int ni = 0; ni < NFW; ++ni) {
    const int nf = wn * NFW + ni;
    col[ni] = n0 + nf * 16 + lr;
    colok[ni] = (nf < NFN) & (col[ni] < p.N);
    bs[ni] = (mb.bias && first_split && colok[ni]) ? mb.bias[col[ni]] : 0.f;
    cs[ni] = (p.colscale && colok[ni]) ? p.colscale[col[ni]] : 1.f;
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row_base + mi * 16 + r;
      rsv[mi][r] = (p.rowscale && row < p.M) ? p.rowscale[(row / p.rs_div) % p.rs_mod] : 1.f;
    }
  uint64_t seed = 0;
  if (p.dropout_p > 0.f) seed = *p.seed_dev;
#pragma unroll
  for (int g0 = 0; g0 < NFW; g0 += GW) {
#pragma unroll
  for (int ni = g0; ni < g0 + GW && ni < NFW; ++ni)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row_base + mi * 16 + r;
        res[ni - g0][mi][r] = (has_res && colok[ni] && row < p.M) ? p.residual[(int64_t)row * p.ldr + col[ni]] : 0.f;
      }
  if (plain) {  // kernel-uniform fast path: bias (+ residual), store or atomic accumulate
#pragma unroll
    for (int ni = g0; ni < g0 + GW && ni < NFW; ++ni) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row_base + mi * 16 + r;
          if (colok[ni] && row < p.M) {
            const float v = acc[mi][ni][r] + bs[ni] + res[ni - g0][mi][r];
            float* dst = mb.D + epi_drow(p, row) * p.ldd + col[ni];
            if (use_atomic) unsafeAtomicAdd(dst, v);
            else *dst = v;
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int ni = g0; ni < g0 + GW && ni < NFW; ++ni) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row_base + mi * 16 + r;
          if (colok[ni] && row < p.M) {
            float v = (acc[mi][ni][r] * cs[ni] + bs[ni]) * mb.alpha;
            if (p.Dpre) p.Dpre[(int64_t)row * p.ldd + col[ni]] = v;
            v = vptr_act(v, p.act) * rsv[mi][r];
            if (p.dropout_p > 0.f) v *= vptr_drop_scale(seed, p.site, (uint64_t)row * (uint64_t)p.N + col[ni], p.dropout_p);
            v += res[ni - g0][mi][r];
            if (p.act_after) v = v > 0.f ? v : 0.f;
            float* dst = mb.D + epi_drow(p, row) * p.ldd + col[ni];
            if (use_atomic) unsafeAtomicAdd(dst, v);
            else *dst = v;
          }
        }
      }
    }
  }
  }
}


// ---- row-major epilogue of the pipelined loop: the accumulator tile goes through LDS (free after the K loop) so that every
// thread then owns float4 pieces of full output rows: 16-byte coalesced stores (and residual / Dpre accesses) of whole
// 704-byte rows instead of 4-byte stores in 64-byte segments, all loads issued before the first store.  Needs N, ldd, ldr
// multiples of 4 and 16-byte aligned pointers (epi_vec_ok); otherwise the fragment-layout epilogue above runs.
// Tile pitch BN + 4 floats: the four 16-lane row groups of a fragment store land in four different 16-bank windows.
template <int NFN>
constexpr int epi_lds_bytes() { return GBM * (16 * NFN + 4) * (int)sizeof(float); }

__device__ __forceinline__ bool epi_vec_ok(const vptr_gemm_desc& p) {
  uintptr_t bits = reinterpret_cast<uintptr_t>(p.D) | reinterpret_cast<uintptr_t>(p.Dpre) | reinterpret_cast<uintptr_t>(p.residual) |
                   reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.colscale);
  if (p.batch > 1 && p.batch_stride_d == 0) bits |= reinterpret_cast<uintptr_t>(p.D_x1) | reinterpret_cast<uintptr_t>(p.D_x2) | reinterpret_cast<uintptr_t>(p.bias_x1) |
                           reinterpret_cast<uintptr_t>(p.bias_x2);
  return ((bits & 15) == 0) && ((p.N & 3) == 0) && ((p.ldd & 3) == 0) && ((p.ldr & 3) == 0);
}

template <int NFN>
__device__ __forceinline__ void gemm_epilogue_rows(const vptr_gemm_desc& p, const Member& mb, f32x4 (&acc)[2][(NFN + 1) / 2], float* sE, const int m0,
                                                   const int n0, const int wm, const int wn, const int lr, const int lq, const int tid,
                                                   const bool first_split, const bool use_atomic) {
  constexpr int NFW = (NFN + 1) / 2, BN = 16 * NFN, PITCH = BN + 4, C4 = BN / 4;
  // fragments -> LDS tile [128][PITCH]  (C/D layout of v_mfma_f32_16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg)
#pragma unroll
  for (int ni = 0; ni < NFW; ++ni) {
    const int nf = wn * NFW + ni;
    if (nf < NFN) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) sE[(wm * 32 + mi * 16 + lq * 4 + r) * PITCH + nf * 16 + lr] = acc[mi][ni][r];
    }
  }
  __syncthreads();
  // thread -> NFN float4 pieces: piece = it * 512 + tid, row = piece / C4, c4 = piece % C4
  const bool has_res = p.residual && first_split;
  const bool has_bias = mb.bias && first_split;
  f32x4 res[NFN], bs[NFN];
  float rsv[NFN];
#pragma unroll
  for (int it = 0; it < NFN; ++it) {
    const int piece = it * GNT + tid;
    const int rl = piece / C4;
    const int row = m0 + rl, col = n0 + (piece - rl * C4) * 4;
    const bool ok = row < p.M && col < p.N;
    bs[it] = res[it] = (f32x4){0.f, 0.f, 0.f, 0.f};   // (`cond ? *ptr : zero` made hipcc park the zero vector in scratch)
    rsv[it] = 1.f;
    if (has_bias && ok) bs[it] = *reinterpret_cast<const f32x4*>(mb.bias + col);
    if (p.rowscale && ok) rsv[it] = p.rowscale[(row / p.rs_div) % p.rs_mod];
    if (has_res && ok) res[it] = *reinterpret_cast<const f32x4*>(p.residual + (int64_t)row * p.ldr + col);
  }
  uint64_t seed = 0;
  if (p.dropout_p > 0.f) seed = *p.seed_dev;
  const bool plain = !p.colscale && !p.Dpre && p.act == VPTR_ACT_NONE && !p.rowscale && p.dropout_p == 0.f && !p.act_after &&
                     mb.alpha == 1.f;
#pragma unroll
  for (int it = 0; it < NFN; ++it) {
    const int piece = it * GNT + tid;
    const int rl = piece / C4;
    const int row = m0 + rl, col = n0 + (piece - rl * C4) * 4;
    if (row < p.M && col < p.N) {
      f32x4 v = *reinterpret_cast<const f32x4*>(&sE[rl * PITCH + (piece - rl * C4) * 4]);
      if (plain) {
        v = v + bs[it] + res[it];
      } else {
        if (p.colscale) v = v * *reinterpret_cast<const f32x4*>(p.colscale + col);   // conv + folded BatchNorm only
        v = (v + bs[it]) * mb.alpha;
        if (p.Dpre) *reinterpret_cast<f32x4*>(p.Dpre + (int64_t)row * p.ldd + col) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = vptr_act(v[e], p.act) * rsv[it];
          if (p.dropout_p > 0.f) t *= vptr_drop_scale(seed, p.site, (uint64_t)row * (uint64_t)p.N + col + e, p.dropout_p);
          t += res[it][e];
          if (p.act_after) t = t > 0.f ? t : 0.f;
          v[e] = t;
        }
      }
      float* dst = mb.D + epi_drow(p, row) * p.ldd + col;
      if (use_atomic) {
#pragma unroll
        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dst + e, v[e]);
      } else {
        *reinterpret_cast<f32x4*>(dst) = v;
      }
    }
  }
  __syncthreads();  // the tile is LDS the caller may reuse (a_rowsum reduction)
}


// The same for the single-image loop (40 KB of LDS per workgroup, two workgroups per CU): the tile goes through LDS in two
// 64-row halves, and the pieces are handled one by one (load -> compute -> store: the other workgroup covers the latency,
// and the 128-VGPR cap leaves no room for a batch of residual vectors next to the accumulators of the waiting half).
template <int NFN>
constexpr int epi_half_lds_bytes() { return (GBM / 2) * (16 * NFN + 4) * (int)sizeof(float); }

template <int NFN>
__device__ __forceinline__ void gemm_epilogue_rows_halves(const vptr_gemm_desc& p, const Member& mb, f32x4 (&acc)[2][(NFN + 1) / 2], float* sE,
                                                          const int m0, const int n0, const int wm, const int wn, const int lr, const int lq,
                                                          const int tid, const bool first_split, const bool use_atomic) {
  constexpr int NFW = (NFN + 1) / 2, BN = 16 * NFN, PITCH = BN + 4, C4 = BN / 4, HR = GBM / 2, NPIECE = HR * C4;
  const bool has_res = p.residual && first_split;
  const bool has_bias = mb.bias && first_split;
  uint64_t seed = 0;
  if (p.dropout_p > 0.f) seed = *p.seed_dev;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if ((wm >> 1) == h) {  // wave-uniform: the four waves of this row half spill their fragments
#pragma unroll
      for (int ni = 0; ni < NFW; ++ni) {
        const int nf = wn * NFW + ni;
        if (nf < NFN) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) sE[((wm & 1) * 32 + mi * 16 + lq * 4 + r) * PITCH + nf * 16 + lr] = acc[mi][ni][r];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < (NPIECE + GNT - 1) / GNT; ++it) {
      const int piece = it * GNT + tid;
      const int rl = piece / C4;
      const int row = m0 + h * HR + rl, col = n0 + (piece - rl * C4) * 4;
      if (piece < NPIECE && row < p.M && col < p.N) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&sE[rl * PITCH + (piece - rl * C4) * 4]);
        if (p.colscale) v = v * *reinterpret_cast<const f32x4*>(p.colscale + col);
        if (has_bias) v = v + *reinterpret_cast<const f32x4*>(mb.bias + col);
        v = v * mb.alpha;
        if (p.Dpre) *reinterpret_cast<f32x4*>(p.Dpre + (int64_t)row * p.ldd + col) = v;
        f32x4 res = {0.f, 0.f, 0.f, 0.f};
        if (has_res) res = *reinterpret_cast<const f32x4*>(p.residual + (int64_t)row * p.ldr + col);
        const float rs = p.rowscale ? p.rowscale[(row / p.rs_div) % p.rs_mod] : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = vptr_act(v[e], p.act) * rs;
          if (p.dropout_p > 0.f) t *= vptr_drop_scale(seed, p.site, (uint64_t)row * (uint64_t)p.N + col + e, p.dropout_p);
          t += res[e];
          if (p.act_after) t = t > 0.f ? t : 0.f;
          v[e] = t;
        }
        if (p.D_planes) {  // the consumer's operand format straight from the producer: hi | lo of this row's 32-channel block
          uint32_t hi[2], lo[2];
          split2(v[0], v[1], hi[0], lo[0]);
          split2(v[2], v[3], hi[1], lo[1]);
          __bf16* o = reinterpret_cast<__bf16*>(p.D_planes) + ((int64_t)row * ((p.N + 31) >> 5) + (col >> 5)) * 64 + (col & 31);
          *reinterpret_cast<uint2*>(o) = make_uint2(hi[0], hi[1]);
          *reinterpret_cast<uint2*>(o + 32) = make_uint2(lo[0], lo[1]);
        }
        if (mb.D) {
          float* dst = mb.D + epi_drow(p, row) * p.ldd + col;
          if (p.d_p16) {   // the consumer GEMM's operand format straight from this epilogue
            vptr_p16_store4(reinterpret_cast<unsigned char*>(mb.D), (int64_t)row * p.ldd + col, make_float4(v[0], v[1], v[2], v[3]));
          } else if (use_atomic) {
#pragma unroll
            for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dst + e, v[e]);
          } else {
            *reinterpret_cast<f32x4*>(dst) = v;
          }
        }
      }
    }
    if (h == 0) __syncthreads();
  }
}


// LEAN: the instantiation for the plain launches (bias, alpha, residual, fp32 or P16 output only -- most launches of a step): the
// other options are compiled out, which keeps the code a workgroup walks through after its K loop short (the full epilogue is
// ~20 k instructions of mostly skipped branches; measured +9 us on a 25 us K = 528 GEMM).
template <int NFN, int EPI>   // EPI: 0 every option, 1 lean (bias / alpha / residual), 2 activation gradient (desc.act_grad_src),
                              //      3 lean + DropPath row scale + dropout (out-projections, linear2: `x + drop_path(dropout(proj(.)))`)
__device__ __forceinline__ void gemm_epilogue_rows_halves_batched(const vptr_gemm_desc& p, const Member& mb, f32x4 (&acc)[2][(NFN + 1) / 2], float* sE,
                                                          const int m0, const int n0, const int wm, const int wn, const int lr, const int lq,
                                                          const int tid, const bool first_split, const bool use_atomic_in, long long* tm = nullptr) {
  constexpr int NFW = (NFN + 1) / 2, BN = 16 * NFN, PITCH = BN + 4, C4 = BN / 4, HR = GBM / 2, NPIECE = HR * C4;
  constexpr bool LEAN = EPI != 0, GRAD = EPI == 2;   // EPI 4 (round 5): lean + activation + saved pre-activation (Dpre) + dropout -- linear1 of every MLP
  const bool has_res = !GRAD && mb.res && first_split;
  const bool has_bias = !GRAD && mb.bias && first_split;
  const bool use_atomic = !LEAN && use_atomic_in;
  const float* const colscale = LEAN ? nullptr : p.colscale;
  float* const Dpre = (LEAN && EPI != 4) ? nullptr : p.Dpre;
  const float* const rowscale = (LEAN && EPI != 3) ? nullptr : p.rowscale;
  const auto D_planes = LEAN ? decltype(p.D_planes)(nullptr) : p.D_planes;
  const int act = (EPI == 1 || EPI == 3) ? VPTR_ACT_NONE : p.act;
  const float dropout_p = EPI == 1 ? 0.f : p.dropout_p;
  const bool act_after = !LEAN && p.act_after;
  uint64_t seed = 0;
  if (dropout_p > 0.f) seed = *p.seed_dev;
  constexpr int NIT = (NPIECE + GNT - 1) / GNT;
  // A lone workgroup has nobody to hide its latencies behind, and D may alias the residual (loads do not move across earlier
  // stores): the per-column operands are applied when the fragments are spilled (one batch of loads per lane), and the
  // residual / row-scale operands of a row half are all requested before its first store.
  float bs[NFW], cs[NFW];
#pragma unroll
  for (int ni = 0; ni < NFW; ++ni) {
    const int nf = wn * NFW + ni, col = n0 + nf * 16 + lr;
    const bool ok = (nf < NFN) & (col < p.N);
    bs[ni] = (has_bias && ok) ? mb.bias[col] : 0.f;
    cs[ni] = (colscale && ok) ? colscale[col] : 1.f;
  }
#ifdef VPTR_P16_TIMING
  if (tm) { float z = 0.f; for (int ni = 0; ni < NFW; ++ni) z += bs[ni]; if (z == 12345.678f) sE[0] = z; tm[0] = wall_clock64(); }
#endif
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x4 res[NIT];
    float rsv[NIT];
    float ssum = 0.f, ssq = 0.f;   // desc.frame_stats: sum / sum of squares of this thread's outputs in this row half
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int piece = it * GNT + tid;
      const int rl = piece / C4;
      const int row = m0 + h * HR + rl, col = n0 + (piece - rl * C4) * 4;
      const bool ok = piece < NPIECE && row < p.M && col < p.N;
      res[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (has_res && ok) res[it] = *reinterpret_cast<const f32x4*>(mb.res + (int64_t)row * mb.ldr + col);
      if (GRAD && ok) res[it] = *reinterpret_cast<const f32x4*>(p.act_grad_src + (int64_t)row * p.ldd + col);   // the saved pre-activations
      rsv[it] = (rowscale && ok) ? rowscale[(row / p.rs_div) % p.rs_mod] : 1.f;
    }
    if ((wm >> 1) == h) {  // wave-uniform: the four waves of this row half spill their fragments
#pragma unroll
      for (int ni = 0; ni < NFW; ++ni) {
        const int nf = wn * NFW + ni;
        if (nf < NFN) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              sE[((wm & 1) * 32 + mi * 16 + lq * 4 + r) * PITCH + nf * 16 + lr] = (acc[mi][ni][r] * cs[ni] + bs[ni]) * mb.alpha;
        }
      }
    }
    __syncthreads();
#ifdef VPTR_P16_TIMING
    if (tm) tm[1 + 2 * h] = wall_clock64();
#endif
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int piece = it * GNT + tid;
      const int rl = piece / C4;
      const int row = m0 + h * HR + rl, col = n0 + (piece - rl * C4) * 4;
      if (piece < NPIECE && row < p.M && col < p.N) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&sE[rl * PITCH + (piece - rl * C4) * 4]);
        if (Dpre) *reinterpret_cast<f32x4*>(Dpre + (int64_t)row * p.ldd + col) = v;
        const float rs = rsv[it];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = GRAD ? v[e] * vptr_act_grad(res[it][e], act) : vptr_act(v[e], act) * rs;
          if (dropout_p > 0.f) t *= vptr_drop_scale(seed, p.site, (uint64_t)row * (uint64_t)p.N + col + e, dropout_p);
          if (!GRAD) t += res[it][e];
          if (act_after) t = t > 0.f ? t : 0.f;
          v[e] = t;
        }
        if (!GRAD && p.frame_stats) {
          ssum += (v[0] + v[1]) + (v[2] + v[3]);
          ssq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        if (D_planes) {  // the consumer's operand format straight from the producer: hi | lo of this row's 32-channel block
          uint32_t hi[2], lo[2];
          split2(v[0], v[1], hi[0], lo[0]);
          split2(v[2], v[3], hi[1], lo[1]);
          __bf16* o = reinterpret_cast<__bf16*>(D_planes) + ((int64_t)row * ((p.N + 31) >> 5) + (col >> 5)) * 64 + (col & 31);
          *reinterpret_cast<uint2*>(o) = make_uint2(hi[0], hi[1]);
          *reinterpret_cast<uint2*>(o + 32) = make_uint2(lo[0], lo[1]);
        }
        if (mb.D) {
          float* dst = mb.D + epi_drow(p, row) * p.ldd + col;
          if (p.d_p16) {   // the consumer GEMM's operand format straight from this epilogue
            vptr_p16_store4(reinterpret_cast<unsigned char*>(mb.D), (int64_t)row * p.ldd + col, make_float4(v[0], v[1], v[2], v[3]));
          } else if (use_atomic) {
#pragma unroll
            for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dst + e, v[e]);
          } else {
            *reinterpret_cast<f32x4*>(dst) = v;
          }
        }
      }
    }
    if (!GRAD && p.frame_stats) {   // kernel-uniform: the 64 rows of this half lie in one frame (frame_rows % 64 == 0)
      const float S = wave_sum(ssum), Q = wave_sum(ssq);
      float* red = sE + HR * PITCH;   // behind the half tile, inside the K loop's stages
      if ((tid & 63) == 0) { red[2 * (tid >> 6)] = S; red[2 * (tid >> 6) + 1] = Q; }
      __syncthreads();
      if (tid == 0 && m0 + h * HR < p.M) {
        float s8 = 0.f, q8 = 0.f;
#pragma unroll
        for (int w = 0; w < GNT / 64; ++w) { s8 += red[2 * w]; q8 += red[2 * w + 1]; }
        const int fr = (m0 + h * HR) / p.frame_rows;
        unsafeAtomicAdd(p.frame_stats + VPTR_FRAME_STATS_STRIDE * fr, s8);
        unsafeAtomicAdd(p.frame_stats + VPTR_FRAME_STATS_STRIDE * fr + 1, q8);
      }
    }
#ifdef VPTR_P16_TIMING
    if (tm) tm[2 + 2 * h] = wall_clock64();
#endif
    if (h == 0) __syncthreads();
  }
}


// ---- epilogue of the single-image loop: element by element (load -> compute -> store).  Two workgroups share the CU there and
// cover each other's latencies, and under its 128-VGPR cap the batched-load form below spills (measured: fc1 156 -> 200 us).
template <int NFN>
__device__ __forceinline__ void gemm_epilogue_serial(const vptr_gemm_desc& p, const Member& mb, f32x4 (&acc)[2][(NFN + 1) / 2], const int m0, const int n0,
                                              const int wm, const int wn, const int lr, const int lq, const bool first_split,
                                              const bool use_atomic) {
  constexpr int NFW = (NFN + 1) / 2;
  // ---- epilogue: C/D fragment layout of v_mfma_f32_16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg.
  // Every acc index is a compile-time constant (fully unrolled, no `continue`).
  const bool plain = !p.colscale && !p.Dpre && p.act == VPTR_ACT_NONE && !p.rowscale && p.dropout_p == 0.f && !p.act_after &&
                     mb.alpha == 1.f;
  const int row_base = m0 + wm * 32 + lq * 4;
  if (plain) {  // kernel-uniform fast path: bias (+ residual), store or atomic accumulate
#pragma unroll
    for (int ni = 0; ni < NFW; ++ni) {
      const int nf = wn * NFW + ni;
      const int col = n0 + nf * 16 + lr;
      const bool colok = (nf < NFN) & (col < p.N);
      const float bs = (mb.bias && first_split && colok) ? mb.bias[col] : 0.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row_base + mi * 16 + r;
          if (colok && row < p.M) {
            float v = acc[mi][ni][r] + bs;
            if (p.residual && first_split) v += p.residual[(int64_t)row * p.ldr + col];
            float* dst = mb.D + epi_drow(p, row) * p.ldd + col;
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
      const float bs = (mb.bias && first_split && colok) ? mb.bias[col] : 0.f;
      const float cs = (p.colscale && colok) ? p.colscale[col] : 1.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row_base + mi * 16 + r;
          if (colok && row < p.M) {
            float v = (acc[mi][ni][r] * cs + bs) * mb.alpha;
            if (p.Dpre) p.Dpre[(int64_t)row * p.ldd + col] = v;
            v = vptr_act(v, p.act);
            if (p.rowscale) v *= p.rowscale[(row / p.rs_div) % p.rs_mod];
            if (p.dropout_p > 0.f) v *= vptr_drop_scale(seed, p.site, (uint64_t)row * (uint64_t)p.N + col, p.dropout_p);
            if (p.residual && first_split) v += p.residual[(int64_t)row * p.ldr + col];
            if (p.act_after) v = v > 0.f ? v : 0.f;
            float* dst = mb.D + epi_drow(p, row) * p.ldd + col;
            if (use_atomic) unsafeAtomicAdd(dst, v);
            else *dst = v;
          }
        }
      }
    }
  }
}


// XCD-aware order: workgroup b runs on XCD b % 8 and every XCD has its own L2; XCD x gets a contiguous range of the
// logical (problem, split, tile_m, tile_n) order so that operand panels shared by neighbouring tiles meet in one L2.
__device__ __forceinline__ int xcd_logical_block() {
  const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
  return xcd * xq + min(xcd, xr) + (blockIdx.x >> 3);
}


// defined in gemm_p16.hip (operands in the P16 plane format); called by vptr_gemm / vptr_gemm_grouped in gemm.hip
int vptr_gemm_p16_launch(vptr_gemm_desc& d, hipStream_t st);
int vptr_wgrad_p16_launch(const vptr_gemm_desc* proto, const vptr_gemm_desc* descs_dev, const int* tile_start_dev, int count, int total_tiles,
                          hipStream_t st);
