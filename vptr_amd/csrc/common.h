// Shared device helpers for the VPTR HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vptr_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

void vptr_set_error(const char* fmt, ...);

#define VPTR_CHECK(cond, ...)       \
  do {                              \
    if (!(cond)) {                  \
      vptr_set_error(__VA_ARGS__);  \
      return -1;                    \
    }                               \
  } while (0)

#define VPTR_LAUNCH_CHECK()                                                      \
  do {                                                                           \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) {                                                      \
      vptr_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return -2;                                                                 \
    }                                                                            \
  } while (0)

// vptr_set_deterministic(1): launchers whose workgroups meet in fp32 atomics switch to geometries with ONE adder per destination (or to
// fixed-order two-pass reductions), so results no longer depend on the order in which workgroups retire (api.hip)
extern int g_vptr_deterministic;

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int64_t hmin64(int64_t a, int64_t b) { return a < b ? a : b; }

// ---- counter-based dropout mask: one 32-bit hash per element ------------------------------------
__device__ __forceinline__ uint32_t vptr_mix32(uint32_t x) {  // "lowbias32" finaliser: full avalanche in 2 multiplies
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
// counter-based random word for element `idx` of dropout site `site` under `seed`.  The key depends only on launch-uniform
// values (it lives on the scalar unit); the per-element cost is one 32-bit mix (the first version ran three 64-bit
// multiplies per element, ~3x the VALU work, in every epilogue / normalise / activation-gradient pass with dropout).
__device__ __forceinline__ uint32_t vptr_hash3(uint64_t seed, uint32_t site, uint64_t idx) {
  const uint32_t key = vptr_mix32((uint32_t)seed ^ vptr_mix32((uint32_t)(seed >> 32) + (site + 1u) * 0x9E3779B9u));
  return vptr_mix32(((uint32_t)idx ^ key) + (uint32_t)(idx >> 32) * 0xC2B2AE35u);
}
// returns 1/keep if kept, 0 if dropped
__device__ __forceinline__ float vptr_drop_scale(uint64_t seed, uint32_t site, uint64_t idx, float p) {
  uint32_t h = vptr_hash3(seed, site, idx);
  uint32_t thr = (uint32_t)((double)p * 4294967296.0);
  return (h >= thr) ? 1.0f / (1.0f - p) : 0.0f;
}

// ---- P16 plane format (include/vptr_hip.h): 16-channel granules of 16 bf16 hi | 16 bf16 lo in the footprint of the fp32 tensor ----
// two fp32 -> one dword of two bf16 (round-to-nearest-even): a single v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t vptr_pk_bf16(const float a, const float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  const f2_t f = {a, b};
  const bf2_t h = __builtin_convertvector(f, bf2_t);
  return *reinterpret_cast<const uint32_t*>(&h);
}
// x = hi + lo + O(2^-17 |x|); hi = bf16(x), lo = bf16(x - hi).  6 VALU per pair.
__device__ __forceinline__ void vptr_split2(const float a, const float b, uint32_t& hi, uint32_t& lo) {
  hi = vptr_pk_bf16(a, b);
  const float fa = __uint_as_float(hi << 16), fb = __uint_as_float(hi & 0xffff0000u);
  lo = vptr_pk_bf16(a - fa, b - fb);
}
// store 4 consecutive channels starting at flat element index e (e % 4 == 0; row pitch % 16 == 0) of a P16 tensor
__device__ __forceinline__ void vptr_p16_store4(unsigned char* __restrict__ base, const int64_t e, const float4 v) {
  uint32_t h0, l0, h1, l1;
  vptr_split2(v.x, v.y, h0, l0);
  vptr_split2(v.z, v.w, h1, l1);
  unsigned char* o = base + (e >> 4) * 64 + (e & 15) * 2;
  *reinterpret_cast<uint2*>(o) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(o + 32) = make_uint2(l0, l1);
}
__device__ __forceinline__ void vptr_p16_store2(unsigned char* __restrict__ base, const int64_t e, const float a, const float b) {  // e even
  uint32_t h, l;
  vptr_split2(a, b, h, l);
  unsigned char* o = base + (e >> 4) * 64 + (e & 15) * 2;
  *reinterpret_cast<uint32_t*>(o) = h;
  *reinterpret_cast<uint32_t*>(o + 32) = l;
}
__device__ __forceinline__ void vptr_p16_store1(unsigned char* __restrict__ base, const int64_t e, const float a) {
  uint32_t h, l;
  vptr_split2(a, 0.f, h, l);
  unsigned char* o = base + (e >> 4) * 64 + (e & 15) * 2;
  *reinterpret_cast<uint16_t*>(o) = (uint16_t)h;
  *reinterpret_cast<uint16_t*>(o + 32) = (uint16_t)l;
}
// fp32 or P16 store of 4 consecutive channels at flat element index e into a tensor of either format
__device__ __forceinline__ void vptr_store4_fmt(float* __restrict__ base, const int64_t e, const float4 v, const int p16) {
  if (p16) vptr_p16_store4(reinterpret_cast<unsigned char*>(base), e, v);
  else *reinterpret_cast<float4*>(base + e) = v;
}

// ---- activations --------------------------------------------------------------------------------
// exact-erf GELU (nn.GELU()) and its derivative.  Phi(x) = 0.5 * (1 + erf(x / sqrt 2)) is evaluated with Abramowitz-Stegun
// 7.1.26 (|error| <= 1.5e-7, far inside the fp32 parity budget): one v_rcp, one v_exp and five FMAs, and the exponential
// e^(-x^2/2) it needs is the one the density term of the derivative needs anyway -- libm's erff cost about as much as
// everything else in the normalise / activation-gradient passes together.
__device__ __forceinline__ void vptr_phi(float x, float& cdf, float& pdf) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float e = __expf(-ax * ax);                    // = exp(-x^2 / 2)
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float h = 0.5f * poly * e;                     // 0.5 * erfc(|x| / sqrt 2)
  cdf = x >= 0.f ? 1.0f - h : h;
  pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float vptr_gelu(float x) {
  float c, d;
  vptr_phi(x, c, d);
  return x * c;
}
__device__ __forceinline__ float vptr_gelu_grad(float x) {
  float c, d;
  vptr_phi(x, c, d);
  return c + x * d;
}
__device__ __forceinline__ float vptr_act(float v, int act) {
  if (act == VPTR_ACT_GELU) return vptr_gelu(v);
  if (act == VPTR_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == VPTR_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
  return v;
}
// d act(h) / dh from the value the forward saved (GELU: the pre-activation; ReLU / LeakyReLU: pre-activation or output, same sign)
__device__ __forceinline__ float vptr_act_grad(float h, int act) {
  if (act == VPTR_ACT_GELU) return vptr_gelu_grad(h);
  if (act == VPTR_ACT_RELU) return h > 0.f ? 1.f : 0.f;
  if (act == VPTR_ACT_LRELU) return h > 0.f ? 1.f : 0.2f;
  return 1.f;
}

// ---- wave / block reductions (wave = 64) ----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` is >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
