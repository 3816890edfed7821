// Layout and elementwise helpers (gfx950): NCHW <-> token-major transposes through LDS, activation backward,
// standalone dropout, row scaling, folded-BN/ReLU backward, and the clip + AdamW optimizer step.  All HBM-bound.
#include "common.h"

// [B, R, S] -> [B, S, R] tiled transpose, 32x32 tiles via LDS (+1 pad), 256 threads.
// mode 0: plain; 1: relu on output; 2: mask by (aux > 0) where aux has the SOURCE layout.
template <int MODE>
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, const float* __restrict__ aux,
                                                        float* __restrict__ dst, int R, int S) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, s0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* sb = src + (int64_t)b * R * S;
  const float* ab = (MODE == 2) ? aux + (int64_t)b * R * S : nullptr;
  float* db = dst + (int64_t)b * R * S;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + i * 8, s = s0 + tx;
    float v = 0.f;
    if (r < R && s < S) {
      v = sb[(int64_t)r * S + s];
      if (MODE == 2) v = ab[(int64_t)r * S + s] > 0.f ? v : 0.f;
    }
    tile[ty + i * 8][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int s = s0 + ty + i * 8, r = r0 + tx;
    if (r < R && s < S) {
      float v = tile[tx][ty + i * 8];
      if (MODE == 1) v = v > 0.f ? v : 0.f;
      db[(int64_t)s * R + r] = v;
    }
  }
}

static void launch_transpose(const float* src, const float* aux, float* dst, int B, int R, int S, int mode, hipStream_t st) {
  dim3 grid(cdiv(S, 32), cdiv(R, 32), B);
  if (mode == 0) transpose_kernel<0><<<grid, 256, 0, st>>>(src, aux, dst, R, S);
  else if (mode == 1) transpose_kernel<1><<<grid, 256, 0, st>>>(src, aux, dst, R, S);
  else transpose_kernel<2><<<grid, 256, 0, st>>>(src, aux, dst, R, S);
}

extern "C" int vptr_nchw_to_tokens(const float* src, float* dst, int B, int C, int HW, vptr_stream_t stream) {
  VPTR_CHECK(B > 0 && C > 0 && HW > 0, "nchw_to_tokens: bad arguments");
  launch_transpose(src, nullptr, dst, B, C, HW, 0, (hipStream_t)stream);
  VPTR_LAUNCH_CHECK();
  return 0;
}
extern "C" int vptr_tokens_to_nchw(const float* src, float* dst, int B, int C, int HW, int relu, vptr_stream_t stream) {
  VPTR_CHECK(B > 0 && C > 0 && HW > 0, "tokens_to_nchw: bad arguments");
  launch_transpose(src, nullptr, dst, B, HW, C, relu ? 1 : 0, (hipStream_t)stream);
  VPTR_LAUNCH_CHECK();
  return 0;
}
extern "C" int vptr_nchw_to_tokens_masked(const float* dout, const float* out, float* dtok, int B, int C, int HW,
                                          vptr_stream_t stream) {
  VPTR_CHECK(B > 0 && C > 0 && HW > 0, "nchw_to_tokens_masked: bad arguments");
  launch_transpose(dout, out, dtok, B, C, HW, 2, (hipStream_t)stream);
  VPTR_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ h,
                                                      float* __restrict__ dx, int rows, int C, int act, float alpha,
                                                      const float* __restrict__ rowscale, int rs_div, int rs_mod, float p,
                                                      const uint64_t* seed_dev, uint32_t site) {
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const int64_t n = (int64_t)rows * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float g = dy[i] * alpha;
    if (p > 0.f) g *= vptr_drop_scale(seed, site, (uint64_t)i, p);
    if (rowscale) g *= rowscale[((int)(i / C) / rs_div) % rs_mod];
    if (act == VPTR_ACT_GELU) g *= vptr_gelu_grad(h[i]);
    else if (act == VPTR_ACT_RELU) g = h[i] > 0.f ? g : 0.f;
    else if (act == VPTR_ACT_LRELU) g = h[i] > 0.f ? g : 0.2f * g;  // h: pre-activation or output (same sign)
    dx[i] = g;
  }
}
// float4 variant (C % 4 == 0): one 16-byte access per stream per thread instead of four
__global__ __launch_bounds__(256) void act_bwd4_kernel(const float4* __restrict__ dy, const float4* __restrict__ h,
                                                       float4* __restrict__ dx, int rows, int C4, int act, float alpha,
                                                       const float* __restrict__ rowscale, int rs_div, int rs_mod, float p,
                                                       const uint64_t* seed_dev, uint32_t site, int out_p16) {
  uint64_t seed = 0;
  if (p > 0.f) seed = *seed_dev;
  const int64_t n = (int64_t)rows * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float4 d = dy[i];
    float g[4] = {d.x * alpha, d.y * alpha, d.z * alpha, d.w * alpha};
    if (p > 0.f) {
#pragma unroll
      for (int u = 0; u < 4; ++u) g[u] *= vptr_drop_scale(seed, site, (uint64_t)i * 4 + u, p);
    }
    if (rowscale) {
      const float r = rowscale[((int)(i / C4) / rs_div) % rs_mod];
#pragma unroll
      for (int u = 0; u < 4; ++u) g[u] *= r;
    }
    if (act != VPTR_ACT_NONE) {
      const float4 hv = h[i];
      const float hh[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (act == VPTR_ACT_GELU) g[u] *= vptr_gelu_grad(hh[u]);
        else if (act == VPTR_ACT_RELU) g[u] = hh[u] > 0.f ? g[u] : 0.f;
        else if (act == VPTR_ACT_LRELU) g[u] = hh[u] > 0.f ? g[u] : 0.2f * g[u];
      }
    }
    vptr_store4_fmt(reinterpret_cast<float*>(dx), i * 4, make_float4(g[0], g[1], g[2], g[3]), out_p16);
  }
}
extern "C" int vptr_act_bwd(const float* dy, const float* h, float* dx, int rows, int C, int act, float alpha,
                            const float* rowscale, int rs_div, int rs_mod, float dropout_p, const uint64_t* seed_dev,
                            uint32_t site, int out_p16, vptr_stream_t stream) {
  VPTR_CHECK(rows > 0 && C > 0, "act_bwd: empty input");
  if (act != VPTR_ACT_NONE) VPTR_CHECK(h != nullptr, "act_bwd: activation backward needs the saved pre-activation");
  if (rowscale) VPTR_CHECK(rs_div >= 1 && rs_mod >= 1, "act_bwd: rowscale needs rs_div, rs_mod >= 1");
  if (dropout_p > 0.f) VPTR_CHECK(seed_dev && dropout_p < 1.f, "act_bwd: dropout needs seed_dev");
  const int64_t n = (int64_t)rows * C;
  const bool al16 = ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(h)) & 15) == 0;
  if (out_p16) VPTR_CHECK(C % 16 == 0 && al16 && (reinterpret_cast<uintptr_t>(dx) & 63) == 0, "act_bwd: a P16 output needs C %% 16 == 0 and aligned pointers");
  if (C % 4 == 0 && al16) {
    const int blocks4 = (int)hmin64((n / 4 + 255) / 256, 16384);
    act_bwd4_kernel<<<blocks4, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(h),
                                                             reinterpret_cast<float4*>(dx), rows, C / 4, act, alpha, rowscale, rs_div,
                                                             rs_mod, dropout_p, seed_dev, site, out_p16);
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  const int blocks = (int)hmin64((n + 255) / 256, 8192);
  act_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(dy, h, dx, rows, C, act, alpha, rowscale, rs_div, rs_mod, dropout_p,
                                                         seed_dev, site);
  VPTR_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float p,
                                                      const uint64_t* seed_dev, uint32_t site) {
  const uint64_t seed = *seed_dev;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = x[i] * vptr_drop_scale(seed, site, (uint64_t)i, p);
}
extern "C" int vptr_dropout(const float* x, float* y, int64_t n, float dropout_p, const uint64_t* seed_dev, uint32_t site,
                            vptr_stream_t stream) {
  VPTR_CHECK(n > 0 && seed_dev && dropout_p > 0.f && dropout_p < 1.f, "dropout: bad arguments");
  const int blocks = (int)hmin64((n + 255) / 256, 8192);
  dropout_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(x, y, n, dropout_p, seed_dev, site);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// dst[f, hd, wd, :] = src[f, hd - off_h, wd - off_w, :] when that lies inside the source grid, else 0: with positive
// offsets the centre padding of PadBlock.pad_if_needed (VidHRFormer_modules.py:546-557), with negative ones its crop
// (depad_if_needed :559-569); each is the other's gradient.
__global__ __launch_bounds__(256) void window_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int frames,
                                                          int Hs, int Ws, int Hd, int Wd, int off_h, int off_w, int C4) {
  const int64_t total = (int64_t)frames * Hd * Wd * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4);
    const int64_t pix = i / C4;
    const int wd = (int)(pix % Wd), hd = (int)((pix / Wd) % Hd);
    const int64_t f = pix / ((int64_t)Wd * Hd);
    const int hs = hd - off_h, ws = wd - off_w;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hs >= 0 && hs < Hs && ws >= 0 && ws < Ws) v = src[((f * Hs + hs) * Ws + ws) * C4 + c];
    dst[i] = v;
  }
}
extern "C" int vptr_window_copy(const float* src, float* dst, int frames, int Hs, int Ws, int Hd, int Wd, int off_h, int off_w, int C,
                                vptr_stream_t stream) {
  VPTR_CHECK(src && dst && frames > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0 && C % 4 == 0, "window_copy: bad arguments");
  const int64_t total = (int64_t)frames * Hd * Wd * (C / 4);
  const int blocks = (int)hmin64((total + 255) / 256, 8192);
  window_copy_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst),
                                                              frames, Hs, Ws, Hd, Wd, off_h, off_w, C / 4);
  VPTR_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void rowscale_kernel(const float* __restrict__ dy, const float* __restrict__ rs,
                                                       float* __restrict__ dx, int rows, int C, int div, int mod) {
  const int64_t total = (int64_t)rows * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / C);
    dx[i] = dy[i] * rs[(row / div) % mod];
  }
}
extern "C" int vptr_rowscale(const float* dy, const float* rowscale, float* dx, int rows, int C, int div, int mod,
                             vptr_stream_t stream) {
  VPTR_CHECK(rows > 0 && C > 0 && div >= 1 && mod >= 1, "rowscale: bad arguments");
  const int64_t total = (int64_t)rows * C;
  const int blocks = (int)hmin64((total + 255) / 256, 8192);
  rowscale_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(dy, rowscale, dx, rows, C, div, mod);
  VPTR_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void bnrelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ scale, float* __restrict__ dx, int64_t rows,
                                                         int C) {
  const int64_t total = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    dx[i] = y[i] > 0.f ? dy[i] * scale[c] : 0.f;
  }
}
extern "C" int vptr_bnrelu_bwd(const float* dy, const float* y, const float* scale, float* dx, int64_t rows, int C,
                               vptr_stream_t stream) {
  VPTR_CHECK(rows > 0 && C > 0, "bnrelu_bwd: bad arguments");
  const int64_t total = rows * C;
  const int blocks = (int)hmin64((total + 255) / 256, 8192);
  bnrelu_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(dy, y, scale, dx, rows, C);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ---- optimizer -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {   // 16-byte loads, four of them in flight per thread
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * 256;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
      const float4 v0 = g4[i], v1 = g4[i + stride], v2 = g4[i + 2 * stride], v3 = g4[i + 3 * stride];
      a0 += (v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w);
      a1 += (v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w);
      a2 += (v2.x * v2.x + v2.y * v2.y) + (v2.z * v2.z + v2.w * v2.w);
      a3 += (v3.x * v3.x + v3.y * v3.y) + (v3.z * v3.z + v3.w * v3.w);
    }
    for (; i < n4; i += stride) {
      const float4 v0 = g4[i];
      a0 += (v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w);
    }
    s = (a0 + a1) + (a2 + a3);
    for (int64_t t = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += stride) s += g[t] * g[t];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += g[i] * g[i];
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(out, s);
}
extern "C" int vptr_sumsq(const float* g, int64_t n, float* sumsq_dev, vptr_stream_t stream) {
  VPTR_CHECK(n > 0 && sumsq_dev, "sumsq: bad arguments");
  const int blocks = (int)hmin64((n + 255) / 256, 2048);
  sumsq_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(g, n, sumsq_dev);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// Fixed-order variant: every block leaves its partial in a caller-owned workspace, one block adds the partials in index order
// (run-to-run reproducible: no atomics; vptr_set_deterministic mode of FlatAdamW)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ ws) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += g[i] * g[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ ws, int nws, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < nws; i += 256) s += ws[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) *out = s;
}
extern "C" int vptr_sumsq_ws(const float* g, int64_t n, float* sumsq_dev, float* ws, int nws, vptr_stream_t stream) {
  VPTR_CHECK(n > 0 && sumsq_dev && ws && nws >= 1 && nws <= 65536, "sumsq_ws: bad arguments");
  sumsq_partial_kernel<<<nws, 256, 0, (hipStream_t)stream>>>(g, n, ws);
  sumsq_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(ws, nws, sumsq_dev);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// torch.optim.AdamW semantics (decoupled weight decay, bias correction), clip_grad_norm_ coefficient folded in.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, const float* __restrict__ step_dev,
                                                    const float* __restrict__ sumsq_dev, float max_norm, float grad_scale) {
  const float step = *step_dev;
  float coef = grad_scale;
  if (sumsq_dev) {
    const float total = sqrtf(*sumsq_dev) * grad_scale;
    coef *= fminf(1.f, max_norm / (total + 1e-6f));
  }
  const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = rsqrtf(bc2);
  auto upd = [&](const float graw, float& pi, float& mi, float& vi) {
    const float gi = graw * coef;
    pi = pi * (1.f - lr * wd);
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    pi -= step_size * mi / denom;
  };
  // 16-byte accesses (round 6: the scalar form issued seven 4-byte memory instructions per element -- 4.8 TB/s on a 3.3 GB stream); same arithmetic
  // per element, so results are bit-identical.  n4 = 0 when a pointer is not 16-byte aligned.
  const int64_t n4 = (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) ? n / 4 : 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 g4 = reinterpret_cast<const float4*>(g)[i];
    float4 p4 = reinterpret_cast<float4*>(p)[i], m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i];
    upd(g4.x, p4.x, m4.x, v4.x);
    upd(g4.y, p4.y, m4.y, v4.y);
    upd(g4.z, p4.z, m4.z, v4.z);
    upd(g4.w, p4.w, m4.w, v4.w);
    reinterpret_cast<float4*>(p)[i] = p4;
    reinterpret_cast<float4*>(m)[i] = m4;
    reinterpret_cast<float4*>(v)[i] = v4;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float pi = p[i], mi = m[i], vi = v[i];
    upd(g[i], pi, mi, vi);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}
extern "C" int vptr_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, const float* step_dev, const float* sumsq_dev, float max_norm, float grad_scale,
                          vptr_stream_t stream) {
  VPTR_CHECK(n > 0 && p && g && m && v && step_dev, "adamw: bad arguments");
  const int blocks = (int)hmin64((n / 4 + 255) / 256 + 1, 4096);
  adamw_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step_dev, sumsq_dev,
                                                       max_norm, grad_scale);
  VPTR_LAUNCH_CHECK();
  return 0;
}
