// Winograd F(4x4, 3x3) transforms for the FROZEN stride-1 3x3 convolutions of the ResNet encoder (ResNetAutoEncoder.py:127-151: the
// 18 ResnetBlock convolutions 528 -> 528 on 8 x 8 maps are 92 % of the encoder's FLOPs, and in stage 2 / inference their weights never
// change -- train_NAR.py:54-56,190-191).  Y = A^T [ (G g G^T) . (B^T d B) ] A per 4 x 4 output tile: 36 multiplies instead of 144, i.e.
// a 3 x 3 convolution becomes 36 independent [tiles x Cin] . [Cin x Cout] products -- ONE strided-batch launch of the P16 nt GEMM
// (vptr_gemm_desc.batch_stride_*), 4x fewer MFMA passes than the implicit GEMM of csrc/gemm.hip.
//
//   vptr_wino_in  : fp32 NHWC map [frames, H, W, C]  ->  V[36][Mpad][C] in the P16 operand format (row = (frame, tile_y, tile_x));
//                   the 6 x 6 input patch of a tile with the convolution's own padding (zero / reflect / replicate) folded in
//   vptr_gemm     : M36[xi nu] = V[xi nu] . U[xi nu]^T        (U: transformed filters, P16, made once per weight version on the host)
//   vptr_wino_out : M36[36][Mpad][C] fp32 -> NHWC map: A^T m A, folded-BN scale / shift, ReLU, residual add, trailing ReLU
//
// Interpolation points 0, +-1, +-2, inf (Lavin & Gray); transforms in fp32, operands split to bf16 hi / lo afterwards: the relative error
// of a nine-block encoder is 6e-5 against 1.4e-5 for the direct 3-pass convolution (tools/winograd_numerics.py), inside the 1e-3 bar.
// Both kernels are memory-bound: one thread = one tile x 4 channels, 36 independent 16-byte loads in flight per thread.
#include "common.h"

__device__ __forceinline__ float4 f4_axpy(const float a, const float4 x, const float4 y) {
  return make_float4(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z), fmaf(a, x.w, y.w));
}
__device__ __forceinline__ float4 f4_add(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_sub(const float4 a, const float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4_mul(const float a, const float4 x) { return make_float4(a * x.x, a * x.y, a * x.z, a * x.w); }

// t = B^T d   (6 -> 6):  rows of B^T: [4 0 -5 0 1 0] [0 -4 -4 1 1 0] [0 4 -4 -1 1 0] [0 -2 -1 2 1 0] [0 2 -1 -2 1 0] [0 4 0 -5 0 1]
__device__ __forceinline__ void wino_bt6(const float4 d0, const float4 d1, const float4 d2, const float4 d3, const float4 d4, const float4 d5,
                                         float4& t0, float4& t1, float4& t2, float4& t3, float4& t4, float4& t5) {
  const float4 a = f4_axpy(-4.f, d2, d4);        // d4 - 4 d2
  const float4 b = f4_axpy(-4.f, d1, d3);        // d3 - 4 d1
  const float4 c = f4_sub(d4, d2);               // d4 - d2
  const float4 e = f4_mul(2.f, f4_sub(d3, d1));  // 2 (d3 - d1)
  t0 = f4_axpy(4.f, d0, f4_axpy(-5.f, d2, d4));
  t1 = f4_add(a, b);
  t2 = f4_sub(a, b);
  t3 = f4_add(c, e);
  t4 = f4_sub(c, e);
  t5 = f4_axpy(4.f, d1, f4_axpy(-5.f, d3, d5));
}
// y = A^T m   (6 -> 4):  rows of A^T: [1 1 1 1 1 0] [0 1 -1 2 -2 0] [0 1 1 4 4 0] [0 1 -1 8 -8 1]
__device__ __forceinline__ void wino_at4(const float4 m0, const float4 m1, const float4 m2, const float4 m3, const float4 m4, const float4 m5,
                                         float4& y0, float4& y1, float4& y2, float4& y3) {
  const float4 s12 = f4_add(m1, m2), d12 = f4_sub(m1, m2), s34 = f4_add(m3, m4), d34 = f4_sub(m3, m4);
  y0 = f4_add(m0, f4_add(s12, s34));
  y1 = f4_axpy(2.f, d34, d12);
  y2 = f4_axpy(4.f, s34, s12);
  y3 = f4_add(f4_axpy(8.f, d34, d12), m5);
}

// source index of padded coordinate i (-1 .. n) under the convolution's padding mode; zero padding returns -1 for "outside"
__device__ __forceinline__ int wino_src(const int i, const int n, const int mode) {
  if (i >= 0 && i < n) return i;
  if (mode == 1) return i < 0 ? -i : 2 * n - 2 - i;   // reflect (ReflectionPad2d(1): -1 -> 1, n -> n - 2)
  if (mode == 2) return i < 0 ? 0 : n - 1;            // replicate
  return -1;
}

// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino_in_kernel(const float* __restrict__ x_, unsigned char* __restrict__ V, const int frames, const int H,
                                                       const int W, const int C4, const int64_t Mpad, const int pad_mode) {
  const int TH = H >> 2, TW = W >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)frames * TH * TW * C4) return;
  const int c4 = (int)(idx % C4);
  const int64_t row = idx / C4;                  // (frame, tile_y, tile_x)
  const int tx = (int)(row % TW), ty = (int)((row / TW) % TH);
  const int64_t f = row / ((int64_t)TW * TH);
  const float4* __restrict__ x = reinterpret_cast<const float4*>(x_);
  int sx[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) sx[j] = wino_src(4 * tx - 1 + j, W, pad_mode);
  float4 d[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int sy = wino_src(4 * ty - 1 + i, H, pad_mode);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const bool ok = sy >= 0 && sx[j] >= 0;
      const float4 v = x[((f * H + (ok ? sy : 0)) * W + (ok ? sx[j] : 0)) * C4 + c4];
      d[i][j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // columns: d <- B^T d
#pragma unroll
  for (int j = 0; j < 6; ++j) wino_bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
  // rows: v = (B^T d) B, stored as it is produced
  const int64_t C = (int64_t)C4 * 4;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float4 v[6];
    wino_bt6(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5], v[0], v[1], v[2], v[3], v[4], v[5]);
#pragma unroll
    for (int j = 0; j < 6; ++j) vptr_p16_store4(V, ((int64_t)(i * 6 + j) * Mpad + row) * C + (int64_t)c4 * 4, v[j]);
  }
}

// y may alias residual: every element is read and written by the same thread
__global__ __launch_bounds__(256) void wino_out_kernel(const float* __restrict__ M_, const float* __restrict__ scale_, const float* __restrict__ shift_,
                                                        const float* residual_, float* y_, const int frames, const int H, const int W, const int C4,
                                                        const int64_t Mpad, const int relu, const int act_after) {
  const int TH = H >> 2, TW = W >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)frames * TH * TW * C4) return;
  const int c4 = (int)(idx % C4);
  const int64_t row = idx / C4;
  const int tx = (int)(row % TW), ty = (int)((row / TW) % TH);
  const int64_t f = row / ((int64_t)TW * TH);
  const float4* __restrict__ M = reinterpret_cast<const float4*>(M_);
  const float4* residual = reinterpret_cast<const float4*>(residual_);
  float4* y = reinterpret_cast<float4*>(y_);
  const int64_t pitch = Mpad * C4;               // float4 per (xi, nu) matrix
  const int64_t base = row * C4 + c4;
  // columns first (xi -> 4 rows), one nu at a time: 6 loads in flight per step, 24 float4 kept
  float4 t[4][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float4 m0 = M[(0 * 6 + j) * pitch + base], m1 = M[(1 * 6 + j) * pitch + base], m2 = M[(2 * 6 + j) * pitch + base];
    const float4 m3 = M[(3 * 6 + j) * pitch + base], m4 = M[(4 * 6 + j) * pitch + base], m5 = M[(5 * 6 + j) * pitch + base];
    wino_at4(m0, m1, m2, m3, m4, m5, t[0][j], t[1][j], t[2][j], t[3][j]);
  }
  const float4 sc = scale_ ? reinterpret_cast<const float4*>(scale_)[c4] : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 sh = shift_ ? reinterpret_cast<const float4*>(shift_)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 o[4];
    wino_at4(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], o[0], o[1], o[2], o[3]);
    const int64_t e = ((f * H + 4 * ty + i) * W + 4 * tx) * C4 + c4;
    float4 r[4];
    if (residual) {
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = residual[e + (int64_t)j * C4];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 v = make_float4(fmaf(o[j].x, sc.x, sh.x), fmaf(o[j].y, sc.y, sh.y), fmaf(o[j].z, sc.z, sh.z), fmaf(o[j].w, sc.w, sh.w));
      if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      if (residual) v = f4_add(v, r[j]);
      if (act_after) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      y[e + (int64_t)j * C4] = v;
    }
  }
}

extern "C" int vptr_wino_in(const float* x, void* V, int frames, int H, int W, int C, int64_t Mpad, int pad_mode, vptr_stream_t stream) {
  VPTR_CHECK(x && V && frames > 0 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0 && C > 0 && C % 16 == 0,
             "wino_in: needs H, W multiples of 4 and C a multiple of 16 (got H %d W %d C %d)", H, W, C);
  VPTR_CHECK(pad_mode >= 0 && pad_mode <= 2, "wino_in: pad_mode 0 zero / 1 reflect / 2 replicate");
  const int64_t rows = (int64_t)frames * (H / 4) * (W / 4);
  VPTR_CHECK(Mpad >= rows, "wino_in: Mpad %lld < tile rows %lld", (long long)Mpad, (long long)rows);
  VPTR_CHECK(((reinterpret_cast<uintptr_t>(x) & 15) | (reinterpret_cast<uintptr_t>(V) & 63)) == 0, "wino_in: x 16-byte, V 64-byte aligned");
  const int64_t total = rows * (C / 4);
  wino_in_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, reinterpret_cast<unsigned char*>(V), frames, H, W, C / 4, Mpad, pad_mode);
  VPTR_LAUNCH_CHECK();
  return 0;
}

extern "C" int vptr_wino_out(const float* M36, const float* scale, const float* shift, const float* residual, float* y, int frames, int H, int W,
                             int C, int64_t Mpad, int relu, int act_after, vptr_stream_t stream) {
  VPTR_CHECK(M36 && y && frames > 0 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0 && C > 0 && C % 4 == 0,
             "wino_out: needs H, W multiples of 4 and C a multiple of 4 (got H %d W %d C %d)", H, W, C);
  const int64_t rows = (int64_t)frames * (H / 4) * (W / 4);
  VPTR_CHECK(Mpad >= rows, "wino_out: Mpad %lld < tile rows %lld", (long long)Mpad, (long long)rows);
  VPTR_CHECK(((reinterpret_cast<uintptr_t>(M36) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift) |
               reinterpret_cast<uintptr_t>(residual)) & 15) == 0, "wino_out: operands must be 16-byte aligned");
  const int64_t total = rows * (C / 4);
  wino_out_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(M36, scale, shift, residual, y, frames, H, W, C / 4, Mpad, relu, act_after);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Output transform of one convolution and input transform of the NEXT in one pass: the activated map of a (frame, 16-quad channel slab)
// unit stays in LDS between the two ([pixel][16 quads] float4, H * W * 256 bytes), so the intermediate map of a ResnetBlock never
// touches HBM and the residual stream is written once and not re-read (per convolution: one 43 MB write and one 2.25x-overlapped read
// less).  Unit = T tiles x 16 quads threads (T = (H/4)(W/4) must divide 16: 4 x 4 ... 16 x 16 maps); 256 / (16 T) units per workgroup,
// 64 KB of LDS.  C4 need not be a multiple of 16: the lanes of the last slab beyond C4 idle.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino_out_in_kernel(const float* __restrict__ M_, const float* __restrict__ scale_, const float* __restrict__ shift_,
                                                           const float* residual_, float* y_, unsigned char* __restrict__ V, const int frames, const int H,
                                                           const int W, const int C4, const int64_t Mpad, const int relu, const int act_after,
                                                           const int pad_mode) {
  extern __shared__ float4 wino_map[];   // [unit][H * W][16]
  const int TH = H >> 2, TW = W >> 2, T = TH * TW, HW = H * W;
  const int upw = 256 / (16 * T);                 // units per workgroup
  const int tid = threadIdx.x;
  const int u = tid / (16 * T), l = tid - u * 16 * T;
  const int c4l = l & 15, tile = l >> 4;
  const int nslab = (C4 + 15) >> 4;
  const int64_t unit = (int64_t)blockIdx.x * upw + u;
  const int64_t f = unit / nslab;
  const int c4 = (int)(unit - f * nslab) * 16 + c4l;
  const bool live = f < frames && c4 < C4;
  const int ty = tile / TW, tx = tile - ty * TW;
  const int64_t row = f * T + tile;
  float4* map = wino_map + (size_t)u * HW * 16;
  if (live) {
    const float4* __restrict__ M = reinterpret_cast<const float4*>(M_);
    const float4* residual = reinterpret_cast<const float4*>(residual_);
    float4* y = reinterpret_cast<float4*>(y_);
    const int64_t pitch = Mpad * C4, base = row * C4 + c4;
    float4 t[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float4 m0 = M[(0 * 6 + j) * pitch + base], m1 = M[(1 * 6 + j) * pitch + base], m2 = M[(2 * 6 + j) * pitch + base];
      const float4 m3 = M[(3 * 6 + j) * pitch + base], m4 = M[(4 * 6 + j) * pitch + base], m5 = M[(5 * 6 + j) * pitch + base];
      wino_at4(m0, m1, m2, m3, m4, m5, t[0][j], t[1][j], t[2][j], t[3][j]);
    }
    const float4 sc = scale_ ? reinterpret_cast<const float4*>(scale_)[c4] : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = shift_ ? reinterpret_cast<const float4*>(shift_)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 o[4];
      wino_at4(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], o[0], o[1], o[2], o[3]);
      const int pix = (4 * ty + i) * W + 4 * tx;
      const int64_t e = (f * HW + pix) * C4 + c4;
      float4 r[4];
      if (residual) {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = residual[e + (int64_t)j * C4];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 v = make_float4(fmaf(o[j].x, sc.x, sh.x), fmaf(o[j].y, sc.y, sh.y), fmaf(o[j].z, sc.z, sh.z), fmaf(o[j].w, sc.w, sh.w));
        if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        if (residual) v = f4_add(v, r[j]);
        if (act_after) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        if (y) y[e + (int64_t)j * C4] = v;
        map[(pix + j) * 16 + c4l] = v;
      }
    }
  }
  __syncthreads();
  if (!live) return;
  int sx[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) sx[j] = wino_src(4 * tx - 1 + j, W, pad_mode);
  float4 d[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int sy = wino_src(4 * ty - 1 + i, H, pad_mode);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const bool ok = sy >= 0 && sx[j] >= 0;
      const float4 v = map[((ok ? sy : 0) * W + (ok ? sx[j] : 0)) * 16 + c4l];
      d[i][j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) wino_bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
  const int64_t C = (int64_t)C4 * 4;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float4 v[6];
    wino_bt6(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5], v[0], v[1], v[2], v[3], v[4], v[5]);
#pragma unroll
    for (int j = 0; j < 6; ++j) vptr_p16_store4(V, ((int64_t)(i * 6 + j) * Mpad + row) * C + (int64_t)c4 * 4, v[j]);
  }
}

extern "C" int vptr_wino_out_in(const float* M36, const float* scale, const float* shift, const float* residual, float* y, void* V_next, int frames,
                                int H, int W, int C, int64_t Mpad, int relu, int act_after, int pad_mode, vptr_stream_t stream) {
  VPTR_CHECK(M36 && V_next && frames > 0 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0 && C > 0 && C % 16 == 0,
             "wino_out_in: needs H, W multiples of 4 and C a multiple of 16 (got H %d W %d C %d)", H, W, C);
  const int T = (H / 4) * (W / 4);
  VPTR_CHECK(T <= 16 && 16 % T == 0, "wino_out_in: (H/4)*(W/4) must divide 16 (got H %d W %d): use vptr_wino_out + vptr_wino_in", H, W);
  VPTR_CHECK(pad_mode >= 0 && pad_mode <= 2, "wino_out_in: pad_mode 0 zero / 1 reflect / 2 replicate");
  const int64_t rows = (int64_t)frames * T;
  VPTR_CHECK(Mpad >= rows, "wino_out_in: Mpad %lld < tile rows %lld", (long long)Mpad, (long long)rows);
  VPTR_CHECK(((reinterpret_cast<uintptr_t>(M36) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift) |
               reinterpret_cast<uintptr_t>(residual)) & 15) == 0 && (reinterpret_cast<uintptr_t>(V_next) & 63) == 0 && M36 != V_next,
             "wino_out_in: operands must be 16-byte aligned, V 64-byte aligned and distinct from M36");
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_out_in_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess) {
      vptr_set_error("wino_out_in: cannot reserve 64 KB of LDS");
      return -1;
    }
    attr_set = true;
  }
  const int upw = 256 / (16 * T);
  const int64_t units = (int64_t)frames * ((C / 4 + 15) / 16);
  wino_out_in_kernel<<<(unsigned)((units + upw - 1) / upw), 256, 65536, (hipStream_t)stream>>>(
      M36, scale, shift, residual, y, reinterpret_cast<unsigned char*>(V_next), frames, H, W, C / 4, Mpad, relu, act_after, pad_mode);
  VPTR_LAUNCH_CHECK();
  return 0;
}
