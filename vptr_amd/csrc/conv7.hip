// Direct 7x7 convolutions at the two ends of the ResNet auto-encoder (gfx950).
//   conv7_in : ReflectionPad2d(3) + Conv7x7(Cimg -> Cout) + folded BN + ReLU    (ResNetAutoEncoder.py:26-29)
//   conv7_out: ReflectionPad2d(3) + Conv7x7(Cin -> Cimg) + bias + Tanh/Sigmoid  (ResNetAutoEncoder.py:89-96)
// K is tiny on one side (Cimg = 1 or 3), so these are not MFMA-shaped: they run on the vector ALUs with the filter
// bank staged in LDS and channel-last (NHWC) stores/loads arranged so that a wave touches contiguous 1-4 KB runs.
#include "common.h"

__device__ __forceinline__ int reflect_idx(int c, int n) { return c < 0 ? -c : (c >= n ? 2 * n - 2 - c : c); }

// thread = (pixel, group of 16 output channels); block = 64 pixels x 4 groups (Cout must be 64)
__global__ __launch_bounds__(256) void conv7_in_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ y, int B, int Cimg, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [Cimg*49][64]
  const int tid = threadIdx.x;
  const int K = Cimg * 49;
  for (int i = tid; i < K * 64; i += 256) {
    const int kk = i >> 6, co = i & 63;
    sw[i] = w[co * K + kk];
  }
  __syncthreads();
  const int cg = tid & 3;
  const int64_t pix = (int64_t)blockIdx.x * 64 + (tid >> 2);
  const int64_t npix = (int64_t)B * H * W;
  if (pix >= npix) return;
  const int ox = (int)(pix % W), oy = (int)((pix / W) % H);
  const int b = (int)(pix / ((int64_t)W * H));
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int ci = 0; ci < Cimg; ++ci) {
    const float* xp = x + ((int64_t)b * Cimg + ci) * H * W;
    for (int ky = 0; ky < 7; ++ky) {
      const int iy = reflect_idx(oy + ky - 3, H);
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const int ix = reflect_idx(ox + kx - 3, W);
        const float xv = xp[iy * W + ix];
        const float4* wp = reinterpret_cast<const float4*>(sw + ((ci * 49 + ky * 7 + kx) << 6) + cg * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 wv = wp[j];
          acc[j * 4 + 0] += xv * wv.x; acc[j * 4 + 1] += xv * wv.y;
          acc[j * 4 + 2] += xv * wv.z; acc[j * 4 + 3] += xv * wv.w;
        }
      }
    }
  }
  float4* yp = reinterpret_cast<float4*>(y + pix * 64 + cg * 16);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = cg * 16 + j * 4;
    float4 o;
    if (scale) {
      o.x = fmaxf(acc[j * 4 + 0] * scale[c + 0] + shift[c + 0], 0.f);
      o.y = fmaxf(acc[j * 4 + 1] * scale[c + 1] + shift[c + 1], 0.f);
      o.z = fmaxf(acc[j * 4 + 2] * scale[c + 2] + shift[c + 2], 0.f);
      o.w = fmaxf(acc[j * 4 + 3] * scale[c + 3] + shift[c + 3], 0.f);
    } else {  // raw convolution output (train-mode BatchNorm is applied by its own passes)
      o = make_float4(acc[j * 4 + 0], acc[j * 4 + 1], acc[j * 4 + 2], acc[j * 4 + 3]);
    }
    yp[j] = o;
  }
}

// Single-channel images (KTH, MNIST), second version: lane = OUTPUT channel with its 49 taps in registers; the reflection-padded
// image rows sit in LDS and the 7 x 10 input window of four neighbouring outputs arrives as wave-uniform (broadcast) 16-byte reads:
// 21 LDS reads per 196 FMAs, no per-thread weight reads (the first version reads 4 weight vectors per tap and thread and is
// LDS-bound), every output pixel vector leaves as one 256-byte store.
__global__ __launch_bounds__(256, 4) void conv7_in_fwd2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ y, int B, int H, int W, int rows_per_wg, int planes) {
  extern __shared__ __attribute__((aligned(16))) float c7_sx[];   // [rows + 6][GW]: tile column c = padded column (ix = c - 3)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bands = (H + rows_per_wg - 1) / rows_per_wg;
  const int b = blockIdx.x / bands, band = blockIdx.x % bands;
  const int oy_a = band * rows_per_wg, oy_b = min(H, oy_a + rows_per_wg);
  const int GW = W + 8, trows = oy_b - oy_a + 6;
  const float* xb = x + (int64_t)b * H * W;
  for (int i = tid; i < trows * GW; i += 256) {
    const int t = i / GW, c = i - t * GW;
    c7_sx[i] = c < W + 6 ? xb[reflect_idx(oy_a - 3 + t, H) * W + reflect_idx(c - 3, W)] : 0.f;
  }
  float wr[7][7];
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) wr[ky][kx] = w[lane * 49 + ky * 7 + kx];
  const float sc = scale ? scale[lane] : 1.f, sh = scale ? shift[lane] : 0.f;
  __syncthreads();
  for (int oy = oy_a + wave; oy < oy_b; oy += 4) {
    float* yrow = y + ((int64_t)b * H + oy) * W * 64 + lane;
    for (int ox0 = 0; ox0 < W; ox0 += 4) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const float4* xp = reinterpret_cast<const float4*>(c7_sx + (oy - oy_a + ky) * GW + ox0);
        const float4 a0 = xp[0], a1 = xp[1], a2 = xp[2];
        const float xw[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] += wr[ky][kx] * xw[j + kx];
        __builtin_amdgcn_sched_barrier(0);   // keep the 21 window reads from all being hoisted to the top (84 live registers)
      }
      if (planes) {   // bf16 hi / lo planes X[pixel][c / 32][hi 32 | lo 32] for the plane-operand convolution that follows (same 256 bytes per pixel)
        unsigned char* prow = reinterpret_cast<unsigned char*>(y) + (((int64_t)b * H + oy) * W + ox0) * 256 + (lane >> 5) * 128 + (lane & 31) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint32_t h, l;
          vptr_split2(scale ? fmaxf(acc[j] * sc + sh, 0.f) : acc[j], 0.f, h, l);
          *reinterpret_cast<uint16_t*>(prow + j * 256) = (uint16_t)h;
          *reinterpret_cast<uint16_t*>(prow + j * 256 + 64) = (uint16_t)l;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) yrow[(int64_t)(ox0 + j) * 64] = scale ? fmaxf(acc[j] * sc + sh, 0.f) : acc[j];
      }
    }
  }
}

// Colour images (BAIR: Cimg = 3), same idea: lane = output channel.  The 49 taps of ONE input channel sit in registers at a time, so
// the accumulators cover a whole row chunk of up to 64 outputs (16 quads) and the channel loop is outermost per chunk: 49 tap loads per
// (row chunk, input channel) against 64 x 49 FMAs.  Input x [B][Cimg][H][W], weights [64][Cimg][7][7].
__global__ __launch_bounds__(256, 2) void conv7_in_fwd2c_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float* __restrict__ y, int B, int Cimg, int H, int W, int rows_per_wg, int planes) {
  extern __shared__ __attribute__((aligned(16))) float c7_sx[];   // [Cimg][rows + 6][GW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bands = (H + rows_per_wg - 1) / rows_per_wg;
  const int b = blockIdx.x / bands, band = blockIdx.x % bands;
  const int oy_a = band * rows_per_wg, oy_b = min(H, oy_a + rows_per_wg);
  const int GW = W + 8, trows = oy_b - oy_a + 6, tsz = trows * GW;
  const float* xb = x + (int64_t)b * Cimg * H * W;
  for (int i = tid; i < Cimg * tsz; i += 256) {
    const int ci = i / tsz, r = i - ci * tsz;
    const int t = r / GW, c = r - t * GW;
    c7_sx[i] = c < W + 6 ? xb[((int64_t)ci * H + reflect_idx(oy_a - 3 + t, H)) * W + reflect_idx(c - 3, W)] : 0.f;
  }
  const float sc = scale ? scale[lane] : 1.f, sh = scale ? shift[lane] : 0.f;
  __syncthreads();
  for (int oy = oy_a + wave; oy < oy_b; oy += 4) {
    for (int xc = 0; xc < W; xc += 64) {
      const int nq = min(16, (W - xc) >> 2);
      float acc[16][4];
#pragma unroll
      for (int qi = 0; qi < 16; ++qi)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[qi][j] = 0.f;
      for (int ci = 0; ci < Cimg; ++ci) {
        float wr[7][7];
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
          for (int kx = 0; kx < 7; ++kx) wr[ky][kx] = w[(lane * Cimg + ci) * 49 + ky * 7 + kx];
        const float* tile = c7_sx + ci * tsz + (oy - oy_a) * GW + xc;
#pragma unroll
        for (int qi = 0; qi < 16; ++qi) {
          if (qi < nq) {   // wave-uniform
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
              const float4* xp = reinterpret_cast<const float4*>(tile + ky * GW + 4 * qi);
              const float4 a0 = xp[0], a1 = xp[1], a2 = xp[2];
              const float xw[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
#pragma unroll
              for (int kx = 0; kx < 7; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[qi][j] += wr[ky][kx] * xw[j + kx];
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
      if (planes) {
        unsigned char* prow = reinterpret_cast<unsigned char*>(y) + (((int64_t)b * H + oy) * W + xc) * 256 + (lane >> 5) * 128 + (lane & 31) * 2;
#pragma unroll
        for (int qi = 0; qi < 16; ++qi) {
          if (qi < nq) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              uint32_t h, l;
              vptr_split2(scale ? fmaxf(acc[qi][j] * sc + sh, 0.f) : acc[qi][j], 0.f, h, l);
              *reinterpret_cast<uint16_t*>(prow + (qi * 4 + j) * 256) = (uint16_t)h;
              *reinterpret_cast<uint16_t*>(prow + (qi * 4 + j) * 256 + 64) = (uint16_t)l;
            }
          }
        }
      } else {
        float* yrow = y + (((int64_t)b * H + oy) * W + xc) * 64 + lane;
#pragma unroll
        for (int qi = 0; qi < 16; ++qi) {
          if (qi < nq) {
#pragma unroll
            for (int j = 0; j < 4; ++j) yrow[(int64_t)(qi * 4 + j) * 64] = scale ? fmaxf(acc[qi][j] * sc + sh, 0.f) : acc[qi][j];
          }
        }
      }
    }
  }
}

extern "C" int vptr_conv7_in_fwd(const float* x, const float* w, const float* scale, const float* shift, float* y, int B,
                                 int Cimg, int H, int W, int Cout, vptr_stream_t stream) {
  VPTR_CHECK(B > 0 && Cimg > 0 && H > 3 && W > 3, "conv7_in_fwd: bad arguments");
  VPTR_CHECK(Cout == 64, "conv7_in_fwd: Cout must be 64 (ngf of the reference encoder), got %d", Cout);
  if (Cimg == 1 && W % 4 == 0 && W <= 1024) {
    const int rpw = 8, bands = (H + rpw - 1) / rpw;
    conv7_in_fwd2_kernel<<<B * bands, 256, sizeof(float) * (rpw + 6) * (W + 8), (hipStream_t)stream>>>(x, w, scale, shift, y, B, H, W, rpw, 0);
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  if (Cimg <= 4 && W % 4 == 0 && W <= 512) {   // colour images (BAIR): the row-wide variant of the same kernel
    const int rpw = 8, bands = (H + rpw - 1) / rpw;
    conv7_in_fwd2c_kernel<<<B * bands, 256, sizeof(float) * Cimg * (rpw + 6) * (W + 8), (hipStream_t)stream>>>(x, w, scale, shift, y, B, Cimg, H, W, rpw, 0);
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  const size_t lds = sizeof(float) * Cimg * 49 * 64;
  VPTR_CHECK(lds <= 64 * 1024, "conv7_in_fwd: too many image channels");
  const int64_t npix = (int64_t)B * H * W;
  conv7_in_fwd_kernel<<<cdiv(npix, 64), 256, lds, (hipStream_t)stream>>>(x, w, scale, shift, y, B, Cimg, H, W);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// the same into the plane operand format of vptr_gemm(VPTR_A_CONV_PLANES): planes [(B*H*W + 1)][2][64] bf16 whose last (all-zero)
// row the caller zeroed once; single-channel images only (the second-generation kernel)
extern "C" int vptr_conv7_in_fwd_planes(const float* x, const float* w, const float* scale, const float* shift, void* planes, int B,
                                        int Cimg, int H, int W, int Cout, vptr_stream_t stream) {
  VPTR_CHECK(B > 0 && Cimg >= 1 && Cimg <= 4 && H > 3 && W > 3 && W % 4 == 0 && W <= (Cimg == 1 ? 1024 : 512) && Cout == 64 && planes,
             "conv7_in_fwd_planes: 1 - 4 image channels, Cout = 64, W %% 4 == 0");
  VPTR_CHECK((reinterpret_cast<uintptr_t>(planes) & 127) == 0, "conv7_in_fwd_planes: the plane buffer must be 128-byte aligned");
  const int rpw = 8, bands = (H + rpw - 1) / rpw;
  if (Cimg > 1) {
    conv7_in_fwd2c_kernel<<<B * bands, 256, sizeof(float) * Cimg * (rpw + 6) * (W + 8), (hipStream_t)stream>>>(x, w, scale, shift, reinterpret_cast<float*>(planes), B,
                                                                                                             Cimg, H, W, rpw, 1);
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  conv7_in_fwd2_kernel<<<B * bands, 256, sizeof(float) * (rpw + 6) * (W + 8), (hipStream_t)stream>>>(x, w, scale, shift, reinterpret_cast<float*>(planes), B, H, W,
                                                                                                   rpw, 1);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// One wave per 2 x 8 patch of output pixels, lane = input channel (Cin == 64): every input pixel vector is one coalesced
// 256-B load that feeds up to 7 x 2 outputs from registers; the 49 taps of the lane's channel sit in VGPRs (filter bank
// staged once per workgroup in LDS as [Cimg][49][64]); each output is finished by a wave reduction.
__global__ __launch_bounds__(256) void conv7_out_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int B,
                                                            int H, int W, int Cimg, int out_act) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [Cimg][49][64]
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < Cimg * 49 * 64; i += 256) {
    const int ci = i & 63, tap = (i >> 6) % 49, co = i / (64 * 49);
    sw[i] = w[(co * 64 + ci) * 49 + tap];
  }
  __syncthreads();
  const int strips_x = (W + 7) / 8, strips_y = (H + 1) / 2;
  const int64_t patch = (int64_t)blockIdx.x * 4 + (tid >> 6);
  if (patch >= (int64_t)B * strips_y * strips_x) return;
  const int sx = (int)(patch % strips_x), sy = (int)((patch / strips_x) % strips_y);
  const int b = (int)(patch / ((int64_t)strips_x * strips_y));
  const int x0 = sx * 8, y0 = sy * 2;
  for (int co = 0; co < Cimg; ++co) {
    float wr[49];
#pragma unroll
    for (int t = 0; t < 49; ++t) wr[t] = sw[(co * 49 + t) * 64 + lane];
    float acc[2][8];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[r][q] = 0.f;
#pragma unroll
    for (int ry = 0; ry < 8; ++ry) {      // input rows y0-3 .. y0+4 cover output rows y0 (ky = ry) and y0+1 (ky = ry-1)
      const int iy = reflect_idx(y0 + ry - 3, H);
      const float* xrow = x + (((int64_t)b * H + iy) * W) * 64 + lane;
#pragma unroll
      for (int cx = 0; cx < 14; ++cx) {   // input cols x0-3 .. x0+10
        const int ix = reflect_idx(min(x0 + cx - 3, W + 2), W);
        const float xv = xrow[(int64_t)ix * 64];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int kx = cx - q;
          if (kx >= 0 && kx < 7) {
            if (ry < 7) acc[0][q] += xv * wr[ry * 7 + kx];
            if (ry >= 1) acc[1][q] += xv * wr[(ry - 1) * 7 + kx];
          }
        }
      }
    }
    const float bco = bias[co];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float v = wave_sum(acc[r][q]) + bco;
        if (out_act == 1) v = tanhf(v);
        else if (out_act == 2) v = 1.f / (1.f + __expf(-v));
        const int oy = y0 + r, ox = x0 + q;
        if (lane == 0 && oy < H && ox < W) y[(((int64_t)b * Cimg + co) * H + oy) * W + ox] = v;
      }
  }
}

// One output channel (KTH, MNIST), second version: a wave owns a 4 x 16 patch of outputs (64 accumulators per lane = input channel),
// reads the 10 x 22 input pixel vectors of the patch once (3.4 coalesced loads per output instead of 7) and finishes all 64 outputs
// with ONE transposing reduction (63 shuffles: after step s a lane keeps the half of the values whose index bit equals its lane
// bit) instead of 64 full wave reductions; lane l then owns output l of the patch.
// Several output channels (BAIR: 3): gridDim.y = output channel -- the same patch is computed once per channel (the input vectors come
// from L2 the second and third time; keeping 3 x 64 accumulators per lane instead would not fit the register file).
__global__ __launch_bounds__(256, 2) void conv7_out_fwd2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W,
                                                             int out_act) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int px = W >> 4, py = H >> 2;
  const int64_t patch = (int64_t)blockIdx.x * 4 + (tid >> 6);
  if (patch >= (int64_t)B * py * px) return;
  const int co = blockIdx.y, Cimg = gridDim.y;
  w += (int64_t)co * 64 * 49;
  const int pxi = (int)(patch % px), pyi = (int)((patch / px) % py), b = (int)(patch / ((int64_t)px * py));
  const int x0 = pxi * 16, y0 = pyi * 4;
  float wr[7][7];
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) wr[ky][kx] = w[lane * 49 + ky * 7 + kx];
  float acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = 0.f;
  const float* xb = x + (int64_t)b * H * W * 64 + lane;
  int coff[22];                             // element offsets of the 22 input columns x0 - 3 .. x0 + 18 (reflected)
#pragma unroll
  for (int c = 0; c < 22; ++c) coff[c] = reflect_idx(x0 + c - 3, W) * 64;
#pragma unroll
  for (int r = 0; r < 10; ++r) {            // input rows y0 - 3 .. y0 + 6
    const float* xrow = xb + (int64_t)reflect_idx(y0 + r - 3, H) * W * 64;
    float xv[22];
#pragma unroll
    for (int c = 0; c < 22; ++c) xv[c] = xrow[coff[c]];
#pragma unroll
    for (int q = 0; q < 4; ++q) {           // output row y0 + q uses this input row with ky = r - q
      const int ky = r - q;
      if (ky < 0 || ky > 6) continue;
#pragma unroll
      for (int kx = 0; kx < 7; ++kx)
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[q * 16 + c] += xv[c + kx] * wr[ky][kx];
    }
    __builtin_amdgcn_sched_barrier(0);      // one input row at a time
  }
  // transposing reduction over the 64 lanes (= input channels)
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const bool up = (lane & s) != 0;
#pragma unroll
    for (int i = 0; i < s; ++i) {
      const float keep = up ? acc[i + s] : acc[i];
      const float send = up ? acc[i] : acc[i + s];
      acc[i] = keep + __shfl_xor(send, s, 64);
    }
  }
  float v = acc[0] + bias[co];
  if (out_act == 1) v = tanhf(v);
  else if (out_act == 2) v = 1.f / (1.f + __expf(-v));
  y[(((int64_t)b * Cimg + co) * H + y0 + (lane >> 4)) * W + x0 + (lane & 15)] = v;
}

extern "C" int vptr_conv7_out_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                                  int Cimg, int out_act, vptr_stream_t stream) {
  VPTR_CHECK(B > 0 && H > 3 && W > 3 && Cimg > 0, "conv7_out_fwd: bad arguments");
  VPTR_CHECK(Cin == 64, "conv7_out_fwd: Cin must be 64 (ngf of the reference decoder), got %d", Cin);
  if (Cimg <= 4 && H % 4 == 0 && W % 16 == 0) {
    const int64_t patches = (int64_t)B * (H / 4) * (W / 16);
    conv7_out_fwd2_kernel<<<dim3((unsigned)cdiv(patches, 4), Cimg), 256, 0, (hipStream_t)stream>>>(x, w, bias, y, B, H, W, out_act);
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  const size_t lds = sizeof(float) * Cimg * 49 * 64;
  VPTR_CHECK(lds <= 64 * 1024, "conv7_out_fwd: filter bank too large for LDS");
  const int64_t patches = (int64_t)B * ((H + 1) / 2) * ((W + 7) / 8);
  conv7_out_fwd_kernel<<<cdiv(patches, 4), 256, lds, (hipStream_t)stream>>>(x, w, bias, y, B, H, W, Cimg, out_act);
  VPTR_LAUNCH_CHECK();
  return 0;
}

__device__ __forceinline__ float out_act_grad(float dy, float y, int out_act) {
  if (out_act == 1) return dy * (1.f - y * y);
  if (out_act == 2) return dy * y * (1.f - y);
  return dy;
}

// backward-data of conv7_out: thread = (input pixel, 16-channel group).  With reflection padding an input pixel
// (iy, ix) is the image of up to 2x2 padded positions; each padded position py feeds outputs oy = py + 3 - ky.
__global__ __launch_bounds__(256) void conv7_out_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                 const float* __restrict__ w, float* __restrict__ dx, int B,
                                                                 int Cin, int H, int W, int Cimg, int out_act) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [Cimg][49][Cin]
  const int tid = threadIdx.x;
  for (int i = tid; i < Cimg * 49 * Cin; i += 256) {
    const int ci = i % Cin, tap = (i / Cin) % 49, co = i / (Cin * 49);
    sw[i] = w[(co * Cin + ci) * 49 + tap];
  }
  __syncthreads();
  const int ngrp = Cin >> 4;
  const int cg = tid % ngrp;
  const int64_t pix = (int64_t)blockIdx.x * (256 / ngrp) + tid / ngrp;
  const int64_t npix = (int64_t)B * H * W;
  if (pix >= npix || tid / ngrp >= 256 / ngrp) return;
  const int ix = (int)(pix % W), iy = (int)((pix / W) % H);
  const int b = (int)(pix / ((int64_t)W * H));
  int pys[3], pxs[3], npy = 1, npx = 1;
  pys[0] = iy; pxs[0] = ix;
  if (iy >= 1 && iy <= 3) pys[npy++] = -iy;
  if (iy <= H - 2 && iy >= H - 4) pys[npy++] = 2 * (H - 1) - iy;
  if (ix >= 1 && ix <= 3) pxs[npx++] = -ix;
  if (ix <= W - 2 && ix >= W - 4) pxs[npx++] = 2 * (W - 1) - ix;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int co = 0; co < Cimg; ++co) {
    const float* dyp = dy + ((int64_t)b * Cimg + co) * H * W;
    const float* yp = y + ((int64_t)b * Cimg + co) * H * W;
    for (int a = 0; a < npy; ++a)
      for (int ky = 0; ky < 7; ++ky) {
        const int oy = pys[a] + 3 - ky;
        if (oy < 0 || oy >= H) continue;
        for (int c = 0; c < npx; ++c)
          for (int kx = 0; kx < 7; ++kx) {
            const int ox = pxs[c] + 3 - kx;
            if (ox < 0 || ox >= W) continue;
            const float g = out_act_grad(dyp[oy * W + ox], yp[oy * W + ox], out_act);
            const float4* wp = reinterpret_cast<const float4*>(sw + (co * 49 + ky * 7 + kx) * Cin + cg * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float4 wv = wp[j];
              acc[j * 4 + 0] += g * wv.x; acc[j * 4 + 1] += g * wv.y;
              acc[j * 4 + 2] += g * wv.z; acc[j * 4 + 3] += g * wv.w;
            }
          }
      }
  }
  float4* dp = reinterpret_cast<float4*>(dx + pix * Cin + cg * 16);
#pragma unroll
  for (int j = 0; j < 4; ++j) dp[j] = make_float4(acc[j * 4 + 0], acc[j * 4 + 1], acc[j * 4 + 2], acc[j * 4 + 3]);
}

// One output channel, Cin = 64, second version: lane = input channel with its 49 taps in registers.  A wave produces one row of dx:
// for each pre-image row of the reflection padding (the row itself, plus its mirror image for the 3 rows next to an edge) it walks
// the PADDED columns four at a time, dpad[q] = sum_k w[k] g[q - k + 3] with the 7 x 10 gradient window as broadcast 16-byte LDS
// reads from a zero-padded tile of the whole frame, and folds the padded columns into the WT pixel accumulators of the row it
// keeps in registers (the fold indices are compile-time: the column loop is fully unrolled).  256-byte stores, no weight reads in
// the loop (the first version reads 4 weight vectors per tap and thread from LDS and is bound by that).
template <int WT>
__global__ __launch_bounds__(256, 2) void conv7_out_bwd_data2_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                     const float* __restrict__ w, float* __restrict__ dx, int B, int H,
                                                                     int out_act, int rows_per_wg, int Cimg) {
  extern __shared__ __attribute__((aligned(16))) float c7_sgd[];   // [Cimg][H + 12][GW]: row t = output row t - 6, column c = output column c - 8
  constexpr int PC = WT + 6, NG = (PC + 3) / 4, GW = 4 * NG + 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bands = (H + rows_per_wg - 1) / rows_per_wg;
  const int b = blockIdx.x / bands, band = blockIdx.x % bands;
  const int iy_a = band * rows_per_wg, iy_b = min(H, iy_a + rows_per_wg);
  const int tsz = (H + 12) * GW;
  for (int i = tid; i < Cimg * tsz; i += 256) {
    const int co = i / tsz, r = i - co * tsz;
    const int t = r / GW, c = r - t * GW;
    const int oy = t - 6, ox = c - 8;
    float g = 0.f;
    if (oy >= 0 && oy < H && ox >= 0 && ox < WT) {
      const int64_t o = (((int64_t)b * Cimg + co) * H + oy) * WT + ox;
      g = out_act_grad(dy[o], y[o], out_act);
    }
    c7_sgd[i] = g;
  }
  float wr[7][7];
  if (Cimg == 1) {
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) wr[ky][kx] = w[lane * 49 + ky * 7 + kx];
  }
  __syncthreads();
  for (int iy = iy_a + wave; iy < iy_b; iy += 4) {
    int qs[3], nq = 1;                       // pre-image rows (coordinates of the unpadded image; the padded row is q + 3)
    qs[0] = iy;
    if (iy >= 1 && iy <= 3) qs[nq++] = -iy;
    if (iy <= H - 2 && iy >= H - 4) qs[nq++] = 2 * (H - 1) - iy;
    float* drow = dx + ((int64_t)b * H + iy) * WT * 64 + lane;
#pragma unroll
    for (int half = 0; half < 2; ++half) {   // the padded row in two halves of NG / 2 groups (72 + 98 registers of accumulators and
      constexpr int HG = NG / 2;              // packed tap pairs do not fit otherwise)
      float dp[4 * HG];                       // dp[l] = padded column 4 * HG * half + l, summed over pre-image rows and taps
#pragma unroll
      for (int i = 0; i < 4 * HG; ++i) dp[i] = 0.f;
      for (int ca = 0; ca < Cimg * nq; ++ca) {
        const int co = ca / nq, a = ca - co * nq;
        if (Cimg > 1 && a == 0) {   // the taps of output channel co (weights [Cimg][64][7][7]); single-channel images keep theirs for the whole kernel
#pragma unroll
          for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) wr[ky][kx] = w[(co * 64 + lane) * 49 + ky * 7 + kx];
        }
        const float* grow = c7_sgd + co * tsz + (qs[a] + 3 + 6) * GW + 4 * HG * half;   // tile row of ky = 0 (output row q + 3); ky steps one row up
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {     // one tile row at a time: its 16-byte reads overlap between neighbouring groups and are shared
#pragma unroll
          for (int gi = 0; gi < HG; ++gi) {
            const float4* gp = reinterpret_cast<const float4*>(grow - ky * GW + 4 * gi);
            const float4 g0 = gp[0], g1 = gp[1], g2 = gp[2];
            const float gw[12] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w};
#pragma unroll
            for (int kx = 0; kx < 7; ++kx)
#pragma unroll
              for (int j = 0; j < 4; ++j) dp[4 * gi + j] += wr[ky][kx] * gw[j - kx + 8];
            __builtin_amdgcn_sched_barrier(0);   // (else the row's reads are all hoisted)
          }
        }
      }
      // fold the padded columns onto the pixels they are (mirror) images of (pixel ix = padded column ix + 3) and store
      if (half == 0) {
#pragma unroll
        for (int i = 1; i <= 3; ++i) dp[i + 3] += dp[3 - i];
#pragma unroll
        for (int ix = 0; ix + 3 < 4 * HG; ++ix) drow[(int64_t)ix * 64] = dp[ix + 3];
      } else {
        constexpr int P0 = 4 * HG;            // first padded column of this half
#pragma unroll
        for (int i = 1; i <= 3; ++i) dp[WT + 2 - i - P0] += dp[WT + 2 + i - P0];
#pragma unroll
        for (int ix = P0 - 3; ix < WT; ++ix) drow[(int64_t)ix * 64] = dp[ix + 3 - P0];
      }
    }
  }
}

extern "C" int vptr_conv7_out_bwd_data(const float* dy, const float* y, const float* w, float* dx, int B, int Cin, int H, int W,
                                       int Cimg, int out_act, vptr_stream_t stream) {
  VPTR_CHECK(B > 0 && Cin > 0 && Cin % 16 == 0 && 256 % (Cin / 16) == 0 && H > 6 && W > 6 && Cimg > 0,
             "conv7_out_bwd_data: bad arguments");
  const size_t lds2 = sizeof(float) * Cimg * (H + 12) * (4 * ((64 + 6 + 3) / 4) + 8);
  if (Cimg <= 4 && Cin == 64 && W == 64 && lds2 <= 80 * 1024) {
    const int rpw = 8, bands = (H + rpw - 1) / rpw;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)conv7_out_bwd_data2_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      attr = true;
    }
    conv7_out_bwd_data2_kernel<64><<<B * bands, 256, lds2, (hipStream_t)stream>>>(dy, y, w, dx, B, H, out_act, rpw, Cimg);
    VPTR_LAUNCH_CHECK();
    return 0;
  }
  const size_t lds = sizeof(float) * Cimg * 49 * Cin;
  VPTR_CHECK(lds <= 64 * 1024, "conv7_out_bwd_data: filter bank too large for LDS");
  const int64_t npix = (int64_t)B * H * W;
  const int ppb = 256 / (Cin / 16);
  conv7_out_bwd_data_kernel<<<cdiv(npix, ppb), 256, lds, (hipStream_t)stream>>>(dy, y, w, dx, B, Cin, H, W, Cimg, out_act);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// backward-weight of conv7_out: block = (8x8 output tile swept over a chunk of frames); thread = (ci, ky) owns the 7 taps
// (ky, 0..6) of its input channel.  The x tile with its 3-pixel reflected halo and the activation-gradient tile are staged
// in LDS; per output row a thread reads the 8 gradient values and the 14 x values of its channel ONCE and slides them over
// the 7 kx taps in registers (56 FMAs per 22 LDS reads; the first version did 2 LDS reads per FMA and was LDS-bound).
__global__ __launch_bounds__(512) void conv7_out_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                   const float* __restrict__ x, float* __restrict__ dw,
                                                                   float* __restrict__ db, int B, int Cin, int H, int W,
                                                                   int Cimg, int out_act, int frames_per_block) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sx = smem;                    // [14*14][64]
  float* sg = sx + 14 * 14 * 64;       // [Cimg][64]
  __shared__ float red[16];
  const int tid = threadIdx.x;
  const int ci = tid & 63, ky = tid >> 6;   // ky = 7: idle in the tap phase (512 = 64 x 8 threads, 7 kernel rows)
  const int tiles_x = W / 8, tiles_y = H / 8;
  const int ty = (blockIdx.x / tiles_x) % tiles_y, tx = blockIdx.x % tiles_x;
  const int b0 = blockIdx.y * frames_per_block, b1 = min(B, b0 + frames_per_block);
  float acc[3][7];
#pragma unroll
  for (int co = 0; co < 3; ++co)
#pragma unroll
    for (int t = 0; t < 7; ++t) acc[co][t] = 0.f;
  float bsum[3] = {0.f, 0.f, 0.f};
  for (int b = b0; b < b1; ++b) {
    __syncthreads();
    for (int i = tid; i < 14 * 14 * 64; i += 512) {
      const int c = i & 63, p = i >> 6;
      const int iy = reflect_idx(ty * 8 + p / 14 - 3, H), ix = reflect_idx(tx * 8 + p % 14 - 3, W);
      sx[i] = x[(((int64_t)b * H + iy) * W + ix) * 64 + c];
    }
    for (int i = tid; i < Cimg * 64; i += 512) {
      const int co = i >> 6, p = i & 63;
      const int64_t o = (((int64_t)b * Cimg + co) * H + ty * 8 + (p >> 3)) * W + tx * 8 + (p & 7);
      sg[i] = out_act_grad(dy[o], y[o], out_act);
    }
    __syncthreads();
    if (ky < 7) {
      for (int py = 0; py < 8; ++py) {
        float xr[14];
#pragma unroll
        for (int u = 0; u < 14; ++u) xr[u] = sx[((py + ky) * 14 + u) * 64 + ci];
#pragma unroll
        for (int co = 0; co < 3; ++co) {
          if (co < Cimg) {
            float gr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) gr[u] = sg[co * 64 + py * 8 + u];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
              float a = 0.f;
#pragma unroll
              for (int u = 0; u < 8; ++u) a += gr[u] * xr[u + kx];
              acc[co][kx] += a;
            }
          }
        }
      }
    }
    for (int co = 0; co < Cimg; ++co) {
      float gs = 0.f;
      if (tid < 64) gs = sg[co * 64 + tid];
      gs = block_sum(gs, red);
      bsum[co] += gs;
    }
  }
  if (ky < 7) {
#pragma unroll
    for (int co = 0; co < 3; ++co)
      if (co < Cimg) {
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) unsafeAtomicAdd(dw + ((int64_t)co * 64 + ci) * 49 + ky * 7 + kx, acc[co][kx]);
      }
  }
  if (tid == 0)
    for (int co = 0; co < Cimg; ++co) unsafeAtomicAdd(db + co, bsum[co]);
}

extern "C" int vptr_conv7_out_bwd_weight(const float* dy, const float* y, const float* x, float* dw, float* db, int B, int Cin,
                                         int H, int W, int Cimg, int out_act, vptr_stream_t stream) {
  VPTR_CHECK(B > 0 && Cin == 64 && H % 8 == 0 && W % 8 == 0 && Cimg >= 1 && Cimg <= 3, "conv7_out_bwd_weight: unsupported geometry");
  const size_t lds = sizeof(float) * (14 * 14 * Cin + Cimg * 64);
  // many frames per workgroup: each workgroup ends with Cimg*64*49 atomics onto the same addresses
  const int fpb = B >= 64 ? 16 : 4;
  dim3 grid((H / 8) * (W / 8), cdiv(B, fpb));
  conv7_out_bwd_weight_kernel<<<grid, 512, lds, (hipStream_t)stream>>>(dy, y, x, dw, db, B, Cin, H, W, Cimg, out_act, fpb);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// Second version of the same gradient (used when the caller provides a workspace): the mirror image of conv7_out_fwd.  A lane is
// an input channel and keeps ITS 49 taps of one output channel as accumulators; a wave walks rows of the reflection-PADDED input
// (every x pixel vector is one coalesced 256-byte load, no halo re-reads), four positions per step, and multiplies each x value
// with the 7 x 7 window of activation gradients around it -- wave-uniform values read from a zero-padded LDS tile as broadcast
// 16-byte reads (21 reads per 196 FMAs; the first version reads 22 LDS words per 56 FMAs and spends most of its time in the
// 2 M device-scope atomics of its 640 workgroups).  Workgroup = (frame, band of padded rows), 8 waves; their accumulators meet in
// LDS and leave as ONE [Cimg][49][64] partial per workgroup; conv7_dw_reduce_kernel adds the partials in a fixed order.
constexpr int C7_BANDS = 3;
__global__ __launch_bounds__(512, 4) void conv7_out_bwd_weight2_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                    const float* __restrict__ x, float* __restrict__ ws, float* __restrict__ db,
                                                                    int B, int H, int W, int Cimg, int out_act) {
  extern __shared__ __attribute__((aligned(16))) float c7_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / C7_BANDS, band = blockIdx.x % C7_BANDS;
  const int PR = H + 6, PC = W + 6, ngroups = (PC + 3) >> 2, GW = 4 * ngroups + 8;
  const int pr_a = band * PR / C7_BANDS, pr_b = (band + 1) * PR / C7_BANDS;
  const int trows = pr_b - pr_a + 6;                       // tile row t holds output row oy = pr_a - 6 + t; tile column c holds ox = c - 8
  const int oy_a = band * H / C7_BANDS, oy_b = (band + 1) * H / C7_BANDS;   // the rows whose gradient sum this workgroup owns (bias)
  float* sg = c7_smem;                                     // [trows][GW]
  for (int co = 0; co < Cimg; ++co) {
    __syncthreads();
    float bsum = 0.f;
    for (int i = tid; i < trows * GW; i += 512) {
      const int t = i / GW, c = i - t * GW;
      const int oy = pr_a - 6 + t, ox = c - 8;
      float g = 0.f;
      if (oy >= 0 && oy < H && ox >= 0 && ox < W) {
        const int64_t o = (((int64_t)b * Cimg + co) * H + oy) * W + ox;
        g = out_act_grad(dy[o], y[o], out_act);
        if (oy >= oy_a && oy < oy_b) bsum += g;
      }
      sg[i] = g;
    }
    bsum = wave_sum(bsum);
    if (lane == 0 && db) unsafeAtomicAdd(db + co, bsum);
    __syncthreads();
    float acc[7][7];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) acc[ky][kx] = 0.f;
    for (int pr = pr_a + wave; pr < pr_b; pr += 8) {
      const float* xr = x + ((int64_t)b * H + reflect_idx(pr - 3, H)) * W * 64 + lane;
      const float* grow = sg + (pr - pr_a + 6) * GW;       // tile row of ky = 0; ky steps one row up
      float xv[4], xn[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xn[j] = xr[(int64_t)reflect_idx(j - 3, W) * 64];
      for (int gi = 0; gi < ngroups; ++gi) {
        const int pc0 = 4 * gi;
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = pc0 + j < PC ? xn[j] : 0.f;
        if (gi + 1 < ngroups) {
#pragma unroll
          for (int j = 0; j < 4; ++j) xn[j] = xr[(int64_t)reflect_idx(min(pc0 + 4 + j, PC - 1) - 3, W) * 64];
        }
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
          const float4* gp = reinterpret_cast<const float4*>(grow - ky * GW + pc0);
          const float4 g0 = gp[0], g1 = gp[1], g2 = gp[2];
          const float gw[12] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w};
#pragma unroll
          for (int kx = 0; kx < 7; ++kx)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[ky][kx] += xv[j] * gw[j - kx + 8];   // ox = pc0 + j - kx  <->  tile column pc0 + j - kx + 8
          __builtin_amdgcn_sched_barrier(0);   // one window row at a time (else all 21 reads are hoisted: 84 more live registers)
        }
      }
    }
    // the 8 waves' accumulators -> one partial per workgroup: 8 -> 4 -> 1 through 4 x 12.5 KB of LDS (the gradient tile is dead)
    float* sacc = c7_smem;                                 // [4][49][64]
    __syncthreads();
    if (wave >= 4) {
#pragma unroll
      for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) sacc[((wave - 4) * 49 + ky * 7 + kx) * 64 + lane] = acc[ky][kx];
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
      for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) acc[ky][kx] += sacc[(wave * 49 + ky * 7 + kx) * 64 + lane];
    }
    __syncthreads();
    if (wave >= 1 && wave < 4) {
#pragma unroll
      for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) sacc[((wave - 1) * 49 + ky * 7 + kx) * 64 + lane] = acc[ky][kx];
    }
    __syncthreads();
    if (wave == 0) {
      float* o = ws + ((int64_t)blockIdx.x * Cimg + co) * 49 * 64 + lane;
#pragma unroll
      for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const int t = ky * 7 + kx;
          o[t * 64] = (acc[ky][kx] + sacc[t * 64 + lane]) + (sacc[(49 + t) * 64 + lane] + sacc[(98 + t) * 64 + lane]);
        }
    }
  }
}
// dw[(co * 64 + ci) * 49 + tap] += sum over workgroups of ws[wg][co][tap][ci]
__global__ __launch_bounds__(256) void conv7_dw_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nwg, int Cimg) {
  __shared__ float part[4][64];
  const int tap = blockIdx.x % 49, co = blockIdx.x / 49, ci = threadIdx.x & 63, qd = threadIdx.x >> 6;
  float s = 0.f;
  int g = qd;
  for (; g + 12 < nwg; g += 16) {
    float t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = ws[(((int64_t)(g + 4 * u) * Cimg + co) * 49 + tap) * 64 + ci];
    s += (t[0] + t[1]) + (t[2] + t[3]);
  }
  for (; g < nwg; g += 4) s += ws[(((int64_t)g * Cimg + co) * 49 + tap) * 64 + ci];
  part[qd][ci] = s;
  __syncthreads();
  if (qd == 0) dw[((int64_t)co * 64 + ci) * 49 + tap] += (part[0][ci] + part[1][ci]) + (part[2][ci] + part[3][ci]);
}
extern "C" int vptr_conv7_out_bwd_weight_workspace(int B, int Cimg) { return B * C7_BANDS * Cimg * 49 * 64; }
extern "C" int vptr_conv7_out_bwd_weight_ws(const float* dy, const float* y, const float* x, float* dw, float* db, int B, int Cin,
                                            int H, int W, int Cimg, int out_act, float* workspace, int workspace_floats,
                                            vptr_stream_t stream) {
  const int PR = H + 6, GW = 4 * ((W + 6 + 3) / 4) + 8;
  const size_t tile = sizeof(float) * ((PR + C7_BANDS - 1) / C7_BANDS + 1 + 6) * GW, red = sizeof(float) * 4 * 49 * 64;
  const size_t lds = tile > red ? tile : red;
  if (!workspace || workspace_floats < vptr_conv7_out_bwd_weight_workspace(B, Cimg) || lds > 150 * 1024 || H < 8 || W < 8)
    return vptr_conv7_out_bwd_weight(dy, y, x, dw, db, B, Cin, H, W, Cimg, out_act, stream);
  VPTR_CHECK(B > 0 && Cin == 64 && Cimg >= 1 && Cimg <= 3, "conv7_out_bwd_weight: unsupported geometry");
  VPTR_CHECK((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "conv7_out_bwd_weight: the workspace must be 16-byte aligned");
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)conv7_out_bwd_weight2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  conv7_out_bwd_weight2_kernel<<<B * C7_BANDS, 512, lds, (hipStream_t)stream>>>(dy, y, x, workspace, db, B, H, W, Cimg, out_act);
  VPTR_LAUNCH_CHECK();
  conv7_dw_reduce_kernel<<<49 * Cimg, 256, 0, (hipStream_t)stream>>>(workspace, dw, B * C7_BANDS, Cimg);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight-gradient helpers of the decoder's ConvTranspose2d + BatchNorm(eval) + ReLU layers.
// ---------------------------------------------------------------------------------------------------------------------
// im2col on NHWC with zero padding: out[(b,oy,ox)][(ky,kx,c)] = x[b, oy*s-p+ky, ox*s-p+kx, c].  The patch matrix is the
// k-strided B operand of the ConvTranspose2d weight-gradient GEMM  dW[ci][(ky,kx,co)] = sum_pix x[pix][ci] * P[pix][...].
__global__ __launch_bounds__(256) void im2col_nhwc_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int IH,
                                                          int IW, int C4, int OH, int OW, int KH, int KW, int stride, int pad,
                                                          int pad_mode) {
  const int64_t total = (int64_t)B * OH * OW * KH * KW * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t t = i / C4;
    const int kx = (int)(t % KW); t /= KW;
    const int ky = (int)(t % KH); t /= KH;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
    int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
    if (pad_mode == VPTR_PAD_REFLECT) { iy = reflect_idx(iy, IH); ix = reflect_idx(ix, IW); }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < IH && ix >= 0 && ix < IW) v = reinterpret_cast<const float4*>(x)[(((int64_t)b * IH + iy) * IW + ix) * C4 + c4];
    reinterpret_cast<float4*>(out)[i] = v;
  }
}
// the same patch matrix in the P16 operand format (C % 16 == 0): the B operand of the decoder's weight-gradient problems on the
// token-major P16 kernel
__global__ __launch_bounds__(256) void im2col_nhwc_p16_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int IH,
                                                              int IW, int C4, int OH, int OW, int KH, int KW, int stride, int pad,
                                                              int pad_mode) {
  const int64_t total = (int64_t)B * OH * OW * KH * KW * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t t = i / C4;
    const int kx = (int)(t % KW); t /= KW;
    const int ky = (int)(t % KH); t /= KH;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
    int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    bool ok = true;
    if (pad_mode == VPTR_PAD_REFLECT) { iy = reflect_idx(iy, IH); ix = reflect_idx(ix, IW); }
    else ok = iy >= 0 && iy < IH && ix >= 0 && ix < IW;
    if (ok) v = reinterpret_cast<const float4*>(x)[(((int64_t)b * IH + iy) * IW + ix) * C4 + c4];
    vptr_p16_store4(reinterpret_cast<unsigned char*>(out), i * 4, v);   // flat element index = ((pixel * taps + tap) * C + c): rows of taps * C
  }
}
extern "C" int vptr_im2col_nhwc_p16(const float* x, float* out_p16, int B, int IH, int IW, int C, int OH, int OW, int KH, int KW,
                                    int stride, int pad, int pad_mode, vptr_stream_t stream) {
  VPTR_CHECK(B > 0 && IH > 0 && IW > 0 && C > 0 && C % 16 == 0 && OH > 0 && OW > 0 && KH > 0 && KW > 0 && stride >= 1,
             "im2col_nhwc_p16: bad arguments (C must be a multiple of 16)");
  VPTR_CHECK(pad_mode == VPTR_PAD_ZERO || pad_mode == VPTR_PAD_REFLECT, "im2col_nhwc_p16: pad_mode must be zero or reflect");
  if (pad_mode == VPTR_PAD_REFLECT) VPTR_CHECK(pad < IH && pad < IW, "im2col_nhwc_p16: reflect padding needs pad < H, W");
  VPTR_CHECK((reinterpret_cast<uintptr_t>(out_p16) & 63) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "im2col_nhwc_p16: unaligned buffers");
  const int64_t total = (int64_t)B * OH * OW * KH * KW * (C / 4);
  const int blocks = (int)hmin64((total + 255) / 256, 16384);
  im2col_nhwc_p16_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(x, out_p16, B, IH, IW, C / 4, OH, OW, KH, KW, stride, pad, pad_mode);
  VPTR_LAUNCH_CHECK();
  return 0;
}
extern "C" int vptr_im2col_nhwc(const float* x, float* out, int B, int IH, int IW, int C, int OH, int OW, int KH, int KW,
                                int stride, int pad, int pad_mode, vptr_stream_t stream) {
  VPTR_CHECK(B > 0 && IH > 0 && IW > 0 && C > 0 && C % 4 == 0 && OH > 0 && OW > 0 && KH > 0 && KW > 0 && stride >= 1,
             "im2col_nhwc: bad arguments");
  VPTR_CHECK(pad_mode == VPTR_PAD_ZERO || pad_mode == VPTR_PAD_REFLECT, "im2col_nhwc: pad_mode must be zero or reflect");
  if (pad_mode == VPTR_PAD_REFLECT) VPTR_CHECK(pad < IH && pad < IW, "im2col_nhwc: reflect padding needs pad < H, W");
  const int64_t total = (int64_t)B * OH * OW * KH * KW * (C / 4);
  const int blocks = (int)hmin64((total + 255) / 256, 16384);
  im2col_nhwc_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(x, out, B, IH, IW, C / 4, OH, OW, KH, KW, stride, pad, pad_mode);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// eval-mode BatchNorm affine gradients behind a ReLU, from the layer OUTPUT y = relu(w*xhat + b):
//   db[c] += sum_{y>0} dy,   dw[c] += sum_{y>0} dy * xhat,  xhat = (y - b)/w.   Channel-last [rows, C]; same tiling as colsum.
__global__ __launch_bounds__(256) void bnrelu_bwd_params_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                const float* __restrict__ w, const float* __restrict__ b,
                                                                float* __restrict__ dw, float* __restrict__ db, int64_t rows,
                                                                int C4) {
  __shared__ float4 redw[8][32], redb[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c4 = blockIdx.x * 32 + tx;
  const int64_t r0 = (int64_t)blockIdx.y * 512, r1 = r0 + 512 < rows ? r0 + 512 : rows;
  float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ab = aw;
  if (c4 < C4) {
    const float4 wv = reinterpret_cast<const float4*>(w)[c4], bv = reinterpret_cast<const float4*>(b)[c4];
    const float4 iw = make_float4(1.f / wv.x, 1.f / wv.y, 1.f / wv.z, 1.f / wv.w);
    for (int64_t r = r0 + ty; r < r1; r += 8) {
      const float4 d = reinterpret_cast<const float4*>(dy)[r * C4 + c4];
      const float4 yv = reinterpret_cast<const float4*>(y)[r * C4 + c4];
      const float gx = yv.x > 0.f ? d.x : 0.f, gy = yv.y > 0.f ? d.y : 0.f, gz = yv.z > 0.f ? d.z : 0.f, gw = yv.w > 0.f ? d.w : 0.f;
      ab.x += gx; ab.y += gy; ab.z += gz; ab.w += gw;
      aw.x += gx * (yv.x - bv.x) * iw.x; aw.y += gy * (yv.y - bv.y) * iw.y;
      aw.z += gz * (yv.z - bv.z) * iw.z; aw.w += gw * (yv.w - bv.w) * iw.w;
    }
  }
  redw[ty][tx] = aw;
  redb[ty][tx] = ab;
  __syncthreads();
  if (ty == 0 && c4 < C4) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const float4 u = redw[k][tx], v = redb[k][tx];
      aw.x += u.x; aw.y += u.y; aw.z += u.z; aw.w += u.w;
      ab.x += v.x; ab.y += v.y; ab.z += v.z; ab.w += v.w;
    }
    unsafeAtomicAdd(dw + c4 * 4 + 0, aw.x); unsafeAtomicAdd(dw + c4 * 4 + 1, aw.y);
    unsafeAtomicAdd(dw + c4 * 4 + 2, aw.z); unsafeAtomicAdd(dw + c4 * 4 + 3, aw.w);
    unsafeAtomicAdd(db + c4 * 4 + 0, ab.x); unsafeAtomicAdd(db + c4 * 4 + 1, ab.y);
    unsafeAtomicAdd(db + c4 * 4 + 2, ab.z); unsafeAtomicAdd(db + c4 * 4 + 3, ab.w);
  }
}
extern "C" int vptr_bnrelu_bwd_params(const float* dy, const float* y, const float* w, const float* b, float* dw, float* db,
                                      int64_t rows, int C, vptr_stream_t stream) {
  VPTR_CHECK(rows > 0 && C > 0 && C % 4 == 0, "bnrelu_bwd_params: bad arguments");
  bnrelu_bwd_params_kernel<<<dim3(cdiv(C / 4, 32), cdiv(rows, 512)), 256, 0, (hipStream_t)stream>>>(dy, y, w, b, dw, db, rows, C / 4);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// Both of the above in one pass over (dy, y), for the decoder's backward: dx = [y > 0] dy * scale (the folded-BN + ReLU input
// gradient that feeds the ConvTranspose2d dgrad / wgrad) and the affine gradients.  One read of the two tensors instead of two
// (882 MB instead of 1.47 GB over the three up-sampling layers of the K64 decoder at N = 16).  A workgroup is cw = min(C/4, 64)
// float4 channel lanes x 256 / cw rows (no idle lanes for C = 64), sweeping 512 rows.
__global__ __launch_bounds__(256) void bnrelu_bwd_fused_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                               const float* __restrict__ scale, const float* __restrict__ w,
                                                               const float* __restrict__ b, float* __restrict__ dx, float* __restrict__ dw,
                                                               float* __restrict__ db, int64_t rows, int C4, int cw) {
  __shared__ float4 redw[256], redb[256];
  const int tx = threadIdx.x % cw, ty = threadIdx.x / cw, nty = 256 / cw;
  const int c4 = blockIdx.x * cw + tx;
  const int64_t r0 = (int64_t)blockIdx.y * 512, r1 = r0 + 512 < rows ? r0 + 512 : rows;
  float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ab = aw;
  if (c4 < C4) {
    const float4 wv = reinterpret_cast<const float4*>(w)[c4], bv = reinterpret_cast<const float4*>(b)[c4];
    const float4 sv = reinterpret_cast<const float4*>(scale)[c4];
    const float4 iw = make_float4(1.f / wv.x, 1.f / wv.y, 1.f / wv.z, 1.f / wv.w);
    auto one = [&](const int64_t r, const float4 d, const float4 yv) {
      const float gx = yv.x > 0.f ? d.x : 0.f, gy = yv.y > 0.f ? d.y : 0.f, gz = yv.z > 0.f ? d.z : 0.f, gw = yv.w > 0.f ? d.w : 0.f;
      reinterpret_cast<float4*>(dx)[r * C4 + c4] = make_float4(gx * sv.x, gy * sv.y, gz * sv.z, gw * sv.w);
      ab.x += gx; ab.y += gy; ab.z += gz; ab.w += gw;
      aw.x += gx * (yv.x - bv.x) * iw.x; aw.y += gy * (yv.y - bv.y) * iw.y;
      aw.z += gz * (yv.z - bv.z) * iw.z; aw.w += gw * (yv.w - bv.w) * iw.w;
    };
    int64_t r = r0 + ty;
    for (; r + 3 * nty < r1; r += 4 * nty) {   // 8 independent 16-byte loads in flight per thread
      float4 d[4], yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d[u] = reinterpret_cast<const float4*>(dy)[(r + u * nty) * C4 + c4];
        yv[u] = reinterpret_cast<const float4*>(y)[(r + u * nty) * C4 + c4];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(r + u * nty, d[u], yv[u]);
    }
    for (; r < r1; r += nty) one(r, reinterpret_cast<const float4*>(dy)[r * C4 + c4], reinterpret_cast<const float4*>(y)[r * C4 + c4]);
  }
  redw[threadIdx.x] = aw;
  redb[threadIdx.x] = ab;
  __syncthreads();
  if (ty == 0 && c4 < C4 && dw) {
    for (int k = 1; k < nty; ++k) {
      const float4 u = redw[k * cw + tx], v = redb[k * cw + tx];
      aw.x += u.x; aw.y += u.y; aw.z += u.z; aw.w += u.w;
      ab.x += v.x; ab.y += v.y; ab.z += v.z; ab.w += v.w;
    }
    unsafeAtomicAdd(dw + c4 * 4 + 0, aw.x); unsafeAtomicAdd(dw + c4 * 4 + 1, aw.y);
    unsafeAtomicAdd(dw + c4 * 4 + 2, aw.z); unsafeAtomicAdd(dw + c4 * 4 + 3, aw.w);
    unsafeAtomicAdd(db + c4 * 4 + 0, ab.x); unsafeAtomicAdd(db + c4 * 4 + 1, ab.y);
    unsafeAtomicAdd(db + c4 * 4 + 2, ab.z); unsafeAtomicAdd(db + c4 * 4 + 3, ab.w);
  }
}
extern "C" int vptr_bnrelu_bwd_fused(const float* dy, const float* y, const float* scale, const float* w, const float* b, float* dx,
                                     float* dw, float* db, int64_t rows, int C, vptr_stream_t stream) {
  VPTR_CHECK(rows > 0 && C > 0 && C % 4 == 0 && dy && y && scale && w && b && dx && dw && db, "bnrelu_bwd_fused: bad arguments");
  const int C4 = C / 4;
  int cw = 1;
  while (cw * 2 <= C4 && cw < 64) cw *= 2;   // a power of two that divides 256
  bnrelu_bwd_fused_kernel<<<dim3(cdiv(C4, cw), cdiv(rows, 512)), 256, 0, (hipStream_t)stream>>>(dy, y, scale, w, b, dx, dw, db, rows, C4, cw);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// adjoint of ReflectionPad2d(p): padded index q mirrors source reflect_idx(q - p, n); the pre-images of source j are
// q = j + p, q = p - j (1 <= j <= p) and q = 2n - 2 - j + p (1 <= n - 1 - j <= p).
__device__ __forceinline__ int fold_preimages(int j, int n, int p, int (&q)[3]) {
  int c = 0;
  q[c++] = j + p;
  if (j >= 1 && j <= p) q[c++] = p - j;
  if (n - 1 - j >= 1 && n - 1 - j <= p) q[c++] = 2 * n - 2 - j + p;
  return c;
}
__global__ __launch_bounds__(256) void reflect_fold_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int B, int H,
                                                           int W, int C4, int p) {
  const int Hp = H + 2 * p, Wp = W + 2 * p;
  const int64_t total = (int64_t)B * H * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4);
    const int64_t pix = i / C4;
    const int xw = (int)(pix % W), yh = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    int qy[3], qx[3];
    const int ny = fold_preimages(yh, H, p, qy), nx = fold_preimages(xw, W, p, qx);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int u = 0; u < ny; ++u)
      for (int v = 0; v < nx; ++v) {
        const float4 s = src[((b * Hp + qy[u]) * Wp + qx[v]) * C4 + c];
        a.x += s.x; a.y += s.y; a.z += s.z; a.w += s.w;
      }
    dst[i] = a;
  }
}
extern "C" int vptr_reflect_fold(const float* dxpad, float* dx, int B, int H, int W, int C, int pad, vptr_stream_t stream) {
  VPTR_CHECK(dxpad && dx && B > 0 && H > pad && W > pad && pad >= 1 && C > 0 && C % 4 == 0, "reflect_fold: bad arguments");
  const int64_t total = (int64_t)B * H * W * (C / 4);
  const int blocks = (int)hmin64((total + 255) / 256, 16384);
  reflect_fold_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const float4*>(dxpad), reinterpret_cast<float4*>(dx),
                                                               B, H, W, C / 4, pad);
  VPTR_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient of the 7x7 input convolution (Cout = 64): thread = (output channel, tap group of 4); a workgroup sweeps a
// chunk of pixels, every thread accumulating its taps in registers, and leaves Cimg*49 atomics per channel.
#define C7W_CHUNK 2048
__global__ __launch_bounds__(256) void conv7_in_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                  float* __restrict__ dw, int B, int Cimg, int H, int W) {
  const int co = threadIdx.x & 63, tg = threadIdx.x >> 6;  // taps tg, tg + 4, ... < 49 (13 or 12 per thread)
  const int64_t npix = (int64_t)B * H * W;
  const int64_t p0 = (int64_t)blockIdx.x * C7W_CHUNK, p1 = p0 + C7W_CHUNK < npix ? p0 + C7W_CHUNK : npix;
  for (int ci = 0; ci < Cimg; ++ci) {
    float acc[13];
#pragma unroll
    for (int t = 0; t < 13; ++t) acc[t] = 0.f;
    for (int64_t pix = p0; pix < p1; ++pix) {
      const int ox = (int)(pix % W), oy = (int)((pix / W) % H);
      const int64_t b = pix / ((int64_t)W * H);
      const float g = dy[pix * 64 + co];
      const float* xp = x + (b * Cimg + ci) * H * W;
#pragma unroll
      for (int t = 0; t < 13; ++t) {
        const int tap = min(tg + 4 * t, 48);
        const int ky = tap / 7, kx = tap - ky * 7;
        acc[t] += g * xp[reflect_idx(oy + ky - 3, H) * W + reflect_idx(ox + kx - 3, W)];
      }
    }
#pragma unroll
    for (int t = 0; t < 13; ++t) {
      const int tap = tg + 4 * t;
      if (tap < 49) unsafeAtomicAdd(dw + ((int64_t)co * Cimg + ci) * 49 + tap, acc[t]);
    }
  }
}
extern "C" int vptr_conv7_in_bwd_weight(const float* dy, const float* x, float* dw, int B, int Cimg, int H, int W, int Cout,
                                        vptr_stream_t stream) {
  VPTR_CHECK(dy && x && dw && B > 0 && Cimg > 0 && H > 3 && W > 3, "conv7_in_bwd_weight: bad arguments");
  VPTR_CHECK(Cout == 64, "conv7_in_bwd_weight: Cout must be 64 (ngf of the reference encoder), got %d", Cout);
  const int64_t npix = (int64_t)B * H * W;
  conv7_in_bwd_weight_kernel<<<(unsigned)cdiv(npix, C7W_CHUNK), 256, 0, (hipStream_t)stream>>>(dy, x, dw, B, Cimg, H, W);
  VPTR_LAUNCH_CHECK();
  return 0;
}
