// Interface between attn.hip (entry points, fp32 vector kernels) and attn_mfma.hip (attention cores on the matrix units).
#pragma once
#include "common.h"

struct AmGeom {
  int mode;            // 0 window, 1 temporal
  int H, W, ws;        // window mode
  int Tq, Tk, HW;      // temporal mode
  int Lq, Lk;          // rows per problem
  int C, nh, hd;
  int groups;          // windows, or N * HW pixels
  // derived by the launchers of attn_mfma.hip (callers leave them zero)
  int ws_shift;        // log2(ws) when ws is a power of two, else -1
  int idx32;           // 1: groups * nh * Lq * Lk < 2^32 -- dropout element indices fit 32 bits (same hash, cheaper index arithmetic)
};

// true when the MFMA kernels cover the geometry AND the tensor size (32-bit row offsets: < 2^31 elements) -- otherwise the fp32 vector
// kernels of attn.hip run; VPTR_ATTN_MFMA=0 disables them
bool vptr_attn_mfma_ok(int Lq, int Lk, int C, int nh, int causal, int64_t groups /* problems: windows, or N * HW pixels */);
int vptr_attn_mfma_fwd(const float* q, const float* k, const float* v, const float* table, const int64_t* rel_index, float* o, const AmGeom& gm, int causal,
                       float p, const uint64_t* seed_dev, uint32_t site, int p16, hipStream_t st);
int vptr_attn_mfma_bwd(const float* q, const float* k, const float* v, const float* table, const int64_t* rel_index, const float* dout, float* dq, float* dk,
                       float* dv, float* dtable, const AmGeom& gm, int causal, float p, const uint64_t* seed_dev, uint32_t site, float dq_scale, int p16,
                       hipStream_t st);

// attn16.hip: problems of at most 16 x 16 tokens (4 x 4 windows, T <= 16) on the matrix units without LDS
struct A16Geom {
  int kind;          // 0: 4 x 4 windows of [B, H, W] frames; 1: temporal
  int H, W;          // kind 0
  int Tq, Tk, HW;    // kind 1
  int C, nh, hd;
  int Lq, Lk;        // rows per problem (<= 16)
  int nprob;         // windows, or N * HW pixels
  int causal;
};
bool vptr_attn16_ok(int kind, int Lq, int Lk, int C, int nh, int ws, int backward);
int vptr_attn16_fwd(const float* q, const float* k, const float* v, const float* table, const int64_t* rel_index, float* o, const A16Geom& g, float p,
                    const uint64_t* seed_dev, uint32_t site, int p16, hipStream_t st);
int vptr_attn16_bwd(const float* q, const float* k, const float* v, const float* table, const int64_t* rel_index, const float* dout, float* dq, float* dk,
                    float* dv, float* dtable, const A16Geom& g, float p, const uint64_t* seed_dev, uint32_t site, float dq_scale, int p16,
                    float* dtable_ws, int64_t ws_floats, hipStream_t st);
int64_t vptr_attn16_ws_floats(int nh);
