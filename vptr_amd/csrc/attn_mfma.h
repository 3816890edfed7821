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
};

// true when the MFMA kernels cover the geometry (otherwise the fp32 vector kernels of attn.hip run); VPTR_ATTN_MFMA=0 disables them
bool vptr_attn_mfma_ok(int Lq, int Lk, int C, int nh, int causal);
int vptr_attn_mfma_fwd(const float* q, const float* k, const float* v, const float* table, const int64_t* rel_index, float* o, const AmGeom& gm, int causal,
                       float p, const uint64_t* seed_dev, uint32_t site, int p16, hipStream_t st);
int vptr_attn_mfma_bwd(const float* q, const float* k, const float* v, const float* table, const int64_t* rel_index, const float* dout, float* dq, float* dk,
                       float* dv, float* dtable, const AmGeom& gm, int causal, float p, const uint64_t* seed_dev, uint32_t site, float dq_scale, int p16,
                       hipStream_t st);
