/*
 * vptr_hip.h -- C ABI of libvptr_hip.so: hand-written HIP kernels (gfx950 / CDNA4) for the VPTR hot path.
 *
 * The reference (XiYe20/VPTR) is pure PyTorch and has no FFI layer of its own; its hot path is a
 * sequence of stock ATen op call sites.  Each entry point below
 * replaces one group of those call sites and cites them (paths relative to the reference root).
 * Python host code (vptr_amd/) binds these with ctypes and passes tensor.data_ptr() values and the
 * current HIP stream; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless stated otherwise; the library is
 *     borrow-only: it never allocates, frees or synchronises; all work is enqueued on `stream`.
 *   - activations are token-major, channel-last:  x[(n,t,h,w)][c]  ==  (N,T,H,W,C) contiguous.
 *   - return value: 0 on success, negative on error; vptr_last_error() gives a thread-local message.
 *   - dropout: p == 0 disables it; otherwise the mask of element i at a call site is
 *     hash(seed_dev[0], site, i) < keep, with seed_dev a DEVICE pointer to one uint64 (so that a
 *     captured hipGraph sees a fresh seed at every replay).
 */
#ifndef VPTR_HIP_H
#define VPTR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vptr_stream_t; /* hipStream_t */

int vptr_abi_version(void); /* 10 */
const char* vptr_last_error(void);

/* Run-to-run reproducibility (ABI 8).  The reference (cuDNN / cuBLAS defaults, train_NAR.py) is not bit-deterministic and neither is the
 * default path here: several backward kernels let workgroups meet in fp32 atomics.  vptr_set_deterministic(1) makes every LAUNCHER that
 * does so pick a geometry with ONE adder per destination (BatchNorm-type norm-act column sums, depthwise-conv weight gradients, row-table
 * sums, column sums); callers use vptr_sumsq_ws instead of vptr_sumsq and leave vptr_gemm_desc.frame_stats / vptr_dwconv3x3_fwd(frame_stats)
 * NULL.  With that, the stage-2 train step (NAR / FAR transformers; <= 16-token attention problems) is bit-reproducible on one device --
 * tests/test_11_deterministic_gpu.py.  Host-side switch, read when a launch is enqueued; returns the previous value. */
int vptr_set_deterministic(int on);
int vptr_get_deterministic(void);

/* Telemetry of the panel-synchronous grouped weight-gradient launch (vptr_gemm_grouped with token-major P16 operands):
 * out_dev[8] (device ints) = per XCD, how many workgroups gave up a bounded wait since the library was loaded.  All zero = every
 * participant was resident whenever somebody waited for it. */
int vptr_wgrad_sync_stats(int* out_dev, vptr_stream_t stream);
/* Library-owned device state (the ONE exception to "no global mutable state"): the persistent panel-synchronous launch keeps its
 * per-XCD arrival counters in a module-scope __device__ array (512 ints, zero at load, left zero by every launch) and assumes that all
 * of its workgroups are resident.  Contract: AT MOST ONE such launch in flight per device, on one stream, with nothing else competing
 * for the CUs -- which is what launches issued from one stream give.  vptr_gemm_grouped takes this path only for prototypes with
 * split_k == -1 (the caller vouches: equal token counts, exclusive use of the device for the duration); callers that run beside other
 * work (a second trainer on another stream, weight-gradient chunks beside RCCL kernels or beside a main-stream backward pass) pass
 * split_k >= 0 and get the plain grouped launch, which has no shared state.  Breaking the contract cannot hang or corrupt results
 * (destination adds are atomic, every wait is bounded) -- it costs bounded-spin time-outs, visible in vptr_wgrad_sync_stats.
 * The A/B switches of INTEGRATION.md's table (VPTR_GEMM_*, VPTR_WGRAD_*, VPTR_NORM_*, VPTR_DWCONV_GEN) are read ONCE, at the first launch
 * that consults them; VPTR_ATTN16 / VPTR_ATTN_MFMA / VPTR_ATTN16_FWD1 are read per launch (the tests switch attention families inside
 * one process). */

/* ------------------------------------------------------------------------------------------------
 * GEMM + fused epilogue on MFMA (bf16 inputs split from fp32 in the staging path, fp32 accumulate).
 *   D[M,N] = epilogue( op(A)[M,K] * op(B)[K,N] )
 * Replaces: F.linear / nn.Linear (MultiHeadAttentionRPE.py:543-545,687-688; VidHRFormer_modules.py:87-89,
 * 190-192; nn.MultiheadAttention in/out projections :79-84,185-187,204-205), the 1x1 convs of MlpDWBN
 * (:430,:436) and their autograd dgrad/wgrad; with a_mode = VPTR_A_CONV also Conv2d / ConvTranspose2d of
 * the ResNet auto-encoder (ResNetAutoEncoder.py:26-48,74-88,138,151) as implicit GEMM.
 *
 * precision: 1 = single bf16 MFMA pass (rel. error ~2e-3 per GEMM),
 *            3 = split-bf16 (hi*hi + hi*lo + lo*hi, fp32-class accuracy ~1e-5).
 * epilogue, in this order (null pointer / zero flag = skipped):
 *   v = acc; v *= colscale[n]; v += bias[n]; v *= alpha; v = act(v) (1 = exact-erf GELU, 2 = ReLU);
 *   v *= rowscale[(m / rs_div) % rs_mod]; dropout(p, site); v += residual[m, n]; if (act_after) v = relu(v);
 *   atomic: D[m,n] += v (split-K / gradient accumulation)  else  D[m,n] = v
 * ---------------------------------------------------------------------------------------------- */
enum { VPTR_A_KCONTIG = 0, VPTR_A_KSTRIDED = 1, VPTR_A_CONV = 2,
       VPTR_A_CONV_PLANES = 3 /* as VPTR_A_CONV, but A holds the NHWC input as bf16 hi / lo planes (vptr_split_planes) */,
       VPTR_A_P16 = 5         /* k-contiguous A[M][K] in the P16 plane format (see vptr_to_p16); needs b_mode = VPTR_B_P16 */,
       VPTR_A_P16T = 6        /* vptr_gemm_grouped only: token-major P16 G[K = tokens][M]; needs b_mode = VPTR_B_P16T */ };
enum { VPTR_B_KCONTIG = 0, VPTR_B_KSTRIDED = 1,
       VPTR_B_PLANES = 2 /* B[n][tap][c / 32][hi 32 | lo 32] bf16: plane form of the k-contiguous conv weight */,
       VPTR_B_P16 = 3    /* k-contiguous B[N][K] in the P16 plane format (weight planes, vptr_weight_planes) */,
       VPTR_B_P16T = 4   /* vptr_gemm_grouped only: token-major P16 X[K = tokens][N] */ };
enum { VPTR_ACT_NONE = 0, VPTR_ACT_GELU = 1, VPTR_ACT_RELU = 2, VPTR_ACT_LRELU = 3 /* LeakyReLU(0.2), VPTR_modules.py:70 */ };
enum { VPTR_PAD_ZERO = 0, VPTR_PAD_REFLECT = 1, VPTR_PAD_REPLICATE = 2 };

/* ABI 10: floats per frame of a frame-statistics buffer (vptr_gemm_desc.frame_stats, vptr_dwconv3x3_fwd(frame_stats), raw_stats of
 * vptr_norm_act_fwd / vptr_dwconv3x3_norm_fwd): [frames][VPTR_FRAME_STATS_STRIDE], sum at [0], sum of squares at [1], the rest unused.  One
 * 128-byte line per frame: the producers' fp32 atomics of different frames never meet in one L2 line (with the [frames][2] layout of
 * ABI <= 9 the 160 frames of a K64 step shared 10 lines and a launch's atomics serialised there: 120 of the 172 us of a fused
 * normalise + depthwise launch, profiles/r06_dwn_probe.log). */
#define VPTR_FRAME_STATS_STRIDE 32

typedef struct vptr_gemm_desc {
  const float* A; /* a_mode 0: [M, lda] (k contiguous); 1: [K, lda] (m contiguous); 2: NHWC image, see conv_* */
  const float* B; /* b_mode 0: [N, ldb] (k contiguous, i.e. nn.Linear weight); 1: [K, ldb] (n contiguous) */
  float* D;       /* [M, ldd] */
  float* Dpre;    /* optional [M, ldd]: value BEFORE the activation (after colscale/bias/alpha), saved for backward */
  int64_t lda, ldb, ldd;
  int M, N, K;
  int a_mode, b_mode;
  int precision; /* 1 or 3 */
  int split_k;   /* >= 1; > 1 forces atomic accumulation into D */
  int atomic;    /* 1: D += result (D must be initialised by the caller) */
  const float* colscale; /* [N] */
  const float* bias;     /* [N] */
  float alpha;
  int act;
  const float* rowscale; /* [rs_mod] */
  int rs_div, rs_mod;
  float dropout_p;
  const uint64_t* seed_dev;
  uint32_t site;
  const float* residual; /* [M, ldr] */
  int64_t ldr;
  int act_after; /* ReLU after the residual add */
  /* implicit-GEMM convolution (a_mode == VPTR_A_CONV): A is the NHWC input [frames, IH, IW, Cin];
     row m = (frame, oy, ox) of the [frames, OH, OW] output grid, k = (ky*KW + kx)*Cin + ci.
     transposed = 0: iy = oy*stride - pad + ky (pad_mode for out-of-range);
     transposed = 1: iy = (oy + pad - ky)/stride when divisible and in range, else zero
                     (gather form of ConvTranspose2d, ResNetAutoEncoder.py:74-88). */
  int conv_IH, conv_IW, conv_Cin, conv_OH, conv_OW, conv_KH, conv_KW, conv_stride, conv_pad, conv_pad_mode,
      conv_transposed;
  /* optional [M], a_mode == VPTR_A_KSTRIDED with the pipelined kernels only: a_rowsum[m] += alpha * sum_k op(A)[m, k], taken from
     the registers of the A staging path by the workgroups of column tile 0.  For a weight gradient dW = dY^T . X this is
     the bias gradient (column sums of dY), which then needs no pass of its own. */
  float* a_rowsum;
  /* Several same-shaped products in ONE launch (0 / 1 = plain GEMM; at most 3; split_k = 1, no conv operand).  A K = 528
     projection alone is 240 tiles on 256 CUs and half of its time is prologue + epilogue, so the q/k/v projections of an
     attention (MultiHeadAttentionRPE.py:543-545; nn.MultiheadAttention's in_proj) and their input gradients are issued as:
       batch  = b: b independent problems; member i > 0 reads A_x<i>, B_x<i> and writes D_x<i> with bias_x<i>, alpha_x<i>
                (lda/ldb/ldd, shapes, modes, act and dropout settings are shared; no Dpre/residual/atomic);
       ksegs  = s: ONE output, D = epilogue( sum_i op(A_i)[M,K] * op(B_i)[K,N] ), segment i > 0 reads A_x<i>, B_x<i>
                (dX = dQ.Wq + dK.Wk + dV.Wv when q, k and v were projected from the same tensor).
     batch and ksegs are mutually exclusive. */
  int batch, ksegs;
  const float *A_x1, *A_x2; /* scalar fields, not arrays: the kernels select them with uniform compares */
  const float *B_x1, *B_x2;
  float *D_x1, *D_x2;
  const float *bias_x1, *bias_x2;
  float alpha_x1, alpha_x2;
  /* plane-operand kernels only (a_mode = VPTR_A_CONV_PLANES / VPTR_A_PLANES): additionally (or, with D = NULL, instead) write
     the result as bf16 hi / lo planes [M (+1)][ceil(N/32)][64] -- the operand format of the NEXT plane GEMM, so no
     vptr_split_planes pass is needed in between.  Pad channels / the extra row are never written (keep the buffer zeroed). */
  void* D_planes;
  /* a_mode = VPTR_A_P16 only: != 0 writes D (and D_x1 / D_x2) in the P16 plane format instead of fp32 (the operand format of the
     GEMM that consumes it; Dpre stays fp32).  Needs N and ldd multiples of 16 and 64-byte aligned outputs. */
  int d_p16;
  /* a_mode = VPTR_A_P16 only: activation-GRADIENT epilogue.  Non-NULL: an [M, N] fp32 tensor (row pitch ldd) of saved
     pre-activations h; D = alpha * acc * act'(h) * dropout(seed, site, row * N + col) -- the input gradient of a Linear whose
     consumer was act(.) + dropout (linear2's dX fused with linear1's activation backward, VidHRFormer_modules.py:89,192), written
     fp32 or P16.  Not combinable with bias / colscale / Dpre / rowscale / residual / act_after / atomic / batch / ksegs. */
  const float* act_grad_src;
  /* a_mode = VPTR_A_P16 only: per-frame statistics of the OUTPUT for the LayerNorm((F,H,W)) that consumes it (MlpDWBN,
     VidHRFormer_modules.py:397-419): frame_stats[S f] += sum, frame_stats[S f + 1] += sum of squares (S = VPTR_FRAME_STATS_STRIDE) of the rows
     [f * frame_rows, (f + 1) * frame_rows) of D (fp32 atomics into a zeroed [M / frame_rows][VPTR_FRAME_STATS_STRIDE] buffer; frame_rows % 64 == 0).
     vptr_norm_act_fwd(raw_stats = ...) turns them into mean / rstd -- no separate statistics pass over D. */
  float* frame_stats;
  int frame_rows;
  /* vptr_gemm_grouped with token-major P16 operands only: != 0 stores the product TRANSPOSED, D[n * ldd + m] (+)= alpha * acc[m][n]
     (D is [N, ldd]), and a_rowsum then receives the column sums of the B operand: a_rowsum[n] += alpha * sum_t B[t][n].  The host
     uses it to put the 176-wide tile side on the dimension it divides: dW[528][2112] = dY^T . X is computed as X^T . dY (A = X,
     B = dY: 17 x 3 tiles instead of 5 x 12 with one eighth-full row tile in five) and lands in dW's own layout; the bias gradient
     (column sums of dY) rides in a wave row of the last row tile that lies beyond M (needs >= 32 such rows: M % 128 in 1..96). */
  int d_transposed;
  /* fp32-staged kernels (a_mode 0 .. 2), fragment-layout epilogues only (no residual / Dpre): d_row_w > 0 sends output row m to row
     m + d_row_w * (m / d_row_w) + d_row_off of D (row pitch ldd).  With ldd = 2 * Cout, d_row_w = IW, d_row_off = py * IW and
     D offset by px * Cout this interleaves the four parity classes of a stride-2 ConvTranspose2d (ResNetAutoEncoder.py:74-88) into
     the NHWC output: each class is an ordinary stride-1 gather over the INPUT grid with 1, 2, 2 or 4 taps instead of a 9-tap
     gather form in which three quarters of the products are zeros. */
  int d_row_w, d_row_off;
  /* a_mode = VPTR_A_P16, plain launches (bias / alpha only) with batch > 1: bit i set = member i ACCUMULATES, D_i += result (its own
     output is its residual).  The input gradients of the encoder memory arrive from the encoder-decoder attention of every decoder
     block (VidHRFormer_modules.py:199-206): with this flag they are summed by the GEMMs instead of one autograd `add` per block. */
  int batch_accum;
  /* ABI 10.  a_mode = VPTR_A_P16, plain launches (alpha / shared bias, fp32 or P16 output): batch = b (ANY b >= 1) members at constant
     strides -- member i reads A + i * batch_stride_a, B + i * batch_stride_b and writes D + i * batch_stride_d (strides in fp32
     elements, i.e. 4-byte units of the P16 image; multiples of 16).  Set batch_stride_d != 0 to select this form; A_x*, B_x*, D_x*,
     bias_x*, alpha_x* are then ignored, every member shares bias / alpha.  First user: the 36 Winograd-domain products of a frozen
     3 x 3 convolution (vptr_wino_in / vptr_wino_out below). */
  int64_t batch_stride_a, batch_stride_b, batch_stride_d;
} vptr_gemm_desc;

int vptr_gemm(const vptr_gemm_desc* desc, vptr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * ABI 10.  Winograd F(4x4, 3x3) for FROZEN stride-1 3x3 convolutions (the 18 ResnetBlock convolutions of VPTREnc under
 * train_NAR.py:54-56 / ResNetAutoEncoder.py:127-151: 92 % of the encoder's FLOPs; weights never change in stage 2 / inference).
 *   y = A^T [ U . (B^T d B) ] A per 4 x 4 output tile, U = G g G^T made once per weight version by the host (P16, [36][Cout][Cin]):
 *   vptr_wino_in   x  [frames, H, W, C] fp32 NHWC  ->  V [36][Mpad][C] P16, row = (frame, tile_y, tile_x), the convolution's padding
 *                  (pad_mode 0 zero / 1 reflect / 2 replicate, one pixel) folded into the 6 x 6 patch of every tile
 *   vptr_gemm      a_mode = VPTR_A_P16, b_mode = VPTR_B_P16, batch = 36, batch_stride_a = batch_stride_d = Mpad * C, batch_stride_b = C * C
 *   vptr_wino_out  M36 [36][Mpad][C] fp32 -> y [frames, H, W, C]: v = (A^T m A) * scale[c] + shift[c]; relu; + residual; act_after
 *                  (the folded eval-mode BatchNorm, ReLU and skip connection of ResnetBlock, ResNetAutoEncoder.py:153-157); y may be the
 *                  residual buffer itself.
 * H, W multiples of 4; C a multiple of 16; Mpad >= frames * (H/4) * (W/4) (rows beyond that are never touched: keep V zeroed).
 * 4x fewer MFMA passes than the implicit GEMM; fp32 transforms, relative error 6e-5 after nine blocks (tools/winograd_numerics.py). */
int vptr_wino_in(const float* x, void* V, int frames, int H, int W, int C, int64_t Mpad, int pad_mode, vptr_stream_t stream);
int vptr_wino_out(const float* M36, const float* scale, const float* shift, const float* residual, float* y, int frames, int H, int W, int C,
                  int64_t Mpad, int relu, int act_after, vptr_stream_t stream);
/* vptr_wino_out followed by vptr_wino_in of the NEXT convolution in one pass: the activated map of a (frame, 64-channel) unit stays in LDS
 * between the two transforms, so a ResnetBlock's intermediate map never reaches HBM.  y may be NULL (the map is not needed outside) or the
 * residual buffer.  (H/4) * (W/4) must divide 16 (4 x 4 ... 16 x 16 maps); V_next != M36. */
int vptr_wino_out_in(const float* M36, const float* scale, const float* shift, const float* residual, float* y, void* V_next, int frames, int H,
                     int W, int C, int64_t Mpad, int relu, int act_after, int pad_mode, vptr_stream_t stream);

/* "Convert once" operand format of the a_mode = VPTR_A_CONV_PLANES path (first user: the frozen VPTREnc of the NAR / FAR
 * steps, ResNetAutoEncoder.py:127-151 under train_NAR.py:54-56): x [rows, C] fp32 -> planes [(rows + 1), ceil(C/32), 64] bf16,
 * block (row, cb) = hi(x[row, 32 cb .. +31]) then lo(...), channels beyond C and the extra last row zero; x = hi + lo to
 * 2^-17 relative.  The GEMM stages such operands with global_load_lds (no conversion, no LDS stores in its main loop).
 * planes must be 128-byte aligned. */
int vptr_split_planes(const float* x, void* planes, int64_t rows, int C, vptr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * P16: the "convert once" operand format of every nn.Linear forward / input-gradient / weight-gradient GEMM of the
 * transformers (MultiHeadAttentionRPE.py:543-545,687-688; VidHRFormer_modules.py:79-89,185-192,204-205,430,436 and their
 * autograd).  A P16 tensor has the shape, pitch and byte size of the fp32 [rows, C] tensor it stands for (C % 16 == 0); each
 * 16-channel granule (64 bytes) holds 16 bf16 `hi` then 16 bf16 `lo` with x = hi + lo to 2^-17 relative.  Producers write it
 * directly (the `p16` flags of vptr_layernorm_fwd, the attention cores, vptr_norm_act_fwd/bwd, vptr_act_bwd and
 * vptr_gemm_desc.d_p16); the GEMMs stage it with global_load_lds (no conversion and no LDS stores in their main loops).
 * vptr_to_p16 is the stand-alone conversion pass for tensors no producer kernel emits in this format. */
int vptr_to_p16(const float* x, void* out, int64_t rows, int C, vptr_stream_t stream);

/* Weight planes for a table of nn.Linear / 1x1-conv weights in ONE launch (after every optimizer step):
 *   W [N, ldw] fp32 (K columns used)  ->  Wp [N, K] P16 (B operand of the forward GEMM, b_mode = VPTR_B_P16)
 *                                         WT [K, N] P16 (B operand of the input-gradient GEMM dX = dY . W)
 * N % 16 == 0, K % 16 == 0.  tile_start_dev[g] = first 32 x 32 tile of entry g (ceil(N/32) * ceil(K/32) tiles each). */
typedef struct vptr_wplane_entry {
  const float* W;
  void* Wp;
  void* WT;
  int64_t ldw;
  int N, K;
} vptr_wplane_entry;
int vptr_weight_planes(const vptr_wplane_entry* table_dev, const int* tile_start_dev, int count, int total_tiles,
                       vptr_stream_t stream);

/* Column width of the output tile vptr_gemm / vptr_gemm_grouped use for an N-column problem (64, 128 or 176);
 * the row height is 128 (vptr_gemm_grouped: see proto->split_k below).  Host-side helper for building vptr_gemm_grouped's tile table. */
int vptr_gemm_tile_cols(int N);

/* Grouped GEMM: `count` independent problems in ONE launch, each tile running its problem's full K range (no split-K).
 * Used for the weight gradients of a whole backward pass (dW = dY^T . X of every nn.Linear, train_NAR.py:101): the
 * per-layer calls are recorded and flushed together, because each one alone (12-60 output tiles, K = all tokens)
 * cannot fill 256 CUs without ~30 K-splits that each pay a prologue and an atomic epilogue.
 * With a_mode = VPTR_A_P16T / b_mode = VPTR_B_P16T both operands are token-major P16 tensors (A = dY [tokens, M] with lda,
 * B = X [tokens, N] with ldb, K = tokens): the convert-once weight-gradient kernel (fragments by ds_read_b64_tr_b16, tile
 * 128 x 176 for every N); a_rowsum then comes out of an MFMA product with a vector of ones.
 *   proto          host copy of any member: a_mode / b_mode (must be k-strided x k-strided), precision and the tile
 *                  class of N (vptr_gemm_tile_cols) are taken from it and must be common to the group
 *   descs_dev      DEVICE array [count] of descriptors (split_k ignored; atomic = 1 accumulates into D)
 *   tile_start_dev DEVICE int[count]: first tile index of problem g; problem g owns
 *                  ceil(M/TR) * ceil(N/tile_cols) consecutive tiles, TR = tile rows (128 unless proto->split_k says otherwise);
 *                  total_tiles = sum over the group
 * Token-major P16 groups: proto->split_k selects the launch geometry (the caller counted the tiles accordingly; atomic = 1):
 *      1   plain launch, 128 x 176 tiles, two workgroups per CU
 *     -1   panel-synchronous persistent launch, 128 x 176 tiles (needs equal token counts and >= 1024 tiles; see the contract above)
 *     -2 / -3   256 x 176 tiles, one workgroup per CU (1.47x the flops per staged operand byte): persistent (>= 512 tiles) / plain
 *     -4 / -5   192 x 176 tiles, three stages, one workgroup per CU: persistent / plain (measured slower; tools/rejected/README.md)
 * vptr_amd.ops.plan_wgrad_launches is the worked example of cutting a backward pass's problems into such launches. */
int vptr_gemm_grouped(const vptr_gemm_desc* proto, const vptr_gemm_desc* descs_dev, const int* tile_start_dev, int count,
                      int total_tiles, vptr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the channel dim (nn.LayerNorm(C), VidHRFormer_modules.py:44-48,56,137-161; VidHRFormer.py:24,26).
 *   y = LN(x); optional y2 = y + tab[((row / tab_div) % tab_mod), :]   (positional adds of :79,176,185,204)
 * ---------------------------------------------------------------------------------------------- */
int vptr_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* y2, const float* tab,
                       int tab_div, int tab_mod, float* mean, float* rstd, int rows, int C, float eps,
                       int p16 /* != 0: y and y2 are written in the P16 plane format (they only feed GEMMs) */, vptr_stream_t stream);
/* dx = LN'(dy + dy2) + dx_add; dgamma/dbeta are ACCUMULATED (+=) with atomics; dy2 and dx_add may be null.
 * dx_add [rows, C] is the gradient that reaches x through the residual connection around the pre-norm sub-layer
 * (x + f(LN(x)), VidHRFormer_modules.py:68-93): added here, it needs no accumulation pass of its own. */
int vptr_layernorm_bwd(const float* dy, const float* dy2, const float* x, const float* gamma, const float* mean,
                       const float* rstd, float* dx, float* dgamma, float* dbeta, int rows, int C, const float* dx_add,
                       vptr_stream_t stream);
/* The same with the parameter gradients DEFERRED: dx is written, the workgroups' gamma / beta sums go to
 * partials[vptr_layernorm_bwd_partials(rows, C)][2][C] (caller-owned, 16-byte aligned) instead of 2 C atomics per workgroup;
 * one vptr_partial_reduce launch at the end of the backward pass adds the partial rows of EVERY deferred call into their
 * destinations.  vptr_layernorm_bwd_partials returns 0 for geometries without a deferred variant (use vptr_layernorm_bwd). */
int vptr_layernorm_bwd_deferred(const float* dy, const float* dy2, const float* x, const float* gamma, const float* mean,
                                const float* rstd, float* dx, int rows, int C, const float* dx_add, float* partials,
                                vptr_stream_t stream);
int vptr_layernorm_bwd_partials(int rows, int C);   /* a plain number, not an error code */
typedef struct vptr_reduce_entry {
  const float* part;   /* [nparts][2][C]; with dst1 == null [nparts][C] */
  float* dst0;         /* [C] += sum over p of part[p][0][:] */
  float* dst1;         /* [C] += sum over p of part[p][1][:], or null */
  int nparts, C;
} vptr_reduce_entry;
/* unique_dst != 0: no two entries name overlapping destinations -> plain read-add-write (16 bytes wide where the destination
 * is aligned) instead of one atomic per element (13 M atomics for the K64 step's LayerNorm((F,H,W)) affines) */
int vptr_partial_reduce(const vptr_reduce_entry* table_dev, int count, int max_C /* of the entries (every C % 4 == 0, 16-byte aligned parts) */,
                        int unique_dst, vptr_stream_t stream);

/* out[(row / div) % mod, :] += src[row, :]   (gradient of a row-broadcast table; out must be zeroed by the caller) */
int vptr_rowmod_sum(const float* src, float* out, int rows, int C, int div, int mod, vptr_stream_t stream);
/* out[c] += sum_rows src[row, c]  (bias gradients) */
int vptr_colsum(const float* src, float* out, int rows, int C, vptr_stream_t stream);
/* y = x + tab[((row / div) % mod), :] */
int vptr_add_rowtab(const float* x, const float* tab, float* y, int rows, int C, int div, int mod, vptr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Local-window attention core with relative-position bias
 * (MultiHeadAttentionRPE.py:586-590,623,629-650,677-682 + window partition VidHRFormer_modules.py:497-525,
 *  done by index arithmetic).  q (pre-scaled), k, v, o: [B*H*W, C] token-major, B = N*T frames.
 *  bias_table [(2ws-1)^2, nh] or null; rel_index int64 [ws*ws, ws*ws].
 *  p16 (all attention cores): != 0 writes the outputs that only feed GEMMs -- o in the forward calls, dq / dk / dv in the
 *  backward calls -- in the P16 plane format (C % 16 == 0) instead of fp32; inputs are always fp32.
 * ---------------------------------------------------------------------------------------------- */
int vptr_winattn_fwd(const float* q, const float* k, const float* v, const float* bias_table, const int64_t* rel_index,
                     float* o, int B, int H, int W, int C, int nh, int ws, float dropout_p, const uint64_t* seed_dev,
                     uint32_t site, int p16, vptr_stream_t stream);
/* dq,dk,dv are written; dbias_table is ACCUMULATED (may be null).  dq is multiplied by dq_scale on the way out: with
 * q = alpha * (x Wq^T + bq) (the head_dim^-0.5 of MultiHeadAttentionRPE.py:586 / nn.MultiheadAttention), dq_scale = alpha
 * makes dq the gradient of the UNSCALED projection, so dQ, dK, dV can feed one K-segmented input-gradient GEMM
 * (vptr_gemm_desc.ksegs); 1.0 = plain gradient w.r.t. q. */
int vptr_winattn_bwd(const float* q, const float* k, const float* v, const float* bias_table, const int64_t* rel_index,
                     const float* dout, float* dq, float* dk, float* dv, float* dbias_table, int B, int H, int W, int C,
                     int nh, int ws, float dropout_p, const uint64_t* seed_dev, uint32_t site, float dq_scale,
                     int p16, vptr_stream_t stream);
/* The same with a caller-owned scratch buffer of vptr_winattn_bwd_workspace(nh) floats (16-byte aligned, contents irrelevant
 * before and after the call): the MFMA kernel for 4 x 4 windows then leaves its bias-table gradient as per-workgroup partial
 * sums that a small second kernel adds into dbias_table in a fixed order -- no atomics onto the table's 49 * nh words
 * (which cost as much as the rest of the kernel) and a run-to-run reproducible table gradient.  workspace == null, a
 * workspace that is too small, or a geometry that other kernels serve: behaves like vptr_winattn_bwd. */
int vptr_winattn_bwd_ws(const float* q, const float* k, const float* v, const float* bias_table, const int64_t* rel_index,
                        const float* dout, float* dq, float* dk, float* dv, float* dbias_table, int B, int H, int W, int C,
                        int nh, int ws, float dropout_p, const uint64_t* seed_dev, uint32_t site, float dq_scale,
                        int p16, float* workspace, int workspace_floats, vptr_stream_t stream);
int vptr_winattn_bwd_workspace(int nh);   /* floats; a plain number, not an error code */

/* ------------------------------------------------------------------------------------------------
 * Temporal attention core (nn.MultiheadAttention slow path, VidHRFormer_modules.py:74-84,183-187,199-206):
 * for every (n, pixel, head) attend over time.  q,o: [(n,tq,p), C]; k,v: [(n,tk,p), C]; q pre-scaled.
 * causal != 0 masks j > i (FAR, :76-82).
 * ---------------------------------------------------------------------------------------------- */
int vptr_tattn_fwd(const float* q, const float* k, const float* v, float* o, int Nb, int Tq, int Tk, int HW, int C, int nh,
                   int causal, float dropout_p, const uint64_t* seed_dev, uint32_t site, int p16, vptr_stream_t stream);
int vptr_tattn_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq, float* dk, float* dv,
                   int Nb, int Tq, int Tk, int HW, int C, int nh, int causal, float dropout_p, const uint64_t* seed_dev,
                   uint32_t site, float dq_scale /* as in vptr_winattn_bwd */, int p16, vptr_stream_t stream);

/* Temporal-spatial window attention (TemporalSpatialLocalMultiheadAttention, VidHRFormer_modules.py:219-284 with the
 * permutes of :444-484 folded into index arithmetic): q [(n,tq,h,w), C] (pre-scaled), k, v [(n,tk,h,w), C]; for every
 * ws x ws window and head the Tq*ws*ws queries attend to the Tk*ws*ws memory tokens of the same window.
 * H, W multiples of ws (PadBlock padding is applied by the caller with vptr_window_copy). */
int vptr_tsattn_fwd(const float* q, const float* k, const float* v, float* o, int Nb, int Tq, int Tk, int H, int W, int ws, int C,
                    int nh, float dropout_p, const uint64_t* seed_dev, uint32_t site, int p16, vptr_stream_t stream);
int vptr_tsattn_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq, float* dk, float* dv, int Nb,
                    int Tq, int Tk, int H, int W, int ws, int C, int nh, float dropout_p, const uint64_t* seed_dev,
                    uint32_t site, int p16, vptr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Conv-FFN pieces (MlpDWBN, VidHRFormer_modules.py:424-442) on channel-last [rows = frames*HW, F].
 * Normalisation statistics:
 *   group_rows == rows : BatchNorm2d batch statistics per channel  -> mean[F], var[F] (biased)
 *   group_rows == HW   : LayerNorm((F,H,W)) statistics per frame   -> mean[frames], var[frames]
 * ---------------------------------------------------------------------------------------------- */
/* rstd (optional, may be NULL) = 1/sqrt(var + eps), written by the same launch */
int vptr_colstats(const float* x, float* mean, float* var, float* rstd, float eps,
                  float* scratch /* >= 2*F*ceil(rows/256) floats */, int rows, int F, vptr_stream_t stream);
/* the same plus BatchNorm2d's train-mode bookkeeping in the same launch (torch/nn/modules/batchnorm.py as used by
 * VidHRFormer_modules.py:397-419): running_mean/var <- (1 - momentum) * running + momentum * (mean | var * rows / (rows - 1)),
 * num_batches_tracked += 1; each of the three may be NULL */
int vptr_colstats_running(const float* x, float* mean, float* var, float* rstd, float eps, float* scratch, int rows, int F,
                          float* running_mean, float* running_var, float momentum, long long* num_batches_tracked,
                          vptr_stream_t stream);
int vptr_groupstats(const float* x, float* mean, float* var, float* rstd, float eps, int groups, int group_elems,
                    vptr_stream_t stream);
/* y = rowscale[(row/rs_div)%rs_mod] * dropout(act( (x - mean)*rstd * w + b )) + residual
 * per_col != 0: stats indexed by column (BN), affine [F];
 * per_col == 0: stats indexed by row / HW (LN over (F,H,W)), affine given channel-last as [HW, F].
 * rowscale (DropPath, VidHRFormer_modules.py:563-575) and residual may be null. */
/* raw_stats != NULL (per_col == 0 only): [frames][2] per-frame sum / sum of squares of x as accumulated by its producer
 * (vptr_gemm_desc::frame_stats, vptr_dwconv3x3_fwd); the kernel derives mean / rstd from them (eps) and WRITES mean[], rstd[]
 * for the backward pass -- otherwise mean[], rstd[] are inputs. */
int vptr_norm_act_fwd(const float* x, float* mean, float* rstd, const float* w, const float* b, float* y,
                      int rows, int F, int HW, int per_col, int act, float dropout_p, const uint64_t* seed_dev,
                      uint32_t site, const float* rowscale, int rs_div, int rs_mod, const float* residual,
                      int p16 /* != 0: y is written in the P16 plane format */, const float* raw_stats, float eps,
                      vptr_stream_t stream);
/* backward: dx written; dw/db ACCUMULATED (same layout as w/b).  scratch (initialised inside): per_col: >= 2*F floats;
 * per-frame: >= 2*frames*(1 + 4*ceil(HW*F/1024)) floats, frames = rows/HW (frame sums + per-wave partials).
 * const_stats != 0: mean/rstd are constants (BatchNorm in eval mode) -> no statistics terms in dx. */
int vptr_norm_act_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* w, const float* b,
                      float* dx, float* dw, float* db, float* scratch, int rows, int F, int HW, int per_col, int act,
                      int const_stats, float dropout_p, const uint64_t* seed_dev, uint32_t site, const float* rowscale,
                      int rs_div, int rs_mod, int p16 /* != 0: dx is written in the P16 plane format */, vptr_stream_t stream);
/* The LayerNorm((F,H,W)) mode (per_col = 0) with the affine gradients DEFERRED like vptr_layernorm_bwd_deferred: dx written, the
 * frame chunks' sums to partials[vptr_norm_act_bwd_partials(rows, F, HW, 0)][2][HW * F] (rows of dw / db), added into their
 * destinations by the backward pass's vptr_partial_reduce launch.  Without atomics the frames are cut into more chunks. */
int vptr_norm_act_bwd_deferred(const float* dy, const float* x, const float* mean, const float* rstd, const float* w, const float* b,
                               float* dx, float* scratch, int rows, int F, int HW, int act, int const_stats, float dropout_p,
                               const uint64_t* seed_dev, uint32_t site, const float* rowscale, int rs_div, int rs_mod, int p16,
                               float* partials, vptr_stream_t stream);
/* ABI 10.  The LayerNorm((F,H,W)) mode of vptr_norm_act_bwd_deferred in ONE pass over dy and x (cooperative: the workgroups of a chunk of 10
 * frames exchange the frames' two sums through sync_ws and wait for each other; 260 MB per [10240 x 2112] call instead of 432, one launch
 * instead of three).  vptr_norm_act_bwd_coop_partials = rows of the partial-sum buffer [chunks][2][HW * F] it writes (0: take the two-phase
 * call -- VPTR_NORM_COOP=0, deterministic mode, fewer than 16 frames, a chunk too large to be co-resident).  sync_ws: [(rows / HW) + 1][32]
 * floats, 128-byte aligned, ZERO on entry, not restored (last line: != 0 if a bounded wait ever ran out). */
int vptr_norm_act_bwd_coop_partials(int rows, int F, int HW);
int vptr_norm_act_bwd_coop(const float* dy, const float* x, const float* mean, const float* rstd, const float* w, const float* b, float* dx,
                           float* sync_ws, int rows, int F, int HW, int act, float dropout_p, const uint64_t* seed_dev, uint32_t site,
                           const float* rowscale, int rs_div, int rs_mod, int p16, float* partials, vptr_stream_t stream);
int vptr_norm_act_bwd_partials(int rows, int F, int HW, int per_col);   /* a plain number (0: use vptr_norm_act_bwd) */
/* depthwise 3x3, pad 1 (VidHRFormer_modules.py:404-409,433); w given tap-major [9, F]. */
/* frame_stats (may be NULL): [frames][2] zeroed buffer that receives each frame's sum / sum of squares of y (see
 * vptr_norm_act_fwd raw_stats); needs W even and (W/2)*(F/4) % 64 == 0 */
int vptr_dwconv3x3_fwd(const float* x, const float* w9, const float* b, float* y, int frames, int H, int W, int F,
                       float* frame_stats, vptr_stream_t stream);
int vptr_dwconv3x3_bwd(const float* dy, const float* x, const float* w9, float* dx, float* dw9, float* db, int frames,
                       int H, int W, int F, vptr_stream_t stream);
/* ABI 9.  The first normalisation of the conv-FFN folded into the depthwise kernel's load path (VidHRFormer_modules.py:430-434: fc1 ->
 * norm1 = LayerNorm((F,H,W)) -> act1 -> dw3x3): x is the RAW output of fc1, raw_stats its per-frame sum / sum of squares as left by the GEMM
 * epilogue (vptr_gemm_desc::frame_stats), aff_w / aff_b the channel-last [H*W, F] affine of the normalisation.  y = dw3x3(act(norm(x))) (+ its
 * own frame_stats, as vptr_dwconv3x3_fwd); mean_out / rstd_out [frames] are written for the normalisation's backward pass
 * (vptr_norm_act_bwd*); a_half (may be NULL) receives act(norm(x)) as fp16 [frames*H*W, F] -- the only thing the backward pass needs of the
 * activated tensor is the x operand of the depthwise weight gradient, vptr_dwconv3x3_bwd_xh.  Needs W even, W / 2 dividing 16 and
 * (W/2)*(F/4) % 64 == 0; no fallback. */
int vptr_dwconv3x3_norm_fwd(const float* x, const float* raw_stats, const float* aff_w, const float* aff_b, float eps, int act,
                            const float* w9, const float* b, float* y, void* a_half, float* mean_out, float* rstd_out,
                            int frames, int H, int W, int F, float* frame_stats, vptr_stream_t stream);
/* vptr_dwconv3x3_bwd with the forward input given as the fp16 copy vptr_dwconv3x3_norm_fwd wrote (W even) */
int vptr_dwconv3x3_bwd_xh(const float* dy, const void* x_half, const float* w9, float* dx, float* dw9, float* db, int frames,
                          int H, int W, int F, vptr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise / layout helpers
 * ---------------------------------------------------------------------------------------------- */
/* [B, C, HW] (NCHW) <-> [B, HW, C] (tokens); relu != 0 applies ReLU to the output
 * (VidHRFormer.py:43,51 permutes + relu_). */
int vptr_nchw_to_tokens(const float* src, float* dst, int B, int C, int HW, vptr_stream_t stream);
int vptr_tokens_to_nchw(const float* src, float* dst, int B, int C, int HW, int relu, vptr_stream_t stream);
/* backward of tokens_to_nchw(relu): dtok = nchw_to_tokens(dout * (out > 0)) */
int vptr_nchw_to_tokens_masked(const float* dout, const float* out, float* dtok, int B, int C, int HW, vptr_stream_t stream);
/* backward of the GEMM epilogue: dx[m,n] = dy[m,n] * dropmask(site) * rowscale[(m / rs_div) % rs_mod] * act'(h[m,n]) * alpha
 * (act 1: exact GELU on the saved pre-activation h; act 2: ReLU mask h > 0; act 0: h unused). rowscale may be null. */
int vptr_act_bwd(const float* dy, const float* h, float* dx, int rows, int C, int act, float alpha, const float* rowscale,
                 int rs_div, int rs_mod, float dropout_p, const uint64_t* seed_dev, uint32_t site,
                 int out_p16 /* != 0: dx is written in the P16 plane format (C % 16 == 0): it only feeds GEMMs */, vptr_stream_t stream);
/* y = x * mask(site)/keep -- standalone dropout (fwd and bwd are the same call) */
int vptr_dropout(const float* x, float* y, int64_t n, float dropout_p, const uint64_t* seed_dev, uint32_t site,
                 vptr_stream_t stream);
/* dx[m,n] = dy[m,n] * rowscale[(m / div) % mod] */
int vptr_rowscale(const float* dy, const float* rowscale, float* dx, int rows, int C, int div, int mod, vptr_stream_t stream);

/* Token grids [frames, H, W, C]: dst[f, hd, wd, :] = src[f, hd - off_h, wd - off_w, :] inside the source grid, 0 outside.
 * off > 0: centre padding of PadBlock.pad_if_needed (VidHRFormer_modules.py:546-557) before the window partition when
 * H or W is not a multiple of the window; off < 0: the crop of depad_if_needed (:559-569).  C % 4 == 0. */
int vptr_window_copy(const float* src, float* dst, int frames, int Hs, int Ws, int Hd, int Wd, int off_h, int off_w, int C,
                     vptr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Direct convolutions of the auto-encoder ends (ResNetAutoEncoder.py:26-29 and :89-96).
 * ---------------------------------------------------------------------------------------------- */
/* first layer: ReflectionPad2d(3) + Conv7x7(Cimg -> Cout) + folded BN + ReLU; x NCHW [B,Cimg,H,W] -> y NHWC [B,H,W,Cout];
 * w is the PyTorch weight [Cout, Cimg, 7, 7].  scale == NULL: the raw convolution output (train-mode BatchNorm follows
 * as its own statistics + normalise passes, stage-1 training train_AutoEncoder.py:44-86). */
int vptr_conv7_in_fwd(const float* x, const float* w, const float* scale, const float* shift, float* y, int B, int Cimg,
                      int H, int W, int Cout, vptr_stream_t stream);
/* the same with the result as bf16 hi / lo planes [(B*H*W + 1)][2][64] (vptr_split_planes format; the caller keeps the last row
 * zero): the A operand of the strided plane-operand convolution that follows in the frozen encoder.  Cimg == 1, Cout == 64. */
int vptr_conv7_in_fwd_planes(const float* x, const float* w, const float* scale, const float* shift, void* planes, int B, int Cimg,
                             int H, int W, int Cout, vptr_stream_t stream);
/* last layer: ReflectionPad2d(3) + Conv7x7(Cin -> Cimg) + bias + Tanh(1)/Sigmoid(2); x NHWC -> y NCHW; w [Cimg,Cin,7,7] */
int vptr_conv7_out_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W, int Cimg,
                       int out_act, vptr_stream_t stream);
/* backward of the last layer w.r.t. its input: dy, y NCHW [B,Cimg,H,W] -> dx NHWC [B,H,W,Cin] */
int vptr_conv7_out_bwd_data(const float* dy, const float* y, const float* w, float* dx, int B, int Cin, int H, int W,
                            int Cimg, int out_act, vptr_stream_t stream);
/* backward of the last layer w.r.t. weight/bias: dw [Cimg,Cin,7,7], db [Cimg] ACCUMULATED */
int vptr_conv7_out_bwd_weight(const float* dy, const float* y, const float* x, float* dw, float* db, int B, int Cin, int H,
                              int W, int Cimg, int out_act, vptr_stream_t stream);
/* The same with a caller-owned scratch buffer of vptr_conv7_out_bwd_weight_workspace(B, Cimg) floats (16-byte aligned): a second
 * kernel (lane = input channel with its 49 taps in registers, the gradient window as broadcast LDS reads) leaves one partial
 * per workgroup there and a reduction adds them in a fixed order -- no atomics on dw.  Without (enough) workspace: the call above. */
int vptr_conv7_out_bwd_weight_ws(const float* dy, const float* y, const float* x, float* dw, float* db, int B, int Cin, int H, int W,
                                 int Cimg, int out_act, float* workspace, int workspace_floats, vptr_stream_t stream);
int vptr_conv7_out_bwd_weight_workspace(int B, int Cimg);   /* floats; a plain number, not an error code */
/* dx = dy * (y > 0) * scale[c]   (ReLU + folded-BN backward on channel-last [rows, C]) */
int vptr_bnrelu_bwd(const float* dy, const float* y, const float* scale, float* dx, int64_t rows, int C, vptr_stream_t stream);
/* eval-mode BatchNorm2d affine gradients behind the ReLU, from the layer output y = relu(w*xhat + b) (ACCUMULATED):
 * db[c] += sum_{y>0} dy;  dw[c] += sum_{y>0} dy * (y - b[c]) / w[c]   (ResNetAutoEncoder.py:79-80 in stage 2) */
int vptr_bnrelu_bwd_params(const float* dy, const float* y, const float* w, const float* b, float* dw, float* db, int64_t rows,
                           int C, vptr_stream_t stream);
/* vptr_bnrelu_bwd and vptr_bnrelu_bwd_params in one pass over (dy, y): dx written, dw / db ACCUMULATED (C % 4 == 0).  The
 * decoder backward of the stage-2 / stage-3 steps (every up-sampling layer's BatchNorm is trainable there although never stepped). */
int vptr_bnrelu_bwd_fused(const float* dy, const float* y, const float* scale, const float* w, const float* b, float* dx,
                          float* dw, float* db, int64_t rows, int C, vptr_stream_t stream);
/* im2col on NHWC: out[(b,oy,ox)][(ky,kx,c)] = x[b, oy*stride-pad+ky, ox*stride-pad+kx, c] (pad_mode: VPTR_PAD_ZERO or
 * VPTR_PAD_REFLECT for out-of-range taps); the patch matrix is the k-strided operand of the convolution weight-gradient
 * GEMMs (autograd of ResNetAutoEncoder.py:33-48,74-88,138,151 and of the PatchGAN convs VPTR_modules.py:70-91). */
int vptr_im2col_nhwc(const float* x, float* out, int B, int IH, int IW, int C, int OH, int OW, int KH, int KW, int stride,
                     int pad, int pad_mode, vptr_stream_t stream);
/* the same matrix as a P16 tensor (C % 16 == 0): token-major B operand of vptr_gemm_grouped (VPTR_B_P16T) */
int vptr_im2col_nhwc_p16(const float* x, float* out_p16, int B, int IH, int IW, int C, int OH, int OW, int KH, int KW,
                         int stride, int pad, int pad_mode, vptr_stream_t stream);
/* adjoint of reflection padding: dxpad [B, H+2p, W+2p, C] (gradient w.r.t. the padded image, e.g. from the gather-form
 * transposed convolution with pad 0) -> dx [B, H, W, C], each border contribution folded back onto the pixel it mirrors
 * (nn.ReflectionPad2d of ResnetBlock, ResNetAutoEncoder.py:127-151). */
int vptr_reflect_fold(const float* dxpad, float* dx, int B, int H, int W, int C, int pad, vptr_stream_t stream);
/* weight gradient of the first layer: dw[Cout=64, Cimg, 7, 7] += sum_pix dy[pix, co] * xpad[pix + tap, ci];
 * dy NHWC [B,H,W,64] (gradient of the RAW convolution output), x NCHW [B,Cimg,H,W] (ResNetAutoEncoder.py:26-27 autograd) */
int vptr_conv7_in_bwd_weight(const float* dy, const float* x, float* dw, int B, int Cimg, int H, int W, int Cout,
                             vptr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Losses of the train steps (model/criterion.py) and the stochastic-depth vectors as plain kernel launches: a whole-step
 * hipGraph must not hold memset nodes (ATen's multi-block reductions issue cudaMemsetAsync).
 * ---------------------------------------------------------------------------------------------- */
/* MSELoss + GDL(alpha = 1), no temporal weights (criterion.py:105-132, 134-204; call sites train_NAR.py:87-88,
 * train_FAR.py:37-38, train_AutoEncoder.py:37-38).  pred, gt: [planes, H, W] (planes = N*T*Cimg).
 * scratch: planes * ceil(H / 16) * 3 floats.  mse_out / gdl_out: one float each.  Deterministic (fixed-order sums). */
int vptr_mse_gdl_fwd(const float* pred, const float* gt, float* scratch, float* mse_out, float* gdl_out, int planes, int H,
                     int W, vptr_stream_t stream);
/* dpred = g_mse[0] * d mse / d pred + g_gdl[0] * d gdl / d pred; g_mse / g_gdl: DEVICE scalars (null = 0). */
int vptr_mse_gdl_bwd(const float* pred, const float* gt, const float* g_mse, const float* g_gdl, float* dpred, int planes,
                     int H, int W, vptr_stream_t stream);
/* BiPatchNCE (criterion.py:206-259) INCLUDING the F.normalize(p = 2, dim = channel) in front of it (train_NAR.py:81-84):
 * g = NCE_projector(gt features), p = NCE_projector(predicted features), token-major [frames * L, C], L = h*w patches
 * per frame.  scratch (kept for the backward): frames * L * (4 + L) + frames floats
 * (inverse norms [2 R] | scores [frames][L][L] | row / column log-sum-exp [R] each | per-frame partials), R = frames * L. */
int vptr_nce_fwd(const float* g, const float* p, float* scratch, float* loss_out, int frames, int L, int C,
                 float temperature, vptr_stream_t stream);
/* dg, dp [frames * L, C] = gout[0] * d loss / d (g, p) (gout: DEVICE scalar, null = 1); C % 4 == 0, C <= 640. */
int vptr_nce_bwd(const float* g, const float* p, const float* scratch, const float* gout, float* dg, float* dp, int frames,
                 int L, int C, float temperature, vptr_stream_t stream);
/* DropPath scale vectors (VidHRFormer_modules.py:563-575): out[r][k] = floor(keep[r] + U(seed, site0 + r, k)) / keep[r] for
 * nreq requests of up to maxcount indices each, from the counter-based hash of the step's dropout seed. */
int vptr_droppath_scales(const float* keep, float* out, int nreq, int maxcount, const uint64_t* seed_dev, uint32_t site0,
                         vptr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer: global-norm clip + AdamW on flat fp32 buffers (train_NAR.py:85-86,205).
 * ---------------------------------------------------------------------------------------------- */
/* sumsq_dev[0] += sum(g^2) */
int vptr_sumsq(const float* g, int64_t n, float* sumsq_dev, vptr_stream_t stream);
/* sumsq_dev[0] = sum(g^2) in a fixed order: nws workgroups leave partials in ws[nws] (caller-owned), one workgroup adds them by index */
int vptr_sumsq_ws(const float* g, int64_t n, float* sumsq_dev, float* ws, int nws, vptr_stream_t stream);
/* p,m,v updated in place. clip coefficient = min(1, max_norm / (sqrt(sumsq_dev[0]) + 1e-6)) if sumsq_dev != null.
 * step_dev: DEVICE pointer to the (float) step count already incremented for this step. grad_scale multiplies g first. */
int vptr_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
               float weight_decay, const float* step_dev, const float* sumsq_dev, float max_norm, float grad_scale,
               vptr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VPTR_HIP_H */
