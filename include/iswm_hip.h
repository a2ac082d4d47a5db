/*
 * iswm_hip.h -- C ABI of libiswm_hip.so, the MI355X (gfx950) kernels behind the
 * DeepLabV3+ training hot path of Alanlee0323/ISWM.
 *
 * The reference has no FFI layer of its own: its hot path is a chain of stock
 * torch.nn modules.  Each entry point below replaces the ATen op
 * that one reference call site runs; the call site is cited next to it (paths
 * relative to the reference checkout).  The host side (iswm_amd/, Python) binds
 * these with ctypes -- plain pointers and sizes, no torch types.
 *
 * Conventions
 *   - all tensors are device pointers to fp32 unless stated otherwise;
 *   - activations are NHWC ("channels last"): element (n,h,w,c) of a tensor with
 *     pixel pitch ld lives at ((n*H+h)*W+w)*ld + c.  ld >= C lets an op read or
 *     write a channel slice of a wider buffer;
 *   - conv weights are OHWI: [Cout][KH][KW][Cin] (a torch [Cout,Cin,KH,KW]
 *     parameter in channels_last memory format has exactly this layout);
 *   - every function enqueues on `stream` and returns immediately; it never
 *     synchronises, allocates or frees (safe under hipGraph capture);
 *   - return 0 on success, non-zero on a rejected argument or a launch error;
 *     iswm_last_error() then holds a thread-local message.
 */
#ifndef ISWM_HIP_H
#define ISWM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* iswm_stream_t; /* hipStream_t */

const char* iswm_last_error(void);
int iswm_version(void);

/* ---- convolution (implicit GEMM on v_mfma_f32_32x32x2_f32) ------------------
 * nn.Conv2d call sites: network/backbone/resnet.py:27-35,144,184-187;
 * network/_deeplab.py:37,44-51,124,134,149,162.  groups == 1, square dilation.
 * Requirements: Cin % 4 == 0, Cout % 4 == 0, ldx % 4 == 0, ldy % 4 == 0,
 * 16-byte aligned pointers (the host pads the 3-channel stem input to 4 and the
 * num_classes-wide classifier to a multiple of 4). */
typedef struct {
    int N, H, W, Cin;   /* input  [N,H,W,Cin]  */
    int Ho, Wo, Cout;   /* output [N,Ho,Wo,Cout] */
    int KH, KW, stride, pad, dil;
    int ldx;            /* pixel pitch of x  (floats) */
    int ldy;            /* pixel pitch of y  (floats) */
} iswm_conv_desc;

/* Arithmetic of the convolution kernels: 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32);
 * 1 = "bf16x6": each fp32 operand is split exactly into three bf16 pieces and the product is formed
 * from six bf16 MFMAs with fp32 accumulation (error ~1e-7 relative, i.e. fp32 level, at up to 2.7x the
 * fp32 MFMA rate).  Default 1; the environment variable ISWM_CONV_MATH=f32 selects 0 at load time.
 * Mode 2 ("bf16", ISWM_CONV_MATH=bf16) is MIXED PRECISION, not fp32-grade: operands rounded to nearest bf16, ONE
 * bf16 MFMA per product, fp32 accumulation and fp32 tensors -- through the packed forward / data-gradient entry
 * points and the weight gradient; the plain entry points stay on the fp32 MFMA kernels in this mode.
 * Geometries the bf16x6 kernels do not cover (gathered channel count not a multiple of 32: the stem, the
 * 304-channel decoder input) always run on the fp32 MFMA kernels. */
int iswm_set_conv_math(int mode);
int iswm_get_conv_math(void);
/* name of the device kernel a call with this geometry launches (kind 0 fwd, 1 dgrad, 2 wgrad, 3 fwd_packed,
 * 4 dgrad_packed, 5 fwd_pl2, 6 dgrad_pl2) --
 * lets a profiler label its timings with the symbol rocprofv3 reports */
int iswm_conv2d_kernel_name(const iswm_conv_desc* d, int kind, char* buf, int buflen);
/* M tiling the forward kernel will use for this geometry: rows per tile (128 or 64) and number of
 * tiles == rows of the BN partials */
int iswm_conv2d_stat_tile_rows(const iswm_conv_desc* d);
int iswm_conv2d_stat_tiles(const iswm_conv_desc* d);
/* y = conv(x, w) (+ bias).  If stat_partials != NULL it receives per-M-tile
 * per-channel statistics [2][tiles][Cout] = {S_t, M2_t} (see iswm_colstat) for the training-mode
 * BatchNorm that follows every conv (network/backbone/resnet.py:89-93). */
int iswm_conv2d_fwd(const iswm_conv_desc* d, const float* x, const float* w, const float* bias,
                    float* y, float* stat_partials, iswm_stream_t stream);
/* dx (=|+=) conv_transpose(dy, w): autograd backward of the conv wrt its input;
 * accumulate != 0 adds into dx (branches that share an input, residual joins) */
int iswm_conv2d_dgrad(const iswm_conv_desc* d, const float* dy, const float* w, float* dx,
                      int accumulate, iswm_stream_t stream);
/* bf16x6 data gradient: needs the weights transposed to [Cin][KH][KW][Cout] (K axis contiguous for the
 * matrix cores).  iswm_conv2d_dgrad_wants_wt tells the host when to use this pair instead of
 * iswm_conv2d_dgrad (current conv math is bf16x6 and Cout % 32 == 0). */
int iswm_transpose_weights(const iswm_conv_desc* d, const float* w, float* wt, iswm_stream_t stream);
int iswm_conv2d_dgrad_wants_wt(const iswm_conv_desc* d);
int iswm_conv2d_dgrad_wt(const iswm_conv_desc* d, const float* dy, const float* wt, float* dx, int accumulate,
                         iswm_stream_t stream);
/* bf16x6 with pre-split weights: iswm_conv2d_pack_weights splits a weight tensor into its three bf16 planes and
 * stores them in MFMA fragment order (kind 0: for the forward conv, 1: for the data gradient, which also transposes);
 * the *_packed kernels then load the weight operand straight into registers.  iswm_conv2d_packed_weight_bytes
 * returns the buffer size, or 0 when the packed path does not apply (conv math f32, or the gathered channel
 * count -- Cin forward, Cout data gradient -- is not a multiple of 32). */
size_t iswm_conv2d_packed_weight_bytes(const iswm_conv_desc* d, int kind);
int iswm_conv2d_pack_weights(const iswm_conv_desc* d, int kind, const float* w, void* packed, iswm_stream_t stream);
/* Batched form: all conv weights of a model in ONE launch (a step otherwise issues two tiny pack kernels per conv).
 * jobs_dev is a DEVICE array sorted by first_block; job i owns workgroups [first_block, first_block +
 * iswm_pack_job_blocks(...)); total_blocks is the sum.  iswm_packed_weight_bytes is the buffer size for
 * (Cout, taps, Cin, kind), 0 if the gathered channel count (Cin forward, Cout data gradient) is not 32-aligned.
 * kind 2 / 3: forward / data-gradient weights in the fragment order of the planes kernels (iswm_conv2d_fwd_pl2 /
 * iswm_conv2d_dgrad_pl2; gathered channel count a multiple of 64). */
typedef struct iswm_pack_job {
    const float* w;         /* [Cout][taps][Cin] (OHWI) */
    void* packed;
    int Cout, taps, Cin, kind;
    int first_block, reserved;
} iswm_pack_job;
size_t iswm_packed_weight_bytes(int Cout, int taps, int Cin, int kind);
int iswm_pack_job_blocks(int Cout, int taps, int Cin, int kind);
int iswm_pack_weights_batch(const iswm_pack_job* jobs_dev, int njobs, int total_blocks, iswm_stream_t stream);
/* layout of the BN partials iswm_conv2d_fwd_packed writes: tiles x Cout floats per plane (sum, centred M2).
 * *tile_rows > 0: every tile holds that many rows (the last one the remainder); *tile_rows == 0: the tiles are
 * image patches of varying size and their row counts follow the planes as floats (partials + 2*tiles*Cout), so
 * the buffer is 2*tiles*Cout + tiles floats -- iswm_bn_finalize takes tile_rows = 0 for that layout. */
int iswm_conv2d_fwd_packed_stat_layout(const iswm_conv_desc* d, int* tiles, int* tile_rows);
int iswm_conv2d_fwd_packed(const iswm_conv_desc* d, const float* x, const void* wpk, const float* bias,
                           float* y, float* stat_partials, iswm_stream_t stream);
int iswm_conv2d_dgrad_packed(const iswm_conv_desc* d, const float* dy, const void* wpk, float* dx, int accumulate,
                             iswm_stream_t stream);
/* ---- "planes": activations stored PRE-SPLIT for the bf16x6 kernels.  A planes tensor is three bf16 tensors
 * [plane][pixel][ld] (hi, mid, lo -- the exact truncation split, hi + mid + lo == the fp32 value), plane_stride
 * bf16 elements apart (one plane under conv math "bf16": the value rounded to nearest).  The producer of an
 * activation writes the split once; the convolution stages its activation operand HBM -> LDS by LDS-DMA with no
 * split arithmetic in the loop.  In the *_planes entry points the pitch of the planes operand (d->ldx forward,
 * d->ldy data gradient) counts bf16 elements and must be a multiple of 8; same call sites as iswm_conv2d_fwd. */
int iswm_split_planes(const float* x, int64_t M, int C, int ldx, void* planes, int ldp, int64_t plane_stride,
                      iswm_stream_t stream);
int iswm_join_planes(const void* planes, int ldp, int64_t plane_stride, int64_t M, int C, float* x, int ldx,
                     iswm_stream_t stream);
/* Planes-aware forms of the memory-bound passes that produce or consume convolution operands.  A tensor argument typed
 * `void*` with a `*_ps` companion is an fp32 tensor when *_ps == 0 (pitch in floats), three bf16 planes when *_ps > 0
 * (pitch and plane stride in bf16 elements) and one rounded bf16 plane when *_ps == -1.  The fp32-only entry points
 * further down are these with every *_ps == 0. */
int iswm_bn_apply_pl(const float* y, int64_t M, int C, int ldy, const float* scale, const float* shift, const float* mean,
                     const void* residual, int ldr, int64_t res_ps, int relu, void* out, int ldo, int64_t out_ps,
                     iswm_stream_t stream);
int iswm_bn_backward_pl(const float* dout, int ldd, const void* out, int ldo, int64_t out_ps, const float* y, int ldy,
                        int64_t M, int C, const float* mean, const float* invstd, const float* gamma,
                        const float* mask_scale, const float* mask_shift, int relu, int training, float* dgamma,
                        float* dbeta, void* dy, int lddy, int64_t dy_ps, float* dres, int lddres, void* workspace,
                        size_t workspace_bytes, iswm_stream_t stream);
int iswm_maxpool3x3s2_fwd_pl(const float* x, int N, int H, int W, int C, void* y, int64_t y_ps, uint8_t* idx, int Ho,
                             int Wo, iswm_stream_t stream);
int iswm_gap_fwd_pl(const void* x, int64_t x_ps, int N, int HW, int C, int ldx, float* y, iswm_stream_t stream);
int iswm_bcast_fwd_pl(const float* v, int N, int HW, int C, void* y, int ldy, int64_t y_ps, iswm_stream_t stream);
int iswm_bilinear_fwd_pl(const float* x, int N, int Hi, int Wi, int C, int ldx, void* y, int64_t y_ps, int Ho, int Wo,
                         int ldy, iswm_stream_t stream);
/* diagnostics: per-stage shader-clock stamps of workgroup 0 of the planes conv kernels into a device buffer of >= 512 uint64
 * (NULL = off, the default; tools/pl2_timeline.py) */
int iswm_set_debug_buffer(void* buf);
/* second-generation planes kernels: (16*rbw) x 128 tiles, weights packed for the 16x16x32 MFMA (own packing) */
size_t iswm_conv2d_pl2_weight_bytes(const iswm_conv_desc* d, int kind);
int iswm_conv2d_pl2_pack_weights(const iswm_conv_desc* d, int kind, const float* w, void* packed, iswm_stream_t stream);
int iswm_conv2d_pl2_tile_rows(const iswm_conv_desc* d, int kind);
int iswm_conv2d_fwd_pl2(const iswm_conv_desc* d, const void* xp, int64_t plane_stride, const void* wpk,
                        const float* bias, float* y, float* stat_partials, iswm_stream_t stream);
int iswm_conv2d_dgrad_pl2(const iswm_conv_desc* d, const void* dyp, int64_t plane_stride, const void* wpk,
                          float* dx, int accumulate, iswm_stream_t stream);
/* The same data gradient, also emitting the first pass of the BatchNorm backward that CONSUMES dx (the BatchNorm of the stage
 * whose activation the conv read -- reference resnet.py:103-118: bn1 / bn2 of a Bottleneck): per tile row t and input channel c
 *   partials[0][t][c] = sum dz,  partials[1][t][c] = sum dz * xhat,   dz = dx * [ReLU pattern],  xhat = (y - mean) * invstd
 * over the finished dx (after accumulation).  y: that stage's raw conv output [N*H*W][ldy]; relu 0 = no activation, 2 = pattern
 * recomputed as (y - mean) * mask_scale + mask_shift > 0, exactly as iswm_bn_backward does; 3 = the producer is a RESIDUAL
 * stage (bn3 of a Bottleneck, resnet.py:110-118): pattern = hi plane of its saved output planes (mask_hi, pitch ld_mask bf16
 * elements) > 0, and dx is stored MASKED -- the one tensor is then both dout of that stage's BatchNorm backward (call it with
 * relu = 0) and the gradient of its identity branch: no reduction pass, no separate dres.  partials: 2 * tiles * Cin doubles,
 * tiles = iswm_conv2d_dgrad_pl2_stat_tiles(d).  iswm_bn_backward_stats_pl then runs only the finalize and apply passes
 * (8 bytes per element of HBM traffic less than iswm_bn_backward_pl). */
int iswm_conv2d_dgrad_pl2_stat_tiles(const iswm_conv_desc* d);
int iswm_conv2d_dgrad_pl2_bn(const iswm_conv_desc* d, const void* dyp, int64_t plane_stride, const void* wpk, float* dx,
                             int accumulate, const float* y, int ldy, const float* mean, const float* invstd,
                             const float* mask_scale, const float* mask_shift, int relu, const void* mask_hi,
                             int ld_mask, double* partials, int tiles, iswm_stream_t stream);
int iswm_bn_backward_stats_pl(const float* dout, int ldd, const void* out, int ldo, int64_t out_ps, const float* y, int ldy,
                              int64_t M, int C, const float* mean, const float* invstd, const float* gamma,
                              const float* mask_scale, const float* mask_shift, int relu, int training, float* dgamma,
                              float* dbeta, void* dy, int lddy, int64_t dy_ps, float* dres, int lddres,
                              const double* partials, int tiles, void* workspace, size_t workspace_bytes,
                              iswm_stream_t stream);
/* ---- ASPP as one launch (conv_mfma_pl2t.hip): the parallel branches of network/_deeplab.py:143-172 -- ASPP.convs[0] (1x1),
 * convs[1..3] (ASPPConv, 3x3 at the three atrous rates, :121-128) -- over ONE tile table.  Rows (pixels) are sorted by their set
 * of in-bounds filter taps so that a tile runs exactly the taps that reach it; the tiles of all branches are dealt heaviest
 * first to a persistent grid; the data gradient of the four branches is a single GEMM over their 28 taps (dx written once).
 * The descriptor gives N, H, W (= Ho, Wo), Cin, Cout (PER BRANCH), stride 1, ldx, ldy; ksize[b] / dil[b] describe branch b
 * (odd square filter, pad = dil * (k - 1) / 2).  The PLAN (tap table, row order, tile table) is built on the host once per
 * geometry -- iswm_aspp_plan_bytes / iswm_aspp_plan -- and copied to the device by the caller; kind 0 forward, 1 data
 * gradient.  Pointer arrays (wpk, y, stats, dw) are host arrays of nbranch device pointers.
 *   iswm_aspp_fwd : y[b] = conv(x planes, w_b) into fp32 [N*H*W][ldy] + optional BatchNorm partials stats[b] = [2][T][Cout],
 *                   T = ceil(N*H*W / 144) tile rows of 144 (feed iswm_bn_finalize with tile_rows = 144);
 *                   wpk[b] = iswm_conv2d_pl2_pack_weights(kind 0) of branch b.
 *   iswm_aspp_bwd : dx (=|+=) sum_b conv^T(dy_b, w_b); dyp = planes of the concatenated gradient [N*H*W][ld_dy], branch b at
 *                   channels [b*Cout, (b+1)*Cout); wpk[b] = iswm_conv2d_pl2_pack_weights(kind 1) of branch b.  With xp / dw /
 *                   workspace it also runs the four weight gradients (iswm_conv2d_wgrad_planes per branch; workspace >= the
 *                   largest iswm_conv2d_wgrad_planes_workspace of the branches). */
size_t iswm_aspp_plan_bytes(const iswm_conv_desc* d, int nbranch, const int* ksize, const int* dil, int kind);
int iswm_aspp_plan(const iswm_conv_desc* d, int nbranch, const int* ksize, const int* dil, int kind, void* host_plan,
                   int grid_hint);
int iswm_aspp_fwd(const iswm_conv_desc* d, int nbranch, const int* ksize, const int* dil, const void* plan_dev, const void* xp,
                  int64_t x_ps, const void* const* wpk, float* const* y, float* const* stats, iswm_stream_t stream);
int iswm_aspp_bwd(const iswm_conv_desc* d, int nbranch, const int* ksize, const int* dil, const void* plan_dev, const void* dyp,
                  int64_t dy_ps, int ld_dy, const void* const* wpk, float* dx, int accumulate, const void* xp, int64_t x_ps,
                  float* const* dw, float* workspace, size_t workspace_bytes, iswm_stream_t stream);
/* weight gradient with BOTH operands pre-split (x: planes of the conv input, pitch d->ldx; dy: planes of the gradient of
 * the conv output, pitch d->ldy; pitches and plane strides in bf16 elements).  Needs Cin % 8 == 0 and Cout % 8 == 0
 * (iswm_conv2d_wgrad_planes_ok); same result layout, workspace protocol and call sites as iswm_conv2d_wgrad. */
int iswm_conv2d_wgrad_planes_ok(const iswm_conv_desc* d);
size_t iswm_conv2d_wgrad_planes_workspace(const iswm_conv_desc* d);
int iswm_conv2d_wgrad_planes(const iswm_conv_desc* d, const void* xp, int64_t x_ps, const void* dyp, int64_t dy_ps, float* dw,
                             float* workspace, size_t workspace_bytes, iswm_stream_t stream);
/* ---- depthwise convolution (groups == channels): first half of AtrousSeparableConvolution,
 * network/_deeplab.py:95-119.  The descriptor has Cin == Cout == channel count of the (possibly zero-padded)
 * activation; w is the torch parameter [Cw][1][KH][KW] as stored, Cw <= Cin (extra channels see zero weights).
 * HBM-bound streaming kernels; the weight gradient reduces in a fixed order (workspace of doubles). */
int iswm_dwconv2d_fwd(const iswm_conv_desc* d, const float* x, const float* w, int Cw, const float* bias, float* y,
                      iswm_stream_t stream);
int iswm_dwconv2d_dgrad(const iswm_conv_desc* d, const float* dy, const float* w, int Cw, float* dx, int accumulate,
                        iswm_stream_t stream);
size_t iswm_dwconv2d_wgrad_workspace(const iswm_conv_desc* d);
int iswm_dwconv2d_wgrad(const iswm_conv_desc* d, const float* x, const float* dy, int Cw, float* dw, void* workspace,
                        size_t workspace_bytes, iswm_stream_t stream);
/* dw[Cout][KH][KW][Cin] = sum over pixels.  workspace holds split-K slabs. */
size_t iswm_conv2d_wgrad_workspace(const iswm_conv_desc* d);
int iswm_conv2d_wgrad(const iswm_conv_desc* d, const float* x, const float* dy, float* dw,
                      float* workspace, size_t workspace_bytes, iswm_stream_t stream);

/* ---- BatchNorm2d (training and eval), ReLU, residual add ---------------------
 * nn.BatchNorm2d eps 1e-5 momentum 0.1 + nn.ReLU + FloatFunctional.add:
 * network/backbone/resnet.py:99-120, network/_deeplab.py:38-39,125-126,135-136. */
/* Tile statistics.  partials[2][tiles][C]: [0] = per-tile column sum S_t, [1] = per-tile
 * sum of squared deviations from the TILE mean (M2_t).  Tile t covers rows
 * [t*tile_rows, min(M, (t+1)*tile_rows)).  The conv forward epilogue emits the same pair
 * with tile_rows = 128 (iswm_conv2d_stat_tiles); iswm_colstat computes it for any [M, C]
 * (pitch ld) tensor with tiles = ceil(M / iswm_colstat_tile_rows(M)) <= iswm_colstat_tiles(M). */
int iswm_colstat_tiles(int64_t M);
int64_t iswm_colstat_tile_rows(int64_t M);
int iswm_colstat(const float* x, int64_t M, int C, int ld, float* partials, iswm_stream_t stream);
/* merge tiles (pairwise/Chan update in double) -> batch mean / biased var, update running
 * stats (unbiased var, momentum), emit scale = gamma*invstd, shift = beta, and save mean /
 * invstd (iswm_bn_apply evaluates (y - mean)*scale + shift). */
int iswm_bn_finalize(const float* partials, int tiles, int C, int64_t count, int64_t tile_rows,
                     const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                     float eps, float* scale, float* shift, float* save_mean, float* save_invstd,
                     iswm_stream_t stream);
/* eval mode: scale/shift from the running statistics */
int iswm_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* scale, float* shift,
                        float* save_mean, float* save_invstd, iswm_stream_t stream);
/* out = act((y - mean[c])*scale[c] + shift[c] (+ residual)); act = relu if relu != 0  * relu: 0 none, 1 ReLU, 6 ReLU6 (clamp to [0, 6]; its backward passes the gradient where 0 < out < 6). */
int iswm_bn_apply(const float* y, int64_t M, int C, int ldy, const float* scale, const float* shift,
                  const float* mean, const float* residual, int ldr, int relu, float* out, int ldo,
                  iswm_stream_t stream);
/* backward of out = act(BN(y) (+ residual)):  dz = dout * (out > 0 if relu);
 *   dbeta = sum dz, dgamma = sum dz*xhat (two deterministic stages, accumulated in double as
 *   ATen's CPU kernel does);
 *   dy = gamma*invstd*(dz - dbeta/M - xhat*dgamma/M) (training) or gamma*invstd*dz (eval);
 *   dres (optional) = dz, the gradient of the identity branch.
 * workspace: iswm_bn_bwd_workspace(M, C) bytes, 16-byte aligned.  * mask_scale / mask_shift (optional, ReLU without residual): the forward's scale and shift -- the ReLU sign pattern is
 * then recomputed as (y - mean)*scale + shift > 0 (bit-identical to iswm_bn_apply) and `out` is not read. */
size_t iswm_bn_bwd_workspace(int64_t M, int C);
int iswm_bn_backward(const float* dout, int ldd, const float* out, int ldo, const float* y, int ldy,
                     int64_t M, int C, const float* mean, const float* invstd, const float* gamma,
                     const float* mask_scale, const float* mask_shift,
                     int relu, int training, float* dgamma, float* dbeta, float* dy, int lddy,
                     float* dres, int lddres, void* workspace, size_t workspace_bytes,
                     iswm_stream_t stream);
/* out[c] = sum over tiles of partials[0][t][c] (bias gradient from iswm_colstat partials);
 * scratch: C floats */
int iswm_colsum_finalize(const float* partials, int tiles, int C, float* out, float* scratch,
                         iswm_stream_t stream);

/* ---- pooling ------------------------------------------------------------------ */
/* nn.MaxPool2d(3, 2, 1), network/backbone/resnet.py:148.  idx[n,ho,wo,c] = winning
 * tap 0..8 (first max in scan order, as ATen). */
int iswm_maxpool3x3s2_fwd(const float* x, int N, int H, int W, int C, float* y, uint8_t* idx, int Ho,
                          int Wo, iswm_stream_t stream);
int iswm_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, int N, int H, int W, int C, int Ho,
                          int Wo, float* dx, iswm_stream_t stream);
/* nn.AdaptiveAvgPool2d(1), network/_deeplab.py:133: y[n,c] = mean over HW */
int iswm_gap_fwd(const float* x, int N, int HW, int C, int ldx, float* y, iswm_stream_t stream);
/* dx[n,p,c] (+)= dy[n,c]/HW  (accumulate != 0 adds into dx) */
int iswm_gap_bwd(const float* dy, int N, int HW, int C, float* dx, int lddx, int accumulate,
                 iswm_stream_t stream);
/* broadcast a [N,C] vector over HW pixels (bilinear upsample of a 1x1 map,
 * network/_deeplab.py:141) into a channel slice, and its backward (sum over HW) */
int iswm_bcast_fwd(const float* v, int N, int HW, int C, float* y, int ldy, iswm_stream_t stream);
int iswm_bcast_bwd(const float* dy, int lddy, int N, int HW, int C, float* dv, iswm_stream_t stream);

/* ---- bilinear resize, align_corners=False ------------------------------------
 * F.interpolate call sites: network/_deeplab.py:58, network/utils.py:22. */
int iswm_bilinear_fwd(const float* x, int N, int Hi, int Wi, int C, int ldx, float* y, int Ho, int Wo,
                      int ldy, iswm_stream_t stream);
int iswm_bilinear_bwd(const float* dy, int N, int Hi, int Wi, int C, int lddy, int Ho, int Wo, float* dx,
                      int lddx, iswm_stream_t stream);
/* final upsample fused with the NHWC -> NCHW layout change of the logits
 * (network/utils.py:22-24): x NHWC [N,Hi,Wi,ldx] (first C channels) -> y NCHW */
int iswm_bilinear_nhwc_to_nchw_fwd(const float* x, int N, int Hi, int Wi, int C, int ldx, float* y,
                                   int Ho, int Wo, iswm_stream_t stream);
int iswm_bilinear_nhwc_to_nchw_bwd(const float* dy, int N, int Hi, int Wi, int C, int lddx, int Ho,
                                   int Wo, float* dx, iswm_stream_t stream);

/* ---- layout / elementwise helpers ---------------------------------------------- */
/* NCHW [N,C,H,W] -> NHWC with Cp >= C channels (extra channels zero) */
int iswm_nchw_to_nhwc(const float* x, int N, int C, int HW, float* y, int Cp, iswm_stream_t stream);
int iswm_nhwc_to_nchw(const float* x, int N, int C, int HW, int ldx, float* y, iswm_stream_t stream);
/* copy C channels between pitched [M, *] buffers (torch.cat / its backward,
 * network/_deeplab.py:59,171) */
int iswm_copy_channels(const float* src, int lds, float* dst, int ldd, int64_t M, int C,
                       iswm_stream_t stream);
int iswm_add_inplace(float* dst, const float* src, int64_t n, iswm_stream_t stream);
int iswm_scale_inplace(float* x, int64_t n, const float* scalar_dev, float host_mul, iswm_stream_t stream);
/* The heads' last two layers with the 1x1 classifier folded into the BatchNorm passes of the stage in front of it --
 * Conv2d(256, 256, 3) -> BatchNorm2d -> ReLU -> Conv2d(256, num_classes, 1), network/_deeplab.py:44-52 (DeepLabHeadV3Plus) and
 * :84-90 (DeepLabHead): forward = one pass over the raw conv output y to logits [M][4] (the 256-channel activation is never
 * stored); backward = the stage's BatchNorm backward fed by dlogit [M][4] and the zero-padded classifier weight wc4 [4][C],
 * returning the classifier's weight gradient dwc4 with it.  C == 256, num_classes <= 4. */
int iswm_bn_apply_classify(const float* y, int64_t M, int C, int ldy, const float* scale, const float* shift, const float* mean,
                           const float* wc4, const float* bias4, float* logits, int ldl, iswm_stream_t stream);
size_t iswm_bn_classify_bwd_workspace(int64_t M, int C);
int iswm_bn_backward_classify(const float* dlogit, int ldl, const float* wc4, const float* y, int ldy, int64_t M, int C,
                              const float* mean, const float* invstd, const float* gamma, const float* mask_scale,
                              const float* mask_shift, int training, float* dgamma, float* dbeta, float* dwc4, void* dy, int lddy,
                              int64_t dy_ps, void* workspace, size_t workspace_bytes, iswm_stream_t stream);
/* Convolutions whose channel counts are not multiples of the kernels' granule (nn.Conv2d(3, 64, 7) at network/backbone/resnet.py:137,
 * Conv2d(304, 256, 3) / Conv2d(256, 48, 1) / Conv2d(256, num_classes, 1) at network/_deeplab.py:36-52): zero-padded OHWI copy of an OIHW
 * parameter with element strides[4] = (O, I, H, W); the inverse for its gradient; zero bytes [byte0, byte1) of every row of a
 * (planes or fp32) activation buffer -- the padding channels themselves.  They replace torch.zeros / slice-copy / zero_() on the step. */
int iswm_pad_weights(const float* w, int cout, int cin, int kh, int kw, const int64_t* strides, int cout_p, int cin_p, float* out_ohwi,
                     iswm_stream_t stream);
int iswm_unpad_weights(const float* dw_ohwi, int cout_p, int cin_p, int cout, int cin, int kh, int kw, float* grad,
                       const int64_t* strides, iswm_stream_t stream);
/* the flat gradient arena of the fused optimizers before a backward pass (optimizer.zero_grad(), train.py:607) */
int iswm_fill_zero(void* p, size_t bytes, iswm_stream_t stream);
int iswm_zero_cols(void* y, int64_t M, int64_t ld_bytes, int byte0, int byte1, int64_t plane_bytes, int nplanes, iswm_stream_t stream);
/* nn.Dropout(p), network/_deeplab.py:165: counter-based Philox mask */
int iswm_dropout_fwd(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed,
                     uint64_t offset, iswm_stream_t stream);
int iswm_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p,
                     iswm_stream_t stream);

/* ---- fused weighted cross-entropy / focal loss ---------------------------------
 * nn.CrossEntropyLoss(weight, ignore_index=255), train.py:454-459 and
 * FocalLoss.forward, utils/loss.py:23-35.  logits NCHW [B,C,H,W]; labels uint8 or
 * int64 [B,H,W].  mode 0: weighted mean (CE); 1: focal mean over all pixels;
 * 2: focal sum.  One pass writes the UNNORMALISED gradient and per-block partial
 * sums {sum w*nll or sum focal, sum w}; iswm_loss_finalize reduces them into
 * sums[2] and loss[1] (deterministic two-stage reduction). */
int iswm_loss_blocks(int64_t npix);
int iswm_loss_fwd(const float* logits, const void* labels, int label_bytes, int B, int C, int64_t HW,
                  const float* class_weight, int ignore_index, float alpha, float gamma, int mode,
                  float* grad_unnorm, float* partials, iswm_stream_t stream);
int iswm_loss_finalize(const float* partials, int blocks, int mode, int64_t npix, float* sums,
                       float* loss, iswm_stream_t stream);
/* grad *= upstream * (mode 0: 1/sums[1]; mode 1: 1/npix; mode 2: 1) */
int iswm_loss_bwd_scale(float* grad, int64_t n, const float* sums, const float* upstream, int mode,
                        int64_t npix, iswm_stream_t stream);
/* logits.max(1)[1], train.py:644,659 -- ties resolve to the lowest class index */
int iswm_argmax_nchw(const float* logits, int B, int C, int64_t HW, int64_t* out, iswm_stream_t stream);

/* ---- validation metrics ------------------------------------------------------------
 * StreamMetrics._fast_hist, metrics/stream_metrics.py:24-31: hist[n_classes][n_classes] (int64, row = ground
 * truth, column = prediction) += bincount(n_classes*label + pred) over the pixels whose label lies in
 * [0, n_classes) (255 = ignore falls outside).  Exact integer counting; accumulates into hist.
 * dtype codes: 0 = uint8, 1 = int64.  The _logits form takes the prediction as logits.max(1)[1]
 * (train.py:644,659) without materialising the mask. */
int iswm_confusion_matrix(const void* labels, int label_dtype, const void* preds, int pred_dtype, int64_t npix,
                          int n_classes, int64_t* hist, iswm_stream_t stream);
int iswm_confusion_matrix_logits(const void* labels, int label_dtype, const float* logits, int B, int C, int64_t HW,
                                 int n_classes, int64_t* hist, iswm_stream_t stream);

/* ---- training-input pipeline -------------------------------------------------------
 * The reference's per-sample PIL chain ExtRandomScale -> ExtRandomCrop(pad_if_needed) -> ExtRandomHorizontalFlip ->
 * ExtToTensor -> ExtNormalize (train.py:355-362, utils/ext_transforms.py:94-115,212-396) for a whole batch in one
 * launch: uint8 HWC source images + uint8 HW labels in, normalised fp32 NCHW batch + uint8 label batch out.
 * Bit-exact with Pillow's BILINEAR (image) / NEAREST (label) resize: the host computes Pillow's per-column /
 * per-row bounds, 22-bit fixed-point weights and nearest-source indices (iswm_amd/utils/ext_transforms.py) and
 * passes them in `tables`: per sample, at tab_off (ints):
 *   xintab[rs_w] | yintab[rs_h] | hbounds[rs_w][2] (first source column, count) | hk[rs_w][ksize_h] |
 *   vbounds[rs_h][2] | vk[rs_h][ksize_v].
 * images / labels / samples / tables / outputs are device pointers; mean3 / std3 are HOST arrays of 3 floats. */
typedef struct iswm_aug_sample {
    long long img_off;      /* byte offset of this sample's uint8 [src_h][src_w][3] image in `images` */
    long long lbl_off;      /* byte offset of its uint8 [src_h][src_w] label in `labels` */
    int src_h, src_w;
    int rs_h, rs_w;         /* size after the random rescale */
    int pad;                /* border added on every side by pad_if_needed (0: none) */
    int crop_i, crop_j;     /* crop origin (row, column) in the padded image */
    int flip;               /* horizontal flip after the crop */
    int tab_off;
    int ksize_h, ksize_v;
    int reserved;
} iswm_aug_sample;
int iswm_augment_batch(const unsigned char* images, const unsigned char* labels, const void* samples,
                       const int* tables, int B, int crop_h, int crop_w, const float* mean3, const float* std3,
                       float* out_nchw, unsigned char* out_labels, iswm_stream_t stream);

/* ---- optimizers over a flat fp32 arena ------------------------------------------
 * torch.optim.SGD(momentum=0.9, nesterov=True, weight_decay) / Adam / AdamW as built
 * by setup_optimizer, train.py:421-444.  lr is read from device memory so that a
 * captured graph can be replayed with a new learning rate. */
int iswm_sgd_step(float* p, const float* g, float* buf, int64_t n, const float* lr_dev, float momentum,
                  float weight_decay, int nesterov, iswm_stream_t stream);
/* hyper_dev[4] = {lr, bias_correction1, bias_correction2, unused} */
int iswm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev,
                   float beta1, float beta2, float eps, float weight_decay, int decoupled,
                   iswm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ISWM_HIP_H */
