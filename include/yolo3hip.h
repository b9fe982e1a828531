/*
 * yolo3hip.h -- C ABI of libyolo3hip.so, the MI355X (gfx950) hot path of the
 * YOLOv3 train / inference pipeline of usnistgov/object-detection-yolov3.
 *
 * The reference has no FFI of its own: its hot path is TensorFlow ops called
 * from Python (model.py) plus NumPy post-processing (bbox_utils.py).  Each
 * entry point below names the reference call site (file:line under the
 * reference checkout) whose arithmetic it replaces.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is DEVICE memory unless it says host
 *   - activations are NHWC fp32 with an explicit row pitch `ld` (floats per
 *     pixel, >= channels) so producers can write straight into concat buffers
 *   - conv kernels are Keras layout [kh][kw][Cin][Cout] (= [tap][Cin][Cout])
 *   - all work is enqueued on `stream` (a hipStream_t) and returns at once
 *   - return 0 on success, negative Y3_E* on error; y3_last_error() = message
 *   - no entry point allocates, frees or synchronises (graph-capture safe)
 */
#ifndef YOLO3HIP_H
#define YOLO3HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Y3_OK 0
#define Y3_EINVAL (-1)   /* bad argument / unsupported shape */
#define Y3_ELAUNCH (-2)  /* HIP launch error */

typedef void* y3_stream_t; /* hipStream_t */

const char* y3_last_error(void);
/* Diagnostics: x / d computed the way the kernels' index decode does it (multiply-high + shift, common.h y3_make_div),
 * on the host; 0 <= x < 2^31, d >= 1.  The CPU tests compare it with the integer division it replaces. */
int y3_debug_div(int x, int d);
int y3_version(void);

/* ---- epilogue flags of y3_conv2d_fwd / y3_conv2d_dgrad ------------------ */
#define Y3_EPI_LRELU 1u  /* v = v > 0 ? v : alpha*v   (tf.nn.leaky_relu, model.py:34) */
#define Y3_EPI_ACCUM 2u  /* dst += v instead of dst = v */
/* Arithmetic selector of y3_conv2d_fwd / y3_conv2d_dgrad / y3_conv2d_dgrad_bn (conv_x3.hip).  Without it the contraction runs on
 * v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain).  With it every operand element is cut into three bf16 pieces that sum to it
 * exactly (8 + 8 + 8 significant bits, round to nearest), the six piece pairs that carry more than 2^-26 of a product go through
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32-class results (not bit-identical to the fmaf chain; tests/test_gpu_kernels.py
 * bounds the error against fp64 by that of the fp32 instruction) at 2.67x fewer matrix-pipe cycles.  THE WEIGHT OPERAND CHANGES
 * with the flag: the kernel needs K contiguous per output column AND the weights already split, so `wt` / `wt_t` then point at the
 * THREE bf16 PLANES y3_x3_split_weights() makes of the copy with that layout -- y3_conv2d_fwd: of [tap][Cout][Cin] (what
 * y3_transpose_weights writes), y3_conv2d_dgrad*: of [tap][Cin][Cout] (the Keras kernel); each entry point the copy the OTHER one
 * takes without the flag.  (The activations are split inside the kernels.)  Shapes: y3_conv2d_x3_ok() / y3_conv2d_dgrad_x3_ok().
 * Everything else -- epilogue, statistics, split-K workspace contract -- is unchanged; ask the *_x queries for tile
 * counts and workspace sizes.  y3_conv2d_wgrad_x takes the same flag (both its operands are activations: no planes involved). */
#define Y3_CONV_X3 4u
/* The weight operand of the Y3_CONV_X3 forward / data-gradient kernels.  w: a kernel with K contiguous per row, [taps][rows][k_per_row]
 * fp32 (forward: rows = Cout, k_per_row = Cin, i.e. what y3_transpose_weights writes; data gradient: rows = Cin, k_per_row = Cout,
 * the Keras kernel); k_per_row a multiple of 16.  planes (bf16, 3 * taps * rows * k_per_row elements, 16-byte aligned):
 *     planes[(((tap * k_per_row/16 + c/16) * rows + row) * 3 + piece) * 16 + c % 16] = piece `piece` of w[tap][row][c],
 * w = piece 0 + piece 1 + piece 2 exactly (each the round-to-nearest bf16 of what the earlier ones leave): the block one K step of
 * the kernels reads -- rows x 3 pieces x 16 k -- is contiguous. */
int y3_x3_split_weights(const float* w, void* planes, int taps, int rows, int k_per_row, y3_stream_t stream);
/* The same for every kernel of a parameter arena in ONE launch.  table_dev: DEVICE int32 [nlayers][5] = {arena offset of the
 * layer's kernel (floats), taps, rows, k_per_row, index of the layer's first block}, blocks of 1024 elements; total_blocks = sum over
 * layers of ceil(taps * rows * k_per_row / 1024).  The planes of a layer are written at planes_arena + 3 * offset (bf16 elements). */
int y3_x3_split_weights_batched(const float* arena, void* planes_arena, const int* table_dev, int nlayers, int total_blocks, y3_stream_t stream);
/* y3_transpose_weights_batched and the two y3_x3_split_weights_batched launches of an optimiser step in ONE pass over the arena:
 * params_t <- the transposed kernels, planes <- piece planes of the Keras copy, planes_t <- piece planes of the transposed copy
 * (each for the layers whose K per row is a multiple of 16; layouts as above, a layer's planes at 3 x its arena offset).
 * table_dev / nlayers / total_tiles: as for y3_transpose_weights_batched.  Bit-identical to the three launches it replaces. */
int y3_x3_prepare_weights_batched(const float* params, float* params_t, void* planes, void* planes_t, const int* table_dev, int nlayers,
                                  int total_tiles, y3_stream_t stream);

/*
 * Tensor view: NHWC, `ld` floats between consecutive pixels.
 */
typedef struct y3_tensor {
    float* ptr;
    int n, h, w, c;
    int ld;
} y3_tensor;

/*
 * conv_layer forward (model.py:29-39 Conv2D part, :108-120 detection_layer).
 *   dst = epi( conv_SAME(src, wt) + bias )
 * epi: optional leaky-relu, then optional per-channel affine v*scale+shift
 * (inference-mode BatchNorm folded), then optional residual add (model.py:47).
 * If `stats` != NULL the kernel also writes per-row-tile partial sums of the
 * post-activation value: stats[tile][0][c] = sum, stats[tile][1][c] = sum of
 * squares (training-mode BatchNorm statistics, model.py:38); the number of
 * tiles is y3_conv2d_stats_tiles().
 * TF 'same' padding (pad_before = pad_total/2, extra at the end).
 */
int y3_conv2d_fwd(const y3_tensor* src, const float* wt, const float* bias, int ksize, int stride,
                  const y3_tensor* dst, unsigned flags, float alpha,
                  const float* scale, const float* shift, const y3_tensor* resid,
                  float* stats, void* workspace, size_t workspace_bytes, y3_stream_t stream);
/* m = output pixels (N*OH*OW).  Launches whose tile count does not fill the 256 CUs evenly are split along K (all
 * tiles, or only the remainder round); the slices park raw partial sums in `workspace` and the slice of a tile that
 * finishes last reduces them in slice order inside the same kernel (bit-reproducible; no second launch).
 * WORKSPACE CONTRACT (y3_conv2d_fwd / _dgrad / _wgrad): the first 256 KiB of a workspace are per-tile tickets.  Zero
 * the workspace once after allocating it (hipMemset, or y3_fill); every launch leaves the tickets at zero.  Launches
 * that share a workspace must be ordered on one stream.  Passing less than y3_conv2d_*_workspace() bytes disables
 * the split (forward / data gradient) or is an error (kernel gradient).  A header that is NOT zero makes the affected
 * tiles keep stale output without any error (no slice draws the last ticket); to find such a caller run with the
 * environment variable Y3_CHECK_TICKETS=1: every ticketed launch then synchronises its stream, reads the header back and
 * fails with Y3_EINVAL / y3_last_error() if a ticket is non-zero (debug aid: it serialises the stream; launches on a stream
 * under graph capture are not checked). */
int y3_conv2d_stats_tiles(int m, int cin, int ksize, int cout);
size_t y3_conv2d_fwd_workspace(int m, int cin, int ksize, int cout);
/* The same two queries for a launch with `flags` (Y3_CONV_X3 changes the tile and the split-K plan). */
int y3_conv2d_stats_tiles_x(int m, int cin, int ksize, int cout, unsigned flags);
size_t y3_conv2d_fwd_workspace_x(int m, int cin, int ksize, int cout, unsigned flags);
/* 1 if the Y3_CONV_X3 kernels take an implicit GEMM of m rows, `ntaps` taps of `c` contracted channels each and `nout` output
 * columns (forward: c = Cin, nout = Cout; stride-1 data gradient: c = Cout, nout = Cin): c a multiple of 16 (a power of two when
 * ntaps > 1), nout >= 32. */
int y3_conv2d_x3_ok(int m, int c, int ntaps, int nout);
/* 1 if y3_conv2d_dgrad / y3_conv2d_dgrad_bn take this data gradient with Y3_CONV_X3: stride 1 as y3_conv2d_x3_ok(m, Cout, ksize^2,
 * Cin); stride 2 (3x3: the merged launch of the four output-parity classes, each with its own K slices): Cin >= 64, Cout a power of
 * two >= 32.  (tape.gradient through the stride-2 convs of darknet-53, model.py:496 / :362-366.) */
int y3_conv2d_dgrad_x3_ok(const y3_tensor* ddst, int ksize, int stride, const y3_tensor* dsrc);
/* Diagnostics (host only, no launch): the plan y3_conv2d_fwd and the stride-1 y3_conv2d_dgrad use for an implicit GEMM of
 * m x cout x (ksize^2 * cin).  out13 = {bm, bn, bk, tiles, f, s0, s1, chunk0, chunk1, grid, stats_tiles, fast, nk}: tiles
 * [0, f) are cut into s0 K slices of chunk0 K steps, tiles [f, tiles) into s1 of chunk1 (nk K steps in all); grid = work
 * items = workgroups; fast: bit 0 = the MFMA kernel with split-K takes the launch, bit 1 = x3 plan whose short last slices are
 * dealt to the blocks dispatched last (a few blocks more than workgroup slots).  Returns the workspace bytes (= y3_conv2d_fwd_workspace).  tests/planner_sweep.cpp replays the
 * kernel's item -> (tile, slice, slab, ticket) mapping from these numbers under AddressSanitizer. */
size_t y3_conv2d_plan(int m, int cin, int ksize, int cout, int* out13);
size_t y3_conv2d_plan_x(int m, int cin, int ksize, int cout, unsigned flags, int* out13);     /* the plan of a launch with `flags` (Y3_CONV_X3) */

/*
 * Gradient w.r.t. the conv input (tape.gradient, model.py:496):
 *   dsrc (+)= conv_transpose(ddst, wt)
 * wt_t is the kernel with the two channel axes swapped: [tap][Cout][Cin]
 * (y3_transpose_weights).  ddst is the gradient at the conv output (dst of
 * the forward), dsrc has the forward's src geometry.
 */
int y3_conv2d_dgrad(const y3_tensor* ddst, const float* wt_t, int ksize, int stride,
                    const y3_tensor* dsrc, unsigned flags, void* workspace, size_t workspace_bytes,
                    y3_stream_t stream);
size_t y3_conv2d_dgrad_workspace(const y3_tensor* ddst, int ksize, int stride, const y3_tensor* dsrc);
size_t y3_conv2d_dgrad_workspace_x(const y3_tensor* ddst, int ksize, int stride, const y3_tensor* dsrc, unsigned flags);
/*
 * y3_conv2d_dgrad whose epilogue also sums the six raw moments of (dsrc after this launch, bn_a) per output column and
 * row tile -- the statistics of the BatchNorm backward of the layer that PRODUCED dsrc's activation (bn_a = that layer's
 * lrelu(z), geometry of dsrc).  Use it for the launch that completes dsrc (the last accumulation).  Fast-path channel
 * counts only; stride 2 (3x3) only when the four parity classes go out as one merged launch: y3_conv2d_dgrad_bn_tiles()
 * returns the number of partial rows (row tiles, over all classes; partials: rows * 6 * dsrc->c floats, 16-byte aligned),
 * or 0 when the shape does not qualify.  Same workspace as y3_conv2d_dgrad.
 */
int y3_conv2d_dgrad_bn(const y3_tensor* ddst, const float* wt_t, int ksize, int stride, const y3_tensor* dsrc,
                       unsigned flags, const y3_tensor* bn_a, float* bn_partials,
                       void* workspace, size_t workspace_bytes, y3_stream_t stream);
int y3_conv2d_dgrad_bn_tiles(const y3_tensor* ddst, int ksize, int stride, const y3_tensor* dsrc);
int y3_conv2d_dgrad_bn_tiles_x(const y3_tensor* ddst, int ksize, int stride, const y3_tensor* dsrc, unsigned flags);

/*
 * Gradient w.r.t. the kernel:  dw[tap][ci][co] = sum_pixels src*ddst.
 * The pixel axis is split over workgroups.  With 2..8 splits the last split of a (k-tile, n-tile) to finish sums the
 * partial slabs in split order inside the kernel (tickets); with more splits (layers whose kernel matrix is small and
 * whose pixel count is large) every split writes a natural-layout slab and slab_reduce_kernel follows in the same call.
 * Either way the sum order is fixed (bit-reproducible).  workspace: y3_conv2d_wgrad_workspace() bytes, zeroed once
 * (contract above).
 */
int y3_conv2d_wgrad(const y3_tensor* src, const y3_tensor* ddst, int ksize, int stride,
                    float* dw, void* workspace, size_t workspace_bytes, y3_stream_t stream);
size_t y3_conv2d_wgrad_workspace(const y3_tensor* src, const y3_tensor* ddst, int ksize, int stride);
/* The same with `flags`: Y3_CONV_X3 runs the contraction over the pixels as three bf16 pieces per operand (conv_wgrad_x3_kernel:
 * both operands are activations, so both are split in the kernel; 128 x 128 tiles of the [K][Nout] gradient, fragments read with
 * ds_read_b64_tr_b16).  Same operands, layouts, workspace contract and reduction order rules as y3_conv2d_wgrad; shapes:
 * y3_conv2d_wgrad_x3_ok() (K = ksize^2 * Cin >= 128, Cout >= 128, Cin a power of two when ksize = 3). */
int y3_conv2d_wgrad_x(const y3_tensor* src, const y3_tensor* ddst, int ksize, int stride,
                      float* dw, unsigned flags, void* workspace, size_t workspace_bytes, y3_stream_t stream);
size_t y3_conv2d_wgrad_workspace_x(const y3_tensor* src, const y3_tensor* ddst, int ksize, int stride, unsigned flags);
int y3_conv2d_wgrad_x3_ok(int m, int cin, int ksize, int cout);
/* Diagnostics: the kernel-gradient plan for m output pixels.  out8 = {bkr, bn, splits, chunk, tiles, in_kernel, grid,
 * pixel_table}: the m pixels are cut into `splits` runs of `chunk`; in_kernel = 1 when the last split of a tile reduces
 * the slabs inside the kernel (splits <= 8), else slab_reduce_kernel follows.  Returns the workspace bytes. */
size_t y3_conv2d_wgrad_plan(int m, int cin, int ksize, int cout, int* out8);
size_t y3_conv2d_wgrad_plan_x(int m, int cin, int ksize, int cout, unsigned flags, int* out8);

/* wt_t[tap][co][ci] = wt[tap][ci][co] */
int y3_transpose_weights(const float* wt, float* wt_t, int taps, int cin, int cout, y3_stream_t stream);
/* The same for every layer of a parameter arena in ONE launch.  table_dev: DEVICE int32 [nlayers][5] =
 * {arena offset (floats), taps, cin, cout, index of the layer's first 32x32 tile}; total_tiles = sum over layers of
 * taps * ceil(cin/32) * ceil(cout/32). */
int y3_transpose_weights_batched(const float* params, float* params_t, const int* table_dev, int nlayers,
                                 int total_tiles, y3_stream_t stream);

/* ---- BatchNormalization(axis=1), eps 1e-3, momentum .99 (model.py:38) ---- */
/*
 * Training forward, step 1: reduce conv-epilogue partials to batch mean and
 * biased variance; write scale = gamma*rsqrt(var+eps), shift = beta-mean*scale;
 * save mean / rstd for backward; update moving stats
 * (moving = moving*mom + batch*(1-mom), variance Bessel-corrected).
 */
int y3_bn_stats_finalize(const float* stats, int tiles, int c, int count,
                         const float* gamma, const float* beta, float eps, float momentum,
                         float* moving_mean, float* moving_var,
                         float* save_mean, float* save_rstd, float* scale, float* shift,
                         y3_stream_t stream);
/* moving-stats (inference) affine: scale = gamma*rsqrt(var+eps), shift = beta-mean*scale */
int y3_bn_fold_inference(const float* gamma, const float* beta, const float* moving_mean,
                         const float* moving_var, float eps, int c, float* scale, float* shift,
                         y3_stream_t stream);
/* The same for every BatchNorm layer in ONE launch.  table_dev: DEVICE int32 [nlayers][7] = float offsets of
 * {gamma, beta} in `params`, {moving mean, moving var} in `moving`, {scale, shift} in `chan`, and the channel count. */
int y3_bn_fold_inference_batched(const float* params, const float* moving, float* chan, const int* table_dev,
                                 int nlayers, float eps, y3_stream_t stream);
/* y = a*scale + shift (+ resid)   (BN apply and the residual add of model.py:47) */
int y3_bn_apply(const y3_tensor* a, const float* scale, const float* shift, const y3_tensor* resid,
                const y3_tensor* y, y3_stream_t stream);
/*
 * Backward of conv_layer's tail  y = BN(lrelu(z)):  given dy and a = lrelu(z)
 *   step 1 (stats):  per-channel raw moments of (dy, a) in fp64 (workspace), then -- by the workgroup whose partial
 *                    arrives last, inside the same launch -- dgamma, dbeta, dbias (bias gradient of the conv) and the
 *                    per-channel coefficients k1,k2,k3 (coef[3][c]).  Optionally the residual fan-in of model.py:47
 *                    rides along while dy streams through: dres = dy (dres_accumulate == 0) or dres += dy.
 *                    Channels: a multiple of 64 up to 1024, or 4 / 8 / 16 / 32 (every layer of this network).
 *                    `workspace`: y3_bn_bwd_workspace(m, c) bytes, 16-byte aligned; its first 1 KiB (tickets) must be
 *                    zero before the FIRST call -- every call leaves it zero again.
 *   step 2 (apply):  dz = (k1*dy + k2*a + k3) * (a > 0 ? 1 : alpha)
 */
int y3_bn_bwd_stats(const y3_tensor* dy, const y3_tensor* a, const y3_tensor* dres, int dres_accumulate,
                    const float* gamma, const float* save_mean, const float* save_rstd, float alpha,
                    float* dgamma, float* dbeta, float* dbias, float* coef,
                    void* workspace, size_t workspace_bytes, y3_stream_t stream);
size_t y3_bn_bwd_workspace(int m, int c); /* 0 if the channel count is unsupported */
int y3_bn_bwd_apply(const y3_tensor* dy, const y3_tensor* a, const float* coef, float alpha,
                    const y3_tensor* dz, y3_stream_t stream);
/* step 2 with the residual fan-in of model.py:47 riding along: dres = dy (dres_accumulate == 0) or dres += dy */
int y3_bn_bwd_apply_fanin(const y3_tensor* dy, const y3_tensor* a, const float* coef, float alpha,
                          const y3_tensor* dz, const y3_tensor* dres, int dres_accumulate, y3_stream_t stream);
/*
 * Step 1 without a pass of its own: when the LAST contribution to dy is written by a stride-1 data gradient
 * (y3_conv2d_dgrad_bn above: its epilogue has dy in registers), that launch leaves per-row-tile partial moments
 * [tiles][6][c] (fp32) and this call turns them into dgamma / dbeta / dbias / coef (fp64 across tiles).
 */
int y3_bn_bwd_finalize_tiles(const float* partials, int tiles, int c, int count, const float* gamma,
                             const float* save_mean, const float* save_rstd, float alpha,
                             float* dgamma, float* dbeta, float* dbias, float* coef, y3_stream_t stream);

/* ---- upsample_2x: frozen all-ones Conv2DTranspose k2 s2 (model.py:94-105) ---- */
/* out[n,2i+a,2j+b,co] = sum_ci in[n,i,j,ci] for every co */
int y3_upsample_sum2x_fwd(const y3_tensor* in, const y3_tensor* out, y3_stream_t stream);
/* din[n,i,j,ci] = sum_{a,b,co} dout[n,2i+a,2j+b,co] for every ci */
int y3_upsample_sum2x_bwd(const y3_tensor* dout, const y3_tensor* din, y3_stream_t stream);

/* ---- bf16 inference path (BASELINE config 5: tiled inference with a bf16 conv path) ----------------
 * Tensors are NHWC with `ld` counted in ELEMENTS; `ptr` addresses bf16 (2-byte) elements unless noted.
 * Replaces the same Conv2D -> LeakyReLU -> BatchNorm(training=False) chain as y3_conv2d_fwd
 * (model.py:71-91,108-137) when inference runs in reduced precision: bf16 operands, fp32 accumulate on
 * v_mfma_f32_32x32x16_bf16, fp32 epilogue (bias, lrelu, folded BN scale/shift, residual), one rounding on store.
 * wt_t_bf16 is the y3_transpose_weights layout [kh][kw][Cout][Cin] converted with y3_f32_to_bf16.
 * Requires Cin % 32 == 0 (the first RGB layer stays on y3_conv2d_fwd).  dst_is_f32 != 0 writes fp32
 * (used for the three head convs so that decode / NMS stay fp32).
 * Which kernel runs is the library's choice and does not change the contract: the 3x3 layers 32 -> 64 and 64 -> 128 with bf16
 * output and 16-byte aligned rows (the 608^2 -> 152^2 stages of the tiled path) take HBM-bound patch kernels (input patch once
 * through LDS, weights in registers), Cin % 64 == 0 with Cout >= 256 and >= 96 tiles of 256 x 256 the ping-pong kernel, the rest
 * the LDS-DMA ring kernel.  All of them accumulate in fp32 and round once.
 * Y3_BF16_NO_PATCH in `flags` keeps a launch off the patch kernels: they win where a layer streams from HBM (hundreds of MB:
 * batches of 25+ tiles of 608^2) and lose 2 % of a whole forward at batch 8, where the next layer finds the ring kernel's
 * output in its own XCD's L2 (the patch kernels write strip-wise).  The caller knows its batch: yolo3/model.py sets the flag for
 * layers that move less than 300 MB. */
#define Y3_BF16_NO_PATCH 8u
int y3_conv2d_fwd_bf16(const y3_tensor* src, const void* wt_t_bf16, const float* bias, int ksize, int stride,
                       const y3_tensor* dst, int dst_is_f32, unsigned flags, float alpha,
                       const float* scale, const float* shift, const y3_tensor* resid, y3_stream_t stream);
/* The same with a caller-owned workspace (y3_conv2d_fwd_bf16_workspace(m = N*OH*OW, cin, ksize, cout) bytes; zero it once, the
 * first 256 KiB are per-tile tickets (the same header as the fp32 entries, so one zeroed workspace can serve both) that every launch leaves at zero; one stream per workspace): the small-M layers
 * (<= 1 024 tiles of 64 x 64 and >= 32 K steps: the 13x13 / 26x26 / 19x19 grids at batch 8) are then split along K into up to 8
 * slices whose fp32 partial sums the last-arriving slice adds in slice order inside the kernel (bit-reproducible).  Without a
 * workspace (or with y3_conv2d_fwd_bf16) every tile walks its whole K. */
size_t y3_conv2d_fwd_bf16_workspace(int m, int cin, int ksize, int cout);
int y3_conv2d_fwd_bf16_ws(const y3_tensor* src, const void* wt_t_bf16, const float* bias, int ksize, int stride,
                          const y3_tensor* dst, int dst_is_f32, unsigned flags, float alpha,
                          const float* scale, const float* shift, const y3_tensor* resid,
                          void* workspace, size_t workspace_bytes, y3_stream_t stream);
/* The first (RGB) conv_layer straight to bf16: src fp32 NHWC with Cin padded to 4, wt the fp32 Keras kernel
 * [3][3][4][32], 3x3 stride 1 SAME, dst bf16 with 32 channels; same epilogue order as above.  fp32-accurate: input and
 * weights are split into three bf16 pieces each and multiplied on the matrix pipe (fp32 accumulation), one rounding on the
 * store.  Any image size; src rows 16-byte aligned. */
int y3_conv2d_first_bf16(const y3_tensor* src, const float* wt, const float* bias, const y3_tensor* dst,
                         unsigned flags, float alpha, const float* scale, const float* shift, y3_stream_t stream);
int y3_f32_to_bf16(const float* src, void* dst, size_t count, y3_stream_t stream); /* round-to-nearest-even */
int y3_upsample_sum2x_fwd_bf16(const y3_tensor* in, const y3_tensor* out, y3_stream_t stream); /* model.py:94-105 */

/* ---- small data movement -------------------------------------------------- */
int y3_copy(const y3_tensor* src, const y3_tensor* dst, y3_stream_t stream);        /* strided copy (tf.concat, model.py:368,375) */
int y3_add_inplace(const y3_tensor* src, const y3_tensor* dst, y3_stream_t stream); /* dst += src (gradient fan-in) */
int y3_fill(float* ptr, size_t count, float value, y3_stream_t stream);
/* model input [N,C,H,W] -> NHWC with channels zero-padded to dst->c (multiple of 4) */
int y3_nchw_to_nhwc(const float* src, int n, int c, int h, int w, const y3_tensor* dst, y3_stream_t stream);
/* NHWC view -> dense [N,C,H,W] (feature-map export, model.py:462) */
int y3_nhwc_to_nchw(const y3_tensor* src, float* dst, y3_stream_t stream);
/* column sums: out[c] = sum_pixels src  (bias gradient of the detection layers) */
int y3_colsum(const y3_tensor* src, float* out, y3_stream_t stream);

/* ---- anchor decode: reorg_layer + convert_feature_map_to_inference_detections
 *      (model.py:122-212).  fm[s]: NHWC [N,G,G,A*(5+K)]; out: [N,Nb,5+K] rows
 *      [x0,y0,x1,y1,obj,cls..], scales coarse->fine, then row, col, anchor.
 *      anchors: host [A][2] (w,h).  stride quirk Q6 reproduced:
 *      x uses img_h/grid_h, y uses img_w/grid_w. */
int y3_decode_fwd(const y3_tensor* fm, int nscales, const float* anchors_host, int num_anchors,
                  int num_classes, int img_h, int img_w, float* out, y3_stream_t stream);

/* ---- loss_layer forward + backward for one scale (model.py:230-354) --------
 * fm NHWC [N,G,G,A*(5+K)], gt dense [N,G,G,A,5+K].  Adds the four terms
 * (xy, wh, obj, class; each / local batch) into loss4[0..3] and writes
 * dfm = d(total/global_batch)/d(fm) with fm's layout.  loss4 is accumulated
 * (zero it before the first scale).  workspace: y3_loss_workspace_bytes().
 */
int y3_loss_fwd_bwd(const y3_tensor* fm, const float* gt, const float* anchors_host, int num_anchors,
                    int num_classes, int img_h, int img_w, float global_batch,
                    float* loss4, const y3_tensor* dfm, void* workspace, y3_stream_t stream);
size_t y3_loss_workspace_bytes(void);

/* ---- tf.keras.optimizers.Adam.apply_gradients (model.py:451,500) ----------
 * m += (g-m)(1-b1); v += (g*g-v)(1-b2); p -= lr_t*m/(sqrt(v)+eps), with
 * lr_t read from DEVICE memory (*lr_t_dev) so the launch is graph-replayable. */
int y3_adam_step(float* param, const float* grad, float* m, float* v, size_t count,
                 const float* lr_t_dev, float beta1, float beta2, float eps, y3_stream_t stream);

/* ---- class-wise NMS (bbox_utils.py:200-281; inference.py:72-79) ------------
 * rows [N,Nb,5+K].  A row is a candidate of class c if w > min_box and
 * h > min_box (strict) and sqrt(cls_c*obj) >= score_thr; greedy suppression
 * keeps iou <= iou_thr, in descending score order (ties: higher row index
 * first).  Optional clip of the corners to [0,clip_w]x[0,clip_h] before
 * everything (inference.py:62-65 intent); pass clip_w <= 0 to disable.
 * keep_idx [N,K,max_keep] int32 row indices in selection order,
 * keep_cnt [N,K] int32, keep_score [N,K,max_keep] fp32.
 * workspace: y3_nms_workspace_bytes(). */
int y3_nms_per_class(const float* rows, int n, int nb, int num_classes, float min_box,
                     float score_thr, float iou_thr, float clip_w, float clip_h,
                     int* keep_idx, int* keep_cnt, float* keep_score, int max_keep,
                     void* workspace, size_t workspace_bytes, y3_stream_t stream);
size_t y3_nms_workspace_bytes(int n, int nb, int num_classes);
/* bbox_utils.single_class_nms (bbox_utils.py:217-237): rows5 [M,5] = x0,y0,x1,y1,score; every row is
 * a candidate, the score is used as is.  keep_idx/keep_score [M], keep_cnt [1].
 * workspace: y3_nms_workspace_bytes(1, M, 1). */
int y3_nms_single_class(const float* rows5, int m, float iou_thr, int* keep_idx, int* keep_cnt,
                        float* keep_score, void* workspace, size_t workspace_bytes, y3_stream_t stream);

/* bbox_utils.filter_small_boxes (bbox_utils.py:274-281): keep_idx[0..*keep_cnt) = indices, in row order, of the rows
 * [x0,y0,x1,y1,...] (pitch ld floats) with (x1-x0) > min_size and (y1-y0) > min_size (strict, Q19).  keep_idx holds m ints. */
int y3_filter_small_boxes(const float* rows, int m, int ld, float min_size, int* keep_idx, int* keep_cnt, y3_stream_t stream);

/* bbox_utils.compute_iou (bbox_utils.py:200-214): iou[i] = IoU(box4, boxes[i*ld .. i*ld+3]), corners, no +1, fp32 in the
 * reference's operation order (0/0 -> NaN as in NumPy). */
int y3_compute_iou(const float* box4, const float* boxes, int m, int ld, float* iou, y3_stream_t stream);

/* ---- imagereader.zscore_normalize (imagereader.py:34-46) -------------------
 * per image: mu = mean, sd = population std over all `count` values;
 * out = sd <= 1 ? x-mu : (x-mu)/sd.  workspace: y3_zscore_workspace_bytes(n). */
int y3_zscore(const float* in, float* out, int n, size_t count, void* workspace, y3_stream_t stream);
size_t y3_zscore_workspace_bytes(int n);

/* ---- inference_tiled.convert_image_to_tiles (inference_tiled.py:29-100) on the device -------------
 * img: HWC image resident in device memory, dtype 0 = uint8, 1 = uint16, 2 = float32.
 * table_dev: DEVICE int32 [ntiles][6] = {y0, ny, pre_y, x0, nx, pre_x}: tile t is the crop img[y0:y0+ny, x0:x0+nx]
 * with pre_y / pre_x reflected rows / columns in front and the rest of tile_h / tile_w reflected behind
 * (np.pad(mode='reflect'), inference_tiled.py:82-92).  out: float32 [ntiles][C][tile_h][tile_w] (astype + transpose,
 * inference_tiled.py:199-203). */
int y3_tile_gather(const void* img, int dtype, int height, int width, int channels, const int* table_dev,
                   int ntiles, int tile_h, int tile_w, float* out, y3_stream_t stream);

/* The same tiles, z-scored per tile (y3_zscore: whole-tile mean / population std, subtract-only when std <= 1) and written
 * straight into a network input buffer: out float32 NHWC [ntiles][tile_h][tile_w][channel_pitch] with the channels beyond C
 * zeroed (the layout y3_nchw_to_nhwc produces for the first conv_layer).  Two passes over the image instead of five over
 * the tiles; the results are the bits of y3_tile_gather -> y3_zscore -> y3_nchw_to_nhwc.  workspace:
 * y3_zscore_workspace_bytes(ntiles). */
int y3_tile_gather_zscore_nhwc(const void* img, int dtype, int height, int width, int channels, const int* table_dev,
                               int ntiles, int tile_h, int tile_w, float* out, int channel_pitch, void* workspace,
                               y3_stream_t stream);

/* ---- gradient exchange: tf.distribute.MirroredStrategy's all-reduce (train.py:38-39, model.py:500,510-515) -------
 * One process per GPU; SUM over the replicas (the loss is already divided by the global batch, model.py:492).  RCCL over
 * xGMI underneath (librccl.so is opened on first use).  Rank 0 calls y3_comm_unique_id and hands the 128 bytes to the
 * other ranks out of band; every rank then calls y3_comm_init with its HIP device current.  The Python host issues the
 * same collective through torch.distributed (backend "nccl" = RCCL) by default and through these entry points with
 * yolo3.parallel.DataParallel(transport='native') / Y3_DP_TRANSPORT=native (the torch group then only carries the id).
 * Exercised so far with ONE-rank communicators only (the builder's boxes have one GPU and RCCL refuses two ranks on
 * one device): N > 1 is first run by the driver's multi-GPU bench, with the default torch transport.
 * y3_comm_info: what the communicator itself reports -- ncclCommCount and ncclGetVersion (-1 where the symbol is missing). */
int y3_comm_unique_id(void* id128);
int y3_comm_init(const void* id128, int nranks, int rank, void** comm);
int y3_allreduce_sum_f32(void* comm, float* buf, size_t count, y3_stream_t stream); /* in place, asynchronous on stream */
int y3_comm_info(void* comm, int* nranks, int* version);
int y3_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* YOLO3HIP_H */
