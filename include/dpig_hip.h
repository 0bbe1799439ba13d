/*
 * dpig_hip.h -- C ABI of libdpig_hip.so, the MI355X (gfx950 / CDNA4) compute library behind the
 * conv hot path of Disentangled-Person-Image-Generation (DPIG).
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes + a HIP stream (passed as
 * `void*`, i.e. a `hipStream_t`), never allocates, never synchronises, and returns an int status:
 *   0 = DPIG_OK, negative = error (see dpig_last_error()).  All tensors are fp32 in memory, activations are
 *   *physically* NHWC with an explicit channel stride (`ld*`, in elements) so that channel-concats
 *   and slices are free views; filters are HWIO exactly as the reference stores them.
 *
 * Which reference interface each entry point replaces (paths relative to the reference repo):
 *   dpig_conv2d_fwd      tf.nn.conv2d(SAME)+bias_add   tflib/ops/conv2d.py:106-120, and every
 *                        slim.conv2d(...) of models.py:396-573 (bias + activation fused)
 *   dpig_conv2d_dgrad    gradient of the above w.r.t. the input (TF autodiff; Optimizer.minimize
 *                        trainer.py:137-140) and tf.nn.conv2d_transpose  tflib/ops/deconv2d.py:97-103
 *   dpig_conv2d_wgrad    gradient w.r.t. the HWIO filter (TF autodiff, same call sites)
 *   dpig_bn_*            tf.nn.fused_batch_norm(training)  tflib/ops/batchnorm.py:30 (+ its gradient)
 *   dpig_ln_*            tf.nn.moments + batch_normalization  tflib/ops/layernorm.py:7-19 (+ grads)
 *   dpig_linear_*        tf.matmul + bias_add  tflib/ops/linear.py:132-146, slim.fully_connected
 *                        models.py:431,464,545,554 (+ grads)
 *   dpig_crop_resize_*   tf.image.crop_and_resize  models.py:415 (+ gradient w.r.t. the image)
 *   dpig_upsample2x_*    tf.image.resize_nearest_neighbor  utils.py:61-72 (+ gradient)
 *   dpig_act_bwd/bias    the ReLU / LeakyReLU / bias_add gradients TF autodiff emits
 *   dpig_adam_step       tf.train.AdamOptimizer  trainer.py:131-146
 *   dpig_sce_* / l1      sigmoid_cross_entropy_with_logits / L1  trainer.py:238-245, 607, 623
 *   dpig_rmsprop_step / dpig_clip   tf.train.RMSPropOptimizer + clip_by_value (wgan mode)  trainer.py:119-128
 *   dpig_gp_*            the WGAN-GP penalty term around tf.gradients  trainer.py:222-236
 *   dpig_bn_sqdev/apply/bwd_sums/bwd_apply   the same batch norm in stages, for statistics over data-parallel ranks
 *   dpig_pose_*          coord2channel_simple_rcv + tf_poseInflate of the input pipeline  utils.py:237-318
 *   dpig_ssim_gray_u8    skimage rgb2gray + compare_ssim of trainer.generate() / score.py  trainer.py:516-521
 * DpigConvDesc.compute selects the matrix-pipe arithmetic of the three conv entry points (fp32, or bf16 operands with
 * fp32 accumulation on unchanged fp32 tensors).
 */
#ifndef DPIG_HIP_H
#define DPIG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPIG_OK 0
#define DPIG_EINVAL (-22)     /* bad descriptor / unsupported shape */
#define DPIG_EALIGN (-14)     /* misaligned pointer or stride */
#define DPIG_ENOMEM (-12)     /* workspace too small */
#define DPIG_ELAUNCH (-5)     /* HIP launch failure */

#define DPIG_ACT_NONE 0
#define DPIG_ACT_RELU 1
#define DPIG_ACT_LRELU 2

#define DPIG_VERSION 283

/* Convolution problem, always described from the FORWARD op's point of view. */
typedef struct DpigConvDesc {
    int32_t N, H, W, C;   /* forward input x: N images of H x W x C                               */
    int32_t K;            /* forward output channels                                              */
    int32_t R, S;         /* filter height / width (HWIO filter [R][S][C][K])                     */
    int32_t stride;       /* 1 or 2 (both spatial dims)                                           */
    int32_t pad_t, pad_l; /* leading pads; pass -1 for TensorFlow 'SAME' (tflib conv2d.py:110)    */
    int32_t ldx;          /* channel stride (elements) of x / dx buffers, >= C                    */
    int32_t ldy;          /* channel stride of y / dy buffers, >= K                               */
    int32_t ldres;        /* channel stride of the residual (fwd) / accumulate (dgrad) tensor     */
    int32_t ldmask;       /* channel stride of the dgrad activation-mask tensor                   */
    int32_t act;          /* fwd: activation after bias(+residual); dgrad: activation whose       */
                          /*      derivative (taken at `mask`) multiplies dx                      */
    float alpha;          /* LeakyReLU slope                                                      */
    int32_t upsample2x;   /* 1: x is nearest-neighbour 2x-upsampled before a 1x1 conv             */
                          /*    (models.py:569-570); computed at low resolution (exact commute)   */
    int32_t res_after_act;/* fwd: 0: y = act(conv+bias+res); 1: y = act(conv+bias) + res, the     */
                          /*      reference res-block order (models.py:398-400,425-427,534-536)   */
    int32_t ldy2;         /* channel stride of the optional y_act output                          */
    int32_t res_class;    /* fwd: 1: `residual` is [N][9][K], indexed by the 3x3 border class of  */
                          /*      the output pixel (top|mid|bottom x left|mid|right): the exact   */
                          /*      contribution of spatially constant input channels to a SAME 3x3 */
                          /*      conv -- the tiled embedding of trainer.py:588-590   */
    int32_t split_k;      /* 0 = library heuristic, otherwise forced split count                  */
    int32_t compute;      /* DPIG_COMPUTE_F32 (default): fp32 MFMA, exact fp32 products.           */
                          /* DPIG_COMPUTE_BF16: tensors stay fp32 in memory, GEMM operands are     */
                          /* rounded to bfloat16 (round-to-nearest-even) and multiplied on the     */
                          /* bf16 matrix pipe with fp32 accumulation (BASELINE configs 3-5);       */
                          /* layers the bf16 loop is not written for (unaligned / fewer than 33    */
                          /* output columns / C % 4 != 0) silently use the fp32 pipe.              */
} DpigConvDesc;
#define DPIG_COMPUTE_F32 0
#define DPIG_COMPUTE_BF16 1
#define DPIG_COMPUTE_BF16X3 2   /* tensors stay fp32; every operand is split into two bf16 terms (hi + lo, 16     */
                                /* significand bits) on its way into LDS and each product block is three bf16     */
                                /* MFMAs (hi*lo + lo*hi + hi*hi) into the fp32 accumulator: the fp32 accuracy     */
                                /* class (<= 2e-5 max|ref|, the exact path's own test bar) at bf16-pipe speed.    */
                                /* Applies where DPIG_COMPUTE_BF16 does; other shapes run the exact fp32 kernels. */
#define DPIG_COMPUTE_BF16_STORE 3 /* DpigCriticDesc.compute only (dpig_gp_double_backward): the critic's activations are stored */
                                /* as bfloat16 inside the workspace and every op runs on the bf16-STORAGE kernels below         */
                                /* (dpig_conv2d_*_bf16, dpig_ln_*_bf16, ...): BASELINE configs[2]-[4]'s mode.                     */

int dpig_version(void);
const char* dpig_last_error(void);

/* TF 'SAME' geometry: out = ceil(in/stride), pad_before = floor(max((out-1)*stride+k-in,0)/2). */
int dpig_same_pad(int in, int k, int stride, int* out, int* pad_before);

/* Scratch requirement (bytes) of the three conv entry points for this descriptor.
 * which: 0 fwd, 1 dgrad, 2 wgrad.  Returns 0 when no workspace is needed. */
size_t dpig_conv2d_workspace_bytes(const DpigConvDesc* d, int which);

/* Replaces tf.nn.conv2d('SAME') + bias_add (+ ReLU) of tflib/ops/conv2d.py:106-120 and slim.conv2d, models.py:396-573.
 * y[N,Ho,Wo,K] = act(conv(x, w) + bias + residual)  (or act(..)+residual, see res_after_act).
 * bias / residual / y_act may be NULL.  y_act (optional) receives the activation output before a
 * post-activation residual add (its sign is the ReLU mask the backward pass needs).
 * With upsample2x, y has spatial size (2H, 2W). */
int dpig_conv2d_fwd(const DpigConvDesc* d, const float* x, const float* w, const float* bias,
                    const float* residual, float* y, float* y_act, void* ws, size_t ws_bytes,
                    void* stream);

/* Replaces Conv2DBackpropInput: the input gradient TF autodiff emits for the convs above (Optimizer.minimize,
 * trainer.py:137-140) and the forward of tf.nn.conv2d_transpose, tflib/ops/deconv2d.py:97-103.
 * dx[N,H,W,C] = (conv_backward_data(dy, w) + accum) * act'(mask).  accum / mask may be NULL. */
int dpig_conv2d_dgrad(const DpigConvDesc* d, const float* dy, const float* w, const float* accum,
                      const float* mask, float* dx, void* ws, size_t ws_bytes, void* stream);

/* Replaces Conv2DBackpropFilter + BiasAddGrad of the same call sites (trainer.py:137-140).
 * dw[R,S,C,K] = beta*dw + conv_backward_filter(x, dy).  beta = 0 overwrites, beta = 1 accumulates.
 * If db is not NULL the same launch also produces the bias gradient db[K] = beta_b*db + sum_pixels dy
 * (TF's BiasAddGrad) from the dy tiles it stages anyway -- no extra pass over dy. */
int dpig_conv2d_wgrad(const DpigConvDesc* d, const float* x, const float* dy, float* dw, float beta,
                      float* db, float beta_b, void* ws, size_t ws_bytes, void* stream);

/* ---- bf16-STORAGE convolution family (BASELINE configs 3-5; csrc/dpig_conv_bf16.hip) ---------------------------
 * Same three operations as dpig_conv2d_{fwd,dgrad,wgrad} (same reference call sites: tflib/ops/conv2d.py:106-120,
 * slim.conv2d of models.py:396-573 and the gradients TF autodiff derives, trainer.py:137-140) on tensors stored as
 * bfloat16 (uint16_t bit patterns, round-to-nearest-even): activations and activation gradients NHWC bf16 with a
 * channel stride; products on the bf16 matrix pipe; accumulation, bias, residual adds, activation and every epilogue
 * in fp32; the filter gradient is written in fp32 (the master weights and the optimizer stay fp32).
 * Filters are read from bf16 SHADOWS of the fp32 HWIO master (dpig_filter_shadow_bf16):
 *   forward  w_t   [R][S][K][C]  (per tap transposed: the GEMM's reduction index contiguous)
 *   dgrad    w     [R][S][C][K]  (the HWIO layout itself)
 * Accepted shapes: C, K >= 32 and multiples of 8, ldx / ldy / ldres / ldmask / ldy2 multiples of 8, every pointer
 * 16-byte aligned (dpig_conv2d_bf16_supported); the few thin layers outside that run on the fp32 entry points
 * between dpig_cvt_* calls.  `compute` of the descriptor is ignored. */
int dpig_conv2d_bf16_supported(const DpigConvDesc* d, int which);
size_t dpig_conv2d_bf16_workspace_bytes(const DpigConvDesc* d, int which);
/* residual (bf16, y-shaped) and residual_class (fp32 [N][9][K], see DpigConvDesc.res_class) are exclusive.  With a
 * post-activation residual and y_act, y = bf16(float(y_act) + residual): the sum is formed from the stored activation. */
int dpig_conv2d_fwd_bf16(const DpigConvDesc* d, const uint16_t* x, const uint16_t* w_t, const float* bias,
                         const uint16_t* residual, const float* residual_class, uint16_t* y, uint16_t* y_act,
                         void* ws, size_t ws_bytes, void* stream);
int dpig_conv2d_dgrad_bf16(const DpigConvDesc* d, const uint16_t* dy, const uint16_t* w, const uint16_t* accum,
                           const uint16_t* mask, uint16_t* dx, void* ws, size_t ws_bytes, void* stream);
/* dw (fp32, [R][S][C][K]) = beta*dw + conv_backward_filter(x, dy); db (fp32, optional) = beta_b*db + sum_pixels dy,
 * accumulated in fp32 on the matrix pipe from the dy tiles the launch stages anyway. */
int dpig_conv2d_wgrad_bf16(const DpigConvDesc* d, const uint16_t* x, const uint16_t* dy, float* dw, float beta,
                           float* db, float beta_b, void* ws, size_t ws_bytes, void* stream);
/* Tile family of dpig_conv2d_fwd_bf16 / _dgrad_bf16 (csrc/dpig_conv_bf16_q.hip): large layers run on 8-wave workgroups
 * with 256 x 256 or 512 x 128 block tiles (one workgroup per CU, counted-vmcnt LDS-DMA pipeline), the rest on the
 * 128 x 128 kernels.  mode 0 = 128 x 128 only, 1 = automatic (default; environment DPIG_BF16_Q), 2 = large tiles whenever
 * the layer is legal for them (channel count a multiple of 64; tests).  variant 0 = automatic, 1 = 256 x 256, 2 = 512 x 128.
 * Results do not depend on the choice beyond fp32 summation order.  Process-wide; not stream-ordered. */
int dpig_conv_bf16_set_large_tile(int mode, int variant);
/* The same switch for dpig_conv2d_wgrad_bf16 (csrc/dpig_conv_bf16_wq.hip: stride-1 SAME layers; variant 1 = 2 (tap, 128-ci)
 * items x 256 co per workgroup, 2 = 4 items x 128 co; environment DPIG_BF16_WQ).  The workspace query follows the setting. */
int dpig_conv_bf16_set_large_tile_wgrad(int mode, int variant);
/* The 128 x 128 kernels with eight waves per workgroup instead of four (csrc/dpig_conv_bf16.hip: 64 x 32 sub-tiles, half the LDS-DMA
 * instructions per wave): bit 0 = forward / dgrad (bg8_kernel, bg8_multi_kernel), bit 1 = filter gradient (bw8_kernel), bit 2 = the
 * halo-patch kernel (bh8_kernel); environment DPIG_BF16_G8.  Same plans, same bits. */
int dpig_conv_bf16_set_wave8(int mode);
/* ---- Winograd F(2x2, 3x3) on the fp32 matrix pipe (csrc/dpig_conv_wino.hip) -------------------------------------------------
 * The 3 x 3 stride-1 SAME convs of slim.conv2d (reference models.py:396-400, 425-427, 458-460, 534-535, 564-565: 89 % of a stage-I
 * step's FLOPs) with 2.25x fewer multiplies: fp32 tensors, fp32 products, fp32 accumulation -- the arithmetic TYPE of
 * dpig_conv2d_fwd / _dgrad, a different (mathematically equivalent) evaluation order, so results agree with the direct kernels to a few
 * fp32 ulps of the largest intermediate instead of bit for bit (tests/test_wino_gpu.py holds both to the same bar against the fp64
 * oracle).  Input transform, 16 position GEMMs, output transform and the fused epilogue are ONE kernel; the filter's transform is made
 * once per optimizer step:
 *   dpig_wino_filter_elems      floats of one transformed image of a [3][3][C][K] filter (16 C K), 0 if C or K is not a multiple of 64
 *   dpig_wino_filter_transform  u_fwd / u_dgrad (either may be NULL) <- HWIO filter w; u_dgrad is the image of the 180-degree rotated,
 *                               channel-transposed filter that makes conv_backward_data the same kernel
 *   dpig_conv2d_wino_eligible   1 if the descriptor (which = 0 forward, 1 dgrad) has a Winograd form (3x3, stride 1, even H and W,
 *                               C and K multiples of 64, 16-byte channel vectors, no class residual) AND the cost model expects it to
 *                               beat the direct kernel (small maps that cannot fill 256 CUs with 64-tile x 64-channel workgroups
 *                               stay direct); dpig_conv_wino_set_mode / DPIG_WINO: 0 never, 1 cost model (default), 2 wherever legal
 *   dpig_conv2d_fwd_wino        dpig_conv2d_fwd's semantics (bias, activation, residual before / after it, second output y_act)
 *   dpig_conv2d_dgrad_wino      dpig_conv2d_dgrad's semantics ((. + accum) * act'(mask))
 * Workspace (dpig_conv2d_wino_workspace_bytes; 0 for layers that fill the chip): layers with fewer workgroups than CUs, or a fractional
 * last round, cut the reduction over input channels into ranges whose partial outputs a second kernel sums in fixed order. */
size_t dpig_wino_filter_elems(int C, int K);
int dpig_wino_filter_transform(const float* w, int C, int K, float* u_fwd, float* u_dgrad, void* stream);
/* A whole parameter set's images in ONE launch (after every optimizer step): describe each filter in a job, let
 * dpig_wino_filter_jobs_plan (host only) fill first_block and return the block total (0: a job has no Winograd form), copy the array to
 * device memory once, then dpig_wino_filter_transform_jobs(jobs_dev, njobs, total, stream) per refresh.  u_fwd / u_dgrad may be null. */
typedef struct DpigWinoFilterJob {
    const float* w;            /* HWIO [3][3][C][K] master */
    float* u_fwd;              /* dpig_wino_filter_elems(C, K) floats each */
    float* u_dgrad;
    int32_t C, K;
    int32_t first_block, reserved;
} DpigWinoFilterJob;
int dpig_wino_filter_jobs_plan(DpigWinoFilterJob* jobs, int njobs);
int dpig_wino_filter_transform_jobs(const DpigWinoFilterJob* jobs_dev, int njobs, int total_blocks, void* stream);
int dpig_conv2d_wino_eligible(const DpigConvDesc* d, int which);
int dpig_conv_wino_set_mode(int mode);
int dpig_conv_wino_get_mode(void);        /* the mode in force: DPIG_WINO from the environment (default 1) until _set_mode changes it */
size_t dpig_conv2d_wino_workspace_bytes(const DpigConvDesc* d, int which);
int dpig_conv2d_fwd_wino(const DpigConvDesc* d, const float* x, const float* u_fwd, const float* bias, const float* residual,
                         float* y, float* y_act, void* ws, size_t ws_bytes, void* stream);
int dpig_conv2d_dgrad_wino(const DpigConvDesc* d, const float* dy, const float* u_dgrad, const float* accum, const float* mask,
                           float* dx, void* ws, size_t ws_bytes, void* stream);
/* The filter gradient by F(3x3, 2x2) minimal filtering: dw[3][3][C][K] = beta dw + conv_backward_filter(x, dy), 16 multiplies per
 * (2x2 tile, c, k) instead of 36; input and output-gradient transforms, the 16 position GEMMs over the tile axis and the output
 * transform are one kernel, split over tile ranges into S partial gradients that a second kernel sums in fixed order
 * (deterministic).  Same shape rules as above.  With db ([K]; may be null) the same launches leave the bias gradient
 * db = beta_b db + sum over pixels of dy, as dpig_conv2d_wgrad does.  Workspace: dpig_conv2d_wgrad_wino_workspace_bytes.  dpig_conv2d_wgrad_wino_eligible: shape has the form AND the cost model prefers it. */
int dpig_conv2d_wgrad_wino_eligible(const DpigConvDesc* d);
size_t dpig_conv2d_wgrad_wino_workspace_bytes(const DpigConvDesc* d);
int dpig_conv2d_wgrad_wino(const DpigConvDesc* d, const float* x, const float* dy, float* dw, float beta, float* db, float beta_b,
                           void* ws, size_t ws_bytes, void* stream);

/* ---- Winograd F(4x4, 3x3) on the fp32 matrix pipe (csrc/dpig_conv_wino4.hip) ------------------------------------------------
 * The same layers (reference models.py:396-400, 425-427, 458-460, 534-535, 564-565) on maps whose sides are multiples of 4, with 4x fewer
 * multiplies than direct summation (36 per 4 x 4 output tile and channel pair instead of 144): fp32 tensors, products and sums; the
 * transforms' larger constants cost about one decimal digit against F(2x2, 3x3) (a few 1e-6 of the largest activation on the golden
 * model, profiles/r06_f43_numerics.txt; tests/test_wino4_gpu.py holds the kernels to 5e-5 against the fp64 oracle).  Entry points
 * mirror the F(2x2, 3x3) family's, with their own images (36 C K floats each, in MFMA-fragment order):
 *   dpig_wino4_filter_elems / dpig_wino4_filter_transform / dpig_wino4_filter_transform_jobs (the job table and plan of
 *   dpig_wino_filter_jobs_plan serve both families: the block numbering is the same)
 *   dpig_conv2d_wino4_eligible   the descriptor has the form (3x3, stride 1, H and W multiples of 4, the tile grid cut into blocks of
 *                                4 x 8, 2 x 16 or 3 x 10 tiles on the stack of all images' tile rows -- W / 4 even or a multiple of 3;
 *                                the last block may be partial --, C and K multiples of 64) AND the cost
 *                                model expects it to beat the F(2x2, 3x3) kernel; dpig_conv_wino4_set_mode / DPIG_WINO4: 0 never,
 *                                1 cost model (default), 2 wherever legal
 *   dpig_conv2d_fwd_wino4 / dpig_conv2d_dgrad_wino4 / dpig_conv2d_wino4_workspace_bytes   as their F(2x2, 3x3) namesakes */
size_t dpig_wino4_filter_elems(int C, int K);
int dpig_wino4_filter_transform(const float* w, int C, int K, float* u_fwd, float* u_dgrad, void* stream);
int dpig_wino4_filter_transform_jobs(const DpigWinoFilterJob* jobs_dev, int njobs, int total_blocks, void* stream);
int dpig_conv2d_wino4_eligible(const DpigConvDesc* d, int which);
int dpig_conv_wino4_set_mode(int mode);
int dpig_conv_wino4_get_mode(void);
size_t dpig_conv2d_wino4_workspace_bytes(const DpigConvDesc* d, int which);
int dpig_conv2d_fwd_wino4(const DpigConvDesc* d, const float* x, const float* u_fwd, const float* bias, const float* residual,
                          float* y, float* y_act, void* ws, size_t ws_bytes, void* stream);
int dpig_conv2d_dgrad_wino4(const DpigConvDesc* d, const float* dy, const float* u_dgrad, const float* accum, const float* mask,
                            float* dx, void* ws, size_t ws_bytes, void* stream);

/* The thin layers of 'bf16' mode on the vector-ALU kernels (csrc/dpig_thin.hip) with their WIDE tensor stored as bf16;
 * the 3-channel image side, the fp32 HWIO filter and the filter gradient stay fp32:
 *   K == 3 (3x3 s1, the generator's image conv models.py:573):        x / dx bf16 [.., C],   y / dy fp32 [.., 3]
 *   C == 3 (3x3 s1 encoder stem models.py:396, 5x5 s2 critic conv1):  x / dx fp32 [.., 3],   y / dy bf16 [.., K]
 * (dgrad towards the image exists for the 5x5 s2 layer only).  DPIG_EINVAL for any other layer.  wgrad workspace:
 * dpig_conv2d_workspace_bytes(d, 2). */
int dpig_conv2d_fwd_thin_bf16(const DpigConvDesc* d, const void* x, const float* w, const float* bias, void* y,
                              void* stream);
int dpig_conv2d_dgrad_thin_bf16(const DpigConvDesc* d, const void* dy, const float* w, void* dx, void* stream);
int dpig_conv2d_wgrad_thin_bf16(const DpigConvDesc* d, const void* x, const void* dy, float* dw, float beta, float* db,
                                float beta_b, void* ws, size_t ws_bytes, void* stream);
/* [rows, cols] matrices with row strides (elements): fp32 <-> bf16 (round-to-nearest-even / exact widening). */
int dpig_cvt_f32_to_bf16(const float* in, int ldi, uint16_t* out, int ldo, int64_t rows, int cols, void* stream);
int dpig_cvt_bf16_to_f32(const uint16_t* in, int ldi, float* out, int ldo, int64_t rows, int cols, void* stream);
/* out[r][0..cols_out) = bf16(in[r][0..cols_in)), zero-padded: widens a thin input (the 18 pose channels of
 * models.py:520-528 -> 32) so that its conv runs on the bf16 matrix-pipe loop with a zero-padded filter. */
int dpig_cvt_f32_to_bf16_pad(const float* in, int ldi, int cols_in, uint16_t* out, int ldo, int cols_out, int64_t rows,
                             void* stream);
/* y = act(x) / dz = dy * act'(y) on bf16 [rows, cols] matrices (cols and strides multiples of 8): dpig_act_fwd / _bwd. */
int dpig_act_fwd_bf16(const uint16_t* x, int ldx, uint16_t* y, int ldy, int64_t rows, int cols, int act, float alpha,
                      void* stream);
int dpig_act_bwd_bf16(const uint16_t* dy, int lddy, const uint16_t* y, int ldy, uint16_t* dz, int lddz, int64_t rows,
                      int cols, int act, float alpha, void* stream);
/* bf16 shadows of an fp32 HWIO filter w[taps][C][K]: plain [taps][C][K] and / or transposed [taps][K][C]
 * (either output may be NULL); refreshed after every optimizer step. */
int dpig_filter_shadow_bf16(const float* w, uint16_t* plain, uint16_t* transposed, int taps, int C, int K,
                            void* stream);
/* The shadows of MANY filters in one launch.  table_dev: device array of ntensors rows of six int64
 * {src offset in floats from `base`, destination offset in elements from plain_base / trans_base, taps, C, K, first
 * tile}, rows ordered by first tile; tiles are 32 x 32 (per filter taps * ceil(C/32) * ceil(K/32)), total_tiles their sum. */
int dpig_filter_shadow_bf16_multi(const float* base, uint16_t* plain_base, uint16_t* trans_base,
                                  const int64_t* table_dev, int ntensors, int total_tiles, void* stream);

/* ---- DPIG_COMPUTE_BF16X3 with the filter's split taken out of the k-loop ------------------------------------------------
 * dpig_filter_shadow_split*: from the fp32 HWIO master, hi = bf16(w) and lo = bf16(w - hi) in both layouts (plain [R,S,C,K],
 * per-tap transposed [R,S,K,C]); the lo plane of a layout sits lo_off ELEMENTS behind its hi plane (same allocation).  Call
 * after every optimizer step (the _multi form: all filters of a flat parameter buffer in one launch, table rows as for
 * dpig_filter_shadow_bf16_multi).  dpig_conv2d_fwd_x3 / _dgrad_x3 = dpig_conv2d_fwd / _dgrad with compute = BF16X3, reading
 * the filter's two terms from those planes by LDS-DMA where the layer allows (else from `w`): same results bit for bit. */
int dpig_filter_shadow_split(const float* w, uint16_t* plain_hi, uint16_t* trans_hi, int64_t lo_off, int taps, int C, int K,
                             void* stream);
int dpig_filter_shadow_split_multi(const float* base, uint16_t* plain_base, uint16_t* trans_base, int64_t lo_off,
                                   const int64_t* table_dev, int ntensors, int total_tiles, void* stream);
int dpig_conv2d_fwd_x3(const DpigConvDesc* d, const float* x, const uint16_t* x32, const float* w, const uint16_t* w_t_hi,
                       const uint16_t* w_t_lo, const float* bias, const float* residual, float* y, float* y_act,
                       uint16_t* y32, int* y32_written, void* ws, size_t ws_bytes, void* stream);
int dpig_conv2d_dgrad_x3(const DpigConvDesc* d, const float* dy, const uint16_t* dy32, const float* w, const uint16_t* w_hi,
                         const uint16_t* w_lo, const float* accum, const float* mask, float* dx, uint16_t* dx32,
                         int* dx32_written, void* ws, size_t ws_bytes, void* stream);
/* x32 / dy32 (optional, may be null): the gathered activation's own two-term split, written once per tensor by dpig_split32 as
 * [pixel][32-channel chunk][32 hi | 32 lo] bf16 (dpig_split32_bytes(rows, C) bytes, rows = N*H*W of that tensor, contiguous).
 * With it BOTH operands reach LDS by DMA and no operand passes through registers; results stay bit-identical.
 * y32 / dx32 (optional): room for the OUTPUT's split32 image; the float4 epilogue of an un-split launch writes it beside the
 * fp32 output (no second pass over it) and sets *written = 1 -- the next conv of the chain takes it as its x32 / dy32;
 * *written = 0 (split-K plan, thin layer, channel count not a multiple of 32, stride-2 dgrad): make it with dpig_split32. */
/* Filter gradient of the same mode from the two images (x32 of the forward input, dy32 of the output gradient): both operands
 * by LDS-DMA, no pass over the fp32 tensors; dw bit-identical to dpig_conv2d_wgrad's, db summed from dy's 16 split bits.
 * Null images / layers it does not serve run dpig_conv2d_wgrad(d, x, dy, ...). */
int dpig_conv2d_wgrad_x3(const DpigConvDesc* d, const float* x, const uint16_t* x32, const float* dy, const uint16_t* dy32,
                         float* dw, float beta, float* db, float beta_b, void* ws, size_t ws_bytes, void* stream);
/* dpig_act_bwd that also leaves dz's split32 image (cols % 32 == 0, 16-byte addressable rows). */
int dpig_act_bwd_s32(const float* dy, int lddy, const float* y, int ldy, float* dz, int lddz, int64_t rows, int cols, int act,
                     float alpha, uint16_t* dz32, void* stream);
size_t dpig_split32_bytes(int64_t rows, int C);
int dpig_split32(const float* x, int ldx, int64_t rows, int C, uint16_t* out, void* stream);

/* ---- elementwise / reductions over a [rows, cols] fp32 matrix with row stride ld ------------- */

/* y = act(x): stand-alone ReLU / LeakyReLU (wgan_gp.py:23-24 `tf.maximum(alpha*x, x)`). */
int dpig_act_fwd(const float* x, int ldx, float* y, int ldy, int64_t rows, int cols, int act, float alpha,
                 void* stream);
/* dz = dy * act'(y)   (y = the activation OUTPUT; relu'/lrelu' decided by y > 0): ReluGrad / the gradient of
 * wgan_gp.py:23-24's maximum, as TF autodiff emits them. */
int dpig_act_bwd(const float* dy, int lddy, const float* y, int ldy, float* dz, int lddz, int64_t rows,
                 int cols, int act, float alpha, void* stream);
/* out[c] = beta*out[c] + sum_r a[r,c]   (BiasAddGrad of conv2d.py:118-120 / linear.py:142-146).  ws: dpig_colsum_workspace_bytes. */
size_t dpig_colsum_workspace_bytes(int64_t rows, int cols);
int dpig_colsum(const float* a, int lda, int64_t rows, int cols, float* out, float beta, void* ws,
                size_t ws_bytes, void* stream);

/* out[N][9][C] = per-image sums of a[N,H,W,C] over the 9 border classes (gradient of the
 * class-indexed residual above; the backward half of the tiled-embedding collapse, trainer.py:588-590 + models.py:520-528). */
size_t dpig_border_class_sum_workspace_bytes(int N, int H, int W, int C);
int dpig_border_class_sum(const float* a, int lda, int N, int H, int W, int C, float* out, void* ws,
                          size_t ws_bytes, void* stream);

/* The same sums of a bf16 tensor ('bf16' storage mode), accumulated and returned in fp32. */
int dpig_border_class_sum_bf16(const uint16_t* a, int lda, int N, int H, int W, int C, float* out, void* ws,
                               size_t ws_bytes, void* stream);

/* Batch-norm statistics carried by the producing conv (conv2d.py:106-120 followed by batchnorm.py:30): the forward conv
 * (+ bias, no activation) also writes, per row tile of 128 output pixels, the column sums and the sums of squared deviations
 * from the tile's mean; dpig_bn_stats_finalize merges them (fixed order, exact pairwise update) into the batch mean and
 * rstd = 1/sqrt(biased variance + eps), after which dpig_bn_apply normalises.  One pass over the conv output instead of
 * three.  dpig_conv2d_bn_stats_tiles(d) = number of row tiles = leading dimension of `stats` ([tiles][2][K] floats), or 0
 * when this problem's plan cannot carry the statistics (split-K, K <= 32, upsample fusion, a batch served in several
 * launches): callers then use dpig_bn_fwd. */
int dpig_conv2d_bn_stats_tiles(const DpigConvDesc* d);
int dpig_conv2d_fwd_stats(const DpigConvDesc* d, const float* x, const float* w, const float* bias, float* y, float* stats,
                          void* stream);
/* ... with a workspace (dpig_conv2d_workspace_bytes(d, 0)): split-K plans carry the statistics too -- the reduction pass that sums the
 * partials leaves the tile statistics (same layout, same values as an un-split launch of the same tile would leave). */
int dpig_conv2d_bn_stats_tiles_ws(const DpigConvDesc* d);
int dpig_conv2d_fwd_stats_ws(const DpigConvDesc* d, const float* x, const float* w, const float* bias, float* y, float* stats,
                             void* ws, size_t ws_bytes, void* stream);
/* the same for bf16 activations (statistics from the fp32 accumulators + bias, before y is rounded to bf16) */
int dpig_conv2d_bf16_bn_stats_tiles(const DpigConvDesc* d);
int dpig_conv2d_fwd_bf16_stats(const DpigConvDesc* d, const uint16_t* x, const uint16_t* w_t, const float* bias, uint16_t* y,
                               float* stats, void* stream);
int dpig_bn_stats_finalize(const float* stats, int tiles, int64_t rows, int rows_per_tile, int C, float eps, float* mean,
                           float* rstd, void* stream);

/* ---- batch norm (training mode, biased variance, eps inside sqrt; batchnorm.py:30) ----------- */
/* x,y: [rows, C] (rows = N*H*W).  save_mean / save_rstd: [C].  Optional fused LeakyReLU/ReLU. */
size_t dpig_bn_workspace_bytes(int64_t rows, int C);
int dpig_bn_fwd(const float* x, int ldx, int64_t rows, int C, const float* scale, const float* offset,
                float eps, int act, float alpha, float* y, int ldy, float* save_mean, float* save_rstd,
                void* ws, size_t ws_bytes, void* stream);
/* dy is the gradient w.r.t. the (activated) output y; y is needed only when act != NONE. */
int dpig_bn_bwd(const float* dy, int lddy, const float* x, int ldx, const float* y, int ldy, int64_t rows,
                int C, const float* scale, const float* save_mean, const float* save_rstd, int act,
                float alpha, float* dx, int lddx, float* dscale, float* doffset, void* ws,
                size_t ws_bytes, void* stream);

/* The same op on bf16 tensors ('bf16' storage mode): x / y / dy / dx bfloat16 [rows, C] with row strides (C and strides multiples of
 * 8, 16-byte aligned), fp32 arithmetic, statistics and parameter gradients.  dpig_bn_apply_bf16 normalises with given statistics
 * (those a conv epilogue left: dpig_conv2d_fwd_bf16_stats + dpig_bn_stats_finalize). */
size_t dpig_bn_bf16_workspace_bytes(int64_t rows, int C);
int dpig_bn_fwd_bf16(const uint16_t* x, int ldx, int64_t rows, int C, const float* scale, const float* offset, float eps, int act,
                     float alpha, uint16_t* y, int ldy, float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* stream);
int dpig_bn_apply_bf16(const uint16_t* x, int ldx, int64_t rows, int C, const float* scale, const float* offset, const float* mean,
                       const float* rstd, int act, float alpha, uint16_t* y, int ldy, void* stream);
int dpig_bn_bwd_bf16(const uint16_t* dy, int lddy, const uint16_t* x, int ldx, const uint16_t* y, int ldy, int64_t rows, int C,
                     const float* scale, const float* save_mean, const float* save_rstd, int act, float alpha, uint16_t* dx, int lddx,
                     float* dscale, float* doffset, void* ws, size_t ws_bytes, void* stream);

/* Staged form of the same op for synchronised batch statistics across data-parallel ranks: the
 * caller sum-all-reduces the [C] vectors between the stages.
 *   fwd: dpig_colsum(x) -> mean = sum/n_total; dpig_bn_sqdev(x, mean) -> rstd = rsqrt(sq/n_total + eps);
 *        dpig_bn_apply.
 *   bwd: dpig_bn_bwd_sums -> (dscale, doffset) summed over ranks; dpig_bn_bwd_apply with inv_count = 1/n_total. */
int dpig_bn_sqdev(const float* x, int ldx, int64_t rows, int C, const float* mean, float* sq_out, void* ws,
                  size_t ws_bytes, void* stream);
int dpig_bn_apply(const float* x, int ldx, int64_t rows, int C, const float* scale, const float* offset,
                  const float* mean, const float* rstd, int act, float alpha, float* y, int ldy, void* stream);
int dpig_bn_bwd_sums(const float* dy, int lddy, const float* x, int ldx, const float* y, int ldy, int64_t rows,
                     int C, const float* mean, const float* rstd, int act, float alpha, float* dscale,
                     float* doffset, void* ws, size_t ws_bytes, void* stream);
int dpig_bn_bwd_apply(const float* dy, int lddy, const float* x, int ldx, const float* y, int ldy, int64_t rows,
                      int C, const float* scale, const float* mean, const float* rstd, const float* dscale,
                      const float* doffset, int act, float alpha, float inv_count, float* dx, int lddx,
                      void* stream);

/* ---- layer norm over (H,W,C) per sample, per-channel scale/offset (layernorm.py:6-20) -------- */
/* x,y: [N, P, C] with P = H*W pixels, dense (ld == C). save_mean/save_rstd: [N].  A sample is cut into chunks, one workgroup
 * each (csrc/dpig_norm.hip): every entry point takes a small workspace for the per-chunk partial sums.  The _bf16 forms read /
 * write bf16 tensors directly ('bf16' storage mode) with the same fp32 arithmetic; parameters, statistics and parameter
 * gradients are fp32 in both. */
size_t dpig_ln_fwd_workspace_bytes(int N, int P, int C);
int dpig_ln_fwd(const float* x, int N, int P, int C, const float* scale, const float* offset, float eps,
                int act, float alpha, float* y, float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* stream);
int dpig_ln_fwd_bf16(const uint16_t* x, int N, int P, int C, const float* scale, const float* offset, float eps,
                     int act, float alpha, uint16_t* y, float* save_mean, float* save_rstd, void* ws, size_t ws_bytes,
                     void* stream);
/* dscale / doffset may both be NULL (input gradient only: the first sweep of the gradient penalty). */
size_t dpig_ln_workspace_bytes(int N, int P, int C);
int dpig_ln_bwd(const float* dy, const float* x, const float* y, int N, int P, int C, const float* scale,
                const float* save_mean, const float* save_rstd, int act, float alpha, float* dx,
                float* dscale, float* doffset, void* ws, size_t ws_bytes, void* stream);
int dpig_ln_bwd_bf16(const uint16_t* dy, const uint16_t* x, const uint16_t* y, int N, int P, int C, const float* scale,
                     const float* save_mean, const float* save_rstd, int act, float alpha, uint16_t* dx,
                     float* dscale, float* doffset, void* ws, size_t ws_bytes, void* stream);

/* Second-order LayerNorm (gradient of dpig_ln_bwd's dx w.r.t. dy, x and scale), the piece of the WGAN-GP
 * double backward (trainer.py:222-236) that does not reduce to the first-order kernels: with
 * u = dPenalty/d(dx) it returns d_dy, d_x [N,P,C] and d_scale [C] (the offset has no second-order term). */
size_t dpig_ln_bwd2_workspace_bytes(int N, int P, int C);
int dpig_ln_bwd2(const float* u, const float* dy, const float* x, const float* y, int N, int P, int C,
                 const float* scale, const float* save_mean, const float* save_rstd, int act, float alpha,
                 float* d_dy, float* d_x, float* d_scale, void* ws, size_t ws_bytes, void* stream);
int dpig_ln_bwd2_bf16(const uint16_t* u, const uint16_t* dy, const uint16_t* x, const uint16_t* y, int N, int P, int C,
                      const float* scale, const float* save_mean, const float* save_rstd, int act, float alpha,
                      uint16_t* d_dy, uint16_t* d_x, float* d_scale, void* ws, size_t ws_bytes, void* stream);

/* ---- fully connected (linear.py:132-146, slim.fully_connected) ------------------------------- */
/* y[M,Nout] = act(x[M,Kin] @ w[Kin,Nout] + bias) */
size_t dpig_linear_workspace_bytes(int M, int Kin, int Nout, int which);
int dpig_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int Kin, int Nout,
                    int act, float alpha, void* ws, size_t ws_bytes, void* stream);
/* dx[M,Kin] = dy[M,Nout] @ w^T */
int dpig_linear_dgrad(const float* dy, const float* w, float* dx, int M, int Kin, int Nout, void* ws,
                      size_t ws_bytes, void* stream);
/* dw[Kin,Nout] = beta*dw + x^T @ dy   (the bias gradient is dpig_colsum(dy)) */
int dpig_linear_wgrad(const float* x, const float* dy, float* dw, float beta, int M, int Kin, int Nout,
                      void* ws, size_t ws_bytes, void* stream);

/* ---- tf.image.crop_and_resize (bilinear, extrapolation 0; models.py:415) ------------------------ */
/* img: [N,H,W,C] (ld = C); boxes: [nbox,4] normalised (y1,x1,y2,x2); box_ind: [nbox];
 * out: [nbox, ch, cw, C]. */
int dpig_crop_resize_fwd(const float* img, int N, int H, int W, int C, const float* boxes,
                         const int32_t* box_ind, int nbox, int ch, int cw, float* out, void* stream);
/* Image gradient (TF CropAndResizeGradImage) as two separable gather passes: deterministic, no atomics;
 * dimg [N,H,W,C] is fully overwritten.  ws: dpig_crop_resize_bwd_workspace_bytes(W, C, nbox, ch) bytes. */
size_t dpig_crop_resize_bwd_workspace_bytes(int W, int C, int nbox, int ch);
int dpig_crop_resize_bwd(const float* dout, int N, int H, int W, int C, const float* boxes,
                         const int32_t* box_ind, int nbox, int ch, int cw, float* dimg, void* ws,
                         size_t ws_bytes, void* stream);

/* The same two ops on bf16 tensors ('bf16' storage mode): bilinear weights and sums in fp32 (the separable backward's
 * intermediate stays fp32), inputs / results bf16. */
int dpig_crop_resize_fwd_bf16(const uint16_t* img, int N, int H, int W, int C, const float* boxes,
                              const int32_t* box_ind, int nbox, int ch, int cw, uint16_t* out, void* stream);
int dpig_crop_resize_bwd_bf16(const uint16_t* dout, int N, int H, int W, int C, const float* boxes,
                              const int32_t* box_ind, int nbox, int ch, int cw, uint16_t* dimg, void* ws,
                              size_t ws_bytes, void* stream);

/* ---- input pipeline: pose target maps (utils.py:237-318) --------------------------------
 * rcv: [B, K, 3] = (row, col, visibility) per keypoint, rows/cols in [-1,1] if is_normalized.  out: [B,H,W,K], ld = ldo.
 * dpig_pose_points   = coord2channel_simple_rcv: 2v-1 at the clipped, truncated keypoint pixel, -1 elsewhere.
 * dpig_pose_inflate  = tf_poseInflate on any [-1,1] map: the map plus its 49 zero-padded shifts (disc of radius 4),
 *                      clipped at 1, back to [-1,1].
 * dpig_pose_rasterize = the two chained in one pass straight from the coordinates (no intermediate map, no 49 shifts). */
int dpig_pose_points(const float* rcv, int B, int K, int H, int W, int is_normalized, float* out, int ldo,
                     void* stream);
int dpig_pose_inflate(const float* pose, int ldp, int B, int K, int H, int W, float* out, int ldo, void* stream);
int dpig_pose_rasterize(const float* rcv, int B, int K, int H, int W, int is_normalized, float* out, int ldo,
                        void* stream);

/* ---- evaluation metric of trainer.generate() / score.py (trainer.py:516-521) -------------------
 * SSIM as skimage.measure.compare_ssim computes it on gray uint8 images: a, b: [B,H,W,3] fp32 pixel values
 * (ld 3); each is clipped to [0,255], truncated to uint8, converted to gray ((0.2125,0.7154,0.0721)/255), 7x7
 * uniform window, sample covariance, data_range = max - min of b's gray image (per image), mean over the windows
 * fully inside the image.  out[B].  H, W >= 7. */
size_t dpig_ssim_workspace_bytes(int B, int H, int W);
int dpig_ssim_gray_u8(const float* a, const float* b, int B, int H, int W, float* out, void* ws, size_t ws_bytes,
                      void* stream);

/* ---- WGAN-GP gradient-penalty term (trainer.py:222-236, wgan_gp.py:605-619) --------------------------------
 * dpig_gp_interpolate: xhat[b,:] = real[b,:] + alpha[b] * (fake[b,:] - real[b,:]),  tensors [B, D] dense.
 * dpig_gp_penalty: given g = grad_xhat D(xhat) [B, D]:  slope_b = ||g[b,:]||_2,
 *     *penalty = lambda * mean_b (slope_b - 1)^2   and   dg[b,:] = lambda * 2 (slope_b - 1) / (B * slope_b) * g[b,:]
 *   (the derivative of the penalty w.r.t. g: the seed of the double-backward sweep through the critic), in one
 *   pass over g (chunked over the chip: per-chunk sums of squares in `ws`, merged in chunk order) + a [B]-sized finalisation;
 *   slopes: [B] output.  A zero slope yields dg = 0. */
int dpig_gp_interpolate(const float* real, const float* fake, const float* alpha, int B, int64_t D, float* xhat,
                        void* stream);
size_t dpig_gp_penalty_workspace_bytes(int B, int64_t D);
int dpig_gp_penalty(const float* g, int B, int64_t D, float lambda, float* penalty, float* dg, float* slopes,
                    void* ws, size_t ws_bytes, void* stream);

/* ---- the whole gradient-penalty term of the DCGAN critic in ONE call (csrc/dpig_gp.hip) ------------------------------------
 * Replaces, for Discriminator = WGAN_GP.DCGANDiscriminator in MODE 'wgan-gp' (wgan_gp.py:407-440: Conv5x5s2 -> LReLU ->
 * [Conv5x5s2 -> LayerNorm -> LReLU] x3 -> reshape [-1, 8*4*8*dim] of the NCHW tensor -> Linear), the graph TF builds for
 *     interpolates = real + alpha*(fake-real); gradients = tf.gradients(D(interpolates), [interpolates])[0]
 *     gradient_penalty = LAMBDA * mean((||gradients||_2 - 1)^2)                               (trainer.py:222-236)
 * AND the part of Optimizer.minimize(disc_cost) that differentiates it w.r.t. the critic's variables (tf.gradients of
 * tf.gradients, i.e. the double backward), written out analytically: sweep 1 (forward + input gradient), dpig_gp_penalty,
 * the adjoint of sweep 1's backward half (forward convs + wgrads + dpig_ln_bwd2) and the ordinary backward pass of the x-adjoints.
 * Interface tensors are fp32: images NHWC [B][H][W][Cin]; filters HWIO [5][5][C][K]; LayerNorm scale/offset [C]; w_out
 * [8*4*8*dim][1]; every gradient.  desc->compute = DPIG_COMPUTE_BF16_STORE keeps the critic's own activations (and their
 * adjoints) as bfloat16 inside the workspace and runs the bf16-storage kernels with fp32 accumulation (Cin = 3, dim in {32, 64,
 * 128, 256}); the other modes keep them fp32.
 *   penalty[0]  = the penalty value;  slopes[B] = ||gradients_b||_2
 *   grads->X    = beta * grads->X + d penalty / d X   for every parameter (the output bias has no penalty gradient and is not
 *                 in the struct); grads == NULL: value only (sweep 1).
 * 256x256 inputs give 8 logit rows per image through the hard-coded reshape, exactly as the reference. */
typedef struct DpigCriticDesc {
    int32_t B, H, W, Cin;   /* interpolated images [B][H][W][Cin]; H, W multiples of 16                         */
    int32_t dim;            /* DIM of wgan_gp.py:407: conv widths dim, 2dim, 4dim, 8dim                            */
    float lrelu_alpha;      /* 0.2 (wgan_gp.py:23)                                                                 */
    float ln_eps;           /* 1e-5 (layernorm.py:17)                                                              */
    float lambda;           /* LAMBDA = 10 (wgan_gp.py:100)                                                        */
    int32_t compute;        /* DPIG_COMPUTE_* of the convolutions; DPIG_COMPUTE_BF16_STORE: bf16 activations       */
} DpigCriticDesc;
typedef struct DpigCriticParams {
    const float* w[4];          /* Discriminator.{1..4}.Filters                                                    */
    const float* b[4];          /* Discriminator.{1..4}.Biases                                                     */
    const float* ln_scale[3];   /* Discriminator.BN{2..4}.scale                                                    */
    const float* ln_offset[3];  /* Discriminator.BN{2..4}.offset                                                   */
    const float* w_out;         /* Discriminator.Output.W                                                          */
} DpigCriticParams;
typedef struct DpigCriticGrads {
    float* w[4];
    float* b[4];
    float* ln_scale[3];
    float* ln_offset[3];
    float* w_out;
} DpigCriticGrads;
size_t dpig_gp_double_backward_workspace_bytes(const DpigCriticDesc* d);
/* Inspection: byte offset / size, inside the workspace of a finished call, of one tensor of the three sweeps.  level 0: the fp32
 * images, slot 0 xhat, 1 g = dD/dxhat, 2 u0 = dpenalty/dg.  level 1..4 (NHWC [B][H_l][W_l][C_l]; fp32, or bfloat16 with
 * DPIG_COMPUTE_BF16_STORE), slot 0 z (conv output; levels 2-4), 1 a (activation), 2 da and 3 dz (input-gradient sweep), 4 v and 5 ub
 * (adjoints of the up-sweep before / after the norm), 6 zb (second-order x-adjoint of the norm), 7 t (what the down-sweep receives
 * from the level above), 8 f (its first-order LayerNorm gradient), 9 zs (zb + f, what flows further down; level 4: use zb).  No slot
 * is written twice, so every link of the chain can be checked against its inputs (tests/test_variants_gpu.py). */
int dpig_gp_double_backward_slot(const DpigCriticDesc* d, int level, int slot, size_t* offset, size_t* bytes);
int dpig_gp_double_backward(const DpigCriticDesc* d, const DpigCriticParams* params, const float* real, const float* fake,
                            const float* alpha, float beta, const DpigCriticGrads* grads, float* penalty, float* slopes,
                            void* ws, size_t ws_bytes, void* stream);

/* ---- nearest-neighbour 2x upsample (unfused form; utils.py:61-72) ---------------------------- */
int dpig_upsample2x_fwd(const float* x, int N, int H, int W, int C, float* y, void* stream);
int dpig_upsample2x_bwd(const float* dy, int N, int H, int W, int C, float* dx, void* stream);

/* ---- TensorFlow-flavoured Adam on one flat tensor (trainer.py:137-140) ------------------------
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v updated in place; p -= lr_t*m/(sqrt(v)+eps).
 * `lr` is read from device memory (the reference keeps g_lr in a tf.Variable, trainer.py:56-59). */
int dpig_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev,
                   float beta1, float beta2, float eps, int step, float grad_scale, void* stream);
/* Same update with the step counter in device memory: state_dev = {int32 t; float corr}, zero-initialised
 * by the caller; every call advances t and the bias correction on the device, so the launch sequence
 * can be captured once in a hipGraph and replayed (a host `step` would be frozen into the graph). */
int dpig_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev,
                       void* state_dev, float beta1, float beta2, float eps, float grad_scale, void* stream);
/* Multi-tensor form: ptrs[4*i+0..3] = {p, g, m, v} device pointers of tensor i, sizes[i] elements.
 * `ptrs` and `sizes` are DEVICE arrays. */
int dpig_adam_multi(const void* const* ptrs_dev, const int64_t* sizes_dev, int ntensors, int64_t max_size,
                    const float* lr_dev, float beta1, float beta2, float eps, int step, float grad_scale,
                    void* stream);

/* ---- TF RMSProp (trainer.py:119-122; wgan / lsgan modes) and WGAN weight clipping (:124-128) -------
 * ms = decay*ms + (1-decay)*g^2; mom = momentum*mom + lr*g/sqrt(ms+eps); p -= mom.
 * TF initialises the `rms` slot to ones and `momentum` to zeros (the caller owns the buffers). */
int dpig_rmsprop_step(float* p, const float* g, float* ms, float* mom, int64_t n, const float* lr_dev,
                      float decay, float momentum, float eps, float grad_scale, void* stream);
int dpig_clip(float* p, int64_t n, float lo, float hi, void* stream);

/* ---- graph wiring between the convolutions (csrc/dpig_glue.hip): one launch each ---------------------------------------
 * x_fg = x * m, x_bg = x * (1 - m) (models.py:402-403); x [rows][C] fp32 or bf16 (is_bf16) with row stride, m [rows] fp32.
 * The gradient dx = dfg * m + dbg * (1 - m); either of dfg / dbg may be NULL. */
int dpig_mask_split_fwd(const void* x, int ldx, const float* m, int64_t rows, int C, void* fg, int ldfg, void* bg, int ldbg,
                        int is_bf16, void* stream);
int dpig_mask_split_bwd(const void* dfg, int ldfg, const void* dbg, int ldbg, const float* m, int64_t rows, int C, void* dx,
                        int ldx, int is_bf16, void* stream);
/* models.py:405-413: bbox [B][P_total][4] pixel (y1, x1, y2, x2), int32 or fp32 (is_float) -> boxes [P*B][4] = (y1/H, x1/W,
 * y2/H, x2/W), part-major (row p*B + b), and box_ind [P*B] = b: the arguments of tf.image.crop_and_resize. */
int dpig_roi_boxes(const void* bbox, int is_float, int B, int P_total, int P, float img_H, float img_W, float* boxes,
                   int32_t* box_ind, void* stream);
/* models.py:433-442, 467-468: all[b][p*z + c] = fea[p*B + b][c] * vis[b][p] (p < P), all[b][P*z + c] = bg[b][c] (c < zbg; bg
 * may be NULL with zbg = 0); the gradient scatters back the same way (dbg may be NULL). */
int dpig_vis_concat_fwd(const float* fea, const float* vis, int ldvis, const float* bg, int B, int P, int z, int zbg, float* all,
                        void* stream);
int dpig_vis_concat_bwd(const float* dall, const float* vis, int ldvis, int B, int P, int z, int zbg, float* dfea, float* dbg,
                        void* stream);
/* Tiled-embedding collapse of the generator's first conv (trainer.py:588-590, models.py:520-528): w [3][3][C][K] HWIO, the
 * first E input channels are the spatially constant embedding.  wmat[e][(cy*3 + cx)*K + k] = sum of the taps border class
 * (cy, cx) of a SAME 3x3 conv sees; _bwd is the transpose into dw[.][.][e < E][.] (dw = beta * dw + ...). */
int dpig_emb_class_weights_fwd(const float* w, int E, int C, int K, float* wmat, void* stream);
int dpig_emb_class_weights_bwd(const float* dwc, int E, int C, int K, float* dw, float beta, void* stream);
/* The generator's first conv (models.py:520-528 on concat(tile(emb), pose)) with the pose given as KEYPOINTS: the reference builds
 * the [B,H,W,P] pose map inside the graph from pose_rcv (trainer.py:556-560: coord2channel_simple_rcv + tf_poseInflate,
 * utils.py:237-318); a pose channel is -1 except a radius-4 disc, so its conv contribution is a border-class constant plus a sparse
 * sum around the keypoint -- y is produced without the map and without the dense P-channel conv.
 *   rcv [B][P][3] (row, col, visibility; normalised to [-1,1] if `normalized`), e9 [B][9][K] = emb @ class weights (dpig_linear_fwd on
 *   dpig_emb_class_weights_fwd's first E rows), cpos [9][K] = column sums of that matrix's pose rows, w [3][3][C = E + P][K] = the filter
 *   (its pose rows are channels E .. E + P - 1);
 *   y [B,H,W,K] fp32 or bf16 = act(e9[class] - cpos[class] + bias + sparse).
 * _wgrad: dwp [9][P][K] = gradient of the pose rows from z9 [B][9][K] (dpig_border_class_sum of dz) and dz [B,H,W,K]. */
int dpig_pose_stem_fwd(const float* rcv, int B, int P, int normalized, const float* e9, const float* cpos, const float* bias,
                       const float* w, int C, int E, int H, int W, int K, int act, float alpha, void* y, int is_bf16, void* stream);
int dpig_pose_stem_wgrad(const float* rcv, int B, int P, int normalized, const float* z9, const void* dz, int H, int W, int K, float* dwp,
                         int is_bf16, void* stream);
/* dz[N,H,W,C] = sum over the 2x2 block of dy[N,2H,2W,C] * act'(y[N,2H,2W,C]): the gradient of act(conv1x1(upsample2x(x)))
 * (models.py:569-570, utils.py:61-72) pulled back to the low-resolution grid the 1x1 conv is computed on -- TF's ReluGrad +
 * ResizeNearestNeighborGrad in one pass; its result feeds plain 1x1 dgrad / wgrad.  fp32 or bf16 (is_bf16) tensors; y may be
 * NULL when act is DPIG_ACT_NONE. */
int dpig_act_bwd_pool2x(const void* dy, int lddy, const void* y, int ldy, void* dz, int N, int H, int W, int C, int act, float alpha,
                        int is_bf16, void* stream);
/* dst[o][r][c] = beta * dst[o][r][c] + src[o][r][c] with independent outer / row strides (elements). */
int dpig_axpby3d(const float* src, int64_t s_outer, int64_t s_row, float* dst, int64_t d_outer, int64_t d_row, int outer, int rows,
                 int cols, float beta, void* stream);
/* y[b][c][a] = x[b][a][c] (2- or 4-byte elements): tf.reshape of the critic's logical NCHW tensor (wgan_gp.py:433) from the
 * physically NHWC activation, and its gradient. */
int dpig_transpose12(const void* x, void* y, int B, int A, int C, int elem_bytes, void* stream);

/* ---- losses (trainer.py:238-245, 607, 623) --------------------------------------------------- */
/* out[0] = mean_i sce(logits_i, label) ; dlogits_i = scale*(sigmoid(x_i)-label)/n (dlogits may be NULL) */
int dpig_sce_mean(const float* logits, int n, float label, float* out, float* dlogits, float scale,
                  void* stream);
/* The critic-output terms of the wgan / wgan-gp and lsgan losses (trainer.py:218-220, 246-248):
 * out[0] = mean_i x_i (squared = 0) or mean_i (x_i - target)^2 (squared = 1); dlogits (may be NULL) = scale * d out / dx_i */
int dpig_logit_mean(const float* logits, int n, int squared, float target, float* out, float* dlogits, float scale,
                    void* stream);
/* out[0] = mean |a-b| ; dgrad (may be NULL) = scale*sign(a-b)/n */
size_t dpig_l1_workspace_bytes(int64_t n);
int dpig_l1_mean(const float* a, const float* b, int64_t n, float* out, float* da, float scale, void* ws,
                 size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DPIG_HIP_H */
