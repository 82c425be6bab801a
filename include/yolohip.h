/* yolohip.h -- C ABI of libyolohip.so, the MI355X (gfx950) kernels behind the YOLO hot path.
 *
 * The reference (KhaledSharif/yolo-from-scratch) has no FFI layer: its hot path is the Python
 * surface of train.py, which dispatches to ATen.  Each entry point below replaces the ATen ops one
 * reference call site dispatches to; the citation after "replaces" is that call site.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless its comment says HOST (small
 *     configuration arrays: anchors, grid sizes, arrays of device pointers, the op list); the
 *     library never allocates or frees device memory and keeps no pointer after returning;
 *   - activations are NHWC fp32; a tensor argument is (ptr, ld) where ld = floats per pixel of the
 *     buffer the view lives in (channel-slice views of a concat buffer have ld > C);
 *   - calls are asynchronous and ordered on `stream` (a hipStream_t passed as void*); no call
 *     synchronises the device or the stream; all are hipGraph-capturable;
 *   - return 0 on success, a positive hipError_t or a negative YH_E_* otherwise; the message is
 *     available from yh_last_error(); no C++ exception crosses this boundary.
 */
#ifndef YOLOHIP_H
#define YOLOHIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define YH_E_BADARG (-1)
#define YH_E_UNSUPPORTED (-2)
#define YH_E_WORKSPACE (-3)

int yh_version(void);
const char *yh_last_error(void);    /* per host thread */

/* ---- execution context ------------------------------------------------------------------------ */
/* The only state the library keeps between calls lives in an explicit handle: the side HIP stream
 * and the fork / join events yh_run uses to overlap independent work.  A context binds to the device that is
 * current at its first run that needs the side lane (calls with another device current are then refused);
 * it serves ONE host thread at a time; distinct contexts share nothing, so concurrent yh_run calls on distinct
 * contexts and streams are safe.  yh_create itself makes no HIP call (it works on a host without a GPU).
 * Every other entry point is stateless apart from the per-thread error string and a mutex-guarded
 * (device, kernel) table of dynamic-LDS opt-ins.
 * replaces: nothing in the reference (ATen's stream/event pool is implicit there). */
typedef struct yh_context yh_context;
int yh_create(yh_context **out);
int yh_destroy(yh_context *ctx);          /* waits for the side stream, then frees it; NULL is a no-op */
/* enable = 0: yh_run executes every op in list order on the caller's stream.  Default: on; the environment variable
 * YH_OVERLAP=0 (read ONCE, by yh_create) makes contexts start with it off (profiling: serialised lanes).
 *
 * Environment variables of the whole package -- nothing is read from the environment on a launch path:
 *   YH_OVERLAP=0      library, per context at yh_create: no side lane (above)
 *   YH_BF16_STREAM=0  library, once per process: stride-1 bf16 layers on the segment kernels (conv_bf16.hip) instead of the
 *                     flat-stream kernels (conv_bf16_stream.hip) -- A/B switch
 *   YH_GENERIC=1      planner (graph.py), when a plan is traced: every convolution on the generic gather-GEMM / wgrad kernels
 *                     instead of the specialised families (the two product paths cross-check each other in the tests)
 *   YH_EVAL_FAST=0    planner: eval plans keep every layer on the gather GEMM (no fused Winograd / pointwise forms)
 *   YH_PW_X6=1        library, once per process (EXPERIMENTAL, default off): the forward 1x1 convolutions it supports run on the split-bf16
 *                     form of the fp32 GEMM (yh_conv_pw_fwd_x6 below)
 *   YH_FUSE_ACT=0     planner: no producer activation is applied by its consumers (every BatchNorm + SiLU pass is launched and the
 *                     normalised tensors exist in memory, as up to round 3) -- A/B switch and cross-check of the fused plans
 *   YH_BENCH_SHAPE / YH_BENCH_DTYPE / YH_BENCH_SIZE   bench.py only: informational shapes, never the reported metric
 * Diagnostic builds are compile-time: make EXTRA=-DYH_PW_STAMPS | -DYH_WINO_STAMPS | -DYH_BF_STAMPS | -DYH_WGS_TUNE. */
int yh_context_set_overlap(yh_context *ctx, int enable);
/* introspection for tests: bound device (-1 = not bound yet), overlap flag, raw handles (NULL
 * until the first forked run); any output pointer may be NULL */
int yh_context_info(const yh_context *ctx, int *device, int *overlap, void **side_stream, void **fork_event,
                    void **join_event);

/* ---- layout ------------------------------------------------------------------------------- */
/* NCHW (B,C,H,W) -> NHWC with ld floats per pixel (channels >= C are zero filled up to cpad).
 * replaces: the NCHW image batch handed to YOLO.forward (train.py:568). */
int yh_nchw_to_nhwc(const float *src, float *dst, int B, int C, int H, int W, int ld, int cpad, void *stream);
/* (B,H,W,C) uint8 image batch -> NHWC fp32, value / 255.0f (true division: bit-identical to the reference's
 * `torch.from_numpy(np.array(img)).permute(2,0,1).float() / 255.0`, train.py:115-117), channels [C, cpad) zeroed. */
int yh_u8hwc_to_nhwc(const uint8_t *src, float *dst, int B, int H, int W, int C, int ld, int cpad, void *stream);
/* NHWC view (C channels, ld) -> NCHW contiguous; `accumulate` adds into dst. */
int yh_nhwc_to_nchw(const float *src, float *dst, int B, int C, int H, int W, int ld, int accumulate, void *stream);
/* OIHW weights -> forward pack [kh*kw][Cin_pad][ldwf] and backward-data pack [kh*kw][Cout][ldwb];
 * either destination may be NULL.  Cin_pad >= Cin rows beyond Cin are zero (stem: 3 -> 4). */
int yh_pack_weights(const float *oihw, float *wf, float *wb, int Cout, int Cin, int k, int cin_pad, int ldwf,
                    int ldwb, void *stream);
/* The same for n_layers convolutions in ONE launch.  `table` is a DEVICE array of n_layers records of
 * 56 bytes: { const float *oihw; float *wf; float *wb; int32 Cout, Cin, k*k, cin_pad, ldwf, ldwb, 0, 0 }. */
int yh_pack_weights_multi(const void *table, int n_layers, void *stream);

/* ---- convolution (implicit GEMM on v_mfma_f32_32x32x2_f32) ---------------------------------- */
/* Forward: y = conv(x, w) (+ bias); square kernel k in {1,3}, stride s in {1,2}, pad k/2.
 * If bn_partials != NULL the epilogue also writes per-workgroup per-channel sum / sum of squares
 * of y: bn_partials[(blk*2 + {0,1})*Cout + c], blk < yh_conv_fwd_blocks(...).
 * replaces: nn.Conv2d.forward inside ConvBlock (train.py:260,265), the inline stem/downsample
 * convs (train.py:402-418), SPPF convs (236,240) and head output convs (455,460,465). */
int yh_conv_fwd(const float *x, int ldx, const float *wf, int ldwf, const float *bias, float *y, int ldy,
                float *bn_partials, int B, int Hi, int Wi, int Cin, int Cout, int k, int s, void *stream);
int yh_conv_fwd_blocks(int B, int Hi, int Wi, int Cout, int k, int s);
/* Inference forward: y = [upsample x2]( silu(conv(x, wf) + bias) + residual ) in ONE kernel; wf / bias come from
 * yh_pack_fold_multi (eval-mode BatchNorm folded in).  act_silu = 0 gives conv + bias only.
 * replaces: model.eval() forward of ConvBlock / inline conv+BN+SiLU (train.py:265, 401-418, 244-251) inside
 * predict (train.py:1140-1141). */
int yh_conv_fwd_fused(const float *x, int ldx, const float *wf, int ldwf, const float *bias, const float *res, int ldr,
                      float *y, int ldy, int B, int Hi, int Wi, int Cin, int Cout, int k, int s, int act_silu,
                      int upsample, void *stream);
/* Small-M (batch-1 inference) form of yh_conv_fwd_fused: when a layer would launch few workgroups its K axis (taps x
 * channels) is split over S workgroups per output tile INSIDE ONE launch; partial tiles go to fp32 slabs in ws and the
 * last workgroup to arrive at a tile (agent-scope ticket) adds them in fixed order and applies bias / SiLU / residual /
 * upsample -- deterministic, no second launch.  ws_floats >= yh_conv_fwd_fused_ws(...) = S*M*ld slab floats followed by
 * one int32 ticket per tile; the TICKETS MUST BE ZERO before the first call (the reducer leaves them zero), and one ws
 * must not be shared by launches that can run concurrently.  With ws == NULL or a layer the heuristic leaves alone
 * (yh_conv_fwd_fused_ws == 0) it is yh_conv_fwd_fused. */
int yh_conv_fwd_fused_splitk(const float *x, int ldx, const float *wf, int ldwf, const float *bias, const float *res, int ldr,
                             float *y, int ldy, float *ws, int64_t ws_floats, int B, int Hi, int Wi, int Cin, int Cout, int k,
                             int s, int act_silu, int upsample, void *stream);
int64_t yh_conv_fwd_fused_ws(int B, int Hi, int Wi, int Cin, int Cout, int k, int s);
/* Fold eval-mode BatchNorm into packed forward weights for n_layers convolutions in one launch.  `table` is a
 * DEVICE array of 88-byte records { const float *oihw, *bias_in, *gamma, *beta, *running_mean, *running_var;
 * float *wf, *bias_out; int32 Cout, Cin, k*k, cin_pad, ldwf; float eps } (gamma == NULL: plain pack + bias copy). */
int yh_pack_fold_multi(const void *table, int n_layers, void *stream);
/* Eval-mode BatchNorm folded into plain OIHW weights + a bias (the input of yh_pack_weights_multi / yh_wino_weights_multi /
 * yh_pw_pack_multi, so every forward kernel family can run the folded layer).  `table`: DEVICE array of 80-byte records
 * { const float *oihw, *bias_in, *gamma, *beta, *running_mean, *running_var; float *oihw_out, *bias_out; int32 Cout,
 * Cin*k*k; float eps; int32 0 } (gamma == NULL: copy). */
int yh_fold_oihw_multi(const void *table, int n_layers, void *stream);
/* Inference forms of the Winograd / pointwise forward kernels: y = [upsample x2]( silu(conv(x) + bias) + residual ) with
 * folded weights (yh_fold_oihw_multi -> yh_wino_weights_multi / yh_pw_pack_multi); same eligibility rules as
 * yh_conv_wino_fwd / yh_conv_pw_fwd.  For large batches (eval_epoch, predict_batch); batch-1 latency uses
 * yh_conv_fwd_fused_splitk.  replaces: the same call sites as yh_conv_fwd_fused (train.py:1130-1141, 960-1032). */
int yh_conv_wino_fwd_fused(const float *x, int ldx, const float *U, int ldu, const float *bias, const float *res, int ldr, float *y,
                           int ldy, int B, int H, int W, int Cin, int Cout, int act_silu, int upsample, void *stream);
int yh_conv_pw_fwd_fused(const float *x, int ldx, const float *wq, int ldw, const float *bias, const float *res, int ldr, float *y,
                         int ldy, int B, int H, int W, int Cin, int Cout, int act_silu, int upsample, void *stream);
/* Backward-data: dx (+)= conv_transpose(dy, w).  replaces: aten::convolution_backward (input
 * gradient) reached from loss.backward() (train.py:913). */
int yh_conv_bwd_data(const float *dy, int lddy, const float *wb, int ldwb, float *dx, int lddx, int B, int Hi,
                     int Wi, int Cin, int Cout, int k, int s, int accumulate, void *stream);
/* Stride-2 3x3 backward-data with the two column parities merged into the channel axis (narrow layers, Cin <= 32): dx must
 * be pixel-dense (lddx == Cin) with an even width; wbm = yh_pack_weights_s2m(oihw): [kh 3][c 2][Cout][ldw >= 2*Cin].
 * Same result as yh_conv_bwd_data(k = 3, s = 2); replaces the input gradient of stem[3], train.py:407. */
int yh_conv_bwd_data_s2m(const float *dy, int lddy, const float *wbm, int ldw, float *dx, int lddx, int B, int Hi, int Wi,
                         int Cin, int Cout, int accumulate, void *stream);
int yh_pack_weights_s2m(const float *oihw, float *wbm, int Cout, int Cin, int ldw, void *stream);
/* Backward-data of TWO pointwise (1x1, stride 1) convolutions that share their input (C3's conv1 / conv2):
 * dx (+)= dy1 * W1^T + dy2 * W2^T as one GEMM over K = cout1 + cout2 (dx written once, no read-modify-write).
 * dy1 / dy2: (B,H,W,*) views with the same ld; wb: the two backward packs stacked, rows [0,cout1) then [cout1, cout1+cout2). */
int yh_conv_bwd_data_pair(const float *dy1, int cout1, const float *dy2, int cout2, int lddy, const float *wb, int ldwb,
                          float *dx, int lddx, int B, int H, int W, int Cin, int accumulate, void *stream);
/* The narrow high-resolution 3x3 layers (16 -> 16 stride 1: the C3 bottleneck at 1/4 resolution; 16 -> 32 stride 2: stem[3]) as
 * a direct convolution on v_mfma_f32_16x16x4_f32: the input halo patch is read from HBM once into LDS, the whole filter sits
 * in registers, 16-wide MFMA tiles (no padding of the 16 output channels).  w: the forward pack [9][Cin][ldw] of
 * yh_pack_weights(_multi); with flip_taps = 1 and the BACKWARD pack [9][Cout][ldw] it computes the stride-1 input gradient
 * (then x = dY, Cin / Cout = the convolution's Cout / Cin).  Same results and bn_partials contract as yh_conv_fwd
 * ([yh_conv_narrow_blocks(...)][2][Cout]).  replaces: train.py:300-306 (Bottleneck convs), 407 (stem[3]), 913. */
int yh_conv_narrow_ok(int Cin, int Cout, int k, int s);
int yh_conv_narrow_blocks(int B, int Hi, int Wi, int Cin, int s);
int yh_conv_narrow(const float *x, int ldx, const float *w, int ldw, const float *bias, float *y, int ldy, float *bn_partials, int B,
                   int Hi, int Wi, int Cin, int Cout, int s, int flip_taps, int accumulate, void *stream);
/* Input gradient of the stride-2 narrow layer (stem[3], Cin = 16, Cout = 32) as a direct kernel: one MFMA tile = 16 dX pixels
 * of one (row, column) parity class, which receive 1, 2, 2 or 4 taps -- no zero-stuffed taps, dY read from HBM once.  wb: the
 * backward pack [9][Cout][ldwb]; Hi, Wi: the size of dX.  Same result as yh_conv_bwd_data(k = 3, s = 2). */
int yh_conv_narrow_dgrad_s2_ok(int Cin, int Cout);
int yh_conv_narrow_dgrad_s2(const float *dy, int lddy, const float *wb, int ldwb, float *dx, int lddx, int B, int Hi, int Wi, int Cin,
                            int Cout, int accumulate, void *stream);
/* Weight gradient of the narrow layers as a direct kernel: 16 -> 16 stride 1, 16 -> 32 stride 2 and the first layer
 * (Cin = 4 padded channels of which cin_real are real, -> 16, stride 2).  The reduction runs over pixels: one 16x16x4 MFMA
 * takes 4 consecutive output pixels as k, the 16 rows are the input channels of one tap (or 4 taps x 4 channels), the
 * columns 16 output channels; persistent workgroups keep all accumulator tiles in registers and write one slab each, summed
 * in fixed order (bitwise reproducible).  Every x and dY element is read from HBM once.  Writes OIHW; same result as
 * yh_conv_bwd_weight(k = 3).  dbias != NULL: also the conv's bias gradient (column sums of dY: the pieces pass through the
 * kernel's registers anyway, which saves the separate yh_colsum pass over dY).  ws_floats >= yh_conv_narrow_bwd_weight_ws(...).  replaces: aten::convolution_backward (weight
 * gradient) of stem[0], stem[3] and the 1/4-resolution Bottleneck convs, train.py:913. */
int yh_conv_narrow_bwd_weight_ok(int Cin, int cin_real, int Cout, int k, int s);
int64_t yh_conv_narrow_bwd_weight_ws(int B, int Hi, int Wi, int Cin, int Cout, int s);
int yh_conv_narrow_bwd_weight(const float *x, int ldx, const float *dy, int lddy, float *dw, float *dbias, float *ws, int64_t ws_floats,
                              int B, int Hi, int Wi, int Cin, int cin_real, int Cout, int s, void *stream);
/* The narrow-layer kernels with bf16 storage (bf16 path, BASELINE configs 3-4): activations, activation gradients and the
 * weight packs of yh_bf16_pack_multi ([tap][kpad / 8][ld][8]) are bf16 in HBM; they are widened when parked in LDS and the fp32
 * MFMA loop of the fp32 kernels runs unchanged (these layers are bound by HBM bytes and latency, products of bf16 values are
 * exact in fp32).  Stored results are rounded to bf16, BatchNorm partial sums are taken over the rounded values, weight
 * gradients are fp32.  Same contracts as the fp32 entry points otherwise; the first layer reads the first 4 channels of its
 * (8-channel padded) input.  Same results as yh_bf16_conv_fwd / _bwd_data / _bwd_weight to fp32 accumulation order. */
int yh_bf16_conv_narrow(const void *x, int ldx, const void *w, int ldw, int kpad, const float *bias, void *y, int ldy, float *bn_partials,
                        int B, int Hi, int Wi, int Cin, int Cout, int s, int flip_taps, int accumulate, void *stream);
int yh_bf16_conv_narrow_dgrad_s2(const void *dy, int lddy, const void *wb, int ldwb, int kpad, void *dx, int lddx, int B, int Hi, int Wi,
                                 int Cin, int Cout, int accumulate, void *stream);
int yh_bf16_conv_narrow_bwd_weight(const void *x, int ldx, const void *dy, int lddy, float *dw, float *dbias, float *ws, int64_t ws_floats,
                                   int B, int Hi, int Wi, int Cin, int cin_real, int Cout, int s, void *stream);
/* Winograd F(2x2,3x3) path for 3x3 / stride-1 / pad-1 convolutions with even H, W and K % 16 == 0 (K = Cin forward,
 * Cout backward): the same results as yh_conv_fwd / yh_conv_bwd_data to fp32 rounding with 4/9 of the multiplies.
 * yh_wino_weights transforms OIHW weights into U[16][K][ldu] (backward = 0: K = Cin, N = Cout; backward = 1: the
 * flipped, transposed filter, K = Cout, N = Cin); ldu % 4 == 0, ldu >= N.  bn_partials: [yh_conv_wino_blocks][2][Cout]
 * per-workgroup channel sums / sums of squares (same contract as yh_conv_fwd), or NULL.
 * replaces: nn.Conv2d(k=3, s=1) forward and input gradient (train.py:260-265, 300-306, 913). */
int yh_wino_weights(const float *oihw, float *U, int Cout, int Cin, int ldu, int backward, void *stream);
/* n transforms in one launch; `table` is a DEVICE array of 32-byte records { const float *oihw; float *U; int32 Cout, Cin,
 * ldu, backward; }. */
int yh_wino_weights_multi(const void *table, int n, void *stream);
int yh_conv_wino_blocks(int B, int H, int W);
int yh_conv_wino_fwd(const float *x, int ldx, const float *U, int ldu, const float *bias, float *y, int ldy,
                     float *bn_partials, int B, int H, int W, int Cin, int Cout, void *stream);
int yh_conv_wino_bwd_data(const float *dy, int lddy, const float *Ub, int ldub, float *dx, int lddx, int B, int H,
                          int W, int Cin, int Cout, int accumulate, void *stream);
/* Latency-oriented fused convolution for layers with FEW PIXELS (batch-1 inference, BASELINE config 5; round 4): conv (k 1|3,
 * s 1|2) + folded-BatchNorm bias + SiLU (+ residual) (+ x2 upsample) like yh_conv_fwd_fused, with K split over the 4-16 waves of a
 * workgroup that owns 32 pixels x 32 channels -- operands straight into the MFMA registers, the partial tiles summed through
 * LDS in wave order: no split-K slabs, fences or tickets (the in-launch split-K of yh_conv_fwd_fused_splitk costs three extra
 * memory round trips per layer).  wq: k-quad interleaved pack Wq[(k >> 2)][ldw][k & 3], k = tap * Cin + ci, written by
 * yh_lat_pack_multi from (folded) OIHW weights; Cin % 8 == 0.  replaces: the eval-mode conv forward of predict()
 * (train.py:253-265, 1140-1141). */
int yh_conv_lat_ok(int B, int Hi, int Wi, int Cin, int Cout, int k, int s);
int yh_conv_lat_fwd_fused(const float *x, int ldx, const float *wq, int ldw, const float *bias, const float *res, int ldr, float *y,
                          int ldy, int B, int Hi, int Wi, int Cin, int Cout, int k, int s, int act_silu, int upsample, void *stream);
/* n packs in one launch; `table` is a DEVICE array of 32-byte records { const float *oihw; float *wq; int32 Cout, Cin, k*k, ldw; }. */
int yh_lat_pack_multi(const void *table, int n, void *stream);
/* 3x3 stride-2 forward with the input patch staged through LDS (round 4: the down-sampling layers with Cin % 8 == 0 and
 * Cout % 32 == 0; yh_conv_s2_ok).  wq: the k-quad interleaved pack of yh_lat_pack_multi (k = tap * Cin + ci); icoef: the input
 * prologue table or NULL; bn_partials: [yh_conv_s2_blocks][2][Cout] or NULL.  Same results as yh_conv_fwd to fp32 summation
 * order.  replaces: nn.Conv2d(k=3, s=2, p=1) forward (train.py:408-418, 593-597). */
int yh_conv_s2_ok(int B, int H, int W, int Cin, int Cout);
int yh_conv_s2_blocks(int B, int H, int W, int Cout);
int yh_conv_s2_fwd_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *wq, int ldw, const float *bias, float *y,
                       int ldy, float *bn_partials, int B, int H, int W, int Cin, int Cout, void *stream);
/* ---- input prologue (round 4) ----------------------------------------------------------------------------------------------
 * Every forward / weight-gradient entry point below has an `_act` form taking the prologue table of its x operand: icoef =
 * three rows [scale | shift | gate] of Cin floats, icoef_ld floats apart (NULL: the plain entry point).  The kernel reads the
 * PRODUCER's raw convolution output and uses z = x * scale[k] + shift[k], silu(z) where gate[k] != 0, as its input -- the
 * producer's BatchNorm + SiLU applied while the operand is staged, zero padding after the activation, so the normalised tensor
 * never exists in memory (train.py:253-265 executed at the consumer).  Same results as the plain entry point on the
 * materialised activation (same expression, same hardware exp / rcp as yh_bn_silu_fwd). */
int yh_conv_fwd_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *wf, int ldwf, const float *bias, float *y, int ldy,
                    float *bn_partials, int B, int Hi, int Wi, int Cin, int Cout, int k, int s, void *stream);
int yh_conv_bwd_weight_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *dy, int lddy, float *dw, float *ws,
                           int64_t ws_floats, int B, int Hi, int Wi, int Cin, int cin_real, int Cout, int k, int s, void *stream);
int yh_conv_bwd_weight_prologue_ok(int B, int Hi, int Wi, int Cin, int Cout, int k, int s);
int yh_conv_pw_prologue_ok(int64_t M, int Cin, int Cout);       /* the streaming and tiled pointwise forward kernels have a prologue */
int yh_conv_pw_fwd_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *wq, int ldw, const float *bias, float *y, int ldy,
                       float *bn_partials, int64_t M, int Cin, int Cout, void *stream);
/* EXPERIMENTAL (round 4): the same forward GEMM on the bf16 matrix pipe -- every fp32 operand split into three bf16 terms in registers,
 * six exact bf16 products per fp32 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16: 0.375 of the fp32 instruction's matrix time,
 * error against fp64 BELOW the fp32 instruction's (only the accumulation rounds).  Same contract and results (to fp32 rounding) as
 * yh_conv_pw_fwd_act; Cin in {16, 32, 64, 128}, Cout <= 128 (Cin * Cout small enough for the weights' three planes in LDS), M >= 4096:
 * yh_conv_pw_x6_blocks returns the partial-sum rows, or 0 when the kernel cannot run the problem.  Not used by the planner unless
 * YH_PW_X6=1 (see csrc/conv_pw.hip for what was measured). */
int yh_conv_pw_x6_blocks(int64_t M, int Cin, int Cout);
int yh_conv_pw_fwd_x6(const float *x, int ldx, const float *icoef, int icoef_ld, const float *wq, int ldw, const float *bias, float *y, int ldy,
                      float *bn_partials, int64_t M, int Cin, int Cout, void *stream);
int yh_conv_pw_fwd2_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *wq, int ldw, const float *bias1, float *y1,
                        int ldy1, float *bn_partials1, int cout1, const float *bias2, float *y2, int ldy2, float *bn_partials2, int cout2,
                        int64_t M, int Cin, void *stream);
int yh_conv_pw_bwd_weight_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *dy, int lddy, float *dw, float *ws,
                              int64_t ws_floats, int64_t M, int Cin, int Cout, void *stream);
int yh_conv_narrow_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *w, int ldw, const float *bias, float *y, int ldy,
                       float *bn_partials, int B, int Hi, int Wi, int Cin, int Cout, int s, void *stream);      /* 16-channel inputs */
int yh_conv_narrow_bwd_weight_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *dy, int lddy, float *dw, float *dbias,
                                  float *ws, int64_t ws_floats, int B, int Hi, int Wi, int Cin, int cin_real, int Cout, int s, void *stream);
/* Forward with an input prologue.  icoef (or NULL = yh_conv_wino_fwd): the input prologue table [scale | shift | gate], three rows of Cin floats icoef_ld floats apart -- the
 * PRODUCER's BatchNorm + SiLU fused into the staging: the kernel reads the producer's raw convolution output x and uses
 * z = x * scale[k] + shift[k], silu(z) where gate[k] != 0, as its input; zero padding is applied after the activation, so the
 * normalised tensor never exists in memory (replaces the ConvBlock.forward hand-over, train.py:253-265).
 * bn_partials: [yh_conv_wino_blocks][2][Cout]. */
int yh_conv_wino_fwd_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *U, int ldu, const float *bias,
                         float *y, int ldy, float *bn_partials, int B, int H, int W, int Cin, int Cout, void *stream);
/* Backward-weight in the Winograd domain (Cin % 32 == 0, Cout % 32 == 0, even H, W): dw (OIHW) = the same sum as
 * yh_conv_bwd_weight for k = 3, s = 1, deterministic (fixed-order reduction of per-workgroup [9][Cin][Cout] slabs
 * through ws, ws_floats >= yh_conv_wino_bwd_weight_ws(...)). */
int yh_conv_wino_bwd_weight(const float *x, int ldx, const float *dy, int lddy, float *dw, float *ws, int64_t ws_floats,
                            int B, int H, int W, int Cin, int Cout, void *stream);
/* The same with the input prologue applied to x (see yh_conv_wino_fwd_act): the weight gradient of a convolution whose input
 * activation was never materialised. */
int yh_conv_wino_bwd_weight_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *dy, int lddy, float *dw, float *ws,
                                int64_t ws_floats, int B, int H, int W, int Cin, int Cout, void *stream);
int64_t yh_conv_wino_bwd_weight_ws(int B, int H, int W, int Cin, int Cout);
/* Backward-weight of a pointwise (1x1, stride 1) convolution over M = B*H*W pixels: dw[co][ci] = sum_p x[p][ci] dy[p][co],
 * MFMA operands loaded straight into registers (no LDS in the loop), deterministic slab reduction through ws
 * (ws_floats >= yh_conv_pw_bwd_weight_ws(...)).  Same result as yh_conv_bwd_weight with k = 1, s = 1.
 * replaces: aten::convolution_backward (weight gradient) of every 1x1 conv, train.py:913. */
int yh_conv_pw_bwd_weight(const float *x, int ldx, const float *dy, int lddy, float *dw, float *ws, int64_t ws_floats,
                          int64_t M, int Cin, int Cout, void *stream);
int64_t yh_conv_pw_bwd_weight_ws(int64_t M, int Cin, int Cout);
/* Forward / backward-data of a pointwise (1x1, stride 1) convolution over M = B*H*W pixels as a register-direct MFMA
 * GEMM (no LDS in the loop).  Weights in the k-quad interleaved layout Wq[K/4][ldw][4] written by yh_pw_pack_multi
 * (`table`: DEVICE array of 48-byte records { const float *oihw; float *wq_fwd, *wq_bwd; int32 Cout, Cin, ldw_fwd,
 * ldw_bwd, koff_bwd, noff_fwd }; either destination may be NULL; koff_bwd / noff_fwd = first K row / first column of this
 * conv inside a stacked backward / forward matrix; the forward pack writes only its own columns: allocate wq_fwd zeroed).  Channel counts feeding K must be multiples of 8.  yh_conv_pw_fwd: same contract as yh_conv_fwd
 * (k = 1), bn_partials [yh_conv_pw_blocks(M, Cin, Cout)][2][Cout] (K = the reduction length: small-K, many-pixel layers run a
 * streaming form with one partial row per workgroup).  yh_conv_pw_bwd_data: dx (+)= dy1 W1^T (+ dy2 W2^T when
 * dy2 != NULL: the C3 sibling pair, K = cout1 + cout2 stacked in wq).
 * replaces: nn.Conv2d(k=1) forward and input gradient, train.py:236-240, 282-296, 402-418, 913. */
int yh_pw_pack_multi(const void *table, int n, void *stream);
int yh_conv_pw_blocks(int64_t M, int K, int Cout);
int yh_conv_pw_fwd(const float *x, int ldx, const float *wq, int ldw, const float *bias, float *y, int ldy, float *bn_partials,
                   int64_t M, int Cin, int Cout, void *stream);
/* Two sibling pointwise convolutions that read the same x (C3's conv1 / conv2, train.py:282-293) as ONE GEMM with
 * N = cout1 + cout2: wq holds both weight matrices side by side (pack records with noff = 0 and noff = cout1 into a
 * zero-initialised buffer), each output tensor keeps its own ld, bias and BatchNorm partials
 * [yh_conv_pw_blocks(M, Cin, cout1 + cout2)][2][cout_i]. */
int yh_conv_pw_fwd2(const float *x, int ldx, const float *wq, int ldw, const float *bias1, float *y1, int ldy1,
                    float *bn_partials1, int cout1, const float *bias2, float *y2, int ldy2, float *bn_partials2, int cout2,
                    int64_t M, int Cin, void *stream);
int yh_conv_pw_bwd_data(const float *dy1, int cout1, const float *dy2, int cout2, int lddy, const float *wq, int ldw, float *dx,
                        int lddx, int64_t M, int Cin, int accumulate, void *stream);
/* Backward-weight: dw[co][ci][kh][kw] = sum_pixels x * dy, deterministic two-stage reduction
 * through `ws` (ws_floats >= yh_conv_bwd_weight_ws(...)).  Writes OIHW (Cin_real input channels)
 * into dw.  replaces: aten::convolution_backward (weight gradient), train.py:913. */
int yh_conv_bwd_weight(const float *x, int ldx, const float *dy, int lddy, float *dw, float *ws,
                       int64_t ws_floats, int B, int Hi, int Wi, int Cin, int cin_real, int Cout, int k, int s,
                       void *stream);
int64_t yh_conv_bwd_weight_ws(int B, int Hi, int Wi, int Cin, int Cout, int k, int s);
/* Per-channel column sum of an NHWC view (bias gradient); ws_floats >= yh_colsum_ws(M, C). */
int yh_colsum(const float *x, int ldx, int64_t M, int C, float *out, float *ws, void *stream);
int64_t yh_colsum_ws(int64_t M, int C);

/* ---- BatchNorm2d + SiLU ----------------------------------------------------------------------- */
/* Reduce conv-epilogue partials to batch mean / biased variance; write scale = gamma*invstd and
 * shift = beta - mean*scale into coef[0..C) / coef[C..2C), mean / invstd into coef[2C..4C);
 * update running_mean / running_var (unbiased) with `momentum` when running_mean != NULL and
 * increment *num_batches_tracked when it is != NULL.
 * replaces: nn.BatchNorm2d training-mode statistics (train.py:261,265; torch native_batch_norm). */
int yh_bn_finalize(const float *partials, int nblk, int64_t count, const float *gamma, const float *beta,
                   float *running_mean, float *running_var, float momentum, float eps, float *coef, int C,
                   int64_t *num_batches_tracked, void *stream);
/* The same, and additionally the rows `scale` / `shift` of a CONSUMER-side input-prologue table (xscale / xshift, both or
 * neither): when the activation of this layer is never materialised, the kernels that read it apply it (yh_conv_*_act). */
int yh_bn_finalize_x(const float *partials, int nblk, int64_t count, const float *gamma, const float *beta, float *running_mean,
                     float *running_var, float momentum, float eps, float *coef, int C, int64_t *num_batches_tracked, float *xscale,
                     float *xshift, int64_t *bwd_acc, void *stream);      /* bwd_acc (or NULL): [2][C] accumulators of the backward, zeroed here */
/* Inference coefficients from running statistics (BN folded to scale/shift). */
int yh_bn_eval_coef(const float *gamma, const float *beta, const float *running_mean, const float *running_var,
                    float eps, float *coef, int C, void *stream);
/* a = silu(y*scale + shift) (+ residual); optional nearest x2 upsample on write (out is then
 * (B,2H,2W)).  replaces: BatchNorm2d.forward + nn.SiLU (train.py:265), the Bottleneck residual add
 * (train.py:306), nn.Upsample (train.py:431,436) and the torch.cat copies (train.py:250,293,585-598)
 * because `out` may be a channel slice of the concat buffer. */
int yh_bn_silu_fwd(const float *y, int ldy, const float *coef, const float *residual, int ldr, float *out,
                   int ldo, int64_t M, int C, int H, int W, int upsample, void *stream);
/* The same with a residual that was never materialised: `residual` is its producer's raw convolution output, rcoef its
 * prologue table rows [scale | shift | gate], rcoef_ld floats apart. */
int yh_bn_silu_fwd_res(const float *y, int ldy, const float *coef, const float *residual, int ldr, const float *rcoef, int rcoef_ld,
                       float *out, int ldo, int64_t M, int C, int H, int W, int upsample, void *stream);
/* Backward stage 1: per-channel partial sums of dz and dz*xhat where dz = da*silu'(z);
 * da is read with a 2x2 sum when `upsample`.  partials: [nblk][2][C], nblk = yh_bn_bwd_blocks(M). */
int yh_bn_silu_bwd_reduce(const float *da, int ldda, const float *y, int ldy, const float *coef, float *partials,
                          int64_t M, int C, int H, int W, int upsample, void *stream);
/* The same two passes WITHOUT the finalize launch between them (round 4): the reduce adds every workgroup's channel sums into
 * acc[2][C] -- 64-bit fixed-point accumulators (2^-36 units; integer atomics, so the totals are bitwise reproducible in any
 * arrival order), which must be ZERO on entry (yh_bn_finalize_x zeroes them in the forward pass of the same step); the apply
 * converts the totals once per workgroup and workgroup 0 writes dgamma / dbeta. */
int yh_bn_silu_bwd_reduce_acc(const float *da, int ldda, const float *y, int ldy, const float *coef, int64_t *acc, int64_t M, int C, int H,
                              int W, int upsample, void *stream);
int yh_bn_silu_bwd_apply_acc(const float *da, int ldda, const float *y, int ldy, const float *coef, const int64_t *acc, float *dgamma,
                             float *dbeta, float *dy, int lddy, float *dres, int lddres, int res_accumulate, int64_t M, int C, int H, int W,
                             int upsample, void *stream);
int yh_bf16_bn_silu_bwd_reduce_acc(const void *da, int ldda, const void *y, int ldy, const float *coef, int64_t *acc, int64_t M, int C,
                                   int H, int W, int upsample, void *stream);
int yh_bf16_bn_silu_bwd_apply_acc(const void *da, int ldda, const void *y, int ldy, const float *coef, const int64_t *acc, float *dgamma,
                                  float *dbeta, void *dy, int lddy, void *dres, int lddres, int res_accumulate, int64_t M, int C, int H,
                                  int W, int upsample, void *stream);
int yh_bn_bwd_blocks(int64_t M, int C);
/* Backward stage 2: dy = scale*(dz - mean(dz) - xhat*mean(dz*xhat)); dgamma, dbeta written from the
 * partials; if dres != NULL the incoming da is also routed to the residual branch
 * (dres = da or dres += da).  replaces: native_batch_norm_backward + silu_backward (train.py:913). */
int yh_bn_silu_bwd_apply(const float *da, int ldda, const float *y, int ldy, const float *coef,
                         const float *partials, int nblk, const float *gamma, float *dgamma, float *dbeta,
                         float *dy, int lddy, float *dres, int lddres, int res_accumulate, int64_t M, int C,
                         int H, int W, int upsample, void *stream);

/* ---- SPPF max-pool 5x5 / stride 1 / pad 2 ------------------------------------------------------ */
/* replaces: nn.MaxPool2d(5,1,2) (train.py:239,246-248); argmax (window tap 0..24, first max wins)
 * is kept for the backward pass. */
int yh_maxpool5_fwd(const float *x, int ldx, float *y, int ldy, uint8_t *argmax, int B, int H, int W, int C,
                    void *stream);
/* dx += route(dy) (gather form, deterministic). */
int yh_maxpool5_bwd(const float *dy, int lddy, const uint8_t *argmax, float *dx, int lddx, int B, int H, int W,
                    int C, void *stream);
/* Inference form of the SPPF pooling cascade (train.py:246-248): y1 = pool5(x), y2 = pool5(y1), y3 = pool5(y2) in one
 * launch, no argmax.  One workgroup per (4 channels, image) keeps the plane in LDS: needs yh_sppf_pool3_ok(H, W)
 * (H*W <= 4096); y1..y3 share ldy. */
int yh_sppf_pool3_ok(int H, int W);
int yh_sppf_pool3_fwd(const float *x, int ldx, float *y1, float *y2, float *y3, int ldy, int B, int H, int W, int C,
                      void *stream);

/* ---- loss: decode + CIoU + BCE, three scales, forward and backward in one pass ----------------- */
/* pred[s], target[s]: (B,G_s,G_s,3,5+nc) contiguous, G_s = grid[s]; grid[s] == 0 marks an absent scale
 * (pred/target/dpred/grid are HOST arrays of device pointers / ints); anchors: HOST, 3x3x2 floats
 * (pixels), captured by value.  loss_w / grad_w: HOST, 9 floats = per scale (box, obj, cls) weights of
 * the total and of its gradient; NULL loss_w = the reference's {0.05, {4.0,1.0,0.4}, 0.5}
 * (train.py:865,879), NULL grad_w = loss_w.
 * out[0..4) = total, sum box, sum obj (unweighted), sum cls (train.py:886); out[4+3s..] = per-scale
 * box/obj/cls.  dpred[s] (NULL array or NULL entry -> forward only) receives
 * sum_k grad_w[s][k] * d loss_k / d pred[s]: EVERY element of it is written (whole rows, padding of a padded row
 * included; no need to clear it first).  ws: >= yh_loss_ws(...) floats, 8-byte aligned (positive counts, per-workgroup
 * partial sums, one int32 per cell for the lists of positive cells).
 * loss_img_size is the decode img_size (reference: always 640, quirk Q1).
 * replaces: decode_predictions + ciou_loss + yolo_loss + yolo_loss_multiscale and their autograd
 * (train.py:634-886, 909, 913). */
int yh_yolo_loss(const float *const pred[3], const float *const target[3], float *const dpred[3],
                 const float *anchors, const int grid[3], int B, int nc, float loss_img_size, const float *loss_w,
                 const float *grad_w, float *out, float *ws, void *stream);
int64_t yh_loss_ws(const int grid[3], int B);
/* The same with a typed gradient: dpred_bf16 != 0 writes bf16 (the bf16 path's head gradient); dpred_ld[s] = elements
 * per PIXEL of dpred[s] (3 anchors x (5+nc), possibly padded to a multiple of 8 so that the backward GEMMs can read it
 * in 16-byte pieces; the padding is zeroed), NULL or 0 = contiguous. */
int yh_yolo_loss_ex(const float *const pred[3], const float *const target[3], void *const dpred[3], int dpred_bf16,
                    const int dpred_ld[3], const float *anchors, const int grid[3], int B, int nc, float loss_img_size,
                    const float *loss_w, const float *grad_w, float *out, float *ws, void *stream);
/* eval_epoch's detection metric (train.py:990-1024): counts[0..3) += TP, FP, FN over all cells of the three
 * scales -- same cell, same anchor; sigmoid(obj) > conf_thr vs target obj > conf_thr; a matched pair is a TP when
 * compute_box_iou (centre format, eps 1e-6) > iou_thr, else an FP.  counts is a DEVICE int64[3] the caller zeroes
 * (integer atomics: order-independent).  decode_img_size: the reference decodes with its default 640 here too. */
int yh_eval_counts(const float *const pred[3], const float *const target[3], const float *anchors, const int grid[3],
                   int B, int nc, float decode_img_size, float conf_thr, float iou_thr, int64_t *counts, void *stream);
/* Standalone pieces of the same math for the reference's public functions (anchors3x2: HOST).
 * yh_ciou: ws >= 2*ceil(N/256) floats, 8-byte aligned; N must be > 0. */
int yh_decode(const float *raw, float *out, const float *anchors3x2, int B, int GH, int GW, int nc,
              float img_size, void *stream);                               /* train.py:712-779 */
int yh_decode_bwd(const float *raw, const float *dout, float *draw, const float *anchors3x2, int B, int GH,
                  int GW, int nc, float img_size, void *stream);
int yh_ciou(const float *pred, const float *tgt, float *dpred, int64_t N, float eps, float grad_scale,
            float *loss_out, float *ws, void *stream);                     /* train.py:634-710 */

/* ---- inference post-process --------------------------------------------------------------------- */
/* Candidate extraction of predict() (train.py:1152-1229) for one image: keeps cells with
 * sigmoid(obj) > conf_thr in scale-major, row-major (i,j,a) order.  boxes (cap,4) corners in
 * original-image pixels, scores (cap), classes (cap) int32, count[0] = M (may exceed cap: the
 * caller must check).  ws: >= yh_candidates_ws(grid) ints.  letterbox_dev: NULL, or a DEVICE array
 * {pad_left, pad_top, scale} that overrides the three scalars (so a captured hipGraph can be replayed
 * with per-image letterbox parameters). */
int yh_candidates(const float *const pred[3], const float *anchors, const int grid[3], int nc, float img_size,
                  float conf_thr, float pad_left, float pad_top, float scale, float *boxes, float *scores,
                  int32_t *classes, int32_t *count, int cap, int32_t *ws, const float *letterbox_dev, void *stream);
int64_t yh_candidates_ws(const int grid[3]);
/* torchvision.ops.batched_nms as called at train.py:1232-1233: stable descending score order (ties: lower candidate
 * index first), greedy suppression of IoU > thr.  Both of torchvision's branches, selected by `mode`:
 *   YH_NMS_PER_CLASS         _batched_nms_vanilla: suppression only between boxes of the same class;
 *   YH_NMS_COORDINATE_TRICK  _batched_nms_coordinate_trick: boxes + float(class) * (boxes.max() + 1) in fp32 (one rounding
 *                            per operation), then ONE class-agnostic NMS over the shifted boxes;
 *   YH_NMS_TORCHVISION_CPU   torchvision's own rule for a CPU tensor -- the reference's CPU path, the parity target:
 *                            4 M > 4000 -> per class, else coordinate trick (decided on the device from count[0]);
 *   YH_NMS_TORCHVISION_CUDA  what the reference executes when predict() runs on a GPU device: the same rule with torchvision's
 *                            limit for GPU tensors (20000) AND its device kernel's float-against-float comparison.
 * Which reference device each mode reproduces: PER_CLASS / COORDINATE_TRICK / TORCHVISION_CPU = torchvision's CPU kernel
 * (`double iou_threshold`: the fp32 IoU is promoted for the comparison, so thr = 0.4 suppresses an IoU of exactly
 * float32(0.4)); TORCHVISION_CUDA = the CUDA kernel (`float iou_threshold`: that pair is kept).  M is read from count[0] on the device (clamped to
 * cap).  keep (cap) int32 receives kept candidate indices in descending score order, nkeep[0] their number.
 * ws: >= yh_nms_ws(cap) bytes, 256-byte aligned. */
#define YH_NMS_PER_CLASS 0
#define YH_NMS_COORDINATE_TRICK 1
#define YH_NMS_TORCHVISION_CPU 2
#define YH_NMS_TORCHVISION_CUDA 3
int yh_nms(const float *boxes, const float *scores, const int32_t *classes, const int32_t *count, int cap,
           double iou_thr, int mode, int32_t *keep, int32_t *nkeep, void *ws, void *stream);
int64_t yh_nms_ws(int cap);
/* The result list of predict() (train.py:1236-1246) as one dense device table, so the host needs ONE copy: out[0], out[1]
 * (int32 bits) = count[0], min(nkeep[0], cap); out[8 + 6 k ...] = x1, y1, x2, y2, score, class (int32 bits) of the k-th
 * kept candidate in NMS order.  out: >= 8 + 6 * cap 32-bit words. */
int yh_gather_detections(const float *boxes, const float *scores, const int32_t *classes, const int32_t *count,
                         const int32_t *keep, const int32_t *nkeep, int cap, float *out, void *stream);

/* ---- input side: label lists -> dense target grids --------------------------------- */
/* labels: DEVICE double [B][maxn][5] = (class, xc, yc, w, h) normalised to the padded square image, nlabels:
 * DEVICE int32 [B].  Writes the three (B,G,G,3,5+nc) fp32 target tensors (zeroed first) with the reference's rule:
 * best shape-IoU anchor over all nine, cell = min(int(c*G), G-1), first label wins a cell/anchor, one-hot class.
 * replaces: the target construction in YOLODataset.__getitem__ (train.py:140-205). */
int yh_assign_targets(const double *labels, const int32_t *nlabels, int B, int maxn, const float *anchors,
                      const int grid[3], int nc, int img_size, float *const target[3], void *stream);

/* ---- optimiser: global-norm clip + Adam over flat buffers -------------------------------------- */
/* norm_out[0] = grad_scale * ||g||_2 (fp32), deterministic two-stage fp64 reduce; ws >= yh_sqnorm_ws(n)
 * doubles.  grad_scale = 1/world_size folds the data-parallel mean into the norm.
 * replaces: torch.nn.utils.clip_grad_norm_ (train.py:916). */
int yh_grad_sqnorm(const float *g, int64_t n, float grad_scale, float *norm_out, double *ws, void *stream);
int64_t yh_sqnorm_ws(int64_t n);
/* g *= grad_scale * min(1, max_norm/(norm+1e-6)) (the clip is skipped when max_norm <= 0 or norm == NULL),
 * then one Adam step (torch.optim.Adam defaults, step is 1-based).  replaces: train.py:916-918. */
int yh_adam_step(float *p, float *g, float *m, float *v, int64_t n, double lr, double beta1, double beta2,
                 double eps, int step, float max_norm, const float *norm, float grad_scale, void *stream);

/* ---- bf16 path (BASELINE configs 3-4: "bf16 MFMA implicit-GEMM conv") --------------------------------------------- */
/* Activations, activation gradients and the per-step weight packs are bf16 (void * below = bf16 elements, ld counted in
 * elements); accumulators, bias, BatchNorm statistics / coefficients, the loss and every parameter gradient are fp32;
 * the master weights stay fp32 OIHW.  Convolutions run on v_mfma_f32_32x32x16_bf16; channel counts feeding a GEMM's K
 * axis and every ld must be multiples of 8 (16-byte pieces), views 16-byte aligned.
 * replaces: the same call sites as the fp32 entry points of the same name (train.py:253-265, 401-466, 913). */
/* OIHW fp32 -> bf16 packs for n convolutions in one launch.  `table`: DEVICE array of 56-byte records
 * { const float *oihw; bf16 *wf, *wb; int32 Cout, Cin, k*k, cin_pad, ldf, ldb, koff_b, kpad_b }:
 * wf[tap][cin_pad/8][ldf][8] (forward: K = cin_pad, N = Cout), wb[tap][kpad_b/8][ldb][8] (backward-data: K rows
 * [koff_b, koff_b + Cout) of a matrix with kpad_b rows -- two C3 sibling convs stack into one --, N = Cin); either
 * destination may be NULL; rows / columns beyond the real channel counts are zero. */
int yh_bf16_pack_multi(const void *table, int n, void *stream);
/* y = conv(x, wf) (+ bias); y is bf16, or fp32 when y_f32 (head outputs feeding the fp32 loss).  bn_partials:
 * [yh_bf16_conv_fwd_blocks(B, Hi, Wi, Cin, Cout, k, s, y_f32, ldx, ldy)][2][Cout] sums / sums of squares of the STORED (bf16-rounded)
 * values, or NULL.  Stride-1 layers with 16 / 32 / 64 / 128 input channels (3x3: up to 64) and bf16 output run as a flat
 * pixel stream with ONE partial row per persistent workgroup (conv_bf16_stream.hip); the rest on the gather GEMM with one
 * row per 128 output pixels (yh_bf16_conv_blocks(M)).  The route is a function of shapes and strides only (the same predicate sizes the
 * partial-row table): when ldx and ldy are multiples of 8, x, wf and y must be 16-byte aligned (argument error otherwise); views with
 * other strides run on the gather kernel at any 2-byte alignment.  yh_bf16_conv_bwd_data: the same rule for lddy / lddx. */
int yh_bf16_conv_fwd(const void *x, int ldx, const void *wf, int ldwf, const float *bias, void *y, int ldy, int y_f32,
                     float *bn_partials, int B, int Hi, int Wi, int Cin, int Cout, int k, int s, void *stream);
int yh_bf16_conv_blocks(int64_t M);
/* The flat-stream kernels forced (tests, benchmarks; the dispatch above picks them where they measured faster): same contracts
 * as yh_bf16_conv_fwd with bf16 output / yh_bf16_conv_bwd_data at stride 1.  yh_bf16_conv_stream_blocks: 0 when the kernel
 * cannot run the problem (K streamed channels in {16, 32, 64, 128}, 3x3: K <= 64; N % 8 == 0), else its BatchNorm partial rows. */
int yh_bf16_conv_stream_blocks(int B, int Hi, int Wi, int K, int N, int k);
int yh_bf16_conv_stream_fwd(const void *x, int ldx, const void *wf, int ldwf, const float *bias, void *y, int ldy, float *bn_partials,
                            int B, int Hi, int Wi, int Cin, int Cout, int k, void *stream);
int yh_bf16_conv_stream_bwd_data(const void *dy, int lddy, const void *dy2, int kcout1, const void *wb, int ldwb, void *dx, int lddx,
                                 int B, int Hi, int Wi, int Cin, int Cout, int k, int accumulate, void *stream);
int yh_bf16_conv_fwd_blocks(int B, int Hi, int Wi, int Cin, int Cout, int k, int s, int y_f32, int ldx, int ldy);
/* dx (+)= conv_transpose(dy, wb); Cout = K rows of wb (a multiple of 8: pad dY with zeros).  dy2 != NULL: pointwise
 * only, K rows [0, kcout1) come from dy, [kcout1, Cout) from dy2 (same ld): the fused C3 sibling pair. */
int yh_bf16_conv_bwd_data(const void *dy, int lddy, const void *dy2, int kcout1, const void *wb, int ldwb, void *dx, int lddx,
                          int B, int Hi, int Wi, int Cin, int Cout, int k, int s, int accumulate, void *stream);
/* dw (OIHW fp32, cin_real input channels) = sum_pixels x * dy; deterministic (fp32 slabs in ws, fixed-order sum).
 * dY must be readable up to roundup8(Cout) channels per pixel. */
int yh_bf16_conv_bwd_weight(const void *x, int ldx, const void *dy, int lddy, float *dw, float *ws, int64_t ws_floats,
                            int B, int Hi, int Wi, int Cin, int cin_real, int Cout, int k, int s, void *stream);
int64_t yh_bf16_conv_bwd_weight_ws(int B, int Hi, int Wi, int Cin, int Cout, int k, int s);
/* bf16 forms of the HBM-bound passes (same arguments as the fp32 entry points; compute in fp32) */
int yh_bf16_nchw_to_nhwc(const float *src, void *dst, int B, int C, int H, int W, int ld, int cpad, void *stream);
int yh_bf16_u8hwc_to_nhwc(const uint8_t *src, void *dst, int B, int H, int W, int C, int ld, int cpad, void *stream);
int yh_bf16_nhwc_to_nchw(const void *src, float *dst, int B, int C, int H, int W, int ld, int accumulate, void *stream);
int yh_bf16_colsum(const void *x, int ldx, int64_t M, int C, float *out, float *ws, void *stream);
int yh_bf16_bn_silu_fwd(const void *y, int ldy, const float *coef, const void *residual, int ldr, void *out, int ldo, int64_t M,
                        int C, int H, int W, int upsample, void *stream);
int yh_bf16_bn_silu_bwd_reduce(const void *da, int ldda, const void *y, int ldy, const float *coef, float *partials, int64_t M,
                               int C, int H, int W, int upsample, void *stream);
int yh_bf16_bn_silu_bwd_apply(const void *da, int ldda, const void *y, int ldy, const float *coef, const float *partials,
                              int nblk, const float *gamma, float *dgamma, float *dbeta, void *dy, int lddy, void *dres,
                              int lddres, int res_accumulate, int64_t M, int C, int H, int W, int upsample, void *stream);
int yh_bf16_maxpool5_fwd(const void *x, int ldx, void *y, int ldy, uint8_t *argmax, int B, int H, int W, int C, void *stream);
int yh_bf16_maxpool5_bwd(const void *dy, int lddy, const uint8_t *argmax, void *dx, int lddx, int B, int H, int W, int C,
                         void *stream);

/* ---- small utilities (stream-ordered) --------------------------------------------------------------- */
int yh_memset(void *p, int value, int64_t bytes, void *stream);
/* *p += v on the device (BatchNorm2d.num_batches_tracked). */
int yh_add_int64(int64_t *p, int64_t v, void *stream);

/* ---- op-list executor --------------------------------------------------------------------------- */
/* One record per kernel launch; `kind` selects the entry point above, the arrays carry its
 * arguments in declaration order (pointers in p[], ints in i[], floats in f[]). */
typedef struct yh_op {
    int32_t kind;
    int32_t lane;      /* 0 = the caller's stream; 1 = the library's side stream (between YH_OP_FORK / YH_OP_JOIN) */
    int32_t i[19];
    float f[4];
    void *p[12];
    int64_t l[2];
} yh_op;
enum {
    YH_OP_NCHW_TO_NHWC = 1, YH_OP_NHWC_TO_NCHW, YH_OP_PACK_WEIGHTS, YH_OP_CONV_FWD, YH_OP_CONV_BWD_DATA,
    YH_OP_CONV_BWD_WEIGHT, YH_OP_COLSUM, YH_OP_BN_FINALIZE, YH_OP_BN_EVAL_COEF, YH_OP_BN_SILU_FWD,
    YH_OP_BN_SILU_BWD_REDUCE, YH_OP_BN_SILU_BWD_APPLY, YH_OP_MAXPOOL5_FWD, YH_OP_MAXPOOL5_BWD, YH_OP_MEMSET,
    YH_OP_ADD_INT64, YH_OP_PACK_WEIGHTS_MULTI, YH_OP_PACK_FOLD_MULTI, YH_OP_CONV_FWD_FUSED,
    YH_OP_CONV_BWD_DATA_PAIR,
    YH_OP_FORK,   /* side lane waits for everything issued on the caller's stream so far */
    YH_OP_JOIN,   /* caller's stream waits for everything issued on the side lane so far */
    YH_OP_WINO_WEIGHTS_MULTI, YH_OP_CONV_WINO_FWD, YH_OP_CONV_WINO_BWD_DATA, YH_OP_CONV_WINO_BWD_WEIGHT,
    YH_OP_CONV_PW_BWD_WEIGHT, YH_OP_PW_PACK_MULTI, YH_OP_CONV_PW_FWD, YH_OP_CONV_PW_BWD_DATA,
    YH_OP_CONV_STEM_FWD, YH_OP_PACK_WEIGHTS_S2M, YH_OP_CONV_BWD_DATA_S2M, YH_OP_CONV_PW_FWD2, YH_OP_NOP,
    /* bf16 path: argument slots as in the fp32 op of the same name unless noted in api.hip */
    YH_OP_BF16_PACK_MULTI, YH_OP_BF16_CONV_FWD, YH_OP_BF16_CONV_BWD_DATA, YH_OP_BF16_CONV_BWD_WEIGHT, YH_OP_BF16_COLSUM,
    YH_OP_BF16_BN_SILU_FWD, YH_OP_BF16_BN_SILU_BWD_REDUCE, YH_OP_BF16_BN_SILU_BWD_APPLY, YH_OP_BF16_MAXPOOL5_FWD,
    YH_OP_BF16_MAXPOOL5_BWD,
    YH_OP_FOLD_OIHW_MULTI, YH_OP_CONV_WINO_FWD_FUSED, YH_OP_CONV_PW_FWD_FUSED,  /* slots of YH_OP_CONV_FWD_FUSED */
    YH_OP_CONV_NARROW,  /* p: x, w, bias, y, partials;  i: ldx, ldw, ldy, B, H, W, Cin, Cout, s, flip_taps, accumulate */
    YH_OP_CONV_NARROW_DGRAD_S2,  /* slots of YH_OP_CONV_BWD_DATA */
    YH_OP_CONV_NARROW_BWD_WEIGHT, /* slots of YH_OP_CONV_BWD_WEIGHT + p[4] = dbias | NULL */
    YH_OP_BF16_CONV_NARROW,       /* slots of YH_OP_CONV_NARROW + i[11] = kpad (padded K rows per tap of the bf16 pack) */
    YH_OP_BF16_CONV_NARROW_DGRAD_S2,   /* slots of YH_OP_CONV_BWD_DATA + i[11] = kpad */
    YH_OP_BF16_CONV_NARROW_BWD_WEIGHT, /* slots of YH_OP_CONV_BWD_WEIGHT + p[4] = dbias | NULL */
    YH_OP_SPPF_POOL3,                  /* p: x, y1, y2, y3;  i: ldx, ldy, B, H, W, C */
    YH_OP_CONV_LAT_FWD_FUSED,          /* slots of YH_OP_CONV_FWD_FUSED (p[1] = the k-quad interleaved pack of yh_lat_pack_multi) */
    YH_OP_LAT_PACK_MULTI,              /* p: descriptor table;  i: n */
    YH_OP_CONV_S2_FWD                  /* slots of YH_OP_CONV_FWD (k = 3, s = 2 implied; p[1] = a yh_lat_pack_multi pack) */
};
/* Runs ops[0..n) in order on `stream`; stops at the first failure and returns its code
 * (failing index in *failed when non-NULL).  Two lanes: ops with lane == 1 run on the context's side
 * stream, ordered against the caller's stream only at YH_OP_FORK / YH_OP_JOIN records (independent
 * sub-graphs, e.g. detection heads next to the neck).  Backward-weight and column-sum ops are forked
 * automatically (they only feed the optimiser).  Everything launched is joined back into `stream` before the
 * call returns -- also when an op fails -- so stream order is preserved for the caller.  ctx == NULL (or
 * yh_context_set_overlap(ctx, 0), or YH_OVERLAP=0) runs the whole list in order on `stream`. */
int yh_run(yh_context *ctx, const yh_op *ops, int n, void *stream, int *failed);

#ifdef __cplusplus
}
#endif
#endif
