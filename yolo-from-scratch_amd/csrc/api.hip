// Error reporting, version and the op-list executor of libyolohip.
#include "common.h"
#include <string.h>
#include <stdlib.h>
#include <mutex>
#include <new>
#include <vector>

static thread_local char g_err[512] = "";

void yh_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *yh_last_error(void) { return g_err; }
extern "C" int yh_version(void) { return 100; }

extern "C" int yh_memset(void *p, int value, int64_t bytes, void *stream) {
    YH_REQUIRE(p && bytes >= 0, "memset: bad argument");
    YH_HIP(hipMemsetAsync(p, value, (size_t)bytes, (hipStream_t)stream));
    return 0;
}

// Dispatch one record to its entry point; argument order = declaration order in yolohip.h.
static int run_one(const yh_op &o, void *st) {
    const int32_t *i = o.i;
    const float *f = o.f;
    void *const *p = o.p;
    switch (o.kind) {
        case YH_OP_NCHW_TO_NHWC:
            return yh_nchw_to_nhwc((const float *)p[0], (float *)p[1], i[0], i[1], i[2], i[3], i[4], i[5], st);
        case YH_OP_NHWC_TO_NCHW:
            return yh_nhwc_to_nchw((const float *)p[0], (float *)p[1], i[0], i[1], i[2], i[3], i[4], i[5], st);
        case YH_OP_PACK_WEIGHTS:
            return yh_pack_weights((const float *)p[0], (float *)p[1], (float *)p[2], i[0], i[1], i[2], i[3], i[4], i[5], st);
        case YH_OP_CONV_FWD:
            return yh_conv_fwd_act((const float *)p[0], i[0], (const float *)p[10], i[17], (const float *)p[1], i[1], (const float *)p[2],
                                   (float *)p[3], i[2], (float *)p[4], i[3], i[4], i[5], i[6], i[7], i[8], i[9], st);
        case YH_OP_CONV_BWD_DATA:
            return yh_conv_bwd_data((const float *)p[0], i[0], (const float *)p[1], i[1], (float *)p[2], i[2], i[3], i[4],
                                    i[5], i[6], i[7], i[8], i[9], i[10], st);
        case YH_OP_CONV_BWD_WEIGHT:
            return yh_conv_bwd_weight_act((const float *)p[0], i[0], (const float *)p[10], i[17], (const float *)p[1], i[1], (float *)p[2],
                                          (float *)p[3], o.l[0], i[2], i[3], i[4], i[5], i[6], i[7], i[8], i[9], st);
        case YH_OP_COLSUM:
            return yh_colsum((const float *)p[0], i[0], o.l[0], i[1], (float *)p[1], (float *)p[2], st);
        case YH_OP_BN_FINALIZE:        /* p[7], p[8]: scale / shift rows of a consumer-side prologue table, or NULL */
            return yh_bn_finalize_x((const float *)p[0], i[0], o.l[0], (const float *)p[1], (const float *)p[2],
                                    (float *)p[3], (float *)p[4], f[0], f[1], (float *)p[5], i[1], (int64_t *)p[6], (float *)p[7],
                                    (float *)p[8], (int64_t *)p[9], st);          /* p[9]: the layer's backward accumulators, zeroed here */
        case YH_OP_BN_EVAL_COEF:
            return yh_bn_eval_coef((const float *)p[0], (const float *)p[1], (const float *)p[2], (const float *)p[3], f[0],
                                   (float *)p[4], i[0], st);
        case YH_OP_BN_SILU_FWD:        /* p[4] / i[7]: prologue table of a residual that was never materialised, or NULL */
            return yh_bn_silu_fwd_res((const float *)p[0], i[0], (const float *)p[1], (const float *)p[2], i[1], (const float *)p[4], i[7],
                                      (float *)p[3], i[2], o.l[0], i[3], i[4], i[5], i[6], st);
        case YH_OP_BN_SILU_BWD_REDUCE:     /* p[9] != NULL: fixed-point accumulators instead of partial rows (no finalize launch) */
            if (p[9])
                return yh_bn_silu_bwd_reduce_acc((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2], (int64_t *)p[9],
                                                 o.l[0], i[2], i[3], i[4], i[5], st);
            return yh_bn_silu_bwd_reduce((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2],
                                         (float *)p[3], o.l[0], i[2], i[3], i[4], i[5], st);
        case YH_OP_BN_SILU_BWD_APPLY:
            if (p[9])
                return yh_bn_silu_bwd_apply_acc((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2], (const int64_t *)p[9],
                                                (float *)p[5], (float *)p[6], (float *)p[7], i[3], (float *)p[8], i[4], i[5], o.l[0], i[6],
                                                i[7], i[8], i[9], st);
            return yh_bn_silu_bwd_apply((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2],
                                        (const float *)p[3], i[2], (const float *)p[4], (float *)p[5], (float *)p[6],
                                        (float *)p[7], i[3], (float *)p[8], i[4], i[5], o.l[0], i[6], i[7], i[8], i[9], st);
        case YH_OP_MAXPOOL5_FWD:
            return yh_maxpool5_fwd((const float *)p[0], i[0], (float *)p[1], i[1], (uint8_t *)p[2], i[2], i[3], i[4], i[5], st);
        case YH_OP_SPPF_POOL3:
            return yh_sppf_pool3_fwd((const float *)p[0], i[0], (float *)p[1], (float *)p[2], (float *)p[3], i[1], i[2], i[3], i[4],
                                     i[5], st);
        case YH_OP_MAXPOOL5_BWD:
            return yh_maxpool5_bwd((const float *)p[0], i[0], (const uint8_t *)p[1], (float *)p[2], i[1], i[2], i[3], i[4],
                                   i[5], st);
        case YH_OP_MEMSET:
            return yh_memset(p[0], i[0], o.l[0], st);
        case YH_OP_ADD_INT64:
            return yh_add_int64((int64_t *)p[0], o.l[0], st);
        case YH_OP_WINO_WEIGHTS_MULTI:
            return yh_wino_weights_multi(p[0], i[0], st);
        /* convolution records carry the input-prologue table of their x operand in p[10] (NULL: plain input), its row stride in i[17] */
        case YH_OP_CONV_WINO_FWD:
            return yh_conv_wino_fwd_act((const float *)p[0], i[0], (const float *)p[10], i[17], (const float *)p[1], i[1], (const float *)p[2],
                                        (float *)p[3], i[2], (float *)p[4], i[3], i[4], i[5], i[6], i[7], st);
        case YH_OP_CONV_WINO_BWD_DATA:
            return yh_conv_wino_bwd_data((const float *)p[0], i[0], (const float *)p[1], i[1], (float *)p[2], i[2], i[3], i[4],
                                         i[5], i[6], i[7], i[8], st);
        case YH_OP_CONV_WINO_BWD_WEIGHT:
            return yh_conv_wino_bwd_weight_act((const float *)p[0], i[0], (const float *)p[10], i[17], (const float *)p[1], i[1], (float *)p[2],
                                               (float *)p[3], o.l[0], i[2], i[3], i[4], i[5], i[7], st);
        case YH_OP_CONV_PW_BWD_WEIGHT:      /* same argument slots as YH_OP_CONV_BWD_WEIGHT */
            return yh_conv_pw_bwd_weight_act((const float *)p[0], i[0], (const float *)p[10], i[17], (const float *)p[1], i[1], (float *)p[2],
                                             (float *)p[3], o.l[0], (int64_t)i[2] * i[3] * i[4], i[5], i[7], st);
        case YH_OP_CONV_STEM_FWD:           /* retired (the first layer runs on narrow_conv_kernel); the enum slot stays for ABI stability */
            yh_set_error("yh_run: YH_OP_CONV_STEM_FWD was retired in round 3");
            return YH_E_UNSUPPORTED;
        case YH_OP_PACK_WEIGHTS_S2M:
            return yh_pack_weights_s2m((const float *)p[0], (float *)p[1], i[0], i[1], i[2], st);
        case YH_OP_CONV_BWD_DATA_S2M:       /* same argument slots as YH_OP_CONV_BWD_DATA */
            return yh_conv_bwd_data_s2m((const float *)p[0], i[0], (const float *)p[1], i[1], (float *)p[2], i[2], i[3], i[4], i[5],
                                        i[6], i[7], i[10], st);
        case YH_OP_CONV_PW_FWD2:            /* p: x, wq, bias1, y1, part1, bias2, y2, part2;  i: ldx, ldw, ldy1, B, H, W, Cin, cout1, ldy2, cout2 */
            return yh_conv_pw_fwd2_act((const float *)p[0], i[0], (const float *)p[10], i[17], (const float *)p[1], i[1], (const float *)p[2],
                                       (float *)p[3], i[2], (float *)p[4], i[7], (const float *)p[5], (float *)p[6], i[8], (float *)p[7], i[9],
                                       (int64_t)i[3] * i[4] * i[5], i[6], st);
        case YH_OP_PW_PACK_MULTI:
            return yh_pw_pack_multi(p[0], i[0], st);
        case YH_OP_CONV_PW_FWD:             /* same argument slots as YH_OP_CONV_FWD */
            return yh_conv_pw_fwd_act((const float *)p[0], i[0], (const float *)p[10], i[17], (const float *)p[1], i[1], (const float *)p[2],
                                      (float *)p[3], i[2], (float *)p[4], (int64_t)i[3] * i[4] * i[5], i[6], i[7], st);
        case YH_OP_CONV_PW_BWD_DATA:        /* p: dy1, dy2 | NULL, wq, dx;  i: cout1, cout2, lddy, ldw, lddx, B, H, W, Cin, accumulate */
            return yh_conv_pw_bwd_data((const float *)p[0], i[0], (const float *)p[1], i[1], i[2], (const float *)p[2], i[3],
                                       (float *)p[3], i[4], (int64_t)i[5] * i[6] * i[7], i[8], i[9], st);
        case YH_OP_NOP:
            return 0;
        case YH_OP_CONV_NARROW:
            if (p[10])
                return yh_conv_narrow_act((const float *)p[0], i[0], (const float *)p[10], i[17], (const float *)p[1], i[1], (const float *)p[2],
                                          (float *)p[3], i[2], (float *)p[4], i[3], i[4], i[5], i[6], i[7], i[8], st);
            return yh_conv_narrow((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2], (float *)p[3], i[2],
                                  (float *)p[4], i[3], i[4], i[5], i[6], i[7], i[8], i[9], i[10], st);
        case YH_OP_CONV_NARROW_DGRAD_S2:    /* same argument slots as YH_OP_CONV_BWD_DATA */
            return yh_conv_narrow_dgrad_s2((const float *)p[0], i[0], (const float *)p[1], i[1], (float *)p[2], i[2], i[3], i[4], i[5],
                                           i[6], i[7], i[10], st);
        case YH_OP_CONV_NARROW_BWD_WEIGHT:  /* same argument slots as YH_OP_CONV_BWD_WEIGHT (k = 3 implied) */
            return yh_conv_narrow_bwd_weight_act((const float *)p[0], i[0], (const float *)p[10], i[17], (const float *)p[1], i[1], (float *)p[2],
                                                 (float *)p[4], (float *)p[3], o.l[0], i[2], i[3], i[4], i[5], i[6], i[7], i[9], st);
        case YH_OP_BF16_CONV_NARROW:
            return yh_bf16_conv_narrow(p[0], i[0], p[1], i[1], i[11], (const float *)p[2], p[3], i[2], (float *)p[4], i[3], i[4], i[5], i[6],
                                       i[7], i[8], i[9], i[10], st);
        case YH_OP_BF16_CONV_NARROW_DGRAD_S2:
            return yh_bf16_conv_narrow_dgrad_s2(p[0], i[0], p[1], i[1], i[11], p[2], i[2], i[3], i[4], i[5], i[6], i[7], i[10], st);
        case YH_OP_BF16_CONV_NARROW_BWD_WEIGHT:
            return yh_bf16_conv_narrow_bwd_weight(p[0], i[0], p[1], i[1], (float *)p[2], (float *)p[4], (float *)p[3], o.l[0], i[2], i[3], i[4],
                                                  i[5], i[6], i[7], i[9], st);
        case YH_OP_FOLD_OIHW_MULTI:
            return yh_fold_oihw_multi(p[0], i[0], st);
        case YH_OP_CONV_WINO_FWD_FUSED:     /* slots of YH_OP_CONV_FWD_FUSED (k = 3, s = 1 implied) */
            return yh_conv_wino_fwd_fused((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2], (const float *)p[3],
                                          i[2], (float *)p[4], i[3], i[4], i[5], i[6], i[7], i[8], i[11], i[12], st);
        case YH_OP_CONV_PW_FWD_FUSED:       /* slots of YH_OP_CONV_FWD_FUSED (k = 1, s = 1 implied) */
            return yh_conv_pw_fwd_fused((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2], (const float *)p[3],
                                        i[2], (float *)p[4], i[3], i[4], i[5], i[6], i[7], i[8], i[11], i[12], st);
        // ---- bf16 path ----------------------------------------------------------------------------------------------
        case YH_OP_BF16_PACK_MULTI:
            return yh_bf16_pack_multi(p[0], i[0], st);
        case YH_OP_BF16_CONV_FWD:           /* slots of YH_OP_CONV_FWD; i[10] = output is fp32 */
            return yh_bf16_conv_fwd(p[0], i[0], p[1], i[1], (const float *)p[2], p[3], i[2], i[10], (float *)p[4], i[3], i[4], i[5],
                                    i[6], i[7], i[8], i[9], st);
        case YH_OP_BF16_CONV_BWD_DATA:      /* slots of YH_OP_CONV_BWD_DATA; p[3] / i[11]: second source dy2 / its first K row */
            return yh_bf16_conv_bwd_data(p[0], i[0], p[3], i[11], p[1], i[1], p[2], i[2], i[3], i[4], i[5], i[6], i[7], i[8], i[9],
                                         i[10], st);
        case YH_OP_BF16_CONV_BWD_WEIGHT:    /* slots of YH_OP_CONV_BWD_WEIGHT */
            return yh_bf16_conv_bwd_weight(p[0], i[0], p[1], i[1], (float *)p[2], (float *)p[3], o.l[0], i[2], i[3], i[4], i[5],
                                           i[6], i[7], i[8], i[9], st);
        case YH_OP_BF16_COLSUM:
            return yh_bf16_colsum(p[0], i[0], o.l[0], i[1], (float *)p[1], (float *)p[2], st);
        case YH_OP_BF16_BN_SILU_FWD:
            return yh_bf16_bn_silu_fwd(p[0], i[0], (const float *)p[1], p[2], i[1], p[3], i[2], o.l[0], i[3], i[4], i[5], i[6], st);
        case YH_OP_BF16_BN_SILU_BWD_REDUCE:
            if (p[9])
                return yh_bf16_bn_silu_bwd_reduce_acc(p[0], i[0], p[1], i[1], (const float *)p[2], (int64_t *)p[9], o.l[0], i[2], i[3], i[4],
                                                      i[5], st);
            return yh_bf16_bn_silu_bwd_reduce(p[0], i[0], p[1], i[1], (const float *)p[2], (float *)p[3], o.l[0], i[2], i[3], i[4],
                                              i[5], st);
        case YH_OP_BF16_BN_SILU_BWD_APPLY:
            if (p[9])
                return yh_bf16_bn_silu_bwd_apply_acc(p[0], i[0], p[1], i[1], (const float *)p[2], (const int64_t *)p[9], (float *)p[5],
                                                     (float *)p[6], p[7], i[3], p[8], i[4], i[5], o.l[0], i[6], i[7], i[8], i[9], st);
            return yh_bf16_bn_silu_bwd_apply(p[0], i[0], p[1], i[1], (const float *)p[2], (const float *)p[3], i[2],
                                             (const float *)p[4], (float *)p[5], (float *)p[6], p[7], i[3], p[8], i[4], i[5], o.l[0],
                                             i[6], i[7], i[8], i[9], st);
        case YH_OP_BF16_MAXPOOL5_FWD:
            return yh_bf16_maxpool5_fwd(p[0], i[0], p[1], i[1], (uint8_t *)p[2], i[2], i[3], i[4], i[5], st);
        case YH_OP_BF16_MAXPOOL5_BWD:
            return yh_bf16_maxpool5_bwd(p[0], i[0], (const uint8_t *)p[1], p[2], i[1], i[2], i[3], i[4], i[5], st);
        case YH_OP_CONV_BWD_DATA_PAIR:
            return yh_conv_bwd_data_pair((const float *)p[0], i[0], (const float *)p[1], i[1], i[2], (const float *)p[2], i[3],
                                         (float *)p[3], i[4], i[5], i[6], i[7], i[8], i[9], st);
        case YH_OP_PACK_WEIGHTS_MULTI:
            return yh_pack_weights_multi(p[0], i[0], st);
        case YH_OP_PACK_FOLD_MULTI:
            return yh_pack_fold_multi(p[0], i[0], st);
        case YH_OP_CONV_FWD_FUSED:          /* p[5] / l[0]: optional split-K workspace (small-M layers) */
            return yh_conv_fwd_fused_splitk((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2],
                                            (const float *)p[3], i[2], (float *)p[4], i[3], (float *)p[5], o.l[0], i[4], i[5], i[6],
                                            i[7], i[8], i[9], i[10], i[11], i[12], st);
        case YH_OP_CONV_LAT_FWD_FUSED:      /* slots of YH_OP_CONV_FWD_FUSED */
            return yh_conv_lat_fwd_fused((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2], (const float *)p[3], i[2],
                                         (float *)p[4], i[3], i[4], i[5], i[6], i[7], i[8], i[9], i[10], i[11], i[12], st);
        case YH_OP_LAT_PACK_MULTI:
            return yh_lat_pack_multi(p[0], i[0], st);
        case YH_OP_CONV_S2_FWD:             /* slots of YH_OP_CONV_FWD */
            return yh_conv_s2_fwd_act((const float *)p[0], i[0], (const float *)p[10], i[17], (const float *)p[1], i[1], (const float *)p[2],
                                      (float *)p[3], i[2], (float *)p[4], i[3], i[4], i[5], i[6], i[7], st);
        default:
            yh_set_error("yh_run: unknown op kind %d", o.kind);
            return YH_E_BADARG;
    }
}

// ---- per-(device, kernel) dynamic-LDS opt-in ----------------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device and must precede the first launch that needs more than
// the default.  The launchers call this helper instead of keeping function-local "already set" flags: one mutex-guarded
// table keyed by (device, kernel), so two host threads or two devices cannot race or skip each other's opt-in.
namespace {
struct SmemKey {
    int dev;
    const void *fn;
    size_t bytes;
};
std::mutex g_smem_mu;
std::vector<SmemKey> g_smem;
}  // namespace

int yh_ensure_dyn_smem(const void *fn, size_t bytes) {
    int dev = 0;
    YH_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_smem_mu);
    for (SmemKey &k : g_smem)
        if (k.dev == dev && k.fn == fn) {
            if (k.bytes >= bytes) return 0;
            YH_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            k.bytes = bytes;
            return 0;
        }
    YH_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    g_smem.push_back(SmemKey{dev, fn, bytes});
    return 0;
}

// ---- the library's environment switches: read ONCE per process, never on a launch path --------------------------------
// YH_BF16_STREAM=0   route the stride-1 bf16 layers through the segment kernels of conv_bf16.hip instead of the flat-stream
//                    kernels of conv_bf16_stream.hip (A/B switch)
// (YH_OVERLAP is per context: yh_create.)  Everything else that used to be an environment knob is a constant now; diagnostic
// builds (in-kernel stamps, tile sweeps) are compile-time: make EXTRA=-DYH_..._STAMPS / -DYH_WGS_TUNE.
// YH_PW_X6=1        (experimental) the forward 1x1 convolutions it supports on the split-bf16 form of the fp32 GEMM (conv_pw.hip)
static std::once_flag g_env_once;
static int g_bf16_stream = 1, g_pw_x6 = 0;
static void read_env_once() {
    std::call_once(g_env_once, [] {
        const char *e = getenv("YH_BF16_STREAM");
        g_bf16_stream = (e && e[0] == '0') ? 0 : 1;
        e = getenv("YH_PW_X6");
        g_pw_x6 = (e && e[0] == '1') ? 1 : 0;
    });
}
int yh_env_bf16_stream() {
    read_env_once();
    return g_bf16_stream;
}
int yh_env_pw_x6() {
    read_env_once();
    return g_pw_x6;
}

// ---- execution context ("no global mutable state besides the explicit handle") ---------------------------
// Weight-gradient work (backward-weight GEMMs, bias column sums) depends only on dY and on saved activations, never on
// the rest of the backward chain, and nothing but the optimiser reads its results.  yh_run therefore forks those ops
// onto the context's side stream (event fork after the op that produced dY, event join before returning control): the
// MFMA-bound weight gradients overlap the HBM-bound BatchNorm backward passes and fill the tail rounds of the
// backward-data GEMMs.  The side stream and its two events belong to ONE yh_context, bound to the device that was
// current at its first forked run; a context serves one host thread at a time, distinct contexts are independent.
struct yh_context {
    int device = -1;            // bound lazily, at the first run that needs the side lane
    int overlap = 1;            // YH_OVERLAP=0 (read ONCE, at yh_create) or yh_context_set_overlap(ctx, 0): run every op in list order
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};

extern "C" int yh_create(yh_context **out) {
    YH_REQUIRE(out, "yh_create: null output");
    *out = new (std::nothrow) yh_context();
    YH_REQUIRE(*out, "yh_create: out of memory");
    const char *e = getenv("YH_OVERLAP");           // the library's environment switches are read here and in yh_env_once()
    (*out)->overlap = (e && e[0] == '0') ? 0 : 1;
    return 0;
}

extern "C" int yh_destroy(yh_context *ctx) {
    if (!ctx) return 0;
    int rc = 0;
    if (ctx->side) {
        int cur = 0;
        bool sw = hipGetDevice(&cur) == hipSuccess && cur != ctx->device && hipSetDevice(ctx->device) == hipSuccess;
        if (hipStreamSynchronize(ctx->side) != hipSuccess) rc = YH_E_BADARG;
        (void)hipEventDestroy(ctx->fork);
        (void)hipEventDestroy(ctx->join);
        (void)hipStreamDestroy(ctx->side);
        if (sw) (void)hipSetDevice(cur);
    }
    delete ctx;
    return rc;
}

extern "C" int yh_context_set_overlap(yh_context *ctx, int enable) {
    YH_REQUIRE(ctx, "yh_context_set_overlap: null context");
    ctx->overlap = enable ? 1 : 0;
    return 0;
}

extern "C" int yh_context_info(const yh_context *ctx, int *device, int *overlap, void **side_stream, void **fork_event,
                               void **join_event) {
    YH_REQUIRE(ctx, "yh_context_info: null context");
    if (device) *device = ctx->device;
    if (overlap) *overlap = ctx->overlap;
    if (side_stream) *side_stream = (void *)ctx->side;
    if (fork_event) *fork_event = (void *)ctx->fork;
    if (join_event) *join_event = (void *)ctx->join;
    return 0;
}

// 1 = the side lane is usable for this run, 0 = run everything in list order on the caller's stream, < 0 = error
static int side_ready(yh_context *ctx) {
    if (!ctx) return 0;
    if (!ctx->overlap) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (!ctx->side) {
        // (an ordinary stream: a low-priority side lane measured 22.21 -> 22.12 ms/step in training -- noise -- and a captured
        // hipGraph with mixed-priority nodes replays 2x slower, 2.4 -> 4.8 ms at bs=1)
        hipError_t e = hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking);
        if (e != hipSuccess || hipEventCreateWithFlags(&ctx->fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->join, hipEventDisableTiming) != hipSuccess) {
            if (ctx->fork) (void)hipEventDestroy(ctx->fork);
            if (ctx->side) (void)hipStreamDestroy(ctx->side);
            ctx->side = nullptr;
            ctx->fork = ctx->join = nullptr;
            ctx->overlap = 0;
            return 0;
        }
        ctx->device = dev;
    }
    if (dev != ctx->device) {
        yh_set_error("yh_run: context is bound to device %d but device %d is current (one context per device)", ctx->device, dev);
        return YH_E_BADARG;
    }
    return 1;
}

thread_local int yh_tls_side_lane = 0;

extern "C" int yh_run(yh_context *ctx, const yh_op *ops, int n, void *stream, int *failed) {
    YH_REQUIRE(ops || n == 0, "yh_run: null op list");
    hipStream_t mainst = (hipStream_t)stream;
    bool forked = false;
    int rc = 0, fail_at = -1;
    auto hip = [&](hipError_t e, const char *what) {           // a failing runtime call on the lane plumbing
        if (e == hipSuccess) return 0;
        yh_set_error("yh_run: %s: %s", what, hipGetErrorString(e));
        return (int)e;
    };
    for (int k = 0; k < n; ++k) {
        const int kind = ops[k].kind;
        const bool sync_op = kind == YH_OP_FORK || kind == YH_OP_JOIN;
        const bool auto_side = (kind == YH_OP_CONV_BWD_WEIGHT || kind == YH_OP_CONV_WINO_BWD_WEIGHT ||
                                kind == YH_OP_CONV_PW_BWD_WEIGHT || kind == YH_OP_CONV_NARROW_BWD_WEIGHT || kind == YH_OP_BF16_CONV_NARROW_BWD_WEIGHT || kind == YH_OP_COLSUM || kind == YH_OP_BF16_CONV_BWD_WEIGHT ||
                                kind == YH_OP_BF16_COLSUM);
        int sr = 0;
        if (sync_op || auto_side || ops[k].lane == 1) sr = side_ready(ctx);
        if (sr < 0) {
            rc = sr;
        } else if (sync_op) {                                  // explicit lane synchronisation points
            if (!sr) continue;                                 // overlap disabled: everything runs in list order on `stream`
            if (kind == YH_OP_FORK) {                          // side lane may proceed past everything enqueued on main so far
                rc = hip(hipEventRecord(ctx->fork, mainst), "fork record");
                if (!rc) rc = hip(hipStreamWaitEvent(ctx->side, ctx->fork, 0), "fork wait");
                forked = true;
            } else if (forked) {                               // main waits for everything enqueued on the side lane so far
                rc = hip(hipEventRecord(ctx->join, ctx->side), "join record");
                if (!rc) rc = hip(hipStreamWaitEvent(mainst, ctx->join, 0), "join wait");
                forked = false;
            }
        } else if (sr == 1) {
            if (auto_side) {     // weight-gradient work: fork right here, after the op that produced dY
                rc = hip(hipEventRecord(ctx->fork, mainst), "fork record");
                if (!rc) rc = hip(hipStreamWaitEvent(ctx->side, ctx->fork, 0), "fork wait");
            }
            forked = true;
            if (!rc) {
                yh_tls_side_lane = 1;                          // see common.h: the launchers may size for a shared CU
                rc = run_one(ops[k], (void *)ctx->side);
                yh_tls_side_lane = 0;
            }
        } else {
            rc = run_one(ops[k], stream);
        }
        if (rc) {
            fail_at = k;
            break;
        }
    }
    if (rc && failed) *failed = fail_at;
    // join the side lane on success AND on failure: the caller's stream order must cover everything this call launched
    if (forked && ctx && ctx->side) {
        hipError_t e = hipEventRecord(ctx->join, ctx->side);
        if (e == hipSuccess) e = hipStreamWaitEvent(mainst, ctx->join, 0);
        if (e != hipSuccess && !rc) rc = hip(e, "joining the side lane");
    }
    return rc;
}
