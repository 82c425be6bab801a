// Error reporting, version and the op-list executor of libyolohip.
#include "common.h"
#include <string.h>
#include <stdlib.h>

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
            return yh_conv_fwd((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2], (float *)p[3],
                               i[2], (float *)p[4], i[3], i[4], i[5], i[6], i[7], i[8], i[9], st);
        case YH_OP_CONV_BWD_DATA:
            return yh_conv_bwd_data((const float *)p[0], i[0], (const float *)p[1], i[1], (float *)p[2], i[2], i[3], i[4],
                                    i[5], i[6], i[7], i[8], i[9], i[10], st);
        case YH_OP_CONV_BWD_WEIGHT:
            return yh_conv_bwd_weight((const float *)p[0], i[0], (const float *)p[1], i[1], (float *)p[2], (float *)p[3],
                                      o.l[0], i[2], i[3], i[4], i[5], i[6], i[7], i[8], i[9], st);
        case YH_OP_COLSUM:
            return yh_colsum((const float *)p[0], i[0], o.l[0], i[1], (float *)p[1], (float *)p[2], st);
        case YH_OP_BN_FINALIZE:
            return yh_bn_finalize((const float *)p[0], i[0], o.l[0], (const float *)p[1], (const float *)p[2],
                                  (float *)p[3], (float *)p[4], f[0], f[1], (float *)p[5], i[1], (int64_t *)p[6], st);
        case YH_OP_BN_EVAL_COEF:
            return yh_bn_eval_coef((const float *)p[0], (const float *)p[1], (const float *)p[2], (const float *)p[3], f[0],
                                   (float *)p[4], i[0], st);
        case YH_OP_BN_SILU_FWD:
            return yh_bn_silu_fwd((const float *)p[0], i[0], (const float *)p[1], (const float *)p[2], i[1], (float *)p[3],
                                  i[2], o.l[0], i[3], i[4], i[5], i[6], st);
        case YH_OP_BN_SILU_BWD_REDUCE:
            return yh_bn_silu_bwd_reduce((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2],
                                         (float *)p[3], o.l[0], i[2], i[3], i[4], i[5], st);
        case YH_OP_BN_SILU_BWD_APPLY:
            return yh_bn_silu_bwd_apply((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2],
                                        (const float *)p[3], i[2], (const float *)p[4], (float *)p[5], (float *)p[6],
                                        (float *)p[7], i[3], (float *)p[8], i[4], i[5], o.l[0], i[6], i[7], i[8], i[9], st);
        case YH_OP_MAXPOOL5_FWD:
            return yh_maxpool5_fwd((const float *)p[0], i[0], (float *)p[1], i[1], (uint8_t *)p[2], i[2], i[3], i[4], i[5], st);
        case YH_OP_MAXPOOL5_BWD:
            return yh_maxpool5_bwd((const float *)p[0], i[0], (const uint8_t *)p[1], (float *)p[2], i[1], i[2], i[3], i[4],
                                   i[5], st);
        case YH_OP_MEMSET:
            return yh_memset(p[0], i[0], o.l[0], st);
        case YH_OP_ADD_INT64:
            return yh_add_int64((int64_t *)p[0], o.l[0], st);
        case YH_OP_WINO_WEIGHTS_MULTI:
            return yh_wino_weights_multi(p[0], i[0], st);
        case YH_OP_CONV_WINO_FWD:
            return yh_conv_wino_fwd((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2], (float *)p[3],
                                    i[2], (float *)p[4], i[3], i[4], i[5], i[6], i[7], st);
        case YH_OP_CONV_WINO_BWD_DATA:      /* p[3] / i[9]: optional BatchNorm-backward table */
            return yh_conv_wino_bwd_data_bn((const float *)p[0], i[0], (const float *)p[1], i[1], (float *)p[2], i[2], i[3], i[4],
                                            i[5], i[6], i[7], i[8], p[3], i[9], st);
        case YH_OP_CONV_WINO_BWD_WEIGHT:
            return yh_conv_wino_bwd_weight((const float *)p[0], i[0], (const float *)p[1], i[1], (float *)p[2], (float *)p[3],
                                           o.l[0], i[2], i[3], i[4], i[5], i[7], st);
        case YH_OP_CONV_PW_BWD_WEIGHT:      /* same argument slots as YH_OP_CONV_BWD_WEIGHT */
            return yh_conv_pw_bwd_weight((const float *)p[0], i[0], (const float *)p[1], i[1], (float *)p[2], (float *)p[3],
                                         o.l[0], (int64_t)i[2] * i[3] * i[4], i[5], i[7], st);
        case YH_OP_CONV_STEM_FWD:           /* same argument slots as YH_OP_CONV_FWD */
            return yh_conv_stem_fwd((const float *)p[0], (const float *)p[1], (const float *)p[2], (float *)p[3], i[2], (float *)p[4],
                                    i[3], i[4], i[5], st);
        case YH_OP_PACK_WEIGHTS_S2M:
            return yh_pack_weights_s2m((const float *)p[0], (float *)p[1], i[0], i[1], i[2], st);
        case YH_OP_CONV_BWD_DATA_S2M:       /* same argument slots as YH_OP_CONV_BWD_DATA */
            return yh_conv_bwd_data_s2m((const float *)p[0], i[0], (const float *)p[1], i[1], (float *)p[2], i[2], i[3], i[4], i[5],
                                        i[6], i[7], i[10], st);
        case YH_OP_CONV_PW_FWD2:            /* p: x, wq, bias1, y1, part1, bias2, y2, part2;  i: ldx, ldw, ldy1, B, H, W, Cin, cout1, ldy2, cout2 */
            return yh_conv_pw_fwd2((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2], (float *)p[3], i[2],
                                   (float *)p[4], i[7], (const float *)p[5], (float *)p[6], i[8], (float *)p[7], i[9],
                                   (int64_t)i[3] * i[4] * i[5], i[6], st);
        case YH_OP_PW_PACK_MULTI:
            return yh_pw_pack_multi(p[0], i[0], st);
        case YH_OP_CONV_PW_FWD:             /* same argument slots as YH_OP_CONV_FWD */
            return yh_conv_pw_fwd((const float *)p[0], i[0], (const float *)p[1], i[1], (const float *)p[2], (float *)p[3], i[2],
                                  (float *)p[4], (int64_t)i[3] * i[4] * i[5], i[6], i[7], st);
        case YH_OP_CONV_PW_BWD_DATA:        /* p: dy1, dy2 | NULL, wq, dx, bn table | NULL;  i: cout1, cout2, lddy, ldw, lddx, B, H, W, Cin, accumulate, n_bn */
            return yh_conv_pw_bwd_data_bn((const float *)p[0], i[0], (const float *)p[1], i[1], i[2], (const float *)p[2], i[3],
                                          (float *)p[3], i[4], (int64_t)i[5] * i[6] * i[7], i[8], i[9], p[4], i[10], st);
        case YH_OP_NOP:
            return 0;
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
        default:
            yh_set_error("yh_run: unknown op kind %d", o.kind);
            return YH_E_BADARG;
    }
}

// Weight-gradient work (backward-weight GEMMs, bias column sums) depends only on dY and on saved
// activations, never on the rest of the backward chain, and nothing but the optimiser reads its
// results.  yh_run therefore forks those ops onto a side stream (event fork after the op that produced
// dY, event join before returning control): the MFMA-bound weight gradients overlap the HBM-bound
// BatchNorm backward passes and fill the tail rounds of the backward-data GEMMs.
static int g_overlap = -1;
static hipStream_t g_side = nullptr;
static hipEvent_t g_fork = nullptr, g_join = nullptr;

extern "C" int yh_set_overlap(int enable) {
    g_overlap = enable ? 1 : 0;
    return 0;
}

static int side_ready() {
    if (g_overlap < 0) {
        const char *e = getenv("YH_OVERLAP");
        g_overlap = (e && e[0] == '0') ? 0 : 1;
    }
    if (!g_overlap) return 0;
    if (!g_side) {
        // YH_SIDE_PRIORITY=low|high: experiment knob.  A low-priority side lane measured 22.21 -> 22.12 ms/step in training
        // (noise level) but a captured hipGraph with mixed-priority nodes replays 2x slower (2.4 -> 4.8 ms at bs=1), so the
        // default is an ordinary stream.
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);           // lo = numerically greatest = lowest priority
        const char *pr = getenv("YH_SIDE_PRIORITY");
        hipError_t e = (pr && (pr[0] == 'l' || pr[0] == 'h'))
                           ? hipStreamCreateWithPriority(&g_side, hipStreamNonBlocking, pr[0] == 'l' ? lo : hi)
                           : hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking);
        if (e != hipSuccess ||
            hipEventCreateWithFlags(&g_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g_join, hipEventDisableTiming) != hipSuccess) {
            g_side = nullptr;
            g_overlap = 0;
            return 0;
        }
    }
    return 1;
}

extern "C" int yh_run(const yh_op *ops, int n, void *stream, int *failed) {
    YH_REQUIRE(ops || n == 0, "yh_run: null op list");
    hipStream_t mainst = (hipStream_t)stream;
    bool forked = false;
    for (int k = 0; k < n; ++k) {
        const int kind = ops[k].kind;
        if (kind == YH_OP_FORK || kind == YH_OP_JOIN) {      // explicit lane synchronisation points
            if (!side_ready()) continue;                       // overlap disabled: everything runs in list order on `stream`
            if (kind == YH_OP_FORK) {                          // side lane may proceed past everything enqueued on main so far
                YH_HIP(hipEventRecord(g_fork, mainst));
                YH_HIP(hipStreamWaitEvent(g_side, g_fork, 0));
                forked = true;
            } else if (forked) {                               // main waits for everything enqueued on the side lane so far
                YH_HIP(hipEventRecord(g_join, g_side));
                YH_HIP(hipStreamWaitEvent(mainst, g_join, 0));
                forked = false;
            }
            continue;
        }
        const bool auto_side = (kind == YH_OP_CONV_BWD_WEIGHT || kind == YH_OP_CONV_WINO_BWD_WEIGHT || kind == YH_OP_CONV_PW_BWD_WEIGHT || kind == YH_OP_COLSUM);
        const bool side = (auto_side || ops[k].lane == 1) && side_ready();
        int rc;
        if (side) {
            if (auto_side) {     // weight-gradient work: fork right here, after the op that produced dY
                YH_HIP(hipEventRecord(g_fork, mainst));
                YH_HIP(hipStreamWaitEvent(g_side, g_fork, 0));
            }
            rc = run_one(ops[k], (void *)g_side);
            forked = true;
        } else {
            rc = run_one(ops[k], stream);
        }
        if (rc) {
            if (failed) *failed = k;
            return rc;
        }
    }
    if (forked) {
        YH_HIP(hipEventRecord(g_join, g_side));
        YH_HIP(hipStreamWaitEvent(mainst, g_join, 0));
    }
    return 0;
}
