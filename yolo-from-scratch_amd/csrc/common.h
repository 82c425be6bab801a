// Shared helpers for libyolohip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/yolohip.h"

void yh_set_error(const char *fmt, ...);
// opt a kernel into `bytes` of dynamic LDS on the current device (api.hip: mutex-guarded (device, kernel) table)
int yh_ensure_dyn_smem(const void *fn, size_t bytes);
// environment switches of the library, read once per process (api.hip)
int yh_env_bf16_stream();
int yh_env_pw_x6();
// Set by yh_run around an op it launches on the context's side lane (the weight-gradient lane), 0 otherwise and for every direct call.
// The split-K weight-gradient launchers of conv_pw.hip and conv_wgrad.hip read it: next to the main lane they reserve YH_SIDE_LDS_BYTES
// of LDS per workgroup, i.e. ONE of their workgroups per CU, and leave the other half of the CU (registers, 76 KB of LDS) to the main
// lane's kernels -- measured 17.06 -> 16.64 ms per training step (bs 64, fp32); alone they are 5 % slower that way, so direct calls and
// YH_OVERLAP=0 runs keep their natural occupancy.  Sweep: 60 KB (two per CU) 16.91, 81-92 KB 16.65-16.71, 100 KB 16.76, 128 KB 16.97.
extern thread_local int yh_tls_side_lane;
constexpr size_t YH_SIDE_LDS_BYTES = 84 * 1024;

// Pointers that are SELECTED at run time (tensor A or tensor B, the input or a zero page, a table entry) lose their address
// space and compile to FLAT loads / stores.  A flat access counts on vmcnt AND lgkmcnt and may retire out of order, so every
// later wait becomes s_waitcnt vmcnt(0) lgkmcnt(0): a software prefetch issued before it is waited for in full (the Winograd
// loop ran 1.8x its MFMA time that way).  Everything the kernels touch through such pointers is global memory: say so.
#define YH_GLOBAL __attribute__((address_space(1)))
typedef YH_GLOBAL float gfloat;
template <typename T>
__device__ __forceinline__ YH_GLOBAL T *yh_global(T *p) { return (YH_GLOBAL T *)p; }
template <typename T>
__device__ __forceinline__ const YH_GLOBAL T *yh_global(const T *p) { return (const YH_GLOBAL T *)p; }

#define YH_REQUIRE(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) {                                         \
            yh_set_error(__VA_ARGS__);                         \
            return YH_E_BADARG;                                \
        }                                                      \
    } while (0)

// Check the launch that was just enqueued (no synchronisation).
#define YH_CHECK_LAUNCH(name)                                                        \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            yh_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return (int)e__;                                                         \
        }                                                                            \
    } while (0)

#define YH_HIP(call)                                                                 \
    do {                                                                             \
        hipError_t e__ = (call);                                                     \
        if (e__ != hipSuccess) {                                                     \
            yh_set_error("%s: %s", #call, hipGetErrorString(e__));                   \
            return (int)e__;                                                         \
        }                                                                            \
    } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// accurate exp (not __expf): parity with the CPU reference is 1e-4 relative on losses
__device__ __forceinline__ float yh_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// hardware exp2 + reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each, 4 instructions instead of ~30): for the BatchNorm+SiLU
// passes, which are otherwise VALU-bound on the sigmoid (0.9 ms of pure VALU time in the backward reduce at bs=64).
// Saturates correctly (exp -> inf gives 0, exp -> 0 gives 1); the loss and the detection threshold keep yh_sigmoid.
__device__ __forceinline__ float yh_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Input prologue of the convolution kernels (round 4): the PRODUCER's BatchNorm + SiLU applied by the consumer while it stages its
// operand, so the normalised activation never exists in memory.  Per input channel a table row [scale | shift | gate]:
// z = x * scale + shift, then silu(z) where gate != 0 (materialised channels of a concatenation carry scale 1, shift 0, gate 0).
// Same expression, same hardware exp / rcp as bn_silu_fwd_kernel: a fused consumer sees the values the separate pass would have stored.
__device__ __forceinline__ float yh_prologue(float x, float scale, float shift, float gate) {
    const float z = x * scale + shift;
    return gate != 0.f ? z * yh_sigmoid_fast(z) : z;
}

// 64-lane butterfly sum; every lane ends with the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
