// Global-norm gradient clipping + Adam over flat fp32 buffers (HBM-bound: 16 B read + 12 B written
// per parameter).  The squared norm is a two-stage fp64 reduction with a fixed order; the clip
// coefficient is recomputed by every thread from the device-resident norm, so there is no host
// round trip between clip_grad_norm_ and optimizer.step() (train.py:916-918).
#include "common.h"

namespace {

constexpr int kNormBlocks = 1024;

__global__ void sqnorm_stage1(const float *__restrict__ g, int64_t n, double *__restrict__ part) {
    __shared__ double red[4];
    double s = 0.0;
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = *(const f32x4 *)(g + 4 * i);
        s += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        float v = g[(n4 << 2) + threadIdx.x];
        s += (double)v * v;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void sqnorm_stage2(const double *__restrict__ part, int nblk, double scale, float *__restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) s += part[b];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)(scale * sqrt(red[0] + red[1] + red[2] + red[3]));
}

__global__ void adam_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                            int64_t n, float omb1, float b2, float omb2, float eps, float step_size, float bc2_sqrt,
                            float max_norm, const float *__restrict__ norm, float gscale) {
    float coef = 1.f;
    const bool clip = norm != nullptr && max_norm > 0.f;
    if (clip) {
        coef = max_norm / (norm[0] + 1e-6f);
        if (coef > 1.f) coef = 1.f;
    }
    const bool write_g = clip || gscale != 1.f;
    coef *= gscale;
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 gv = *(const f32x4 *)(g + 4 * i), mv = *(const f32x4 *)(m + 4 * i);
        f32x4 vv = *(const f32x4 *)(v + 4 * i), pv = *(const f32x4 *)(p + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float ge = gv[e] * coef;
            gv[e] = ge;
            mv[e] = mv[e] + (ge - mv[e]) * omb1;                       // lerp_(grad, 1-beta1)
            vv[e] = vv[e] * b2 + omb2 * ge * ge;
            float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
            pv[e] = pv[e] - step_size * (mv[e] / denom);
        }
        if (write_g) *(f32x4 *)(g + 4 * i) = gv;
        *(f32x4 *)(m + 4 * i) = mv;
        *(f32x4 *)(v + 4 * i) = vv;
        *(f32x4 *)(p + 4 * i) = pv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        int64_t i = (n4 << 2) + threadIdx.x;
        float ge = g[i] * coef;
        if (write_g) g[i] = ge;
        float me = m[i] + (ge - m[i]) * omb1;
        float ve = v[i] * b2 + omb2 * ge * ge;
        m[i] = me; v[i] = ve;
        p[i] = p[i] - step_size * (me / (sqrtf(ve) / bc2_sqrt + eps));
    }
}

}  // namespace

extern "C" int64_t yh_sqnorm_ws(int64_t n) { (void)n; return kNormBlocks; }

extern "C" int yh_grad_sqnorm(const float *g, int64_t n, float grad_scale, float *norm_out, double *ws, void *stream) {
    YH_REQUIRE(g && norm_out && ws && n > 0, "grad_sqnorm: bad argument");
    YH_REQUIRE(((uintptr_t)g & 15) == 0, "grad_sqnorm: buffer must be 16-byte aligned");
    int64_t want = cdiv64(n / 4 + 1, 256);
    int nblk = (int)(want > kNormBlocks ? kNormBlocks : want);
    hipLaunchKernelGGL(sqnorm_stage1, dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, n, ws);
    YH_CHECK_LAUNCH("sqnorm_stage1");
    hipLaunchKernelGGL(sqnorm_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, nblk, (double)grad_scale, norm_out);
    YH_CHECK_LAUNCH("sqnorm_stage2");
    return 0;
}

extern "C" int yh_adam_step(float *p, float *g, float *m, float *v, int64_t n, double lr, double beta1, double beta2,
                            double eps, int step, float max_norm, const float *norm, float grad_scale, void *stream) {
    YH_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad argument");
    YH_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: buffers must be 16-byte aligned");
    double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    int64_t want = cdiv64(n / 4 + 1, 256);
    int nblk = (int)(want > 2048 ? 2048 : want);
    hipLaunchKernelGGL(adam_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)eps, step_size, bc2_sqrt, max_norm, norm, grad_scale);
    YH_CHECK_LAUNCH("adam");
    return 0;
}
