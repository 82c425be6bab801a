// HBM-bound kernels of the hot path: layout changes, weight packing, BatchNorm(+SiLU) forward and
// backward, SPPF max-pool, column sums.  All are float4-vectorised over the channel axis (NHWC, so
// the channel axis is the contiguous one) and use grid-stride loops capped at a few thousand
// workgroups; every cross-workgroup reduction is two-stage with a fixed summation order, so results
// are bitwise reproducible run to run.
#include "common.h"
#include <initializer_list>
#include <stdlib.h>

namespace {

constexpr int kMaxBlocks = 2048;

// element access for the two activation types of the path: fp32 (16-byte float4) and bf16 (8-byte groups of four,
// widened to fp32 in registers: every kernel below computes in fp32 whatever the storage type)
typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld4(const float *p) { return *(const f32x4 *)p; }
__device__ __forceinline__ f32x4 ld4(const bf16 *p) { return __builtin_convertvector(*(const bf16x4 *)p, f32x4); }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *(f32x4 *)p = v; }
__device__ __forceinline__ void st4(bf16 *p, f32x4 v) { *(bf16x4 *)p = __builtin_convertvector(v, bf16x4); }
__device__ __forceinline__ float ld1(const float *p) { return *p; }
__device__ __forceinline__ float ld1(const bf16 *p) { return (float)*p; }
__device__ __forceinline__ void st1(float *p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16 *p, float v) { *p = (bf16)v; }

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float *__restrict__ src, T *__restrict__ dst, int C, int HW, int ld,
                                    int cpad, int64_t npix) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        int64_t b = p / HW, hw = p - b * HW;
        const float *s = src + b * C * (int64_t)HW + hw;
        T *d = dst + p * ld;
        if (C <= 4 && (cpad & 3) == 0 && (ld & 3) == 0 && (((uintptr_t)dst) & 15) == 0) {
            // image batches (3 channels padded to 4 or 8): whole 4-channel groups per store -- element-wise bf16 stores are
            // 2-byte writes that the memory side turns into read-modify-write of the line
            f32x4 v = {s[0], C > 1 ? s[(int64_t)HW] : 0.f, C > 2 ? s[(int64_t)2 * HW] : 0.f, C > 3 ? s[(int64_t)3 * HW] : 0.f};
            st4(d, v);
            for (int c = 4; c < cpad; c += 4) st4(d + c, f32x4{0.f, 0.f, 0.f, 0.f});
            continue;
        }
        for (int c = 0; c < cpad; ++c) st1(d + c, c < C ? s[(int64_t)c * HW] : 0.f);
    }
}

// uint8 HWC image batch -> NHWC fp32 in [0,1] (channel-padded): the reference's `from_numpy(...).float() / 255.0`
// (train.py:115-117) evaluated on the device -- a true division, so the values are bit-identical -- 4x less PCIe traffic.
template <typename T>
__global__ void u8hwc_to_nhwc_kernel(const uint8_t *__restrict__ src, T *__restrict__ dst, int C, int ld, int cpad,
                                     int64_t npix) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const uint8_t *s = src + p * C;
        T *d = dst + p * ld;
        for (int c = 0; c < cpad; ++c) st1(d + c, c < C ? (float)s[c] / 255.0f : 0.f);
    }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T *__restrict__ src, float *__restrict__ dst, int C, int HW, int ld,
                                    int accumulate, int64_t total) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t hw = i % HW, q = i / HW;
        int c = (int)(q % C);
        int64_t b = q / C;
        float v = ld1(src + (b * HW + hw) * ld + c);
        dst[i] = accumulate ? dst[i] + v : v;
    }
}

__global__ void pack_weights_kernel(const float *__restrict__ w, float *__restrict__ wf, float *__restrict__ wb,
                                    int Cout, int Cin, int kk, int cin_pad, int ldwf, int ldwb) {
    const int nf = wf ? kk * cin_pad * ldwf : 0;
    const int nb = wb ? kk * Cout * ldwb : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf + nb; i += gridDim.x * blockDim.x) {
        if (i < nf) {
            int n = i % ldwf, q = i / ldwf;
            int ci = q % cin_pad, t = q / cin_pad;
            wf[i] = (n < Cout && ci < Cin) ? w[((size_t)n * Cin + ci) * kk + t] : 0.f;
        } else {
            int j = i - nf;
            int ci = j % ldwb, q = j / ldwb;
            int co = q % Cout, t = q / Cout;
            wb[j] = (ci < Cin) ? w[((size_t)co * Cin + ci) * kk + t] : 0.f;
        }
    }
}

// all layers of a plan in one launch: blockIdx.y = layer, descriptor table in device memory
struct PackDesc {
    const float *w;
    float *wf, *wb;
    int Cout, Cin, kk, cin_pad, ldwf, ldwb, pad0, pad1;
};
__global__ void pack_weights_multi_kernel(const PackDesc *__restrict__ tab) {
    const PackDesc d = tab[blockIdx.y];
    const int nf = d.wf ? d.kk * d.cin_pad * d.ldwf : 0;
    const int nb = d.wb ? d.kk * d.Cout * d.ldwb : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf + nb; i += gridDim.x * blockDim.x) {
        if (i < nf) {
            int n = i % d.ldwf, q = i / d.ldwf;
            int ci = q % d.cin_pad, t = q / d.cin_pad;
            d.wf[i] = (n < d.Cout && ci < d.Cin) ? d.w[((size_t)n * d.Cin + ci) * d.kk + t] : 0.f;
        } else {
            int j = i - nf;
            int ci = j % d.ldwb, q = j / d.ldwb;
            int co = q % d.Cout, t = q / d.Cout;
            d.wb[j] = (ci < d.Cin) ? d.w[((size_t)co * d.Cin + ci) * d.kk + t] : 0.f;
        }
    }
}

// Inference: fold eval-mode BatchNorm into the packed forward weights and a per-channel bias,
//   w'[.., n] = w[n, ..] * gamma[n] / sqrt(rv[n] + eps),   b'[n] = (b[n] - rm[n]) * that + beta[n].
struct FoldDesc {
    const float *w, *bias_in, *gamma, *beta, *rmean, *rvar;
    float *wf, *bias_out;
    int Cout, Cin, kk, cin_pad, ldwf;
    float eps;
};
__global__ void pack_fold_multi_kernel(const FoldDesc *__restrict__ tab) {
    const FoldDesc d = tab[blockIdx.y];
    const int nf = d.kk * d.cin_pad * d.ldwf;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += gridDim.x * blockDim.x) {
        int n = i % d.ldwf, q = i / d.ldwf;
        int ci = q % d.cin_pad, t = q / d.cin_pad;
        float v = 0.f;
        if (n < d.Cout && ci < d.Cin) {
            float sc = d.gamma ? d.gamma[n] * (1.0f / sqrtf(d.rvar[n] + d.eps)) : 1.f;
            v = d.w[((size_t)n * d.Cin + ci) * d.kk + t] * sc;
        }
        d.wf[i] = v;
    }
    if (blockIdx.x == 0 && d.bias_out)
        for (int n = threadIdx.x; n < d.Cout; n += blockDim.x) {
            float b = d.bias_in ? d.bias_in[n] : 0.f;
            if (d.gamma) {
                float sc = d.gamma[n] * (1.0f / sqrtf(d.rvar[n] + d.eps));
                b = (b - d.rmean[n]) * sc + d.beta[n];
            }
            d.bias_out[n] = b;
        }
}

// Inference: eval-mode BatchNorm folded into OIHW weights (the input of every per-kernel weight transform) and a bias,
//   w'[n][...] = w[n][...] * gamma[n] / sqrt(rv[n] + eps),   b'[n] = (b[n] - rm[n]) * that + beta[n];  gamma == NULL: copy.
struct FoldOihwDesc {
    const float *w, *bias_in, *gamma, *beta, *rmean, *rvar;
    float *w_out, *bias_out;
    int Cout, per;          // per = Cin * k * k elements per output channel
    float eps;
    int pad;
};
__global__ void fold_oihw_multi_kernel(const FoldOihwDesc *__restrict__ tab) {
    const FoldOihwDesc d = tab[blockIdx.y];
    const int total = d.Cout * d.per;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int n = i / d.per;
        const float sc = d.gamma ? d.gamma[n] * (1.0f / sqrtf(d.rvar[n] + d.eps)) : 1.f;
        d.w_out[i] = d.w[i] * sc;
    }
    if (blockIdx.x == 0 && d.bias_out)
        for (int n = threadIdx.x; n < d.Cout; n += blockDim.x) {
            float b = d.bias_in ? d.bias_in[n] : 0.f;
            if (d.gamma) {
                float sc = d.gamma[n] * (1.0f / sqrtf(d.rvar[n] + d.eps));
                b = (b - d.rmean[n]) * sc + d.beta[n];
            }
            d.bias_out[n] = b;
        }
}

// ---------------------------------------------------------------------------------------------
// column sums: stage 1 -> partial[blk][C], stage 2 -> out[C]
// vector path (C % 4 == 0, 16-byte addressable rows): float4 columns x row groups, like the BN reductions
template <typename T>
__global__ void colsum_stage1_vec(const T *__restrict__ x, int ldx, int64_t M, int C, float *__restrict__ part,
                                  int64_t rows_per_blk) {
    __shared__ float red[256 * 4];
    const int t = threadIdx.x, cq = C >> 2;
    const int rg = 256 / cq, c4 = t % cq, r_in = t / cq;
    int64_t r0 = blockIdx.x * rows_per_blk, r1 = r0 + rows_per_blk;
    if (r1 > M) r1 = M;
    f32x4 s = {0, 0, 0, 0};
    if (r_in < rg) {
        // four independent rows per trip: one load in flight per thread left the pass latency-bound (1.5-1.9 TB/s); the
        // order of the additions is fixed, so the sums stay reproducible
        f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0}, s3 = {0, 0, 0, 0};
        int64_t r = r0 + r_in;
        const T *px = x + 4 * c4;
        for (; r + 3 * rg < r1; r += 4 * rg) {
            const f32x4 v0 = ld4(px + r * ldx), v1 = ld4(px + (r + rg) * ldx);
            const f32x4 v2 = ld4(px + (r + 2 * rg) * ldx), v3 = ld4(px + (r + 3 * rg) * ldx);
            s += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; r < r1; r += rg) s += ld4(px + r * ldx);
        s = (s + s1) + (s2 + s3);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[t * 4 + e] = s[e];
    __syncthreads();
    if (t < cq) {
        f32x4 a = {0, 0, 0, 0};
        for (int k = 0; k < rg; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += red[(k * cq + t) * 4 + e];
        *(f32x4 *)(part + (size_t)blockIdx.x * C + 4 * t) = a;
    }
}
template <typename T>
__global__ void colsum_stage1(const T *__restrict__ x, int ldx, int64_t M, int C, float *__restrict__ part,
                              int64_t rows_per_blk) {
    __shared__ float red[256];
    const int t = threadIdx.x;
    int cw = 1;
    while (cw < C && cw < 256) cw <<= 1;
    const int rg = 256 / cw, c_in = t % cw, r_in = t / cw;
    int64_t r0 = blockIdx.x * rows_per_blk, r1 = r0 + rows_per_blk;
    if (r1 > M) r1 = M;
    for (int cb = 0; cb < C; cb += cw) {
        int c = cb + c_in;
        float s = 0.f;
        if (c < C) {
            // eight independent rows per trip (one 4-byte load in flight per thread ran the nc = 80 head biases, C = 255, at 1 TB/s);
            // fixed order of the additions: reproducible
            float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
            int64_t r = r0 + r_in;
            const T *px = x + c;
            for (; r + 7 * rg < r1; r += 8 * rg) {
                const float v0 = ld1(px + r * ldx), v1 = ld1(px + (r + rg) * ldx), v2 = ld1(px + (r + 2 * rg) * ldx),
                            v3 = ld1(px + (r + 3 * rg) * ldx), v4 = ld1(px + (r + 4 * rg) * ldx), v5 = ld1(px + (r + 5 * rg) * ldx),
                            v6 = ld1(px + (r + 6 * rg) * ldx), v7 = ld1(px + (r + 7 * rg) * ldx);
                s += v0; s1 += v1; s2 += v2; s3 += v3; s4 += v4; s5 += v5; s6 += v6; s7 += v7;
            }
            for (; r < r1; r += rg) s += ld1(px + r * ldx);
            s = ((s + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
        }
        red[t] = s;
        __syncthreads();
        if (r_in == 0 && c < C) {
            float tot = 0.f;
            for (int k = 0; k < rg; ++k) tot += red[k * cw + c_in];
            part[(size_t)blockIdx.x * C + c] = tot;
        }
        __syncthreads();
    }
}
__global__ void colsum_stage2(const float *__restrict__ part, int nblk, int C, float *__restrict__ out) {
    const int c = blockIdx.x;            // one wave per channel, lanes stride the partials, fixed butterfly order
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += (double)part[(size_t)b * C + c];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) out[c] = (float)s;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm statistics: one workgroup per channel reduces the conv epilogue's partials in double.
__global__ void bn_finalize_kernel(const float *__restrict__ part, int nblk, double count,
                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                   float *__restrict__ rmean, float *__restrict__ rvar, float momentum, float eps,
                                   float *__restrict__ coef, int C, int64_t *__restrict__ nbt,
                                   float *__restrict__ xscale, float *__restrict__ xshift, long long *__restrict__ zacc) {
    __shared__ double rs[16], rq[16];
    const int c = blockIdx.x, t = threadIdx.x;
    double s = 0.0, q = 0.0;
#pragma unroll 4
    for (int b = t; b < nblk; b += blockDim.x) {     // latency-bound strided reads: many threads, loads kept in flight
        s += (double)part[((size_t)b * 2 + 0) * C + c];
        q += (double)part[((size_t)b * 2 + 1) * C + c];
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if ((t & 63) == 0) { rs[t >> 6] = s; rq[t >> 6] = q; }
    __syncthreads();
    if (t == 0) {
        double S = 0.0, Q = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { S += rs[w]; Q += rq[w]; }
        double mean = S / count, var = Q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        float invstd = (float)(1.0 / sqrt(var + (double)eps));
        float scale = gamma[c] * invstd;
        coef[c] = scale;
        coef[C + c] = beta[c] - (float)mean * scale;
        coef[2 * C + c] = (float)mean;
        coef[3 * C + c] = invstd;
        if (xscale) {            // rows of the CONSUMERS' input-prologue table (the activation is applied where it is read)
            xscale[c] = scale;
            xshift[c] = beta[c] - (float)mean * scale;
        }
        if (zacc) zacc[c] = zacc[C + c] = 0;      // this layer's BatchNorm-backward accumulators (yh_bn_silu_bwd_reduce_acc), once per step
        if (rmean) {
            double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
        }
        if (nbt && c == 0) *nbt += 1;
    }
}

__global__ void bn_eval_coef_kernel(const float *gamma, const float *beta, const float *rmean, const float *rvar,
                                    float eps, float *coef, int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float invstd = 1.0f / sqrtf(rvar[c] + eps);
    float scale = gamma[c] * invstd;
    coef[c] = scale;
    coef[C + c] = beta[c] - rmean[c] * scale;
    coef[2 * C + c] = rmean[c];
    coef[3 * C + c] = invstd;
}

__device__ __forceinline__ float silu_f(float z) { return z * yh_sigmoid_fast(z); }
__device__ __forceinline__ float silu_grad(float z) {
    float s = yh_sigmoid_fast(z);
    return s * (1.f + z * (1.f - s));
}

// Channel groups: G = 4 channels per thread and trip (fp32: one 16-byte access) or G = 8 (bf16 with C % 8 == 0: again one
// 16-byte access -- 8-byte accesses reach only ~4 TB/s on these passes); all arithmetic in fp32.
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int G> struct VecOf;
template <> struct VecOf<4> { typedef f32x4 type; };
template <> struct VecOf<8> { typedef f32x8 type; };
template <int G> __device__ __forceinline__ typename VecOf<G>::type ldg(const float *p) { return *(const typename VecOf<G>::type *)p; }
template <int G> __device__ __forceinline__ typename VecOf<G>::type ldg(const bf16 *p);
template <> __device__ __forceinline__ f32x4 ldg<4>(const bf16 *p) { return __builtin_convertvector(*(const bf16x4 *)p, f32x4); }
template <> __device__ __forceinline__ f32x8 ldg<8>(const bf16 *p) { return __builtin_convertvector(*(const bf16x8 *)p, f32x8); }
template <int G> __device__ __forceinline__ void stg(float *p, typename VecOf<G>::type v) { *(typename VecOf<G>::type *)p = v; }
__device__ __forceinline__ void stg4b(bf16 *p, f32x4 v) { *(bf16x4 *)p = __builtin_convertvector(v, bf16x4); }
template <int G> __device__ __forceinline__ void stg(bf16 *p, typename VecOf<G>::type v);
template <> __device__ __forceinline__ void stg<4>(bf16 *p, f32x4 v) { *(bf16x4 *)p = __builtin_convertvector(v, bf16x4); }
template <> __device__ __forceinline__ void stg<8>(bf16 *p, f32x8 v) { *(bf16x8 *)p = __builtin_convertvector(v, bf16x8); }

// a = silu(y*scale+shift) (+res); one channel group per thread and trip
template <typename T, int G>
__global__ __launch_bounds__(256) void bn_silu_fwd_kernel(const T *__restrict__ y, int ldy, const float *__restrict__ coef,
                                   const T *__restrict__ res, int ldr, T *__restrict__ out, int ldo,
                                   int64_t M, int C, int H, int W, int upsample, const float *__restrict__ rcoef, int rcoef_ld) {
    typedef typename VecOf<G>::type V;
    // rcoef: the residual tensor was never materialised either -- `res` holds its producer's raw convolution output and the
    // prologue table rows [scale | shift | gate] (rcoef_ld apart) turn it into the activation here (see yh_prologue)
    auto resv = [&](int64_t mm, int c) __attribute__((always_inline)) {
        V r = ldg<G>(res + mm * ldr + c);
        if (rcoef) {
#pragma unroll
            for (int e = 0; e < G; ++e) r[e] = yh_prologue(r[e], rcoef[c + e], rcoef[rcoef_ld + c + e], rcoef[2 * rcoef_ld + c + e]);
        }
        return r;
    };
    // (row, channel-group) cursor advanced incrementally: the grid-stride index i = m * cq + c4 is never divided inside
    // the loop (a 64-bit division per group cost more VALU time than the sigmoids)
    const int cq = C / G;
    const int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    int64_t m = i0 / cq;
    int c4 = (int)(i0 - m * cq);
    const int64_t dm = stride / cq;
    const int dc = (int)(stride - dm * cq);
    auto finish = [&](int64_t mm, int c, V a) __attribute__((always_inline)) {
        if (!upsample) {
            stg<G>(out + mm * ldo + c, a);
        } else {
            const unsigned mu32 = (unsigned)mm, q = mu32 / (unsigned)W, w = mu32 - q * (unsigned)W;      // M < 2^31
            const unsigned b = q / (unsigned)H, h = q - b * (unsigned)H;
            T *o = out + (((size_t)b * 2 * H + 2 * h) * 2 * W + 2 * w) * ldo + c;
            stg<G>(o, a);
            stg<G>(o + ldo, a);
            stg<G>(o + (size_t)2 * W * ldo, a);
            stg<G>(o + (size_t)2 * W * ldo + ldo, a);
        }
    };
    if (sizeof(T) == 2 && dc == 0) {      // (the fp32 pass already runs at the HBM rate: measured neutral there)
        // The grid stride is a whole number of rows (every BatchNorm layer of the model: C / G divides 256), so a thread keeps
        // its channel group: the coefficients are loaded once and two rows are in flight per trip.  One 16-byte load per
        // thread and trip left the bf16 pass latency-bound at 4.5 TB/s (same bytes in flight as fp32, twice the arithmetic
        // per byte).  Same expression per element as the general loop: bit-identical output.
        const int c = c4 * G;
        const V sc = *(const V *)(coef + c), sh = *(const V *)(coef + C + c);
        for (; m + dm < M; m += 2 * dm) {
            const V v0 = ldg<G>(y + m * ldy + c), v1 = ldg<G>(y + (m + dm) * ldy + c);
            V r0, r1;
            if (res) { r0 = resv(m, c); r1 = resv(m + dm, c); }
            V a0, a1;
#pragma unroll
            for (int e = 0; e < G; ++e) a0[e] = silu_f(v0[e] * sc[e] + sh[e]);
#pragma unroll
            for (int e = 0; e < G; ++e) a1[e] = silu_f(v1[e] * sc[e] + sh[e]);
            if (res) { a0 += r0; a1 += r1; }
            finish(m, c, a0);
            finish(m + dm, c, a1);
        }
        if (m < M) {
            const V v = ldg<G>(y + m * ldy + c);
            V a;
#pragma unroll
            for (int e = 0; e < G; ++e) a[e] = silu_f(v[e] * sc[e] + sh[e]);
            if (res) a += resv(m, c);
            finish(m, c, a);
        }
        return;
    }
    for (; m < M; m += dm, c4 += dc) {
        if (c4 >= cq) { c4 -= cq; if (++m >= M) break; }
        const int c = c4 * G;
        V v = ldg<G>(y + m * ldy + c);
        V sc = *(const V *)(coef + c), sh = *(const V *)(coef + C + c);
        V a;
#pragma unroll
        for (int e = 0; e < G; ++e) a[e] = silu_f(v[e] * sc[e] + sh[e]);
        if (res) a += resv(m, c);
        finish(m, c, a);
    }
}

template <int G, typename T>
__device__ __forceinline__ typename VecOf<G>::type load_da(const T *__restrict__ da, int ldda, int64_t m, int c, int H, int W,
                                                          int upsample) {
    if (!upsample) return ldg<G>(da + m * ldda + c);
    const unsigned mu32 = (unsigned)m, q = mu32 / (unsigned)W, w = mu32 - q * (unsigned)W;              // M < 2^31
    const unsigned b = q / (unsigned)H, h = q - b * (unsigned)H;
    const T *p = da + (((size_t)b * 2 * H + 2 * h) * 2 * W + 2 * w) * ldda + c;
    typename VecOf<G>::type a = ldg<G>(p), b1 = ldg<G>(p + ldda);
    typename VecOf<G>::type c1 = ldg<G>(p + (size_t)2 * W * ldda), d1 = ldg<G>(p + (size_t)2 * W * ldda + ldda);
    return (a + b1) + (c1 + d1);
}

// stage 1 of the backward: per-workgroup partial sums of dz and dz*xhat (per channel)
template <typename T, int G>
__global__ void bn_silu_bwd_reduce_kernel(const T *__restrict__ da, int ldda, const T *__restrict__ y,
                                          int ldy, const float *__restrict__ coef, float *__restrict__ part,
                                          int64_t M, int C, int H, int W, int upsample, int64_t rows_per_blk,
                                          unsigned long long *__restrict__ acc) {
    typedef typename VecOf<G>::type V;
    extern __shared__ float red[];   // [256][2 G]
    const int t = threadIdx.x, cq = C / G;
    const int rg = 256 / cq;          // row groups (cq <= 256)
    const int c4 = t % cq, r_in = t / cq;
    int64_t r0 = blockIdx.x * rows_per_blk, r1 = r0 + rows_per_blk;
    if (r1 > M) r1 = M;
    V s1, s2;
#pragma unroll
    for (int e = 0; e < G; ++e) s1[e] = s2[e] = 0.f;
    if (r_in < rg) {
        const int c = c4 * G;
        V sc = *(const V *)(coef + c), sh = *(const V *)(coef + C + c);
        V mu = *(const V *)(coef + 2 * C + c), is = *(const V *)(coef + 3 * C + c);
        int64_t m = r0 + r_in;
        for (; m + rg < r1; m += 2 * rg) {          // two rows per trip: four 16-byte loads in flight per thread
            V yv0 = ldg<G>(y + m * ldy + c), yv1 = ldg<G>(y + (m + rg) * ldy + c);
            V g0 = load_da<G>(da, ldda, m, c, H, W, upsample), g1 = load_da<G>(da, ldda, m + rg, c, H, W, upsample);
#pragma unroll
            for (int e = 0; e < G; ++e) {
                float dz = g0[e] * silu_grad(yv0[e] * sc[e] + sh[e]);
                s1[e] += dz;
                s2[e] += dz * ((yv0[e] - mu[e]) * is[e]);
            }
#pragma unroll
            for (int e = 0; e < G; ++e) {
                float dz = g1[e] * silu_grad(yv1[e] * sc[e] + sh[e]);
                s1[e] += dz;
                s2[e] += dz * ((yv1[e] - mu[e]) * is[e]);
            }
        }
        for (; m < r1; m += rg) {
            V yv = ldg<G>(y + m * ldy + c);
            V g = load_da<G>(da, ldda, m, c, H, W, upsample);
#pragma unroll
            for (int e = 0; e < G; ++e) {
                float dz = g[e] * silu_grad(yv[e] * sc[e] + sh[e]);
                s1[e] += dz;
                s2[e] += dz * ((yv[e] - mu[e]) * is[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < G; ++e) { red[t * 2 * G + e] = s1[e]; red[t * 2 * G + G + e] = s2[e]; }
    __syncthreads();
    // row groups summed in a fixed order, one COLUMN (channel, which sum) per thread: consecutive threads read consecutive
    // floats (the first form gave this loop to C / G threads with rg * 2 G reads each: as long as the main loop at 8-channel groups)
    const int ncol = cq * 2 * G;                 // = 2 C
    for (int col = t; col < ncol; col += 256) {
        float a = 0.f;
        for (int k = 0; k < rg; ++k) a += red[k * ncol + col];
        const int c4o = col / (2 * G), e = col - c4o * 2 * G;
        if (acc) {
            // Totals without a finalize launch: the workgroup's sum goes into a 64-bit FIXED-POINT accumulator (2^-36 units) with an
            // integer atomic -- integer addition is associative, so the total is bitwise reproducible whatever order the workgroups
            // arrive in (a float atomic would not be).  Resolution 1.5e-11 per workgroup, range +-1.3e8.
            const long long q = __double2ll_rn((double)a * 68719476736.0);
            atomicAdd(acc + (e >= G ? C : 0) + c4o * G + (e >= G ? e - G : e), (unsigned long long)q);
        } else {
            part[((size_t)blockIdx.x * 2 + (e >= G ? 1 : 0)) * C + c4o * G + (e >= G ? e - G : e)] = a;
        }
    }
}

// totals of the partials -> dbeta (sum dz) and dgamma (sum dz*xhat); one workgroup per channel
__global__ void bn_bwd_finalize_kernel(const float *__restrict__ part, int nblk, float *__restrict__ dgamma,
                                       float *__restrict__ dbeta, int C) {
    __shared__ double rs[16], rq[16];
    const int c = blockIdx.x, t = threadIdx.x;
    double s = 0.0, q = 0.0;
#pragma unroll 4
    for (int b = t; b < nblk; b += blockDim.x) {
        s += (double)part[((size_t)b * 2 + 0) * C + c];
        q += (double)part[((size_t)b * 2 + 1) * C + c];
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if ((t & 63) == 0) { rs[t >> 6] = s; rq[t >> 6] = q; }
    __syncthreads();
    if (t == 0) {
        double S = 0.0, Q = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { S += rs[w]; Q += rq[w]; }
        dbeta[c] = (float)S;
        dgamma[c] = (float)Q;
    }
}

template <typename T, int G>
__global__ __launch_bounds__(256) void bn_silu_bwd_apply_kernel(const T *__restrict__ da, int ldda, const T *__restrict__ y,
                                         int ldy, const float *__restrict__ coef, const float *__restrict__ dgamma,
                                         const float *__restrict__ dbeta, T *__restrict__ dy, int lddy,
                                         T *__restrict__ dres, int lddres, int res_acc, int64_t M, int C, int H,
                                         int W, int upsample, const long long *__restrict__ acc, float *__restrict__ dgo,
                                         float *__restrict__ dbo) {
    typedef typename VecOf<G>::type V;
    const int cq = C / G;
    const float inv_n = 1.0f / (float)M;
    // accumulator mode (acc != null): the totals of yh_bn_silu_bwd_reduce_acc are turned into floats once per workgroup (LDS);
    // workgroup 0 also writes them out as the parameter gradients
    extern __shared__ __attribute__((aligned(16))) float tot[];        // [2][C]: dbeta | dgamma
    if (acc) {
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
            const float v = (float)((double)acc[i] * (1.0 / 68719476736.0));
            tot[i] = v;
            if (blockIdx.x == 0) (i < C ? dbo[i] : dgo[i - C]) = v;
        }
        __syncthreads();
    }
    const int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    int64_t m = i0 / cq;                       // division-free cursor, see bn_silu_fwd_kernel
    int c4 = (int)(i0 - m * cq);
    const int64_t dm = stride / cq;
    const int dc = (int)(stride - dm * cq);
    auto coefs = [&](int c, V &sc, V &sh, V &mu, V &is, V &dg, V &db) __attribute__((always_inline)) {
        sc = *(const V *)(coef + c); sh = *(const V *)(coef + C + c);
        mu = *(const V *)(coef + 2 * C + c); is = *(const V *)(coef + 3 * C + c);
#pragma unroll
        for (int h = 0; h < G; h += 4) {       // (views of the flat gradient buffer: 16-byte aligned only)
            f32x4 g4, b4;
            if (acc) { g4 = *(const f32x4 *)(tot + C + c + h); b4 = *(const f32x4 *)(tot + c + h); }     // LDS (never a flat load)
            else { g4 = *(const f32x4 *)(dgamma + c + h); b4 = *(const f32x4 *)(dbeta + c + h); }
#pragma unroll
            for (int e = 0; e < 4; ++e) { dg[h + e] = g4[e]; db[h + e] = b4[e]; }
        }
    };
    auto apply = [&](int64_t mm, int c, const V &yv, V g, const V &sc, const V &sh, const V &mu, const V &is, const V &dg,
                     const V &db) __attribute__((always_inline)) {
        V o;
#pragma unroll
        for (int e = 0; e < G; ++e) {
            float dz = g[e] * silu_grad(yv[e] * sc[e] + sh[e]);
            float xh = (yv[e] - mu[e]) * is[e];
            o[e] = sc[e] * (dz - db[e] * inv_n - xh * dg[e] * inv_n);
        }
        stg<G>(dy + mm * lddy + c, o);
        if (dres) {
            T *r = dres + mm * lddres + c;
            if (res_acc) g += ldg<G>(r);
            stg<G>(r, g);
        }
    };
    if (sizeof(T) == 2 && dc == 0) {       // thread-invariant channel group: coefficients loaded once, two rows in flight (see bn_silu_fwd_kernel)
        const int c = c4 * G;
        V sc, sh, mu, is, dg, db;
        coefs(c, sc, sh, mu, is, dg, db);
        for (; m + dm < M; m += 2 * dm) {
            const V y0 = ldg<G>(y + m * ldy + c), y1 = ldg<G>(y + (m + dm) * ldy + c);
            const V g0 = load_da<G>(da, ldda, m, c, H, W, upsample), g1 = load_da<G>(da, ldda, m + dm, c, H, W, upsample);
            apply(m, c, y0, g0, sc, sh, mu, is, dg, db);
            apply(m + dm, c, y1, g1, sc, sh, mu, is, dg, db);
        }
        if (m < M) apply(m, c, ldg<G>(y + m * ldy + c), load_da<G>(da, ldda, m, c, H, W, upsample), sc, sh, mu, is, dg, db);
        return;
    }
    for (; m < M; m += dm, c4 += dc) {
        if (c4 >= cq) { c4 -= cq; if (++m >= M) break; }
        const int c = c4 * G;
        V yv = ldg<G>(y + m * ldy + c);
        V g = load_da<G>(da, ldda, m, c, H, W, upsample);
        V sc, sh, mu, is, dg, db;
        coefs(c, sc, sh, mu, is, dg, db);
        apply(m, c, yv, g, sc, sh, mu, is, dg, db);
    }
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool5_fwd_kernel(const T *__restrict__ x, int ldx, T *__restrict__ y, int ldy,
                                    uint8_t *__restrict__ arg, int H, int W, int C, int64_t total) {
    const int cq = C >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t m = i / cq;
        int c = (int)(i - m * cq) << 2;
        int w = (int)(m % W);
        int64_t q = m / W;
        int h = (int)(q % H);
        int64_t b = q / H;
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        bool first = true;
        for (int kh = 0; kh < 5; ++kh) {
            int ih = h + kh - 2;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < 5; ++kw) {
                int iw = w + kw - 2;
                if (iw < 0 || iw >= W) continue;
                f32x4 v = ld4(x + ((b * H + ih) * W + iw) * ldx + c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (first || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = kh * 5 + kw; }
                first = false;
            }
        }
        st4(y + m * ldy + c, best);
        uint32_t packed = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
        *(uint32_t *)(arg + m * C + c) = packed;
    }
}

template <typename T>
__global__ void maxpool5_bwd_kernel(const T *__restrict__ dy, int lddy, const uint8_t *__restrict__ arg,
                                    T *__restrict__ dx, int lddx, int H, int W, int C, int64_t total) {
    const int cq = C >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t m = i / cq;
        int c = (int)(i - m * cq) << 2;
        int w = (int)(m % W);
        int64_t q = m / W;
        int h = (int)(q % H);
        int64_t b = q / H;
        f32x4 acc = {0, 0, 0, 0};
        for (int kh = 0; kh < 5; ++kh) {
            int oh = h - kh + 2;
            if (oh < 0 || oh >= H) continue;
            for (int kw = 0; kw < 5; ++kw) {
                int ow = w - kw + 2;
                if (ow < 0 || ow >= W) continue;
                int64_t om = (b * H + oh) * W + ow;
                uint32_t packed = *(const uint32_t *)(arg + om * C + c);
                f32x4 g = ld4(dy + om * lddy + c);
                uint32_t want = (uint32_t)(kh * 5 + kw);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (((packed >> (8 * e)) & 0xffu) == want) acc[e] += g[e];
            }
        }
        T *d = dx + m * lddx + c;
        st4(d, ld4(d) + acc);
    }
}

__global__ void add_int64_kernel(int64_t *p, int64_t v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *p += v;
}

inline int grid_for(int64_t work_items, int threads = 256) {
    constexpr int cap = kMaxBlocks;
    int64_t g = cdiv64(work_items, threads);
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

template <typename T>
static int nchw_to_nhwc_t(const float *src, T *dst, int B, int C, int H, int W, int ld, int cpad, void *stream) {
    YH_REQUIRE(src && dst && B > 0 && C > 0 && cpad >= C && ld >= cpad, "nchw_to_nhwc: bad argument");
    int64_t npix = (int64_t)B * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, src, dst, C,
                       H * W, ld, cpad, npix);
    YH_CHECK_LAUNCH("nchw_to_nhwc");
    return 0;
}
extern "C" int yh_nchw_to_nhwc(const float *src, float *dst, int B, int C, int H, int W, int ld, int cpad, void *stream) {
    return nchw_to_nhwc_t<float>(src, dst, B, C, H, W, ld, cpad, stream);
}
extern "C" int yh_bf16_nchw_to_nhwc(const float *src, void *dst, int B, int C, int H, int W, int ld, int cpad, void *stream) {
    return nchw_to_nhwc_t<bf16>(src, (bf16 *)dst, B, C, H, W, ld, cpad, stream);
}

template <typename T>
static int u8hwc_to_nhwc_t(const uint8_t *src, T *dst, int B, int H, int W, int C, int ld, int cpad, void *stream) {
    YH_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && cpad >= C && ld >= cpad, "u8hwc_to_nhwc: bad argument");
    int64_t npix = (int64_t)B * H * W;
    hipLaunchKernelGGL(u8hwc_to_nhwc_kernel<T>, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, src, dst, C, ld, cpad, npix);
    YH_CHECK_LAUNCH("u8hwc_to_nhwc");
    return 0;
}
extern "C" int yh_u8hwc_to_nhwc(const uint8_t *src, float *dst, int B, int H, int W, int C, int ld, int cpad, void *stream) {
    return u8hwc_to_nhwc_t<float>(src, dst, B, H, W, C, ld, cpad, stream);
}
extern "C" int yh_bf16_u8hwc_to_nhwc(const uint8_t *src, void *dst, int B, int H, int W, int C, int ld, int cpad, void *stream) {
    return u8hwc_to_nhwc_t<bf16>(src, (bf16 *)dst, B, H, W, C, ld, cpad, stream);
}

template <typename T>
static int nhwc_to_nchw_t(const T *src, float *dst, int B, int C, int H, int W, int ld, int accumulate, void *stream) {
    YH_REQUIRE(src && dst && B > 0 && C > 0 && ld >= C, "nhwc_to_nchw: bad argument");
    int64_t total = (int64_t)B * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, C,
                       H * W, ld, accumulate, total);
    YH_CHECK_LAUNCH("nhwc_to_nchw");
    return 0;
}
extern "C" int yh_nhwc_to_nchw(const float *src, float *dst, int B, int C, int H, int W, int ld, int accumulate, void *stream) {
    return nhwc_to_nchw_t<float>(src, dst, B, C, H, W, ld, accumulate, stream);
}
extern "C" int yh_bf16_nhwc_to_nchw(const void *src, float *dst, int B, int C, int H, int W, int ld, int accumulate, void *stream) {
    return nhwc_to_nchw_t<bf16>((const bf16 *)src, dst, B, C, H, W, ld, accumulate, stream);
}

// merged stride-2 backward pack: W'[kh][c][co][pw*Cin + ci], see yh_conv_bwd_data_s2m
__global__ void pack_weights_s2m_kernel(const float *__restrict__ w, float *__restrict__ wbm, int Cout, int Cin, int ldw) {
    const int total = 6 * Cout * ldw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int n = i % ldw, q = i / ldw;
        int co = q % Cout, t = q / Cout;
        int c = t & 1, kh = t >> 1;
        float v = 0.f;
        if (n < 2 * Cin) {
            int pw = n / Cin, ci = n - pw * Cin;
            int kw = pw == 0 ? (c == 0 ? 1 : -1) : (c == 0 ? 2 : 0);
            if (kw >= 0) v = w[((size_t)co * Cin + ci) * 9 + kh * 3 + kw];
        }
        wbm[i] = v;
    }
}

extern "C" int yh_pack_weights_s2m(const float *oihw, float *wbm, int Cout, int Cin, int ldw, void *stream) {
    YH_REQUIRE(oihw && wbm && Cout > 0 && Cin > 0 && ldw >= 2 * Cin && ldw % 4 == 0, "pack_weights_s2m: bad argument");
    hipLaunchKernelGGL(pack_weights_s2m_kernel, dim3(grid_for((int64_t)6 * Cout * ldw)), dim3(256), 0, (hipStream_t)stream, oihw, wbm,
                       Cout, Cin, ldw);
    YH_CHECK_LAUNCH("pack_weights_s2m");
    return 0;
}

extern "C" int yh_pack_weights(const float *oihw, float *wf, float *wb, int Cout, int Cin, int k, int cin_pad,
                               int ldwf, int ldwb, void *stream) {
    YH_REQUIRE(oihw && (wf || wb) && cin_pad >= Cin, "pack_weights: bad argument");
    YH_REQUIRE(!wf || (ldwf >= Cout && ldwf % 4 == 0), "pack_weights: ldwf=%d must be >= Cout and a multiple of 4", ldwf);
    YH_REQUIRE(!wb || (ldwb >= Cin && ldwb % 4 == 0), "pack_weights: ldwb=%d must be >= Cin and a multiple of 4", ldwb);
    int64_t n = (wf ? (int64_t)k * k * cin_pad * ldwf : 0) + (wb ? (int64_t)k * k * Cout * ldwb : 0);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, oihw, wf, wb, Cout,
                       Cin, k * k, cin_pad, ldwf, ldwb);
    YH_CHECK_LAUNCH("pack_weights");
    return 0;
}

extern "C" int yh_pack_weights_multi(const void *table, int n_layers, void *stream) {
    YH_REQUIRE(table && n_layers > 0, "pack_weights_multi: bad argument");
    static_assert(sizeof(PackDesc) == 56, "descriptor layout is part of the ABI");
    hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(256, n_layers), dim3(256), 0, (hipStream_t)stream,
                       (const PackDesc *)table);
    YH_CHECK_LAUNCH("pack_weights_multi");
    return 0;
}

extern "C" int yh_pack_fold_multi(const void *table, int n_layers, void *stream) {
    YH_REQUIRE(table && n_layers > 0, "pack_fold_multi: bad argument");
    static_assert(sizeof(FoldDesc) == 88, "descriptor layout is part of the ABI");
    hipLaunchKernelGGL(pack_fold_multi_kernel, dim3(32, n_layers), dim3(256), 0, (hipStream_t)stream,
                       (const FoldDesc *)table);
    YH_CHECK_LAUNCH("pack_fold_multi");
    return 0;
}

extern "C" int yh_fold_oihw_multi(const void *table, int n_layers, void *stream) {
    YH_REQUIRE(table && n_layers > 0, "fold_oihw_multi: bad argument");
    static_assert(sizeof(FoldOihwDesc) == 80, "descriptor layout is part of the ABI");
    hipLaunchKernelGGL(fold_oihw_multi_kernel, dim3(32, n_layers), dim3(256), 0, (hipStream_t)stream, (const FoldOihwDesc *)table);
    YH_CHECK_LAUNCH("fold_oihw_multi");
    return 0;
}

static int colsum_blocks(int64_t M) {
    int64_t b = cdiv64(M, 512);
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
extern "C" int64_t yh_colsum_ws(int64_t M, int C) { return (int64_t)colsum_blocks(M) * C; }
template <typename T>
static int colsum_t(const T *x, int ldx, int64_t M, int C, float *out, float *ws, void *stream) {
    YH_REQUIRE(x && out && ws && M > 0 && C > 0 && ldx >= C, "colsum: bad argument");
    int nblk = colsum_blocks(M);
    int64_t rows = cdiv64(M, nblk);
    if (C % 4 == 0 && C <= 1024 && ldx % 4 == 0 && ((uintptr_t)x & (4 * sizeof(T) - 1)) == 0)
        hipLaunchKernelGGL(colsum_stage1_vec<T>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, ldx, M, C, ws, rows);
    else
        hipLaunchKernelGGL(colsum_stage1<T>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, ldx, M, C, ws, rows);
    YH_CHECK_LAUNCH("colsum_stage1");
    hipLaunchKernelGGL(colsum_stage2, dim3(C), dim3(64), 0, (hipStream_t)stream, ws, nblk, C, out);
    YH_CHECK_LAUNCH("colsum_stage2");
    return 0;
}
extern "C" int yh_colsum(const float *x, int ldx, int64_t M, int C, float *out, float *ws, void *stream) {
    return colsum_t<float>(x, ldx, M, C, out, ws, stream);
}
extern "C" int yh_bf16_colsum(const void *x, int ldx, int64_t M, int C, float *out, float *ws, void *stream) {
    return colsum_t<bf16>((const bf16 *)x, ldx, M, C, out, ws, stream);
}

extern "C" int yh_bn_finalize_x(const float *partials, int nblk, int64_t count, const float *gamma, const float *beta,
                                float *running_mean, float *running_var, float momentum, float eps, float *coef, int C,
                                int64_t *num_batches_tracked, float *xscale, float *xshift, int64_t *bwd_acc, void *stream) {
    YH_REQUIRE(partials && gamma && beta && coef && nblk > 0 && count > 0 && C > 0 && !xscale == !xshift, "bn_finalize: bad argument");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(nblk >= 2048 ? 1024 : 256), 0, (hipStream_t)stream, partials, nblk, (double)count,
                       gamma, beta, running_mean, running_var, momentum, eps, coef, C, num_batches_tracked, xscale, xshift, (long long *)bwd_acc);
    YH_CHECK_LAUNCH("bn_finalize");
    return 0;
}
extern "C" int yh_bn_finalize(const float *partials, int nblk, int64_t count, const float *gamma, const float *beta,
                              float *running_mean, float *running_var, float momentum, float eps, float *coef, int C,
                              int64_t *num_batches_tracked, void *stream) {
    return yh_bn_finalize_x(partials, nblk, count, gamma, beta, running_mean, running_var, momentum, eps, coef, C, num_batches_tracked,
                            nullptr, nullptr, nullptr, stream);
}

extern "C" int yh_bn_eval_coef(const float *gamma, const float *beta, const float *running_mean,
                               const float *running_var, float eps, float *coef, int C, void *stream) {
    YH_REQUIRE(gamma && beta && running_mean && running_var && coef && C > 0, "bn_eval_coef: bad argument");
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, coef, C);
    YH_CHECK_LAUNCH("bn_eval_coef");
    return 0;
}

// 8-channel groups (one 16-byte access per bf16 tensor): bf16 only, every channel count / stride a multiple of 8, every
// view 16-byte aligned, and few enough groups per row for the reduce kernel's 256-thread layout
template <typename T>
static bool wide_groups(int C, std::initializer_list<int> lds, std::initializer_list<const void *> ptrs) {
    if (sizeof(T) != 2 || C % 8 != 0 || C / 8 > 256) return false;
    for (int ld : lds)
        if (ld % 8 != 0) return false;
    for (const void *p : ptrs)
        if (((uintptr_t)p & 15) != 0) return false;
    return true;
}

#define YH_REQ_VEC4(name, C, ...)                                                                     \
    YH_REQUIRE((C) % 4 == 0 && (C) <= 1024, name ": C=%d must be a multiple of 4 (<= 1024)", (C));       \
    do {                                                                                              \
        const int lds__[] = {__VA_ARGS__};                                                            \
        for (int ld__ : lds__) YH_REQUIRE(ld__ % 4 == 0, name ": ld=%d must be a multiple of 4", ld__); \
    } while (0)

template <typename T>
static int bn_silu_fwd_t(const T *y, int ldy, const float *coef, const T *residual, int ldr, T *out, int ldo, int64_t M, int C,
                         int H, int W, int upsample, const float *rcoef, int rcoef_ld, void *stream) {
    YH_REQUIRE(y && coef && out && M > 0 && M < (1ll << 31) && (!rcoef || (residual && rcoef_ld >= C)), "bn_silu_fwd: bad argument");
    YH_REQ_VEC4("bn_silu_fwd", C, ldy, ldo, residual ? ldr : 0);
    YH_REQUIRE(!upsample || (H > 0 && W > 0 && M % ((int64_t)H * W) == 0), "bn_silu_fwd: upsample needs H, W");
    if (wide_groups<T>(C, {ldy, ldo, residual ? ldr : 0}, {y, out, residual}))
        hipLaunchKernelGGL((bn_silu_fwd_kernel<T, 8>), dim3(grid_for(M * (C / 8))), dim3(256), 0, (hipStream_t)stream, y, ldy, coef,
                           residual, ldr, out, ldo, M, C, H, W, upsample, rcoef, rcoef_ld);
    else
        hipLaunchKernelGGL((bn_silu_fwd_kernel<T, 4>), dim3(grid_for(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, y, ldy, coef,
                           residual, ldr, out, ldo, M, C, H, W, upsample, rcoef, rcoef_ld);
    YH_CHECK_LAUNCH("bn_silu_fwd");
    return 0;
}
extern "C" int yh_bn_silu_fwd(const float *y, int ldy, const float *coef, const float *residual, int ldr, float *out,
                              int ldo, int64_t M, int C, int H, int W, int upsample, void *stream) {
    return bn_silu_fwd_t<float>(y, ldy, coef, residual, ldr, out, ldo, M, C, H, W, upsample, nullptr, 0, stream);
}
extern "C" int yh_bn_silu_fwd_res(const float *y, int ldy, const float *coef, const float *residual, int ldr, const float *rcoef,
                                  int rcoef_ld, float *out, int ldo, int64_t M, int C, int H, int W, int upsample, void *stream) {
    return bn_silu_fwd_t<float>(y, ldy, coef, residual, ldr, out, ldo, M, C, H, W, upsample, rcoef, rcoef_ld, stream);
}
extern "C" int yh_bf16_bn_silu_fwd(const void *y, int ldy, const float *coef, const void *residual, int ldr, void *out,
                                   int ldo, int64_t M, int C, int H, int W, int upsample, void *stream) {
    return bn_silu_fwd_t<bf16>((const bf16 *)y, ldy, coef, (const bf16 *)residual, ldr, (bf16 *)out, ldo, M, C, H, W, upsample, nullptr, 0, stream);
}

extern "C" int yh_bn_bwd_blocks(int64_t M, int C) {
    int rg = 256 / (C / 4 > 0 ? C / 4 : 1);
    if (rg < 1) rg = 1;
    int64_t b = cdiv64(M, (int64_t)rg * 16);
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

template <typename T>
static int bn_silu_bwd_reduce_t(const T *da, int ldda, const T *y, int ldy, const float *coef, float *partials, int64_t *acc, int64_t M, int C,
                                int H, int W, int upsample, void *stream) {
    YH_REQUIRE(da && y && coef && (partials || acc) && M > 0 && M < (1ll << 31), "bn_silu_bwd_reduce: bad argument");
    YH_REQ_VEC4("bn_silu_bwd_reduce", C, ldda, ldy);
    YH_REQUIRE(C <= 1024, "bn_silu_bwd_reduce: C too large");
    int nblk = yh_bn_bwd_blocks(M, C);
    int64_t rows = cdiv64(M, nblk);
    // bf16: 8-channel groups = 16-byte loads (8-byte loads top out at ~3.3 TB/s on this pass: the address path, not HBM); they
    // measured slower in round 2 (0.87 -> 1.01 ms) because the final sum over the row groups ran on C / 8 threads -- now one
    // column per thread
    if (sizeof(T) == 2 && wide_groups<T>(C, {ldda, ldy}, {da, y}))
        hipLaunchKernelGGL((bn_silu_bwd_reduce_kernel<T, 8>), dim3(nblk), dim3(256), 256 * 16 * sizeof(float), (hipStream_t)stream,
                           da, ldda, y, ldy, coef, partials, M, C, H, W, upsample, rows, (unsigned long long *)acc);
    else
        hipLaunchKernelGGL((bn_silu_bwd_reduce_kernel<T, 4>), dim3(nblk), dim3(256), 256 * 8 * sizeof(float), (hipStream_t)stream,
                           da, ldda, y, ldy, coef, partials, M, C, H, W, upsample, rows, (unsigned long long *)acc);
    YH_CHECK_LAUNCH("bn_silu_bwd_reduce");
    return 0;
}
extern "C" int yh_bn_silu_bwd_reduce(const float *da, int ldda, const float *y, int ldy, const float *coef,
                                     float *partials, int64_t M, int C, int H, int W, int upsample, void *stream) {
    return bn_silu_bwd_reduce_t<float>(da, ldda, y, ldy, coef, partials, nullptr, M, C, H, W, upsample, stream);
}
extern "C" int yh_bn_silu_bwd_reduce_acc(const float *da, int ldda, const float *y, int ldy, const float *coef, int64_t *acc, int64_t M,
                                         int C, int H, int W, int upsample, void *stream) {
    return bn_silu_bwd_reduce_t<float>(da, ldda, y, ldy, coef, nullptr, acc, M, C, H, W, upsample, stream);
}
extern "C" int yh_bf16_bn_silu_bwd_reduce_acc(const void *da, int ldda, const void *y, int ldy, const float *coef, int64_t *acc, int64_t M,
                                              int C, int H, int W, int upsample, void *stream) {
    return bn_silu_bwd_reduce_t<bf16>((const bf16 *)da, ldda, (const bf16 *)y, ldy, coef, nullptr, acc, M, C, H, W, upsample, stream);
}
extern "C" int yh_bf16_bn_silu_bwd_reduce(const void *da, int ldda, const void *y, int ldy, const float *coef,
                                          float *partials, int64_t M, int C, int H, int W, int upsample, void *stream) {
    return bn_silu_bwd_reduce_t<bf16>((const bf16 *)da, ldda, (const bf16 *)y, ldy, coef, partials, nullptr, M, C, H, W, upsample, stream);
}

template <typename T>
static int bn_silu_bwd_apply_t(const T *da, int ldda, const T *y, int ldy, const float *coef,
                               const float *partials, int nblk, const int64_t *acc, const float *gamma, float *dgamma, float *dbeta,
                               T *dy, int lddy, T *dres, int lddres, int res_accumulate, int64_t M, int C,
                               int H, int W, int upsample, void *stream) {
    (void)gamma;
    YH_REQUIRE(da && y && coef && (acc || (partials && nblk > 0)) && dgamma && dbeta && dy && M > 0 && M < (1ll << 31), "bn_silu_bwd_apply: bad argument");
    YH_REQ_VEC4("bn_silu_bwd_apply", C, ldda, ldy, lddy, dres ? lddres : 0);
    if (!acc) {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(nblk >= 2048 ? 1024 : 256), 0, (hipStream_t)stream, partials, nblk, dgamma,
                           dbeta, C);
        YH_CHECK_LAUNCH("bn_bwd_finalize");
    }
    const size_t smem = acc ? (size_t)2 * C * sizeof(float) : 0;
    if (wide_groups<T>(C, {ldda, ldy, lddy, dres ? lddres : 0}, {da, y, dy, dres}))
        hipLaunchKernelGGL((bn_silu_bwd_apply_kernel<T, 8>), dim3(grid_for(M * (C / 8))), dim3(256), smem, (hipStream_t)stream, da, ldda,
                           y, ldy, coef, dgamma, dbeta, dy, lddy, dres, lddres, res_accumulate, M, C, H, W, upsample, (const long long *)acc,
                           dgamma, dbeta);
    else
        hipLaunchKernelGGL((bn_silu_bwd_apply_kernel<T, 4>), dim3(grid_for(M * (C / 4))), dim3(256), smem, (hipStream_t)stream, da, ldda,
                           y, ldy, coef, dgamma, dbeta, dy, lddy, dres, lddres, res_accumulate, M, C, H, W, upsample, (const long long *)acc,
                           dgamma, dbeta);
    YH_CHECK_LAUNCH("bn_silu_bwd_apply");
    return 0;
}
extern "C" int yh_bn_silu_bwd_apply_acc(const float *da, int ldda, const float *y, int ldy, const float *coef, const int64_t *acc,
                                        float *dgamma, float *dbeta, float *dy, int lddy, float *dres, int lddres, int res_accumulate,
                                        int64_t M, int C, int H, int W, int upsample, void *stream) {
    return bn_silu_bwd_apply_t<float>(da, ldda, y, ldy, coef, nullptr, 0, acc, nullptr, dgamma, dbeta, dy, lddy, dres, lddres,
                                      res_accumulate, M, C, H, W, upsample, stream);
}
extern "C" int yh_bf16_bn_silu_bwd_apply_acc(const void *da, int ldda, const void *y, int ldy, const float *coef, const int64_t *acc,
                                             float *dgamma, float *dbeta, void *dy, int lddy, void *dres, int lddres, int res_accumulate,
                                             int64_t M, int C, int H, int W, int upsample, void *stream) {
    return bn_silu_bwd_apply_t<bf16>((const bf16 *)da, ldda, (const bf16 *)y, ldy, coef, nullptr, 0, acc, nullptr, dgamma, dbeta, (bf16 *)dy,
                                     lddy, (bf16 *)dres, lddres, res_accumulate, M, C, H, W, upsample, stream);
}
extern "C" int yh_bn_silu_bwd_apply(const float *da, int ldda, const float *y, int ldy, const float *coef,
                                    const float *partials, int nblk, const float *gamma, float *dgamma, float *dbeta,
                                    float *dy, int lddy, float *dres, int lddres, int res_accumulate, int64_t M, int C,
                                    int H, int W, int upsample, void *stream) {
    return bn_silu_bwd_apply_t<float>(da, ldda, y, ldy, coef, partials, nblk, nullptr, gamma, dgamma, dbeta, dy, lddy, dres, lddres,
                                      res_accumulate, M, C, H, W, upsample, stream);
}
extern "C" int yh_bf16_bn_silu_bwd_apply(const void *da, int ldda, const void *y, int ldy, const float *coef,
                                         const float *partials, int nblk, const float *gamma, float *dgamma, float *dbeta,
                                         void *dy, int lddy, void *dres, int lddres, int res_accumulate, int64_t M, int C,
                                         int H, int W, int upsample, void *stream) {
    return bn_silu_bwd_apply_t<bf16>((const bf16 *)da, ldda, (const bf16 *)y, ldy, coef, partials, nblk, nullptr, gamma, dgamma, dbeta,
                                     (bf16 *)dy, lddy, (bf16 *)dres, lddres, res_accumulate, M, C, H, W, upsample, stream);
}

template <typename T>
static int maxpool5_fwd_t(const T *x, int ldx, T *y, int ldy, uint8_t *argmax, int B, int H, int W, int C, void *stream) {
    YH_REQUIRE(x && y && argmax && B > 0 && H > 0 && W > 0, "maxpool5_fwd: bad argument");
    YH_REQ_VEC4("maxpool5_fwd", C, ldx, ldy);
    int64_t total = (int64_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool5_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy,
                       argmax, H, W, C, total);
    YH_CHECK_LAUNCH("maxpool5_fwd");
    return 0;
}
extern "C" int yh_maxpool5_fwd(const float *x, int ldx, float *y, int ldy, uint8_t *argmax, int B, int H, int W, int C,
                               void *stream) {
    return maxpool5_fwd_t<float>(x, ldx, y, ldy, argmax, B, H, W, C, stream);
}
extern "C" int yh_bf16_maxpool5_fwd(const void *x, int ldx, void *y, int ldy, uint8_t *argmax, int B, int H, int W, int C,
                                    void *stream) {
    return maxpool5_fwd_t<bf16>((const bf16 *)x, ldx, (bf16 *)y, ldy, argmax, B, H, W, C, stream);
}

template <typename T>
static int maxpool5_bwd_t(const T *dy, int lddy, const uint8_t *argmax, T *dx, int lddx, int B, int H, int W, int C, void *stream) {
    YH_REQUIRE(dy && dx && argmax && B > 0 && H > 0 && W > 0, "maxpool5_bwd: bad argument");
    YH_REQ_VEC4("maxpool5_bwd", C, lddy, lddx);
    int64_t total = (int64_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool5_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, lddy, argmax,
                       dx, lddx, H, W, C, total);
    YH_CHECK_LAUNCH("maxpool5_bwd");
    return 0;
}
extern "C" int yh_maxpool5_bwd(const float *dy, int lddy, const uint8_t *argmax, float *dx, int lddx, int B, int H,
                               int W, int C, void *stream) {
    return maxpool5_bwd_t<float>(dy, lddy, argmax, dx, lddx, B, H, W, C, stream);
}
extern "C" int yh_bf16_maxpool5_bwd(const void *dy, int lddy, const uint8_t *argmax, void *dx, int lddx, int B, int H,
                                    int W, int C, void *stream) {
    return maxpool5_bwd_t<bf16>((const bf16 *)dy, lddy, argmax, (bf16 *)dx, lddx, B, H, W, C, stream);
}

// ---- inference SPPF: the three cascaded 5x5 pools of one image in ONE launch ---------------------------------------------
// y1 = pool5(x), y2 = pool5(y1), y3 = pool5(y2) (train.py:246-248).  A workgroup owns four channels of one image and keeps
// the whole H x W plane in LDS (two planes of float4): each stage is a 5-tap row maximum followed by a 5-tap column maximum
// (the 5x5 window maximum is separable; NaN wins like torch's max_pool2d), written to the stage's output and left in LDS as
// the next stage's input.  At batch 1 the three launches of the training kernel were 12 us each for a 400-pixel plane.
__device__ __forceinline__ f32x4 nan_max4(f32x4 a, f32x4 b) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (b[e] > a[e] || b[e] != b[e]) a[e] = b[e];
    return a;
}

__global__ __launch_bounds__(256) void sppf_pool3_kernel(const float *__restrict__ x, int ldx, float *__restrict__ y1,
                                                         float *__restrict__ y2, float *__restrict__ y3, int ldy, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float pool_smem[];
    f32x4 *P = (f32x4 *)pool_smem, *Q = P + H * W;
    const int c = blockIdx.x * 4, HW = H * W;
    const size_t img = (size_t)blockIdx.y * HW;
    for (int p = threadIdx.x; p < HW; p += 256) P[p] = ld4(x + (img + p) * ldx + c);
    __syncthreads();
    float *const outs[3] = {y1, y2, y3};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        for (int p = threadIdx.x; p < HW; p += 256) {
            const int h = p / W, w = p - h * W;
            const int lo = w < 2 ? -w : -2, hi = w + 2 >= W ? W - 1 - w : 2;
            f32x4 m = P[p + lo];
            for (int d = lo + 1; d <= hi; ++d) m = nan_max4(m, P[p + d]);
            Q[p] = m;
        }
        __syncthreads();
        for (int p = threadIdx.x; p < HW; p += 256) {
            const int h = p / W;
            const int lo = h < 2 ? -h : -2, hi = h + 2 >= H ? H - 1 - h : 2;
            f32x4 m = Q[p + lo * W];
            for (int d = lo + 1; d <= hi; ++d) m = nan_max4(m, Q[p + d * W]);
            P[p] = m;
            st4(outs[s] + (img + p) * ldy + c, m);
        }
        __syncthreads();
    }
}

extern "C" int yh_sppf_pool3_ok(int H, int W) { return H > 0 && W > 0 && (int64_t)H * W <= 4096; }

extern "C" int yh_sppf_pool3_fwd(const float *x, int ldx, float *y1, float *y2, float *y3, int ldy, int B, int H, int W, int C,
                                 void *stream) {
    YH_REQUIRE(x && y1 && y2 && y3 && B > 0 && B < 65536, "sppf_pool3_fwd: bad argument");
    YH_REQUIRE(yh_sppf_pool3_ok(H, W), "sppf_pool3_fwd: a %d x %d plane does not fit the LDS form (H*W <= 4096); use yh_maxpool5_fwd", H, W);
    YH_REQ_VEC4("sppf_pool3_fwd", C, ldx, ldy);
    YH_REQUIRE((((uintptr_t)y1 | (uintptr_t)y2 | (uintptr_t)y3) & 15) == 0, "sppf_pool3_fwd: outputs must be 16-byte aligned");
    const size_t smem = (size_t)2 * H * W * 16;
    if (int rc = yh_ensure_dyn_smem((const void *)sppf_pool3_kernel, smem)) return rc;
    hipLaunchKernelGGL(sppf_pool3_kernel, dim3(C / 4, B), dim3(256), smem, (hipStream_t)stream, x, ldx, y1, y2, y3, ldy, H, W);
    YH_CHECK_LAUNCH("sppf_pool3_fwd");
    return 0;
}

extern "C" int yh_add_int64(int64_t *p, int64_t v, void *stream) {
    YH_REQUIRE(p, "add_int64: null pointer");
    hipLaunchKernelGGL(add_int64_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p, v);
    YH_CHECK_LAUNCH("add_int64");
    return 0;
}
