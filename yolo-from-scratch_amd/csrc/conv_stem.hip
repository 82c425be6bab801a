// First convolution of the network (train.py:402-403: Conv2d(3, 16, 3, stride 2, padding 1) on the 640x640 image).
// K = 27 is too small for the MFMA path to amortise anything and the layer is HBM-bound (420 MB in, 420 MB out at
// bs = 64): a direct VALU convolution -- one thread = one output pixel x 16 channels, weights through scalar loads --
// reaches the memory roofline where the implicit-GEMM kernel does not (0.45 ms -> see DESIGN.md).  Input NHWC4 (RGB + zero
// channel, ld = 4), weights in the forward pack layout [tap][4][16], output NHWC (ld >= 16), optional bias and the
// per-workgroup BatchNorm partial sums [blocks][2][16] (1024 output pixels per workgroup).
#include "common.h"

namespace {

constexpr int PPB = 1024;       // output pixels per workgroup (4 per thread)

__global__ __launch_bounds__(256) void stem_conv_kernel(const float *__restrict__ x, const float *__restrict__ wf,
                                                        const float *__restrict__ bias, float *__restrict__ y, int ldy,
                                                        float *__restrict__ stats, int Hi, int Wi, int Ho, int Wo, int M) {
    __shared__ float red[4][32];
    const int t = threadIdx.x;
    float s[16], q[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) s[c] = q[c] = 0.f;
    for (int i = 0; i < PPB / 256; ++i) {
        const int p = blockIdx.x * PPB + i * 256 + t;
        if (p >= M) break;
        const int ox = p % Wo, r = p / Wo, oy = r % Ho, b = r / Ho;
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = bias ? bias[c] : 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = 2 * oy - 1 + kh;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = 2 * ox - 1 + kw;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) v = *(const f32x4 *)(x + ((size_t)(b * Hi + iy) * Wi + ix) * 4);
                const float *w = wf + (kh * 3 + kw) * 64;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                    for (int c = 0; c < 16; ++c) acc[c] = fmaf(v[ci], w[ci * 16 + c], acc[c]);
            }
        }
        float *o = y + (size_t)p * ldy;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) *(f32x4 *)(o + 4 * c4) = f32x4{acc[4 * c4], acc[4 * c4 + 1], acc[4 * c4 + 2], acc[4 * c4 + 3]};
#pragma unroll
        for (int c = 0; c < 16; ++c) { s[c] += acc[c]; q[c] = fmaf(acc[c], acc[c], q[c]); }
    }
    if (stats) {
#pragma unroll
        for (int c = 0; c < 16; ++c) { s[c] = wave_sum(s[c]); q[c] = wave_sum(q[c]); }
        if ((t & 63) == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) { red[t >> 6][c] = s[c]; red[t >> 6][16 + c] = q[c]; }
        }
        __syncthreads();
        if (t < 32) stats[(size_t)blockIdx.x * 32 + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    }
}

}  // namespace

extern "C" int yh_conv_stem_blocks(int B, int Hi, int Wi) { return cdiv(B * ((Hi - 1) / 2 + 1) * ((Wi - 1) / 2 + 1), PPB); }

extern "C" int yh_conv_stem_fwd(const float *x, const float *wf, const float *bias, float *y, int ldy, float *bn_partials, int B,
                                int Hi, int Wi, void *stream) {
    YH_REQUIRE(x && wf && y && B > 0 && Hi > 0 && Wi > 0 && ldy >= 16 && ldy % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0,
               "conv_stem_fwd: bad argument");
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    YH_REQUIRE((int64_t)B * Hi * Wi * 4 < (1ll << 31), "conv_stem_fwd: input too large");
    const int M = B * Ho * Wo;
    hipLaunchKernelGGL(stem_conv_kernel, dim3(cdiv(M, PPB)), dim3(256), 0, (hipStream_t)stream, x, wf, bias, y, ldy, bn_partials, Hi,
                       Wi, Ho, Wo, M);
    YH_CHECK_LAUNCH("conv_stem_fwd");
    return 0;
}
