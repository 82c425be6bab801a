// Bare MFMA issue-rate probe (fp32 matrix instructions, operands in registers, random data): TFLOP/s for NACC independent
// accumulator chains at 1, 2 and 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int SHAPE>
__global__ __launch_bounds__(256) void probe(const float *in, float *out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[4], b[4];
    for (int e = 0; e < 4; ++e) { a[e] = in[(t * 8 + e) & 65535]; b[e] = in[(t * 8 + 4 + e) & 65535]; }
    if constexpr (SHAPE == 32) {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[(e + i) & 3], acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[t] = s;
    } else {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[(e + i) & 3], acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[t] = s;
    }
}

template <int NACC, int SHAPE>
void run(const float *din, float *dout, int wgs, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<NACC, SHAPE>), dim3(wgs), dim3(256), 0, 0, din, dout, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL((probe<NACC, SHAPE>), dim3(wgs), dim3(256), 0, 0, din, dout, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flop = (double)wgs * 4 * iters * 4 * NACC * (SHAPE == 32 ? 4096.0 : 2048.0);
    printf("shape %dx%d nacc %d wgs %4d (%.1f waves/SIMD): %.3f ms  %.1f TFLOP/s\n", SHAPE, SHAPE, NACC, wgs, wgs / 256.0, ms, flop / ms / 1e9);
}

int main() {
    std::vector<float> h(65536);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *din, *dout;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, 4096 * 256 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int wgs : {256, 512, 1024}) {
        run<1, 32>(din, dout, wgs, 4000);
        run<2, 32>(din, dout, wgs, 2000);
        run<4, 32>(din, dout, wgs, 1000);
        run<8, 32>(din, dout, wgs, 500);
        run<4, 16>(din, dout, wgs, 2000);
        run<9, 16>(din, dout, wgs, 1000);
    }
    return 0;
}
