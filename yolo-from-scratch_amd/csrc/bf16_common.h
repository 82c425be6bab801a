// Types and small helpers shared by the bf16 convolution kernels (conv_bf16.hip, conv_bf16_stream.hip).
#pragma once
#include "common.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// n / d for 0 <= n < 2^31 with (magic, shift) from yh_set_magic(d); shift < 0 means d == 1
__device__ __forceinline__ int yh_fast_div(int n, unsigned magic, int shift) {
    return shift < 0 ? n : (int)(__umulhi((unsigned)n, magic) >> shift);
}
static inline void yh_set_magic(unsigned d, unsigned &magic, int &shift) {
    int l = 0;
    while ((1u << l) < d) ++l;
    magic = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
    shift = l - 1;
}

// v_mfma_f32_32x32x16_bf16: D register q of lane (col = lane & 31, h = lane >> 5) holds row (q & 3) + 8 (q >> 2) + 4 h
__device__ __forceinline__ int yh_mfma_row(int q, int lh) { return (q & 3) + 8 * (q >> 2) + 4 * lh; }

// Flat-stream weight gradient of the stride-1 convolutions (conv_bf16_stream.hip); slabs in the layout of
// bf16_wgrad_reduce_kernel: ws[nsplit][k*k][Cin][Cout] fp32.  *nsplit receives the slab count.
bool yh_bf16_wgrad_stream_ok(int W, int Cin, int Cout, int k, int s);
int64_t yh_bf16_wgrad_stream_ws(int B, int H, int W, int Cin, int Cout, int k);
int yh_bf16_wgrad_stream(const void *x, int ldx, const void *dy, int lddy, float *ws, int64_t ws_floats, int B, int H, int W, int Cin,
                         int Cout, int k, int *nsplit, hipStream_t st);

// Flat-stream forward / backward-data of the stride-1 layers (K = 16 / 32 / 64 / 128 streamed channels, bf16 output).
// stats: [yh_bf16_fstream_blocks(...)][2][N] BatchNorm partial rows (ONE per persistent workgroup).
bool yh_bf16_fstream_supported(int B, int H, int W, int K, int N, int k, int s);   // the kernel can run the problem
bool yh_bf16_fstream_ok(int B, int H, int W, int K, int N, int k, int s);          // ... and measured faster than the gather GEMM
int yh_bf16_fstream_blocks(int B, int H, int W, int K, int N, int k);
int yh_bf16_fstream(const void *in, const void *in2, int ksplit, int ldi, const void *w, int ldw, const float *bias, void *out, int ldo,
                    float *stats, int B, int H, int W, int K, int N, int k, int accumulate, const int *tap_dy, const int *tap_dx,
                    const int *tap_w, hipStream_t st);
