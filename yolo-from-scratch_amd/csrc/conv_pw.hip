// Pointwise (1x1, stride 1) convolution kernels whose MFMA operands are loaded straight into the operand registers
// (no LDS in the main loop), gfx950, NHWC fp32.
//
// Backward-weight:  dW[ci][co] = sum_p x[p][ci] * dy[p][co]  -- rows = ci, cols = co, k = pixels.  One
// v_mfma_f32_32x32x2_f32 k-step consumes a PAIR of pixels: lane (c = lane & 31, h = lane >> 5) supplies pixel 2s+h.
// A lane loads XV consecutive input channels and DV consecutive output channels of its pixel with one vector load
// each; element e of the vector is the operand of row/column tile e, i.e. row r of row tile i is channel XV*r + i
// (a fixed permutation, undone when the slab is written).  XV x DV accumulator tiles per wave, (XV + DV) loads per
// XV*DV MFMAs.  The four waves of a workgroup walk disjoint pixel ranges of the same (ci, co) tile and are summed
// through LDS; workgroups write [Cin][Cout] slabs that a fixed-order reduction adds up (bitwise reproducible).
#include "common.h"
#include <stdlib.h>
#ifdef YH_PW_STAMPS
#include <stdio.h>
#include <vector>
#endif

namespace {

template <int V>
struct Vec;
template <>
struct Vec<1> { typedef float type; };
template <>
struct Vec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <>
struct Vec<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int V>
__device__ __forceinline__ typename Vec<V>::type buf_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    if constexpr (V == 1) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    else if constexpr (V == 2) return __builtin_bit_cast(typename Vec<2>::type, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    else return __builtin_bit_cast(typename Vec<4>::type, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
template <int V>
__device__ __forceinline__ float vget(const typename Vec<V>::type &v, int e) {
    if constexpr (V == 1) return v;
    else return v[e];
}

struct PwW {
    const float *x, *dy;
    const float *icoef;         // input prologue table of x ([scale | shift | gate] rows, icoef_ld apart; yh_prologue), or null
    int icoef_ld;
    float *ws;
    int ldx, lddy, Cin, Cout;
    int M, pps;                 // pixels, pixels per split (multiple of 64)
    unsigned x_bytes, dy_bytes;
};

template <int XV, int DV>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(const PwW g) {
    typedef typename Vec<XV>::type xv_t;
    typedef typename Vec<DV>::type dv_t;
    constexpr int SPS = 8;                     // k-steps (pixel pairs) per pipeline stage
    constexpr int RT = 32 * XV, CT = 32 * DV;   // channels of a workgroup tile
    extern __shared__ __attribute__((aligned(16))) float smem[];   // epilogue only: [4][32][CT]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int ci0 = blockIdx.y * RT, co0 = blockIdx.z * CT;
    // pixel range of this wave: a quarter of the split's range
    const int p_begin = blockIdx.x * g.pps + wave * (g.pps / 4);
    const int p_end = min(p_begin + g.pps / 4, g.M);

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)g.x, 0, g.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)g.dy, 0, g.dy_bytes, 0x00020000);
    const unsigned xlane = (unsigned)(ci0 + XV * lr) * 4u, dlane = (unsigned)(co0 + DV * lr) * 4u;
    const unsigned ldx4 = (unsigned)g.ldx * 4u, ldd4 = (unsigned)g.lddy * 4u;
    int p = p_begin + lh;                      // this lane's pixel of the next k-step

    // input prologue: a lane's XV channels are the same for every pixel -- coefficients loaded once.  Pixels past the range load
    // zeros on BOTH operands, so whatever the prologue makes of an x zero meets a dY zero.
    const bool act = g.icoef != nullptr;                   // workgroup-uniform
    float psc[XV], psh[XV], pgt[XV];
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        int ch = ci0 + XV * lr + i;
        ch = ch < g.Cin ? ch : g.Cin - 1;
        psc[i] = act ? g.icoef[ch] : 1.f; psh[i] = act ? g.icoef[g.icoef_ld + ch] : 0.f; pgt[i] = act ? g.icoef[2 * g.icoef_ld + ch] : 0.f;
    }
    xv_t XA[SPS], XB[SPS];
    dv_t DA[SPS], DB[SPS];
    auto load_stage = [&](xv_t (&X)[SPS], dv_t (&D)[SPS]) {
#pragma unroll
        for (int q = 0; q < SPS; ++q) {
            const unsigned inv = (unsigned)(p >= p_end) << 31;     // bit 31: beyond num_records -> 0
            X[q] = buf_load<XV>(rx, ((unsigned)p * ldx4 + xlane) | inv);
            D[q] = buf_load<DV>(rd, ((unsigned)p * ldd4 + dlane) | inv);
            p += 2;
        }
    };

    f32x16 acc[XV][DV];
#pragma unroll
    for (int i = 0; i < XV; ++i)
#pragma unroll
        for (int j = 0; j < DV; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto compute = [&](const xv_t (&X)[SPS], const dv_t (&D)[SPS]) {
#pragma unroll
        for (int q = 0; q < SPS; ++q)
#pragma unroll
            for (int i = 0; i < XV; ++i) {
                float xv = vget<XV>(X[q], i);
                if (act) xv = yh_prologue(xv, psc[i], psh[i], pgt[i]);
#pragma unroll
                for (int j = 0; j < DV; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv, vget<DV>(D[q], j), acc[i][j], 0, 0, 0);
            }
    };

    const int nstages = (g.pps / 4) / (2 * SPS);
    load_stage(XA, DA);
    int s = 0;
    for (; s + 2 <= nstages; s += 2) {
        load_stage(XB, DB);
        __builtin_amdgcn_sched_barrier(0);
        compute(XA, DA);
        __builtin_amdgcn_sched_barrier(0);
        load_stage(XA, DA);                    // past the range: zeros
        __builtin_amdgcn_sched_barrier(0);
        compute(XB, DB);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (s < nstages) compute(XA, DA);

    // ---- sum the four waves through LDS (one row tile per pass), undo the channel permutation, write the slab ------
    float *slab = g.ws + (size_t)blockIdx.x * g.Cin * g.Cout;
#pragma unroll
    for (int i = 0; i < XV; ++i) {
        if (i) __syncthreads();
        float *T = smem + wave * 32 * CT;
#pragma unroll
        for (int j = 0; j < DV; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                T[row * CT + DV * lr + j] = acc[i][j][r];
            }
        __syncthreads();
        for (int e = t; e < 32 * CT; e += 256) {
            int col = e % CT, row = e / CT;
            int ci = ci0 + XV * row + i, co = co0 + col;
            if (ci < g.Cin && co < g.Cout)
                slab[(size_t)ci * g.Cout + co] = (smem[e] + smem[32 * CT + e]) + (smem[2 * 32 * CT + e] + smem[3 * 32 * CT + e]);
        }
    }
}

// ws [nsplit][Cin][Cout] -> dw [Cout][Cin] (OIHW, 1x1): 16 split-lanes each add every 16th slab, lanes combined in order.
__global__ __launch_bounds__(256) void pw_wgrad_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dw, int nsplit,
                                                              int Cin, int Cout) {
    __shared__ float red[16][17];
    const int n = Cin * Cout;
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + e;
    float s = 0.f;
    if (i < n)
        for (int k = sl; k < nsplit; k += 16) s += ws[(size_t)k * n + i];
    red[sl][e] = s;
    __syncthreads();
    if (sl == 0 && i < n) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += red[k][e];
        int co = i % Cout, ci = i / Cout;
        dw[(size_t)co * Cin + ci] = tot;
    }
}

// vmax_x / vmax_d: widest vector load the operand's address and ld allow (1, 2 or 4 floats)
int pw_plan(PwW &g, int &nsplit, int &XV, int &DV, int64_t M, int Cin, int Cout, int vmax_x, int vmax_d) {
    YH_REQUIRE(M > 0 && M < (1ll << 30) && Cin > 0 && Cout > 0, "conv_pw_bwd_weight: bad shape");
    g.M = (int)M; g.Cin = Cin; g.Cout = Cout;
    XV = Cin <= 32 ? 1 : 2;                       // <= 32 channels: one tile, no wasted MFMA rows / columns
    DV = Cout <= 32 ? 1 : (Cout % 128 == 0 ? 4 : 2);
    while (XV > vmax_x) XV >>= 1;
    while (DV > vmax_d) DV >>= 1;
    int pairs = cdiv(Cin, 32 * XV) * cdiv(Cout, 32 * DV);
    // 256 workgroups measured best on every 1x1 layer of the network (slab traffic grows with the split count)
    constexpr int target = 256;
    nsplit = target / pairs;
    if (nsplit < 1) nsplit = 1;
    int pps = cdiv(cdiv((int)M, nsplit), 64) * 64;
    if (pps < 256) pps = 256;
    nsplit = cdiv((int)M, pps);
    g.pps = pps;
    return 0;
}

int vec_width(const void *p, int ld) {
    uintptr_t a = (uintptr_t)p;
    if (ld % 4 == 0 && (a & 15) == 0) return 4;
    if (ld % 2 == 0 && (a & 7) == 0) return 2;
    return 1;
}

}  // namespace

extern "C" int64_t yh_conv_pw_bwd_weight_ws(int64_t M, int Cin, int Cout) {
    int64_t need = 0;                              // the split count depends on the vector widths the views allow
    for (int vx = 1; vx <= 4; vx <<= 1)
        for (int vd = 1; vd <= 4; vd <<= 1) {
            PwW g{};
            int nsplit, XV, DV;
            if (pw_plan(g, nsplit, XV, DV, M, Cin, Cout, vx, vd)) return -1;
            if ((int64_t)nsplit * Cin * Cout > need) need = (int64_t)nsplit * Cin * Cout;
        }
    return need;
}

extern "C" int yh_conv_pw_bwd_weight_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *dy, int lddy, float *dw,
                                         float *ws, int64_t ws_floats, int64_t M, int Cin, int Cout, void *stream);
extern "C" int yh_conv_pw_bwd_weight(const float *x, int ldx, const float *dy, int lddy, float *dw, float *ws, int64_t ws_floats,
                                     int64_t M, int Cin, int Cout, void *stream) {
    return yh_conv_pw_bwd_weight_act(x, ldx, nullptr, 0, dy, lddy, dw, ws, ws_floats, M, Cin, Cout, stream);
}
extern "C" int yh_conv_pw_bwd_weight_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *dy, int lddy, float *dw,
                                         float *ws, int64_t ws_floats, int64_t M, int Cin, int Cout, void *stream) {
    YH_REQUIRE(x && dy && dw && ws && ldx >= Cin && lddy >= Cout && (!icoef || icoef_ld >= Cin), "conv_pw_bwd_weight: bad argument");
    PwW g{};
    g.icoef = icoef; g.icoef_ld = icoef_ld;
    int nsplit, XV, DV;
    int rc = pw_plan(g, nsplit, XV, DV, M, Cin, Cout, vec_width(x, ldx), vec_width(dy, lddy));
    if (rc) return rc;
    YH_REQUIRE(ws_floats >= (int64_t)nsplit * Cin * Cout, "conv_pw_bwd_weight: workspace too small");
    YH_REQUIRE(((M - 1) * ldx + Cin) * 4 < (1ll << 31) && ((M - 1) * lddy + Cout) * 4 < (1ll << 31),
               "conv_pw_bwd_weight: views must span less than 2 GiB");
    g.x = x; g.dy = dy; g.ws = ws; g.ldx = ldx; g.lddy = lddy;
    g.x_bytes = (unsigned)(((M - 1) * ldx + Cin) * 4);
    g.dy_bytes = (unsigned)(((M - 1) * lddy + Cout) * 4);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(nsplit, cdiv(Cin, 32 * XV), cdiv(Cout, 32 * DV));
    const bool shared_cu = yh_tls_side_lane != 0;              // next to the main lane: one workgroup per CU (common.h)
    const size_t smem = shared_cu ? YH_SIDE_LDS_BYTES : (size_t)4 * 32 * 32 * DV * sizeof(float);
#define YH_PW_LAUNCH(xv, dv)                                                                                   \
    do {                                                                                                       \
        if (shared_cu) if (int rc2 = yh_ensure_dyn_smem((const void *)pw_wgrad_kernel<xv, dv>, smem)) return rc2; \
        hipLaunchKernelGGL((pw_wgrad_kernel<xv, dv>), grid, dim3(256), smem, st, g);                           \
    } while (0)
    if (XV == 2 && DV == 4) YH_PW_LAUNCH(2, 4);
    else if (XV == 2 && DV == 2) YH_PW_LAUNCH(2, 2);
    else if (XV == 2) YH_PW_LAUNCH(2, 1);
    else if (DV == 4) YH_PW_LAUNCH(1, 4);
    else if (DV == 2) YH_PW_LAUNCH(1, 2);
    else YH_PW_LAUNCH(1, 1);
#undef YH_PW_LAUNCH
    YH_CHECK_LAUNCH("pw_wgrad");
    int n = Cin * Cout;
    hipLaunchKernelGGL(pw_wgrad_reduce_kernel, dim3(cdiv(n, 16)), dim3(256), 0, st, ws, dw, nsplit, Cin, Cout);
    YH_CHECK_LAUNCH("pw_wgrad_reduce");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Forward / backward-data of a pointwise convolution:  out[p][n] (+)= sum_k in[p][k] * W[k][n] (+ bias).
// Rows = pixels, cols = channels.  Lane (r = lane & 31, q = lane >> 5) loads channels k0+4q .. k0+4q+3 of pixel r with
// one float4 -- the A operand of four consecutive MFMAs (k-group = lane half) -- and the matching B fragments with one
// float4 each from the k-quad interleaved weights Wq[K/4][ldw][4] (same layout as the Winograd U).  No LDS, no
// barrier in the loop; two register sets, the loads of chunk c+1 are pinned ahead of the MFMAs of chunk c.
// A wave owns TM x 32 pixels x NT x 32 channels; the four waves of a workgroup own consecutive pixel groups (their B
// fragments are the same lines: L1 hits).  Epilogue as the other conv kernels: 128-byte row stores, bias /
// accumulate, per-workgroup BatchNorm partial sums.  K may come from two tensors (sibling-pair backward-data).
namespace {

struct PwG {
    const float *in, *in2, *Wq, *bias, *bias2;
    float *out, *stats, *out2, *stats2;     // out2 != NULL: columns [N1, N) go to a second tensor (sibling convolutions that
    int ldi, ldw, ldo, ldo2;                // share their input: one read of x, one GEMM with N = N1 + N2)
    int M, K, K1, N, N1;        // K1: channels taken from `in` (the rest, K - K1, from `in2`); N1 = N when out2 == NULL
    int accumulate;
    unsigned in_bytes, in2_bytes;
    const float *res;           // inference epilogue (tiled kernel only): SiLU on (acc + bias), + residual, x2 upsample on write
#ifdef YH_PW_STAMPS
    unsigned long long *dbg;    // diagnostic build only: per-workgroup phase stamps (tools/pw_probe.py)
#endif
    int ldr, act, up2, H, W;    // H, W: image size (up2 needs the pixel's row / column)
    const float *icoef;         // input prologue table of `in` ([scale | shift | gate] rows, icoef_ld apart; yh_prologue), or null:
    int icoef_ld;               // tiled and streaming kernels, single-source K only
};

template <int TM, int NT>
__global__ __launch_bounds__(256) void pw_gemm_kernel(const PwG g) {
    __shared__ float red[4][32 * NT][2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int p0 = (blockIdx.x * 4 + wave) * 32 * TM, n0 = blockIdx.y * 32 * NT;

    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void *)g.in, 0, g.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void *)(g.in2 ? g.in2 : g.in), 0, g.in2 ? g.in2_bytes : 0u, 0x00020000);
    unsigned aoff[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int p = p0 + i * 32 + lr;
        aoff[i] = ((unsigned)p * (unsigned)g.ldi + 4u * lh) * 4u | ((unsigned)(p >= g.M) << 31);
    }
    const float *wb[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int n = n0 + j * 32 + lr;
        wb[j] = g.Wq + ((size_t)lh * g.ldw + (n < g.ldw ? n : 0)) * 4;     // columns past N are never stored
    }
    const size_t wchunk = (size_t)2 * g.ldw * 4;

    f32x4 aA[TM], bA[NT], aB[TM], bB[NT];
    f32x16 acc[TM][NT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto compute = [&](const f32x4 (&a)[TM], const f32x4 (&b)[NT]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    };
    // one K segment (all of it from one tensor): chunks [c0, c0 + nch) of the weight rows, channels from 0 of `r`
    auto segment = [&](const __amdgpu_buffer_rsrc_t r, int c0, int nch) {
        auto load = [&](int c, f32x4 (&a)[TM], f32x4 (&b)[NT]) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = buf_load<4>(r, aoff[i] + (unsigned)c * 32u);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = *(const f32x4 *)(wb[j] + (size_t)(c0 + c) * wchunk);
        };
        load(0, aA, bA);
        int c = 0;
        for (; c + 2 <= nch; c += 2) {
            load(c + 1, aB, bB);
            __builtin_amdgcn_sched_barrier(0);
            compute(aA, bA);
            __builtin_amdgcn_sched_barrier(0);
            load(c + 2 < nch ? c + 2 : c, aA, bA);
            __builtin_amdgcn_sched_barrier(0);
            compute(aB, bB);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c < nch) compute(aA, bA);
    };
    segment(r1, 0, g.K1 / 8);
    if (g.K > g.K1) segment(r2, g.K1 / 8, (g.K - g.K1) / 8);

    // ---- epilogue ---------------------------------------------------------------------------------------------
    float csum[NT], csq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 32 + lr;
        const bool nok = n < g.N, second = n >= g.N1;
        const int nl = second ? n - g.N1 : n;                       // column inside its own tensor
        const gfloat *bp = second ? yh_global(g.bias2) : yh_global(g.bias);
        const float bias = (bp && nok) ? bp[nl] : 0.f;
        gfloat *const ob = (second ? yh_global(g.out2) : yh_global(g.out)) + nl;
        const int ldo = second ? g.ldo2 : g.ldo;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (nok && p < g.M) {
                    gfloat *o = ob + (size_t)p * ldo;
                    float v = acc[i][j][r] + bias;
                    if (g.accumulate) v += *o;
                    if (g.act | g.up2 | (g.res != nullptr)) {      // inference form (uniform branch)
                        if (g.act) v = v * yh_sigmoid(v);
                        if (g.res) v += g.res[(size_t)p * g.ldr + n];
                        if (g.up2) {
                            const int q = p / g.W, x = p - q * g.W, b = q / g.H, y = q - b * g.H;
                            gfloat *u = ob + (((size_t)b * 2 * g.H + 2 * y) * 2 * g.W + 2 * x) * ldo;
                            const size_t rs = (size_t)(2 * g.W) * ldo;
                            u[0] = v; u[ldo] = v; u[rs] = v; u[rs + ldo] = v;
                            continue;
                        }
                    }
                    *o = v;
                    s += v;
                    q += v * v;
                }
            }
        csum[j] = s;
        csq[j] = q;
    }
    if (g.stats) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float s = csum[j] + __shfl_xor(csum[j], 32), q = csq[j] + __shfl_xor(csq[j], 32);
            if (lh == 0) { red[wave][j * 32 + lr][0] = s; red[wave][j * 32 + lr][1] = q; }
        }
        __syncthreads();
        if (t < 32 * NT && n0 + t < g.N) {
            float a0 = (red[0][t][0] + red[1][t][0]) + (red[2][t][0] + red[3][t][0]);
            float a1 = (red[0][t][1] + red[1][t][1]) + (red[2][t][1] + red[3][t][1]);
            const int n = n0 + t;
            gfloat *sp = n >= g.N1 ? yh_global(g.stats2) : yh_global(g.stats);   // each tensor has its own [blocks][2][C] partials
            const int C = n >= g.N1 ? g.N - g.N1 : g.N1, nl = n >= g.N1 ? n - g.N1 : n;
            sp[((size_t)blockIdx.x * 2 + 0) * C + nl] = a0;
            sp[((size_t)blockIdx.x * 2 + 1) * C + nl] = a1;
        }
    }
}

// The same GEMM with the A operand staged through LDS (training-path layers with K >= 64 that do not stream).  PMC on the
// register-direct kernel above (128 -> 128 at 40x40, 75 us = 45 TFLOP/s): waves sit in s_waitcnt 57 % of their cycles, the L1
// is stalled on pending fills, and L2 -> L1 traffic is 3.5x the operands -- a lane's 16-byte A loads touch 32 different
// cache lines per instruction and every wave streams the full weight panel by itself.  Here a workgroup owns 128 pixels x
// (32 WN) channels: the A chunk (128 pixels x up to 128 channels) is read ONCE with row-contiguous 16-byte loads (next chunk
// prefetched into registers under the MFMAs) and parked in LDS with a 132-float row stride (conflict-free ds_read_b128 of a
// lane's four consecutive k); wave (wm, wn) multiplies TMW x 32 rows by ITS OWN 32 columns, so each weight fragment is
// fetched by exactly one wave of the workgroup.  Same operand maps, k order per 8-channel group, epilogue and partial-sum
// contract as pw_gemm_kernel.
template <int TMW, int WN, bool ACT = false>
__global__ __launch_bounds__(256, 2) void pw_tile_kernel(const PwG g) {
    constexpr int WM = 4 / WN, BM = 32 * TMW * WM, KC = 128, LDA = KC + 4, NP = BM * (KC / 4) / 256;
    extern __shared__ __attribute__((aligned(16))) float pw_as[];     // [BM][LDA]; reused for the partial-sum exchange
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int n = (blockIdx.y * WN + wn) * 32 + lr;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void *)g.in, 0, g.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void *)(g.in2 ? g.in2 : g.in), 0, g.in2 ? g.in2_bytes : 0u, 0x00020000);
    const int nch1 = (g.K1 + KC - 1) / KC, nch = nch1 + (g.K - g.K1 + KC - 1) / KC;
    const int ntiles = (g.M + BM - 1) / BM;

    // staging plan: piece j of this thread = 16 bytes of row (t >> 5) + 8 j, channels 4 (t & 31) .. + 3 of the chunk
    const int col4 = t & 31, row0 = t >> 5;
    f32x4 rx[NP];
    // quarter q (0..3) of the loads of chunk ci of `tile`: issued a quarter at a time between the MFMA groups of the
    // previous chunk, so that the weight fragments those groups wait for are never queued behind a burst of A loads
    // (vmcnt retires in order)
    auto fetch_q = [&](int tile, int ci, int q) {
        const bool second = ci >= nch1;
        const int kb = (second ? ci - nch1 : ci) * KC;
        const int kc = (second ? g.K - g.K1 : g.K1) - kb;                   // channels left in this tensor (>= 1)
        const unsigned dead = (unsigned)(4 * col4 >= kc) << 31;
        const int p0 = tile * BM + row0;
        const unsigned base = ((unsigned)p0 * (unsigned)g.ldi + 4u * col4 + (unsigned)kb) * 4u, rstep = 8u * (unsigned)g.ldi * 4u;
#pragma unroll
        for (int jj = 0; jj < NP / 4; ++jj) {
            const int j = q * (NP / 4) + jj;
            const unsigned off = (base + (unsigned)j * rstep) | dead | ((unsigned)(p0 + 8 * j >= g.M) << 31);
            rx[j] = second ? buf_load<4>(r2, off) : buf_load<4>(r1, off);
        }
    };
    auto fetch = [&](int tile, int ci) {
        fetch_q(tile, ci, 0); fetch_q(tile, ci, 1); fetch_q(tile, ci, 2); fetch_q(tile, ci, 3);
    };
    // input prologue: the chunk's coefficients for this thread's channel quad come from the table when the chunk is parked (three
    // 16-byte L1 / L2 hits per 128-channel chunk).  Rows past M and channels past K are parked as whatever the prologue makes of
    // their zeros: those rows are never stored or summed, those channels never multiplied (ns = kc >> 3 groups).
    auto park = [&](int kb) {
        if constexpr (ACT) {
            const float *cp = g.icoef + kb + 4 * col4;
            const bool live = kb + 4 * col4 < g.K1;
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 sc = live ? *(const f32x4 *)cp : z4, sh = live ? *(const f32x4 *)(cp + g.icoef_ld) : z4,
                        gt = live ? *(const f32x4 *)(cp + 2 * g.icoef_ld) : z4;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                f32x4 v = rx[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = yh_prologue(v[e], sc[e], sh[e], gt[e]);
                *(f32x4 *)(pw_as + (row0 + 8 * j) * LDA + 4 * col4) = v;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) *(f32x4 *)(pw_as + (row0 + 8 * j) * LDA + 4 * col4) = rx[j];
    };

    const float *wbase = g.Wq + ((size_t)lh * g.ldw + (n < g.ldw ? n : 0)) * 4;      // columns past N are never stored
    const size_t wstep = (size_t)2 * g.ldw * 4;                                        // one 8-channel group of weight rows
    const float *abase = pw_as + (wm * TMW * 32 + lr) * LDA + 4 * lh;

    // per-column epilogue state
    const bool nok = n < g.N, second = n >= g.N1;
    const int nl = second ? n - g.N1 : n;
    const gfloat *bp = second ? yh_global(g.bias2) : yh_global(g.bias);
    const float bias = (bp && nok) ? bp[nl] : 0.f;
    gfloat *const ob = (second ? yh_global(g.out2) : yh_global(g.out)) + nl;
    const int ldo = second ? g.ldo2 : g.ldo;
    const int nb0 = (blockIdx.y * WN + wn) * 32;                            // this wave's first column
    const bool cols_whole = nb0 + 32 <= g.N && (nb0 >= g.N1 || nb0 + 32 <= g.N1);          // wave-uniform
    const bool sec_u = nb0 >= g.N1;                                          // scalar copies of the per-lane selections
    const int ldo_u = sec_u ? g.ldo2 : g.ldo;
    gfloat *const ob_u = sec_u ? yh_global(g.out2) : yh_global(g.out);

    // weight fragments: a ring of PD groups (8 channels each) kept PD groups ahead of the MFMAs, running on across A chunks
    // and (wrapping to group 0) across tiles; vmcnt retires in order, so the ring is filled BEFORE the A loads are issued
    constexpr int PD = 4;
    const int G = g.K >> 3;                                                 // K % 32 == 0 here (launcher)
    f32x4 bq[PD];
#pragma unroll
    for (int u = 0; u < PD; ++u) bq[u] = *(const f32x4 *)(wbase + (size_t)u * wstep);

    // A persistent workgroup walks pixel tiles blockIdx.x, + gridDim.x, ...: the first A chunk of the NEXT tile is already in
    // flight while the last chunk of the current one is multiplied and while its epilogue stores drain, so load, MFMA and
    // store phases of the chip overlap instead of alternating (one tile per workgroup ran 7.5 + 13.6 + 7 us per round of
    // 512 workgroups on 128 -> 128 at 160x160: every workgroup loading, then multiplying, then storing at the same time).
    float cs = 0.f, cq = 0.f;                                               // BatchNorm sums of ALL tiles of this workgroup: one row
    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile, 0);
    for (; tile < ntiles; tile += gridDim.x) {
#ifdef YH_PW_STAMPS
        const unsigned long long st0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long st1 = 0, st2 = 0;
#endif
        const int m0 = tile * BM;
        f32x16 acc[TMW];
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        int gs = 0;                                                         // first weight group of the current chunk
        for (int ci = 0; ci < nch; ++ci) {
            const bool seg2 = ci >= nch1;
            const int kb = (seg2 ? ci - nch1 : ci) * KC;
            int kc = (seg2 ? g.K - g.K1 : g.K1) - kb;
            if (kc > KC) kc = KC;
            __syncthreads();                                                // the previous chunk's fragments / sums are consumed
            park(kb);
            __syncthreads();
#ifdef YH_PW_STAMPS
            if (ci == 0) st1 = __builtin_amdgcn_s_memtime();
#endif
            // the next chunk (of this tile, or the first one of the next tile) is fetched under the MFMAs below
            const bool more = ci + 1 < nch || tile + (int)gridDim.x < ntiles;
            const int nt_tile = ci + 1 < nch ? tile : tile + (int)gridDim.x, nt_ci = ci + 1 < nch ? ci + 1 : 0;
            const int ns = kc >> 3;
            f32x4 a0[TMW], a1[TMW];
#pragma unroll
            for (int i = 0; i < TMW; ++i) a0[i] = *(const f32x4 *)(abase + i * 32 * LDA);
            // one group: take the weight fragment loaded PD groups ago, refill its ring slot, read the NEXT group's A
            // fragments from LDS, then 4 TMW MFMAs -- everything issued ahead of the MFMAs that hide it
            auto group = [&](int sc, int u, const f32x4 (&ac)[TMW], f32x4 (&an)[TMW]) {
                const f32x4 b = bq[u];
                int nx = gs + sc + PD;
                if (nx >= G) nx -= G;
                bq[u] = *(const f32x4 *)(wbase + (size_t)nx * wstep);
                const int sn = sc + 1 < ns ? sc + 1 : sc;
#pragma unroll
                for (int i = 0; i < TMW; ++i) an[i] = *(const f32x4 *)(abase + i * 32 * LDA + 8 * sn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TMW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[i][e], b[e], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            // vmcnt retires in order: a weight fragment requested after a burst of A loads is not "there" before they are.
            // Inside a tile the next chunk's A loads go out at once at the chunk's start (one possible stall per chunk, PD
            // groups later); the next TILE's first chunk goes out after the last MFMA group, ahead of the epilogue, whose
            // stores need no waiting -- stamps: loads spread between the groups cost 1.65x the MFMA time of the loop.
            if (more && ci + 1 < nch) fetch(nt_tile, nt_ci);
            for (int s8 = 0; s8 < ns; s8 += PD) {                           // ns % PD == 0: K segments are multiples of 32 here
                group(s8 + 0, 0, a0, a1);
                group(s8 + 1, 1, a1, a0);
                group(s8 + 2, 2, a0, a1);
                group(s8 + 3, 3, a1, a0);
            }
            if (more && ci + 1 == nch) fetch(nt_tile, nt_ci);
            gs += ns;
        }

#ifdef YH_PW_STAMPS
        st2 = __builtin_amdgcn_s_memtime();
#endif
        // ---- epilogue: bias / accumulate, 128-byte row stores, column sums ---------------------------------------------
        if (cols_whole && m0 + BM <= g.M) {
            // full tile, all 32 columns in one tensor: no per-element tests; row offsets are compile-time multiples of ldo
            gfloat *const o0 = ob_u + (size_t)(m0 + wm * TMW * 32) * ldo_u + ((size_t)(4 * lh) * ldo_u + nl);
            if (!g.accumulate) {
#pragma unroll
                for (int i = 0; i < TMW; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][r] + bias;
                        o0[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldo_u] = v;
                        cs += v;
                        cq += v * v;
                    }
            } else {
#pragma unroll
                for (int i = 0; i < TMW; ++i) {
                    float old[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[r] = o0[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldo_u];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][r] + bias + old[r];
                        o0[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldo_u] = v;
                        cs += v;
                        cq += v * v;
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = m0 + (wm * TMW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (nok && p < g.M) {
                        gfloat *o = ob + (size_t)p * ldo;
                        float v = acc[i][r] + bias;
                        if (g.accumulate) v += *o;
                        *o = v;
                        cs += v;
                        cq += v * v;
                    }
                }
        }
#ifdef YH_PW_STAMPS
        if (g.dbg && t == 0 && blockIdx.y == 0) {
            unsigned long long *d = g.dbg + (size_t)tile * 6;
            d[0] = st0; d[1] = st1; d[2] = st2; d[3] = __builtin_amdgcn_s_memtime(); d[4] = rt0; d[5] = __builtin_amdgcn_s_memrealtime();
        }
#endif
    }
    if (g.stats) {
        float(*red)[32][2] = (float(*)[32][2])pw_as;                        // [4 waves][32][2]
        cs += __shfl_xor(cs, 32);
        cq += __shfl_xor(cq, 32);
        __syncthreads();                                                    // all waves are done with the A tile
        if (lh == 0) { red[wave][lr][0] = cs; red[wave][lr][1] = cq; }
        __syncthreads();
        if (t < 32 * WN) {
            const int cw = t >> 5, cl = t & 31, nn = (blockIdx.y * WN + cw) * 32 + cl;
            if (nn < g.N) {
                float a0s = 0.f, a1s = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) { a0s += red[w * WN + cw][cl][0]; a1s += red[w * WN + cw][cl][1]; }
                gfloat *sp = nn >= g.N1 ? yh_global(g.stats2) : yh_global(g.stats);    // each tensor has its own [workgroups][2][C] partials
                const int C = nn >= g.N1 ? g.N - g.N1 : g.N1, nnl = nn >= g.N1 ? nn - g.N1 : nn;
                sp[((size_t)blockIdx.x * 2 + 0) * C + nnl] = a0s;
                sp[((size_t)blockIdx.x * 2 + 1) * C + nnl] = a1s;
            }
        }
    }
}

// Small-K, small-N, many-pixel pointwise layers (the 160^2 / 80^2 1x1 convs: K <= 64, memory-bound): a STREAMING form of the
// same GEMM.  All B fragments of the layer (K/8 x NT float4 per lane) are loaded once and stay in registers; a fixed grid of
// workgroups walks the pixel groups (32 pixels per wave and trip), prefetching the next group's A operand while the
// current one is multiplied and stored.  No per-tile prologue, one BatchNorm partial row per workgroup (<= 1024 rows to
// finalize instead of M / 128).
template <int NT, int KC>
__global__ __launch_bounds__(256) void pw_stream_kernel(const PwG g) {
    __shared__ float red[4][32 * NT][2];
    __shared__ __attribute__((aligned(16))) float ctab[3][8 * KC];      // input prologue table of the layer's K = 8 KC channels
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void *)g.in, 0, g.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void *)(g.in2 ? g.in2 : g.in), 0, g.in2 ? g.in2_bytes : 0u, 0x00020000);
    const int kc1 = g.K1 >> 3;

    f32x4 b[KC][NT];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = j * 32 + lr;
            b[c][j] = *(const f32x4 *)(g.Wq + ((size_t)(2 * c + lh) * g.ldw + (n < g.ldw ? n : 0)) * 4);
        }
    // per-column epilogue state (same semantics as pw_gemm_kernel)
    bool nok[NT];
    float bias[NT], csum[NT], csq[NT];
    gfloat *ob[NT];
    int ldo[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + lr;
        nok[j] = n < g.N;
        const bool second = n >= g.N1;
        const int nl = second ? n - g.N1 : n;
        const gfloat *bp = second ? yh_global(g.bias2) : yh_global(g.bias);
        bias[j] = (bp && nok[j]) ? bp[nl] : 0.f;
        ob[j] = (second ? yh_global(g.out2) : yh_global(g.out)) + nl;
        ldo[j] = second ? g.ldo2 : g.ldo;
        csum[j] = csq[j] = 0.f;
    }

    const bool plain = 32 * NT <= g.N;                     // wave-uniform: every lane's column exists
    const int ngroups = (g.M + 31) >> 5, stride = gridDim.x * 4;
    auto load = [&](int grp, f32x4 (&a)[KC]) {
        const int p = grp * 32 + lr;
        const unsigned off = (((unsigned)p * (unsigned)g.ldi + 4u * lh) * 4u) | ((unsigned)(p >= g.M || grp >= ngroups) << 31);
#pragma unroll
        for (int c = 0; c < KC; ++c) a[c] = c < kc1 ? buf_load<4>(r1, off + (unsigned)c * 32u) : buf_load<4>(r2, off + (unsigned)(c - kc1) * 32u);
    };
    auto compute = [&](int grp, const f32x4 (&a)[KC]) {
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][e], b[c][j][e], acc[j], 0, 0, 0);
        const int p0 = grp * 32;
        if (plain && p0 + 32 <= g.M) {
            // all 32 pixels and every column valid, no BatchNorm table: no per-element tests, one 64-bit add per row
            // (measured on the tested form: ~25 instructions per element, more VALU time than the group's MFMAs)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                gfloat *o = ob[j] + (size_t)(p0 + 4 * lh) * ldo[j];
                const size_t step = (size_t)ldo[j];
                if (!g.accumulate) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[j][r] + bias[j];
                        *o = v;
                        o += (r & 3) == 3 ? 5 * step : step;
                        csum[j] += v;
                        csq[j] += v * v;
                    }
                } else {
                    float old[16];
                    gfloat *q = o;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { old[r] = *q; q += (r & 3) == 3 ? 5 * step : step; }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[j][r] + bias[j] + old[r];
                        *o = v;
                        o += (r & 3) == 3 ? 5 * step : step;
                        csum[j] += v;
                        csq[j] += v * v;
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (nok[j] && p < g.M) {
                    gfloat *o = ob[j] + (size_t)p * ldo[j];
                    float v = acc[j][r] + bias[j];
                    if (g.accumulate) v += *o;
                    *o = v;
                    csum[j] += v;
                    csq[j] += v * v;
                }
            }
    };
    // input prologue (the layers this kernel runs are HBM-bound: the sigmoid work of one wave hides under the loads of the others).
    // Pixels past M load zeros and come out as whatever the prologue makes of them: their rows are never stored or summed.
    const bool act = g.icoef != nullptr;                   // workgroup-uniform
    if (act) {
        for (int i = t; i < 3 * 8 * KC; i += 256) ctab[i / (8 * KC)][i % (8 * KC)] = g.icoef[(i / (8 * KC)) * g.icoef_ld + i % (8 * KC)];
        __syncthreads();
    }
    auto prologue = [&](f32x4 (&a)[KC]) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const f32x4 sc = *(const f32x4 *)&ctab[0][8 * c + 4 * lh], sh = *(const f32x4 *)&ctab[1][8 * c + 4 * lh],
                        gt = *(const f32x4 *)&ctab[2][8 * c + 4 * lh];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[c][e] = yh_prologue(a[c][e], sc[e], sh[e], gt[e]);
        }
    };
    f32x4 aA[KC], aB[KC];
    int grp = blockIdx.x * 4 + wave;
    load(grp, aA);
    for (; grp < ngroups; grp += 2 * stride) {
        load(grp + stride, aB);
        __builtin_amdgcn_sched_barrier(0);
        if (act) prologue(aA);
        compute(grp, aA);
        __builtin_amdgcn_sched_barrier(0);
        load(grp + 2 * stride, aA);
        __builtin_amdgcn_sched_barrier(0);
        if (grp + stride < ngroups) {
            if (act) prologue(aB);
            compute(grp + stride, aB);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (g.stats) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float s = csum[j] + __shfl_xor(csum[j], 32), q = csq[j] + __shfl_xor(csq[j], 32);
            if (lh == 0) { red[wave][j * 32 + lr][0] = s; red[wave][j * 32 + lr][1] = q; }
        }
        __syncthreads();
        if (t < 32 * NT && t < g.N) {
            float a0 = (red[0][t][0] + red[1][t][0]) + (red[2][t][0] + red[3][t][0]);
            float a1 = (red[0][t][1] + red[1][t][1]) + (red[2][t][1] + red[3][t][1]);
            const int n = t;
            gfloat *sp = n >= g.N1 ? yh_global(g.stats2) : yh_global(g.stats);
            const int C = n >= g.N1 ? g.N - g.N1 : g.N1, nl = n >= g.N1 ? n - g.N1 : n;
            sp[((size_t)blockIdx.x * 2 + 0) * C + nl] = a0;
            sp[((size_t)blockIdx.x * 2 + 1) * C + nl] = a1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same fp32 GEMM on the bf16 matrix pipe ("x6" form, round 4).  MI355X multiplies fp32 at 157 TFLOP/s and bf16 at 2.5 PFLOP/s: an
// fp32 operand split into three bf16 terms (x = h + m + l exactly up to 2^-24 |x|: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m))
// makes a product a b = hh + (hm + mh) + (hl + lh + mm) + O(2^-24): SIX exact bf16 products accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16 -- 6 x 32 cycles per 16 channels of a 32 x 32 tile against 8 x 64 for v_mfma_f32_32x32x2_f32, 0.375 of the
// matrix time, and measured MORE accurate than the fp32 instruction (each product is exact, only the fp32 accumulation rounds: relative
// error 1.2e-7 against 3.0e-7 for K = 64 ... 2304 against an fp64 reference).  The weights' three planes are built once per workgroup in
// LDS in the B-operand layout ([plane][16-channel chunk][k half][column] x 8 bf16: conflict-free 16-byte reads); a wave owns 32 pixels x
// NT x 32 columns per trip, loads its pixels' K channels as fp32 (two float4 per chunk and lane: k = 16 c + 8 (lane >> 5) ...), applies
// the input prologue, splits in registers and runs 6 NT MFMAs per chunk; the next group's loads are in flight meanwhile.  Same epilogue,
// partial-sum rows and sibling-pair outputs as pw_stream_kernel.
typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
typedef float xf32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void x6_split(const xf32x8 v, xbf16x8 &h, xbf16x8 &m, xbf16x8 &l) {
    h = __builtin_convertvector(v, xbf16x8);
    const xf32x8 r1 = v - __builtin_convertvector(h, xf32x8);
    m = __builtin_convertvector(r1, xbf16x8);
    const xf32x8 r2 = r1 - __builtin_convertvector(m, xf32x8);
    l = __builtin_convertvector(r2, xbf16x8);
}

template <int NT, int KC, bool ACT>         // NT x 32 columns, K = 16 KC
__global__ __launch_bounds__(256) void pw_x6_kernel(const PwG g) {
    constexpr int NP = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    xbf16x8 *const bp = (xbf16x8 *)xsm;                                  // [3][KC][2][NP]
    float *const red = (float *)(xsm + (size_t)3 * KC * 2 * NP * 16);   // [4][NP][2]
    float *const ctab = red + 4 * NP * 2;                                // [3][16 KC]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void *)g.in, 0, g.in_bytes, 0x00020000);
    for (int i = t; i < KC * 2 * NP; i += 256) {
        const int n = i % NP, cg = i / NP;                               // cg = 2 c + k half: channels 8 cg ... 8 cg + 7
        xf32x8 w;
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = 0.f;
        if (n < g.ldw) {
            const f32x4 w0 = *(const f32x4 *)(g.Wq + ((size_t)(2 * cg) * g.ldw + n) * 4);
            const f32x4 w1 = *(const f32x4 *)(g.Wq + ((size_t)(2 * cg + 1) * g.ldw + n) * 4);
            w = xf32x8{w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
        }
        xbf16x8 h, m, l;
        x6_split(w, h, m, l);
        bp[(0 * KC * 2 + cg) * NP + n] = h;
        bp[(1 * KC * 2 + cg) * NP + n] = m;
        bp[(2 * KC * 2 + cg) * NP + n] = l;
    }
    if (ACT)
        for (int i = t; i < 3 * 16 * KC; i += 256) ctab[i] = g.icoef[(i / (16 * KC)) * g.icoef_ld + i % (16 * KC)];
    __syncthreads();

    bool nok[NT];
    float bias[NT], csum[NT], csq[NT];
    gfloat *ob[NT];
    int ldo[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + lr;
        nok[j] = n < g.N;
        const bool second = n >= g.N1;
        const int nl = second ? n - g.N1 : n;
        const gfloat *bpt = second ? yh_global(g.bias2) : yh_global(g.bias);
        bias[j] = (bpt && nok[j]) ? bpt[nl] : 0.f;
        ob[j] = (second ? yh_global(g.out2) : yh_global(g.out)) + nl;
        ldo[j] = second ? g.ldo2 : g.ldo;
        csum[j] = csq[j] = 0.f;
    }
    const bool plain = NP <= g.N;                          // wave-uniform: every lane's column exists
    const int ngroups = (g.M + 31) >> 5, stride = gridDim.x * 4;
    auto load = [&](int grp, f32x4 (&a)[KC][2]) {
        const int p = grp * 32 + lr;
        const unsigned off = (((unsigned)p * (unsigned)g.ldi + 8u * lh) * 4u) | ((unsigned)(p >= g.M || grp >= ngroups) << 31);
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            a[c][0] = buf_load<4>(r1, off + (unsigned)c * 64u);
            a[c][1] = buf_load<4>(r1, off + (unsigned)c * 64u + 16u);
        }
    };
    auto compute = [&](int grp, const f32x4 (&a)[KC][2]) {
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            xf32x8 v = {a[c][0][0], a[c][0][1], a[c][0][2], a[c][0][3], a[c][1][0], a[c][1][1], a[c][1][2], a[c][1][3]};
            if (ACT) {
                const int k0 = 16 * c + 8 * lh;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = yh_prologue(v[e], ctab[k0 + e], ctab[16 * KC + k0 + e], ctab[32 * KC + k0 + e]);
            }
            xbf16x8 ah, am, al;
            x6_split(v, ah, am, al);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const xbf16x8 bh = bp[((0 * KC + c) * 2 + lh) * NP + j * 32 + lr];
                const xbf16x8 bm = bp[((1 * KC + c) * 2 + lh) * NP + j * 32 + lr];
                const xbf16x8 bl = bp[((2 * KC + c) * 2 + lh) * NP + j * 32 + lr];
                // smallest terms first
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
            }
        }
        const int p0 = grp * 32;
        if (plain && p0 + 32 <= g.M) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                gfloat *o = ob[j] + (size_t)(p0 + 4 * lh) * ldo[j];
                const size_t step = (size_t)ldo[j];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[j][r] + bias[j];
                    *o = v;
                    o += (r & 3) == 3 ? 5 * step : step;
                    csum[j] += v;
                    csq[j] += v * v;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (nok[j] && p < g.M) {
                    const float v = acc[j][r] + bias[j];
                    ob[j][(size_t)p * ldo[j]] = v;
                    csum[j] += v;
                    csq[j] += v * v;
                }
            }
    };
    f32x4 aA[KC][2], aB[KC][2];
    int grp = blockIdx.x * 4 + wave;
    load(grp, aA);
    for (; grp < ngroups; grp += 2 * stride) {
        load(grp + stride, aB);
        __builtin_amdgcn_sched_barrier(0);
        compute(grp, aA);
        __builtin_amdgcn_sched_barrier(0);
        load(grp + 2 * stride, aA);
        __builtin_amdgcn_sched_barrier(0);
        if (grp + stride < ngroups) compute(grp + stride, aB);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (g.stats) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float sm = csum[j] + __shfl_xor(csum[j], 32), q = csq[j] + __shfl_xor(csq[j], 32);
            if (lh == 0) { red[(wave * NP + j * 32 + lr) * 2] = sm; red[(wave * NP + j * 32 + lr) * 2 + 1] = q; }
        }
        __syncthreads();
        if (t < NP && t < g.N) {
            const float a0 = (red[t * 2] + red[(NP + t) * 2]) + (red[(2 * NP + t) * 2] + red[(3 * NP + t) * 2]);
            const float a1 = (red[t * 2 + 1] + red[(NP + t) * 2 + 1]) + (red[(2 * NP + t) * 2 + 1] + red[(3 * NP + t) * 2 + 1]);
            const int n = t;
            gfloat *sp = n >= g.N1 ? yh_global(g.stats2) : yh_global(g.stats);
            const int C = n >= g.N1 ? g.N - g.N1 : g.N1, nl = n >= g.N1 ? n - g.N1 : n;
            sp[((size_t)blockIdx.x * 2 + 0) * C + nl] = a0;
            sp[((size_t)blockIdx.x * 2 + 1) * C + nl] = a1;
        }
    }
}

// Status: EXPERIMENTAL, off unless YH_PW_X6=1 (read once per process).  Alone the kernel is faster (1x1 64->64 sibling pairs at 80^2:
// 61 -> 44 us per launch, 32->32 at 160^2 91 -> 81, 128->64 at 80^2 97 -> 78 us), but inside the training step the launches AROUND it
// slow down by more than it gains (step 16.61 -> 16.70 ms with 1024 workgroups per launch, 16.9-17.05 with 512, 16.63 with 256; the
// serial per-launch sum shows the other kernels +0.16 ms): a fixed fp32 Winograd layer runs 9 % slower behind a burst of these
// launches than behind the fp32 ones (tools/x6_aftermath.py) -- the chip's clock after dense bf16-MFMA work.  Kept as the tested starting point for moving the fp32 GEMMs to the bf16 pipe: yh_conv_pw_fwd_x6 forces it.
inline bool pw_x6_shape_ok(int64_t M, int K, int N) {
    return M >= 4096 && K % 16 == 0 && K >= 16 && K <= 128 && N <= 128 && (int64_t)K * (N > 64 ? 128 : (N > 32 ? 64 : 32)) <= 64 * 128;
}
inline bool pw_x6_can(int64_t M, int K, int N) {         // ... and an instantiation exists (K = 16, 32, 64, 128)
    return pw_x6_shape_ok(M, K, N) && (K == 16 || K == 32 || K == 64 || K == 128);
}
inline bool pw_x6_takes(int64_t M, int K, int N) {       // the automatic route (planner and launcher agree: both call this)
    return yh_env_pw_x6() && pw_x6_can(M, K, N);
}
inline int pw_x6_blocks(int64_t M) {                     // one workgroup per CU (measured best inside the step: see above)
    int64_t b = ((M + 31) / 32 + 3) / 4;
    return (int)(b > 256 ? 256 : b);
}
inline size_t pw_x6_smem(int K, int NT) {
    const int KC = K / 16, NP = 32 * NT;
    return (size_t)3 * KC * 2 * NP * 16 + (size_t)4 * NP * 2 * 4 + (size_t)3 * 16 * KC * 4;
}

// streaming form: K in {16, 32, 64}, N <= 64 and enough pixels to stream (measured: 160^2 32->32 0.104 -> 0.085 ms = 4.9 TB/s;
// N = 128 with K = 32 was slower than the tiled kernel and is left to it)
inline bool pw_use_stream(int64_t M, int K, int N) {
    return M >= 32768 && (K == 16 || K == 32 || K == 64) && N <= 64;
}
inline int pw_stream_blocks(int64_t M) {
    int64_t b = ((M + 31) / 32 + 3) / 4;
    return (int)(b > 1024 ? 1024 : b);
}

// LDS-staged form: every tiled (non-streaming) training-path layer with K >= 64
// pixels per workgroup of the LDS-staged form: 64 (more, smaller workgroups: 3-4 per CU instead of 2) unless N <= 32
inline int pw_tile_bm(int N) {
    if (N <= 32) return 128;
    return N <= 64 ? 64 : 128;                 // measured: 64 wins for N = 64 (33 vs 40 us at 128 -> 64, 40x40), 128 for N >= 128
}
inline int pw_tile_slots(int BM) {
    (void)BM;
    return 512;                                // 2 workgroups per CU (67 KB of LDS at 128 rows; ~200 VGPRs at 64)
}
// grid.x of the LDS-staged persistent kernel = rows of BatchNorm partial sums it writes
inline int pw_tile_gx(int64_t M, int N) {
    const int BM = pw_tile_bm(N), WN = N > 64 ? 4 : (N > 32 ? 2 : 1);
    const int ntile = cdiv((int)M, BM), ncol = cdiv(N, 32 * WN);
    int gx = cdiv(pw_tile_slots(BM), ncol);
    if (gx > ntile) gx = ntile;
    return cdiv(ntile, cdiv(ntile, gx));                                   // same number of rounds, evenly filled
}
inline bool pw_use_tile(int K, int N) {
    (void)N;
    return K >= 64 && K % 32 == 0;
}

struct PwPackDesc {
    const float *w;             // [Cout][Cin] (OIHW, 1x1)
    float *wf, *wb;             // forward: Wq[Cin/4][ldwf][4]; backward-data: Wq[(koff + Cout)/4 rows...][ldwb][4]
    int Cout, Cin, ldwf, ldwb, koff, noff;   // noff: first column of this conv inside a stacked forward matrix
};

// forward  Wq_f[ci >> 2][co][ci & 3] = w[co][ci];   backward  Wq_b[(koff + co) >> 2][ci][(koff + co) & 3] = w[co][ci]
__global__ void pw_pack_multi_kernel(const PwPackDesc *__restrict__ tab) {
    const PwPackDesc d = tab[blockIdx.y];
    // forward: only this conv's own columns are written (padding columns keep the zeros of the allocation)
    const int nf = d.wf ? d.Cin * d.Cout : 0, nb = d.wb ? d.Cout * d.ldwb : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf + nb; i += gridDim.x * blockDim.x) {
        if (i < nf) {
            int n = i % d.Cout, k = i / d.Cout;
            d.wf[((size_t)(k >> 2) * d.ldwf + d.noff + n) * 4 + (k & 3)] = d.w[(size_t)n * d.Cin + k];
        } else {
            int j = i - nf;
            int n = j % d.ldwb, k = j / d.ldwb, kk = d.koff + k;
            d.wb[((size_t)(kk >> 2) * d.ldwb + n) * 4 + (kk & 3)] = n < d.Cin ? d.w[(size_t)k * d.Cin + n] : 0.f;
        }
    }
}

int launch_pw_gemm(PwG &g, hipStream_t st, bool force_x6 = false) {
    YH_REQUIRE(g.K % 8 == 0 && g.K1 % 8 == 0 && g.K1 > 0 && g.K1 <= g.K && g.ldi % 4 == 0 && (((uintptr_t)g.in | (uintptr_t)g.Wq) & 15) == 0 &&
                   (!g.in2 || (((uintptr_t)g.in2) & 15) == 0) && g.ldw >= g.N && g.M > 0,
               "conv_pw: channels must be multiples of 8, operands 16-byte addressable");
    YH_REQUIRE(((int64_t)(g.M - 1) * g.ldi + g.K) * 4 < (1ll << 31), "conv_pw: input view must span less than 2 GiB");
    if (!g.out2) g.N1 = g.N;
    g.in_bytes = (unsigned)(((int64_t)(g.M - 1) * g.ldi + g.K1) * 4);
    g.in2_bytes = (unsigned)(((int64_t)(g.M - 1) * g.ldi + (g.K - g.K1)) * 4);
    const int NT = g.N > 64 ? 4 : (g.N > 32 ? 2 : 1);
    const bool fused = g.act || g.up2 || g.res;          // the streaming form has no inference epilogue
    YH_REQUIRE(!g.icoef || (!fused && !g.in2 && (((uintptr_t)g.icoef) & 15) == 0 && g.icoef_ld % 4 == 0 && g.icoef_ld >= g.K &&
                            (force_x6 || pw_x6_takes(g.M, g.K, g.N) || pw_use_stream(g.M, g.K, g.N) || (g.K % 32 == 0 && pw_use_tile(g.K, g.N)))),
               "conv_pw: the input prologue runs on the streaming and the tiled forward kernels only (yh_conv_pw_prologue_ok)");
    if (!fused && !g.in2 && !g.accumulate && (force_x6 ? pw_x6_can(g.M, g.K, g.N) : pw_x6_takes(g.M, g.K, g.N))) {
        dim3 sg(pw_x6_blocks(g.M));
        const int KC = g.K / 16;
        const size_t smem = pw_x6_smem(g.K, NT);
        int rc = 0;
        bool ok = true;
#define YH_X6(nt, kc)                                                                                         \
    do {                                                                                                      \
        if (g.icoef) {                                                                                        \
            rc = yh_ensure_dyn_smem((const void *)pw_x6_kernel<nt, kc, true>, smem);                          \
            if (!rc) hipLaunchKernelGGL((pw_x6_kernel<nt, kc, true>), sg, dim3(256), smem, st, g);           \
        } else {                                                                                              \
            rc = yh_ensure_dyn_smem((const void *)pw_x6_kernel<nt, kc, false>, smem);                         \
            if (!rc) hipLaunchKernelGGL((pw_x6_kernel<nt, kc, false>), sg, dim3(256), smem, st, g);          \
        }                                                                                                     \
    } while (0)
#define YH_X6K(nt)                                                                                            \
    do {                                                                                                      \
        if (KC == 1) YH_X6(nt, 1); else if (KC == 2) YH_X6(nt, 2); else if (KC == 4) YH_X6(nt, 4);            \
        else if (KC == 8) YH_X6(nt, 8); else ok = false;                                                      \
    } while (0)
        if (NT == 1) YH_X6K(1); else if (NT == 2) YH_X6K(2); else YH_X6K(4);
#undef YH_X6K
#undef YH_X6
        if (ok) {
            if (rc) return rc;
            YH_CHECK_LAUNCH("pw_x6");
            return 0;
        }
    }
    if (!fused && pw_use_stream(g.M, g.K, g.N)) {
        dim3 sg(pw_stream_blocks(g.M));
        const int KC = g.K / 8;
#define YH_PWS(nt, kc) hipLaunchKernelGGL((pw_stream_kernel<nt, kc>), sg, dim3(256), 0, st, g)
        bool ok = true;
        if (NT == 1) { if (KC == 2) YH_PWS(1, 2); else if (KC == 4) YH_PWS(1, 4); else if (KC == 8) YH_PWS(1, 8); else ok = false; }
        else if (NT == 2) { if (KC == 2) YH_PWS(2, 2); else if (KC == 4) YH_PWS(2, 4); else if (KC == 8) YH_PWS(2, 8); else ok = false; }
        else ok = false;
#undef YH_PWS
        YH_REQUIRE(ok, "conv_pw: streaming form needs K in {16, 32, 64}");
        YH_CHECK_LAUNCH("pw_stream");
        return 0;
    }
    if (!fused && g.K1 % 32 == 0 && (g.K - g.K1) % 32 == 0 && pw_use_tile(g.K, g.N)) {
        const int WN = g.N > 64 ? 4 : (g.N > 32 ? 2 : 1);
        const int BM = pw_tile_bm(g.N);
        const int ntile = cdiv(g.M, BM), ncol = cdiv(g.N, 32 * WN);
        const int gx = pw_tile_gx(g.M, g.N);
        dim3 tg(gx, ncol);
#ifdef YH_PW_STAMPS
        static unsigned long long *dbgbuf = nullptr;
        if (!dbgbuf) (void)hipMalloc((void **)&dbgbuf, (size_t)1 << 24);
        g.dbg = getenv("YH_PW_DBG") && (size_t)ntile * 48 <= ((size_t)1 << 24) ? dbgbuf : nullptr;
#endif
        const size_t smem = (size_t)BM * 132 * sizeof(float);
        int rc = 0;
#define YH_PWT(tmw, wn)                                                                                  \
    do {                                                                                                 \
        if (g.icoef) {                                                                                   \
            rc = yh_ensure_dyn_smem((const void *)pw_tile_kernel<tmw, wn, true>, smem);                  \
            if (!rc) hipLaunchKernelGGL((pw_tile_kernel<tmw, wn, true>), tg, dim3(256), smem, st, g);    \
        } else {                                                                                         \
            rc = yh_ensure_dyn_smem((const void *)pw_tile_kernel<tmw, wn>, smem);                        \
            if (!rc) hipLaunchKernelGGL((pw_tile_kernel<tmw, wn>), tg, dim3(256), smem, st, g);          \
        }                                                                                                \
    } while (0)
        if (WN == 4) { if (BM == 64) YH_PWT(2, 4); else YH_PWT(4, 4); }
        else if (WN == 2) { if (BM == 64) YH_PWT(1, 2); else YH_PWT(2, 2); }
        else YH_PWT(1, 1);
#undef YH_PWT
        if (rc) return rc;
        YH_CHECK_LAUNCH("pw_tile");
#ifdef YH_PW_STAMPS
        if (getenv("YH_PW_DBG")) {
            (void)hipStreamSynchronize(st);
            std::vector<unsigned long long> h((size_t)ntile * 6);
            (void)hipMemcpy(h.data(), g.dbg, h.size() * 8, hipMemcpyDeviceToHost);
            double a = 0, b = 0, c = 0, rt = 0; unsigned long long lo = ~0ull, hi = 0; int cnt = 0;
            for (int i = gx; i < ntile; ++i) {       // tiles after each workgroup's first one (steady state)
                a += (double)(h[6 * i + 1] - h[6 * i]); b += (double)(h[6 * i + 2] - h[6 * i + 1]); c += (double)(h[6 * i + 3] - h[6 * i + 2]);
                rt += (double)(h[6 * i + 5] - h[6 * i + 4]); ++cnt;
            }
            for (int i = 0; i < ntile; ++i) { if (h[6 * i + 4] < lo) lo = h[6 * i + 4]; if (h[6 * i + 5] > hi) hi = h[6 * i + 5]; }
            if (cnt) fprintf(stderr, "[pw stamps] tiles %d grid %d: park+barriers %.0f, multiply %.0f, epilogue %.0f ticks per tile = %.2f us (clock %.2f GHz); kernel span %.1f us\n",
                             ntile, gx, a / cnt, b / cnt, c / cnt, rt / cnt / 100.0, (a + b + c) / rt * 0.1, (double)(hi - lo) / 100.0);
        }
#endif
        return 0;
    }
    // (the tile shape is a function of the problem alone, shared with yh_conv_pw_blocks / yh_conv_pw_bwd_data_bn_rows, which size
    // the BatchNorm partial-sum rows: no run-time override may change grid.x behind the planner's back)
    const int TM = (NT <= 2 && cdiv(g.M, 256) * cdiv(g.N, 32 * NT) >= 1024) ? 2 : 1;   // <2,4> would run one wave per SIMD
    dim3 grid(cdiv(g.M, 128 * TM), cdiv(g.N, 32 * NT));
#define YH_PWG(tm, nt) hipLaunchKernelGGL((pw_gemm_kernel<tm, nt>), grid, dim3(256), 0, st, g)
    if (TM == 2) { if (NT == 2) YH_PWG(2, 2); else YH_PWG(2, 1); }
    else { if (NT == 4) YH_PWG(1, 4); else if (NT == 2) YH_PWG(1, 2); else YH_PWG(1, 1); }
#undef YH_PWG
    YH_CHECK_LAUNCH("pw_gemm");
    return 0;
}

}  // namespace

extern "C" int yh_conv_pw_blocks(int64_t M, int K, int Cout) {
    if (pw_x6_takes(M, K, Cout)) return pw_x6_blocks(M);
    if (pw_use_stream(M, K, Cout)) return pw_stream_blocks(M);
    if (pw_use_tile(K, Cout)) return pw_tile_gx(M, Cout);
    const int NT = Cout > 64 ? 4 : (Cout > 32 ? 2 : 1);
    const int TM = (NT <= 2 && cdiv((int)M, 256) * cdiv(Cout, 32 * NT) >= 1024) ? 2 : 1;
    return cdiv((int)M, 128 * TM);
}

extern "C" int yh_pw_pack_multi(const void *table, int n, void *stream) {
    YH_REQUIRE(table && n > 0, "pw_pack_multi: bad argument");
    static_assert(sizeof(PwPackDesc) == 48, "descriptor layout is part of the ABI");
    hipLaunchKernelGGL(pw_pack_multi_kernel, dim3(128, n), dim3(256), 0, (hipStream_t)stream, (const PwPackDesc *)table);
    YH_CHECK_LAUNCH("pw_pack_multi");
    return 0;
}

extern "C" int yh_conv_pw_prologue_ok(int64_t M, int Cin, int Cout) {
    return (pw_x6_takes(M, Cin, Cout) || pw_use_stream(M, Cin, Cout) || (Cin % 32 == 0 && pw_use_tile(Cin, Cout))) ? 1 : 0;
}
extern "C" int yh_conv_pw_fwd_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *wq, int ldw, const float *bias,
                                  float *y, int ldy, float *bn_partials, int64_t M, int Cin, int Cout, void *stream);
extern "C" int yh_conv_pw_fwd(const float *x, int ldx, const float *wq, int ldw, const float *bias, float *y, int ldy,
                              float *bn_partials, int64_t M, int Cin, int Cout, void *stream) {
    return yh_conv_pw_fwd_act(x, ldx, nullptr, 0, wq, ldw, bias, y, ldy, bn_partials, M, Cin, Cout, stream);
}
extern "C" int yh_conv_pw_fwd_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *wq, int ldw, const float *bias,
                                  float *y, int ldy, float *bn_partials, int64_t M, int Cin, int Cout, void *stream) {
    YH_REQUIRE(x && wq && y && M > 0 && M < (1ll << 30) && ldx >= Cin && ldy >= Cout, "conv_pw_fwd: bad argument");
    PwG g{};
    g.icoef = icoef; g.icoef_ld = icoef_ld;
    g.in = x; g.Wq = wq; g.bias = bias; g.out = y; g.stats = bn_partials;
    g.ldi = ldx; g.ldw = ldw; g.ldo = ldy; g.M = (int)M; g.K = Cin; g.K1 = Cin; g.N = Cout;
    return launch_pw_gemm(g, (hipStream_t)stream);
}

extern "C" int yh_conv_pw_x6_blocks(int64_t M, int Cin, int Cout) { return pw_x6_can(M, Cin, Cout) ? pw_x6_blocks(M) : 0; }
extern "C" int yh_conv_pw_fwd_x6(const float *x, int ldx, const float *icoef, int icoef_ld, const float *wq, int ldw, const float *bias,
                                 float *y, int ldy, float *bn_partials, int64_t M, int Cin, int Cout, void *stream) {
    YH_REQUIRE(x && wq && y && M > 0 && M < (1ll << 30) && ldx >= Cin && ldy >= Cout, "conv_pw_fwd_x6: bad argument");
    YH_REQUIRE(pw_x6_can(M, Cin, Cout), "conv_pw_fwd_x6: unsupported problem (yh_conv_pw_x6_blocks == 0)");
    PwG g{};
    g.icoef = icoef; g.icoef_ld = icoef_ld;
    g.in = x; g.Wq = wq; g.bias = bias; g.out = y; g.stats = bn_partials;
    g.ldi = ldx; g.ldw = ldw; g.ldo = ldy; g.M = (int)M; g.K = Cin; g.K1 = Cin; g.N = Cout;
    return launch_pw_gemm(g, (hipStream_t)stream, true);
}

extern "C" int yh_conv_pw_fwd_fused(const float *x, int ldx, const float *wq, int ldw, const float *bias, const float *res, int ldr,
                                    float *y, int ldy, int B, int H, int W, int Cin, int Cout, int act_silu, int upsample, void *stream) {
    const int64_t M = (int64_t)B * H * W;
    YH_REQUIRE(x && wq && y && M > 0 && M < (1ll << 30) && ldx >= Cin && ldy >= Cout && (!res || ldr >= Cout), "conv_pw_fwd_fused: bad argument");
    PwG g{};
    g.in = x; g.Wq = wq; g.bias = bias; g.out = y; g.stats = nullptr;
    g.ldi = ldx; g.ldw = ldw; g.ldo = ldy; g.M = (int)M; g.K = Cin; g.K1 = Cin; g.N = Cout;
    g.res = res; g.ldr = ldr; g.act = act_silu ? 1 : 0; g.up2 = upsample ? 1 : 0; g.H = H; g.W = W;
    return launch_pw_gemm(g, (hipStream_t)stream);
}

extern "C" int yh_conv_pw_fwd2_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *wq, int ldw, const float *bias1,
                                   float *y1, int ldy1, float *bn_partials1, int cout1, const float *bias2, float *y2, int ldy2,
                                   float *bn_partials2, int cout2, int64_t M, int Cin, void *stream);
extern "C" int yh_conv_pw_fwd2(const float *x, int ldx, const float *wq, int ldw, const float *bias1, float *y1, int ldy1,
                               float *bn_partials1, int cout1, const float *bias2, float *y2, int ldy2, float *bn_partials2,
                               int cout2, int64_t M, int Cin, void *stream) {
    return yh_conv_pw_fwd2_act(x, ldx, nullptr, 0, wq, ldw, bias1, y1, ldy1, bn_partials1, cout1, bias2, y2, ldy2, bn_partials2, cout2, M, Cin,
                               stream);
}
extern "C" int yh_conv_pw_fwd2_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *wq, int ldw, const float *bias1,
                                   float *y1, int ldy1, float *bn_partials1, int cout1, const float *bias2, float *y2, int ldy2,
                                   float *bn_partials2, int cout2, int64_t M, int Cin, void *stream) {
    YH_REQUIRE(x && wq && y1 && y2 && M > 0 && M < (1ll << 30) && ldx >= Cin && ldy1 >= cout1 && ldy2 >= cout2 && cout1 > 0 && cout2 > 0 &&
                   (!bn_partials1) == (!bn_partials2), "conv_pw_fwd2: bad argument");
    PwG g{};
    g.icoef = icoef; g.icoef_ld = icoef_ld;
    g.in = x; g.Wq = wq; g.bias = bias1; g.bias2 = bias2; g.out = y1; g.out2 = y2; g.stats = bn_partials1; g.stats2 = bn_partials2;
    g.ldi = ldx; g.ldw = ldw; g.ldo = ldy1; g.ldo2 = ldy2; g.M = (int)M; g.K = Cin; g.K1 = Cin; g.N = cout1 + cout2; g.N1 = cout1;
    return launch_pw_gemm(g, (hipStream_t)stream);
}

extern "C" int yh_conv_pw_bwd_data(const float *dy1, int cout1, const float *dy2, int cout2, int lddy, const float *wq, int ldw,
                                   float *dx, int lddx, int64_t M, int Cin, int accumulate, void *stream) {
    YH_REQUIRE(dy1 && wq && dx && M > 0 && M < (1ll << 30) && cout1 > 0 && (dy2 ? cout2 > 0 : cout2 == 0) && lddx >= Cin,
               "conv_pw_bwd_data: bad argument");
    PwG g{};
    g.in = dy1; g.in2 = dy2; g.Wq = wq; g.out = dx;
    g.ldi = lddy; g.ldw = ldw; g.ldo = lddx; g.M = (int)M; g.K = cout1 + cout2; g.K1 = cout1; g.N = Cin; g.accumulate = accumulate;
    return launch_pw_gemm(g, (hipStream_t)stream);
}
