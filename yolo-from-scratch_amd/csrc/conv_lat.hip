// Latency-oriented fused convolution for SMALL-M inference (batch 1: BASELINE config 5), gfx950, NHWC fp32.
//
// At batch 1 a layer of this network is 0.1 - 0.5 GFLOP on at most a few thousand pixels: its time is not arithmetic but the
// CHAIN OF DEPENDENT MEMORY ROUND TRIPS inside one launch.  The gather-GEMM with in-launch split-K (conv_gemm.hip) spends
// 15 - 22 us per layer: a K loop of 32-channel chunks through LDS with a barrier each (one round trip per chunk), then fp32
// slabs, an agent-scope release, a ticket, an acquire and a slab reduction (three more round trips) -- 57 such launches are
// 1.08 of the 1.24 ms the device spends on an image.
//
// Here K is split over the WAVES of one workgroup instead of over workgroups:
//   * a workgroup owns 32 output pixels x 32 output channels; wave w of NW (4 or 8) owns a contiguous range of 8-channel
//     K chunks (k = tap * Cin + ci);
//   * operands go straight into the MFMA operand registers, no LDS and no barrier in the loop (the scheme of the pointwise and
//     Winograd kernels): lane (r = lane & 31, h = lane >> 5) loads the float4 of channels ci0 + 4h .. + 3 of ITS pixel at the
//     chunk's tap (A operand of four v_mfma_f32_32x32x2_f32) and the float4 Wq[(k >> 2)][n][0..3] of its column from the
//     k-quad interleaved weight pack (B operand); padding pixels load a valid address and are zeroed by a select;
//   * every wave keeps two batches of LB chunks in flight behind its MFMAs;
//   * the NW partial 32 x 32 tiles are summed through LDS in wave order (fixed order: bitwise reproducible), then bias, SiLU,
//     residual, x2 upsample and 128-byte row stores -- no slabs, no fences, no tickets, nothing another workgroup waits for.
// Same results as yh_conv_fwd_fused to fp32 summation order.  replaces: the eval-mode ConvBlock / Conv2d forward of predict()
// (train.py:253-265, 1140-1141) for layers with few pixels.
#include "common.h"

namespace {

// K chunks (8 channels each) per batch and wave.  Measured per layer at batch 1 (tools/lat_bench.py, us; waves x LB): 80x80 64->64 3x3
// 18.5 (8 x 6) / 15.9 (8 x 2) / 14.7 (4 x 3) / 18.8 (16 x 2); 20x20 256->256 3x3 27.8 (16 x 6) / 21.8 (8 x 2); 40x40 256->64 1x1
// 7.4 (4 x 6) / 5.2 (8 x 2); 80x80 128->32 1x1 6.8 / 4.8 -- short batches keep the register count low (more waves resident) and
// waste fewer MFMAs on the dead chunks of a ragged last batch; eight waves beat four and sixteen.
constexpr int LB = 2;

struct LatConv {
    const float *in, *Wq, *bias, *res;
    float *out;
    int ldi, ldw, ldo, ldr;
    int B, Hi, Wi, Ho, Wo, Cin, N, k, s, M;
    int act, up2;
    int nchunks, cpw;           // K / 8; chunks per wave
    unsigned wo_magic, ho_magic;
    int wo_shift, ho_shift;
};

__device__ __forceinline__ int lat_div(int n, unsigned magic, int shift) {
    return shift < 0 ? n : (int)(__umulhi((unsigned)n, magic) >> shift);
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void lat_conv_kernel(const LatConv g) {
    __shared__ __attribute__((aligned(16))) float red[NW][16][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int pad = g.k >> 1;

    // this lane's output pixel and the input pixel under tap (0, 0)
    const int m = m0 + lr;
    const bool mv = m < g.M;
    const int mc = mv ? m : 0;
    const int q = lat_div(mc, g.wo_magic, g.wo_shift), ox = mc - q * g.Wo;
    const int b = lat_div(q, g.ho_magic, g.ho_shift), oy = q - b * g.Ho;
    const int iy0 = oy * g.s - pad, ix0 = ox * g.s - pad;
    const gfloat *const pin = yh_global(g.in) + ((ptrdiff_t)(b * g.Hi + iy0) * g.Wi + ix0) * g.ldi + 4 * lh;
    unsigned okm = 0;           // bit tap: the tap's input pixel exists
    for (int tap = 0; tap < g.k * g.k; ++tap) {
        const int dy = tap / g.k, dx = tap - dy * g.k;
        okm |= (unsigned)(mv && (unsigned)(iy0 + dy) < (unsigned)g.Hi && (unsigned)(ix0 + dx) < (unsigned)g.Wi) << tap;
    }
    const int n = n0 + lr;
    const gfloat *const pw = yh_global(g.Wq) + ((size_t)lh * g.ldw + (n < g.ldw ? n : 0)) * 4;
    const size_t wstep = (size_t)2 * g.ldw * 4;                    // one 8-channel chunk of weight rows

    const int c_begin = wave * g.cpw;
    int c_end = c_begin + g.cpw;
    if (c_end > g.nchunks) c_end = g.nchunks;
    const int cpt = g.Cin >> 3;                                    // chunks per tap

    f32x4 aA[LB], bA[LB], aB[LB], bB[LB];
    // UNCONDITIONAL loads (a padding pixel / a chunk past the range re-reads a valid address and is zeroed by a select): nothing
    // under a branch, so the compiler's counted waits let the next batch stay in flight behind the MFMAs of the current one
    auto load_batch = [&](int c0, f32x4 (&a)[LB], f32x4 (&bq)[LB]) {
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int c = c0 + i;
            const bool live = c < c_end;
            const int cc = live ? c : 0;                           // (a wave past the end of K has c_begin >= nchunks: chunk 0 always exists)
            const int tap = cc / cpt, ci = (cc - tap * cpt) << 3;  // wave-uniform
            const int dy = tap / g.k, dx = tap - dy * g.k;
            const bool ok = live && (okm >> tap & 1);
            const f32x4 v = *(const YH_GLOBAL f32x4 *)(pin + (ok ? (ptrdiff_t)((dy * g.Wi + dx) * g.ldi + ci) : (ptrdiff_t)((pad * g.Wi + pad) * g.ldi)));
            a[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            bq[i] = *(const YH_GLOBAL f32x4 *)(pw + (size_t)cc * wstep);
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto compute = [&](const f32x4 (&a)[LB], const f32x4 (&bq)[LB]) {
#pragma unroll
        for (int i = 0; i < LB; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], bq[i][e], acc, 0, 0, 0);
    };
    // (the centre tap of pixel `m` always exists for mv lanes; lanes past M read pixel 0's centre: in bounds)
    load_batch(c_begin, aA, bA);
    for (int c0 = c_begin; c0 < c_end; c0 += 2 * LB) {
        load_batch(c0 + LB, aB, bB);
        __builtin_amdgcn_sched_barrier(0);
        compute(aA, bA);
        __builtin_amdgcn_sched_barrier(0);
        load_batch(c0 + 2 * LB, aA, bA);
        __builtin_amdgcn_sched_barrier(0);
        if (c0 + LB < c_end) compute(aB, bB);                      // wave-uniform (the loads above stay unconditional)
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- sum the NW partial tiles in wave order; thread (register rr, lane) of the result: row = MFMA row of (rr, lane >> 5),
    // column = lane & 31 -- 32 consecutive channels per 32 lanes, 128-byte row stores
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    for (int o = t; o < 16 * 64; o += 64 * NW) {
        const int rr = o >> 6, ln = o & 63;
        float v = red[0][rr][ln];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[w][rr][ln];
        const int row = (rr & 3) + 8 * (rr >> 2) + 4 * (ln >> 5), col = ln & 31;
        const int mo = m0 + row, no = n0 + col;
        if (mo >= g.M || no >= g.N) continue;
        v += g.bias ? g.bias[no] : 0.f;
        if (g.act) v = v * yh_sigmoid(v);
        if (g.res) v += g.res[(size_t)mo * g.ldr + no];
        if (!g.up2) {
            g.out[(size_t)mo * g.ldo + no] = v;
        } else {
            const int q2 = lat_div(mo, g.wo_magic, g.wo_shift), x2 = mo - q2 * g.Wo;
            const int b2 = lat_div(q2, g.ho_magic, g.ho_shift), y2 = q2 - b2 * g.Ho;
            float *dst = g.out + (((size_t)b2 * (2 * g.Ho) + 2 * y2) * (2 * g.Wo) + 2 * x2) * g.ldo + no;
            const size_t rs = (size_t)(2 * g.Wo) * g.ldo;
            dst[0] = v; dst[g.ldo] = v; dst[rs] = v; dst[rs + g.ldo] = v;
        }
    }
}

// Wq[(k >> 2)][n][k & 3] = w[n][ci][kh][kw] with k = (kh * kw_count + kw) * Cin + ci: the k-quad interleaved pack of the kernel above
struct LatPackDesc {
    const float *w;             // OIHW (BatchNorm already folded in)
    float *wq;
    int Cout, Cin, kk, ldw;
};
__global__ void lat_pack_multi_kernel(const LatPackDesc *__restrict__ tab) {
    const LatPackDesc d = tab[blockIdx.y];
    const int K = d.kk * d.Cin, total = K * d.ldw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int n = i % d.ldw, kidx = i / d.ldw;
        const int tap = kidx / d.Cin, ci = kidx - tap * d.Cin;
        d.wq[((size_t)(kidx >> 2) * d.ldw + n) * 4 + (kidx & 3)] = n < d.Cout ? d.w[((size_t)n * d.Cin + ci) * d.kk + tap] : 0.f;
    }
}

void lat_magic(unsigned d, unsigned &magic, int &shift) {
    int l = 0;
    while ((1u << l) < d) ++l;
    magic = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
    shift = l - 1;
}

// waves per workgroup: eight (measured above), four when K has fewer than eight chunks
int lat_waves(int nchunks) { return nchunks < 8 ? 4 : 8; }

}  // namespace

extern "C" int yh_conv_lat_ok(int B, int Hi, int Wi, int Cin, int Cout, int k, int s) {
    const int pad = k / 2, Ho = (Hi + 2 * pad - k) / s + 1, Wo = (Wi + 2 * pad - k) / s + 1;
    const int64_t M = (int64_t)B * Ho * Wo;
    return ((k == 1 || k == 3) && (s == 1 || s == 2) && !(k == 1 && s == 2) && Cin % 8 == 0 && Cout > 0 && M > 0 && M < (1 << 24) &&
            Hi >= k && Wi >= k) ? 1 : 0;
}

extern "C" int yh_lat_pack_multi(const void *table, int n, void *stream) {
    YH_REQUIRE(table && n > 0, "lat_pack_multi: bad argument");
    static_assert(sizeof(LatPackDesc) == 32, "descriptor layout is part of the ABI");
    hipLaunchKernelGGL(lat_pack_multi_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, (const LatPackDesc *)table);
    YH_CHECK_LAUNCH("lat_pack_multi");
    return 0;
}

extern "C" int yh_conv_lat_fwd_fused(const float *x, int ldx, const float *wq, int ldw, const float *bias, const float *res, int ldr,
                                     float *y, int ldy, int B, int Hi, int Wi, int Cin, int Cout, int k, int s, int act_silu,
                                     int upsample, void *stream) {
    YH_REQUIRE(x && wq && y && yh_conv_lat_ok(B, Hi, Wi, Cin, Cout, k, s), "conv_lat_fwd_fused: unsupported problem");
    YH_REQUIRE(ldx >= Cin && ldx % 4 == 0 && ldw % 4 == 0 && ldw >= Cout && ldy >= Cout && (!res || ldr >= Cout) &&
                   (((uintptr_t)x | (uintptr_t)wq) & 15) == 0,
               "conv_lat_fwd_fused: 16-byte addressable input rows and weight pack required");
    YH_REQUIRE((int64_t)B * Hi * Wi * ldx < (1ll << 31), "conv_lat_fwd_fused: input exceeds 32-bit element offsets");
    LatConv g{};
    const int pad = k / 2;
    g.in = x; g.Wq = wq; g.bias = bias; g.res = res; g.out = y;
    g.ldi = ldx; g.ldw = ldw; g.ldo = ldy; g.ldr = ldr;
    g.B = B; g.Hi = Hi; g.Wi = Wi; g.Ho = (Hi + 2 * pad - k) / s + 1; g.Wo = (Wi + 2 * pad - k) / s + 1;
    g.Cin = Cin; g.N = Cout; g.k = k; g.s = s; g.M = B * g.Ho * g.Wo;
    g.act = act_silu ? 1 : 0; g.up2 = upsample ? 1 : 0;
    g.nchunks = k * k * Cin / 8;
    const int NW = lat_waves(g.nchunks);
    g.cpw = cdiv(g.nchunks, NW);
    lat_magic((unsigned)g.Wo, g.wo_magic, g.wo_shift);
    lat_magic((unsigned)g.Ho, g.ho_magic, g.ho_shift);
    dim3 grid(cdiv(g.M, 32), cdiv(Cout, 32));
    hipStream_t st = (hipStream_t)stream;
    if (NW == 4) hipLaunchKernelGGL((lat_conv_kernel<4>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((lat_conv_kernel<8>), grid, dim3(512), 0, st, g);
    YH_CHECK_LAUNCH("conv_lat");
    return 0;
}
