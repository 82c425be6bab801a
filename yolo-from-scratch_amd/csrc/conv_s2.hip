// 3x3 stride-2 convolution (the five down-sampling layers with >= 32 input channels), forward, on v_mfma_f32_32x32x2_f32 with
// the INPUT PATCH STAGED THROUGH LDS (round 4), gfx950, NHWC fp32.
//
// These layers ran on the gather implicit GEMM at 66 - 77 TFLOP/s (conv_gemm.hip: a 32-channel K chunk through LDS per barrier,
// every output pixel's nine taps gathered separately).  Here a workgroup owns a 2-D block of R x C output pixels and, per chunk
// of 8 input channels, reads the block's (2R+1) x (2C+1) input patch ONCE into LDS (the staging code of wino_lds_kernel: linear
// output rows over batch x image rows with one gap row per image boundary, even / odd column planes so that a stride-2 operand
// read is a conflict-free ds_read_b128, optional input prologue = the producer's BatchNorm + SiLU applied while staging).
// Wave w owns one 32-pixel x 32-channel accumulator tile (column tile w % NCT, pixel block w / NCT): per chunk nine operand
// reads (one per tap), nine private float4 weight fragments from the k-quad interleaved pack Wq[(tap * Cin + ci) >> 2][n][4]
// (each re-loaded in place for the next chunk right behind the four MFMAs that consumed it) and 36 MFMAs.  ~110 VGPRs: four to
// five workgroups per CU hide each other's barriers and load latency.  Epilogue: bias, 128-byte row stores, one BatchNorm
// partial row per workgroup (fixed order: bitwise reproducible).  Same results as yh_conv_fwd to fp32 summation order.
// replaces: nn.Conv2d(k=3, s=2, p=1) forward of the down-sampling layers (train.py:408-418, 593-597).
#include "common.h"

namespace {

constexpr int KC = 8;

struct S2 {
    const float *in, *Wq, *bias, *icoef;
    float *out, *stats;
    int icoef_ld, ldi, ldw, ldo;
    int B, H, W, Ho, Wo, K, N;
    int R, C, ncb, ncolb, NCT;      // output block R x C (= 32 * 4 / NCT pixels); column blocks per row; N blocks; column tiles per workgroup
    int PCh, plane, bufsz, tab_ofs; // patch half-row / plane stride (float4 units), one buffer (floats), pixel table offset (floats)
};

template <bool ACT, int NP>
__global__ __launch_bounds__(256) void s2_lds_kernel(const S2 g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int nwg = gridDim.x, orig = blockIdx.x;                  // XCD-aware bijective remap (see wino_lds_kernel)
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int pgrp = lin / g.ncolb, colb = lin - pgrp * g.ncolb;
    const int n0 = colb * g.NCT * 32;
    const int rbk = pgrp / g.ncb, cbk = pgrp - rbk * g.ncb;
    const int g0 = rbk * g.R, c0 = cbk * g.C;                      // first linear output row (over B * Ho) and first output column
    const int b0 = g0 / g.Ho, oy0 = g0 - b0 * g.Ho;
    const int GT = g.B * g.Ho;
    const int Reff = min(g.R, GT - g0), Ceff = min(g.C, g.Wo - c0);
    const int cross = (g0 + Reff - 1) / g.Ho - b0;                 // image boundaries inside the block: one extra patch row each
    const int rows_needed = 2 * Reff + 1 + cross;
    const int PC = 2 * g.C + 1;
    const int wj = wave % g.NCT, wpb = wave / g.NCT;               // this wave's column tile and pixel block

    int *const otab = (int *)(smem + g.tab_ofs);                   // output pixel index of each of the block's pixels (-1: none)
    int rd[3][2];                                                  // float offsets of this lane's pixel: patch rows 0..2, per column parity
    {
        const int p = wpb * 32 + lr;
        int r = p / g.C, c = p - r * g.C;
        const bool pv = r < Reff && c < Ceff;
        const int grow = g0 + r, b = grow / g.Ho, oy = grow - b * g.Ho;
        if (wj == 0 && lh == 0) otab[p] = pv ? (b * g.Ho + oy) * g.Wo + c0 + c : -1;
        const int prow0 = pv ? 2 * r + (b - b0) : 0;               // empty slots read pixel 0's patch (in bounds; never stored)
        if (!pv) c = 0;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int par = 0; par < 2; ++par) rd[dy][par] = (((lh * 2 + par) * g.plane) + (prow0 + dy) * g.PCh + c) * 4;
    }
    // staging plan (as wino_lds_kernel): piece = (k-quad, patch pixel) -> element offset in the input (-1: padding, -2: none) and
    // float offset in a patch buffer
    int gofs[NP], ldst[NP];
    const int npieces = rows_needed * PC * 2;
    const int qq = t & 1;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int pid = t + 256 * k, pix = pid >> 1;
        const int prow = pix / PC, pcol = pix - prow * PC;
        int b, iy;
        const int seg0 = 2 * (g.Ho - oy0) + 1;                     // patch rows of the first image
        if (prow < seg0) { b = b0; iy = 2 * oy0 - 1 + prow; }
        else { const int pr = prow - seg0, sgm = pr / (2 * g.Ho + 1); b = b0 + 1 + sgm; iy = pr - sgm * (2 * g.Ho + 1) - 1; }
        const int ix = 2 * c0 - 1 + pcol;
        const bool ok = b < g.B && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        gofs[k] = pid < npieces ? (ok ? ((b * g.H + iy) * g.W + ix) * g.ldi + 4 * qq : -1) : -2;
        ldst[k] = (((qq * 2 + (pcol & 1)) * g.plane) + prow * g.PCh + (pcol >> 1)) * 4;
    }
    const gfloat *const ing = yh_global(g.in);
    const int kq4 = g.K >> 2;
    const int n = n0 + 32 * wj + lr;
    const float *const pw = g.Wq + ((size_t)lh * g.ldw + (n < g.ldw ? n : 0)) * 4;
    const size_t wtap = (size_t)kq4 * g.ldw * 4, wchunk = (size_t)2 * g.ldw * 4;

    typedef const __attribute__((address_space(4))) f32x4 cf32x4;
    f32x4 csc0, csc1, csh0, csh1, cg0, cg1;
    f32x4 sr[NP];
    auto stage_load = [&](int c) {     // unconditional loads (see wino_lds_kernel)
#pragma unroll
        for (int k = 0; k < NP; ++k) sr[k] = *(const YH_GLOBAL f32x4 *)(ing + (gofs[k] > 0 ? gofs[k] : 0) + c * KC);
        if constexpr (ACT) {
            cf32x4 *ps = (cf32x4 *)(g.icoef + c * KC), *ph = (cf32x4 *)(g.icoef + g.icoef_ld + c * KC), *pg = (cf32x4 *)(g.icoef + 2 * g.icoef_ld + c * KC);
            csc0 = ps[0]; csc1 = ps[1]; csh0 = ph[0]; csh1 = ph[1]; cg0 = pg[0]; cg1 = pg[1];
        }
    };
    auto stage_store = [&](float *buf) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            f32x4 v = sr[k];
            if constexpr (ACT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = yh_prologue(v[e], qq ? csc1[e] : csc0[e], qq ? csh1[e] : csh0[e], qq ? cg1[e] : cg0[e]);
            }
            if (gofs[k] < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};        // padding is zero AFTER the activation
            if (gofs[k] != -2) *(f32x4 *)(buf + ldst[k]) = v;
        }
    };

    const int nchunks = g.K / KC, last = nchunks - 1;
    f32x4 bq[9];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    stage_load(0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) bq[tap] = *(const f32x4 *)(pw + tap * wtap);
    stage_store(smem);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const float *const cur = smem + (c & 1) * g.bufsz;
        float *const nxt = smem + ((c & 1) ^ 1) * g.bufsz;
        const bool more = c < last;                                // workgroup-uniform
        const int cn = more ? c + 1 : c;
        stage_load(cn);                                            // in flight under this chunk's MFMAs (the tail re-loads: nothing branches)
        f32x4 a[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - 3 * dy;
            a[tap] = *(const f32x4 *)(cur + rd[dy][dx & 1] + 4 * (dx >> 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tap][e], bq[tap][e], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bq[tap] = *(const f32x4 *)(pw + tap * wtap + cn * wchunk);      // next chunk's fragment, in place
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) stage_store(nxt);
        __syncthreads();                                           // cur is consumed by every wave, nxt is complete
    }

    // ---- epilogue: register r of lane (lr, lh) = pixel (r & 3) + 8 (r >> 2) + 4 lh of the wave's block, channel n ------------
    const bool nok = n < g.N;
    const float bias = (g.bias && nok) ? g.bias[n] : 0.f;
    int op[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const i32x4 x = *(const i32x4 *)(otab + wpb * 32 + 8 * q + 4 * lh);
        op[4 * q] = x[0]; op[4 * q + 1] = x[1]; op[4 * q + 2] = x[2]; op[4 * q + 3] = x[3];
    }
    const bool whole = Reff * Ceff == 32 * (4 / g.NCT) && n0 + 32 * g.NCT <= g.N;   // workgroup-uniform: every pixel slot and column valid
    gfloat *const outg = yh_global(g.out);
    float cs = 0.f, cq = 0.f;
    if (whole) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r] + bias;
            outg[(unsigned)(op[r] * g.ldo + n)] = v;
            cs += v;
            cq += v * v;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (nok && op[r] >= 0) {
                const float v = acc[r] + bias;
                outg[(unsigned)(op[r] * g.ldo + n)] = v;
                cs += v;
                cq += v * v;
            }
    }
    if (g.stats) {
        float *red = smem;                                         // [4 waves][32][2]  (the patch buffers are consumed: barrier above)
        cs += __shfl_xor(cs, 32);
        cq += __shfl_xor(cq, 32);
        if (lh == 0) { red[(wave * 32 + lr) * 2] = cs; red[(wave * 32 + lr) * 2 + 1] = cq; }
        __syncthreads();
        if (t < 32 * g.NCT) {
            const int j = t >> 5, cl = t & 31, nn = n0 + t;
            if (nn < g.N) {
                float s0 = 0.f, s1 = 0.f;
                for (int pb = 0; pb < 4 / g.NCT; ++pb) {           // pixel blocks in order
                    s0 += red[((pb * g.NCT + j) * 32 + cl) * 2];
                    s1 += red[((pb * g.NCT + j) * 32 + cl) * 2 + 1];
                }
                g.stats[((size_t)pgrp * 2 + 0) * g.N + nn] = s0;
                g.stats[((size_t)pgrp * 2 + 1) * g.N + nn] = s1;
            }
        }
    }
}

// block geometry: R x C output pixels (R C = 32 PB), the fewest workgroups, then the smallest patch; PCh searched for conflict-free
// ds_read_b128 of the stride-2 operand reads (lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: MI355X_MICROARCH.md, LDS)
struct S2Geom {
    int R, C, PR, PCh, plane, bufsz, nrb, ncb;
};
int s2_read_conflicts(int C, int PCh) {
    static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                   {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int worst = 0;
    for (int gi = 0; gi < 2; ++gi) {
        int cnt[16] = {0};
        for (int k = 0; k < 16; ++k) {
            const int l = grp[gi][k], r = l / C, c = l - r * C;
            ++cnt[(2 * r * PCh + c) & 15];
        }
        for (int k = 0; k < 16; ++k) worst = cnt[k] > worst ? cnt[k] : worst;
    }
    return worst;
}
bool s2_geom(int B, int Ho, int Wo, int PB, S2Geom &o) {
    long best = -1;
    const int GT = B * Ho, npx = 32 * PB;
    for (int C = 1; C <= npx && C <= Wo; ++C) {
        int R = npx / C;
        if (R > GT) R = GT;
        if (R < 1) continue;
        const int maxcross = R > 1 ? (R - 1 + Ho - 1) / Ho : 0;
        const int PR = 2 * R + 1 + maxcross, PC = 2 * C + 1;
        if (PR * PC * 2 > 768) continue;
        const int nrb = cdiv(GT, R), ncb = cdiv(Wo, C);
        const long cost = (long)nrb * ncb * 4096 + PR * PC;
        if (best < 0 || cost < best) { best = cost; o.R = R; o.C = C; o.PR = PR; o.nrb = nrb; o.ncb = ncb; }
    }
    if (best < 0) return false;
    int bp = o.C + 1, bw = 1 << 30;
    for (int p = o.C + 1; p < o.C + 17; ++p) {
        const int w = s2_read_conflicts(o.C, p);
        if (w < bw) { bw = w; bp = p; }
    }
    o.PCh = bp;
    o.plane = o.PR * o.PCh;
    while ((o.plane & 7) != 2) ++o.plane;
    o.bufsz = 4 * o.plane * 4;
    return true;
}
inline int s2_nct(int N) { return N >= 128 ? 4 : 2; }

template <bool ACT>
int s2_launch(S2 &g, const S2Geom &gm, hipStream_t st) {
    const int tab = 2 * gm.bufsz > 256 ? 2 * gm.bufsz : 256;
    g.tab_ofs = tab;
    const size_t smem = (size_t)(tab + 32 * (4 / g.NCT) + 4) * sizeof(float);
    const int np = gm.PR * (2 * gm.C + 1) * 2 <= 512 ? 2 : 3;
    dim3 grid(gm.nrb * gm.ncb * g.ncolb);
    if (np == 2) {
        if (int rc = yh_ensure_dyn_smem((const void *)s2_lds_kernel<ACT, 2>, smem)) return rc;
        hipLaunchKernelGGL((s2_lds_kernel<ACT, 2>), grid, dim3(256), smem, st, g);
    } else {
        if (int rc = yh_ensure_dyn_smem((const void *)s2_lds_kernel<ACT, 3>, smem)) return rc;
        hipLaunchKernelGGL((s2_lds_kernel<ACT, 3>), grid, dim3(256), smem, st, g);
    }
    YH_CHECK_LAUNCH("conv_s2");
    return 0;
}

}  // namespace

extern "C" int yh_conv_s2_ok(int B, int H, int W, int Cin, int Cout) {
    S2Geom gm{};
    if (B <= 0 || H < 3 || W < 3 || Cin % 8 || Cin < 8 || Cout % 32) return 0;
    return s2_geom(B, (H - 1) / 2 + 1, (W - 1) / 2 + 1, 4 / s2_nct(Cout), gm) ? 1 : 0;
}

extern "C" int yh_conv_s2_blocks(int B, int H, int W, int Cout) {
    S2Geom gm{};
    if (!s2_geom(B, (H - 1) / 2 + 1, (W - 1) / 2 + 1, 4 / s2_nct(Cout), gm)) return -1;
    return gm.nrb * gm.ncb;
}

extern "C" int yh_conv_s2_fwd_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *wq, int ldw, const float *bias,
                                  float *y, int ldy, float *bn_partials, int B, int H, int W, int Cin, int Cout, void *stream) {
    YH_REQUIRE(x && wq && y && yh_conv_s2_ok(B, H, W, Cin, Cout), "conv_s2_fwd: unsupported problem (3x3 stride 2, Cin %% 8 == 0, Cout %% 32 == 0)");
    YH_REQUIRE(ldx >= Cin && ldx % 4 == 0 && ldy >= Cout && ldw >= Cout && ldw % 4 == 0 && (((uintptr_t)x | (uintptr_t)wq) & 15) == 0,
               "conv_s2_fwd: 16-byte addressable input rows and weight pack required");
    YH_REQUIRE(!icoef || ((((uintptr_t)icoef) & 15) == 0 && icoef_ld % 4 == 0 && icoef_ld >= Cin), "conv_s2_fwd: misaligned prologue table");
    S2 g{};
    g.in = x; g.Wq = wq; g.bias = bias; g.icoef = icoef; g.icoef_ld = icoef_ld; g.out = y; g.stats = bn_partials;
    g.ldi = ldx; g.ldw = ldw; g.ldo = ldy; g.B = B; g.H = H; g.W = W; g.Ho = (H - 1) / 2 + 1; g.Wo = (W - 1) / 2 + 1; g.K = Cin; g.N = Cout;
    YH_REQUIRE((int64_t)B * H * W * ldx < (1ll << 31) && (int64_t)B * g.Ho * g.Wo * ldy < (1ll << 31), "conv_s2_fwd: tensors exceed 32-bit element offsets");
    g.NCT = s2_nct(Cout);
    g.ncolb = cdiv(Cout, 32 * g.NCT);
    S2Geom gm{};
    YH_REQUIRE(s2_geom(B, g.Ho, g.Wo, 4 / g.NCT, gm), "conv_s2_fwd: no block geometry");
    g.R = gm.R; g.C = gm.C; g.ncb = gm.ncb; g.PCh = gm.PCh; g.plane = gm.plane; g.bufsz = gm.bufsz;
    return icoef ? s2_launch<true>(g, gm, (hipStream_t)stream) : s2_launch<false>(g, gm, (hipStream_t)stream);
}
