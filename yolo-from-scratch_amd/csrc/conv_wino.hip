// Winograd F(2x2, 3x3) convolution on v_mfma_f32_32x32x2_f32 (gfx950), NHWC fp32, stride 1, pad 1.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, 4x4 input tile d, 3x3 filter g
//
// With channels the element-wise product becomes 16 independent GEMMs (one per position (xi, nu) of the 4x4
// transformed tile):  M[p][tile][n] = sum_k V[p][tile][k] * U[p][k][n]  -- 4/9 of the multiplies of the direct
// form (2.25x fewer MFMA cycles).  Everything is fused in one kernel (wino_lds_kernel):
//   * a workgroup owns a 2-D block of 32 output tiles x 32*NT channels; wave w owns transform row xi = w (positions 4w .. 4w+3);
//   * per chunk of 8 input channels the block's input patch is staged ONCE through LDS (rounds 1-3 loaded it straight into
//     registers: every pixel was requested 3.4 times and the loop ran at the CU's cache-line rate, not at the MFMA rate);
//   * lane (tile = lane & 31, q = lane >> 5) reads the two patch rows that row xi of B^T d needs (4 pixels x float4 of
//     channels 4q..4q+3 each), applies the transform in registers and holds V[nu][tile][4q..4q+3] -- which is exactly the
//     A operand this lane feeds to the MFMA (row = lane & 31, k-group = lane >> 5);
//   * U is pre-transformed into a k-quad interleaved layout [16][K/4][N][4], so a B fragment (4 k values of one
//     column) is one float4 load, coalesced over the 32 columns of a tile; it is private to the wave;
//   * epilogue: the nu-contraction of A^T M A is lane-local (the four positions of a wave), the xi-contraction
//     goes through LDS; then bias / accumulate / BatchNorm partial sums exactly as the direct kernel.
// The same kernel computes backward-data (correlation of dY with the flipped, transposed filter).
// Numerics: fp32 throughout; F(2,3) transform constants are {0, +-1, +-1/2}, error growth is a few ulp.
#include "common.h"
#ifdef YH_WINO_STAMPS
#include <stdio.h>
#include <vector>
#endif

namespace {

constexpr int KC = 8;           // input channels per chunk (2 k-quads: one per lane half)
constexpr int TPB = 32;         // tiles per workgroup

struct Wino {
    const float *in, *U, *bias;
    float *out, *stats;
    int ldi, ldu, ldo;
    int B, H, W, K, N;
    int TW, TPI, ntiles;        // tile columns per row, tiles per image, total tiles
    int ncol;                   // column blocks (set by the launcher)
    int accumulate;
    const float *res;           // inference epilogue (eval-mode BatchNorm folded into U / bias): + residual after the activation,
    int ldr, act, up2;          // act: SiLU on (acc + bias); up2: each output pixel replicated 2x2 (out is (B,2H,2W))
    unsigned tw_magic, tpi_magic;
    int tw_shift, tpi_shift;
    // LDS-staged kernel (wino_lds_kernel): fused producer activation, block geometry and patch layout (set by the launcher)
    const float *icoef;             // input prologue table [scale | shift | gate] (icoef_ld floats apart): input = z = x * scale[k] + shift[k],
    int icoef_ld;                   // then silu(z) where gate[k] != 0; applied while the patch is staged; null = plain input
    int TH, R, C, ncb;              // tile rows per image; tile rows x tile columns per workgroup (R C <= 32); column blocks per tile row
    unsigned dm[6];                 // division by multiplication (fdiv) for the block's index arithmetic: ncol, ncb, TH, C, 2 C + 2, 2 TH + 2
    int ds[6];
    int PCh, plane, bufsz, toff_ofs;// half-row stride and plane stride of the patch (float4 units); one patch buffer, tile-offset table (floats)
#ifdef YH_WINO_STAMPS
    unsigned long long *dbg;    // diagnostic build only: per-workgroup phase stamps
#endif
};

__device__ __forceinline__ int fdiv(int n, unsigned magic, int shift) {
    return shift < 0 ? n : (int)(__umulhi((unsigned)n, magic) >> shift);
}

// Output transform + store + BatchNorm partial sums.  `toff[32]` (LDS):
// pixel index of each tile's top-left output ((b H + 2 ty) W + 2 tx), -1 for an empty tile slot; `whole_tiles`: all 32 slots hold tiles
// (wave-uniform); `tgrp`: the workgroup's row of the BatchNorm partial-sum table.  Every wave has passed a barrier after its last use
// of `smem` before the call.
template <int NT>
__device__ __forceinline__ void wino_epilogue(const Wino &g, f32x16 (&acc)[4][NT], float *smem, const int *toff, bool whole_tiles,
                                              int tgrp, int n0) {
    constexpr int BNW = 32 * NT;
    constexpr int IMG_FLOATS = 6 * 4 * NT * 256;                   // training epilogue: six register images of 16 NT floats per lane
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    // ---- output transform, training form (no activation / residual / upsample / BatchNorm table) ----------------------------
    // Y = A^T M A per tile.  The nu-contraction is lane-local: T[c] = (m0 + m1) + m2 | (m1 - m2) - m3 (output column c).  The
    // xi-contraction needs three of the four waves' T: wave w produces output row w >> 1, column w & 1 of every tile, so the
    // waves swap whole REGISTER IMAGES through LDS (six images, 16-byte writes and reads, lane-contiguous: conflict-free) and
    // each stores its quarter of the outputs straight from the MFMA layout (32 consecutive channels = 128 bytes per row).
    // Same values bit for bit as the general epilogue below ((T0 + T1) + T2, (T1 - T2) - T3); in-kernel stamps: that one
    // took 11 000 cycles per workgroup at one wave per SIMD (64 4-byte LDS writes + 64 reads per lane, per-tile divisions).
    if (!(g.act | g.up2 | (g.res != nullptr))) {
        float T[2][NT][16];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m0 = acc[0][j][r], m1 = acc[1][j][r], m2 = acc[2][j][r], m3 = acc[3][j][r];
                T[0][j][r] = m0 + m1 + m2;
                T[1][j][r] = m1 - m2 - m3;
            }
        auto put = [&](int id, const float (&v)[NT][16]) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(f32x4 *)(smem + ((size_t)((id * NT + j) * 4 + q) * 64 + lane) * 4) = f32x4{v[j][4 * q], v[j][4 * q + 1], v[j][4 * q + 2], v[j][4 * q + 3]};
        };
        auto get = [&](int id, float (&v)[NT][16]) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 x = *(const f32x4 *)(smem + ((size_t)((id * NT + j) * 4 + q) * 64 + lane) * 4);
                    v[j][4 * q] = x[0]; v[j][4 * q + 1] = x[1]; v[j][4 * q + 2] = x[2]; v[j][4 * q + 3] = x[3];
                }
        };
        // images: 0 = T0[1], 1 = T1[0], 2 = T1[1], 3 = T2[0], 4 = T2[1], 5 = T3[0]
        if (wave == 0) put(0, T[1]);
        else if (wave == 1) { put(1, T[0]); put(2, T[1]); }
        else if (wave == 2) { put(3, T[0]); put(4, T[1]); }
        else put(5, T[0]);
        __syncthreads();
        float Y[NT][16], L1[NT][16], L2[NT][16];
        if (wave == 0) {            // row 0, column 0: (T0[0] + T1[0]) + T2[0]
            get(1, L1); get(3, L2);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[j][r] = (T[0][j][r] + L1[j][r]) + L2[j][r];
        } else if (wave == 1) {     // row 0, column 1: (T0[1] + T1[1]) + T2[1]
            get(0, L1); get(4, L2);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[j][r] = (L1[j][r] + T[1][j][r]) + L2[j][r];
        } else if (wave == 2) {     // row 1, column 0: (T1[0] - T2[0]) - T3[0]
            get(1, L1); get(5, L2);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[j][r] = (L1[j][r] - T[0][j][r]) - L2[j][r];
        } else {                    // row 1, column 1: (T1[1] - T2[1]) - T3[1]
            get(2, L1); get(4, L2);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[j][r] = (L1[j][r] - L2[j][r]) - T[1][j][r];
        }
        // register r of this lane = tile (r & 3) + 8 (r >> 2) + 4 lh of the group: four consecutive offsets per 16-byte read
        int tp[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const i32x4 x = *(const i32x4 *)(toff + 8 * q + 4 * lh);
            tp[4 * q] = x[0]; tp[4 * q + 1] = x[1]; tp[4 * q + 2] = x[2]; tp[4 * q + 3] = x[3];
        }
        const int pc = (wave >> 1) * g.W + (wave & 1);
        gfloat *const outg = yh_global(g.out);
        const bool whole = whole_tiles && n0 + BNW <= g.N;                   // wave-uniform
        float cs[NT], cq[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + 32 * j + lr;
            const bool nok = n < g.N;
            const float bias = (g.bias && nok) ? g.bias[n] : 0.f;
            cs[j] = cq[j] = 0.f;
            if (whole) {
                if (!g.accumulate) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = Y[j][r] + bias;
                        outg[(unsigned)((tp[r] + pc) * g.ldo + n)] = v;
                        cs[j] += v;
                        cq[j] += v * v;
                    }
                } else {
                    float old[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[r] = outg[(unsigned)((tp[r] + pc) * g.ldo + n)];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = Y[j][r] + bias + old[r];
                        outg[(unsigned)((tp[r] + pc) * g.ldo + n)] = v;
                        cs[j] += v;
                        cq[j] += v * v;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (nok && tp[r] >= 0) {
                        gfloat *o = outg + (unsigned)((tp[r] + pc) * g.ldo + n);
                        float v = Y[j][r] + bias;
                        if (g.accumulate) v += *o;
                        *o = v;
                        cs[j] += v;
                        cq[j] += v * v;
                    }
            }
        }
        if (g.stats) {
            float *red = smem + IMG_FLOATS + 32;                          // [4 waves][BNW][2]
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float s = cs[j] + __shfl_xor(cs[j], 32), q = cq[j] + __shfl_xor(cq[j], 32);
                if (lh == 0) { red[(wave * BNW + 32 * j + lr) * 2] = s; red[(wave * BNW + 32 * j + lr) * 2 + 1] = q; }
            }
            __syncthreads();
            if (t < BNW && n0 + t < g.N) {
                const float a0 = (red[t * 2] + red[(BNW + t) * 2]) + (red[(2 * BNW + t) * 2] + red[(3 * BNW + t) * 2]);
                const float a1 = (red[t * 2 + 1] + red[(BNW + t) * 2 + 1]) + (red[(2 * BNW + t) * 2 + 1] + red[(3 * BNW + t) * 2 + 1]);
                g.stats[((size_t)tgrp * 2 + 0) * g.N + n0 + t] = a0;
                g.stats[((size_t)tgrp * 2 + 1) * g.N + n0 + t] = a1;
            }
        }
        return;
    }

    // ---- output transform, general form.  nu-contraction (lane-local): s0 = m0+m1+m2, s1 = m1-m2-m3 -------------------
    float *S = smem;                           // [xi 4][j 2][tile 32][ch BNW]
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            float m0 = acc[0][j][r], m1 = acc[1][j][r], m2 = acc[2][j][r], m3 = acc[3][j][r];
            S[((wave * 2 + 0) * TPB + row) * BNW + j * 32 + lr] = m0 + m1 + m2;
            S[((wave * 2 + 1) * TPB + row) * BNW + j * 32 + lr] = m1 - m2 - m3;
        }
    __syncthreads();
    // xi-contraction + store: thread = (channel, tile group); 256 / BNW tile groups
    constexpr int TG = 256 / BNW;
    const int ch = t % BNW, tgp = t / BNW, n = n0 + ch;
    const bool nok = n < g.N;
    const float bias = (g.bias && nok) ? g.bias[n] : 0.f;
    float csum = 0.f, csq = 0.f;
    for (int it = 0; it < TPB / TG; ++it) {
        const int tl = tgp + TG * it, tp0 = toff[tl];      // pixel index of the tile's top-left output; -1 = no such tile
        if (tp0 < 0 || !nok) continue;
        float s[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int j = 0; j < 2; ++j) s[xi][j] = S[((xi * 2 + j) * TPB + tl) * BNW + ch];
        float *o = g.out + (size_t)tp0 * g.ldo + n;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float y0 = s[0][j] + s[1][j] + s[2][j] + bias;
            float y1 = s[1][j] - s[2][j] - s[3][j] + bias;
            float *o0 = o + (size_t)j * g.ldo, *o1 = o0 + (size_t)g.W * g.ldo;
            if (g.accumulate) { y0 += *o0; y1 += *o1; }
            if (g.act | g.up2 | (g.res != nullptr)) {          // inference form (uniform branch)
                if (g.act) { y0 = y0 * yh_sigmoid(y0); y1 = y1 * yh_sigmoid(y1); }
                const size_t p0 = (size_t)tp0 + j;
                if (g.res) { y0 += g.res[p0 * g.ldr + n]; y1 += g.res[(p0 + g.W) * g.ldr + n]; }
                if (g.up2) {
                    // output (b, 2 ty, 2 tx + j) of the (B, H, W) lattice = row tp0 / W, column tp0 % W + j; replicated 2 x 2 into (B, 2H, 2W)
                    const unsigned rowg = (unsigned)tp0 / (unsigned)g.W, colg = (unsigned)tp0 - rowg * (unsigned)g.W + j;
                    const size_t rs = (size_t)(2 * g.W) * g.ldo;
                    float *u0 = g.out + ((size_t)(2 * rowg) * 2 * g.W + 2 * colg) * g.ldo + n;
                    float *u1 = u0 + 2 * rs;
                    u0[0] = y0; u0[g.ldo] = y0; u0[rs] = y0; u0[rs + g.ldo] = y0;
                    u1[0] = y1; u1[g.ldo] = y1; u1[rs] = y1; u1[rs + g.ldo] = y1;
                    continue;
                }
            }
            *o0 = y0; *o1 = y1;
            csum += y0 + y1;
            csq += y0 * y0 + y1 * y1;
        }
    }
    if (g.stats) {
        float *red = smem + 4 * 2 * TPB * BNW;   // [TG][BNW][2]
        red[(tgp * BNW + ch) * 2 + 0] = csum;
        red[(tgp * BNW + ch) * 2 + 1] = csq;
        __syncthreads();
        if (tgp == 0 && nok) {                   // ch == t: the thread that looked its column up
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int w = 0; w < TG; ++w) { a0 += red[(w * BNW + t) * 2]; a1 += red[(w * BNW + t) * 2 + 1]; }
            g.stats[((size_t)tgrp * 2 + 0) * g.N + n] = a0;
            g.stats[((size_t)tgrp * 2 + 1) * g.N + n] = a1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same convolution with the INPUT PATCH STAGED THROUGH LDS (round 4).  What bounded wino_kernel above was not bytes but cache
// lines: a lane's 16-byte pixel loads touch 64 different lines per instruction and the four transform-row waves request every
// pixel 3.4 times, so the loop ran at the CU's line rate (1.45x its MFMA time).  Here a workgroup owns a 2-D block of R x C tiles
// (R C <= 32; rows are LINEAR tile rows over batch x image rows, so blocks may span images) and, per chunk of 8 input channels,
// reads the block's (2R+2) x (2C+2) input patch ONCE (16 bytes per thread and piece, two or three pieces per thread), applies the
// producer's BatchNorm scale / shift + SiLU on the way if asked to (the normalised activation then never exists in HBM; padding
// pixels are zeros AFTER the activation, as in the reference), and parks it in LDS as P[k-quad][column parity][row][column / 2]
// float4s.  A wave's operand read is a conflict-free ds_read_b128 for the block shapes the launcher picks (the half-row stride
// PCh is searched on the host); two patch buffers, one barrier per chunk.  Weights: as before (private coalesced float4 loads of
// U), but each transform row's fragments are re-loaded IN PLACE for the next chunk right behind the eight MFMAs that consumed
// them -- prefetch distance one chunk at no register cost.  Transform arithmetic, MFMA order per accumulator and the epilogue are
// those of wino_kernel: outputs are bit-identical to it.
template <int NT, bool ACT, int NP>                               // NP: staging pieces per thread (2, or 3 for the tallest patches)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, NT == 1 ? 3 : 2))) void wino_lds_kernel(const Wino g) {
    constexpr int BNW = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // patch buffers during the loop, the epilogue's images afterwards
#ifdef YH_WINO_STAMPS
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int nwg = gridDim.x, orig = blockIdx.x;                  // XCD-aware bijective remap (see wino_kernel)
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int tgrp = fdiv(lin, g.dm[0], g.ds[0]), colb = lin - tgrp * g.ncol;
    const int n0 = colb * BNW;
    const int rbk = fdiv(tgrp, g.dm[1], g.ds[1]), cbk = tgrp - rbk * g.ncb;
    const int g0 = rbk * g.R, c0 = cbk * g.C;                      // first linear tile row (over B * TH) and first tile column
    const int b0 = fdiv(g0, g.dm[2], g.ds[2]), ty0 = g0 - b0 * g.TH;
    const int GT = g.B * g.TH;
    const int Reff = min(g.R, GT - g0), Ceff = min(g.C, g.TW - c0);
    const int cross = fdiv(g0 + Reff - 1, g.dm[2], g.ds[2]) - b0;                 // image boundaries inside the block: two extra patch rows each
    const int rows_needed = 2 * Reff + 2 + 2 * cross;
    const int PC = 2 * g.C + 2;

    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg = wave == 1 ? 1.f : -1.f;

    int *const toff = (int *)(smem + g.toff_ofs);
    int rdA[2], rdB[2];                                            // float offsets of patch rows ra / rb of this lane's tile, per column parity
    {
        int r = fdiv(lr, g.dm[3], g.ds[3]), c = lr - r * g.C;
        const bool tv = r < Reff && c < Ceff;
        const int grow = g0 + r, b = fdiv(grow, g.dm[2], g.ds[2]), ty = grow - b * g.TH;
        if (wave == 0 && lh == 0) toff[lr] = tv ? (b * g.H + 2 * ty) * g.W + 2 * (c0 + c) : -1;
        const int prow0 = tv ? 2 * r + 2 * (b - b0) : 0;           // empty slots read tile 0's pixels (in bounds; results never stored)
        if (!tv) c = 0;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            rdA[par] = (((lh * 2 + par) * g.plane) + (prow0 + ra) * g.PCh + c) * 4;
            rdB[par] = (((lh * 2 + par) * g.plane) + (prow0 + rb) * g.PCh + c) * 4;
        }
    }
    // staging plan, fixed for the kernel: piece = (k-quad, patch pixel) -> element offset in the input (-1: padding, stays zero;
    // -2: no such piece) and float offset in a patch buffer
    int gofs[NP], ldst[NP];
    const int npieces = rows_needed * PC * 2;
    const int qq = t & 1;                                          // the k-quad of every piece of this thread (256 is even)
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int pid = t + 256 * k, pix = pid >> 1;
        const int prow = fdiv(pix, g.dm[4], g.ds[4]), pcol = pix - prow * PC;
        int b, iy;
        const int seg0 = 2 * (g.TH - ty0) + 2;                     // patch rows of the first image
        if (prow < seg0) { b = b0; iy = 2 * ty0 - 1 + prow; }
        else { const int pr = prow - seg0, sgm = fdiv(pr, g.dm[5], g.ds[5]); b = b0 + 1 + sgm; iy = pr - sgm * (2 * g.TH + 2) - 1; }
        const int ix = 2 * c0 - 1 + pcol;
        const bool ok = b < g.B && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        gofs[k] = pid < npieces ? (ok ? ((b * g.H + iy) * g.W + ix) * g.ldi + 4 * qq : -1) : -2;
        ldst[k] = (((qq * 2 + (pcol & 1)) * g.plane) + prow * g.PCh + (pcol >> 1)) * 4;
    }
    const gfloat *const ing = yh_global(g.in);

    const int kq4 = g.K >> 2;
    const float *ub[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 32 + lr;
        ub[j] = g.U + ((size_t)((4 * wave) * kq4 + lh) * g.ldu + (n < g.ldu ? n : 0)) * 4;
    }
    const size_t upos = (size_t)kq4 * g.ldu * 4;      // stride between positions
    const size_t uchunk = (size_t)2 * g.ldu * 4;      // stride between chunks

    f32x16 acc[4][NT];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[v][j][r] = 0.f;

    // the producer's scale / shift of the chunk's 8 channels: wave-uniform, so they travel through the scalar cache into SGPRs
    // (written by an earlier kernel: coherent at the launch boundary) and cost no vector register while the MFMAs run
    typedef const __attribute__((address_space(4))) f32x4 cf32x4;
    f32x4 csc0, csc1, csh0, csh1, cg0, cg1;
    // UNCONDITIONAL loads (padding / absent pieces re-read element 0 of the chunk and are zeroed / dropped at the store): a load
    // under a branch makes every later counted s_waitcnt a vmcnt(0), which would wait for the prefetches in front of the MFMAs
    auto stage_load = [&](int c, f32x4 (&sr)[NP]) {
#pragma unroll
        for (int k = 0; k < NP; ++k) sr[k] = *(const YH_GLOBAL f32x4 *)(ing + (gofs[k] > 0 ? gofs[k] : 0) + c * KC);
    };
    auto coef_load = [&](int c) {
        if constexpr (ACT) {
            cf32x4 *ps = (cf32x4 *)(g.icoef + c * KC), *ph = (cf32x4 *)(g.icoef + g.icoef_ld + c * KC),
                   *pg = (cf32x4 *)(g.icoef + 2 * g.icoef_ld + c * KC);
            csc0 = ps[0]; csc1 = ps[1]; csh0 = ph[0]; csh1 = ph[1]; cg0 = pg[0]; cg1 = pg[1];
        }
    };
    // the activation of the parked pieces, in place in the staging registers (padding is zero AFTER the activation)
    auto stage_math1 = [&](f32x4 (&sr)[NP], int i) {               // element i of the NP x 4 staged values
        const int k = i >> 2, e = i & 3;
        float v = sr[k][e];
        // (pure arithmetic is not ordered against sched_barrier by the instruction selector: the empty volatile asm statements tie
        // the slice to its place in the instruction stream, between two MFMAs)
        asm volatile("" : "+v"(v));
        if constexpr (ACT) v = yh_prologue(v, qq ? csc1[e] : csc0[e], qq ? csh1[e] : csh0[e], qq ? cg1[e] : cg0[e]);
        v = gofs[k] < 0 ? 0.f : v;
        asm volatile("" : "+v"(v));
        sr[k][e] = v;
    };
    auto stage_math = [&](f32x4 (&sr)[NP]) {
#pragma unroll
        for (int i = 0; i < 4 * NP; ++i) stage_math1(sr, i);
    };
    auto stage_store = [&](float *buf, const f32x4 (&sr)[NP]) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (gofs[k] != -2) *(f32x4 *)(buf + ldst[k]) = sr[k];
        }
    };

    // Software pipeline over chunks of 8 input channels (nchunks is even).  In iteration c a wave
    //   issues the piece loads of chunk c + 2 (two register sets: the HBM latency has two chunks of MFMAs to hide behind),
    //   runs MFMA groups 0, 1 (transform rows' positions nu = 0, 1; each group re-loads its weight fragments for chunk c + 1 in place),
    //   applies the activation to the pieces of chunk c + 1 and parks them in the other patch buffer, BARRIER,
    //   issues the eight operand reads of chunk c + 1, runs group 2 over their latency, transforms (V[0..2] of chunk c + 1 overwrite
    //   the dead ones), runs group 3, finishes V[3]:
    // the matrix pipe sees back-to-back MFMAs across the chunk boundary from a SINGLE wave (the first form of this loop relied on the
    // second resident workgroup to fill ~2 900 idle cycles per chunk: 2.4x the MFMA time whenever a workgroup ran alone on its CU).
    const int nchunks = g.K / KC, last = nchunks - 1;
    f32x4 u[4][NT], sA[NP], sB[NP], V[4], tt[4], d[2][4];
    auto group = [&](int v, int cn) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[v][e], u[v][j][e], acc[v][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NT; ++j) u[v][j] = *(const f32x4 *)(ub[j] + v * upos + cn * uchunk);   // next chunk, in place
        __builtin_amdgcn_sched_barrier(0);
    };
    auto read_patch = [&](const float *buf) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            d[0][cc] = *(const f32x4 *)(buf + rdA[cc & 1] + 4 * (cc >> 1));
            d[1][cc] = *(const f32x4 *)(buf + rdB[cc & 1] + 4 * (cc >> 1));
        }
    };
    auto iteration = [&](int c, f32x4 (&sNext)[NP], f32x4 (&sFar)[NP]) {
        const int c1 = c + 1 < last ? c + 1 : last, c2 = c + 2 < last ? c + 2 : last;     // clamped: the tail re-loads, nothing branches
        float *const nxt = smem + ((c & 1) ^ 1) * g.bufsz;
        stage_load(c2, sFar);
        coef_load(c1);
        __builtin_amdgcn_sched_barrier(0);
        group(0, c1);
        {   // group 1 with the activation of the next chunk's pieces issued BETWEEN its MFMAs (a slice of the ~100 vector instructions
            // behind each one, order pinned): in one lump ahead of the store they left the matrix pipe idle once per chunk
            constexpr int NM = 4 * NT, NE = 4 * NP;
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const int e = i / NT, j = i - e * NT;
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[1][e], u[1][j][e], acc[1][j], 0, 0, 0);
#pragma unroll
                for (int q = i * NE / NM; q < (i + 1) * NE / NM; ++q) stage_math1(sNext, q);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) u[1][j] = *(const f32x4 *)(ub[j] + upos + c1 * uchunk);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c < last) stage_store(nxt, sNext);                     // wave-uniform branch around LDS stores only
        __syncthreads();                                           // chunk c + 1 is parked; chunk c - 1's buffer is free again
        read_patch(nxt);
        __builtin_amdgcn_sched_barrier(0);
        group(2, c1);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) tt[cc] = d[0][cc] + sg * d[1][cc];
        V[0] = tt[0] - tt[2];
        V[1] = tt[1] + tt[2];
        V[2] = tt[2] - tt[1];
        __builtin_amdgcn_sched_barrier(0);
        group(3, c1);
        V[3] = tt[1] - tt[3];
        __builtin_amdgcn_sched_barrier(0);
    };
    stage_load(0, sA);
    coef_load(0);
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int j = 0; j < NT; ++j) u[v][j] = *(const f32x4 *)(ub[j] + v * upos);
    stage_math(sA);
    stage_store(smem, sA);
    stage_load(last > 0 ? 1 : 0, sA);
    __syncthreads();
    read_patch(smem);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) tt[cc] = d[0][cc] + sg * d[1][cc];
    V[0] = tt[0] - tt[2];
    V[1] = tt[1] + tt[2];
    V[2] = tt[2] - tt[1];
    V[3] = tt[1] - tt[3];
#ifdef YH_WINO_STAMPS
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
    for (int c = 0; c < nchunks; c += 2) {
        iteration(c, sA, sB);
        iteration(c + 1, sB, sA);
    }
    __syncthreads();                                               // every wave has read its last operands: the epilogue reuses the buffers
#ifdef YH_WINO_STAMPS
    const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif
    wino_epilogue<NT>(g, acc, smem, toff, Reff * Ceff == TPB, tgrp, n0);
#ifdef YH_WINO_STAMPS
    if (g.dbg && t == 0) {
        unsigned long long *d = g.dbg + (size_t)blockIdx.x * 6;
        d[0] = st0; d[1] = st1; d[2] = st2; d[3] = __builtin_amdgcn_s_memtime(); d[4] = rt0; d[5] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// U[pos][k/4][n][k%4] = (G g G^T)[pos] with g = w[n][k][.][.] (forward, k = ci, n = co) or the flipped filter of
// the transposed convolution, g[a][b] = w[k][n][2-a][2-b] (backward-data, k = co, n = ci).
struct WinoWDesc {
    const float *w;
    float *U;
    int Cout, Cin, ldu, bwd;
};

__device__ __forceinline__ void wino_weights_body(const WinoWDesc d, int first, int step) {
    const float *__restrict__ w = d.w;
    float *__restrict__ U = d.U;
    const int Cin = d.Cin, ldu = d.ldu, bwd = d.bwd;
    const int K = bwd ? d.Cout : Cin, N = bwd ? Cin : d.Cout;
    const int total = K * ldu;
    const size_t pstride = (size_t)K * ldu;
    for (int i = first; i < total; i += step) {
        int n = i % ldu, k = i / ldu;
        float gg[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float v = 0.f;
                if (n < N) v = bwd ? w[((size_t)k * Cin + n) * 9 + (2 - a) * 3 + (2 - b)] : w[((size_t)n * Cin + k) * 9 + a * 3 + b];
                gg[a][b] = v;
            }
        float t4[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t4[0][b] = gg[0][b];
            t4[1][b] = 0.5f * (gg[0][b] + gg[1][b] + gg[2][b]);
            t4[2][b] = 0.5f * (gg[0][b] - gg[1][b] + gg[2][b]);
            t4[3][b] = gg[2][b];
        }
        const size_t base = ((size_t)(k >> 2) * ldu + n) * 4 + (k & 3);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float u0 = t4[a][0], u1 = 0.5f * (t4[a][0] + t4[a][1] + t4[a][2]), u2 = 0.5f * (t4[a][0] - t4[a][1] + t4[a][2]),
                  u3 = t4[a][2];
            U[base + (size_t)(a * 4 + 0) * pstride] = u0;
            U[base + (size_t)(a * 4 + 1) * pstride] = u1;
            U[base + (size_t)(a * 4 + 2) * pstride] = u2;
            U[base + (size_t)(a * 4 + 3) * pstride] = u3;
        }
    }
}

__global__ void wino_weights_kernel(const WinoWDesc d) {
    wino_weights_body(d, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}
__global__ void wino_weights_multi_kernel(const WinoWDesc *__restrict__ tab) {
    wino_weights_body(tab[blockIdx.y], blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// ---------------------------------------------------------------------------------------------------------------
// Backward-weight in the Winograd domain:  dU[p][ci][co] = sum_tiles V[p][tile][ci] * dM[p][tile][co],
// dM = A dY A^T (4x4 from the 2x2 output-gradient tile), then dg = G^T dU G.  One MFMA k-step consumes a PAIR of
// tiles: lane (c = lane & 31, h = lane >> 5) transforms tile 2s+h for input channel ci0 + c (A operand) and for
// output channels co0 + NT*c + {0..NT-1} (B operands, a float / float2 load: column c of column-tile nt is channel
// co0 + NT*c + nt) -- both operands are produced in the registers that feed the MFMA, no LDS in the loop.
// Wave w owns transform row xi = w.  The tile range is split over blockIdx.x; every workgroup applies G^T . G to its
// partial dU (nu lane-local, xi through LDS) and writes a [9][Cin][Cout] slab; a fixed-order reduction sums the
// slabs (bitwise reproducible) into OIHW.
struct WinoW {
    const float *x, *dy;
    float *ws;
    int ldx, lddy;
    int B, H, W, Cin, Cout;
    int TW, TPI, ntiles, tps;   // tps: tiles per split (multiple of 8)
    unsigned x_bytes, dy_bytes; // extent of the two views (buffer-load bounds)
    unsigned tw_magic, tpi_magic;
    int tw_shift, tpi_shift;
};

template <int NT>
struct DyVec;
template <>
struct DyVec<1> { typedef float type; };
template <>
struct DyVec<2> { typedef float type __attribute__((ext_vector_type(2))); };

template <int NT>
__device__ __forceinline__ float dy_elem(const typename DyVec<NT>::type &v, int j);
template <>
__device__ __forceinline__ float dy_elem<1>(const float &v, int) { return v; }
template <>
__device__ __forceinline__ float dy_elem<2>(const DyVec<2>::type &v, int j) { return v[j]; }

template <int NT, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void wino_wgrad_kernel(const WinoW g) {
    typedef typename DyVec<NT>::type dyv;
    constexpr int SPS = 4;                     // k-steps (tile pairs) per pipeline stage
    extern __shared__ __attribute__((aligned(16))) float smem[];   // epilogue only: T[4][32][32*NT]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 32 * NT;
    const int t_begin = blockIdx.x * g.tps;
    const int t_end = min(t_begin + g.tps, g.ntiles);

    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg = wave == 1 ? 1.f : -1.f;
    const float ca = wave == 3 ? 0.f : 1.f;                    // row xi of A: (1,0) (1,1) (1,-1) (0,-1)
    const float cb = wave == 0 ? 0.f : (wave == 1 ? 1.f : -1.f);

    // Buffer loads: out-of-range offsets return 0, so padding pixels and tiles past the range cost one select on a
    // 32-bit offset (no pointer arithmetic, no branches).  The per-lane tile cursor (b, ty, tx) advances by two
    // tiles per k-step.
    // Buffer loads: out-of-range offsets return 0, so padding pixels and tiles past the range cost one OR on a
    // 32-bit offset (bit 31 set = beyond num_records: views span < 2 GiB); no pointer arithmetic, no branches.
    // The per-lane tile cursor (b, ty, tx) advances by two tiles per k-step.
    // (Measured: moving the eight wave-uniform pixel displacements into the scalar-offset operand is 20% SLOWER.)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)g.x, 0, g.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)g.dy, 0, g.dy_bytes, 0x00020000);
    const unsigned xlane = (unsigned)(ci0 + lr) * 4u, dlane = (unsigned)(co0 + NT * lr) * 4u;
    const int ldx4 = g.ldx * 4, ldd4 = g.lddy * 4, TH = g.H / 2;
    int dxo[2][4];                             // wave-uniform byte offsets of the 2 x 4 input pixels from the tile origin
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) dxo[rr][cc] = ((((rr ? rb : ra) - 1) * g.W) + cc - 1) * ldx4;
    int tg = t_begin + lh, cb_, cty, ctx;
    {
        int tgc = tg < g.ntiles ? tg : 0;
        cb_ = fdiv(tgc, g.tpi_magic, g.tpi_shift);
        int r = tgc - cb_ * g.TPI;
        cty = fdiv(r, g.tw_magic, g.tw_shift);
        ctx = r - cty * g.TW;
    }

    float XA[SPS][2][4], XB[SPS][2][4];
    dyv DA[SPS][2][2], DB[SPS][2][2];
    auto load_stage = [&](float (&X)[SPS][2][4], dyv (&D)[SPS][2][2]) {
#pragma unroll
        for (int q = 0; q < SPS; ++q) {
            const bool tv = tg < t_end;
            const int pix = __mul24(__mul24(cb_, g.H) + 2 * cty, g.W) + 2 * ctx;    // top-left OUTPUT pixel of the tile
            const unsigned xo = (unsigned)__mul24(pix, ldx4) + xlane, dof = (unsigned)__mul24(pix, ldd4) + dlane;
            const bool c0 = ctx > 0, c3 = ctx < g.TW - 1;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int iy = 2 * cty + (rr ? rb : ra) - 1;
                const bool rok = tv & ((unsigned)iy < (unsigned)g.H);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const bool ok = rok & (cc == 0 ? c0 : (cc == 3 ? c3 : true));
                    X[q][rr][cc] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (xo + (unsigned)dxo[rr][cc]) | ((unsigned)!ok << 31), 0, 0));
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned o = (dof + (unsigned)((i * g.W + j) * ldd4)) | ((unsigned)!tv << 31);
                    if constexpr (NT == 1) D[q][i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, o, 0, 0));
                    else D[q][i][j] = __builtin_bit_cast(dyv, __builtin_amdgcn_raw_buffer_load_b64(rd, o, 0, 0));
                }
            tg += 2;                           // branch-free cursor advance
            ctx += 2;
            const int wrap = ctx >= g.TW ? 1 : 0;
            ctx -= wrap ? g.TW : 0;
            cty += wrap;
            const int wrap2 = cty >= TH ? 1 : 0;
            cty = wrap2 ? 0 : cty;
            cb_ += wrap2;
        }
    };

    f32x16 acc[4][NT];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[v][j][r] = 0.f;

    auto compute = [&](const float (&X)[SPS][2][4], const dyv (&D)[SPS][2][2]) {
#pragma unroll
        for (int q = 0; q < SPS; ++q) {
            float tt[4], V[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) tt[cc] = X[q][0][cc] + sg * X[q][1][cc];
            V[0] = tt[0] - tt[2];
            V[1] = tt[1] + tt[2];
            V[2] = tt[2] - tt[1];
            V[3] = tt[1] - tt[3];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float r0 = ca * dy_elem<NT>(D[q][0][0], j) + cb * dy_elem<NT>(D[q][1][0], j);
                float r1 = ca * dy_elem<NT>(D[q][0][1], j) + cb * dy_elem<NT>(D[q][1][1], j);
                float dM[4] = {r0, r0 + r1, r0 - r1, -r1};
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[v], dM[v], acc[v][j], 0, 0, 0);
            }
        }
    };

    // one wave per SIMD by design (amdgpu_waves_per_eu(1, 1)): the scheduler is free to use the whole register file
    // and interleaves the next stage's loads and address arithmetic with the MFMAs of the current one
    const int nstages = (g.tps / 2) / SPS;
    load_stage(XA, DA);
    int sidx = 0;
    for (; sidx + 2 <= nstages; sidx += 2) {
        load_stage(XB, DB);
        compute(XA, DA);
        load_stage(XA, DA);                    // past the range: every tile invalid -> zeros
        compute(XB, DB);
    }
    if (sidx < nstages) compute(XA, DA);

    // ---- dg = G^T dU G.  nu-contraction lane-local: T[b] = sum_nu dU[xi][nu] G[nu][b]; xi-contraction through LDS,
    // one filter column b per pass (32 KB of LDS instead of 96) --------------------------------------------------------
    constexpr int BNW = 32 * NT;
    float *T = smem;                           // [xi 4][row 32][col BNW]
    float *slab = g.ws + (size_t)blockIdx.x * 9 * g.Cin * g.Cout;
    const size_t tapstride = (size_t)g.Cin * g.Cout;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        if (b) __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float a0 = acc[0][j][r], a1 = acc[1][j][r], a2 = acc[2][j][r], a3 = acc[3][j][r];
                float v = b == 0 ? a0 + 0.5f * (a1 + a2) : (b == 1 ? 0.5f * (a1 - a2) : 0.5f * (a1 + a2) + a3);
                T[(wave * 32 + row) * BNW + NT * lr + j] = v;
            }
        __syncthreads();
        for (int e = t; e < 32 * BNW; e += 256) {
            int col = e % BNW, row = e / BNW;
            float T0 = T[(0 * 32 + row) * BNW + col], T1 = T[(1 * 32 + row) * BNW + col];
            float T2 = T[(2 * 32 + row) * BNW + col], T3 = T[(3 * 32 + row) * BNW + col];
            float h = 0.5f * (T1 + T2);
            size_t o = (size_t)(ci0 + row) * g.Cout + co0 + col;
            slab[(0 * 3 + b) * tapstride + o] = T0 + h;
            slab[(1 * 3 + b) * tapstride + o] = 0.5f * (T1 - T2);
            slab[(2 * 3 + b) * tapstride + o] = h + T3;
        }
    }
}

// The same computation with the operands STAGED THROUGH LDS.  In-kernel stamps of the register-direct kernel above: the loop
// runs at 2.7x its MFMA issue time -- per 8 MFMAs a wave issues 8 + 4 load instructions of 4 (8) bytes per lane, and the four
// waves of a workgroup load overlapping pixels: the texture path, not the matrix pipe, sets the pace.  Here a workgroup walks
// STRIPS of G consecutive tiles of one tile row: the 4 x (2G + 2) input pixels and the 2 x 2G output-gradient pixels of the
// strip are fetched once with 16-byte loads (next strip in flight in registers under the current one's MFMAs), parked as
// Xs[row][pixel][32 channels] / Ds[row][pixel][32 NT channels], and every operand read is a conflict-free ds_read_b32 /
// ds_read_b64 of consecutive channels.  Same transforms, accumulators, slab layout and fixed-order reduction.
struct WinoWL {
    const float *x, *dy;
    float *ws;
    const float *icoef;         // input prologue table of x ([scale | shift | gate], icoef_ld apart; see yh_prologue) or null
    int icoef_ld;
    int ldx, lddy;
    int B, H, W, Cin, Cout;
    int TW, TH, G, spr, nstrips, sps;   // G tiles per strip (even), strips per tile row, total strips, strips per split
};

template <int NT, bool ACT>
__global__ __launch_bounds__(256, 2) void wino_wgrad_lds_kernel(const WinoWL g) {
    constexpr int GMAX = 16, PWMAX = 2 * GMAX + 2;
    constexpr int XPIECES_MAX = 4 * PWMAX * 8, DPIECES_MAX = 2 * 2 * GMAX * 8 * NT;
    constexpr int NXP = (XPIECES_MAX + 255) / 256, NDP = (DPIECES_MAX + 255) / 256;
    constexpr int BNW = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // Xs | Ds during the loop, T[4][32][BNW] in the epilogue
    float *Xs = smem, *Ds = smem + 4 * PWMAX * 32;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * BNW;
    const int G = g.G, PW = 2 * G + 2, DW = 2 * G;

    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg = wave == 1 ? 1.f : -1.f;
    const float ca = wave == 3 ? 0.f : 1.f;                    // row xi of A: (1,0) (1,1) (1,-1) (0,-1)
    const float cb = wave == 0 ? 0.f : (wave == 1 ? 1.f : -1.f);

    // staging plan (fixed for the kernel): piece k of this thread -> (row, pixel, channel quad); -1 = none
    int xrow[NXP], xpix[NXP], xq[NXP], drow[NDP], dpix[NDP], dq[NDP];
#pragma unroll
    for (int k = 0; k < NXP; ++k) {
        const int i = t + 256 * k, q = i & 7, pp = i >> 3;
        xq[k] = q; xpix[k] = pp % PW; xrow[k] = pp / PW;
        if (xrow[k] >= 4) xrow[k] = -1;
    }
#pragma unroll
    for (int k = 0; k < NDP; ++k) {
        const int i = t + 256 * k, q = i % (8 * NT), pp = i / (8 * NT);
        dq[k] = q; dpix[k] = pp % DW; drow[k] = pp / DW;
        if (drow[k] >= 2) drow[k] = -1;
    }
    // element offsets of the pieces from the strip's origin pixel (fixed for the kernel: one 64-bit base per strip, then an add
    // and two compares per piece -- the full address arithmetic per piece cost ~40 % of a strip's MFMA time)
    int xoff[NXP], doff[NDP];
#pragma unroll
    for (int k = 0; k < NXP; ++k) xoff[k] = (xrow[k] * g.W + xpix[k]) * g.ldx + 4 * xq[k];
#pragma unroll
    for (int k = 0; k < NDP; ++k) doff[k] = (drow[k] * g.W + dpix[k]) * g.lddy + 4 * dq[k];
    f32x4 rx[NXP], rd[NDP];
    // input prologue: every x piece of this thread covers the same channel quad (256 % 8 == 0), so its coefficients are loaded once;
    // padding must stay zero AFTER the activation: one validity bit per piece, refreshed by fetch
    f32x4 psc, psh, pgt;
    unsigned xok = 0;
    if constexpr (ACT) {
        const int ch = ci0 + 4 * (t & 7);
        psc = *(const f32x4 *)(g.icoef + ch); psh = *(const f32x4 *)(g.icoef + g.icoef_ld + ch); pgt = *(const f32x4 *)(g.icoef + 2 * g.icoef_ld + ch);
    }
    auto fetch = [&](int sid) {
        const int sx = sid % g.spr, rest = sid / g.spr;
        const int ty = rest % g.TH, b = rest / g.TH;
        const int ix0 = 2 * sx * G - 1, iy0 = 2 * ty - 1;
        const float *xbase = g.x + ((ptrdiff_t)(b * g.H + iy0) * g.W + ix0) * g.ldx + ci0;
        const float *dbase = g.dy + ((ptrdiff_t)(b * g.H + 2 * ty) * g.W + 2 * sx * G) * g.lddy + co0;
        const int wleft = g.W - 2 * sx * G;                // dY pixels of this strip that exist
        xok = 0;
#pragma unroll
        for (int k = 0; k < NXP; ++k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (xrow[k] >= 0 && (unsigned)(iy0 + xrow[k]) < (unsigned)g.H && (unsigned)(ix0 + xpix[k]) < (unsigned)g.W) {
                v = *(const f32x4 *)(xbase + xoff[k]);
                xok |= 1u << k;
            }
            rx[k] = v;
        }
#pragma unroll
        for (int k = 0; k < NDP; ++k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (drow[k] >= 0 && dpix[k] < wleft) v = *(const f32x4 *)(dbase + doff[k]);
            rd[k] = v;
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int k = 0; k < NXP; ++k)
            if (xrow[k] >= 0) {
                f32x4 v = rx[k];
                if constexpr (ACT) {
                    if (xok >> k & 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = yh_prologue(v[e], psc[e], psh[e], pgt[e]);
                    }
                }
                *(f32x4 *)(Xs + (xrow[k] * PWMAX + xpix[k]) * 32 + 4 * xq[k]) = v;
            }
#pragma unroll
        for (int k = 0; k < NDP; ++k)
            if (drow[k] >= 0) *(f32x4 *)(Ds + (drow[k] * 2 * GMAX + dpix[k]) * BNW + 4 * dq[k]) = rd[k];
    };

    f32x16 acc[4][NT];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[v][j][r] = 0.f;

    const int s_begin = blockIdx.x * g.sps;
    int s_end = s_begin + g.sps;
    if (s_end > g.nstrips) s_end = g.nstrips;
    // operand addresses of this lane inside a strip: tile 2 s + lh, channel lr (x) / NT lr (dy)
    const float *xa = Xs + (ra * PWMAX + 2 * lh) * 32 + lr, *xb = Xs + (rb * PWMAX + 2 * lh) * 32 + lr;
    const float *d0 = Ds + (2 * lh) * BNW + NT * lr, *d1 = d0 + 2 * GMAX * BNW;
    if (s_begin < s_end) fetch(s_begin);
    for (int sid = s_begin; sid < s_end; ++sid) {
        __syncthreads();                                   // the previous strip's operands are consumed
        park();
        __syncthreads();
        if (sid + 1 < s_end) fetch(sid + 1);               // in flight under the MFMAs below
        for (int ks = 0; ks < G / 2; ++ks) {               // one k-step = the tile pair (2 ks, 2 ks + 1)
            const int po = 4 * ks * 32;                    // 4 pixels per tile pair
            float tt[4], V[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) tt[cc] = xa[po + cc * 32] + sg * xb[po + cc * 32];
            V[0] = tt[0] - tt[2];
            V[1] = tt[1] + tt[2];
            V[2] = tt[2] - tt[1];
            V[3] = tt[1] - tt[3];
            const int dofs = 4 * ks * BNW;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float r0 = ca * d0[dofs + j] + cb * d1[dofs + j];
                const float r1 = ca * d0[dofs + BNW + j] + cb * d1[dofs + BNW + j];
                const float dM[4] = {r0, r0 + r1, r0 - r1, -r1};
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[v], dM[v], acc[v][j], 0, 0, 0);
            }
        }
    }

    // ---- dg = G^T dU G (as in wino_wgrad_kernel): nu-contraction lane-local, xi-contraction through LDS, one filter column per pass
    __syncthreads();
    float *T = smem;                           // [xi 4][row 32][col BNW]
    float *slab = g.ws + (size_t)blockIdx.x * 9 * g.Cin * g.Cout;
    const size_t tapstride = (size_t)g.Cin * g.Cout;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        if (b) __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float a0 = acc[0][j][r], a1 = acc[1][j][r], a2 = acc[2][j][r], a3 = acc[3][j][r];
                float v = b == 0 ? a0 + 0.5f * (a1 + a2) : (b == 1 ? 0.5f * (a1 - a2) : 0.5f * (a1 + a2) + a3);
                T[(wave * 32 + row) * BNW + NT * lr + j] = v;
            }
        __syncthreads();
        for (int e = t; e < 32 * BNW; e += 256) {
            int col = e % BNW, row = e / BNW;
            float T0 = T[(0 * 32 + row) * BNW + col], T1 = T[(1 * 32 + row) * BNW + col];
            float T2 = T[(2 * 32 + row) * BNW + col], T3 = T[(3 * 32 + row) * BNW + col];
            float h = 0.5f * (T1 + T2);
            size_t o = (size_t)(ci0 + row) * g.Cout + co0 + col;
            slab[(0 * 3 + b) * tapstride + o] = T0 + h;
            slab[(1 * 3 + b) * tapstride + o] = 0.5f * (T1 - T2);
            slab[(2 * 3 + b) * tapstride + o] = h + T3;
        }
    }
}

// ws [nsplit][9][Cin][Cout] -> dw OIHW: 16 split-lanes each add every 16th slab, lanes combined in order (the
// summation order is fixed).
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dw, int nsplit,
                                                                int Cin, int Cout) {
    __shared__ float red[16][17];
    const int n = 9 * Cin * Cout;
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
        int co = i % Cout, q = i / Cout;
        int ci = q % Cin, tap = q / Cin;
        dw[((size_t)co * Cin + ci) * 9 + tap] = tot;
    }
}

void set_magic(unsigned d, unsigned &magic, int &shift) {
    int l = 0;
    while ((1u << l) < d) ++l;
    magic = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
    shift = l - 1;
}

// ---- LDS-staged kernel: block geometry ------------------------------------------------------------------------------------
// R x C tiles per workgroup (R C <= 32) over B*TH linear tile rows x TW tile columns: the fewest workgroups (= the fewest MFMA
// tile slots), then the smallest patch.  PCh (half-row stride of the patch planes, in 16-byte slots) is searched so that the 16
// lanes ds_read_b128 serves per LDS cycle ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: MI355X_MICROARCH.md, LDS) hit 16
// different slots of the 256-byte bank row; the plane stride is padded to 2 mod 8 slots, which makes the staging ds_write_b128
// (8 consecutive lanes = 4 consecutive pixels x 2 k-quads) conflict-free as well.
struct WinoGeom {
    int R, C, PCh, PR, plane, bufsz, nrb, ncb;
};
static int wino_read_conflicts(int R, int C, int PCh) {
    static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                   {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int worst = 0;
    for (int gi = 0; gi < 2; ++gi) {
        int cnt[16] = {0};
        bool zero_seen = false;
        for (int k = 0; k < 16; ++k) {
            int l = grp[gi][k];
            if (l >= R * C) l = 0;                  // empty slots read tile 0 (identical addresses broadcast)
            if (l == 0) { if (zero_seen) continue; zero_seen = true; }
            const int r = l / C, c = l - r * C;
            ++cnt[(2 * r * PCh + c) & 15];
        }
        for (int k = 0; k < 16; ++k) worst = cnt[k] > worst ? cnt[k] : worst;
    }
    return worst;
}
static bool wino_lds_geom(int B, int TH, int TW, WinoGeom &o) {
    long best_cost = -1;
    const int GT = B * TH;
    for (int maxp = 512; maxp <= 768 && best_cost < 0; maxp += 256)     // two pieces per thread if any block shape allows it
    for (int C = 1; C <= 32 && C <= TW; ++C) {
        int R = 32 / C;
        if (R > GT) R = GT;
        const int maxcross = R > 1 ? (R - 1 + TH - 1) / TH : 0;
        const int PR = 2 * R + 2 + 2 * maxcross, PC = 2 * C + 2;
        if (PR * PC * 2 > maxp) continue;           // 16-byte pieces per thread: two, three at most
        const int nrb = cdiv(GT, R), ncb = cdiv(TW, C);
        const long cost = (long)nrb * ncb * 4096 + PR * PC;
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            o.R = R; o.C = C; o.PR = PR; o.nrb = nrb; o.ncb = ncb;
        }
    }
    if (best_cost < 0) return false;
    int bestp = o.C + 1, bestw = 1 << 30;
    for (int p = o.C + 1; p < o.C + 1 + 16; ++p) {
        const int w = wino_read_conflicts(o.R, o.C, p);
        if (w < bestw) { bestw = w; bestp = p; }
    }
    o.PCh = bestp;
    o.plane = o.PR * o.PCh;
    while ((o.plane & 7) != 2) ++o.plane;
    o.bufsz = 4 * o.plane * 4;
    return true;
}

template <int NT, bool ACT, int NP>
int launch_lds_np(Wino &g, const WinoGeom &gm, hipStream_t st) {
    constexpr int BNW = 32 * NT;
    g.ncol = cdiv(g.N, BNW);
    const int epi = 4 * 2 * TPB * BNW + 256 * 2;
    g.toff_ofs = 2 * gm.bufsz > epi ? 2 * gm.bufsz : epi;
    const size_t smem = (size_t)(g.toff_ofs + TPB) * sizeof(float);
    YH_REQUIRE(smem <= 80 * 1024, "conv_wino (LDS-staged): patch buffers exceed the budget of two workgroups per CU");
    if (int rc = yh_ensure_dyn_smem((const void *)wino_lds_kernel<NT, ACT, NP>, smem)) return rc;
#ifdef YH_WINO_STAMPS
    static unsigned long long *dbgbuf = nullptr;
    const int nwg_dbg = gm.nrb * gm.ncb * g.ncol;
    if (!dbgbuf) (void)hipMalloc((void **)&dbgbuf, (size_t)1 << 24);
    g.dbg = getenv("YH_WINO_DBG") && (size_t)nwg_dbg * 48 <= ((size_t)1 << 24) ? dbgbuf : nullptr;
#endif
    {
        const unsigned dv[6] = {(unsigned)g.ncol, (unsigned)g.ncb, (unsigned)g.TH, (unsigned)g.C, (unsigned)(2 * g.C + 2), (unsigned)(2 * g.TH + 2)};
        for (int i = 0; i < 6; ++i) set_magic(dv[i], g.dm[i], g.ds[i]);
    }
    hipLaunchKernelGGL((wino_lds_kernel<NT, ACT, NP>), dim3(gm.nrb * gm.ncb * g.ncol), dim3(256), smem, st, g);
    YH_CHECK_LAUNCH("wino_lds");
#ifdef YH_WINO_STAMPS
    if (g.dbg) {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h((size_t)nwg_dbg * 6);
        (void)hipMemcpy(h.data(), g.dbg, h.size() * 8, hipMemcpyDeviceToHost);
        double a = 0, b = 0, c = 0, rt = 0; unsigned long long lo = ~0ull, hi = 0;
        for (int i = 0; i < nwg_dbg; ++i) {
            a += (double)(h[6 * i + 1] - h[6 * i]); b += (double)(h[6 * i + 2] - h[6 * i + 1]); c += (double)(h[6 * i + 3] - h[6 * i + 2]);
            rt += (double)(h[6 * i + 5] - h[6 * i + 4]);
            if (h[6 * i + 4] < lo) lo = h[6 * i + 4];
            if (h[6 * i + 5] > hi) hi = h[6 * i + 5];
        }
        const double mf = (double)(g.K / 8) * 16 * NT * 64;
        fprintf(stderr, "[wino lds stamps] NT %d ACT %d K %d N %d R %d C %d wgs %d: setup %.0f, loop %.0f (MFMA issue floor %.0f), epilogue %.0f cycles per workgroup = %.2f us (clock %.2f GHz); span %.1f us\n",
                NT, (int)ACT, g.K, g.N, g.R, g.C, nwg_dbg, a / nwg_dbg, b / nwg_dbg, mf, c / nwg_dbg, rt / nwg_dbg / 100.0, (a + b + c) / rt * 0.1, (double)(hi - lo) / 100.0);
    }
#endif
    return 0;
}
template <int NT, bool ACT>
int launch_lds_nt(Wino &g, const WinoGeom &gm, hipStream_t st) {
    return gm.PR * (2 * gm.C + 2) * 2 <= 512 ? launch_lds_np<NT, ACT, 2>(g, gm, st) : launch_lds_np<NT, ACT, 3>(g, gm, st);
}

int launch_wino_lds(Wino &g, hipStream_t st) {
    YH_REQUIRE(g.H % 2 == 0 && g.W % 2 == 0, "conv_wino: H and W must be even");
    YH_REQUIRE(g.K % (2 * KC) == 0 && g.ldi % 4 == 0 && (((uintptr_t)g.in | (uintptr_t)g.U) & 15) == 0 && g.ldu >= g.N,
               "conv_wino: K must be a multiple of 16, buffers 16-byte addressable");
    YH_REQUIRE(!g.icoef || ((((uintptr_t)g.icoef) & 15) == 0 && g.icoef_ld % 4 == 0 && g.icoef_ld >= g.K),
               "conv_wino: the input prologue table must be 16-byte aligned with a stride that is a multiple of 4");
    YH_REQUIRE((int64_t)g.B * g.H * g.W * g.ldi < (1ll << 31) && (int64_t)g.B * g.H * g.W * g.ldo * (g.up2 ? 4 : 1) < (1ll << 31),
               "conv_wino: tensors exceed 32-bit element offsets");
    g.TW = g.W / 2; g.TH = g.H / 2; g.TPI = g.TH * g.TW; g.ntiles = g.B * g.TPI;
    WinoGeom gm{};
    YH_REQUIRE(wino_lds_geom(g.B, g.TH, g.TW, gm), "conv_wino: no block geometry for this shape");
    g.R = gm.R; g.C = gm.C; g.ncb = gm.ncb; g.PCh = gm.PCh; g.plane = gm.plane; g.bufsz = gm.bufsz;
    if (g.icoef) return g.N <= 32 ? launch_lds_nt<1, true>(g, gm, st) : launch_lds_nt<2, true>(g, gm, st);
    return g.N <= 32 ? launch_lds_nt<1, false>(g, gm, st) : launch_lds_nt<2, false>(g, gm, st);
}

}  // namespace

namespace {
// strips of the LDS-staged kernel: G tiles (even, <= 16) per strip with the least padding of the tile row
int wgrad_lds_plan(WinoWL &g, int &nsplit, int &NT, int B, int H, int W, int Cin, int Cout) {
    g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout;
    g.TW = W / 2; g.TH = H / 2;
    int best = 16, bestpad = 1 << 30;
    for (int G = 16; G >= 8; G -= 2) {
        const int pad = cdiv(g.TW, G) * G;
        if (pad < bestpad) { bestpad = pad; best = G; }
    }
    if (g.TW < 8) best = (g.TW + 1) & ~1;
    g.G = best;
    g.spr = cdiv(g.TW, g.G);
    g.nstrips = B * g.TH * g.spr;
    NT = Cout % 64 == 0 ? 2 : 1;
    const int pairs = (Cin / 32) * (Cout / (32 * NT));
    constexpr int target = 512;
    nsplit = target / pairs;
    if (nsplit < 1) nsplit = 1;
    g.sps = cdiv(g.nstrips, nsplit);
    if (g.sps < 4) g.sps = 4;
    nsplit = cdiv(g.nstrips, g.sps);
    return 0;
}
inline bool wino_wgrad_use_lds() {
    constexpr bool on = true;
    return on;
}

int wgrad_plan(WinoW &g, int &nsplit, int &NT, int B, int H, int W, int Cin, int Cout) {
    YH_REQUIRE(H % 2 == 0 && W % 2 == 0 && Cin % 32 == 0 && Cout % 32 == 0, "conv_wino_bwd_weight: even H, W and channels % 32 == 0");
    g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout;
    g.TW = W / 2; g.TPI = (H / 2) * g.TW; g.ntiles = B * g.TPI;
    set_magic((unsigned)g.TW, g.tw_magic, g.tw_shift);
    set_magic((unsigned)g.TPI, g.tpi_magic, g.tpi_shift);
    NT = Cout % 64 == 0 ? 2 : 1;
    int pairs = (Cin / 32) * (Cout / (32 * NT));
    constexpr int target = 512;
    nsplit = target / pairs;
    if (nsplit < 1) nsplit = 1;
    int tps = cdiv(cdiv(g.ntiles, nsplit), 8) * 8;
    if (tps < 64) tps = 64;
    nsplit = cdiv(g.ntiles, tps);
    g.tps = tps;
    return 0;
}
}  // namespace

extern "C" int64_t yh_conv_wino_bwd_weight_ws(int B, int H, int W, int Cin, int Cout) {
    WinoW g{};
    int nsplit, NT;
    if (wgrad_plan(g, nsplit, NT, B, H, W, Cin, Cout)) return -1;
    WinoWL gl{};
    int nsplit2 = 0, NT2;
    wgrad_lds_plan(gl, nsplit2, NT2, B, H, W, Cin, Cout);
    return (int64_t)(nsplit > nsplit2 ? nsplit : nsplit2) * 9 * Cin * Cout;       // either kernel may run (YH_WINO_WGRAD_LDS)
}

static int wino_bwd_weight_impl(const float *x, int ldx, const float *icoef, int icoef_ld, const float *dy, int lddy, float *dw, float *ws,
                                int64_t ws_floats, int B, int H, int W, int Cin, int Cout, void *stream);
extern "C" int yh_conv_wino_bwd_weight(const float *x, int ldx, const float *dy, int lddy, float *dw, float *ws, int64_t ws_floats,
                                       int B, int H, int W, int Cin, int Cout, void *stream) {
    return wino_bwd_weight_impl(x, ldx, nullptr, 0, dy, lddy, dw, ws, ws_floats, B, H, W, Cin, Cout, stream);
}
extern "C" int yh_conv_wino_bwd_weight_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *dy, int lddy, float *dw,
                                           float *ws, int64_t ws_floats, int B, int H, int W, int Cin, int Cout, void *stream) {
    return wino_bwd_weight_impl(x, ldx, icoef, icoef_ld, dy, lddy, dw, ws, ws_floats, B, H, W, Cin, Cout, stream);
}
static int wino_bwd_weight_impl(const float *x, int ldx, const float *icoef, int icoef_ld, const float *dy, int lddy, float *dw, float *ws,
                                int64_t ws_floats, int B, int H, int W, int Cin, int Cout, void *stream) {
    YH_REQUIRE(x && dy && dw && ws && ldx >= Cin && lddy >= Cout, "conv_wino_bwd_weight: bad argument");
    YH_REQUIRE(!icoef || ((((uintptr_t)icoef) & 15) == 0 && icoef_ld % 4 == 0 && icoef_ld >= Cin),
               "conv_wino_bwd_weight: the input prologue table must be 16-byte aligned with a stride that is a multiple of 4");
    WinoW g{};
    int nsplit, NT;
    int rc = wgrad_plan(g, nsplit, NT, B, H, W, Cin, Cout);
    if (rc) return rc;
    YH_REQUIRE(ws_floats >= (int64_t)nsplit * 9 * Cin * Cout, "conv_wino_bwd_weight: workspace too small");
    YH_REQUIRE((int64_t)B * H * W * (ldx > lddy ? ldx : lddy) < (1ll << 31), "conv_wino_bwd_weight: tensor exceeds 32-bit element offsets");
    YH_REQUIRE(NT == 1 || ((((uintptr_t)dy) & 7) == 0 && lddy % 2 == 0), "conv_wino_bwd_weight: dy must be 8-byte addressable");
    g.x = x; g.dy = dy; g.ws = ws; g.ldx = ldx; g.lddy = lddy;
    const int64_t npix = (int64_t)B * H * W;
    YH_REQUIRE(((npix - 1) * ldx + Cin) * 4 < (1ll << 31) && ((npix - 1) * lddy + Cout) * 4 < (1ll << 31) && W >= 4,
               "conv_wino_bwd_weight: views must span less than 2 GiB");
    g.x_bytes = (unsigned)(((npix - 1) * ldx + Cin) * 4);
    g.dy_bytes = (unsigned)(((npix - 1) * lddy + Cout) * 4);
    hipStream_t st = (hipStream_t)stream;
    if (wino_wgrad_use_lds() && ldx % 4 == 0 && lddy % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)dy)) & 15) == 0) {
        WinoWL gl{};
        int ns2, NT2;
        wgrad_lds_plan(gl, ns2, NT2, B, H, W, Cin, Cout);
        YH_REQUIRE(ws_floats >= (int64_t)ns2 * 9 * Cin * Cout, "conv_wino_bwd_weight: workspace too small");
        gl.x = x; gl.dy = dy; gl.ws = ws; gl.ldx = ldx; gl.lddy = lddy; gl.icoef = icoef; gl.icoef_ld = icoef_ld;
        dim3 grid2(ns2, Cin / 32, Cout / (32 * NT2));
        const size_t stage = (size_t)(4 * 34 * 32 + 2 * 32 * 32 * NT2) * sizeof(float), epi = (size_t)4 * 32 * 32 * NT2 * sizeof(float);
        const size_t smem2 = stage > epi ? stage : epi;
        if (icoef) {
            if (NT2 == 2) hipLaunchKernelGGL((wino_wgrad_lds_kernel<2, true>), grid2, dim3(256), smem2, st, gl);
            else hipLaunchKernelGGL((wino_wgrad_lds_kernel<1, true>), grid2, dim3(256), smem2, st, gl);
        } else {
            if (NT2 == 2) hipLaunchKernelGGL((wino_wgrad_lds_kernel<2, false>), grid2, dim3(256), smem2, st, gl);
            else hipLaunchKernelGGL((wino_wgrad_lds_kernel<1, false>), grid2, dim3(256), smem2, st, gl);
        }
        YH_CHECK_LAUNCH("wino_wgrad_lds");
        const int n2 = 9 * Cin * Cout;
        hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(cdiv(n2, 16)), dim3(256), 0, st, ws, dw, ns2, Cin, Cout);
        YH_CHECK_LAUNCH("wino_wgrad_reduce");
        return 0;
    }
    YH_REQUIRE(!icoef, "conv_wino_bwd_weight: the input prologue needs 16-byte addressable views (LDS-staged kernel)");
    dim3 grid(nsplit, Cin / 32, Cout / (32 * NT));
    // two waves per SIMD hide each other's address arithmetic for the 64-column variant (0.222 -> 0.187 ms on 64->64 @80^2);
    // the 32-column variant spills at that register budget and stays at one
    const size_t smem = (size_t)4 * 32 * 32 * NT * sizeof(float);
    if (NT == 2) hipLaunchKernelGGL((wino_wgrad_kernel<2, 2>), grid, dim3(256), smem, st, g);
    else hipLaunchKernelGGL((wino_wgrad_kernel<1, 1>), grid, dim3(256), smem, st, g);
    YH_CHECK_LAUNCH("wino_wgrad");
    int n = 9 * Cin * Cout;
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(cdiv(n, 16)), dim3(256), 0, st, ws, dw, nsplit, Cin, Cout);
    YH_CHECK_LAUNCH("wino_wgrad_reduce");
    return 0;
}


extern "C" int yh_wino_weights(const float *oihw, float *U, int Cout, int Cin, int ldu, int backward, void *stream) {
    YH_REQUIRE(oihw && U && Cout > 0 && Cin > 0 && ldu >= (backward ? Cin : Cout) && (backward ? Cout : Cin) % 4 == 0, "wino_weights: bad argument");
    int total = (backward ? Cout : Cin) * ldu;
    int blocks = cdiv(total, 256);
    WinoWDesc d{oihw, U, Cout, Cin, ldu, backward};
    hipLaunchKernelGGL(wino_weights_kernel, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, (hipStream_t)stream, d);
    YH_CHECK_LAUNCH("wino_weights");
    return 0;
}

extern "C" int yh_wino_weights_multi(const void *table, int n, void *stream) {
    YH_REQUIRE(table && n > 0, "wino_weights_multi: bad argument");
    static_assert(sizeof(WinoWDesc) == 32, "descriptor layout is part of the ABI");
    hipLaunchKernelGGL(wino_weights_multi_kernel, dim3(256, n), dim3(256), 0, (hipStream_t)stream, (const WinoWDesc *)table);
    YH_CHECK_LAUNCH("wino_weights_multi");
    return 0;
}

extern "C" int yh_conv_wino_fwd(const float *x, int ldx, const float *U, int ldu, const float *bias, float *y, int ldy,
                                float *bn_partials, int B, int H, int W, int Cin, int Cout, void *stream) {
    YH_REQUIRE(x && U && y && B > 0 && H > 0 && W > 0 && ldx >= Cin && ldy >= Cout, "conv_wino_fwd: bad argument");
    Wino g{};
    g.in = x; g.U = U; g.bias = bias; g.out = y; g.stats = bn_partials;
    g.ldi = ldx; g.ldu = ldu; g.ldo = ldy; g.B = B; g.H = H; g.W = W; g.K = Cin; g.N = Cout; g.accumulate = 0;
    return launch_wino_lds(g, (hipStream_t)stream);
}

extern "C" int yh_conv_wino_fwd_fused(const float *x, int ldx, const float *U, int ldu, const float *bias, const float *res, int ldr,
                                      float *y, int ldy, int B, int H, int W, int Cin, int Cout, int act_silu, int upsample,
                                      void *stream) {
    YH_REQUIRE(x && U && y && B > 0 && H > 0 && W > 0 && ldx >= Cin && ldy >= Cout && (!res || ldr >= Cout), "conv_wino_fwd_fused: bad argument");
    Wino g{};
    g.in = x; g.U = U; g.bias = bias; g.out = y; g.stats = nullptr;
    g.ldi = ldx; g.ldu = ldu; g.ldo = ldy; g.B = B; g.H = H; g.W = W; g.K = Cin; g.N = Cout; g.accumulate = 0;
    g.res = res; g.ldr = ldr; g.act = act_silu ? 1 : 0; g.up2 = upsample ? 1 : 0;
    return launch_wino_lds(g, (hipStream_t)stream);
}

extern "C" int yh_conv_wino_bwd_data(const float *dy, int lddy, const float *Ub, int ldub, float *dx, int lddx, int B, int H,
                                     int W, int Cin, int Cout, int accumulate, void *stream) {
    YH_REQUIRE(dy && Ub && dx && B > 0 && H > 0 && W > 0 && lddy >= Cout && lddx >= Cin, "conv_wino_bwd_data: bad argument");
    Wino g{};
    g.in = dy; g.U = Ub; g.bias = nullptr; g.out = dx; g.stats = nullptr;
    g.ldi = lddy; g.ldu = ldub; g.ldo = lddx; g.B = B; g.H = H; g.W = W; g.K = Cout; g.N = Cin; g.accumulate = accumulate;
    return launch_wino_lds(g, (hipStream_t)stream);
}

extern "C" int yh_conv_wino_blocks(int B, int H, int W) {
    WinoGeom gm{};
    if (H <= 0 || W <= 0 || B <= 0 || (H | W) & 1 || !wino_lds_geom(B, H / 2, W / 2, gm)) return -1;
    return gm.nrb * gm.ncb;
}

extern "C" int yh_conv_wino_fwd_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *U, int ldu,
                                    const float *bias, float *y, int ldy, float *bn_partials, int B, int H, int W, int Cin, int Cout,
                                    void *stream) {
    YH_REQUIRE(x && U && y && B > 0 && H > 0 && W > 0 && ldx >= Cin && ldy >= Cout, "conv_wino_fwd_act: bad argument");
    Wino g{};
    g.in = x; g.U = U; g.bias = bias; g.out = y; g.stats = bn_partials; g.icoef = icoef; g.icoef_ld = icoef_ld;
    g.ldi = ldx; g.ldu = ldu; g.ldo = ldy; g.B = B; g.H = H; g.W = W; g.K = Cin; g.N = Cout; g.accumulate = 0;
    return launch_wino_lds(g, (hipStream_t)stream);
}

