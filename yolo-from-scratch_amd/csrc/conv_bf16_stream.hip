// bf16 convolutions as FLAT PIXEL STREAMS (gfx950): the stride-1 layers (k = 1 and k = 3), which carry most of the bf16
// path's bytes and are bound by HBM, not by the matrix pipe.
//
// The padded image batch is ONE 1-D sequence: position f = (b * VH + vy) * XW + vx with XW = W + 2 pad, VH = H + 2 pad
// (pad = k / 2; padding positions hold zeros).  A 3x3 tap (kh, kw) is then a constant displacement (kh - 1) * XW + (kw - 1)
// of that sequence -- no per-row segments, no per-segment halo reloads, no bounds arithmetic in the multiply loop:
//
//   dW[tap][ci][co] = sum_f  x[f + off(tap)][ci] * dy[f][co]          (dy = 0 at padding positions)
//
// A workgroup owns a (32 NI input channels) x (32 NJ output channels) x all-taps slab and a contiguous range of 64-position
// blocks of the stream.  Both operands travel HBM -> registers -> LDS rings ONCE (the x ring keeps the 2 XW + 2 positions of
// halo that the nine taps share); LDS images are planes of 32 channels with a 64-byte pixel stride, which makes the
// transposing fragment reads (ds_read_b64_tr_b16: the reduction index -- pixels -- is the slow axis of NHWC) conflict-free
// without padding.  Loads run PD = 3 blocks ahead in three register sets; one barrier per block; two workgroups per CU.
//
// Slabs are written in the layout of bf16_wgrad_reduce_kernel (conv_bf16.hip), whose fixed-order sum gives the OIHW fp32
// gradient: bitwise reproducible.
#include "bf16_common.h"
#include <type_traits>
#ifdef YH_FS_STAMPS
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#endif
#include <stdlib.h>

namespace {

struct WgS {
    const bf16 *x, *dy;
    float *ws;
    int ldx, lddy, Cin, Cout, cout8;
    int H, W, XW, VH, pad, Mflat;
    unsigned xw_magic, vh_magic;
    int xw_shift, vh_shift;
    int n_ci_tiles, ntiles, nsplit, nblk_total, blk_per_split;
    int HB;          // halo of the nine taps in 64-position blocks: ceil((XW + 1) / 64), 0 for k = 1
    int RX, RD;      // ring sizes in positions: (2 HB + 3) * 64 and 192
    int tapoff[9];
};

// block = BP stream positions per pipeline step, PD = blocks requested ahead (PD - 1 register sets in flight).  What bounds a
// workgroup's streaming rate is bytes in flight: (PD - 1) blocks against ~2 us of HBM latency.  3x3: BP 128, PD 4, one
// 512-thread workgroup per CU (72 KB in flight); 1x1: BP 64, PD 3, up to three 256-thread workgroups per CU.
// NT = threads per workgroup: the 3x3 form runs 8 waves on one set of rings (two waves per SIMD hide each other's LDS latency).
template <int KK> struct WgsCfg { static constexpr int BP = KK == 9 ? 128 : 64, PD = KK == 9 ? 4 : 3, NT = KK == 9 ? 512 : 256, WPE = 2; };

// flat position -> pixel index of the NHWC tensor; false for padding / out-of-range positions
template <int KK>
__device__ __forceinline__ bool wgs_pixel(const WgS &g, int f, int &pix) {
    if (KK == 1) {
        pix = f;
        return (unsigned)f < (unsigned)g.Mflat;
    }
    const bool in = (unsigned)f < (unsigned)g.Mflat;
    const int ff = in ? f : 0;
    const int q1 = yh_fast_div(ff, g.xw_magic, g.xw_shift), vx = ff - q1 * g.XW - g.pad;
    const int b = yh_fast_div(q1, g.vh_magic, g.vh_shift), vy = q1 - b * g.VH - g.pad;
    pix = (b * g.H + vy) * g.W + vx;
    return in && (unsigned)vx < (unsigned)g.W && (unsigned)vy < (unsigned)g.H;
}

__device__ __forceinline__ int wgs_mod(int v, int m) {        // v may be negative
    int r = v % m;
    return r < 0 ? r + m : r;
}

template <int KK, int NI, int NJ>
__global__ __launch_bounds__(WgsCfg<KK>::NT, WgsCfg<KK>::WPE) void bf16_wgrad_stream_kernel(const WgS g) {
    constexpr int BP = WgsCfg<KK>::BP, PD = WgsCfg<KK>::PD, NT = WgsCfg<KK>::NT;
    constexpr int WK = (NT / 64) / (NI * NJ);    // wave groups that split the 16-position steps of a block
    constexpr int PPT = NT / 4;                  // positions covered by one pass of the workgroup's threads
    constexpr int NPIX = BP / PPT;               // positions per thread and block (thread t: positions (t >> 2) + PPT i)
    static_assert(BP % PPT == 0 && (BP / 16) % WK == 0, "block shape");
    constexpr int TG = KK == 9 ? 3 : 1;          // taps per round of the end-of-kernel wave reduction
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);       // provably wave-uniform: scalar branches, not EXEC masks
    const int wi = wave_u % NI, wj = (wave_u / NI) % NJ, wk = wave_u / (NI * NJ);
    const int RXp = g.RX + 16;                   // + mirror of the first 16 positions: a 16-position read never wraps
    unsigned char *Xs = smem;                    // [NI][RXp][64 bytes]
    unsigned char *Ds = smem + NI * RXp * 64;    // [NJ][RD][64 bytes]
    // XCD-aware placement (workgroups b and b + 8 share an XCD and its L2): the ntiles workgroups that stream the SAME block
    // range (different channel tiles) get consecutive slots of one XCD, so the second reader of a line finds it in L2.
    // Bijective: whole groups of 8 splits are remapped, a trailing partial group keeps the plain order.
    int split, tile;
    {
        const int lin = blockIdx.x, per = 8 * g.ntiles, full = (g.nsplit >> 3) * per;
        if (lin < full) {
            const int grp = lin / per, r = lin - grp * per, xcd = r & 7, w = r >> 3;
            split = grp * 8 + xcd;
            tile = w;
        } else {
            const int r = lin - full, rem = g.nsplit & 7;
            split = (g.nsplit & ~7) + r % rem;
            tile = r / rem;
        }
    }
    const int ci0 = (tile % g.n_ci_tiles) * 32 * NI;
    const int co0 = (tile / g.n_ci_tiles) * 32 * NJ;
    const int j0 = split * g.blk_per_split;
    int j1 = j0 + g.blk_per_split;
    if (j1 > g.nblk_total) j1 = g.nblk_total;
    const int nX = g.RX / BP, nD = g.RD / BP;

    f32x16 acc[KK];
#pragma unroll
    for (int u = 0; u < KK; ++u)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[u][q] = 0.f;

    // ---- staging: thread = (positions (t >> 2) + 64 i of the block, 16-byte chunk ch of a 32-channel plane) ---------------
    const int pix = t >> 2, ch = t & 3;
    const unsigned char *xb = (const unsigned char *)g.x, *db = (const unsigned char *)g.dy;
    u32x4 rx[PD][NPIX * NI], rd[PD][NPIX * NJ];
    unsigned mk[PD];                             // bits 0..: x pieces valid, bits 16..: dy pieces valid
    // L(n): request x block n + HB + PD and dy block n + PD (stored PD - 1 iterations later).  Every load is issued
    // unconditionally (a clamped address, zeroed when it is stored) so that the compiler's vmcnt bookkeeping stays exact.
    auto issue = [&](int n, u32x4 (&ax)[NPIX * NI], u32x4 (&ad)[NPIX * NJ], unsigned &mko) __attribute__((always_inline)) {
        unsigned m = 0;
        const int bx = n + g.HB + PD, bd = n + PD;
        const bool bxok = bx >= j0 - g.HB && bx < j1 + g.HB, bdok = bd >= j0 && bd < j1;
#pragma unroll
        for (int i = 0; i < NPIX; ++i) {
            int px, pd;
            const bool okx = wgs_pixel<KK>(g, BP * bx + pix + PPT * i, px) && bxok;
            const bool okd = wgs_pixel<KK>(g, BP * bd + pix + PPT * i, pd) && bdok;
            const unsigned ox = okx ? (unsigned)(px * g.ldx + ci0 + 8 * ch) * 2u : 0u;
            const unsigned od = okd ? (unsigned)(pd * g.lddy + co0 + 8 * ch) * 2u : 0u;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const bool ok = okx && ci0 + 32 * j + 8 * ch < g.Cin;
                m |= (unsigned)ok << (i * NI + j);
                ax[i * NI + j] = *(const u32x4 *)(xb + (ok ? ox + 64u * j : 0u));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bool ok = okd && co0 + 32 * j + 8 * ch < g.cout8;
                m |= (unsigned)ok << (16 + i * NJ + j);
                ad[i * NJ + j] = *(const u32x4 *)(db + (ok ? od + 64u * j : 0u));
            }
        }
        mko = m;
    };
    // ring cursors of iteration n, advanced by one block per iteration (wave-uniform: scalar registers, no division in the loop)
    const int nfirst = j0 - 2 * g.HB - 1;
    int st_x = wgs_mod(nfirst + g.HB + 1, nX) * BP, st_d = wgs_mod(nfirst + 1, nD) * BP;     // where S(n) stores
    int cd = wgs_mod(nfirst, nD) * BP;                                                        // dy block n
    int cx[KK];                                                                               // x position BP n + off(tap)
#pragma unroll
    for (int u = 0; u < KK; ++u) cx[u] = wgs_mod(BP * nfirst + g.tapoff[u], g.RX);
    auto advance = [&]() __attribute__((always_inline)) {
        st_x += BP; if (st_x >= g.RX) st_x -= g.RX;
        st_d += BP; if (st_d >= g.RD) st_d -= g.RD;
        cd += BP; if (cd >= g.RD) cd -= g.RD;
#pragma unroll
        for (int u = 0; u < KK; ++u) { cx[u] += BP; if (cx[u] >= g.RX) cx[u] -= g.RX; }
    };
    // S(n): x block n + HB + 1 and dy block n + 1 go to their ring slots (zeros where the mask says so)
    auto store = [&](const u32x4 (&ax)[NPIX * NI], const u32x4 (&ad)[NPIX * NJ], unsigned mki) __attribute__((always_inline)) {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NPIX; ++i) {
            const int sx = st_x + pix + PPT * i, sd = st_d + pix + PPT * i;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const u32x4 v = (mki >> (i * NI + j)) & 1u ? ax[i * NI + j] : z;
                unsigned char *p = Xs + (j * RXp + sx) * 64 + ch * 16;
                *(u32x4 *)p = v;
                if (sx < 16) *(u32x4 *)(p + g.RX * 64) = v;
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                *(u32x4 *)(Ds + (j * g.RD + sd) * 64 + ch * 16) = (mki >> (16 + i * NJ + j)) & 1u ? ad[i * NJ + j] : z;
        }
    };

    // ---- fragments: transposing reads (lane = 16 gi + 4 q + p addresses position 8 (lane >> 5) + q [+ 4], channels
    // 16 (gi & 1) + 4 p .. + 3 of its wave's plane; it receives the MFMA row / column lane & 31, k = 8 (lane >> 5) + 0..7) -----
    const int gi = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3, lh = lane >> 5;
    const int lane_off = (8 * lh + qq) * 64 + (16 * (gi & 1) + 4 * pp) * 2;
    const unsigned char *a_lane = Xs + wi * RXp * 64 + lane_off;
    const unsigned char *b_lane = Ds + wj * g.RD * 64 + lane_off;
    typedef __attribute__((address_space(3))) bf16x4 *lds_b4;
    auto frag = [&](const unsigned char *p) __attribute__((always_inline)) {
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(p));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(p + 256));
        return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto a_ptr = [&](int u, int st) __attribute__((always_inline)) {
        int sl = cx[u] + 16 * st;
        if (sl >= g.RX) sl -= g.RX;
        return a_lane + sl * 64;
    };
    // this wave's steps of the block; the x fragment of tap u + 1 is requested before the MFMA of tap u
    auto compute = [&]() __attribute__((always_inline)) {
        for (int st = wk; st < BP / 16; st += WK) {
            const bf16x8 bv = frag(b_lane + (cd + 16 * st) * 64);
            bf16x8 av = frag(a_ptr(0, st));
#pragma unroll
            for (int u = 0; u < KK; ++u) {
                bf16x8 an = av;
                if (u + 1 < KK) an = frag(a_ptr(u + 1, st));
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[u], 0, 0, 0);
                av = an;
            }
        }
    };

    // ---- pipeline: iteration n stores what was requested PD - 1 iterations earlier, requests PD - 1 blocks ahead, multiplies
    // block n (from n = j0 on; the first 2 HB + 1 iterations only fill the x halo).  Unrolled PD times: static register sets. --
#pragma unroll
    for (int p = 0; p < PD - 1; ++p) issue(nfirst - (PD - 1) + p, rx[p], rd[p], mk[p]);
    for (int n = nfirst; n < j1; n += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            if (n + p < j1) {
                issue(n + p, rx[(p + PD - 1) % PD], rd[(p + PD - 1) % PD], mk[(p + PD - 1) % PD]);
                store(rx[p], rd[p], mk[p]);
                __syncthreads();
                if (n + p >= j0) compute();
                advance();
            }
        }
    }
    __syncthreads();

    // ---- the WK pixel groups of a slab are summed into the wk == 0 waves through LDS, TG taps per round, in a fixed order ----
    if (WK > 1) {
        float *red = (float *)smem;                 // [NI * NJ][TG][32][33]
        float *r = red + (size_t)((wj * NI + wi) * TG) * 32 * 33;
        const int lr = lane & 31;
#pragma unroll
        for (int u0 = 0; u0 < KK; u0 += TG) {
            for (int w = 1; w < WK; ++w) {
                if (wk == w) {
#pragma unroll
                    for (int u = 0; u < TG; ++u)
#pragma unroll
                        for (int q = 0; q < 16; ++q) r[(u * 32 + yh_mfma_row(q, lh)) * 33 + lr] = acc[u0 + u][q];
                }
                __syncthreads();
                if (wk == 0) {
#pragma unroll
                    for (int u = 0; u < TG; ++u)
#pragma unroll
                        for (int q = 0; q < 16; ++q) acc[u0 + u][q] += r[(u * 32 + yh_mfma_row(q, lh)) * 33 + lr];
                }
                __syncthreads();
            }
        }
        if (wk != 0) return;
    }
    // slab write [tap][ci][co] fp32 straight from the accumulators (a register row = 32 consecutive output channels)
    float *slab = g.ws + (size_t)split * KK * g.Cin * g.Cout;
    const int co = co0 + 32 * wj + (lane & 31);
#pragma unroll
    for (int u = 0; u < KK; ++u)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ci = ci0 + 32 * wi + yh_mfma_row(q, lh);
            if (ci < g.Cin && co < g.Cout) slab[((size_t)u * g.Cin + ci) * g.Cout + co] = acc[u][q];
        }
}

struct WgsPlan {
    WgS g;
    int NI, NJ, KK, nsplit, ntiles;
    size_t smem;
};

void plan_wgs(WgsPlan &pl, int B, int H, int W, int Cin, int Cout, int k) {
    WgS &g = pl.g;
    g.Cin = Cin; g.Cout = Cout; g.cout8 = (Cout + 7) & ~7;
    g.H = H; g.W = W; g.pad = k / 2;
    g.XW = W + 2 * g.pad; g.VH = H + 2 * g.pad;
    g.Mflat = B * g.VH * g.XW;
    yh_set_magic((unsigned)g.XW, g.xw_magic, g.xw_shift);
    yh_set_magic((unsigned)g.VH, g.vh_magic, g.vh_shift);
    pl.KK = k * k;
    const int BP = k == 3 ? WgsCfg<9>::BP : WgsCfg<1>::BP;
    g.nblk_total = cdiv(g.Mflat, BP);
    // 64-channel tiles read every line whole and once; for the short streams of the 40x40 / 20x20 3x3 layers one input plane
    // per workgroup (twice the tiles, half the splits and slab bytes) measured 5-15 % faster (tools/wgs_sweep.sh)
    pl.NI = Cin > 32 && (k == 1 || Cin >= 256 || (g.nblk_total >= 800 && (Cin >= 128 || g.nblk_total >= 1600))) ? 2 : 1;
    pl.NJ = Cout > 32 ? 2 : 1;
    {   // the rings must fit the CU's 160 KB (wide rows: a 160-pixel row needs a 912-position x ring)
        const int hb = k == 1 ? 0 : cdiv(g.XW + 1, BP);
        auto rings_b = [&](int ni, int nj) { return (size_t)ni * ((2 * hb + 3) * BP + 16) * 64 + (size_t)nj * 3 * BP * 64; };
        if (rings_b(pl.NI, pl.NJ) > 160 * 1024) pl.NI = 1;
        if (rings_b(pl.NI, pl.NJ) > 160 * 1024) pl.NJ = 1;
    }
#ifdef YH_WGS_TUNE
    if (getenv("YH_WGS_NJ") && Cout > 32) pl.NJ = atoi(getenv("YH_WGS_NJ"));
    if (getenv("YH_WGS_NI") && Cin > 32) pl.NI = atoi(getenv("YH_WGS_NI"));
#endif
    g.n_ci_tiles = cdiv(Cin, 32 * pl.NI);
    pl.ntiles = g.n_ci_tiles * cdiv(Cout, 32 * pl.NJ);
    g.HB = k == 1 ? 0 : cdiv(g.XW + 1, BP);
    g.RX = (2 * g.HB + 3) * BP;
    g.RD = 3 * BP;
    for (int u = 0; u < 9; ++u) g.tapoff[u] = k == 3 ? (u / 3 - 1) * g.XW + (u % 3 - 1) : 0;
    // one (3x3) or two (1x1) workgroups per CU; a split no shorter than twice its halo fill (every split re-reads 2 HB + 1
    // blocks of x)
    int want = (k == 3 ? 256 : 512) / pl.ntiles;
#ifdef YH_WGS_TUNE
    if (getenv("YH_WGS_SPLITS")) want = atoi(getenv("YH_WGS_SPLITS"));
#endif
    if (want < 1) want = 1;
    int bps = cdiv(g.nblk_total, want);
    const int min_bps = 2 * (2 * g.HB + 1) + 2;
    if (bps < min_bps) bps = min_bps;
    if (bps > g.nblk_total) bps = g.nblk_total;
    g.blk_per_split = bps;
    pl.nsplit = cdiv(g.nblk_total, bps);
    g.ntiles = pl.ntiles;
    g.nsplit = pl.nsplit;
    const size_t rings = (size_t)pl.NI * (g.RX + 16) * 64 + (size_t)pl.NJ * g.RD * 64;
    const int WK = ((k == 3 ? WgsCfg<9>::NT : WgsCfg<1>::NT) / 64) / (pl.NI * pl.NJ);
    const size_t red = WK > 1 ? (size_t)pl.NI * pl.NJ * (pl.KK == 9 ? 3 : 1) * 32 * 33 * sizeof(float) : 0;
    pl.smem = rings > red ? rings : red;
}

template <int KK, int NI, int NJ>
int launch_wgs(const WgsPlan &pl, hipStream_t st) {
    auto kern = bf16_wgrad_stream_kernel<KK, NI, NJ>;
    if (int rc = yh_ensure_dyn_smem((const void *)kern, pl.smem)) return rc;
    hipLaunchKernelGGL(kern, dim3(pl.nsplit * pl.ntiles), dim3(WgsCfg<KK>::NT), pl.smem, st, pl.g);
    YH_CHECK_LAUNCH("bf16_wgrad_stream");
    return 0;
}

}  // namespace

bool yh_bf16_wgrad_stream_ok(int W, int Cin, int Cout, int k, int s) {
    // (k = 3: the x ring holds 2 (W + 2) + 2 positions of halo around a block; rows up to 638 pixels fit one 32-channel plane)
    return s == 1 && (k == 1 || (k == 3 && W <= 600)) && Cin % 8 == 0 && Cin >= 8 && Cout >= 1;
}

int64_t yh_bf16_wgrad_stream_ws(int B, int H, int W, int Cin, int Cout, int k) {
    WgsPlan pl{};
    plan_wgs(pl, B, H, W, Cin, Cout, k);
    return (int64_t)pl.nsplit * k * k * Cin * Cout;
}

int yh_bf16_wgrad_stream(const void *x, int ldx, const void *dy, int lddy, float *ws, int64_t ws_floats, int B, int H, int W, int Cin,
                         int Cout, int k, int *nsplit, hipStream_t st) {
    WgsPlan pl{};
    plan_wgs(pl, B, H, W, Cin, Cout, k);
    YH_REQUIRE(ws_floats >= (int64_t)pl.nsplit * k * k * Cin * Cout, "bf16_wgrad_stream: workspace too small");
    YH_REQUIRE((int64_t)B * H * W * ldx * 2 < (1ll << 31) && (int64_t)B * H * W * lddy * 2 < (1ll << 31) && pl.g.Mflat < (1 << 29),
               "bf16_wgrad_stream: tensor exceeds the 32-bit byte-offset range");
    pl.g.x = (const bf16 *)x; pl.g.dy = (const bf16 *)dy; pl.g.ws = ws; pl.g.ldx = ldx; pl.g.lddy = lddy;
    *nsplit = pl.nsplit;
    if (k == 3) {
        if (pl.NI == 2) return pl.NJ == 2 ? launch_wgs<9, 2, 2>(pl, st) : launch_wgs<9, 2, 1>(pl, st);
        return pl.NJ == 2 ? launch_wgs<9, 1, 2>(pl, st) : launch_wgs<9, 1, 1>(pl, st);
    }
    if (pl.NI == 2) return pl.NJ == 2 ? launch_wgs<1, 2, 2>(pl, st) : launch_wgs<1, 2, 1>(pl, st);
    return pl.NJ == 2 ? launch_wgs<1, 1, 2>(pl, st) : launch_wgs<1, 1, 1>(pl, st);
}

// =====================================================================================================================
// Forward / backward-data of the stride-1 layers as a flat stream:  out[f][n] (+)= sum_{tap, c} in[f + off(tap)][c] w[tap][c][n]
// (backward-data = the same sum over dY with the flipped, transposed filter pack).
//
// A persistent 256-thread workgroup owns a contiguous range of 128-position tiles and a column block of 32 TN channels:
//   * each wave keeps its B fragments (taps x k-steps x column tiles, <= 144 registers) in REGISTERS for its whole tile range;
//   * input positions travel HBM -> registers -> an LDS ring ONCE: the ring holds the 2 (XW + 1) positions of halo that the
//     nine taps of consecutive tiles share (the segment kernel re-requested every input element nine times from L1 / L2);
//     a slot is Cin * 2 + 16 bytes, so the 16 rows of a ds_read_b128 lane group start in 16 distinct bank quads;
//   * loads run PD - 1 = 3 tiles ahead in three register sets, one barrier per tile publishes the next tile's 128 positions;
//   * wave w multiplies rows 32 w .. 32 w + 31 of the tile against all 32 TN columns (A fragment shared by TN MFMAs);
//   * epilogue: bias, rounding to bf16, BatchNorm sums of the ROUNDED values over real (non-padding) positions kept in
//     registers across the whole tile range (ONE partial row per workgroup: a cheap bn_finalize), transpose through LDS,
//     16-byte stores of the real positions only.
namespace {

struct FsP {
    const bf16 *in, *in2;      // in2: channels >= ksplit come from a second tensor (the C3 sibling pair's backward-data)
    const bf16 *w;             // pack [tap][Cin/8][ldw][8]
    const float *bias;
    bf16 *out;
    float *stats;              // [gridDim.x][2][N]
    int ksplit, ldw, ldi, ldo, N, accumulate;
    int H, W, XW, VH, pad, Mflat;
    unsigned xw_magic, vh_magic;
    int xw_shift, vh_shift;
    int ntile, tiles_per_wg, gx, gy;
    int R, halo;               // ring slots; XW + 1 (3x3) or 0 (1x1)
    int tapoff[9], tapw[9];
#ifdef YH_FS_STAMPS
    unsigned long long *dbg;   // diagnostic build: per-workgroup phase cycle sums
#endif
};

template <int KK>
__device__ __forceinline__ bool fs_pixel(const FsP &g, int f, int &pix) {
    if (KK == 1) {
        pix = f;
        return (unsigned)f < (unsigned)g.Mflat;
    }
    const bool in = (unsigned)f < (unsigned)g.Mflat;
    const int ff = in ? f : 0;
    const int q1 = yh_fast_div(ff, g.xw_magic, g.xw_shift), vx = ff - q1 * g.XW - g.pad;
    const int b = yh_fast_div(q1, g.vh_magic, g.vh_shift), vy = q1 - b * g.VH - g.pad;
    pix = (b * g.H + vy) * g.W + vx;
    return in && (unsigned)vx < (unsigned)g.W && (unsigned)vy < (unsigned)g.H;
}

// WN = 1: four waves, each 32 rows x all 32 TN columns.  WN = 2: eight waves (two per SIMD: one wave's fragment reads, epilogue
// and staging run under the other's MFMAs), wave (wm, wn) = 32 rows x columns [16 TN wn, 16 TN (wn + 1)).
// OCC = 2: two four-wave workgroups per CU (column blocks of 32: the two blocks of a 64-column layer are independent workgroups
// whose stage / multiply / epilogue phases drift apart and overlap, which the barriers of one eight-wave workgroup prevent).
template <int KK, int CIN, int TN, int WN, int OCC>
__global__ __launch_bounds__(256 * WN, WN * OCC) void bf16_fstream_kernel(const FsP g) {
    constexpr int BM = 128, BN = 32 * TN, C8 = CIN / 8, KS = CIN / 16, PS = CIN * 2 + 16, PD = 4;
    constexpr int NTH = 256 * WN, TW = TN / WN;  // threads; column tiles per wave
    constexpr int NP = (BM * C8) / NTH;          // 16-byte pieces per thread and tile
    constexpr int NW = 4 * WN;                   // waves
    constexpr int CSW = 64 * TW + 16;            // bytes per row of a wave's private staging tile (32 rows x 32 TW columns)
    constexpr int PCW = 4 * TW;                  // 16-byte pieces per row of it
    static_assert(NP >= 1 && (BM * C8) % NTH == 0 && TN % WN == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *As = smem;                                  // [R][PS]
    unsigned char *Cs = As + g.R * PS;                         // [NW][32][CSW]
    int *rowpix = (int *)(Cs + NW * 32 * CSW);                 // [NW][32] output pixel of each of the wave's rows, -1 = padding position
    float *red = (float *)(rowpix + NW * 32);                  // [4][BN][2]
    const int t = threadIdx.x, lane = t & 63, wave_all = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wave = wave_all & 3, cbw = 32 * TW * (wave_all >> 2);      // row group; first column of this wave's tiles
    const int lr = lane & 31, lh = lane >> 5;
    // 1-D grid; the gy column blocks that stream the SAME tile range get consecutive slots of one XCD (workgroups b and b + 8
    // share an XCD and its L2): the second reader of a line finds it there.  Bijective (partial trailing group: plain order).
    int bx, by;
    {
        const int lin = blockIdx.x, per = 8 * g.gy, full = (g.gx >> 3) * per;
        if (lin < full) {
            const int grp = lin / per, r = lin - grp * per;
            bx = grp * 8 + (r & 7);
            by = r >> 3;
        } else {
            const int r = lin - full, rem = g.gx & 7;
            bx = (g.gx & ~7) + r % rem;
            by = r / rem;
        }
    }
    const int n0 = by * BN;
    const int T0 = bx * g.tiles_per_wg;
    int T1 = T0 + g.tiles_per_wg;
    if (T1 > g.ntile) T1 = g.ntile;

    // ---- weights: this wave's B fragments live in REGISTERS for the whole kernel (the pack layout [tap][K/8][ldw][8] is the
    // fragment layout: lane (col r, h) takes the 16 bytes of octet 2 ks + h, column r).  No LDS footprint, no LDS read per MFMA
    // for B: at two waves per SIMD the LDS port was as busy as the matrix pipe with both operands coming from it. -----------
    bf16x8 bfr[KK * KS][TW];
#pragma unroll
    for (int u = 0; u < KK; ++u)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                const int n = n0 + cbw + 32 * j + lr;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (n < g.ldw) v = *(const u32x4 *)(g.w + ((size_t)(g.tapw[u] * C8 + 2 * ks + lh) * g.ldw + n) * 8);
                bfr[u * KS + ks][j] = __builtin_bit_cast(bf16x8, v);
            }

    // ---- staging ------------------------------------------------------------------------------------------------------------
    const unsigned char *ib = (const unsigned char *)g.in, *ib2 = (const unsigned char *)g.in2;
    int ppix[NP], pch[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int e = t + NTH * i;
        ppix[i] = e / C8;
        pch[i] = e % C8;
    }
    u32x4 ra[PD][NP];
    unsigned mk[PD];
    // L(m): request the 128 positions [BM (m + PD) + halo, ...) (stored PD - 1 iterations later as "the new positions of tile m + PD")
    auto issue = [&](int m, u32x4 (&a)[NP], unsigned &mko) __attribute__((always_inline)) {
        unsigned mm = 0;
        const int p0 = BM * (m + PD) + g.halo;
        const bool live = m + PD <= T1;               // positions beyond the last tile's halo are never read
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int px;
            const bool ok = fs_pixel<KK>(g, p0 + ppix[i], px) && live;
            const int c = 8 * pch[i];
            const bool second = g.in2 && c >= g.ksplit;
            const unsigned off = ok ? (unsigned)(px * g.ldi + (second ? c - g.ksplit : c)) * 2u : 0u;
            mm |= (unsigned)ok << i;
            a[i] = *(const u32x4 *)((second ? ib2 : ib) + off);
        }
        mko = mm;
    };
    // ring cursors (wave-uniform): st = slot of position BM (n + 1) + halo, cb[u] = slot of position BM n + off(tap u)
    const int warm = (2 * g.halo + BM - 1) / BM + 1;
    const int nfirst = T0 - warm;
    int st = wgs_mod(BM * (nfirst + 1) + g.halo, g.R);
    int cb[KK];
#pragma unroll
    for (int u = 0; u < KK; ++u) cb[u] = wgs_mod(BM * nfirst + g.tapoff[u], g.R);
    auto advance = [&]() __attribute__((always_inline)) {
        st += BM; if (st >= g.R) st -= g.R;
#pragma unroll
        for (int u = 0; u < KK; ++u) { cb[u] += BM; if (cb[u] >= g.R) cb[u] -= g.R; }
    };
    auto store = [&](const u32x4 (&a)[NP], unsigned mki) __attribute__((always_inline)) {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int s = st + ppix[i];
            if (s >= g.R) s -= g.R;
            *(u32x4 *)(As + s * PS + pch[i] * 16) = (mki >> i) & 1u ? a[i] : z;
        }
    };

    // ---- multiply: wave w = rows 32 w .. 32 w + 31; lane (r, h) holds A[row r][k = 8 h ..] and B[k = 8 h ..][col r] ----------
    f32x16 acc[TW];
    float csum[TW], csq[TW];
#pragma unroll
    for (int j = 0; j < TW; ++j) csum[j] = csq[j] = 0.f;
    auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
        for (int u = 0; u < KK; ++u) {
            int s = cb[u] + 32 * wave + lr;
            if (s >= g.R) s -= g.R;
            const unsigned char *ap = As + s * PS + 16 * lh;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 av = *(const bf16x8 *)(ap + 32 * ks);
#pragma unroll
                for (int j = 0; j < TW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bfr[u * KS + ks][j], acc[j], 0, 0, 0);
            }
        }
    };
    float bias_v[TW];
#pragma unroll
    for (int j = 0; j < TW; ++j) {
        const int n = n0 + cbw + 32 * j + lr;
        bias_v[j] = (g.bias && n < g.N) ? g.bias[n] : 0.f;
    }
#ifdef YH_FS_STAMPS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tk = __builtin_amdgcn_s_memtime();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#define FS_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph[i] += now_ - tk; tk = now_; } while (0)
#else
#define FS_STAMP(i) do { } while (0)
#endif
    // The epilogue of a wave touches only that wave's LDS (its 32 row pixels, its 32 x 32 TW staging tile): no barrier, so the
    // waves of a workgroup are coupled by the ONE barrier per tile that publishes the ring.
    int *const rpw = rowpix + 32 * wave_all;
    unsigned char *const csw = Cs + wave_all * 32 * CSW;
    auto epilogue = [&](int n) __attribute__((always_inline)) {
        if (lane < 32) {
            int px;
            rpw[lane] = fs_pixel<KK>(g, BM * n + 32 * wave + lane, px) ? px : -1;
        }
        float rv[16];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const i32x4 p4 = *(const i32x4 *)(rpw + 8 * q4 + 4 * lh);
#pragma unroll
            for (int e = 0; e < 4; ++e) rv[4 * q4 + e] = p4[e] >= 0 ? 1.f : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            unsigned char *row = csw + yh_mfma_row(q, lh) * CSW + lr * 2;
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                const bf16 hv = (bf16)(acc[j][q] + bias_v[j]);
                *(bf16 *)(row + j * 64) = hv;
                const float vq = (float)hv * rv[q];
                csum[j] += vq;
                csq[j] += vq * (float)hv;
            }
        }
#pragma unroll
        for (int i = 0; i < (32 * PCW) / 64; ++i) {
            const int e = lane + 64 * i, rl = e / PCW, oc = e % PCW;
            const int px = rpw[rl], nn = n0 + cbw + 8 * oc;
            if (px >= 0 && nn < g.N) {
                bf16x8 v = *(const bf16x8 *)(csw + rl * CSW + oc * 16);
                bf16 *o = g.out + (size_t)px * g.ldo + nn;
                if (g.accumulate) {
                    const bf16x8 old = *(const bf16x8 *)o;
                    const f32x8 sum = __builtin_convertvector(v, f32x8) + __builtin_convertvector(old, f32x8);
                    v = __builtin_convertvector(sum, bf16x8);
                }
                *(bf16x8 *)o = v;
            }
        }
        FS_STAMP(4);
    };

    // ---- pipeline (same shape as the weight-gradient stream): iteration n stores the new positions of tile n + 1, requests
    // those of tile n + PD, multiplies tile n; the first `warm` iterations only fill the ring ---------------------------------
    // (Tried and dropped: letting the second wave group of an eight-wave workgroup write tile n - 1 out under the first group's
    // MFMAs of tile n -- two copies of the loop, or one copy with a run-time order, cost 25+ registers and spilled at the 256
    // the 144 B-fragment registers leave: 64 -> 64 3x3 at 80x80 50 -> 70 us.)
#pragma unroll
    for (int p = 0; p < PD - 1; ++p) issue(nfirst - (PD - 1) + p, ra[p], mk[p]);
    for (int n = nfirst; n < T1; n += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            if (n + p < T1) {
                issue(n + p, ra[(p + PD - 1) % PD], mk[(p + PD - 1) % PD]);
                store(ra[p], mk[p]);
                __syncthreads();
                FS_STAMP(0);
                if (n + p >= T0) {
                    compute();
                    FS_STAMP(1);
                    epilogue(n + p);
                }
                advance();
            }
        }
    }
#ifdef YH_FS_STAMPS
    if (g.dbg && t == 0) {
        unsigned long long *d = g.dbg + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 5; ++i) d[i] = ph[i];
        d[5] = (unsigned long long)(T1 - T0);
        d[6] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
    if (g.stats) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const float s = csum[j] + __shfl_xor(csum[j], 32), q = csq[j] + __shfl_xor(csq[j], 32);
            if (lh == 0) {
                red[((wave * BN) + cbw + 32 * j + lr) * 2 + 0] = s;
                red[((wave * BN) + cbw + 32 * j + lr) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (t < BN && n0 + t < g.N) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                s += red[((w * BN) + t) * 2 + 0];
                q += red[((w * BN) + t) * 2 + 1];
            }
            g.stats[((size_t)bx * 2 + 0) * g.N + n0 + t] = s;
            g.stats[((size_t)bx * 2 + 1) * g.N + n0 + t] = q;
        }
    }
}

struct FsPlan {
    FsP g;
    int KK, CIN, TN, WN, OCC, gx, gy;
    size_t smem;
    bool ok;
};

// K = channels of the streamed tensor (16 / 32 / 64 / 128), N = output channels (bf16 output, a multiple of 8)
void plan_fs(FsPlan &pl, int B, int H, int W, int K, int N, int k) {
    FsP &g = pl.g;
    pl.ok = false;
    pl.KK = k * k;
    pl.CIN = K;
    if (!(K == 16 || K == 32 || K == 64 || K == 128) || (k != 1 && k != 3) || N % 8 != 0 || N < 8) return;
    if (k == 3 && K > 64) return;                               // the weight block + halo ring of a 128-channel 3x3 layer exceed the LDS
    g.H = H; g.W = W; g.pad = k / 2; g.XW = W + 2 * g.pad; g.VH = H + 2 * g.pad; g.N = N;
    g.Mflat = B * g.VH * g.XW;
    if (g.Mflat >= (1 << 29)) return;
    yh_set_magic((unsigned)g.XW, g.xw_magic, g.xw_shift);
    yh_set_magic((unsigned)g.VH, g.vh_magic, g.vh_shift);
    g.halo = k == 3 ? g.XW + 1 : 0;
    g.R = (3 * 128 + 2 * g.halo + 15) & ~15;                      // one tile more than the taps span: a wave may store the next
                                                                // tile's positions while another still multiplies (one barrier per tile)
    // column block: all of N up to 128 columns (k = 1) / 64 columns (k = 3), so the input is streamed once
    const int cap = k == 3 ? 64 : 128;
    pl.TN = N > 64 ? (cap >= 128 ? 4 : 2) : (N > 32 ? 2 : 1);
    // four-wave workgroups with one 32-column block can run TWO per CU (their phases drift apart and overlap; measured on the
    // 3x3 layers, us at one / two per CU: 32 -> 32 at 80x80 34 / 26, 16 -> 16 at 160x160 74 / 52; 64 -> 64 would need column
    // blocks of 32 and two passes over the input: 50 / 57, so it keeps the eight-wave form)
    pl.OCC = (pl.TN == 1 && K <= (k == 3 ? 32 : 64)) ? 2 : 1;
    const int BN = 32 * pl.TN;
    pl.gy = cdiv(N, BN);
    g.ntile = cdiv(g.Mflat, 128);
    int gx = 256 * pl.OCC / pl.gy;                               // OCC workgroups per CU
    if (gx < 1) gx = 1;
    if (gx > g.ntile) gx = g.ntile;
    g.tiles_per_wg = cdiv(g.ntile, gx);
    pl.gx = cdiv(g.ntile, g.tiles_per_wg);
    g.gx = pl.gx; g.gy = pl.gy;
    const int PS = K * 2 + 16;
    pl.WN = (pl.TN >= 2 && K >= 32) ? 2 : 1;                    // eight waves wherever a wave still owns a whole 32-column tile
    const int NW = 4 * pl.WN, CSW = 64 * (pl.TN / pl.WN) + 16;
    pl.smem = (size_t)g.R * PS + (size_t)NW * 32 * CSW + (size_t)NW * 32 * sizeof(int) + (size_t)4 * BN * 2 * sizeof(float);
    if (pl.smem > 160 * 1024) return;
    // (measured and dropped: 32-column blocks at two per CU for K = 64 pointwise layers -- the input is streamed twice: 24.8 ->
    // 26.5 us for 64 -> 64 at 80x80; four waves x 64 columns at two per CU: 24.8 -> 28.9 us)
    if (pl.KK * (K / 16) * (pl.TN / pl.WN) * 4 > 160) return;   // B fragments in registers: taps * k-steps * column tiles * 4 VGPRs
    pl.ok = true;
}

template <int KK, int CIN, int TN, int WN, int OCC = 1>
int launch_fs(const FsPlan &pl, hipStream_t st) {
    auto kern = bf16_fstream_kernel<KK, CIN, TN, WN, OCC>;
    if (int rc = yh_ensure_dyn_smem((const void *)kern, pl.smem)) return rc;
#ifdef YH_FS_STAMPS
    static unsigned long long *dbgbuf = nullptr;
    if (!dbgbuf) (void)hipMalloc((void **)&dbgbuf, (size_t)1 << 20);
    const bool dbg_on = getenv("YH_FS_DBG") != nullptr;
    FsPlan pd = pl;
    pd.g.dbg = dbg_on ? dbgbuf : nullptr;
    hipLaunchKernelGGL(kern, dim3(pl.gx * pl.gy), dim3(256 * WN), pl.smem, st, pd.g);
    if (dbg_on) {
        (void)hipStreamSynchronize(st);
        const int nb = pl.gx * pl.gy;
        std::vector<unsigned long long> h((size_t)nb * 8);
        (void)hipMemcpy(h.data(), dbgbuf, h.size() * 8, hipMemcpyDeviceToHost);
        double a[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < nb; ++i)
            for (int k = 0; k < 7; ++k) a[k] += (double)h[8 * i + k];
        const double tiles = a[5] / nb;
        fprintf(stderr, "fstream<%d,%d,%d,%d,%d> wgs %d tiles/wg %.1f | cycles per tile: stage+barrier %.0f  mfma %.0f  xch/rowpix barrier %.0f  "
                        "Cs write+barrier %.0f  stores %.0f | wall %.1f us/wg\n", KK, CIN, TN, WN, OCC, nb, tiles, a[0] / a[5], a[1] / a[5],
                a[2] / a[5], a[3] / a[5], a[4] / a[5], a[6] / nb / 100.0);
    }
    return 0;
#else
    hipLaunchKernelGGL(kern, dim3(pl.gx * pl.gy), dim3(256 * WN), pl.smem, st, pl.g);
    YH_CHECK_LAUNCH("bf16_fstream");
    return 0;
#endif
}

template <int KK, int CIN>
int launch_fs_tn(const FsPlan &pl, hipStream_t st) {
    constexpr int W2 = CIN >= 32 ? 2 : 1;
    if (pl.TN == 1 && pl.OCC == 2) {
        if constexpr (CIN <= (KK == 9 ? 32 : 64)) return launch_fs<KK, CIN, 1, 1, 2>(pl, st);
    }
    if (pl.TN == 1) return launch_fs<KK, CIN, 1, 1>(pl, st);
    if (pl.TN == 2) return launch_fs<KK, CIN, 2, W2>(pl, st);
    if (KK == 1 && pl.TN == 4) return launch_fs<1, CIN, 4, W2>(pl, st);
    yh_set_error("bf16_fstream: column block of %d not instantiated", 32 * pl.TN);
    return YH_E_UNSUPPORTED;
}

}  // namespace

// Where the flat stream measured faster than (or equal to) the gather GEMM (tools/fs_bench.py, batch 64, us stream / gather):
// every supported 3x3 (64 -> 64 at 80x80: 50 / 68-74, at 40x40: 20 / 24; 32 -> 32 at 80x80: 32-34 / 32-36; 16 -> 16 at
// 160x160: 78 / 93); pointwise layers over >= 1 M pixels (32 -> 16 at 160x160: 51 / 60) or with N >= 64 and K <= N
// (64 -> 64 at 80x80: 24.5 / 25.0, 128 -> 128 at 40x40: 17.9 / 18.1 -- equal, and ONE BatchNorm partial row per workgroup
// instead of one per 128 pixels) or with N <= 32 and K <= 64 (two workgroups per CU: 64 -> 32 at 80x80 18.8 / 19.8); the 128 -> 32
// layers stay on the gather GEMM (28.7 / 25.9).
bool yh_bf16_fstream_supported(int B, int H, int W, int K, int N, int k, int s) {
    if (s != 1) return false;
    FsPlan pl{};
    plan_fs(pl, B, H, W, K, N, k);
    return pl.ok;
}
bool yh_bf16_fstream_ok(int B, int H, int W, int K, int N, int k, int s) {
    if (!yh_bf16_fstream_supported(B, H, W, K, N, k, s)) return false;
    return k == 3 || (int64_t)B * H * W >= (1 << 20) || (N >= 64 && K <= N) || (N <= 32 && K <= 64);
}

int yh_bf16_fstream_blocks(int B, int H, int W, int K, int N, int k) {
    FsPlan pl{};
    plan_fs(pl, B, H, W, K, N, k);
    return pl.ok ? pl.gx : 0;
}

// taps: tapoff (dy, dx as flat displacement) and pack index per tap, nTaps = k * k entries
int yh_bf16_fstream(const void *in, const void *in2, int ksplit, int ldi, const void *w, int ldw, const float *bias, void *out, int ldo,
                    float *stats, int B, int H, int W, int K, int N, int k, int accumulate, const int *tap_dy, const int *tap_dx,
                    const int *tap_w, hipStream_t st) {
    FsPlan pl{};
    plan_fs(pl, B, H, W, K, N, k);
    YH_REQUIRE(pl.ok, "bf16_fstream: unsupported problem K=%d N=%d k=%d", K, N, k);
    YH_REQUIRE((int64_t)B * H * W * ldi * 2 < (1ll << 31), "bf16_fstream: tensor exceeds the 32-bit byte-offset range");
    FsP &g = pl.g;
    g.in = (const bf16 *)in; g.in2 = (const bf16 *)in2; g.ksplit = ksplit; g.ldi = ldi; g.w = (const bf16 *)w; g.ldw = ldw;
    g.bias = bias; g.out = (bf16 *)out; g.ldo = ldo; g.stats = stats; g.accumulate = accumulate;
    for (int u = 0; u < k * k; ++u) {
        g.tapoff[u] = tap_dy[u] * g.XW + tap_dx[u];
        g.tapw[u] = tap_w[u];
    }
    if (k == 3) {
        switch (K) {
            case 16: return launch_fs_tn<9, 16>(pl, st);
            case 32: return launch_fs_tn<9, 32>(pl, st);
            default: return launch_fs_tn<9, 64>(pl, st);
        }
    }
    switch (K) {
        case 16: return launch_fs_tn<1, 16>(pl, st);
        case 32: return launch_fs_tn<1, 32>(pl, st);
        case 64: return launch_fs_tn<1, 64>(pl, st);
        default: return launch_fs_tn<1, 128>(pl, st);
    }
}
