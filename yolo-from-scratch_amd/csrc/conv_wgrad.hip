// Backward-weight convolution on v_mfma_f32_32x32x2_f32 (gfx950), NHWC fp32.
//
//   dW[tap][ci][co] = sum over output pixels p of  x[p*s + tap - pad][ci] * dy[p][co]
//
// GEMM view: rows = (tap, ci) flattened, cols = co, reduction = pixels.  In NHWC both MFMA operands
// are "pixel-major with channels contiguous", which is exactly the A[i][k] / B[k][j] lane maps of
// the 32x32x2 instruction (lane l: row/col l&31, k = l>>5): a lane pair of pixels (p, p+1) supplies
// k = 0,1, and consecutive lanes read consecutive channels, so LDS fragment reads are conflict free
// without any transposition.
//
// A workgroup (4 waves) owns a (CIT input channels x COT output channels x all taps) slab of dW and a
// contiguous range of pixel segments (32 output pixels of one output row).  Per segment it stages the
// input halo rows and the dy row in LDS, then every wave accumulates its share of the slab's 32x32
// tiles in registers across the whole range.  Partial slabs go to a workspace [split][tap][ci][co];
// a second kernel adds the splits in a fixed order and writes OIHW -- deterministic, no atomics.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int PMAX = 40;   // output pixels per segment (even, chosen per layer so rows split evenly)

struct Wgrad {
    const float *x, *dy;
    const float *icoef;         // input prologue table of x ([scale | shift | gate] rows, icoef_ld apart; yh_prologue), or null
    int icoef_ld;
    float *ws;
    int ldx, lddy;
    int B, Hi, Wi, Ho, Wo;
    int Cin, Cout, k, s, pad;
    int CIT, COT, n_ci_tiles;
    int nseg_row, nseg_total, segs_per_split;
    int vec_dy, P, MT;
};

// XL / DL: compile-time upper bounds of the float4 staging loads per thread for the input halo and
// the dy row, so the loads of the NEXT segment are issued back to back into registers and stay in
// flight while the MFMAs of the current segment run (register-staged double buffering).
template <int MT> struct WMfma;
template <> struct WMfma<32> {
    typedef f32x16 Acc;
    static constexpr int NR = 16;
    static __device__ __forceinline__ Acc run(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }
};
template <> struct WMfma<16> {
    typedef f32x4 Acc;
    static constexpr int NR = 4;
    static __device__ __forceinline__ Acc run(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int r, int lh) { return 4 * lh + r; }
};

// MT = MFMA tile edge: 32 (32x32x2, two pixels per instruction) or 16 (16x16x4, four pixels; used when
// the slab has <= 16 output channels or few (tap, ci) rows, where 32-wide tiles would be mostly padding).
template <int NT, int XL, int DL, int MT>
__global__ __launch_bounds__(256) void wgrad_kernel(const Wgrad g) {
    typedef WMfma<MT> MF;
    constexpr int LG = 64 / MT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & (MT - 1), lh = lane / MT;
    const int KK = g.k * g.k;
    const int P = g.P;
    const int XW = (P - 1) * g.s + g.k;
    const int CIT = g.CIT, COT = g.COT;
    float *Xs = smem;                          // [k][XW][CIT]
    float *Ds = smem + g.k * XW * CIT;         // [P][COT]

    const int ci0 = (blockIdx.y % g.n_ci_tiles) * CIT;
    const int co0 = (blockIdx.y / g.n_ci_tiles) * COT;
    const int rows = KK * CIT;
    const int RT = (rows + MT - 1) / MT, OT = (COT + MT - 1) / MT;

    // tile ownership: a wave owns ONE column tile (so a single dy fragment per step feeds all its MFMAs)
    // and every WR-th row tile: ot = wave % OT, rt = wave / OT + WR * u, WR = 4 / OT (OT is 1, 2 or 4).
    // Rows / columns past the slab read in-bounds garbage whose products are never stored: no masks.
    const int WR = 4 / OT;
    const int ot = wave % OT, wr = wave / OT;
    int aoff[NT];
    typename MF::Acc acc[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        int row = (wr + WR * u) * MT + lr;
        if (row >= rows) row = lr % rows;
        int tap = row / CIT, ci = row % CIT;
        aoff[u] = ((tap / g.k) * XW + (tap % g.k)) * CIT + ci;
#pragma unroll
        for (int r = 0; r < MF::NR; ++r) acc[u][r] = 0.f;
    }
    const int bcol = ot * MT + lr;
    const int boff = bcol < COT ? bcol : 0;

    // fixed staging slots of this thread: slot j handles float4 index t + 256 j
    const int citq = CIT >> 2, cotq = COT >> 2;
    const int nx = g.k * XW * citq, nd = P * cotq;
    int xkh[XL], xcol[XL], xc[XL], dp[DL], dc[DL];
#pragma unroll
    for (int j = 0; j < XL; ++j) {
        int i = t + 256 * j;
        int c4 = i % citq, q = i / citq;
        xcol[j] = q % XW; xkh[j] = q / XW; xc[j] = 4 * c4;
        if (i >= nx) xkh[j] = -1;
    }
#pragma unroll
    for (int j = 0; j < DL; ++j) {
        int i = t + 256 * j;
        dc[j] = 4 * (i % cotq); dp[j] = i / cotq;
        if (i >= nd) dp[j] = -1;
    }

    const int seg_begin = blockIdx.x * g.segs_per_split;
    int seg_end = seg_begin + g.segs_per_split;
    if (seg_end > g.nseg_total) seg_end = g.nseg_total;

    f32x4 rx[XL], rd[DL];
    // input prologue: applied when a segment is parked.  The channel quad of a thread's slots is xc[j] (the same for every slot when
    // 256 % (CIT / 4) == 0, which the launcher requires with a table); padding must stay zero AFTER the activation: validity bits.
    const bool act = g.icoef != nullptr;                   // workgroup-uniform
    f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f}, pgt = {0.f, 0.f, 0.f, 0.f};
    unsigned xokm = 0;
    if (act) {
        int ch = ci0 + xc[0];
        ch = ch + 4 <= g.Cin ? ch : 0;
        psc = *(const f32x4 *)(g.icoef + ch); psh = *(const f32x4 *)(g.icoef + g.icoef_ld + ch); pgt = *(const f32x4 *)(g.icoef + 2 * g.icoef_ld + ch);
    }
    auto load_seg = [&](int seg) {
        const int sr = seg % g.nseg_row, rowid = seg / g.nseg_row;
        const int ho = rowid % g.Ho, b = rowid / g.Ho;
        const int w0 = sr * P;
        const int pv = (g.Wo - w0) < P ? (g.Wo - w0) : P;
        const int xcols = (pv - 1) * g.s + g.k;
        xokm = 0;
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            int hi = ho * g.s + xkh[j] - g.pad, wi = w0 * g.s - g.pad + xcol[j];
            int c = ci0 + xc[j];
            bool ok = xkh[j] >= 0 && xcol[j] < xcols && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi && c < g.Cin;
            rx[j] = ok ? *(const f32x4 *)(g.x + ((size_t)(b * g.Hi + hi) * g.Wi + wi) * g.ldx + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            xokm |= (unsigned)ok << j;
        }
        if (g.vec_dy) {
#pragma unroll
            for (int j = 0; j < DL; ++j) {
                int c = co0 + dc[j];
                bool ok = dp[j] >= 0 && dp[j] < pv && c < g.Cout;
                rd[j] = ok ? *(const f32x4 *)(g.dy + ((size_t)(b * g.Ho + ho) * g.Wo + w0 + dp[j]) * g.lddy + c)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_seg = [&](int seg) {
#pragma unroll
        for (int j = 0; j < XL; ++j)
            if (xkh[j] >= 0) {
                f32x4 v = rx[j];
                if (act && (xokm >> j & 1)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = yh_prologue(v[e], psc[e], psh[e], pgt[e]);
                }
                *(f32x4 *)(Xs + (xkh[j] * XW + xcol[j]) * CIT + xc[j]) = v;
            }
        if (g.vec_dy) {
#pragma unroll
            for (int j = 0; j < DL; ++j)
                if (dp[j] >= 0) *(f32x4 *)(Ds + dp[j] * COT + dc[j]) = rd[j];
        } else {   // dy rows that are not 16-byte addressable (head outputs: 18 or 255 channels): scalar staging
            const int sr = seg % g.nseg_row, rowid = seg / g.nseg_row;
            const int ho = rowid % g.Ho, b = rowid / g.Ho;
            const int w0 = sr * P;
            const int pv = (g.Wo - w0) < P ? (g.Wo - w0) : P;
            for (int i = t; i < P * COT; i += 256) {
                int c = i % COT, p = i / COT;
                float v = 0.f;
                if (p < pv && co0 + c < g.Cout) v = g.dy[((size_t)(b * g.Ho + ho) * g.Wo + w0 + p) * g.lddy + co0 + c];
                Ds[p * COT + c] = v;
            }
        }
    };

    if (seg_begin < seg_end) load_seg(seg_begin);
    for (int seg = seg_begin; seg < seg_end; ++seg) {
        const int w0 = (seg % g.nseg_row) * P;
        const int pv = (g.Wo - w0) < P ? (g.Wo - w0) : P;
        __syncthreads();   // previous segment's fragment reads are done
        store_seg(seg);
        __syncthreads();
        if (seg + 1 < seg_end) load_seg(seg + 1);
        const int npairs = (pv + LG - 1) / LG;      // MFMA steps: LG pixels each
        // two steps per iteration: the fragment reads of step pp+1 are in flight behind the MFMAs of step pp.
        // Branch-free: tiles past the slab (at most one per wave) multiply zeros / unused columns, never stored.
        int pp = 0;
        for (; pp + 1 < npairs; pp += 2) {
            const int p0 = LG * pp + lh, p1 = p0 + LG;
            const float *x0 = Xs + p0 * g.s * CIT, *x1 = Xs + p1 * g.s * CIT;
            const float b0 = Ds[boff + p0 * COT], b1 = Ds[boff + p1 * COT];
            float a0[NT], a1[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) { a0[u] = x0[aoff[u]]; a1[u] = x1[aoff[u]]; }
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[u] = MF::run(a0[u], b0, acc[u]);
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[u] = MF::run(a1[u], b1, acc[u]);
        }
        if (pp < npairs) {
            const int p0 = LG * pp + lh;
            const float *x0 = Xs + p0 * g.s * CIT;
            const float b0 = Ds[boff + p0 * COT];
            float a0[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) a0[u] = x0[aoff[u]];
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[u] = MF::run(a0[u], b0, acc[u]);
        }
    }

    // write this split's partial slab: ws[split][tap][ci][co]
    float *wsp = g.ws + (size_t)blockIdx.x * KK * g.Cin * g.Cout;
    if (bcol < COT && co0 + bcol < g.Cout) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int rt = wr + WR * u;
#pragma unroll
            for (int r = 0; r < MF::NR; ++r) {
                int row = rt * MT + MF::row(r, lh);
                if (row >= rows) continue;
                int tap = row / CIT, ci = ci0 + row % CIT;
                if (ci >= g.Cin) continue;
                wsp[((size_t)tap * g.Cin + ci) * g.Cout + co0 + bcol] = acc[u][r];
            }
        }
    }
}

// dw[co][ci][tap] (ci < cin_real) = sum_split ws[split][tap][ci][co].  A workgroup owns 16 consecutive
// elements; 16 split-lanes each add every 16th split, then the lanes are combined in lane order:
// the summation order is fixed, so the result is bitwise reproducible.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dw,
                                                           int nsplit, int KK, int Cin, int cin_real, int Cout) {
    __shared__ float red[16][17];
    const int n = KK * Cin * Cout;
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
        if (ci < cin_real) dw[((size_t)co * cin_real + ci) * KK + tap] = tot;
    }
}

struct Plan {
    Wgrad g;
    int nsplit, ntiles, NT;
    size_t smem;
};

int make_plan(Plan &pl, int B, int Hi, int Wi, int Cin, int Cout, int k, int s) {
    Wgrad &g = pl.g;
    g = Wgrad{};
    g.k = k; g.s = s; g.pad = k / 2; g.Cin = Cin; g.Cout = Cout;
    if (k == 1) {   // pointwise: pixels are independent, walk them as one long row
        if (s != 1) return YH_E_UNSUPPORTED;
        int64_t M = (int64_t)B * Hi * Wi;
        if (M >= (1ll << 31)) return YH_E_UNSUPPORTED;
        g.B = 1; g.Hi = 1; g.Wi = (int)M; g.Ho = 1; g.Wo = (int)M;
    } else {
        g.B = B; g.Hi = Hi; g.Wi = Wi;
        g.Ho = (Hi + 2 * g.pad - k) / s + 1; g.Wo = (Wi + 2 * g.pad - k) / s + 1;
    }
    // slab shape: keep <= 36 tiles (9 per wave) and LDS modest
    // column tiles per slab must be 1, 2 or 4 (wave ownership): COT is 128, 64 or the whole (<= 64) channel count
    if (k == 1) { g.CIT = Cin >= 128 ? 128 : Cin; g.COT = Cout >= 128 ? 128 : (Cout > 64 ? 64 : (Cout + 3) / 4 * 4); }
    else {
        constexpr int cit3 = 32;
        g.CIT = (s == 2) ? (Cin >= 32 ? 32 : Cin) : (Cin >= cit3 ? cit3 : Cin);
        g.COT = Cout >= 64 ? 64 : (Cout + 3) / 4 * 4;
    }
    if (g.CIT % 4) return YH_E_UNSUPPORTED;
    g.n_ci_tiles = cdiv(Cin, g.CIT);
    pl.ntiles = g.n_ci_tiles * cdiv(Cout, g.COT);
    {   // MFMA shape: cycles per pixel = tiles32 * 64/2 vs tiles16 * 32/4
        int t32 = cdiv(k * k * g.CIT, 32) * cdiv(g.COT, 32), t16 = cdiv(k * k * g.CIT, 16) * cdiv(g.COT, 16);
        constexpr int force = 0;
        const int ot16 = cdiv(g.COT, 16);
        const bool ok16 = t16 <= 20 && (ot16 == 1 || ot16 == 2 || ot16 == 4);
        g.MT = (t16 < 4 * t32 && ok16) ? 16 : 32;
        if (force == 16 && ok16) g.MT = 16;
        if (force == 32) g.MT = 32;
        int RT = cdiv(k * k * g.CIT, g.MT), OT = cdiv(g.COT, g.MT);
        if (OT != 1 && OT != 2 && OT != 4) return YH_E_UNSUPPORTED;
        pl.NT = cdiv(RT, 4 / OT);
    }
    // segment length: the candidate with the least padded work that fits the staging-register budget
    {
        const int cand[4] = {40, 32, 20, 16};
        int best = 32, best_work = 1 << 30;
        for (int c : cand) {
            int xw = (c - 1) * s + k;
            int xl = cdiv(k * xw * (g.CIT / 4), 256), dl = cdiv(c * (g.COT / 4), 256);
            if (!((xl <= 4 && dl <= 4) || (xl <= 7 && dl <= 2))) continue;   // staging variants that exist
            int work = cdiv(g.Wo, c) * c;
            if (work < best_work) { best_work = work; best = c; }
        }
        g.P = best;
    }
    const int P = g.P;
    g.nseg_row = cdiv(g.Wo, P);
    g.nseg_total = g.B * g.Ho * g.nseg_row;
    int64_t wsize = (int64_t)k * k * Cin * Cout;
    constexpr int target = 512;
    int want = target / pl.ntiles;                    // ~2 workgroups per CU in total
    if (want < 1) want = 1;
    int64_t cap = (48ll << 20) / wsize;               // <= 192 MiB of partials
    if (cap < 1) cap = 1;
    if (want > cap) want = (int)cap;
    if (want > g.nseg_total) want = g.nseg_total;
    g.segs_per_split = cdiv(g.nseg_total, want);
    pl.nsplit = cdiv(g.nseg_total, g.segs_per_split);
    int XW = (P - 1) * s + k;
    pl.smem = (size_t)(k * XW * g.CIT + P * g.COT) * sizeof(float);
    return 0;
}

template <int NT, int XL, int DL, int MT>
int launch_wgrad_cfg(const Plan &pl, hipStream_t st) {
    auto kern = wgrad_kernel<NT, XL, DL, MT>;
    // next to the main lane: one workgroup per CU (common.h)
    const size_t smem = yh_tls_side_lane && pl.smem < YH_SIDE_LDS_BYTES ? YH_SIDE_LDS_BYTES : pl.smem;
    if (int rc = yh_ensure_dyn_smem((const void *)kern, smem)) return rc;
    hipLaunchKernelGGL(kern, dim3(pl.nsplit, pl.ntiles), dim3(256), smem, st, pl.g);
    YH_CHECK_LAUNCH("wgrad");
    return 0;
}

template <int NT>
int launch_wgrad(const Plan &pl, hipStream_t st) {
    const Wgrad &g = pl.g;
    const int XW = (g.P - 1) * g.s + g.k;
    const int xl = cdiv(g.k * XW * (g.CIT / 4), 256), dl = cdiv(g.P * (g.COT / 4), 256);
    if (g.MT == 16) {
        if constexpr (NT <= 5) {
            if (xl <= 2 && dl <= 2) return launch_wgrad_cfg<NT, 2, 2, 16>(pl, st);
            if (xl <= 4 && dl <= 4) return launch_wgrad_cfg<NT, 4, 4, 16>(pl, st);
            if (xl <= 7 && dl <= 2) return launch_wgrad_cfg<NT, 7, 2, 16>(pl, st);
        }
    } else {
        if (xl <= 2 && dl <= 2) return launch_wgrad_cfg<NT, 2, 2, 32>(pl, st);
        if (xl <= 4 && dl <= 4) return launch_wgrad_cfg<NT, 4, 4, 32>(pl, st);
        if (xl <= 7 && dl <= 2) return launch_wgrad_cfg<NT, 7, 2, 32>(pl, st);
    }
    yh_set_error("conv_bwd_weight: staging shape (%d, %d) unsupported", xl, dl);
    return YH_E_UNSUPPORTED;
}

}  // namespace

extern "C" int64_t yh_conv_bwd_weight_ws(int B, int Hi, int Wi, int Cin, int Cout, int k, int s) {
    Plan pl;
    if (make_plan(pl, B, Hi, Wi, Cin, Cout, k, s)) return -1;
    return (int64_t)pl.nsplit * k * k * Cin * Cout;
}

extern "C" int yh_conv_bwd_weight_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *dy, int lddy, float *dw, float *ws,
                                      int64_t ws_floats, int B, int Hi, int Wi, int Cin, int cin_real, int Cout, int k, int s, void *stream);
extern "C" int yh_conv_bwd_weight(const float *x, int ldx, const float *dy, int lddy, float *dw, float *ws,
                                  int64_t ws_floats, int B, int Hi, int Wi, int Cin, int cin_real, int Cout, int k,
                                  int s, void *stream) {
    return yh_conv_bwd_weight_act(x, ldx, nullptr, 0, dy, lddy, dw, ws, ws_floats, B, Hi, Wi, Cin, cin_real, Cout, k, s, stream);
}
extern "C" int yh_conv_bwd_weight_prologue_ok(int B, int Hi, int Wi, int Cin, int Cout, int k, int s) {
    Plan pl;
    if (make_plan(pl, B, Hi, Wi, Cin, Cout, k, s)) return 0;
    const int citq = pl.g.CIT / 4;
    return (citq > 0 && 256 % citq == 0 && pl.g.CIT % 4 == 0) ? 1 : 0;       // every staging slot of a thread covers the same channel quad
}
extern "C" int yh_conv_bwd_weight_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *dy, int lddy, float *dw, float *ws,
                                      int64_t ws_floats, int B, int Hi, int Wi, int Cin, int cin_real, int Cout, int k, int s, void *stream) {
    YH_REQUIRE((k == 1 || k == 3) && (s == 1 || s == 2), "conv_bwd_weight: unsupported k=%d s=%d", k, s);
    YH_REQUIRE(!icoef || ((((uintptr_t)icoef) & 15) == 0 && icoef_ld % 4 == 0 && icoef_ld >= Cin && cin_real == Cin &&
                          yh_conv_bwd_weight_prologue_ok(B, Hi, Wi, Cin, Cout, k, s)),
               "conv_bwd_weight: the input prologue needs a 16-byte aligned table and a slab whose channel tile divides the staging plan");
    YH_REQUIRE(x && dy && dw && ws && B > 0 && Cin > 0 && Cout > 0 && cin_real > 0 && cin_real <= Cin,
               "conv_bwd_weight: bad argument");
    YH_REQUIRE(Cin % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0,
               "conv_bwd_weight: input channels / ld must be multiples of 4 and x 16-byte aligned");
    YH_REQUIRE(ldx >= Cin && lddy >= Cout, "conv_bwd_weight: ld smaller than channel count");
    Plan pl;
    int rc = make_plan(pl, B, Hi, Wi, Cin, Cout, k, s);
    if (rc) { yh_set_error("conv_bwd_weight: unsupported shape"); return rc; }
    int64_t need = (int64_t)pl.nsplit * k * k * Cin * Cout;
    if (ws_floats < need) { yh_set_error("conv_bwd_weight: workspace %lld < %lld floats", (long long)ws_floats, (long long)need); return YH_E_WORKSPACE; }
    pl.g.x = x; pl.g.dy = dy; pl.g.ws = ws; pl.g.ldx = ldx; pl.g.lddy = lddy; pl.g.icoef = icoef; pl.g.icoef_ld = icoef_ld;
    pl.g.vec_dy = (lddy % 4 == 0) && (((uintptr_t)dy & 15) == 0) && (pl.g.COT % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    switch (pl.NT) {
        case 1: rc = launch_wgrad<1>(pl, st); break;
        case 2: rc = launch_wgrad<2>(pl, st); break;
        case 3: rc = launch_wgrad<3>(pl, st); break;
        case 4: rc = launch_wgrad<4>(pl, st); break;
        case 5: rc = launch_wgrad<5>(pl, st); break;
        case 6: case 7: case 8: case 9: rc = launch_wgrad<9>(pl, st); break;
        default: yh_set_error("conv_bwd_weight: %d tiles per wave unsupported", pl.NT); return YH_E_UNSUPPORTED;
    }
    if (rc) return rc;
    int n = k * k * Cin * Cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(n, 16)), dim3(256), 0, st, ws, dw, pl.nsplit, k * k, Cin, cin_real, Cout);
    YH_CHECK_LAUNCH("wgrad_reduce");
    return 0;
}
