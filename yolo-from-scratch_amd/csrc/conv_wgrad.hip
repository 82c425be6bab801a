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

namespace {

constexpr int P = 32;   // output pixels per segment

struct Wgrad {
    const float *x, *dy;
    float *ws;
    int ldx, lddy;
    int B, Hi, Wi, Ho, Wo;
    int Cin, Cout, k, s, pad;
    int CIT, COT, n_ci_tiles;
    int nseg_row, nseg_total, segs_per_split;
    int vec_dy;
};

template <int NT>
__global__ __launch_bounds__(256) void wgrad_kernel(const Wgrad g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int KK = g.k * g.k;
    const int XW = (P - 1) * g.s + g.k;
    const int CIT = g.CIT, COT = g.COT;
    float *Xs = smem;                          // [k][XW][CIT]
    float *Ds = smem + g.k * XW * CIT;         // [P][COT]

    const int ci0 = (blockIdx.y % g.n_ci_tiles) * CIT;
    const int co0 = (blockIdx.y / g.n_ci_tiles) * COT;
    const int rows = KK * CIT;
    const int RT = (rows + 31) / 32, OT = (COT + 31) / 32;

    // tiles of this wave: tau = wave + 4u -> (rt, ot)
    int aoff[NT], boff[NT];
    bool avalid[NT], tvalid[NT];
    f32x16 acc[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        int tau = wave + 4 * u;
        tvalid[u] = tau < RT * OT;
        int rt = tvalid[u] ? tau / OT : 0, ot = tvalid[u] ? tau % OT : 0;
        int row = rt * 32 + lr;
        avalid[u] = row < rows;
        int tap = avalid[u] ? row / CIT : 0, ci = avalid[u] ? row % CIT : 0;
        aoff[u] = ((tap / g.k) * XW + (tap % g.k)) * CIT + ci;
        int col = ot * 32 + lr;
        boff[u] = col < COT ? col : 0;   // columns >= COT only exist when COT < 32; their results are never stored
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    }

    const int seg_begin = blockIdx.x * g.segs_per_split;
    int seg_end = seg_begin + g.segs_per_split;
    if (seg_end > g.nseg_total) seg_end = g.nseg_total;
    const int citq = CIT >> 2;

    for (int seg = seg_begin; seg < seg_end; ++seg) {
        const int sr = seg % g.nseg_row, rowid = seg / g.nseg_row;
        const int ho = rowid % g.Ho, b = rowid / g.Ho;
        const int w0 = sr * P;
        const int pv = (g.Wo - w0) < P ? (g.Wo - w0) : P;
        const int xcols = (pv - 1) * g.s + g.k;
        __syncthreads();   // previous segment's fragment reads are done
        // stage input halo rows (zero outside the image, beyond the segment or beyond Cin)
        for (int i = t; i < g.k * XW * citq; i += 256) {
            int c4 = i % citq, q = i / citq;
            int col = q % XW, kh = q / XW;
            int hi = ho * g.s + kh - g.pad, wi = w0 * g.s - g.pad + col;
            int c = ci0 + 4 * c4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (col < xcols && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi && c < g.Cin)
                v = *(const f32x4 *)(g.x + ((size_t)(b * g.Hi + hi) * g.Wi + wi) * g.ldx + c);
            *(f32x4 *)(Xs + (kh * XW + col) * CIT + 4 * c4) = v;
        }
        // stage the dy row segment
        if (g.vec_dy) {
            const int cotq = COT >> 2;
            for (int i = t; i < P * cotq; i += 256) {
                int c4 = i % cotq, p = i / cotq;
                int c = co0 + 4 * c4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (p < pv && c < g.Cout)
                    v = *(const f32x4 *)(g.dy + ((size_t)(b * g.Ho + ho) * g.Wo + w0 + p) * g.lddy + c);
                *(f32x4 *)(Ds + p * COT + 4 * c4) = v;
            }
        } else {
            for (int i = t; i < P * COT; i += 256) {
                int c = i % COT, p = i / COT;
                float v = 0.f;
                if (p < pv && co0 + c < g.Cout) v = g.dy[((size_t)(b * g.Ho + ho) * g.Wo + w0 + p) * g.lddy + co0 + c];
                Ds[p * COT + c] = v;
            }
        }
        __syncthreads();
        const int npairs = (pv + 1) >> 1;
        for (int pp = 0; pp < npairs; ++pp) {
            const int p = 2 * pp + lh;
            const int xo = p * g.s * CIT, dofs = p * COT;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                if (!tvalid[u]) continue;
                float a = avalid[u] ? Xs[aoff[u] + xo] : 0.f;
                float bb = Ds[boff[u] + dofs];
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[u], 0, 0, 0);
            }
        }
    }

    // write this split's partial slab: ws[split][tap][ci][co]
    float *wsp = g.ws + (size_t)blockIdx.x * KK * g.Cin * g.Cout;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        if (!tvalid[u]) continue;
        int tau = wave + 4 * u;
        int rt = tau / OT, ot = tau % OT;
        int col = ot * 32 + lr;
        if (col >= COT || co0 + col >= g.Cout) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row >= rows) continue;
            int tap = row / CIT, ci = ci0 + row % CIT;
            if (ci >= g.Cin) continue;
            wsp[((size_t)tap * g.Cin + ci) * g.Cout + co0 + col] = acc[u][r];
        }
    }
}

// dw[co][ci][tap] (ci < cin_real) = sum_split ws[split][tap][ci][co], splits added in index order
__global__ void wgrad_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dw, int nsplit, int KK, int Cin,
                                    int cin_real, int Cout) {
    const int n = KK * Cin * Cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int co = i % Cout, q = i / Cout;
        int ci = q % Cin, tap = q / Cin;
        if (ci >= cin_real) continue;
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += ws[(size_t)k * n + i];
        dw[((size_t)co * cin_real + ci) * KK + tap] = s;
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
    if (k == 1) { g.CIT = Cin >= 128 ? 128 : Cin; g.COT = Cout >= 128 ? 128 : (Cout + 3) / 4 * 4; }
    else {
        g.CIT = (s == 2) ? (Cin >= 32 ? 32 : Cin) : (Cin >= 64 ? 64 : Cin);
        g.COT = Cout >= 64 ? 64 : (Cout + 3) / 4 * 4;
    }
    if (g.CIT % 4) return YH_E_UNSUPPORTED;
    g.n_ci_tiles = cdiv(Cin, g.CIT);
    pl.ntiles = g.n_ci_tiles * cdiv(Cout, g.COT);
    int RT = cdiv(k * k * g.CIT, 32), OT = cdiv(g.COT, 32);
    pl.NT = cdiv(RT * OT, 4);
    g.nseg_row = cdiv(g.Wo, P);
    g.nseg_total = g.B * g.Ho * g.nseg_row;
    int64_t wsize = (int64_t)k * k * Cin * Cout;
    int want = 1536 / pl.ntiles;                      // ~6 workgroups per CU in total
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

template <int NT>
int launch_wgrad(const Plan &pl, hipStream_t st) {
    static size_t attr = 0;
    auto kern = wgrad_kernel<NT>;
    if (pl.smem > attr) {
        YH_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
        attr = pl.smem;
    }
    hipLaunchKernelGGL(kern, dim3(pl.nsplit, pl.ntiles), dim3(256), pl.smem, st, pl.g);
    YH_CHECK_LAUNCH("wgrad");
    return 0;
}

}  // namespace

extern "C" int64_t yh_conv_bwd_weight_ws(int B, int Hi, int Wi, int Cin, int Cout, int k, int s) {
    Plan pl;
    if (make_plan(pl, B, Hi, Wi, Cin, Cout, k, s)) return -1;
    return (int64_t)pl.nsplit * k * k * Cin * Cout;
}

extern "C" int yh_conv_bwd_weight(const float *x, int ldx, const float *dy, int lddy, float *dw, float *ws,
                                  int64_t ws_floats, int B, int Hi, int Wi, int Cin, int cin_real, int Cout, int k,
                                  int s, void *stream) {
    YH_REQUIRE((k == 1 || k == 3) && (s == 1 || s == 2), "conv_bwd_weight: unsupported k=%d s=%d", k, s);
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
    pl.g.x = x; pl.g.dy = dy; pl.g.ws = ws; pl.g.ldx = ldx; pl.g.lddy = lddy;
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
    int blocks = cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, ws, dw, pl.nsplit, k * k, Cin, cin_real, Cout);
    YH_CHECK_LAUNCH("wgrad_reduce");
    return 0;
}
