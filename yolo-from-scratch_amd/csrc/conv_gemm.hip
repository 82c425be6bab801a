// Implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (gfx950), NHWC fp32.
//
// One "gather GEMM" kernel serves the forward convolution and the backward-data convolution:
//     out[p][n] (+)= sum_{tap, c} in[gather(p, tap)][c] * w[tap][c][n]       (+ bias[n])
// p runs over an output lattice (B x Yo x Xo) that may be a strided sub-lattice of the output
// tensor (the four parity classes of a stride-2 backward-data pass), gather(p, tap) is the input
// pixel (y*sy + dy[tap], x*sx + dx[tap]) and is skipped (zero) when it falls outside the input.
//
// GEMM view: M = B*Yo*Xo pixels, N = output channels, K = taps*Cin.  A workgroup of 4 waves owns a
// BM x BN tile; K is walked in chunks of 32 staged through LDS (register-staged double buffer).
// MFMA operand maps (cdna guide section 3): A lane l = A[row l&31][k l>>5], B lane l = B[k l>>5][col l&31],
// D reg r of lane l = D[row (r&3)+8(r>>2)+4(l>>5)][col l&31]: output channels sit on lanes, so a
// store instruction writes 128 contiguous bytes per half wave and the BatchNorm column sums are
// lane-local.  The A tile is kept [pixel][k] with a 36-float row stride: a lane's four consecutive
// k values come from one conflict-free ds_read_b128 and feed four MFMAs (the k order inside a
// chunk is permuted consistently for A and B, which a sum over k does not care about).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BK = 32;
constexpr int LDA = BK + 4;

struct GatherGemm {
    const float *icoef;   // input prologue table of `in` ([scale | shift | gate] rows, icoef_ld apart; yh_prologue) or null: forward only,
    int icoef_ld;         // 16-byte staging (VEC == 4), single-source K
    const float *in, *w, *bias;
    const float *in2;     // optional second input (same geometry and ld): k >= ksplit reads in2 (pointwise K-concatenation)
    int ksplit;
    float *out, *stats;
    int Hi, Wi, ldi, Cin;
    int ldw;
    int Ho_f, Wo_f, ldo, N;
    int B, Yo, Xo, M;
    int osy, osx, ooy, oox;
    int sy, sx;
    int nTaps, Ktot;
    int accumulate, dense;
    int off32;            // every output element offset fits 31 bits (set by the launcher)
    const float *res;     // inference epilogue: + residual (after the activation), ld = ldr
    int ldr, act, up2;    // act: 1 = SiLU on (acc + bias); up2: replicate each output pixel 2x2 (nearest upsample)
    int nblk_n;
    // split-K (small-M inference layers, SK template flag): blockIdx.y = split, each split multiplies a contiguous range of
    // K chunks into its fp32 slab sk_ws[split][M][sk_ldws]; the LAST split to arrive at a tile (agent-scope ticket in
    // sk_cnt[tile], zero before the launch and reset by the reducer) adds the slabs in split order and runs the epilogue
    int sk_splits, sk_ldws;
    float *sk_ws;
    int *sk_cnt;
    unsigned cin_magic;   // floor(2^32 / Cin) + 1: k / Cin == umulhi(k, magic) for k < 2^16
    unsigned xo_magic, yo_magic;   // exact division of a pixel index < 2^31 by Xo / Yo (see fast_div)
    int xo_shift, yo_shift;
    int tap_dy[9], tap_dx[9], tap_w[9];
};

// Up to four output-parity classes of a stride-2 backward-data pass share one launch (blockIdx.y).
struct GatherGemmSet {
    GatherGemm c[4];
};

// floor(n / d) for 0 <= n < 2^31: magic = ceil(2^(31+l) / d) with l = ceil(log2 d) fits 32 bits and
// (n * magic) >> (31 + l) is exact (Granlund-Montgomery); stored shift = l - 1, shift < 0 means d == 1.
__device__ __forceinline__ int fast_div(int n, unsigned magic, int shift) {
    return shift < 0 ? n : (int)(__umulhi((unsigned)n, magic) >> shift);
}

template <int MT> struct Mfma;
template <> struct Mfma<32> {
    typedef f32x16 Acc;
    static constexpr int NR = 16;
    static __device__ __forceinline__ Acc run(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }
};
template <> struct Mfma<16> {
    typedef f32x4 Acc;
    static constexpr int NR = 4;
    static __device__ __forceinline__ Acc run(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int r, int lh) { return 4 * lh + r; }
};

template <int BM, int BN, int WM, int WN, int VEC, int NCLS, int MT, int SK = 0>
__global__ __launch_bounds__(256) void gather_gemm_kernel(const GatherGemmSet gs) {
    static_assert(!SK || NCLS == 1, "split-K launches carry one problem");
    const GatherGemm &g = gs.c[(NCLS == 1 || SK) ? 0 : blockIdx.y];
    typedef Mfma<MT> MF;
    typedef typename MF::Acc Acc;
    constexpr int LG = 64 / MT;            // lane groups = physical k values one MFMA consumes per lane slot
    constexpr int KSTEP = 4 * LG;          // k consumed per (ds_read_b128 -> 4 MFMAs) step
    constexpr int TM = BM / WM / MT, TN = BN / WN / MT;
    constexpr int AROWS = BM / 32;
    constexpr int BQ = BN / 4;
    constexpr int BROWS = 256 / BQ;
    constexpr int BPASS = (BK + BROWS - 1) / BROWS;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && BK % KSTEP == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                          // [2][BM][LDA]
    float *Bs = smem + 2 * BM * LDA;           // [2][BK][BN]
    int *tapt = (int *)(Bs + 2 * BK * BN);     // [3][9] (+ the split-K "last arriver" flag at [27])

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // Workgroups are dealt round-robin over the 8 XCDs (private L2 each): remap the linear id so that
    // every XCD walks a contiguous range of tiles with the n-tiles of one m-tile adjacent -- the A
    // tile re-read by the next n-tile and the 3x3 halo shared with the next m-tile then hit that L2.
    // Bijective for any grid size (guide section 5, "XCD swizzle must be bijective").
    const int nwg = ((g.M + BM - 1) / BM) * g.nblk_n, orig = blockIdx.x;
    if (orig >= nwg) return;   // classes of one launch have slightly different tile counts
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int mblk = tile / g.nblk_n, nblk = tile - mblk * g.nblk_n;
    const int m0 = mblk * BM, n0 = nblk * BN;
    if (t < 9) {
        tapt[t] = g.tap_dy[t];
        tapt[9 + t] = g.tap_dx[t];
        tapt[18 + t] = g.tap_w[t];
    }

    // per-thread A rows: r = (t>>3) + 32*i, float4 column kq = t&7.  roff = element offset of the
    // row's centre pixel; a tap adds the block-uniform-or-per-thread delta (dy*Wi + dx)*ldi.
    const int kq = t & 7;
    int roff[AROWS], riy[AROWS], rix[AROWS];
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        int m = m0 + (t >> 3) + 32 * i;
        if (m < g.M) {
            int q = fast_div(m, g.xo_magic, g.xo_shift), x = m - q * g.Xo;
            int b = fast_div(q, g.yo_magic, g.yo_shift), y = q - b * g.Yo;
            riy[i] = y * g.sy;
            rix[i] = x * g.sx;
            roff[i] = ((b * g.Hi + riy[i]) * g.Wi + rix[i]) * g.ldi;
        } else {
            roff[i] = 0;
            riy[i] = -(1 << 20);
            rix[i] = 0;
        }
    }
    __syncthreads();

    // two register sets: chunk c+1 and chunk c+2 are both in flight while chunk c is multiplied
    f32x4 raA[AROWS], rbA[BPASS], raB[AROWS], rbB[BPASS];
    // input prologue: a chunk's coefficients (the thread's channel quad) are loaded with the chunk, the activation is applied when the
    // chunk is parked in LDS; padding / rows past M stay zero: one validity bit per row and register set
    const bool act = VEC == 4 && g.icoef != nullptr;       // workgroup-uniform
    f32x4 pcA[3], pcB[3];
    unsigned okA = 0, okB = 0;
    auto load_tiles = [&](int c, f32x4 (&ra)[AROWS], f32x4 (&rb)[BPASS], f32x4 (&pc)[3], unsigned &okm) {
        const int k = c * BK + 4 * kq;
        okm = 0;
        if (VEC == 4) {
            if (k < g.Ktot) {
                int tap = (int)__umulhi((unsigned)k, g.cin_magic), ci = k - tap * g.Cin;
                if (act) {
                    pc[0] = *(const f32x4 *)(g.icoef + ci); pc[1] = *(const f32x4 *)(g.icoef + g.icoef_ld + ci);
                    pc[2] = *(const f32x4 *)(g.icoef + 2 * g.icoef_ld + ci);
                }
                int dy = tapt[tap], dx = tapt[9 + tap];
                int toff = (dy * g.Wi + dx) * g.ldi + ci;
                const float *inb = (g.in2 && k >= g.ksplit) ? g.in2 - g.ksplit : g.in;   // two-source K (pointwise only)
#pragma unroll
                for (int i = 0; i < AROWS; ++i) {
                    bool ok = (unsigned)(riy[i] + dy) < (unsigned)g.Hi && (unsigned)(rix[i] + dx) < (unsigned)g.Wi;
                    ra[i] = ok ? *(const f32x4 *)(inb + (roff[i] + toff)) : f32x4{0.f, 0.f, 0.f, 0.f};
                    okm |= (unsigned)ok << i;
                }
            } else {
#pragma unroll
                for (int i = 0; i < AROWS; ++i) ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int i = 0; i < AROWS; ++i) ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int ke = k + e;
                if (ke < g.Ktot) {
                    int tap = (int)__umulhi((unsigned)ke, g.cin_magic), ci = ke - tap * g.Cin;
                    int dy = tapt[tap], dx = tapt[9 + tap];
                    int toff = (dy * g.Wi + dx) * g.ldi + ci;
                    const float *inb = (g.in2 && ke >= g.ksplit) ? g.in2 - g.ksplit : g.in;
#pragma unroll
                    for (int i = 0; i < AROWS; ++i) {
                        bool ok = (unsigned)(riy[i] + dy) < (unsigned)g.Hi && (unsigned)(rix[i] + dx) < (unsigned)g.Wi;
                        if (ok) ra[i][e] = inb[roff[i] + toff];
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            int kr = c * BK + t / BQ + p * BROWS;
            int n = n0 + 4 * (t % BQ);
            if (t / BQ + p * BROWS < BK && kr < g.Ktot && n < g.ldw) {
                int tap = (int)__umulhi((unsigned)kr, g.cin_magic), ci = kr - tap * g.Cin;
                rb[p] = *(const f32x4 *)(g.w + ((tapt[18 + tap] * g.Cin + ci) * g.ldw + n));
            } else {
                rb[p] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_tiles = [&](int buf, f32x4 (&ra)[AROWS], f32x4 (&rb)[BPASS], const f32x4 (&pc)[3], unsigned okm) {
        float *a = As + buf * BM * LDA;
        float *b = Bs + buf * BK * BN;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            f32x4 v = ra[i];
            if (act && (okm >> i & 1)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = yh_prologue(v[e], pc[0][e], pc[1][e], pc[2][e]);
            }
            *(f32x4 *)(a + ((t >> 3) + 32 * i) * LDA + 4 * kq) = v;
        }
#pragma unroll
        for (int p = 0; p < BPASS; ++p)
            if (t / BQ + p * BROWS < BK) *(f32x4 *)(b + (t / BQ + p * BROWS) * BN + 4 * (t % BQ)) = rb[p];
    };

    Acc acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < MF::NR; ++r) acc[i][j][r] = 0.f;

    const int nch_all = (g.Ktot + BK - 1) / BK;
    // split-K: this workgroup owns chunks [cbeg, cbeg + nchunks) of the K axis
    const int cbeg = SK ? (int)(((long long)nch_all * blockIdx.y) / g.sk_splits) : 0;
    const int nchunks = SK ? (int)(((long long)nch_all * (blockIdx.y + 1)) / g.sk_splits) - cbeg : nch_all;
    const int lr = lane & (MT - 1), lh = lane / MT;
    auto compute = [&](int buf, int kvalid) {
        const float *a = As + buf * BM * LDA + (wm * TM * MT + lr) * LDA + 4 * lh;
        const float *b = Bs + buf * BK * BN + (4 * lh) * BN + wn * TN * MT + lr;
#pragma unroll
        for (int kk = 0; kk < BK; kk += KSTEP) {
            if (kk >= kvalid) break;               // the last chunk of a short K skips its empty k-steps
            f32x4 av[TM];
            float bv[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *(const f32x4 *)(a + i * MT * LDA + kk);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[j][e] = b[(kk + e) * BN + j * MT];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = MF::run(av[i][e], bv[j][e], acc[i][j]);
        }
    };
    // software pipeline, prefetch distance 2: at step c the loads of chunk c+2 are issued, chunk c is
    // multiplied from LDS, then chunk c+1 (loaded one step earlier) is written to the other LDS buffer.
    load_tiles(cbeg, raA, rbA, pcA, okA);
    store_tiles(0, raA, rbA, pcA, okA);
    if (nchunks > 1) load_tiles(cbeg + 1, raB, rbB, pcB, okB);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
        if (c + 2 < nchunks) load_tiles(cbeg + c + 2, raA, rbA, pcA, okA);
        compute(0, g.Ktot - (cbeg + c) * BK);
        if (c + 1 < nchunks) store_tiles(1, raB, rbB, pcB, okB);
        __syncthreads();
        if (c + 1 >= nchunks) break;
        if (c + 3 < nchunks) load_tiles(cbeg + c + 3, raB, rbB, pcB, okB);
        compute(1, g.Ktot - (cbeg + c + 1) * BK);
        if (c + 2 < nchunks) store_tiles(0, raA, rbA, pcA, okA);
        __syncthreads();
    }

    if (SK) {
        // publish this split's partial tile, then the last arriver of the tile reduces (cdna guide section 5, "in-launch
        // split-K reduction": plain slab stores, vmcnt drain, barrier, one agent-scope release + relaxed ticket; the
        // reducer takes one agent-scope acquire and reads every slab with plain loads in split order -- deterministic)
        float *slab = g.sk_ws + (size_t)blockIdx.y * g.M * g.sk_ldws;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < MF::NR; ++r) {
                const int m = m0 + wm * TM * MT + i * MT + MF::row(r, lh);
                if (m >= g.M) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * TN * MT + j * MT + lr;
                    if (n < g.N) slab[(size_t)m * g.sk_ldws + n] = acc[i][j][r];
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int ticket = __hip_atomic_fetch_add(&g.sk_cnt[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = ticket == g.sk_splits - 1;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&g.sk_cnt[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
            tapt[27] = last;
        }
        __syncthreads();
        if (!tapt[27]) return;
        // reducer: float4 pieces of the tile, all slabs of a piece loaded back to back (independent loads), summed in split
        // order, then bias / SiLU / residual / x2 upsample and the store -- the whole epilogue of the inference form
        constexpr int PCS = BN / 4;
        const int S = g.sk_splits;
        for (int pc = t; pc < BM * PCS; pc += 256) {
            const int rl = pc / PCS, n = n0 + 4 * (pc - rl * PCS);
            const int m = m0 + rl;
            if (m >= g.M || n >= g.N) continue;
            const float *src = g.sk_ws + (size_t)m * g.sk_ldws + n;
            const size_t sstride = (size_t)g.M * g.sk_ldws;
            f32x4 v = *(const f32x4 *)src;
            int sp = 1;
            for (; sp + 3 < S; sp += 4) {
                f32x4 a0 = *(const f32x4 *)(src + sp * sstride), a1 = *(const f32x4 *)(src + (sp + 1) * sstride);
                f32x4 a2 = *(const f32x4 *)(src + (sp + 2) * sstride), a3 = *(const f32x4 *)(src + (sp + 3) * sstride);
                v += a0; v += a1; v += a2; v += a3;
            }
            for (; sp < S; ++sp) v += *(const f32x4 *)(src + sp * sstride);
            size_t opix = (size_t)m;
            if (g.up2) {
                int q = fast_div(m, g.xo_magic, g.xo_shift), x = m - q * g.Xo;
                int b = fast_div(q, g.yo_magic, g.yo_shift), y = q - b * g.Yo;
                opix = ((size_t)b * (2 * g.Ho_f) + 2 * y) * (2 * g.Wo_f) + 2 * x;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e >= g.N) break;
                float o = v[e] + (g.bias ? g.bias[n + e] : 0.f);
                if (g.act) o = o * yh_sigmoid(o);
                if (g.res) o += g.res[(size_t)m * g.ldr + n + e];
                float *dst = g.out + opix * g.ldo + n + e;
                dst[0] = o;
                if (g.up2) {
                    const size_t rs = (size_t)(2 * g.Wo_f) * g.ldo;
                    dst[g.ldo] = o; dst[rs] = o; dst[rs + g.ldo] = o;
                }
            }
        }
        return;
    }

    // ---- epilogue: bias, optional accumulate, store, optional BatchNorm partial sums ----------
    float csum[TN], csq[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) csum[j] = csq[j] = 0.f;
    float bias_v[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int n = n0 + wn * TN * MT + j * MT + lr;
        bias_v[j] = (g.bias && n < g.N) ? g.bias[n] : 0.f;
    }
    if (MT == 32 && !(g.act | g.up2 | (g.res != nullptr)) && g.off32) {
        // training form.  The output pixel of each of the tile's BM rows is computed ONCE (one thread per row, two divisions for
        // a strided output lattice) and parked in LDS; a lane then reads its 16 TM rows' offsets with 16-byte LDS reads instead
        // of dividing 16 TM times, and a whole tile (every row and column valid) stores without per-element tests -- on the
        // one-tap parity class of a stride-2 backward-data pass the tested epilogue cost more than the 32-MFMA main loop.
        int *orow = (int *)(smem + 2 * WM * BN);                          // behind the partial-sum exchange area
        if (t < BM) {
            const int m = m0 + t;
            int op = -1;
            if (m < g.M) {
                if (g.dense) {
                    op = m;
                } else {
                    int q = fast_div(m, g.xo_magic, g.xo_shift), x = m - q * g.Xo;
                    int b = fast_div(q, g.yo_magic, g.yo_shift), y = q - b * g.Yo;
                    op = (b * g.Ho_f + (y * g.osy + g.ooy)) * g.Wo_f + (x * g.osx + g.oox);
                }
            }
            orow[t] = op;
        }
        __syncthreads();
        const bool whole = m0 + BM <= g.M && n0 + BN <= g.N;              // workgroup-uniform
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int op[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const i32x4 v = *(const i32x4 *)(orow + wm * TM * MT + i * MT + 8 * q + 4 * lh);
                op[4 * q] = v[0]; op[4 * q + 1] = v[1]; op[4 * q + 2] = v[2]; op[4 * q + 3] = v[3];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * TN * MT + j * MT + lr;
                if (whole) {
                    if (!g.accumulate) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[i][j][r] + bias_v[j];
                            g.out[(unsigned)(op[r] * g.ldo + n)] = v;
                            csum[j] += v;
                            csq[j] += v * v;
                        }
                    } else {
                        float old[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) old[r] = g.out[(unsigned)(op[r] * g.ldo + n)];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[i][j][r] + bias_v[j] + old[r];
                            g.out[(unsigned)(op[r] * g.ldo + n)] = v;
                            csum[j] += v;
                            csq[j] += v * v;
                        }
                    }
                } else if (n < g.N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (op[r] >= 0) {
                            float *o = g.out + (unsigned)(op[r] * g.ldo + n);
                            float v = acc[i][j][r] + bias_v[j];
                            if (g.accumulate) v += *o;
                            *o = v;
                            csum[j] += v;
                            csq[j] += v * v;
                        }
                }
            }
        }
        __syncthreads();                                                   // orow is consumed before the partial sums reuse LDS
    } else
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < MF::NR; ++r) {
            int m = m0 + wm * TM * MT + i * MT + MF::row(r, lh);
            if (m >= g.M) continue;
            size_t opix;
            if (g.dense) {
                opix = (size_t)m;
            } else {
                int q = fast_div(m, g.xo_magic, g.xo_shift), x = m - q * g.Xo;
                int b = fast_div(q, g.yo_magic, g.yo_shift), y = q - b * g.Yo;
                opix = ((size_t)b * g.Ho_f + (y * g.osy + g.ooy)) * g.Wo_f + (x * g.osx + g.oox);
            }
            size_t opix_up = 0;
            if (g.up2) {   // output tensor is (B, 2*Ho_f, 2*Wo_f): top-left pixel of the 2x2 replica
                int q = fast_div(m, g.xo_magic, g.xo_shift), x = m - q * g.Xo;
                int b = fast_div(q, g.yo_magic, g.yo_shift), y = q - b * g.Yo;
                opix_up = ((size_t)b * (2 * g.Ho_f) + 2 * y) * (2 * g.Wo_f) + 2 * x;
            }
            float *orow = g.out + (g.up2 ? opix_up : opix) * g.ldo;
            const float *rrow = g.res ? g.res + opix * g.ldr : nullptr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int n = n0 + wn * TN * MT + j * MT + lr;
                if (n < g.N) {
                    float v = acc[i][j][r] + bias_v[j];
                    if (g.act) v = v * yh_sigmoid(v);
                    if (rrow) v += rrow[n];
                    if (g.accumulate) v += orow[n];
                    orow[n] = v;
                    if (g.up2) {
                        const size_t rs = (size_t)(2 * g.Wo_f) * g.ldo;
                        orow[g.ldo + n] = v; orow[rs + n] = v; orow[rs + g.ldo + n] = v;
                    }
                    csum[j] += v;
                    csq[j] += v * v;
                }
            }
        }
    }
    if (g.stats) {
        float *red = smem;   // [WM][BN][2]; the main loop ended with a barrier
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = csum[j], q = csq[j];
#pragma unroll
            for (int o = 32; o >= MT; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }   // lanes sharing a column
            if (lh == 0) {
                int col = wn * TN * MT + j * MT + lr;
                red[(wm * BN + col) * 2 + 0] = s;
                red[(wm * BN + col) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (t < BN && n0 + t < g.N) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                s += red[(w * BN + t) * 2 + 0];
                q += red[(w * BN + t) * 2 + 1];
            }
            g.stats[((size_t)mblk * 2 + 0) * g.N + n0 + t] = s;
            g.stats[((size_t)mblk * 2 + 1) * g.N + n0 + t] = q;
        }
    }
}

void set_magic(unsigned d, unsigned &magic, int &shift) {
    int l = 0;
    while ((1u << l) < d) ++l;
    magic = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
    shift = l - 1;
}

template <int BM, int BN, int WM, int WN, int VEC, int NCLS, int MT, int SK = 0>
int launch_cfg(GatherGemmSet gs, hipStream_t st) {
    constexpr size_t smem = (size_t)(2 * BM * LDA + 2 * BK * BN) * sizeof(float) + 28 * sizeof(int);
    auto kern = gather_gemm_kernel<BM, BN, WM, WN, VEC, NCLS, MT, SK>;
    if (int rc = yh_ensure_dyn_smem((const void *)kern, smem)) return rc;
    int maxblk = 0;
    for (int c = 0; c < NCLS; ++c) {
        GatherGemm &g = gs.c[c];
        g.nblk_n = cdiv(g.N, BN);
        g.cin_magic = (unsigned)((1ull << 32) / (unsigned)g.Cin) + 1u;
        g.off32 = (int64_t)g.B * g.Ho_f * g.Wo_f * g.ldo < (1ll << 31) ? 1 : 0;
        int blk = cdiv(g.M, BM) * g.nblk_n;
        if (blk > maxblk) maxblk = blk;
    }
    hipLaunchKernelGGL(kern, dim3(maxblk, SK ? gs.c[0].sk_splits : NCLS), dim3(256), smem, st, gs);
    YH_CHECK_LAUNCH("gather_gemm");
    return 0;
}

// Tile choice.  The MFMA pipe is the bound for most layers, so what matters is (a) not padding N
// (BN = 32/64/128 by channel count) and (b) wave quantisation: a layer with few tiles leaves CUs idle
// in its last round, so small-M layers take the 64-row tile.
int pick_bm(int M, int nblk_n) {
    constexpr int force = 0;
    if (force == 64 || force == 128) return force;
    // measured on MI355X (tools/layer_bench.py): 64-row tiles win whenever the 128-row grid is small,
    // except when it fills the chip exactly once at two workgroups per CU (256 < blocks <= 512)
    const int blocks128 = cdiv(M, 128) * nblk_n;
    if (blocks128 >= 8 * 256) return 128;
    return (blocks128 > 256 && blocks128 <= 512) ? 128 : 64;
}

int stats_bm(int M, int N) {   // rows per BatchNorm partial-sum block = the BM the forward launch will use
    int bn = N <= 32 ? 32 : (N <= 64 ? 64 : 128);
    return bn <= 32 ? 128 : pick_bm(M, cdiv(N, bn));
}

template <int NCLS>
int launch_set(GatherGemmSet &gs, hipStream_t st) {
    for (int c = 0; c < NCLS; ++c) {
        GatherGemm &g = gs.c[c];
        YH_REQUIRE(g.M > 0 && g.N > 0 && g.Ktot > 0, "gather_gemm: empty problem M=%d N=%d K=%d", g.M, g.N, g.Ktot);
        YH_REQUIRE(g.ldw % 4 == 0 && g.ldw >= g.N, "gather_gemm: weight row stride %d must be a multiple of 4 and >= N=%d",
                   g.ldw, g.N);
        YH_REQUIRE(((uintptr_t)g.w & 15) == 0, "gather_gemm: weights must be 16-byte aligned");
        YH_REQUIRE(g.Ktot < 65536 && (int64_t)g.B * g.Hi * g.Wi * g.ldi < (1ll << 31) && (int64_t)g.Ktot * g.ldw < (1ll << 31),
                   "gather_gemm: problem exceeds the 32-bit element-offset range");
        set_magic((unsigned)g.Xo, g.xo_magic, g.xo_shift);
        set_magic((unsigned)g.Yo, g.yo_magic, g.yo_shift);
    }
    const GatherGemm &g = gs.c[0];
    const bool vec = (g.Cin % 4 == 0) && (g.ldi % 4 == 0) && (((uintptr_t)g.in & 15) == 0);
    const int bn = g.N <= 16 ? 16 : (g.N <= 32 ? 32 : (g.N <= 64 ? 64 : 128));
    int mtot = 0;
    for (int c = 0; c < NCLS; ++c) mtot += gs.c[c].M;
    const int bm = bn <= 32 ? 128 : pick_bm(mtot, cdiv(g.N, bn));
#define YH_CFG(BM_, BN_, WM_, WN_) \
    (vec ? launch_cfg<BM_, BN_, WM_, WN_, 4, NCLS, 32>(gs, st) : launch_cfg<BM_, BN_, WM_, WN_, 1, NCLS, 32>(gs, st))
    // N <= 16: the 16x16x4 MFMA shape (same FLOP rate, no padding of the channel axis to 32)
    if (bn == 16) return vec ? launch_cfg<128, 16, 4, 1, 4, NCLS, 16>(gs, st) : launch_cfg<128, 16, 4, 1, 1, NCLS, 16>(gs, st);
    if (bm == 128) {
        if (bn == 32) return YH_CFG(128, 32, 4, 1);
        if (bn == 64) return YH_CFG(128, 64, 2, 2);
        return YH_CFG(128, 128, 2, 2);
    }
    if (bn == 64) return YH_CFG(64, 64, 2, 2);
    return YH_CFG(64, 128, 2, 2);
#undef YH_CFG
}

int launch(const GatherGemm &g, hipStream_t st) {
    GatherGemmSet gs{};
    gs.c[0] = g;
    return launch_set<1>(gs, st);
}

}  // namespace

extern "C" int yh_conv_fwd_blocks(int B, int Hi, int Wi, int Cout, int k, int s) {
    int p = k / 2, Ho = (Hi + 2 * p - k) / s + 1, Wo = (Wi + 2 * p - k) / s + 1;
    int M = B * Ho * Wo;
    return cdiv(M, stats_bm(M, Cout));
}

extern "C" int yh_conv_fwd_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *wf, int ldwf, const float *bias, float *y,
                               int ldy, float *bn_partials, int B, int Hi, int Wi, int Cin, int Cout, int k, int s, void *stream);
extern "C" int yh_conv_fwd(const float *x, int ldx, const float *wf, int ldwf, const float *bias, float *y, int ldy,
                           float *bn_partials, int B, int Hi, int Wi, int Cin, int Cout, int k, int s,
                           void *stream) {
    return yh_conv_fwd_act(x, ldx, nullptr, 0, wf, ldwf, bias, y, ldy, bn_partials, B, Hi, Wi, Cin, Cout, k, s, stream);
}
extern "C" int yh_conv_fwd_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *wf, int ldwf, const float *bias, float *y,
                               int ldy, float *bn_partials, int B, int Hi, int Wi, int Cin, int Cout, int k, int s, void *stream) {
    YH_REQUIRE((k == 1 || k == 3) && (s == 1 || s == 2), "conv_fwd: unsupported k=%d s=%d", k, s);
    YH_REQUIRE(!icoef || ((((uintptr_t)icoef) & 15) == 0 && icoef_ld % 4 == 0 && icoef_ld >= Cin && Cin % 4 == 0 && ldx % 4 == 0 &&
                          (((uintptr_t)x) & 15) == 0),
               "conv_fwd: the input prologue needs 16-byte addressable input rows and table");
    YH_REQUIRE(x && wf && y && B > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "conv_fwd: bad argument");
    YH_REQUIRE(ldx >= Cin && ldy >= Cout, "conv_fwd: ld smaller than channel count");
    YH_REQUIRE((int64_t)B * Hi * Wi * (int64_t)ldx < (1ll << 31), "conv_fwd: input too large for 32-bit pixel index");
    GatherGemm g{};
    const int p = k / 2;
    g.icoef = icoef; g.icoef_ld = icoef_ld;
    g.in = x; g.w = wf; g.bias = bias; g.out = y; g.stats = bn_partials;
    g.Hi = Hi; g.Wi = Wi; g.ldi = ldx; g.Cin = Cin; g.ldw = ldwf;
    g.Ho_f = (Hi + 2 * p - k) / s + 1; g.Wo_f = (Wi + 2 * p - k) / s + 1; g.ldo = ldy; g.N = Cout;
    g.B = B; g.Yo = g.Ho_f; g.Xo = g.Wo_f; g.M = B * g.Yo * g.Xo;
    g.osy = g.osx = 1; g.ooy = g.oox = 0; g.sy = g.sx = s;
    g.nTaps = k * k; g.Ktot = g.nTaps * Cin; g.accumulate = 0; g.dense = 1;
    for (int kh = 0; kh < k; ++kh)
        for (int kw = 0; kw < k; ++kw) {
            int t = kh * k + kw;
            g.tap_dy[t] = kh - p; g.tap_dx[t] = kw - p; g.tap_w[t] = t;
        }
    return launch(g, (hipStream_t)stream);
}

extern "C" int yh_conv_fwd_fused(const float *x, int ldx, const float *wf, int ldwf, const float *bias, const float *res,
                                 int ldr, float *y, int ldy, int B, int Hi, int Wi, int Cin, int Cout, int k, int s,
                                 int act_silu, int upsample, void *stream) {
    YH_REQUIRE((k == 1 || k == 3) && (s == 1 || s == 2), "conv_fwd_fused: unsupported k=%d s=%d", k, s);
    YH_REQUIRE(x && wf && y && B > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "conv_fwd_fused: bad argument");
    YH_REQUIRE(ldx >= Cin && ldy >= Cout && (!res || ldr >= Cout), "conv_fwd_fused: ld smaller than channel count");
    GatherGemm g{};
    const int p = k / 2;
    g.in = x; g.w = wf; g.bias = bias; g.out = y; g.stats = nullptr;
    g.Hi = Hi; g.Wi = Wi; g.ldi = ldx; g.Cin = Cin; g.ldw = ldwf;
    g.Ho_f = (Hi + 2 * p - k) / s + 1; g.Wo_f = (Wi + 2 * p - k) / s + 1; g.ldo = ldy; g.N = Cout;
    g.B = B; g.Yo = g.Ho_f; g.Xo = g.Wo_f; g.M = B * g.Yo * g.Xo;
    g.osy = g.osx = 1; g.ooy = g.oox = 0; g.sy = g.sx = s;
    g.nTaps = k * k; g.Ktot = g.nTaps * Cin; g.accumulate = 0; g.dense = 1;
    g.res = res; g.ldr = ldr; g.act = act_silu ? 1 : 0; g.up2 = upsample ? 1 : 0;
    for (int kh = 0; kh < k; ++kh)
        for (int kw = 0; kw < k; ++kw) {
            int t = kh * k + kw;
            g.tap_dy[t] = kh - p; g.tap_dx[t] = kw - p; g.tap_w[t] = t;
        }
    return launch(g, (hipStream_t)stream);
}

// ---- small-M inference: split-K inside one launch ------------------------------------------------------------------------
// At batch 1 a layer has few output tiles (7 x 4 workgroups for 256 -> 256 at 20x20) and a long K loop (72 chunks): the
// layer's latency is one workgroup's K loop on a mostly idle chip.  The K axis (taps x channels) is cut into S contiguous
// chunk ranges, one workgroup per (tile, range); partial tiles go to fp32 slabs and the last workgroup to arrive at a tile
// (an agent-scope ticket) adds them in range order and applies bias / SiLU / residual / x2 upsample -- no second launch,
// bitwise reproducible.  ws layout: [S][M][ldws] floats, then one int32 ticket per tile (zero between launches).
namespace {
struct SkPlan {
    int bm, bn, tiles, splits, ldws;
};
SkPlan sk_plan(int B, int Hi, int Wi, int Cin, int Cout, int k, int s) {
    SkPlan p{};
    const int pad = k / 2, Ho = (Hi + 2 * pad - k) / s + 1, Wo = (Wi + 2 * pad - k) / s + 1;
    const int64_t M = (int64_t)B * Ho * Wo;
    p.bn = Cout <= 16 ? 16 : (Cout <= 32 ? 32 : 64);
    p.bm = p.bn == 64 ? 64 : 128;
    p.tiles = (int)(((M + p.bm - 1) / p.bm) * ((Cout + p.bn - 1) / p.bn));
    p.ldws = (Cout + 3) / 4 * 4;
    const int nch = (k * k * Cin + BK - 1) / BK;
#ifdef YH_WGS_TUNE
    static const int target = getenv("YH_SK_TARGET") ? atoi(getenv("YH_SK_TARGET")) : 192;
    static const int minch = getenv("YH_SK_MINCH") ? atoi(getenv("YH_SK_MINCH")) : 2;
#else
    constexpr int target = 192;      // workgroups per layer
    constexpr int minch = 2;         // at least two K chunks per split
#endif
    int S = target / p.tiles;
    if (S > nch / minch) S = nch / minch;
    if (S > 32) S = 32;
    p.splits = (M > 65536 || S < 2) ? 1 : S;
    return p;
}
}  // namespace

extern "C" int64_t yh_conv_fwd_fused_ws(int B, int Hi, int Wi, int Cin, int Cout, int k, int s) {
    const SkPlan p = sk_plan(B, Hi, Wi, Cin, Cout, k, s);
    if (p.splits == 1) return 0;
    const int pad = k / 2, Ho = (Hi + 2 * pad - k) / s + 1, Wo = (Wi + 2 * pad - k) / s + 1;
    return (int64_t)p.splits * B * Ho * Wo * p.ldws + p.tiles;      // slabs + tickets (int32, one float slot each)
}

extern "C" int yh_conv_fwd_fused_splitk(const float *x, int ldx, const float *wf, int ldwf, const float *bias, const float *res,
                                        int ldr, float *y, int ldy, float *ws, int64_t ws_floats, int B, int Hi, int Wi, int Cin,
                                        int Cout, int k, int s, int act_silu, int upsample, void *stream) {
    const SkPlan p = sk_plan(B, Hi, Wi, Cin, Cout, k, s);
    const bool vec = (Cin % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0);
    if (p.splits == 1 || !ws || !vec)
        return yh_conv_fwd_fused(x, ldx, wf, ldwf, bias, res, ldr, y, ldy, B, Hi, Wi, Cin, Cout, k, s, act_silu, upsample, stream);
    YH_REQUIRE((k == 1 || k == 3) && (s == 1 || s == 2), "conv_fwd_fused_splitk: unsupported k=%d s=%d", k, s);
    YH_REQUIRE(x && wf && y && B > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "conv_fwd_fused_splitk: bad argument");
    YH_REQUIRE(ldx >= Cin && ldy >= Cout && (!res || ldr >= Cout), "conv_fwd_fused_splitk: ld smaller than channel count");
    YH_REQUIRE(ws_floats >= yh_conv_fwd_fused_ws(B, Hi, Wi, Cin, Cout, k, s), "conv_fwd_fused_splitk: workspace too small");
    const int pad = k / 2, Ho = (Hi + 2 * pad - k) / s + 1, Wo = (Wi + 2 * pad - k) / s + 1;
    const int64_t M = (int64_t)B * Ho * Wo;
    GatherGemm g{};
    g.in = x; g.w = wf; g.bias = bias; g.out = y; g.stats = nullptr;
    g.Hi = Hi; g.Wi = Wi; g.ldi = ldx; g.Cin = Cin; g.ldw = ldwf;
    g.Ho_f = Ho; g.Wo_f = Wo; g.ldo = ldy; g.N = Cout;
    g.B = B; g.Yo = Ho; g.Xo = Wo; g.M = (int)M;
    g.osy = g.osx = 1; g.ooy = g.oox = 0; g.sy = g.sx = s;
    g.nTaps = k * k; g.Ktot = g.nTaps * Cin; g.accumulate = 0; g.dense = 1;
    g.res = res; g.ldr = ldr; g.act = act_silu ? 1 : 0; g.up2 = upsample ? 1 : 0;
    for (int kh = 0; kh < k; ++kh)
        for (int kw = 0; kw < k; ++kw) {
            int t = kh * k + kw;
            g.tap_dy[t] = kh - pad; g.tap_dx[t] = kw - pad; g.tap_w[t] = t;
        }
    g.sk_splits = p.splits; g.sk_ldws = p.ldws; g.sk_ws = ws;
    g.sk_cnt = (int *)(ws + (size_t)p.splits * M * p.ldws);
    YH_REQUIRE(ldwf % 4 == 0 && ldwf >= Cout && ((uintptr_t)wf & 15) == 0, "conv_fwd_fused_splitk: weight pack misaligned");
    YH_REQUIRE(g.Ktot < 65536 && (int64_t)B * Hi * Wi * ldx < (1ll << 31) && (int64_t)g.Ktot * ldwf < (1ll << 31),
               "conv_fwd_fused_splitk: problem exceeds the 32-bit element-offset range");
    set_magic((unsigned)g.Xo, g.xo_magic, g.xo_shift);
    set_magic((unsigned)g.Yo, g.yo_magic, g.yo_shift);
    GatherGemmSet gs{};
    gs.c[0] = g;
    hipStream_t st = (hipStream_t)stream;
    if (p.bn == 16) return launch_cfg<128, 16, 4, 1, 4, 1, 16, 1>(gs, st);
    if (p.bn == 32) return launch_cfg<128, 32, 4, 1, 4, 1, 32, 1>(gs, st);
    return launch_cfg<64, 64, 2, 2, 4, 1, 32, 1>(gs, st);
}

extern "C" int yh_conv_bwd_data(const float *dy, int lddy, const float *wb, int ldwb, float *dx, int lddx, int B,
                                int Hi, int Wi, int Cin, int Cout, int k, int s, int accumulate, void *stream) {
    YH_REQUIRE((k == 1 || k == 3) && (s == 1 || s == 2), "conv_bwd_data: unsupported k=%d s=%d", k, s);
    YH_REQUIRE(dy && wb && dx && B > 0 && Cin > 0 && Cout > 0, "conv_bwd_data: bad argument");
    YH_REQUIRE(lddy >= Cout && lddx >= Cin, "conv_bwd_data: ld smaller than channel count");
    const int p = k / 2, Ho = (Hi + 2 * p - k) / s + 1, Wo = (Wi + 2 * p - k) / s + 1;
    YH_REQUIRE(!(k == 1 && s == 2), "conv_bwd_data: 1x1 stride-2 is not used by this network");
    // one workgroup set per residue class of the input pixel modulo the stride; the (up to four)
    // classes of a stride-2 pass go out as ONE launch (blockIdx.y = class)
    GatherGemmSet gs{};
    int ncls = 0;
    for (int ph = 0; ph < s; ++ph)
        for (int pw = 0; pw < s; ++pw) {
            GatherGemm g{};
            g.in = dy; g.w = wb; g.bias = nullptr; g.out = dx; g.stats = nullptr;
            g.Hi = Ho; g.Wi = Wo; g.ldi = lddy; g.Cin = Cout; g.ldw = ldwb;
            g.Ho_f = Hi; g.Wo_f = Wi; g.ldo = lddx; g.N = Cin;
            g.B = B; g.Yo = (Hi - ph + s - 1) / s; g.Xo = (Wi - pw + s - 1) / s; g.M = B * g.Yo * g.Xo;
            g.osy = g.osx = s; g.ooy = ph; g.oox = pw; g.sy = g.sx = 1;
            g.accumulate = accumulate; g.dense = (s == 1);
            int nt = 0;
            for (int kh = 0; kh < k; ++kh) {
                if ((ph + p - kh) % s != 0) continue;
                for (int kw = 0; kw < k; ++kw) {
                    if ((pw + p - kw) % s != 0) continue;
                    g.tap_dy[nt] = (ph + p - kh) / s; g.tap_dx[nt] = (pw + p - kw) / s; g.tap_w[nt] = kh * k + kw;
                    ++nt;
                }
            }
            YH_REQUIRE(nt > 0, "conv_bwd_data: residue class without taps");
            g.nTaps = nt; g.Ktot = nt * Cout;
            if (g.M == 0) continue;
            gs.c[ncls++] = g;
        }
    hipStream_t st = (hipStream_t)stream;
    switch (ncls) {
        case 0: return 0;
        case 1: return launch_set<1>(gs, st);
        case 2: return launch_set<2>(gs, st);
        case 3: return launch_set<3>(gs, st);
        default: return launch_set<4>(gs, st);
    }
}

// Stride-2 3x3 backward-data with the two column parities MERGED into the channel axis (for narrow layers: Cin <= 32).
// dx is dense in pixels (lddx == Cin), so the pixel pair (2q, 2q+1) of a row is 2*Cin contiguous floats: view dx as
// (B, Hi, Wi/2, 2*Cin) and compute, per row parity ph, out[(y, q)][(pw, ci)] from the dy pixels (oy, q) and (oy, q+1):
//   ph = 0: kh = 1, oy = y/2;   ph = 1: kh = 0 -> oy + 1, kh = 2 -> oy
//   pw = 0: kw = 1 from column q;   pw = 1: kw = 2 from column q, kw = 0 from column q + 1
// The merged weights W'[kh][c][co][(pw, ci)] (yh_pack_weights_s2m) hold zero blocks where a (column, parity) pair has no
// tap: 4/3 of the exact multiplies, but N doubles (no half-empty MFMA column tiles at Cin = 16), stores are full
// 128-byte lines instead of every other pixel, and two classes instead of four share the launch.
extern "C" int yh_conv_bwd_data_s2m(const float *dy, int lddy, const float *wbm, int ldw, float *dx, int lddx, int B, int Hi,
                                    int Wi, int Cin, int Cout, int accumulate, void *stream) {
    YH_REQUIRE(dy && wbm && dx && B > 0 && Cin > 0 && Cout > 0 && Hi > 0, "conv_bwd_data_s2m: bad argument");
    YH_REQUIRE(lddx == Cin && Wi % 2 == 0 && lddy >= Cout && ldw >= 2 * Cin, "conv_bwd_data_s2m: needs a pixel-dense dx and an even width");
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    GatherGemmSet gs{};
    int ncls = 0;
    for (int ph = 0; ph < 2; ++ph) {
        GatherGemm g{};
        g.in = dy; g.w = wbm; g.bias = nullptr; g.out = dx; g.stats = nullptr;
        g.Hi = Ho; g.Wi = Wo; g.ldi = lddy; g.Cin = Cout; g.ldw = ldw;
        g.Ho_f = Hi; g.Wo_f = Wi / 2; g.ldo = 2 * lddx; g.N = 2 * Cin;
        g.B = B; g.Yo = (Hi - ph + 1) / 2; g.Xo = Wi / 2; g.M = B * g.Yo * g.Xo;
        g.osy = 2; g.osx = 1; g.ooy = ph; g.oox = 0; g.sy = g.sx = 1;
        g.accumulate = accumulate; g.dense = 0;
        int nt = 0;
        for (int kh = 0; kh < 3; ++kh) {
            if ((ph + 1 - kh) % 2 != 0) continue;
            for (int c = 0; c < 2; ++c) {
                g.tap_dy[nt] = (ph + 1 - kh) / 2; g.tap_dx[nt] = c; g.tap_w[nt] = kh * 2 + c;
                ++nt;
            }
        }
        g.nTaps = nt; g.Ktot = nt * Cout;
        if (g.M == 0) continue;
        gs.c[ncls++] = g;
    }
    hipStream_t st = (hipStream_t)stream;
    if (ncls == 0) return 0;
    return ncls == 1 ? launch_set<1>(gs, st) : launch_set<2>(gs, st);
}

// Backward-data of two pointwise (1x1, stride 1) convolutions that read the SAME input x (the conv1 / conv2 pair
// of a C3 block): dx (+)= dy1 * W1^T + dy2 * W2^T as ONE GEMM with K = Cout1 + Cout2, so dx is written once
// instead of written and then read-modified-written.  wb holds the two backward packs stacked: rows [0, Cout1)
// from conv1, [Cout1, Cout1+Cout2) from conv2.
extern "C" int yh_conv_bwd_data_pair(const float *dy1, int cout1, const float *dy2, int cout2, int lddy, const float *wb,
                                     int ldwb, float *dx, int lddx, int B, int H, int W, int Cin, int accumulate,
                                     void *stream) {
    YH_REQUIRE(dy1 && dy2 && wb && dx && B > 0 && H > 0 && W > 0 && Cin > 0 && cout1 > 0 && cout2 > 0,
               "conv_bwd_data_pair: bad argument");
    YH_REQUIRE(lddy >= cout1 && lddy >= cout2 && lddx >= Cin, "conv_bwd_data_pair: ld smaller than channel count");
    YH_REQUIRE(cout1 % 4 == 0 && lddy % 4 == 0 && (((uintptr_t)dy1 | (uintptr_t)dy2) & 15) == 0,
               "conv_bwd_data_pair: sources must be 16-byte addressable and Cout1 a multiple of 4");
    GatherGemm g{};
    g.in = dy1; g.in2 = dy2; g.ksplit = cout1; g.w = wb; g.bias = nullptr; g.out = dx; g.stats = nullptr;
    g.Hi = H; g.Wi = W; g.ldi = lddy; g.Cin = cout1 + cout2; g.ldw = ldwb;
    g.Ho_f = H; g.Wo_f = W; g.ldo = lddx; g.N = Cin;
    g.B = B; g.Yo = H; g.Xo = W; g.M = B * H * W;
    g.osy = g.osx = 1; g.ooy = g.oox = 0; g.sy = g.sx = 1;
    g.nTaps = 1; g.Ktot = cout1 + cout2; g.accumulate = accumulate; g.dense = 1;
    g.tap_dy[0] = 0; g.tap_dx[0] = 0; g.tap_w[0] = 0;
    return launch(g, (hipStream_t)stream);
}
