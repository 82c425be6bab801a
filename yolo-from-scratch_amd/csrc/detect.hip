// Inference post-process of predict(): order-preserving candidate compaction and class-aware NMS.
//
// NMS index selection must match the CPU definition bit for bit, so this file is compiled with
// floating-point contraction OFF (no FMA fusion): IoU is evaluated as
//   inter / ((area_i + area_j) - inter) > thr      with area = (x2-x1)*(y2-y1),
// one IEEE rounding per operation, exactly as torchvision's CPU kernel does; that kernel takes `thr` as a C double and
// promotes the fp32 IoU for the comparison, which is the same as comparing against the largest float <= thr (host side).
//
// torchvision.ops.batched_nms (train.py:1232-1233) has TWO branches and both are here (yh_nms `mode`):
//   per class ("vanilla", numel > 4000 on CPU tensors / > 20000 on GPU tensors): suppression only inside a class;
//   coordinate trick (otherwise): every box is shifted by float(class) * (max coordinate + 1) -- three fp32 operations,
//   each rounded -- and ONE class-agnostic NMS runs over the shifted boxes (areas and intersections of the SHIFTED
//   coordinates, boxes of different classes may meet when coordinates are negative).  The branch is picked on the device
//   from the candidate count, like everything else here.
//
// Pipeline (all sizes are device-side: M is read from count[0], launches are sized by `cap`):
//   1. keys = (descending-orderable score, candidate index), all distinct -> position in the sorted order = number of
//      smaller keys (rank counting, one thread per candidate over LDS-staged key tiles, whole chip); ties in score keep
//      the lower candidate index first = a stable descending sort.  Boxes / classes are scattered to sorted order.
//   2. 64x64-tiled suppression bit matrix (upper triangle): bit (i, j) = j > i, same class, IoU > thr; a fixed grid
//      walks the blocks the device-side M needs.
//   3. one workgroup resolves the greedy scan: per 64-box block one wave walks the diagonal word with a scalar chain and
//      ORs the next removal word itself; the other waves stage the diagonal band through LDS and OR the kept rows into
//      the later removal words, every mask load requested several steps ahead of its use.
#pragma clang fp contract(off)
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
struct CandArgs {
    const float *pred[3];
    float anchors[18];
    int grid[3];
    int cell_begin[4];
    int nc, cap;
    float img, thr, pad_left, pad_top, scale;
    const float *lb;      // optional device {pad_left, pad_top, scale}: per-image values under hipGraph replay
    float *boxes, *scores;
    int32_t *classes, *count, *blk;   // blk: [nblk] counts then [nblk] offsets
    int nblk;
};

__device__ __forceinline__ bool cand_flag(const CandArgs &a, int cell, int &s, int &local) {
    if (cell >= a.cell_begin[3]) return false;
    s = cell >= a.cell_begin[2] ? 2 : (cell >= a.cell_begin[1] ? 1 : 0);
    local = cell - a.cell_begin[s];
    float obj = yh_sigmoid(a.pred[s][(size_t)local * (5 + a.nc) + 4]);
    return obj > a.thr;
}

__global__ void cand_count_kernel(const CandArgs a) {
    __shared__ int wsum[4];
    int s, local;
    bool f = cand_flag(a, blockIdx.x * 256 + threadIdx.x, s, local);
    unsigned long long m = __ballot(f);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) a.blk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(64) void cand_scan_kernel(const CandArgs a) {      // one wave: exclusive prefix of the block counts
    const int lane = threadIdx.x;
    int run = 0;
    for (int b0 = 0; b0 < a.nblk; b0 += 64) {
        const int b = b0 + lane;
        const int c = b < a.nblk ? a.blk[b] : 0;
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        if (b < a.nblk) a.blk[a.nblk + b] = run + inc - c;
        run += __shfl(inc, 63);
    }
    if (lane == 0) a.count[0] = run;
}

__global__ void cand_write_kernel(const CandArgs a) {
    __shared__ int wsum[4];
    int s = 0, local = 0;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    bool f = cand_flag(a, cell, s, local);
    unsigned long long m = __ballot(f);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wsum[w] = __popcll(m);
    __syncthreads();
    if (!f) return;
    int pos = a.blk[a.nblk + blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
    for (int k = 0; k < w; ++k) pos += wsum[k];
    if (pos >= a.cap) return;
    const int ch = 5 + a.nc, G = a.grid[s];
    const float *p = a.pred[s] + (size_t)local * ch;
    int an = local % 3, q = local / 3;
    int j = q % G, i = (q / G) % G;
    // decode_predictions(pred, anchors, img_size) (train.py:758-774)
    float bx = ((yh_sigmoid(p[0]) * 2.0f - 0.5f) + (float)j) / (float)G;
    float by = ((yh_sigmoid(p[1]) * 2.0f - 0.5f) + (float)i) / (float)G;
    float tw = 2.0f * yh_sigmoid(p[2]), th = 2.0f * yh_sigmoid(p[3]);
    float bw = (a.anchors[(s * 3 + an) * 2 + 0] / a.img) * (tw * tw);
    float bh = (a.anchors[(s * 3 + an) * 2 + 1] / a.img) * (th * th);
    float obj = yh_sigmoid(p[4]);
    float cp;
    int cid = 0;
    if (a.nc == 1) {
        cp = yh_sigmoid(p[5]);
    } else {
        cp = -INFINITY;
        for (int c = 0; c < a.nc; ++c) {
            float v = yh_sigmoid(p[5 + c]);
            if (v > cp) { cp = v; cid = c; }     // first maximum wins
        }
    }
    // pixels -> corners -> un-letterbox (train.py:1192-1213), same operation order
    float xc = bx * a.img, yc = by * a.img, wp = bw * a.img, hp = bh * a.img;
    float x1 = xc - wp / 2, y1 = yc - hp / 2, x2 = xc + wp / 2, y2 = yc + hp / 2;
    const float pl = a.lb ? a.lb[0] : a.pad_left, pt = a.lb ? a.lb[1] : a.pad_top, sc = a.lb ? a.lb[2] : a.scale;
    x1 = (x1 - pl) / sc; y1 = (y1 - pt) / sc;
    x2 = (x2 - pl) / sc; y2 = (y2 - pt) / sc;
    a.boxes[4 * pos + 0] = x1; a.boxes[4 * pos + 1] = y1; a.boxes[4 * pos + 2] = x2; a.boxes[4 * pos + 3] = y2;
    a.scores[pos] = obj * cp;
    a.classes[pos] = cid;
}

// ------------------------------------------------------------------------------------------------
constexpr int kNmsThreads = 1024;
constexpr int kRankTile = 1024;   // keys staged in LDS per pass of the rank kernel
constexpr int kMaxWords = 4096;   // mask words per row the scan keeps in LDS (cap <= 64 * kMaxWords)

__device__ __forceinline__ uint64_t make_key(float score, int idx) {
    uint32_t u = __float_as_uint(score);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending-orderable
    return ((uint64_t)(~u) << 32) | (uint32_t)idx;    // ascending key = descending score, then index
}

// torchvision's branch rule, evaluated on the device-side candidate count (numel = 4 M)
__device__ __forceinline__ bool nms_uses_trick(int mode, int M) {
    if (mode == YH_NMS_PER_CLASS) return false;
    if (mode == YH_NMS_COORDINATE_TRICK) return true;
    return !(4 * (int64_t)M > (mode == YH_NMS_TORCHVISION_CPU ? 4000 : 20000));
}

// torch.max semantics: a NaN operand wins
__device__ __forceinline__ float max_nan(float a, float b) { return (a != a || b != b) ? NAN : fmaxf(a, b); }

struct NmsArgs {
    const float *boxes, *scores;
    const int32_t *classes, *count;
    int cap, W;                 // W = words per mask row = ceil(cap/64)
    float thr;                  // largest float <= the caller's double threshold
    int mode;                   // YH_NMS_*
    float *pmax;                // [ceil(cap/64)] per-workgroup maximum coordinate (coordinate trick)
    float *sboxes;              // [cap][4]
    int32_t *sclasses, *order;  // [cap]
    uint64_t *mask;             // [cap][W]
    int32_t *keep, *nkeep;
};

// Stable descending-score order by RANK COUNTING: the keys (score, candidate index) are unique, so the position of a
// candidate in the sorted order is the number of keys smaller than its own.  A workgroup owns 64 candidates; its four
// waves each walk a quarter of every LDS-staged key tile (broadcast 16-byte reads) and the four partial counts meet in
// LDS -- M / 64 workgroups use all four SIMDs of their CU (at M = 3000 only 47 of the 256 CUs have work, so the walk per
// SIMD is what the kernel's duration is made of).
constexpr int kRankWaves = 4;
__global__ __launch_bounds__(64 * kRankWaves) void nms_rank_kernel(const NmsArgs a) {
    __shared__ __attribute__((aligned(16))) uint64_t tile[kRankTile];
    __shared__ int part[kRankWaves][64];
    int M = a.count[0];
    if (M > a.cap) M = a.cap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= M) return;                     // whole workgroup out of range (uniform)
    const uint64_t mine = i < M ? make_key(a.scores[i], i) : ~0ull;
    int rank = 0;
    typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
    constexpr int QT = kRankTile / kRankWaves;
    for (int base = 0; base < M; base += kRankTile) {
        __syncthreads();
        for (int j = threadIdx.x; j < kRankTile; j += 64 * kRankWaves) {
            int c = base + j;
            tile[j] = c < M ? make_key(a.scores[c], c) : ~0ull;       // padding keys are never smaller than a real key
        }
        __syncthreads();
        int lim = M - base - wave * QT;                   // keys of this wave's quarter that exist (padding counts nothing)
        lim = lim <= 0 ? 0 : (lim < QT ? ((lim + 1) & ~1) : QT);
        const uint64_t *q = tile + wave * QT;
#pragma unroll 8
        for (int j = 0; j < lim; j += 2) {                // broadcast 16-byte reads: two keys per LDS instruction
            const u64x2 k2 = *(const u64x2 *)&q[j];
            rank += (k2[0] < mine ? 1 : 0) + (k2[1] < mine ? 1 : 0);
        }
    }
    part[wave][lane] = rank;
    __syncthreads();
    if (wave != 0) return;
    rank = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
    float cmax = -INFINITY;
    if (i < M) {
        a.order[rank] = i;
        a.sclasses[rank] = a.classes[i];
        const f32x4 b = *(const f32x4 *)(a.boxes + 4 * i);
        *(f32x4 *)(a.sboxes + 4 * rank) = b;
        cmax = max_nan(max_nan(b[0], b[1]), max_nan(b[2], b[3]));
    }
    // boxes.max() of the coordinate trick: this workgroup's share (max is exact, so any reduction order gives the same bits)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cmax = max_nan(cmax, __shfl_xor(cmax, o));
    if (lane == 0) a.pmax[blockIdx.x] = cmax;
}

// 64 x 64 blocks of the suppression bit matrix, upper triangle only (row block <= column block); a fixed grid of
// workgroups walks the blocks the actual candidate count needs (M lives on the device).
__global__ __launch_bounds__(64) void nms_mask_kernel(const NmsArgs a) {
    int M = a.count[0];
    if (M > a.cap) M = a.cap;
    const int nw = (M + 63) >> 6;
    const int npairs = nw * (nw + 1) / 2;
    __shared__ float cbx[64][4];
    __shared__ int ccl[64];
    const int t = threadIdx.x;
    const bool trick = nms_uses_trick(a.mode, M);
    float unit = 0.f;                                     // max_coordinate + 1 (fp32), the per-class shift unit
    if (trick && blockIdx.x < npairs) {
        float m = -INFINITY;
        for (int j = t; j < nw; j += 64) m = max_nan(m, a.pmax[j]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = max_nan(m, __shfl_xor(m, o));
        unit = m + 1.0f;
    }
    for (int p = blockIdx.x; p < npairs; p += gridDim.x) {
        // p -> (rb, cb), rb <= cb: row rb starts at rb * nw - rb (rb - 1) / 2
        int rb = (int)(((2.0f * nw + 1.0f) - sqrtf((2.0f * nw + 1.0f) * (2.0f * nw + 1.0f) - 8.0f * (float)p)) * 0.5f);
        if (rb < 0) rb = 0;
        if (rb > nw - 1) rb = nw - 1;
        while (rb > 0 && rb * nw - rb * (rb - 1) / 2 > p) --rb;
        while ((rb + 1) * nw - (rb + 1) * rb / 2 <= p) ++rb;
        const int cb = rb + (p - (rb * nw - rb * (rb - 1) / 2));
        const int cj = cb * 64 + t;
        __syncthreads();
        if (cj < M) {
            f32x4 bj = *(const f32x4 *)(a.sboxes + 4 * cj);
            const int cl = a.sclasses[cj];
            if (trick) {                                   // boxes + offsets[:, None], offsets = idxs.to(boxes) * unit
                const float off = (float)cl * unit;
                bj[0] = bj[0] + off; bj[1] = bj[1] + off; bj[2] = bj[2] + off; bj[3] = bj[3] + off;
            }
            *(f32x4 *)cbx[t] = bj;
            ccl[t] = trick ? 0 : cl;
        }
        __syncthreads();
        const int i = rb * 64 + t;
        if (i >= M) continue;
        f32x4 bi = *(const f32x4 *)(a.sboxes + 4 * i);
        int ci = a.sclasses[i];
        if (trick) {
            const float off = (float)ci * unit;
            bi[0] = bi[0] + off; bi[1] = bi[1] + off; bi[2] = bi[2] + off; bi[3] = bi[3] + off;
            ci = 0;
        }
        const float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
        uint64_t bits = 0;
        int ncol = M - cb * 64;
        if (ncol > 64) ncol = 64;
        for (int c = (rb == cb ? t + 1 : 0); c < ncol; ++c) {
            if (ccl[c] != ci) continue;
            float xx1 = fmaxf(bi[0], cbx[c][0]), yy1 = fmaxf(bi[1], cbx[c][1]);
            float xx2 = fminf(bi[2], cbx[c][2]), yy2 = fminf(bi[3], cbx[c][3]);
            float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
            float inter = w * h;
            float aj = (cbx[c][2] - cbx[c][0]) * (cbx[c][3] - cbx[c][1]);
            float iou = inter / ((ai + aj) - inter);
            if (iou > a.thr) bits |= 1ull << c;
        }
        a.mask[(size_t)i * a.W + cb] = bits;
    }
}

// Greedy scan in descending score order, 64 candidates per step, one workgroup of 16 waves with fixed roles.  The removal
// words live in LDS.  The mask was written by other CUs a moment ago, so every read of it is a ~2 us trip past this XCD's
// L2: no step may wait for a load issued less than a few steps earlier.
//
//   wave 0       resolves block bi: diagonal word from the LDS band ring, then a SCALAR chain (the removal word sits in an
//                SGPR pair: bit test, select, or -- three dependent scalar instructions per candidate; a kept candidate's own
//                bit is never set later, so kept = ~removal when the chain ends), publishes the kept list, and ORs the kept
//                rows' word bi+1 (also in the band ring) across the wave with DPP rotations -- the only removal word the
//                next step needs from this one.
//   waves 1-4    band loaders: the diagonal word, the four words after it and the original index of every row of a block
//                go into a four-slot LDS ring; a loader owns every fourth block and holds its loads four steps in registers.
//   wave 5       ORs the kept rows' words bi+2..bi+4 out of the band ring one step later (LDS to LDS).
//   waves 6-14   three groups taking turns: the group whose turn it is requests the kept rows' words >= bi+5 from memory
//                and ORs in what it requested three steps ago -- block bi+5 still finds its removal word complete.
// LDS ORs are atomic and order independent: the result is deterministic.
constexpr int kBandWords = 5, kBandRing = 4, kFarSlots = 6, kFarThreads = 192;   // 16 waves: 128 VGPRs per lane, no spills

__device__ __forceinline__ unsigned wave_or(unsigned v) {          // OR over the 64 lanes (all active), uniform result
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);    // row_ror:8
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);    // row_ror:4
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false);    // row_ror:2
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false);    // row_ror:1 -> every lane holds its row's OR
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0) | (unsigned)__builtin_amdgcn_readlane((int)v, 16) |
           (unsigned)__builtin_amdgcn_readlane((int)v, 32) | (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}

__global__ __launch_bounds__(kNmsThreads) void nms_scan_kernel(const NmsArgs a) {
    static_assert(kNmsThreads == 1024, "sixteen waves with fixed roles");
    __shared__ uint64_t remv[kMaxWords];
    __shared__ uint64_t band[kBandRing][kBandWords][64];
    __shared__ int bord[kBandRing][64];
    __shared__ int klist[2][64];
    __shared__ int nkept_sh[2];
    int M = a.count[0];
    if (M > a.cap) M = a.cap;
    const int nw = (M + 63) >> 6;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int w = t; w < nw; w += kNmsThreads) remv[w] = (w == nw - 1 && (M & 63)) ? ~0ull << (M & 63) : 0ull;   // boxes past M never win
    if (nw == 0) {
        if (t == 0) a.nkeep[0] = 0;
        return;
    }
    auto lds_or = [&](int w, uint64_t v) {
        unsigned *r32 = (unsigned *)&remv[w];
        if ((unsigned)v) atomicOr(r32, (unsigned)v);
        if ((unsigned)(v >> 32)) atomicOr(r32 + 1, (unsigned)(v >> 32));
    };
    // Every role runs its own loop of exactly nw steps with one barrier per step (the roles are wave-uniform), so the
    // registers a role keeps across steps -- loads in flight -- are its own.

    if (wave == 0) {
        // ---- resolver ----
        int nk = 0;
        __syncthreads();
        for (int bi = 0; bi < nw; ++bi) {
            const int slot = bi & (kBandRing - 1);
            const uint64_t dg = band[slot][0][lane], d1 = band[slot][1][lane];
            const int ord = bord[slot][lane];
            const uint64_t r0 = remv[bi];
            uint64_t cur = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(r0 >> 32)) << 32) |
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)r0);
            const unsigned dlo = (unsigned)dg, dhi = (unsigned)(dg >> 32);
#pragma unroll
            for (int b = 0; b < 64; ++b) {
                const uint64_t d = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)dhi, b) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane((int)dlo, b);
                uint64_t sel;                          // cur |= bit b of cur set ? 0 : d
                asm("s_bitcmp1_b64 %0, %2\n\ts_cselect_b64 %1, 0, %3\n\ts_or_b64 %0, %0, %1"
                    : "+s"(cur), "=&s"(sel) : "n"(b), "s"(d) : "scc");
            }
            const uint64_t kept = ~cur;
            const bool mine = (kept >> lane) & 1ull;
            const unsigned o0 = wave_or(mine ? (unsigned)d1 : 0u), o1 = wave_or(mine ? (unsigned)(d1 >> 32) : 0u);
            if (lane == 0) {
                if (bi + 1 < nw) lds_or(bi + 1, ((uint64_t)o1 << 32) | o0);
                nkept_sh[bi & 1] = __popcll(kept);
            }
            if (mine) {
                const int pos = __popcll(kept & ((1ull << lane) - 1ull));
                a.keep[nk + pos] = ord;
                klist[bi & 1][pos] = lane;
            }
            nk += __popcll(kept);
            __syncthreads();
        }
        if (lane == 0) a.nkeep[0] = nk;
    } else if (wave <= kBandRing) {
        // ---- band loader: blocks j0, j0 + 4, ...; loads are unconditional (clamped addresses) and validity is applied when
        // the registers go to LDS, so a load's destination is the register that waits four steps
        const int j0 = wave - 1;
        uint64_t hb[kBandWords];
        int ho;
        auto band_fetch = [&](int b) {
            const int bc = b < nw ? b : nw - 1;
            int r = bc * 64 + lane;
            r = r < M ? r : M - 1;
            const uint64_t *row = a.mask + (size_t)r * a.W;
#pragma unroll
            for (int k = 0; k < kBandWords; ++k) hb[k] = row[bc + k < nw ? bc + k : nw - 1];
            ho = a.order[r];
        };
        auto band_store = [&](int b) {
            const bool rv = b * 64 + lane < M;
#pragma unroll
            for (int k = 0; k < kBandWords; ++k) band[b & (kBandRing - 1)][k][lane] = (rv && b + k < nw) ? hb[k] : 0ull;
            bord[b & (kBandRing - 1)][lane] = ho;
        };
        band_fetch(j0);
        if (j0 < nw) band_store(j0);
        band_fetch(j0 + kBandRing);
        __syncthreads();
        for (int bi = 0; bi < nw; ++bi) {
            const int b = bi + 1;                     // the block the next step resolves
            if (b >= kBandRing && ((b - j0) & (kBandRing - 1)) == 0) {
                if (b < nw) band_store(b);            // requested four steps ago
                band_fetch(b + kBandRing);
            }
            __syncthreads();
        }
    } else if (wave == 5) {
        // ---- near helper: rows kept in step bi-1, band words 2..4 -> removal words bi+1..bi+3
        __syncthreads();
        for (int bi = 0; bi < nw; ++bi) {
            if (bi >= 1) {
                const int b = bi - 1, src = b & 1, slot = b & (kBandRing - 1), total = nkept_sh[src] * 3;
                for (int idx = lane; idx < total; idx += 64) {
                    const int kb = idx / 3, k = 2 + (idx - 3 * kb), w = b + k;
                    if (w < nw) {
                        const uint64_t v = band[slot][k][klist[src][kb]];
                        if (v) lds_or(w, v);
                    }
                }
            }
            __syncthreads();
        }
    } else if (wave < 15) {
        // ---- far helpers: group g takes the blocks b = g, g + 3, ...; at step b + 1 it ORs in what it requested at step
        // b - 2 (rows kept in step b - 3) and requests the rows kept in step b, words >= b + 5
        const int fgroup = (wave - 6) / 3;
        const int ft = (wave - 6 - 3 * fgroup) * 64 + lane;
        uint64_t pv[kFarSlots];
        int pw[kFarSlots];
        int pn = 0;                                   // this thread's valid requests
#pragma unroll
        for (int q = 0; q < kFarSlots; ++q) { pv[q] = 0ull; pw[q] = 0; }
        __syncthreads();
        for (int bi = 0; bi < nw; ++bi) {
            if (bi >= 1 && (bi - 1) % 3 == fgroup) {
#pragma unroll
                for (int q = 0; q < kFarSlots; ++q)
                    if (q < pn && pv[q]) lds_or(pw[q], pv[q]);
                pn = 0;
                const int b = bi - 1, w0 = b + kBandWords, nrem = nw - w0;
                const int src = b & 1, total = nrem > 0 ? nkept_sh[src] * nrem : 0;
                if (ft < total) {
                    const uint64_t *rows = a.mask + (size_t)b * 64 * a.W;
#pragma unroll
                    for (int q = 0; q < kFarSlots; ++q) {
                        int idx = ft + q * kFarThreads;
                        if (idx < total) pn = q + 1;
                        idx = idx < total ? idx : total - 1;             // unconditional load, clamped
                        const int kb = idx / nrem, w = w0 + (idx - kb * nrem);
                        pv[q] = rows[(size_t)klist[src][kb] * a.W + w];
                        pw[q] = w;
                    }
                    for (int idx = ft + kFarSlots * kFarThreads; idx < total; idx += kFarThreads) {   // beyond the slots: right away
                        const int kb = idx / nrem, w = w0 + (idx - kb * nrem);
                        const uint64_t v = rows[(size_t)klist[src][kb] * a.W + w];
                        if (v) lds_or(w, v);
                    }
                }
            }
            __syncthreads();
        }
    } else {
        __syncthreads();
        for (int bi = 0; bi < nw; ++bi) __syncthreads();
    }
}

// Kept detections as one dense table: header {count, nkeep} + nkeep rows (x1, y1, x2, y2, score, class bits) in NMS order,
// so the host reads the result of predict() with ONE device->host copy (train.py:1236-1246 gathers boxes / scores /
// classes with three indexed reads and one .item() per scalar).
__global__ void gather_detections_kernel(const float *__restrict__ boxes, const float *__restrict__ scores,
                                         const int32_t *__restrict__ classes, const int32_t *__restrict__ count,
                                         const int32_t *__restrict__ keep, const int32_t *__restrict__ nkeep, int cap,
                                         float *__restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    int n = nkeep[0];
    if (n > cap) n = cap;
    if (k == 0) {
        ((int32_t *)out)[0] = count[0];
        ((int32_t *)out)[1] = n;
    }
    if (k >= n) return;
    const int i = keep[k];
    const f32x4 b = *(const f32x4 *)(boxes + 4 * i);
    float *r = out + 8 + 6 * (size_t)k;
    r[0] = b[0]; r[1] = b[1]; r[2] = b[2]; r[3] = b[3];
    r[4] = scores[i];
    r[5] = __int_as_float(classes[i]);
}

// ------------------------------------------------------------------------------------------------
// Target assignment of YOLODataset.__getitem__ (train.py:164-205) on the device: one thread per image walks
// its labels IN ORDER (first writer wins), so the result is identical to the sequential host rule.
// The reference mixes python floats (double) and float32 tensors; the same types are used here:
// cell index and stored box from doubles, shape-IoU against the anchors in float32, no FMA contraction.
struct AssignArgs {
    const double *labels;     // [B][maxn][5] = class, xc, yc, w, h (normalised to the padded square image)
    const int32_t *nlabels;   // [B]
    float *tgt[3];
    float anchors[18];
    int grid[3];
    int B, maxn, nc;
    double img;
};

__global__ void assign_targets_kernel(const AssignArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const int ch = 5 + a.nc;
    const int n = a.nlabels[b] < a.maxn ? a.nlabels[b] : a.maxn;
    for (int k = 0; k < n; ++k) {
        const double *lab = a.labels + ((size_t)b * a.maxn + k) * 5;
        const int cid = (int)lab[0];
        const double xc = lab[1], yc = lab[2], w = lab[3], h = lab[4];
        const float bw = (float)(w * a.img), bh = (float)(h * a.img);
        float best = -1.f;
        int bs = 0, ba = 0;
        for (int s = 0; s < 3; ++s) {
            float smax = -1.f;
            int sarg = 0;
            for (int an = 0; an < 3; ++an) {
                float aw = a.anchors[(s * 3 + an) * 2], ah = a.anchors[(s * 3 + an) * 2 + 1];
                float inter = fminf(bw, aw) * fminf(bh, ah);
                float uni = (bw * bh + aw * ah) - inter;
                float iou = inter / (uni + 1e-16f);
                if (iou > smax) { smax = iou; sarg = an; }
            }
            if (smax > best) { best = smax; bs = s; ba = sarg; }
        }
        const int G = a.grid[bs];
        int gx = (int)(xc * (double)G), gy = (int)(yc * (double)G);
        gx = gx < G - 1 ? gx : G - 1;
        gy = gy < G - 1 ? gy : G - 1;
        if (gx < 0 || gy < 0) continue;
        float *t = a.tgt[bs] + ((((size_t)b * G + gy) * G + gx) * 3 + ba) * ch;
        if (t[4] == 0.f) {
            t[0] = (float)xc; t[1] = (float)yc; t[2] = (float)w; t[3] = (float)h;
            t[4] = 1.f;
            t[a.nc == 1 ? 5 : 5 + cid] = 1.f;
        }
    }
}

inline int pow2_ge(int v) { int n = 1; while (n < v) n <<= 1; return n; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" int64_t yh_candidates_ws(const int grid[3]) {
    int64_t cells = 0;
    for (int s = 0; s < 3; ++s) cells += (int64_t)grid[s] * grid[s] * 3;
    return 2 * cdiv64(cells, 256);
}

extern "C" int yh_candidates(const float *const pred[3], const float *anchors, const int grid[3], int nc, float img_size,
                             float conf_thr, float pad_left, float pad_top, float scale, float *boxes, float *scores,
                             int32_t *classes, int32_t *count, int cap, int32_t *ws, const float *letterbox_dev,
                             void *stream) {
    YH_REQUIRE(pred && anchors && grid && boxes && scores && classes && count && ws && cap > 0 && nc >= 1,
               "candidates: bad argument");
    YH_REQUIRE(((uintptr_t)boxes & 15) == 0, "candidates: boxes must be 16-byte aligned");
    CandArgs a{};
    int cells = 0;
    for (int s = 0; s < 3; ++s) {
        YH_REQUIRE(pred[s] && grid[s] > 0, "candidates: scale %d missing", s);
        a.pred[s] = pred[s]; a.grid[s] = grid[s];
        a.cell_begin[s] = cells;
        cells += grid[s] * grid[s] * 3;
    }
    a.cell_begin[3] = cells;
    for (int k = 0; k < 18; ++k) a.anchors[k] = anchors[k];
    a.nc = nc; a.cap = cap; a.img = img_size; a.thr = conf_thr;
    a.pad_left = pad_left; a.pad_top = pad_top; a.scale = scale; a.lb = letterbox_dev;
    a.boxes = boxes; a.scores = scores; a.classes = classes; a.count = count; a.blk = ws;
    a.nblk = cdiv(cells, 256);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cand_count_kernel, dim3(a.nblk), dim3(256), 0, st, a);
    YH_CHECK_LAUNCH("cand_count");
    hipLaunchKernelGGL(cand_scan_kernel, dim3(1), dim3(64), 0, st, a);
    YH_CHECK_LAUNCH("cand_scan");
    hipLaunchKernelGGL(cand_write_kernel, dim3(a.nblk), dim3(256), 0, st, a);
    YH_CHECK_LAUNCH("cand_write");
    return 0;
}

extern "C" int64_t yh_nms_ws(int cap) {
    if (cap <= 0) return 0;
    size_t W = (size_t)(cap + 63) / 64;
    size_t b = align_up((size_t)cap * 16, 256);           // sorted boxes
    b += 2 * align_up((size_t)cap * 4, 256);              // sorted classes, order
    b += align_up(W * 4, 256);                            // per-workgroup coordinate maxima
    b += align_up((size_t)cap * W * 8, 256);              // suppression bit matrix
    return (int64_t)b;
}

extern "C" int yh_nms(const float *boxes, const float *scores, const int32_t *classes, const int32_t *count, int cap,
                      double iou_thr, int mode, int32_t *keep, int32_t *nkeep, void *ws, void *stream) {
    YH_REQUIRE(boxes && scores && classes && count && keep && nkeep && ws && cap > 0, "nms: bad argument");
    YH_REQUIRE(mode >= YH_NMS_PER_CLASS && mode <= YH_NMS_TORCHVISION_CUDA, "nms: unknown mode %d", mode);
    YH_REQUIRE(cap <= 64 * kMaxWords, "nms: capacity %d above the supported %d", cap, 64 * kMaxWords);
    YH_REQUIRE(((uintptr_t)boxes & 15) == 0 && ((uintptr_t)ws & 255) == 0, "nms: boxes 16-byte / workspace 256-byte alignment");
    NmsArgs a{};
    a.boxes = boxes; a.scores = scores; a.classes = classes; a.count = count;
    a.cap = cap; a.W = (cap + 63) / 64; a.keep = keep; a.nkeep = nkeep; a.mode = mode;
    // (double)iou > thr  <=>  iou > (largest float <= thr): the CPU kernel's promoted comparison, done in fp32
    // mode YH_NMS_TORCHVISION_CUDA reproduces torchvision's DEVICE kernel, which compares float against float
    a.thr = (float)iou_thr;
    if (mode != YH_NMS_TORCHVISION_CUDA && (double)a.thr > iou_thr) a.thr = nextafterf(a.thr, -INFINITY);
    char *p = (char *)ws;
    a.sboxes = (float *)p;    p += align_up((size_t)cap * 16, 256);
    a.sclasses = (int32_t *)p; p += align_up((size_t)cap * 4, 256);
    a.order = (int32_t *)p;   p += align_up((size_t)cap * 4, 256);
    a.pmax = (float *)p;      p += align_up((size_t)a.W * 4, 256);
    a.mask = (uint64_t *)p;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(nms_rank_kernel, dim3(cdiv(cap, 64)), dim3(64 * kRankWaves), 0, st, a);
    YH_CHECK_LAUNCH("nms_rank");
    const int64_t pairs = (int64_t)a.W * (a.W + 1) / 2;
    hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)(pairs < 2048 ? pairs : 2048)), dim3(64), 0, st, a);
    YH_CHECK_LAUNCH("nms_mask");
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(kNmsThreads), 0, st, a);
    YH_CHECK_LAUNCH("nms_scan");
    return 0;
}

extern "C" int yh_gather_detections(const float *boxes, const float *scores, const int32_t *classes, const int32_t *count,
                                    const int32_t *keep, const int32_t *nkeep, int cap, float *out, void *stream) {
    YH_REQUIRE(boxes && scores && classes && count && keep && nkeep && out && cap > 0, "gather_detections: bad argument");
    YH_REQUIRE(((uintptr_t)boxes & 15) == 0, "gather_detections: boxes must be 16-byte aligned");
    hipLaunchKernelGGL(gather_detections_kernel, dim3(cdiv(cap, 256)), dim3(256), 0, (hipStream_t)stream, boxes, scores,
                       classes, count, keep, nkeep, cap, out);
    YH_CHECK_LAUNCH("gather_detections");
    return 0;
}

extern "C" int yh_assign_targets(const double *labels, const int32_t *nlabels, int B, int maxn, const float *anchors,
                                 const int grid[3], int nc, int img_size, float *const target[3], void *stream) {
    YH_REQUIRE(labels && nlabels && anchors && grid && target && B > 0 && maxn > 0 && nc >= 1, "assign_targets: bad argument");
    AssignArgs a{};
    a.labels = labels; a.nlabels = nlabels; a.B = B; a.maxn = maxn; a.nc = nc; a.img = (double)img_size;
    for (int k = 0; k < 18; ++k) a.anchors[k] = anchors[k];
    hipStream_t st = (hipStream_t)stream;
    for (int s = 0; s < 3; ++s) {
        YH_REQUIRE(target[s] && grid[s] > 0, "assign_targets: scale %d missing", s);
        a.tgt[s] = target[s]; a.grid[s] = grid[s];
        YH_HIP(hipMemsetAsync(target[s], 0, (size_t)B * grid[s] * grid[s] * 3 * (5 + nc) * sizeof(float), st));
    }
    hipLaunchKernelGGL(assign_targets_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, a);
    YH_CHECK_LAUNCH("assign_targets");
    return 0;
}
