// Fused multi-scale YOLO loss (decode + CIoU + BCE-with-logits), forward and backward in one pass
// over the three head outputs, plus the standalone decode / CIoU entry points.
//
// Latency-bound scan work, not MFMA work: one thread per (image, row, col, anchor) cell.  The three
// scales are processed by one launch; every workgroup belongs to exactly one scale so its partial
// sums (fp64) are per scale.  Sums over workgroups are added in index order by a final one-workgroup
// kernel: no float atomics, bitwise reproducible.  The number of positives per scale (the mean's
// denominator, needed by the gradient) comes from an integer-atomic counting pre-pass.
//
// CIoU's gradient is obtained with forward-mode dual numbers over the four decoded box coordinates;
// tie conventions follow torch's autograd (min/max split the gradient evenly on ties, clamp passes
// the gradient at the boundary).
#include "common.h"

namespace {

struct D4 {
    float v, d[4];
};
__device__ __forceinline__ D4 d4_const(float c) { return D4{c, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 d4_var(float x, int i) {
    D4 r = d4_const(x);
    r.d[i] = 1.f;
    return r;
}
__device__ __forceinline__ D4 operator+(D4 a, D4 b) {
    D4 r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
__device__ __forceinline__ D4 operator-(D4 a, D4 b) {
    D4 r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
__device__ __forceinline__ D4 operator*(D4 a, D4 b) {
    D4 r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
__device__ __forceinline__ D4 operator/(D4 a, D4 b) {
    D4 r; r.v = a.v / b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
    return r;
}
__device__ __forceinline__ D4 d4_scale(D4 a, float s) {
    D4 r; r.v = a.v * s;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * s;
    return r;
}
__device__ __forceinline__ D4 d4_addc(D4 a, float c) { a.v += c; return a; }
// min / max against a constant (the target box is constant)
__device__ __forceinline__ D4 d4_minc(D4 a, float c) {
    if (a.v < c) return a;
    if (a.v > c) return d4_const(c);
    return D4{c, {0.5f * a.d[0], 0.5f * a.d[1], 0.5f * a.d[2], 0.5f * a.d[3]}};
}
__device__ __forceinline__ D4 d4_maxc(D4 a, float c) {
    if (a.v > c) return a;
    if (a.v < c) return d4_const(c);
    return D4{c, {0.5f * a.d[0], 0.5f * a.d[1], 0.5f * a.d[2], 0.5f * a.d[3]}};
}
__device__ __forceinline__ D4 d4_clamp0(D4 a) { return a.v >= 0.f ? a : d4_const(0.f); }
__device__ __forceinline__ D4 d4_sq(D4 a) {
    D4 r; r.v = a.v * a.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = 2.f * a.v * a.d[i];
    return r;
}
__device__ __forceinline__ D4 d4_atan(D4 a) {
    D4 r; r.v = atanf(a.v);
    float g = 1.f / (1.f + a.v * a.v);
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = g * a.d[i];
    return r;
}

// 1 - CIoU of one (pred, target) pair and its gradient w.r.t. (px, py, pw, ph); train.py:646-708.
__device__ __forceinline__ float ciou_term(const float p[4], const float t[4], float eps, float grad[4]) {
    D4 px = d4_var(p[0], 0), py = d4_var(p[1], 1), pw = d4_var(p[2], 2), ph = d4_var(p[3], 3);
    const float tx = t[0], ty = t[1], tw = t[2], th = t[3];
    D4 hw = d4_scale(pw, 0.5f), hh = d4_scale(ph, 0.5f);        // pw / 2
    D4 px1 = px - hw, px2 = px + hw, py1 = py - hh, py2 = py + hh;
    float tx1 = tx - tw / 2, tx2 = tx + tw / 2, ty1 = ty - th / 2, ty2 = ty + th / 2;
    D4 iw = d4_clamp0(d4_minc(px2, tx2) - d4_maxc(px1, tx1));
    D4 ih = d4_clamp0(d4_minc(py2, ty2) - d4_maxc(py1, ty1));
    D4 inter = iw * ih;
    D4 uni = d4_addc(pw * ph, tw * th) - inter;
    D4 iou = inter / d4_addc(uni, eps);
    D4 rho2 = d4_sq(d4_addc(px, -tx)) + d4_sq(d4_addc(py, -ty));
    D4 cw = d4_maxc(px2, tx2) - d4_minc(px1, tx1);
    D4 ch = d4_maxc(py2, ty2) - d4_minc(py1, ty1);
    D4 c2 = d4_addc(d4_sq(cw) + d4_sq(ch), eps);
    D4 dist = rho2 / c2;
    float at_t = atanf(tw / (th + eps));
    D4 dat = d4_addc(d4_atan(pw / d4_addc(ph, eps)), -at_t);
    const float k = (float)(4.0 / (3.14159265358979323846 * 3.14159265358979323846));
    D4 v = d4_scale(d4_sq(dat), k);
    float alpha = v.v / (1.f - iou.v + v.v + eps);              // constant w.r.t. the gradient (no_grad)
    D4 ciou = iou - dist - d4_scale(v, alpha);
#pragma unroll
    for (int i = 0; i < 4; ++i) grad[i] = -ciou.d[i];
    return 1.f - ciou.v;
}

__device__ __forceinline__ float bce_logits(float x, float z) {
    return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
}

struct LossArgs {
    const float *pred[3], *tgt[3];
    void *dpred[3];         // float, or bf16 when dbf16 (the bf16 path's head gradient)
    int dbf16, ldd[3];      // ldd: elements per PIXEL of dpred (3 anchors x ch, possibly padded); 0 = contiguous
    float anchors[18];
    int grid[3];
    int64_t cells[3];       // B*G*G*3 per scale
    int blk_begin[4];       // workgroup ranges per scale
    int B, nc;
    float img;
    float lw[9], gw[9];     // per scale (box, obj, cls): weights of the total, and of its gradient
    int *counts;            // [4]
    int *poslist[3];        // per scale: cells with an object (filled by loss_main_kernel, any order), capacity = cells[s]
    double *part;           // [nblk][3]
    float *out;             // [13]
};

// One pass over the cells.  Thread = cell: objectness term and its gradient for every cell; for a positive cell also the box
// and class terms of the LOSS (their normalisation by the positive count happens in loss_final) and an entry in the scale's
// positive list -- the count (atomic, per wave) is what the gradient of those terms needs, so loss_pos_kernel writes them
// afterwards.  The gradient tensor is written as WHOLE ROWS by the workgroup that owns the cells: the span of its 256 cells
// (zeros, the objectness gradients, the padding elements of a pixel-padded bf16 tensor) goes out in aligned 16-byte pieces --
// no memset of the tensor before the launch, no 2-byte scattered stores into it (nc = 80: 0.28 GB fill + one partial-sector
// write per 340-byte cell before).
__global__ __launch_bounds__(256) void loss_main_kernel(const LossArgs a) {
    __shared__ double red[4][3];
    __shared__ float gobj_sh[256];
    const int s = blockIdx.x >= a.blk_begin[2] ? 2 : (blockIdx.x >= a.blk_begin[1] ? 1 : 0);
    const int64_t c0 = (int64_t)(blockIdx.x - a.blk_begin[s]) * 256;
    const int64_t cell = c0 + threadIdx.x;
    const int ch = 5 + a.nc, G = a.grid[s];
    const float gobj = a.gw[3 * s + 1];
    const bool hasd = a.dpred[s] != nullptr;
    double lbox = 0.0, lobj = 0.0, lcls = 0.0;
    float go = 0.f;
    bool pos = false;
    if (cell < a.cells[s]) {
        const float *p = a.pred[s] + cell * ch;
        const float *t = a.tgt[s] + cell * ch;
        const float x = p[4], z = t[4];
        lobj = (double)bce_logits(x, z);
        go = (yh_sigmoid(x) - z) * (gobj / (float)a.cells[s]);
        pos = z > 0.5f;
        if (pos) {
            int an = (int)(cell % 3);
            int64_t q = cell / 3;
            int j = (int)(q % G);
            int i = (int)((q / G) % G);
            float sg[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) sg[e] = yh_sigmoid(p[e]);
            float aw = a.anchors[(s * 3 + an) * 2 + 0] / a.img, ah = a.anchors[(s * 3 + an) * 2 + 1] / a.img;
            float box[4], tb[4] = {t[0], t[1], t[2], t[3]}, gb[4];
            box[0] = ((sg[0] * 2.0f - 0.5f) + (float)j) / (float)G;
            box[1] = ((sg[1] * 2.0f - 0.5f) + (float)i) / (float)G;
            float two_w = 2.0f * sg[2], two_h = 2.0f * sg[3];
            box[2] = aw * (two_w * two_w);
            box[3] = ah * (two_h * two_h);
            lbox = (double)ciou_term(box, tb, 1e-7f, gb);
        }
    }
    {   // positive count and list (order irrelevant: every entry is handled on its own)
        const unsigned long long m = __ballot(pos);
        const int lane = threadIdx.x & 63;
        // class term of this wave's positive cells: the whole wave walks one cell's classes (coalesced reads, a fixed-order
        // wave sum) instead of the owning thread walking nc = 80 of them while its workgroup waits
        for (unsigned long long mm = m; mm; mm &= mm - 1) {
            const int b = __ffsll((long long)mm) - 1;
            const int64_t cb = c0 + (threadIdx.x & ~63) + b;
            const float *pb = a.pred[s] + cb * ch, *tb2 = a.tgt[s] + cb * ch;
            double v = 0.0;
            for (int c = lane; c < a.nc; c += 64) v += (double)bce_logits(pb[5 + c], tb2[5 + c]);
            v = wave_sum_d(v);
            if (lane == b) lcls = v;
        }
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&a.counts[s], __popcll(m));
        base = __shfl(base, 0);
        if (pos) a.poslist[s][base + __popcll(m & ((1ull << lane) - 1ull))] = (int)cell;
    }
    gobj_sh[threadIdx.x] = go;
    lbox = wave_sum_d(lbox); lobj = wave_sum_d(lobj); lcls = wave_sum_d(lcls);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w][0] = lbox; red[w][1] = lobj; red[w][2] = lcls; }
    __syncthreads();
    if (threadIdx.x < 3)
        a.part[(size_t)blockIdx.x * 3 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (!hasd) return;

    // ---- this workgroup's rows of the gradient tensor: elements [e0, e1) of the (possibly pixel-padded) layout ---------------
    const unsigned LDD = a.ldd[s] ? (unsigned)a.ldd[s] : 3u * (unsigned)ch;           // elements per pixel
    const int64_t c1 = c0 + 256 < a.cells[s] ? c0 + 256 : a.cells[s];
    auto elem_of = [&](int64_t c) { return (unsigned)(c / 3) * LDD + (unsigned)(c % 3) * (unsigned)ch; };
    const unsigned e0 = elem_of(c0), e1 = c1 == a.cells[s] ? (unsigned)(a.cells[s] / 3) * LDD : elem_of(c1);
    const unsigned pix0 = (unsigned)(c0 / 3);
    auto value = [&](unsigned pixel, unsigned r) -> float {                         // r = element within the pixel's row
        const unsigned an = r / (unsigned)ch, chn = r - an * (unsigned)ch;
        if (an >= 3u || chn != 4u) return 0.f;
        return gobj_sh[(int)((int64_t)3 * pixel + an - c0)];
    };
    const unsigned esz = a.dbf16 ? 2u : 4u, epp = 16u / esz;                        // element size, elements per 16-byte piece
    unsigned char *const dp = (unsigned char *)a.dpred[s];
    const uintptr_t b0 = (uintptr_t)dp + (uintptr_t)e0 * esz, b1 = (uintptr_t)dp + (uintptr_t)e1 * esz;
    uintptr_t ab = (b0 + 15) & ~(uintptr_t)15, ae = b1 & ~(uintptr_t)15;
    if (ab > b1) ab = b1;
    if (ae < ab) ae = ab;
    auto put1 = [&](unsigned e) {                                                     // one element (ragged ends of the span)
        const unsigned pixel = e / LDD, r = e - pixel * LDD;
        const float v = value(pixel, r);
        if (a.dbf16) ((__bf16 *)dp)[e] = (__bf16)v;
        else ((float *)dp)[e] = v;
    };
    const unsigned nhead = (unsigned)((ab - b0) / esz), ntail = (unsigned)((b1 - ae) / esz);
    if (threadIdx.x < nhead) put1(e0 + threadIdx.x);
    if (threadIdx.x < ntail) put1((unsigned)((ae - (uintptr_t)dp) / esz) + threadIdx.x);
    const unsigned npiece = (unsigned)((ae - ab) >> 4), ea = (unsigned)((ab - (uintptr_t)dp) / esz);
    (void)pix0;
    for (unsigned k = threadIdx.x; k < npiece; k += 256) {
        const unsigned e = ea + k * epp;
        unsigned pixel = e / LDD, r = e - pixel * LDD;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = 0.f;
            if ((unsigned)j < epp) {
                v[j] = value(pixel, r);
                if (++r == LDD) { r = 0; ++pixel; }
            }
        }
        if (a.dbf16) {
            typedef __bf16 lbf16x8 __attribute__((ext_vector_type(8)));
            const lbf16x8 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3], (__bf16)v[4], (__bf16)v[5], (__bf16)v[6], (__bf16)v[7]};
            *(lbf16x8 *)(ab + (uintptr_t)k * 16) = o;
        } else {
            *(f32x4 *)(ab + (uintptr_t)k * 16) = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
}

// Box and class gradients of the positive cells: one WAVE per list entry (lane e < 4 stores box element e, lane c the class
// elements c, c + 64, ... -- coalesced; one thread per entry walked nc = 80 classes with a load and a store each: 190 us), the
// same expressions in the same order as the single-pass form, with the positive count now known.  Runs after
// loss_main_kernel wrote the rows.
__global__ __launch_bounds__(256) void loss_pos_kernel(const LossArgs a) {
    const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = gridDim.x * 4;
    for (int s = 0; s < 3; ++s) {
        if (!a.dpred[s] || a.cells[s] == 0) continue;
        const int npos = a.counts[s];
        const int ch = 5 + a.nc, G = a.grid[s];
        const float gbox = a.gw[3 * s + 0], gcls = a.gw[3 * s + 2];
        for (int i = wid; i < npos; i += nwave) {
            const int64_t cell = a.poslist[s][i];
            const float *p = a.pred[s] + cell * ch;
            const float *t = a.tgt[s] + cell * ch;
            const int64_t dbase = a.ldd[s] ? (cell / 3) * a.ldd[s] + (cell % 3) * ch : cell * ch;
            auto dstore = [&](int e, float v) {
                if (a.dbf16) ((__bf16 *)a.dpred[s])[dbase + e] = (__bf16)v;
                else ((float *)a.dpred[s])[dbase + e] = v;
            };
            int an = (int)(cell % 3);
            int64_t q = cell / 3;
            int j = (int)(q % G);
            int ii = (int)((q / G) % G);
            float sg[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) sg[e] = yh_sigmoid(p[e]);
            float aw = a.anchors[(s * 3 + an) * 2 + 0] / a.img, ah = a.anchors[(s * 3 + an) * 2 + 1] / a.img;
            float box[4], tb[4] = {t[0], t[1], t[2], t[3]}, gb[4];
            box[0] = ((sg[0] * 2.0f - 0.5f) + (float)j) / (float)G;
            box[1] = ((sg[1] * 2.0f - 0.5f) + (float)ii) / (float)G;
            float two_w = 2.0f * sg[2], two_h = 2.0f * sg[3];
            box[2] = aw * (two_w * two_w);
            box[3] = ah * (two_h * two_h);
            (void)ciou_term(box, tb, 1e-7f, gb);
            const float kb = gbox / (float)npos;
            float g4[4];
            g4[0] = gb[0] * (2.0f * sg[0] * (1.f - sg[0]) / (float)G) * kb;
            g4[1] = gb[1] * (2.0f * sg[1] * (1.f - sg[1]) / (float)G) * kb;
            g4[2] = gb[2] * (aw * 8.0f * sg[2] * sg[2] * (1.f - sg[2])) * kb;
            g4[3] = gb[3] * (ah * 8.0f * sg[3] * sg[3] * (1.f - sg[3])) * kb;
            if (lane < 4) dstore(lane, lane == 0 ? g4[0] : (lane == 1 ? g4[1] : (lane == 2 ? g4[2] : g4[3])));
            const float kc = a.nc > 0 ? gcls / ((float)npos * (float)a.nc) : 0.f;
            for (int c = lane; c < a.nc; c += 64) dstore(5 + c, (yh_sigmoid(p[5 + c]) - t[5 + c]) * kc);
        }
    }
}

__global__ __launch_bounds__(1024) void loss_final_kernel(const LossArgs a) {
    __shared__ double sh[3][3][16];
    const int t = threadIdx.x, w = t >> 6;
    for (int s = 0; s < 3; ++s) {
        double v[3] = {0.0, 0.0, 0.0};
        for (int b = a.blk_begin[s] + t; b < a.blk_begin[s + 1]; b += 1024)
            for (int k = 0; k < 3; ++k) v[k] += a.part[(size_t)b * 3 + k];
        for (int k = 0; k < 3; ++k) {
            double r = wave_sum_d(v[k]);
            if ((t & 63) == 0) sh[s][k][w] = r;
        }
    }
    __syncthreads();
    if (t == 0) {
        float total = 0.f, tb = 0.f, to = 0.f, tc = 0.f;
        for (int s = 0; s < 3; ++s) {
            if (a.cells[s] == 0) { a.out[4 + 3 * s] = a.out[5 + 3 * s] = a.out[6 + 3 * s] = 0.f; continue; }
            double sb = 0.0, so = 0.0, sc = 0.0;
            for (int w2 = 0; w2 < 16; ++w2) { sb += sh[s][0][w2]; so += sh[s][1][w2]; sc += sh[s][2][w2]; }
            int n = a.counts[s];
            float box = n > 0 ? (float)(sb / (double)n) : 0.f;
            float obj = (float)(so / (double)a.cells[s]);
            float cls = (n > 0 && a.nc > 0) ? (float)(sc / ((double)n * (double)a.nc)) : 0.f;
            float weighted = a.lw[3 * s] * box + a.lw[3 * s + 1] * obj + a.lw[3 * s + 2] * cls;   // train.py:879 / 836
            total += weighted; tb += box; to += obj; tc += cls;
            a.out[4 + 3 * s + 0] = box; a.out[4 + 3 * s + 1] = obj; a.out[4 + 3 * s + 2] = cls;
        }
        a.out[0] = total; a.out[1] = tb; a.out[2] = to; a.out[3] = tc;
    }
}

// ---- standalone decode -------------------------------------------------------------------------
struct Anc3 { float v[6]; };
__global__ void decode_kernel(const float *__restrict__ raw, float *__restrict__ out, const Anc3 ancs,
                              int GH, int GW, int ch, float img, int64_t cells) {
    const float *anc = ancs.v;
    for (int64_t cell = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; cell < cells;
         cell += (int64_t)gridDim.x * blockDim.x) {
        int an = (int)(cell % 3);
        int64_t q = cell / 3;
        int j = (int)(q % GW), i = (int)((q / GW) % GH);
        const float *p = raw + cell * ch;
        float *o = out + cell * ch;
        float sx = yh_sigmoid(p[0]), sy = yh_sigmoid(p[1]), sw = yh_sigmoid(p[2]), sh = yh_sigmoid(p[3]);
        o[0] = ((sx * 2.0f - 0.5f) + (float)j) / (float)GW;
        o[1] = ((sy * 2.0f - 0.5f) + (float)i) / (float)GH;
        float tw = 2.0f * sw, th = 2.0f * sh;
        o[2] = (anc[an * 2 + 0] / img) * (tw * tw);
        o[3] = (anc[an * 2 + 1] / img) * (th * th);
        for (int c = 4; c < ch; ++c) o[c] = p[c];
    }
}
__global__ void decode_bwd_kernel(const float *__restrict__ raw, const float *__restrict__ dout,
                                  float *__restrict__ draw, const Anc3 ancs, int GH, int GW, int ch,
                                  float img, int64_t cells) {
    const float *anc = ancs.v;
    for (int64_t cell = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; cell < cells;
         cell += (int64_t)gridDim.x * blockDim.x) {
        int an = (int)(cell % 3);
        const float *p = raw + cell * ch, *g = dout + cell * ch;
        float *o = draw + cell * ch;
        float sx = yh_sigmoid(p[0]), sy = yh_sigmoid(p[1]), sw = yh_sigmoid(p[2]), sh = yh_sigmoid(p[3]);
        o[0] = g[0] * (2.0f * sx * (1.f - sx) / (float)GW);
        o[1] = g[1] * (2.0f * sy * (1.f - sy) / (float)GH);
        o[2] = g[2] * ((anc[an * 2 + 0] / img) * 8.0f * sw * sw * (1.f - sw));
        o[3] = g[3] * ((anc[an * 2 + 1] / img) * 8.0f * sh * sh * (1.f - sh));
        for (int c = 4; c < ch; ++c) o[c] = g[c];
    }
}

// ---- standalone CIoU (mean over N) ---------------------------------------------------------------
__global__ void ciou_kernel(const float *__restrict__ pred, const float *__restrict__ tgt, float *__restrict__ dpred,
                            int64_t N, float eps, float gscale, double *__restrict__ part) {
    __shared__ double red[4];
    int64_t n = blockIdx.x * (int64_t)256 + threadIdx.x;
    double l = 0.0;
    if (n < N) {
        float p[4] = {pred[4 * n], pred[4 * n + 1], pred[4 * n + 2], pred[4 * n + 3]};
        float t[4] = {tgt[4 * n], tgt[4 * n + 1], tgt[4 * n + 2], tgt[4 * n + 3]}, g[4];
        l = (double)ciou_term(p, t, eps, g);
        if (dpred)
            for (int e = 0; e < 4; ++e) dpred[4 * n + e] = g[e] * gscale / (float)N;
    }
    l = wave_sum_d(l);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void ciou_final_kernel(const double *__restrict__ part, int nblk, int64_t N, float *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += part[b];
        out[0] = (float)(s / (double)N);
    }
}

// ---- eval_epoch's same-cell / same-anchor TP, FP, FN counting (train.py:990-1024) ------------------------
struct EvalArgs {
    const float *pred[3], *tgt[3];
    float anchors[18];
    int grid[3];
    int64_t cells[3];
    int blk_begin[4];
    int nc;
    float img, conf, iou;
    unsigned long long *counts;   // [3] = tp, fp, fn
};

__global__ void eval_counts_kernel(const EvalArgs a) {
    const int s = blockIdx.x >= a.blk_begin[2] ? 2 : (blockIdx.x >= a.blk_begin[1] ? 1 : 0);
    const int64_t cell = (int64_t)(blockIdx.x - a.blk_begin[s]) * 256 + threadIdx.x;
    int tp = 0, fp = 0, fn = 0;
    if (cell < a.cells[s]) {
        const int ch = 5 + a.nc, G = a.grid[s];
        const float *p = a.pred[s] + cell * ch, *t = a.tgt[s] + cell * ch;
        const bool pobj = yh_sigmoid(p[4]) > a.conf, tobj = t[4] > a.conf;
        if (pobj && tobj) {
            int an = (int)(cell % 3);
            int64_t q = cell / 3;
            int j = (int)(q % G), i = (int)((q / G) % G);
            // decode_predictions(pred, anchors) with the default img_size (train.py:993)
            float bx = ((yh_sigmoid(p[0]) * 2.0f - 0.5f) + (float)j) / (float)G;
            float by = ((yh_sigmoid(p[1]) * 2.0f - 0.5f) + (float)i) / (float)G;
            float tw = 2.0f * yh_sigmoid(p[2]), th = 2.0f * yh_sigmoid(p[3]);
            float bw = (a.anchors[(s * 3 + an) * 2 + 0] / a.img) * (tw * tw);
            float bh = (a.anchors[(s * 3 + an) * 2 + 1] / a.img) * (th * th);
            // compute_box_iou (train.py:928-958): centre format, eps 1e-6 in the denominator
            float ax1 = bx - bw / 2, ax2 = bx + bw / 2, ay1 = by - bh / 2, ay2 = by + bh / 2;
            float cx1 = t[0] - t[2] / 2, cx2 = t[0] + t[2] / 2, cy1 = t[1] - t[3] / 2, cy2 = t[1] + t[3] / 2;
            float iw = fmaxf(fminf(ax2, cx2) - fmaxf(ax1, cx1), 0.f), ih = fmaxf(fminf(ay2, cy2) - fmaxf(ay1, cy1), 0.f);
            float inter = iw * ih;
            float uni = (ax2 - ax1) * (ay2 - ay1) + (cx2 - cx1) * (cy2 - cy1) - inter;
            if (inter / (uni + 1e-6f) > a.iou) tp = 1; else fp = 1;
        } else if (pobj) {
            fp = 1;
        } else if (tobj) {
            fn = 1;
        }
    }
    unsigned long long m;
    m = __ballot(tp); if ((threadIdx.x & 63) == 0 && m) atomicAdd(&a.counts[0], (unsigned long long)__popcll(m));
    m = __ballot(fp); if ((threadIdx.x & 63) == 0 && m) atomicAdd(&a.counts[1], (unsigned long long)__popcll(m));
    m = __ballot(fn); if ((threadIdx.x & 63) == 0 && m) atomicAdd(&a.counts[2], (unsigned long long)__popcll(m));
}

int fill_args(LossArgs &a, const float *const pred[3], const float *const target[3], void *const dpred[3],
              const float *anchors, const int grid[3], int B, int nc, float img, const float *loss_w,
              const float *grad_w, float *out, float *ws) {
    YH_REQUIRE(pred && target && anchors && grid && out && ws && B > 0 && nc >= 0 && img > 0.f, "yolo_loss: bad argument");
    int nb = 0;
    for (int s = 0; s < 3; ++s) {
        YH_REQUIRE(grid[s] >= 0 && (grid[s] == 0 || (pred[s] && target[s])), "yolo_loss: scale %d missing", s);
        a.pred[s] = pred[s]; a.tgt[s] = target[s]; a.dpred[s] = dpred ? dpred[s] : nullptr;
        a.grid[s] = grid[s];
        a.cells[s] = (int64_t)B * grid[s] * grid[s] * 3;
        a.blk_begin[s] = nb;
        nb += (int)cdiv64(a.cells[s], 256);
    }
    a.blk_begin[3] = nb;
    for (int k = 0; k < 18; ++k) a.anchors[k] = anchors[k];
    static const float kDefault[9] = {0.05f, 4.0f, 0.5f, 0.05f, 1.0f, 0.5f, 0.05f, 0.4f, 0.5f};   // train.py:865,879
    const float *lw = loss_w ? loss_w : kDefault, *gw = grad_w ? grad_w : lw;
    for (int k = 0; k < 9; ++k) { a.lw[k] = lw[k]; a.gw[k] = gw[k]; }
    a.B = B; a.nc = nc; a.img = img;
    a.counts = (int *)ws;
    a.part = (double *)(ws + 8);
    int *lists = (int *)(ws + 8 + 6 * (size_t)nb);
    for (int s = 0; s < 3; ++s) {
        a.poslist[s] = lists;
        lists += a.cells[s];
    }
    a.out = out;
    return 0;
}

}  // namespace

extern "C" int64_t yh_loss_ws(const int grid[3], int B) {
    int64_t nb = 0, cells = 0;
    for (int s = 0; s < 3; ++s) {
        nb += cdiv64((int64_t)B * grid[s] * grid[s] * 3, 256);
        cells += (int64_t)B * grid[s] * grid[s] * 3;
    }
    return 8 + 6 * nb + cells;      // positive counts | per-workgroup partial sums (double x 3) | positive lists (int32 per cell)
}

extern "C" int yh_yolo_loss(const float *const pred[3], const float *const target[3], float *const dpred[3],
                            const float *anchors, const int grid[3], int B, int nc, float loss_img_size,
                            const float *loss_w, const float *grad_w, float *out, float *ws, void *stream) {
    return yh_yolo_loss_ex(pred, target, (void *const *)dpred, 0, nullptr, anchors, grid, B, nc, loss_img_size, loss_w, grad_w, out, ws,
                           stream);
}

extern "C" int yh_yolo_loss_ex(const float *const pred[3], const float *const target[3], void *const dpred[3], int dpred_bf16,
                               const int dpred_ld[3], const float *anchors, const int grid[3], int B, int nc, float loss_img_size,
                               const float *loss_w, const float *grad_w, float *out, float *ws, void *stream) {
    LossArgs a{};
    int rc = fill_args(a, pred, target, dpred, anchors, grid, B, nc, loss_img_size, loss_w, grad_w, out, ws);
    if (rc) return rc;
    a.dbf16 = dpred_bf16 ? 1 : 0;
    for (int s = 0; s < 3; ++s) {
        a.ldd[s] = dpred_ld ? dpred_ld[s] : 0;
        YH_REQUIRE(a.ldd[s] == 0 || a.ldd[s] >= 3 * (5 + nc), "yolo_loss: dpred_ld[%d]=%d smaller than 3*(5+nc)", s, a.ldd[s]);
    }
    YH_REQUIRE(((uintptr_t)ws & 7) == 0, "yolo_loss: workspace must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    YH_HIP(hipMemsetAsync(a.counts, 0, 4 * sizeof(int), st));
    const int nb = a.blk_begin[3];
    YH_REQUIRE(nb > 0, "yolo_loss: no cells");
    for (int s = 0; s < 3; ++s)
        YH_REQUIRE((int64_t)(a.cells[s] / 3 + 1) * (a.ldd[s] ? a.ldd[s] : 3 * (5 + nc)) < (1ll << 32),
                   "yolo_loss: gradient tensor of scale %d exceeds the 32-bit element index", s);
    hipLaunchKernelGGL(loss_main_kernel, dim3(nb), dim3(256), 0, st, a);      // writes every element of dpred (rows, padding included)
    YH_CHECK_LAUNCH("loss_main");
    if (a.dpred[0] || a.dpred[1] || a.dpred[2]) {
        hipLaunchKernelGGL(loss_pos_kernel, dim3(256), dim3(256), 0, st, a);
        YH_CHECK_LAUNCH("loss_pos");
    }
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1024), 0, st, a);
    YH_CHECK_LAUNCH("loss_final");
    return 0;
}

static int decode_grid(int64_t cells) {
    int64_t g = cdiv64(cells, 256);
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

extern "C" int yh_decode(const float *raw, float *out, const float *anchors3x2, int B, int GH, int GW, int nc,
                         float img_size, void *stream) {
    YH_REQUIRE(raw && out && anchors3x2 && B > 0 && GH > 0 && GW > 0 && nc >= 0 && img_size > 0.f, "decode: bad argument");
    Anc3 an;
    for (int k = 0; k < 6; ++k) an.v[k] = anchors3x2[k];
    int64_t cells = (int64_t)B * GH * GW * 3;
    hipLaunchKernelGGL(decode_kernel, dim3(decode_grid(cells)), dim3(256), 0, (hipStream_t)stream, raw, out, an, GH, GW,
                       5 + nc, img_size, cells);
    YH_CHECK_LAUNCH("decode");
    return 0;
}

extern "C" int yh_decode_bwd(const float *raw, const float *dout, float *draw, const float *anchors3x2, int B, int GH,
                             int GW, int nc, float img_size, void *stream) {
    YH_REQUIRE(raw && dout && draw && anchors3x2 && B > 0 && GH > 0 && GW > 0 && nc >= 0, "decode_bwd: bad argument");
    Anc3 an;
    for (int k = 0; k < 6; ++k) an.v[k] = anchors3x2[k];
    int64_t cells = (int64_t)B * GH * GW * 3;
    hipLaunchKernelGGL(decode_bwd_kernel, dim3(decode_grid(cells)), dim3(256), 0, (hipStream_t)stream, raw, dout, draw, an,
                       GH, GW, 5 + nc, img_size, cells);
    YH_CHECK_LAUNCH("decode_bwd");
    return 0;
}

extern "C" int yh_ciou(const float *pred, const float *tgt, float *dpred, int64_t N, float eps, float grad_scale,
                       float *loss_out, float *ws, void *stream) {
    YH_REQUIRE(pred && tgt && loss_out && ws && N > 0, "ciou: bad argument (N must be > 0)");
    YH_REQUIRE(((uintptr_t)ws & 7) == 0, "ciou: workspace must be 8-byte aligned");
    int nblk = (int)cdiv64(N, 256);
    hipLaunchKernelGGL(ciou_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, pred, tgt, dpred, N, eps, grad_scale,
                       (double *)ws);
    YH_CHECK_LAUNCH("ciou");
    hipLaunchKernelGGL(ciou_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double *)ws, nblk, N, loss_out);
    YH_CHECK_LAUNCH("ciou_final");
    return 0;
}

extern "C" int yh_eval_counts(const float *const pred[3], const float *const target[3], const float *anchors,
                              const int grid[3], int B, int nc, float decode_img_size, float conf_thr, float iou_thr,
                              int64_t *counts, void *stream) {
    YH_REQUIRE(pred && target && anchors && grid && counts && B > 0 && nc >= 0, "eval_counts: bad argument");
    EvalArgs a{};
    int nb = 0;
    for (int s = 0; s < 3; ++s) {
        YH_REQUIRE(grid[s] >= 0 && (grid[s] == 0 || (pred[s] && target[s])), "eval_counts: scale %d missing", s);
        a.pred[s] = pred[s]; a.tgt[s] = target[s]; a.grid[s] = grid[s];
        a.cells[s] = (int64_t)B * grid[s] * grid[s] * 3;
        a.blk_begin[s] = nb;
        nb += (int)cdiv64(a.cells[s], 256);
    }
    a.blk_begin[3] = nb;
    YH_REQUIRE(nb > 0, "eval_counts: no cells");
    for (int k = 0; k < 18; ++k) a.anchors[k] = anchors[k];
    a.nc = nc; a.img = decode_img_size; a.conf = conf_thr; a.iou = iou_thr;
    a.counts = (unsigned long long *)counts;
    hipLaunchKernelGGL(eval_counts_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
    YH_CHECK_LAUNCH("eval_counts");
    return 0;
}
