// Direct 3x3 convolution for the NARROW, high-resolution layers (16 -> 16 at 160x160, 16 -> 32 stride 2 at 320x320):
// fp32, NHWC, v_mfma_f32_16x16x4_f32.
//
// These layers carry little arithmetic per byte (24 .. 36 flop/B): they are HBM-bound, and on the 32-wide MFMA tiles of the
// other convolution kernels half of every tile is padding (N = 16).  Here a workgroup owns a TH x 32 patch of output pixels:
//   * the input halo patch is read from HBM ONCE, with 16-byte loads, into LDS laid out [channel quad][pixel][4] -- an A
//     fragment of the 16x16x4 instruction (lane = pixel, k = 4 consecutive channels) is then one conflict-free
//     ds_read_b32 per lane (64 consecutive floats per wave);
//   * the whole filter (9 taps x CIN x COUT) lives in registers as B fragments, loaded once per workgroup;
//   * 16 output pixels x 16 output channels per MFMA tile: no padding for N = 16, the full fp32 MFMA rate;
//   * epilogue: bias, optional accumulate, 64-byte row stores, per-workgroup BatchNorm partial sums (fixed order).
// The same kernel computes backward-data of a stride-1 layer (flipped taps, the backward weight pack).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// Storage type of activations / activation gradients / weight packs: float, or bf16 for the bf16 path (BASELINE configs 3-4).
// The bf16 forms load 8-byte pieces (4 channels), widen them when they are parked in LDS and run the SAME fp32 MFMA loop --
// the layers are bound by HBM bytes and latency, not by the matrix rate, and products of bf16 values are exact in fp32.
// Stored results are rounded to bf16 and the BatchNorm sums are taken over the ROUNDED values, as in conv_bf16.hip.
typedef __bf16 nbf16;
typedef __bf16 nbf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct NarrowIO;
// raw4: four channels as they sit in memory.  A prefetch keeps raw4 registers in flight and widens them when they are parked in
// LDS: widening at the load (the first bf16 form) made every load's result live at once -- `s_waitcnt vmcnt(0)` after each of
// the five loads of a patch, 270 us for the first layer against 167 us in fp32 on twice the bytes.
template <> struct NarrowIO<float> {
    typedef f32x4 raw4;
    static __device__ __forceinline__ raw4 load4raw(const float *p) { return *(const f32x4 *)p; }
    static __device__ __forceinline__ raw4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ f32x4 widen(raw4 v) { return v; }
    static __device__ __forceinline__ float load1(const float *p) { return *p; }
    static __device__ __forceinline__ float store1(float *p, float v) { *p = v; return v; }
    static __device__ __forceinline__ f32x4 load4v(const float *p) { return *(const f32x4 *)p; }             // 16-byte aligned
    static __device__ __forceinline__ f32x4 store4v(float *p, f32x4 v) { *(f32x4 *)p = v; return v; }
    // element (tap, k, n) of a weight pack [tap][K][ld]
    static __device__ __forceinline__ float weight(const float *w, int tap, int k, int n, int K, int kpad, int ld) {
        (void)kpad;
        return w[((size_t)tap * K + k) * ld + n];
    }
};
template <> struct NarrowIO<nbf16> {
    typedef u32x2 raw4;
    static __device__ __forceinline__ raw4 load4raw(const nbf16 *p) { return *(const u32x2 *)p; }
    static __device__ __forceinline__ raw4 zero4() { return u32x2{0u, 0u}; }
    static __device__ __forceinline__ f32x4 widen(raw4 v) {          // bf16 -> fp32 is a 16-bit shift
        return f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
                     __uint_as_float(v[1] & 0xffff0000u)};
    }
    static __device__ __forceinline__ float load1(const nbf16 *p) { return (float)*p; }
    static __device__ __forceinline__ float store1(nbf16 *p, float v) { const nbf16 h = (nbf16)v; *p = h; return (float)h; }
    static __device__ __forceinline__ f32x4 load4v(const nbf16 *p) { return __builtin_convertvector(*(const nbf16x4 *)p, f32x4); }   // 8-byte aligned
    static __device__ __forceinline__ f32x4 store4v(nbf16 *p, f32x4 v) {
        const nbf16x4 h = __builtin_convertvector(v, nbf16x4);
        *(nbf16x4 *)p = h;
        return __builtin_convertvector(h, f32x4);
    }
    // element (tap, k, n) of a bf16 pack [tap][kpad / 8][ld][8] (yh_bf16_pack_multi)
    static __device__ __forceinline__ float weight(const nbf16 *w, int tap, int k, int n, int K, int kpad, int ld) {
        (void)K;
        return (float)w[(((size_t)tap * (kpad >> 3) + (k >> 3)) * ld + n) * 8 + (k & 7)];
    }
};

struct Narrow {
    const float *icoef;             // input prologue table [scale | shift | gate] of the x operand (see yh_prologue), or null
    int icoef_ld;
    const void *in, *w;             // w: [tap][CIN][ldw] (forward pack) or [tap][COUT_of_conv = K][ldw] (backward pack); bf16: see NarrowIO
    const float *bias;
    void *out;
    float *stats;
    int kpad;                       // bf16 packs: padded K rows per tap
    int ldi, ldw, ldo;
    int B, Hi, Wi, Ho, Wo;
    int tiles_x, tiles_y;
    int flip, accumulate;
};

template <int CIN, int COUT, int S, typename T, bool ACT = false>
__global__ __launch_bounds__(256) void narrow_conv_kernel(const Narrow g) {
    typedef NarrowIO<T> IO;
    const T *const gin = (const T *)g.in;
    T *const gout = (T *)g.out;
    constexpr int TW = 32, TH = (S == 1 || CIN == 4) ? 8 : 4;   // output patch
    constexpr int IW = (TW - 1) * S + 3, IH = (TH - 1) * S + 3;
    constexpr int Q = CIN / 4, NT = COUT / 16;
    constexpr int MT = TH * 2 / 4;                        // 16-pixel row segments per wave (TH rows x 2 halves over 4 waves)
    constexpr int NPX = IH * IW, NPC = NPX * Q, NX = (NPC + 255) / 256;
    // stride 2: even and odd input columns are stored as two planes so that the 16 pixels of a fragment stay contiguous
    constexpr int PLANE = S == 1 ? IW : (IW + 1) / 2;
    constexpr int ROWSZ = (S == 1 ? IW : 2 * PLANE);     // pixels per stored input row
    __shared__ __attribute__((aligned(16))) float xs[Q * IH * ROWSZ * 4 + 4 * COUT * 2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 15, kk = lane >> 4;
    const int npatch = g.B * g.tiles_x * g.tiles_y;

    // ---- the filter as B fragments, once per (persistent) workgroup: lane (col, kk) holds w[tap][4q + kk][n0 + col] ---------
    float bw[9 * Q][NT];
#pragma unroll
    for (int tq = 0; tq < 9 * Q; ++tq) {
        const int tap = tq / Q, q = tq - tap * Q;
        const int wt = g.flip ? 8 - tap : tap;
#pragma unroll
        for (int n = 0; n < NT; ++n) bw[tq][n] = IO::weight((const T *)g.w, wt, 4 * q + kk, 16 * n + col, CIN, g.kpad, g.ldw);
    }
    // The MFMA runs TRANSPOSED (weights as the row operand, pixels as the column operand): D register r of lane (col, kk) is
    // channel 16 n + 4 kk + r of pixel `col` -- four consecutive channels of one pixel per lane, so a tile row goes out as ONE
    // 16-byte (bf16: 8-byte) store per lane, 16 pixels x 64 bytes contiguous per instruction (the pixel-major form stored four
    // 4-byte elements per lane and spent a quarter of a patch's cycles on stores and their address arithmetic).
    f32x4 bias4[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[n][r] = g.bias ? g.bias[16 * n + 4 * kk + r] : 0.f;
    const bool vec_ok = (g.ldo & 3) == 0 && ((uintptr_t)g.out & (4 * sizeof(T) - 1)) == 0;      // workgroup-uniform

    // ---- staging plan (fixed for the kernel): piece k of this thread = 16 bytes (patch pixel (py, px), channel quad q);
    // gx = element offset from the patch origin, mx = LDS float offset | py << 16 | px << 24 (py = 255: past the end)
    int gx[NX];
    unsigned mx[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int i = t + 256 * k;
        const int q = i % Q, p = i / Q;
        const int py = p / IW, px = p - py * IW;
        const int sx = S == 1 ? px : (px & 1) * PLANE + (px >> 1);
        gx[k] = (py * g.Wi + px) * g.ldi + 4 * q;
        mx[k] = i < NPC ? (unsigned)(((q * IH + py) * ROWSZ + sx) * 4) | (unsigned)py << 16 | (unsigned)px << 24 : 255u << 16;
    }
    typename IO::raw4 rx[NX];
    // input prologue (ACT): the producer's BatchNorm + SiLU applied when a piece is parked.  Every piece of a thread covers the same
    // channel quad (256 % Q == 0); padding stays zero AFTER the activation: one validity bit per piece, set by fetch
    f32x4 psc, psh, pgt;
    unsigned rok = 0;
    if constexpr (ACT) {
        const int ch = 4 * (t % Q);
        psc = *(const f32x4 *)(g.icoef + ch); psh = *(const f32x4 *)(g.icoef + g.icoef_ld + ch); pgt = *(const f32x4 *)(g.icoef + 2 * g.icoef_ld + ch);
    }
    // patch index (tx, ty, b) of pid, advanced by the grid size without divisions
    struct PIdx { int tx, ty, b; };
    const int gsx = (int)gridDim.x % g.tiles_x, gsr = (int)gridDim.x / g.tiles_x, gsy = gsr % g.tiles_y, gsb = gsr / g.tiles_y;
    auto advance = [&](PIdx &p) {
        p.tx += gsx;
        int c = 0;
        if (p.tx >= g.tiles_x) { p.tx -= g.tiles_x; c = 1; }
        p.ty += gsy + c;
        if (p.ty >= g.tiles_y) { p.ty -= g.tiles_y; p.b += 1; }
        p.b += gsb;
    };
    auto fetch = [&](const PIdx &pi) {
        const int tx = pi.tx, ty = pi.ty, b = pi.b;
        const int iy0 = ty * TH * S - 1, ix0 = tx * TW * S - 1;
        const T *xb = gin + ((ptrdiff_t)(b * g.Hi + iy0) * g.Wi + ix0) * g.ldi;
        rok = 0;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int iy = iy0 + (int)((mx[k] >> 16) & 255u), ix = ix0 + (int)(mx[k] >> 24);
            typename IO::raw4 v = IO::zero4();
            if ((unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi) {
                v = IO::load4raw(xb + gx[k]);
                if constexpr (ACT) rok |= 1u << k;
            }
            rx[k] = v;
        }
    };

    int abase[MT];                                        // LDS float offset of the fragment's pixel `col` at tap (0,0), quad 0
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = wave + 4 * (m >> 1), half = m & 1;
        const int px = (16 * half + col) * S;             // input column of this lane's pixel at tap dx = 0
        const int sx = S == 1 ? px : (px >> 1);           // (even input column -> plane 0)
        abase[m] = ((row * S) * ROWSZ + sx) * 4 + kk;
    }

    // A persistent workgroup walks the patches blockIdx.x, + gridDim.x, ...; the next patch's halo is in flight (registers)
    // while the current one is multiplied and stored.
    f32x4 csum[NT], csq[NT];                              // BatchNorm partial sums of ALL patches of this workgroup: one row
#pragma unroll
    for (int n = 0; n < NT; ++n) csum[n] = csq[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    int pid = blockIdx.x;
    PIdx cur, nxt;
    {
        const int rest = pid / g.tiles_x;
        cur.tx = pid - rest * g.tiles_x;
        cur.ty = rest % g.tiles_y;
        cur.b = rest / g.tiles_y;
    }
    nxt = cur;
    if (pid < npatch) fetch(cur);
    for (; pid < npatch; pid += gridDim.x) {
        __syncthreads();                                  // the previous patch's fragments are consumed
#pragma unroll
        for (int k = 0; k < NX; ++k)
            if (t + 256 * k < NPC) {
                f32x4 v = IO::widen(rx[k]);
                if constexpr (ACT) {
                    if (rok >> k & 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = yh_prologue(v[e], psc[e], psh[e], pgt[e]);
                    }
                }
                *(f32x4 *)(xs + (mx[k] & 0xffffu)) = v;
            }
        __syncthreads();
        advance(nxt);
        if (pid + (int)gridDim.x < npatch) fetch(nxt);

        // ---- multiply: wave w owns output rows {w, w + 4} (8-row patches) or row w, two 16-pixel halves each -----------
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - 3 * dy;
            // stride 2: dx = 0, 2 read the even plane at +0 / +1 pixel, dx = 1 the odd plane
            const int doff = S == 1 ? (dy * ROWSZ + dx) * 4 : (dy * ROWSZ + (dx == 1 ? PLANE : (dx >> 1))) * 4;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                float a[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = xs[q * IH * ROWSZ * 4 + abase[m] + doff];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)      // rows = output channels (the filter fragment), columns = pixels
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[tap * Q + q][n], a[m], acc[m][n], 0, 0, 0);
            }
        }

        // ---- epilogue: D register r of lane (col, kk) = channel 16 n + 4 kk + r of pixel `col` of the segment -------------------
        const int b = cur.b, oy0 = cur.ty * TH, ox0 = cur.tx * TW;
        const bool whole = oy0 + TH <= g.Ho && ox0 + TW <= g.Wo;      // whole patch: no per-pixel tests
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = oy0 + wave + 4 * (m >> 1), ox = ox0 + 16 * (m & 1) + col;
            if (whole || (oy < g.Ho && ox < g.Wo)) {
                T *o = gout + ((size_t)(b * g.Ho + oy) * g.Wo + ox) * g.ldo + 4 * kk;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    f32x4 v = acc[m][n] + bias4[n];
                    if (vec_ok) {
                        if (g.accumulate) v += IO::load4v(o + 16 * n);
                        v = IO::store4v(o + 16 * n, v);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (g.accumulate) v[r] += IO::load1(o + 16 * n + r);
                            v[r] = IO::store1(o + 16 * n + r, v[r]);
                        }
                    }
                    csum[n] += v;
                    csq[n] += v * v;
                }
            }
        }
        cur = nxt;
    }
    if (g.stats) {
        float *red = xs + Q * IH * ROWSZ * 4;                          // [4 waves][COUT][2]
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = csum[n][r], q = csq[n][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }      // the 16 pixels of this lane group
                if (col == 0) {
                    red[(wave * COUT + 16 * n + 4 * kk + r) * 2 + 0] = s;
                    red[(wave * COUT + 16 * n + 4 * kk + r) * 2 + 1] = q;
                }
            }
        __syncthreads();
        if (t < COUT) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s += red[(w * COUT + t) * 2]; q += red[(w * COUT + t) * 2 + 1]; }
            g.stats[((size_t)blockIdx.x * 2 + 0) * COUT + t] = s;
            g.stats[((size_t)blockIdx.x * 2 + 1) * COUT + t] = q;
        }
    }
}

// The first layer (3(4) -> 16, stride 2) with bf16 storage on v_mfma_f32_16x16x32_bf16.  With half the bytes the fp32 16x16x4
// form is bound by its matrix rate (9 instructions of 32 cycles per 16-pixel tile; 165 us for 420 MB); here K = 9 taps x 4
// channels = 36 sits in TWO instructions of 16 cycles: k slot 8 g + j of instruction i (g = lane >> 4) is tap 8 i + 2 g + (j >> 2),
// channel j & 3 -- a lane's A fragment is the 8-byte pixels of two taps (two ds_read_b64 of the raw bf16 patch, parity planes
// as in the fp32 form), taps 9..15 meet zero weights.  Transposed orientation (filter = row operand): a lane ends up with
// four consecutive channels of one pixel, one 8-byte store.  Same patch walk, prefetch and statistics as narrow_conv_kernel.
__global__ __launch_bounds__(256) void narrow_first_bf16_kernel(const Narrow g) {
    typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
    typedef NarrowIO<nbf16> IO;
    const nbf16 *const gin = (const nbf16 *)g.in;
    nbf16 *const gout = (nbf16 *)g.out;
    constexpr int S = 2, TW = 32, TH = 8, COUT = 16;
    constexpr int IW = (TW - 1) * S + 3, IH = (TH - 1) * S + 3, MT = 4;
    constexpr int NPX = IH * IW, NX = (NPX + 255) / 256;
    constexpr int PLANE = (IW + 1) / 2, ROWSZ = 2 * PLANE;                  // even / odd input columns as two planes per row
    __shared__ __attribute__((aligned(16))) unsigned char xs[IH * ROWSZ * 8 + 4 * COUT * 2 * 4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 15, kk = lane >> 4;
    const int npatch = g.B * g.tiles_x * g.tiles_y;

    // filter fragments (row operand): lane (col = output channel, kk) holds, for instruction i, taps 8 i + 2 kk and + 1
    bf16x8v wfr[2];
    int toff[2][2];                                                         // LDS byte offset of those taps' pixels
    unsigned tlive[2][2];                                                   // all-ones / zero: taps 9..15 multiply zeros (not 0 x NaN)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        u32x2 h[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int tap = 8 * i + 2 * kk + e;
            const bool live = tap < 9;
            const int tp = live ? tap : 0;
            h[e] = live ? *(const u32x2 *)((const nbf16 *)g.w + ((size_t)tp * (g.kpad >> 3) * g.ldw + col) * 8) : u32x2{0u, 0u};
            const int dy = tp / 3, dx = tp - 3 * dy;
            toff[i][e] = ((dy * 2 + (dx & 1)) * PLANE + (dx >> 1)) * 8;
            tlive[i][e] = live ? ~0u : 0u;
        }
        typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
        wfr[i] = __builtin_bit_cast(bf16x8v, u32x4v{h[0][0], h[0][1], h[1][0], h[1][1]});
    }
    f32x4 bias4;
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = g.bias ? g.bias[4 * kk + r] : 0.f;
    const bool vec_ok = (g.ldo & 3) == 0 && ((uintptr_t)g.out & 7) == 0;

    // staging plan: piece k of this thread = the 8 bytes of patch pixel (py, px)
    int gx[NX];
    unsigned mx[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int p = t + 256 * k;
        const int py = p / IW, px = p - py * IW;
        gx[k] = (py * g.Wi + px) * g.ldi;
        mx[k] = p < NPX ? (unsigned)(((py * 2 + (px & 1)) * PLANE + (px >> 1)) * 8) | (unsigned)py << 16 | (unsigned)px << 24 : 255u << 16;
    }
    struct PIdx { int tx, ty, b; };
    const int gsx = (int)gridDim.x % g.tiles_x, gsr = (int)gridDim.x / g.tiles_x, gsy = gsr % g.tiles_y, gsb = gsr / g.tiles_y;
    auto advance = [&](PIdx &p) {
        p.tx += gsx;
        int c = 0;
        if (p.tx >= g.tiles_x) { p.tx -= g.tiles_x; c = 1; }
        p.ty += gsy + c;
        if (p.ty >= g.tiles_y) { p.ty -= g.tiles_y; p.b += 1; }
        p.b += gsb;
    };
    u32x2 rx[NX];
    auto fetch = [&](const PIdx &pi) {
        const int iy0 = pi.ty * TH * S - 1, ix0 = pi.tx * TW * S - 1;
        const nbf16 *xb = gin + ((ptrdiff_t)(pi.b * g.Hi + iy0) * g.Wi + ix0) * g.ldi;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int iy = iy0 + (int)((mx[k] >> 16) & 255u), ix = ix0 + (int)(mx[k] >> 24);
            u32x2 v = {0u, 0u};
            if ((unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi) v = *(const u32x2 *)(xb + gx[k]);
            rx[k] = v;
        }
    };
    int abase[MT];                                                          // LDS byte offset of this lane's pixel at tap (0, 0)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = wave + 4 * (m >> 1), half = m & 1;
        abase[m] = ((row * S) * ROWSZ + 16 * half + col) * 8;
    }
    f32x4 csum = {0.f, 0.f, 0.f, 0.f}, csq = {0.f, 0.f, 0.f, 0.f};
    int pid = blockIdx.x;
    PIdx cur, nxt;
    {
        const int rest = pid / g.tiles_x;
        cur.tx = pid - rest * g.tiles_x;
        cur.ty = rest % g.tiles_y;
        cur.b = rest / g.tiles_y;
    }
    nxt = cur;
    if (pid < npatch) fetch(cur);
    for (; pid < npatch; pid += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NX; ++k)
            if (t + 256 * k < NPX) *(u32x2 *)(xs + (mx[k] & 0xffffu)) = rx[k];
        __syncthreads();
        advance(nxt);
        if (pid + (int)gridDim.x < npatch) fetch(nxt);
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x2 p0 = *(const u32x2 *)(xs + abase[m] + toff[i][0]), p1 = *(const u32x2 *)(xs + abase[m] + toff[i][1]);
                typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
                const bf16x8v xf = __builtin_bit_cast(bf16x8v, u32x4v{p0[0] & tlive[i][0], p0[1] & tlive[i][0], p1[0] & tlive[i][1], p1[1] & tlive[i][1]});
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[i], xf, acc[m], 0, 0, 0);
            }
        }
        const int b = cur.b, oy0 = cur.ty * TH, ox0 = cur.tx * TW;
        const bool whole = oy0 + TH <= g.Ho && ox0 + TW <= g.Wo;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = oy0 + wave + 4 * (m >> 1), ox = ox0 + 16 * (m & 1) + col;
            if (whole || (oy < g.Ho && ox < g.Wo)) {
                nbf16 *o = gout + ((size_t)(b * g.Ho + oy) * g.Wo + ox) * g.ldo + 4 * kk;
                f32x4 v = acc[m] + bias4;
                if (vec_ok) {
                    if (g.accumulate) v += IO::load4v(o);
                    v = IO::store4v(o, v);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (g.accumulate) v[r] += IO::load1(o + r);
                        v[r] = IO::store1(o + r, v[r]);
                    }
                }
                csum += v;
                csq += v * v;
            }
        }
        cur = nxt;
    }
    if (g.stats) {
        float *red = (float *)(xs + IH * ROWSZ * 8);                        // [4 waves][COUT][2]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = csum[r], q = csq[r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
            if (col == 0) {
                red[(wave * COUT + 4 * kk + r) * 2 + 0] = s;
                red[(wave * COUT + 4 * kk + r) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (t < COUT) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s += red[(w * COUT + t) * 2]; q += red[(w * COUT + t) * 2 + 1]; }
            g.stats[((size_t)blockIdx.x * 2 + 0) * COUT + t] = s;
            g.stats[((size_t)blockIdx.x * 2 + 1) * COUT + t] = q;
        }
    }
}

// stem[3] (16 -> 32, stride 2) forward with bf16 storage on v_mfma_f32_16x16x32_bf16, same construction as the first layer:
// K = 9 taps x 16 channels = 144 in FIVE instructions per 16-pixel x 16-channel tile (k slot 8 g + j of instruction i = tap
// 2 i + (g >> 1), channel 8 (g & 1) + j; the tenth tap slot is masked), against 36 of the fp32 form.  LDS holds the raw bf16
// patch, 32 bytes per pixel in parity planes; a lane's fragment is one ds_read_b128.
__global__ __launch_bounds__(256) void narrow_s2_16x32_bf16_kernel(const Narrow g) {
    typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    typedef NarrowIO<nbf16> IO;
    const nbf16 *const gin = (const nbf16 *)g.in;
    nbf16 *const gout = (nbf16 *)g.out;
    constexpr int S = 2, TW = 32, TH = 4, COUT = 32, NT = 2, MT = 2;
    constexpr int IW = (TW - 1) * S + 3, IH = (TH - 1) * S + 3;
    constexpr int NPC = IH * IW * 2, NX = (NPC + 255) / 256;                // 16-byte pieces: (pixel, channel octet)
    constexpr int PLANE = (IW + 1) / 2, ROWSZ = 2 * PLANE;
    __shared__ __attribute__((aligned(16))) unsigned char xs[IH * ROWSZ * 32 + 4 * COUT * 2 * 4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 15, kk = lane >> 4;
    const int npatch = g.B * g.tiles_x * g.tiles_y;

    bf16x8v wfr[5][NT];
    int toff[5];
    unsigned tlive[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int tap = 2 * i + (kk >> 1);
        const bool live = tap < 9;
        const int tp = live ? tap : 0;
#pragma unroll
        for (int n = 0; n < NT; ++n)
            wfr[i][n] = __builtin_bit_cast(bf16x8v, live ? *(const u32x4v *)((const nbf16 *)g.w + ((size_t)(tp * (g.kpad >> 3) + (kk & 1)) * g.ldw + 16 * n + col) * 8)
                                                         : u32x4v{0u, 0u, 0u, 0u});
        const int dy = tp / 3, dx = tp - 3 * dy;
        toff[i] = ((dy * 2 + (dx & 1)) * PLANE + (dx >> 1)) * 32 + (kk & 1) * 16;
        tlive[i] = live ? ~0u : 0u;
    }
    f32x4 bias4[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[n][r] = g.bias ? g.bias[16 * n + 4 * kk + r] : 0.f;
    const bool vec_ok = (g.ldo & 3) == 0 && ((uintptr_t)g.out & 7) == 0;

    int gx[NX];
    unsigned mx[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int i = t + 256 * k;
        const int oc = i & 1, p = i >> 1;
        const int py = p / IW, px = p - py * IW;
        gx[k] = (py * g.Wi + px) * g.ldi + 8 * oc;
        mx[k] = i < NPC ? (unsigned)(((py * 2 + (px & 1)) * PLANE + (px >> 1)) * 32 + oc * 16) | (unsigned)py << 16 | (unsigned)px << 24 : 255u << 16;
    }
    struct PIdx { int tx, ty, b; };
    const int gsx = (int)gridDim.x % g.tiles_x, gsr = (int)gridDim.x / g.tiles_x, gsy = gsr % g.tiles_y, gsb = gsr / g.tiles_y;
    auto advance = [&](PIdx &p) {
        p.tx += gsx;
        int c = 0;
        if (p.tx >= g.tiles_x) { p.tx -= g.tiles_x; c = 1; }
        p.ty += gsy + c;
        if (p.ty >= g.tiles_y) { p.ty -= g.tiles_y; p.b += 1; }
        p.b += gsb;
    };
    u32x4v rx[NX];
    auto fetch = [&](const PIdx &pi) {
        const int iy0 = pi.ty * TH * S - 1, ix0 = pi.tx * TW * S - 1;
        const nbf16 *xb = gin + ((ptrdiff_t)(pi.b * g.Hi + iy0) * g.Wi + ix0) * g.ldi;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int iy = iy0 + (int)((mx[k] >> 16) & 255u), ix = ix0 + (int)(mx[k] >> 24);
            u32x4v v = {0u, 0u, 0u, 0u};
            if ((unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi) v = *(const u32x4v *)(xb + gx[k]);
            rx[k] = v;
        }
    };
    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) abase[m] = ((wave * S) * ROWSZ + 16 * m + col) * 32;       // row = wave, half = m
    f32x4 csum[NT], csq[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) csum[n] = csq[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    int pid = blockIdx.x;
    PIdx cur, nxt;
    {
        const int rest = pid / g.tiles_x;
        cur.tx = pid - rest * g.tiles_x;
        cur.ty = rest % g.tiles_y;
        cur.b = rest / g.tiles_y;
    }
    nxt = cur;
    if (pid < npatch) fetch(cur);
    for (; pid < npatch; pid += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NX; ++k)
            if (t + 256 * k < NPC) *(u32x4v *)(xs + (mx[k] & 0xffffu)) = rx[k];
        __syncthreads();
        advance(nxt);
        if (pid + (int)gridDim.x < npatch) fetch(nxt);
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                u32x4v p = *(const u32x4v *)(xs + abase[m] + toff[i]);
                p[0] &= tlive[i]; p[1] &= tlive[i]; p[2] &= tlive[i]; p[3] &= tlive[i];
                const bf16x8v xf = __builtin_bit_cast(bf16x8v, p);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[i][n], xf, acc[m][n], 0, 0, 0);
            }
        const int b = cur.b, oy0 = cur.ty * TH, ox0 = cur.tx * TW;
        const bool whole = oy0 + TH <= g.Ho && ox0 + TW <= g.Wo;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = oy0 + wave, ox = ox0 + 16 * m + col;
            if (whole || (oy < g.Ho && ox < g.Wo)) {
                nbf16 *o = gout + ((size_t)(b * g.Ho + oy) * g.Wo + ox) * g.ldo + 4 * kk;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    f32x4 v = acc[m][n] + bias4[n];
                    if (vec_ok) {
                        if (g.accumulate) v += IO::load4v(o + 16 * n);
                        v = IO::store4v(o + 16 * n, v);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (g.accumulate) v[r] += IO::load1(o + 16 * n + r);
                            v[r] = IO::store1(o + 16 * n + r, v[r]);
                        }
                    }
                    csum[n] += v;
                    csq[n] += v * v;
                }
            }
        }
        cur = nxt;
    }
    if (g.stats) {
        float *red = (float *)(xs + IH * ROWSZ * 32);                       // [4 waves][COUT][2]
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = csum[n][r], q = csq[n][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
                if (col == 0) {
                    red[(wave * COUT + 16 * n + 4 * kk + r) * 2 + 0] = s;
                    red[(wave * COUT + 16 * n + 4 * kk + r) * 2 + 1] = q;
                }
            }
        __syncthreads();
        if (t < COUT) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s += red[(w * COUT + t) * 2]; q += red[(w * COUT + t) * 2 + 1]; }
            g.stats[((size_t)blockIdx.x * 2 + 0) * COUT + t] = s;
            g.stats[((size_t)blockIdx.x * 2 + 1) * COUT + t] = q;
        }
    }
}

// Backward-data of the stride-2 layer (stem[3]: dY 32 channels at 160x160 -> dX 16 channels at 320x320), same ideas.
// dX pixel (y, x) of parity (py, px) = (y & 1, x & 1), a = y >> 1, b = x >> 1, receives
//   py = 0: kh = 1 from dY row a            py = 1: kh = 0 from row a + 1, kh = 2 from row a
//   px = 0: kw = 1 from dY column b         px = 1: kw = 0 from column b + 1, kw = 2 from column b
// i.e. 1, 2, 2 or 4 taps of 32 channels.  An MFMA tile is 16 dX pixels of ONE parity class in one row (x = 2 j + px), so the
// tile shares its taps' weights; the A fragment (lane = j, k = 4 consecutive dY channels) is a conflict-free ds_read_b32 of
// the staged dY patch.  A workgroup owns 8 dX rows x 64 columns; each wave takes one even and one odd row (balanced: 48 + 96
// k-steps).  No zero-stuffed taps, no masked half tiles, every dY element read from HBM once.
template <int KC, int NC, typename T>       // KC = dY channels (32), NC = dX channels (16)
__global__ __launch_bounds__(256) void narrow_dgrad_s2_kernel(const Narrow g) {
    typedef NarrowIO<T> IO;
    const T *const gin = (const T *)g.in;
    T *const gout = (T *)g.out;
    constexpr int TW = 64, TH = 8;                         // dX patch
    constexpr int PW = TW / 2 + 1, PH = TH / 2 + 1;        // dY patch (one extra row / column for the +1 neighbours)
    constexpr int Q = KC / 4;
    __shared__ __attribute__((aligned(16))) float ds[Q * PH * PW * 4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 15, kk = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % g.tiles_x;
    bid /= g.tiles_x;
    const int ty = bid % g.tiles_y, b = bid / g.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;                  // dX origin (even)
    const int a0 = y0 >> 1, b0 = x0 >> 1;                  // dY origin
    // g.Hi, g.Wi: dY size; g.Ho, g.Wo: dX size
    {   // all of this thread's pieces are requested before the first one is widened and parked (one round trip, not six)
        constexpr int NPC = PH * PW * Q, NL = (NPC + 255) / 256;
        typename IO::raw4 rv[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = t + 256 * k;
            const int q = i % Q, p = i / Q;
            const int pr = p / PW, pc = p - pr * PW;
            const int oy = a0 + pr, ox = b0 + pc;
            rv[k] = IO::zero4();
            if (i < NPC && oy < g.Hi && ox < g.Wi) rv[k] = IO::load4raw(gin + ((size_t)(b * g.Hi + oy) * g.Wi + ox) * g.ldi + 4 * q);
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = t + 256 * k;
            const int q = i % Q, p = i / Q;
            const int pr = p / PW, pc = p - pr * PW;
            if (i < NPC) *(f32x4 *)(ds + ((size_t)(q * PH + pr) * PW + pc) * 4) = IO::widen(rv[k]);
        }
    }
    // weights: backward pack wb[tap][co][ldw] (K = co, N = ci); lane (col = ci, kk) holds w[tap][4q + kk][col]
    float bw[9 * Q];
#pragma unroll
    for (int tq = 0; tq < 9 * Q; ++tq) {
        const int tap = tq / Q, q = tq - tap * Q;
        bw[tq] = IO::weight((const T *)g.w, tap, 4 * q + kk, col, KC, g.kpad, g.ldw);
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {                       // this wave's even row (rr = 0) and odd row (rr = 1)
        const int yl = 2 * wave + rr, y = y0 + yl, ar = yl >> 1;     // dY patch row of kh = 1 (even) / kh = 2 (odd)
#pragma unroll
        for (int pxp = 0; pxp < 2; ++pxp)
#pragma unroll
            for (int seg = 0; seg < 2; ++seg) {            // 16 pixels: x = x0 + 2 (16 seg + j) + pxp
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const int jb = 16 * seg + col;             // dY patch column of the "+0" neighbour
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    if ((rr == 0) != (kh == 1)) continue;  // even rows: kh = 1; odd rows: kh = 0, 2
                    const int prow = ar + (kh == 0 ? 1 : 0);
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        if ((pxp == 0) != (kw == 1)) continue;
                        const int pcol = jb + (kw == 0 ? 1 : 0);
                        const float *ap = ds + ((size_t)prow * PW + pcol) * 4 + kk;
#pragma unroll
                        for (int q = 0; q < Q; ++q)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(size_t)q * PH * PW * 4], bw[(kh * 3 + kw) * Q + q], acc, 0, 0, 0);
                    }
                }
                if (y0 + TH <= g.Ho && x0 + TW <= g.Wo) {      // whole patch (workgroup-uniform): no per-element tests
                    T *o = gout + ((size_t)(b * g.Ho + y) * g.Wo + x0 + 2 * (16 * seg + 4 * kk) + pxp) * g.ldo + col;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[r];
                        if (g.accumulate) v += IO::load1(o + (size_t)(2 * r) * g.ldo);
                        IO::store1(o + (size_t)(2 * r) * g.ldo, v);
                    }
                } else if (y < g.Ho) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int x = x0 + 2 * (16 * seg + 4 * kk + r) + pxp;
                        if (x < g.Wo) {
                            T *o = gout + ((size_t)(b * g.Ho + y) * g.Wo + x) * g.ldo + col;
                            float v = acc[r];
                            if (g.accumulate) v += IO::load1(o);
                            IO::store1(o, v);
                        }
                    }
                }
            }
    }
}

// The same layer with bf16 storage.  Halving the bytes (315 MB: 63 us at the HBM rate) leaves the fp32 16x16x4 instruction as
// the bound (144 of them per wave and patch: ~100 us for the layer, 272 us measured), so this form multiplies on
// v_mfma_f32_16x16x32_bf16: the 32 dY channels of a tap are ONE instruction (16 cycles instead of 8 x 32).  LDS holds the dY
// patch as bf16 in four planes of 8 channels, [k-octet][pixel][16 bytes], plane size a multiple of 256 bytes: the 16 rows of
// an A fragment are 16 consecutive 16-byte slots and the four k-octets of a ds_read_b128 lane group fall into disjoint slot
// ranges -- conflict-free.  B fragments: the backward pack [tap][K/8][ld][8] is the fragment layout (one 16-byte load per tap).
__global__ __launch_bounds__(256) void narrow_dgrad_s2_bf16_kernel(const Narrow g) {
    typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    typedef NarrowIO<nbf16> IO;
    const nbf16 *const gin = (const nbf16 *)g.in;
    nbf16 *const gout = (nbf16 *)g.out;
    constexpr int TW = 64, TH = 8, PW = TW / 2 + 1, PH = TH / 2 + 1, NPIX = PH * PW;
    constexpr int PLANE = (NPIX * 16 + 255) / 256 * 256;    // bytes
    __shared__ __attribute__((aligned(16))) unsigned char ds[4 * PLANE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 15, kk = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % g.tiles_x;
    bid /= g.tiles_x;
    const int ty = bid % g.tiles_y, b = bid / g.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;                  // dX origin (even)
    const int a0 = y0 >> 1, b0 = x0 >> 1;                  // dY origin
    {   // g.Hi, g.Wi: dY size.  Piece = (pixel, k-octet): 16 bytes
        constexpr int NPC = NPIX * 4, NL = (NPC + 255) / 256;
        u32x4v rv[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = t + 256 * k;
            const int q = i & 3, p = i >> 2;
            const int pr = p / PW, pc = p - pr * PW;
            const int oy = a0 + pr, ox = b0 + pc;
            rv[k] = u32x4v{0u, 0u, 0u, 0u};
            if (i < NPC && oy < g.Hi && ox < g.Wi) rv[k] = *(const u32x4v *)(gin + ((size_t)(b * g.Hi + oy) * g.Wi + ox) * g.ldi + 8 * q);
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = t + 256 * k;
            if (i < NPC) *(u32x4v *)(ds + (i & 3) * PLANE + (i >> 2) * 16) = rv[k];
        }
    }
    // lane (col = ci, kk) holds w[tap][k = 8 kk .. 8 kk + 7][col]
    bf16x8v bw[9];
    {
        const nbf16 *wp = (const nbf16 *)g.w;
        const int oct = g.kpad >> 3;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            bw[tap] = __builtin_bit_cast(bf16x8v, *(const u32x4v *)(wp + ((size_t)(tap * oct + kk) * g.ldw + col) * 8));
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {                       // this wave's even row (rr = 0) and odd row (rr = 1)
        const int yl = 2 * wave + rr, y = y0 + yl, ar = yl >> 1;
#pragma unroll
        for (int pxp = 0; pxp < 2; ++pxp)
#pragma unroll
            for (int seg = 0; seg < 2; ++seg) {            // 16 pixels: x = x0 + 2 (16 seg + j) + pxp
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const int jb = 16 * seg + col;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    if ((rr == 0) != (kh == 1)) continue;  // even rows: kh = 1; odd rows: kh = 0, 2
                    const int prow = ar + (kh == 0 ? 1 : 0);
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        if ((pxp == 0) != (kw == 1)) continue;
                        const int pcol = jb + (kw == 0 ? 1 : 0);
                        const bf16x8v a = *(const bf16x8v *)(ds + kk * PLANE + (prow * PW + pcol) * 16);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bw[kh * 3 + kw], acc, 0, 0, 0);
                    }
                }
                if (y0 + TH <= g.Ho && x0 + TW <= g.Wo) {      // whole patch (workgroup-uniform): no per-element tests
                    nbf16 *o = gout + ((size_t)(b * g.Ho + y) * g.Wo + x0 + 2 * (16 * seg + 4 * kk) + pxp) * g.ldo + col;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[r];
                        if (g.accumulate) v += IO::load1(o + (size_t)(2 * r) * g.ldo);
                        IO::store1(o + (size_t)(2 * r) * g.ldo, v);
                    }
                } else if (y < g.Ho) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int x = x0 + 2 * (16 * seg + 4 * kk + r) + pxp;
                        if (x < g.Wo) {
                            nbf16 *o = gout + ((size_t)(b * g.Ho + y) * g.Wo + x) * g.ldo + col;
                            float v = acc[r];
                            if (g.accumulate) v += IO::load1(o);
                            IO::store1(o, v);
                        }
                    }
                }
            }
    }
}

// Backward-weight of the narrow layers (16 -> 16 stride 1 at 160x160, stem[3] 16 -> 32 stride 2, stem[0] 3(4) -> 16 stride 2):
//     dW[tap][ci][co] = sum over output pixels p of x[p * S + tap - 1][ci] * dY[p][co]
// The reduction runs over PIXELS, so one v_mfma_f32_16x16x4_f32 takes 4 consecutive output pixels of a row as its k:
// A (rows) = the 16 input channels of one tap (CIN = 16) or 4 taps x 4 channels (CIN = 4), B (columns) = 16 output
// channels.  With x staged [pixel][CIN] and dY staged [16-channel block][pixel][16], both fragments are ONE conflict-free
// ds_read_b32 of 64 consecutive floats (stride 2 keeps even / odd input columns in two planes for that), and a dY fragment
// serves all 9 taps.  A persistent workgroup walks patches of TH x TW output pixels (next patch's 16-byte global loads in
// flight under the MFMAs of the current one), keeps the 9 (x COUT/16) accumulator tiles in registers the whole time and
// writes one raw slab at the end; a fixed-order reduction turns the slabs into OIHW.  Every x and dY element is read once.
struct NarrowW {
    const float *icoef;           // input prologue table of x, or null
    int icoef_ld;
    const void *x, *dy;
    float *ws;
    float *bws;                   // != NULL: per-workgroup column sums of dY (the conv's bias gradient), [grid][COUT]
    int ldx, lddy;
    int B, Hi, Wi, Ho, Wo;
    int tiles_x, tiles_y, npatch;
};

template <int CIN, int COUT, int S> struct NarrowWCfg {
    static constexpr int TH = 8, TW = (S == 1 ? 32 : (CIN == 16 ? 16 : 32));
    static constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    static constexpr int PLANE = (IW + 1) / 2;                         // stride 2: pixels per parity plane of a row
    static constexpr int ROWPX = S == 1 ? IW : 2 * PLANE;              // stored pixels per input row
    static constexpr int QX = CIN / 4, QD = COUT / 4, NB = COUT / 16;
    static constexpr int RG = CIN == 16 ? 9 : 3;                       // MFMA row groups (taps, or groups of 4 taps)
    static constexpr int XPIECES = IH * IW * QX, DPIECES = TH * TW * QD;
    static constexpr int NX = (XPIECES + 255) / 256, ND = (DPIECES + 255) / 256;
    static constexpr int XS = IH * ROWPX * CIN, DS = TH * TW * COUT;   // floats
    static constexpr int SLAB = RG * NB * 256;
};

template <int CIN, int COUT, int S, typename T, bool ACT = false>
__global__ __launch_bounds__(256) void narrow_wgrad_kernel(const NarrowW g) {
    typedef NarrowWCfg<CIN, COUT, S> C;
    typedef NarrowIO<T> IO;
    constexpr int TH = C::TH, TW = C::TW, IH = C::IH, IW = C::IW, PLANE = C::PLANE, ROWPX = C::ROWPX;
    constexpr int QX = C::QX, QD = C::QD, NB = C::NB, RG = C::RG, NX = C::NX, ND = C::ND;
    constexpr int GPR = TW / 4, NGRP = TH * GPR / 4;                   // 4-pixel groups per row / per wave
    constexpr int RED = RG * NB * 4 * 64;                              // floats one wave parks for the cross-wave sum
    constexpr int SMEM = (C::XS + C::DS) > 2 * RED ? (C::XS + C::DS) : 2 * RED;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float *xs = smem, *ds = smem + C::XS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 15, j = lane >> 4;

    // per-thread staging plan, fixed for the whole kernel: piece k of this thread = 16 bytes (pixel (py, px) of the patch,
    // channel quad q); gx/gd = its element offset from the patch origin in global memory, mx/md = LDS float offset | py << 16
    // | px << 24 (py = 255 marks a slot past the end of the patch: its bounds check never passes)
    int gx[NX], gd[ND];
    unsigned mx[NX], md[ND];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int i = t + 256 * k;
        const int q = i % QX, p = i / QX;
        const int py = p / IW, px = p - py * IW;
        const int sx = S == 1 ? px : (px & 1) * PLANE + (px >> 1);
        gx[k] = (py * g.Wi + px) * g.ldx + 4 * q;
        mx[k] = i < C::XPIECES ? (unsigned)((py * ROWPX + sx) * CIN + 4 * q) | (unsigned)py << 16 | (unsigned)px << 24 : 255u << 16;
    }
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        const int i = t + 256 * k;
        const int q = i % QD, p = i / QD;                               // channels 4q..4q+3 of pixel p -> 16-channel block q / 4
        const int py = p / TW, px = p - py * TW;
        gd[k] = (py * g.Wo + px) * g.lddy + 4 * q;
        md[k] = i < C::DPIECES ? (unsigned)(((q >> 2) * TH * TW + p) * 16 + 4 * (q & 3)) | (unsigned)py << 16 | (unsigned)px << 24 : 255u << 16;
    }
    typename IO::raw4 rx[NX], rd[ND];
    f32x4 psc, psh, pgt;                                                // input prologue, as in narrow_conv_kernel
    unsigned rok = 0;
    if constexpr (ACT) {
        const int ch = 4 * (t % QX);
        psc = *(const f32x4 *)(g.icoef + ch); psh = *(const f32x4 *)(g.icoef + g.icoef_ld + ch); pgt = *(const f32x4 *)(g.icoef + 2 * g.icoef_ld + ch);
    }
    auto fetch = [&](int pid) {
        const int tx = pid % g.tiles_x;
        const int rest = pid / g.tiles_x;
        const int ty = rest % g.tiles_y, b = rest / g.tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
        const T *xb = (const T *)g.x + ((ptrdiff_t)(b * g.Hi + iy0) * g.Wi + ix0) * g.ldx;
        const T *db = (const T *)g.dy + ((ptrdiff_t)(b * g.Ho + oy0) * g.Wo + ox0) * g.lddy;
        rok = 0;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int iy = iy0 + (int)((mx[k] >> 16) & 255u), ix = ix0 + (int)(mx[k] >> 24);
            typename IO::raw4 v = IO::zero4();
            if ((unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi) {
                v = IO::load4raw(xb + gx[k]);
                if constexpr (ACT) rok |= 1u << k;
            }
            rx[k] = v;
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int oy = oy0 + (int)((md[k] >> 16) & 255u), ox = ox0 + (int)(md[k] >> 24);
            typename IO::raw4 v = IO::zero4();
            if (oy < g.Ho && ox < g.Wo) v = IO::load4raw(db + gd[k]);
            rd[k] = v;
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int k = 0; k < NX; ++k)
            if (t + 256 * k < C::XPIECES) {
                f32x4 v = IO::widen(rx[k]);
                if constexpr (ACT) {
                    if (rok >> k & 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = yh_prologue(v[e], psc[e], psh[e], pgt[e]);
                    }
                }
                *(f32x4 *)(xs + (mx[k] & 0xffffu)) = v;
            }
#pragma unroll
        for (int k = 0; k < ND; ++k)
            if (t + 256 * k < C::DPIECES) *(f32x4 *)(ds + (md[k] & 0xffffu)) = IO::widen(rd[k]);
    };

    // per-lane LDS offsets (floats) of the A operand relative to the group's first pixel at tap (0, 0)
    int aoff[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        int tap = CIN == 16 ? r : 4 * r + (col >> 2);
        if (tap > 8) tap = 8;                                          // padding rows of the last 4-tap group: computed, never stored
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int pix = S == 1 ? ky * ROWPX + kx : ky * ROWPX + (kx & 1) * PLANE + (kx >> 1);
        aoff[r] = (pix + j) * CIN + (CIN == 16 ? col : (col & 3));
    }

    f32x4 acc[RG][NB];
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[r][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    int pid = blockIdx.x;
    if (pid < g.npatch) fetch(pid);
    for (; pid < g.npatch; pid += gridDim.x) {
        __syncthreads();                                               // the previous patch's fragments are consumed
        if (g.bws) {                                                   // bias gradient: this thread's dY pieces all carry channel quad t % QD
#pragma unroll
            for (int k = 0; k < ND; ++k) bsum += IO::widen(rd[k]);
        }
        park();
        __syncthreads();
        if (pid + (int)gridDim.x < g.npatch) fetch(pid + gridDim.x);   // in flight under the MFMAs below
#pragma unroll 2
        for (int k = 0; k < NGRP; ++k) {
            const int gi = wave + 4 * k;
            const int row = gi / GPR, c0 = 4 * (gi - row * GPR);
            const float *ap = xs + (size_t)((row * S) * ROWPX + c0) * CIN;
            const float *bp = ds + (size_t)(row * TW + c0 + j) * 16 + col;
            float bv[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) bv[n] = bp[n * TH * TW * 16];
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const float a = ap[aoff[r]];
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[n], acc[r][n], 0, 0, 0);
            }
        }
    }

    // ---- (w0 + w2) + (w1 + w3), fixed order, through LDS; wave 0 writes the raw slab ------------------------------------
    __syncthreads();
    float *red = smem;
    auto put = [&](float *dst) {
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[((r * NB + n) * 4 + e) * 64 + lane] = acc[r][n][e];
    };
    auto add = [&](const float *src) {
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[r][n][e] += src[((r * NB + n) * 4 + e) * 64 + lane];
    };
    if (wave >= 2) put(red + (wave - 2) * RED);
    __syncthreads();
    if (wave < 2) add(red + wave * RED);
    __syncthreads();
    if (wave == 1) put(red);
    __syncthreads();
    if (wave == 0) {
        add(red);
        put(g.ws + (size_t)blockIdx.x * C::SLAB);
    }
    if (g.bws) {                                                       // column sums: threads t, t + QD, ... own the same 4 channels
        __syncthreads();
        *(f32x4 *)(smem + 4 * t) = bsum;
        __syncthreads();
        if (t < QD) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            for (int u = t; u < 256; u += QD) a += *(const f32x4 *)(smem + 4 * u);     // fixed order
            *(f32x4 *)(g.bws + (size_t)blockIdx.x * COUT + 4 * t) = a;
        }
    }
}

// dW (OIHW, cin_real input channels) = sum of the raw slabs in workgroup order.  16 outputs per workgroup, 16 partial
// sums each (slab s, s + 16, ...), combined by a fixed tree.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void narrow_wgrad_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dw, int nslab,
                                                                   int cin_real, const float *__restrict__ bws, float *__restrict__ dbias) {
    constexpr int NB = COUT / 16, RG = CIN == 16 ? 9 : 3, SLAB = RG * NB * 256;
    __shared__ float part[16][17];
    const int t = threadIdx.x, e = blockIdx.x * 16 + (t & 15), s0 = t >> 4;
    const int total = COUT * cin_real * 9;
    float v = 0.f;
    if (e < total) {
        const int co = e / (cin_real * 9), rem = e - co * cin_real * 9;
        const int ci = rem / 9, tap = rem - 9 * ci;
        const int r = CIN == 16 ? tap : tap >> 2;
        const int ln = (CIN == 16 ? (ci >> 2) : (tap & 3)) * 16 + (co & 15);
        const int el = CIN == 16 ? (ci & 3) : ci;
        const int idx = ((r * NB + (co >> 4)) * 4 + el) * 64 + ln;
        for (int s = s0; s < nslab; s += 16) v += ws[(size_t)s * SLAB + idx];
    } else if (dbias && e < total + COUT) {                            // the blocks past dW: the bias gradient
        for (int s = s0; s < nslab; s += 16) v += bws[(size_t)s * COUT + (e - total)];
    }
    part[t & 15][s0] = v;
    __syncthreads();
    const int eo = blockIdx.x * 16 + t;
    if (t < 16 && eo < total + (dbias ? COUT : 0)) {
        float a = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) a += part[t][s];
        if (eo < total) dw[eo] = a;
        else dbias[eo - total] = a;
    }
}

// persistent grid: measured best at 2 workgroups per CU for the 16-channel layers and 4 for the first layer (64 x 640 x 640:
// 0.099 / 0.205 / 0.174 ms at 512 / 512 / 1024 workgroups)
int narrow_wgrad_grid(int npatch, int Cin) {
    const int target = Cin == 16 ? 512 : 1024;
    return npatch < target ? npatch : target;
}

template <int CIN, int COUT, int S, typename T>
int narrow_wgrad_launch(NarrowW g, float *dw, float *dbias, int cin_real, hipStream_t st) {
    static_assert(NarrowWCfg<CIN, COUT, S>::NX <= 32 && (256 % (CIN / 4)) == 0, "piece validity mask / per-thread channel quad");
    typedef NarrowWCfg<CIN, COUT, S> C;
    g.tiles_x = cdiv(g.Wo, C::TW);
    g.tiles_y = cdiv(g.Ho, C::TH);
    g.npatch = g.B * g.tiles_x * g.tiles_y;
    const int grid = narrow_wgrad_grid(g.npatch, CIN);
    g.bws = dbias ? g.ws + (size_t)grid * C::SLAB : nullptr;
    if (g.icoef) {
        if constexpr (CIN == 16) hipLaunchKernelGGL((narrow_wgrad_kernel<CIN, COUT, S, T, true>), dim3(grid), dim3(256), 0, st, g);
        else YH_REQUIRE(false, "conv_narrow_bwd_weight: no input prologue for the first layer");
    } else hipLaunchKernelGGL((narrow_wgrad_kernel<CIN, COUT, S, T>), dim3(grid), dim3(256), 0, st, g);
    YH_CHECK_LAUNCH("conv_narrow_bwd_weight");
    hipLaunchKernelGGL((narrow_wgrad_reduce_kernel<CIN, COUT>), dim3(cdiv(COUT * cin_real * 9 + (dbias ? COUT : 0), 16)), dim3(256), 0, st,
                       g.ws, dw, grid, cin_real, g.bws, dbias);
    YH_CHECK_LAUNCH("conv_narrow_bwd_weight_reduce");
    return 0;
}

// persistent grid of the forward / stride-1 backward-data kernel
int narrow_conv_grid(int npatch) {
    const int target = 512;     // 2 per CU: measured 0.162 vs 0.183 ms (first layer), others equal at 1024
    return npatch < target ? npatch : target;
}

int narrow_tiles(int Ho, int Wo, int Cin, int s, int &tx, int &ty) {
    const int TH = (s == 1 || Cin == 4) ? 8 : 4;
    tx = cdiv(Wo, 32);
    ty = cdiv(Ho, TH);
    return tx * ty;
}

}  // namespace

extern "C" int yh_conv_narrow_ok(int Cin, int Cout, int k, int s) {
    return k == 3 && ((Cin == 16 && Cout == 16 && s == 1) || (Cin == 16 && Cout == 32 && s == 2) || (Cin == 4 && Cout == 16 && s == 2)) ? 1 : 0;
}

extern "C" int yh_conv_narrow_blocks(int B, int Hi, int Wi, int Cin, int s) {
    const int Ho = (Hi - 1) / s + 1, Wo = (Wi - 1) / s + 1;
    int tx, ty;
    return narrow_conv_grid(B * narrow_tiles(Ho, Wo, Cin, s, tx, ty));      // one partial row per persistent workgroup
}

extern "C" int yh_conv_narrow_dgrad_s2_ok(int Cin, int Cout) { return Cin == 16 && Cout == 32 ? 1 : 0; }

template <typename T>
static int narrow_dgrad_s2_t(const T *dy, int lddy, const T *wb, int ldwb, int kpad, T *dx, int lddx, int B, int Hi, int Wi, int Cin, int Cout,
                             int accumulate, void *stream) {
    YH_REQUIRE(dy && wb && dx && B > 0 && Hi > 0 && Wi > 0, "conv_narrow_dgrad_s2: bad argument");
    YH_REQUIRE(yh_conv_narrow_dgrad_s2_ok(Cin, Cout), "conv_narrow_dgrad_s2: unsupported shape %d <- %d", Cin, Cout);
    YH_REQUIRE(lddy >= Cout && lddy % 4 == 0 && ((uintptr_t)dy & (4 * sizeof(T) - 1)) == 0 && lddx >= Cin && ldwb >= Cin,
               "conv_narrow_dgrad_s2: views must be addressable in 4-channel pieces");
    Narrow g{};
    g.in = dy; g.w = wb; g.out = dx; g.ldi = lddy; g.ldw = ldwb; g.ldo = lddx; g.B = B; g.kpad = kpad;
    g.Ho = Hi; g.Wo = Wi;                                  // dX
    g.Hi = (Hi - 1) / 2 + 1; g.Wi = (Wi - 1) / 2 + 1;      // dY
    g.accumulate = accumulate ? 1 : 0;
    g.tiles_x = cdiv(Wi, 64); g.tiles_y = cdiv(Hi, 8);
    if constexpr (std::is_same<T, nbf16>::value) {
        // the bf16-MFMA form reads dY in 16-byte pieces and the pack as fragments: K = 32 rows per tap exactly
        if (lddy % 8 == 0 && ((uintptr_t)dy & 15) == 0 && kpad == 32 && ((uintptr_t)wb & 15) == 0) {
            hipLaunchKernelGGL(narrow_dgrad_s2_bf16_kernel, dim3(B * g.tiles_x * g.tiles_y), dim3(256), 0, (hipStream_t)stream, g);
            YH_CHECK_LAUNCH("conv_narrow_dgrad_s2");
            return 0;
        }
    }
    hipLaunchKernelGGL((narrow_dgrad_s2_kernel<32, 16, T>), dim3(B * g.tiles_x * g.tiles_y), dim3(256), 0, (hipStream_t)stream, g);
    YH_CHECK_LAUNCH("conv_narrow_dgrad_s2");
    return 0;
}
extern "C" int yh_conv_narrow_dgrad_s2(const float *dy, int lddy, const float *wb, int ldwb, float *dx, int lddx, int B, int Hi, int Wi,
                                       int Cin, int Cout, int accumulate, void *stream) {
    return narrow_dgrad_s2_t<float>(dy, lddy, wb, ldwb, 0, dx, lddx, B, Hi, Wi, Cin, Cout, accumulate, stream);
}
extern "C" int yh_bf16_conv_narrow_dgrad_s2(const void *dy, int lddy, const void *wb, int ldwb, int kpad, void *dx, int lddx, int B, int Hi,
                                            int Wi, int Cin, int Cout, int accumulate, void *stream) {
    YH_REQUIRE(kpad >= Cout && kpad % 8 == 0, "bf16_conv_narrow_dgrad_s2: bad pack padding");
    return narrow_dgrad_s2_t<nbf16>((const nbf16 *)dy, lddy, (const nbf16 *)wb, ldwb, kpad, (nbf16 *)dx, lddx, B, Hi, Wi, Cin, Cout, accumulate,
                                    stream);
}

template <typename T>
static int narrow_conv_t(const T *x, int ldx, const float *icoef, int icoef_ld, const T *w, int ldw, int kpad, const float *bias, T *y, int ldy,
                         float *bn_partials, int B, int Hi, int Wi, int Cin, int Cout, int s, int flip_taps, int accumulate, void *stream) {
    YH_REQUIRE(x && w && y && B > 0 && Hi > 0 && Wi > 0, "conv_narrow: bad argument");
    YH_REQUIRE(!icoef || (Cin == 16 && !flip_taps && ((uintptr_t)icoef & 15) == 0 && icoef_ld % 4 == 0 && icoef_ld >= Cin),
               "conv_narrow: the input prologue needs a 16-channel forward layer and a 16-byte aligned table");
    YH_REQUIRE(yh_conv_narrow_ok(Cin, Cout, 3, s), "conv_narrow: unsupported shape %d -> %d stride %d", Cin, Cout, s);
    YH_REQUIRE(ldx >= Cin && ldx % 4 == 0 && ((uintptr_t)x & (4 * sizeof(T) - 1)) == 0 && ldy >= Cout && ldw >= Cout,
               "conv_narrow: views must be addressable in 4-channel pieces");
    YH_REQUIRE(!(flip_taps && s != 1), "conv_narrow: flipped taps (backward-data) only for stride 1");
    Narrow g{};
    g.in = x; g.w = w; g.bias = bias; g.out = y; g.stats = bn_partials; g.kpad = kpad; g.icoef = icoef; g.icoef_ld = icoef_ld;
    g.ldi = ldx; g.ldw = ldw; g.ldo = ldy; g.B = B; g.Hi = Hi; g.Wi = Wi;
    g.Ho = (Hi - 1) / s + 1; g.Wo = (Wi - 1) / s + 1;
    g.flip = flip_taps ? 1 : 0; g.accumulate = accumulate ? 1 : 0;
    const int nt = narrow_tiles(g.Ho, g.Wo, Cin, s, g.tiles_x, g.tiles_y);
    hipStream_t st = (hipStream_t)stream;
    const int grid = narrow_conv_grid(B * nt);
    if (icoef) {
        if (s == 1) hipLaunchKernelGGL((narrow_conv_kernel<16, 16, 1, T, true>), dim3(grid), dim3(256), 0, st, g);
        else hipLaunchKernelGGL((narrow_conv_kernel<16, 32, 2, T, true>), dim3(grid), dim3(256), 0, st, g);
    } else if (s == 1) hipLaunchKernelGGL((narrow_conv_kernel<16, 16, 1, T>), dim3(grid), dim3(256), 0, st, g);
    else if (Cin == 4) {
        bool fast = false;
        if constexpr (std::is_same<T, nbf16>::value)      // the bf16-MFMA form reads pixels and filter octets in 8-byte pieces
            fast = ldx % 4 == 0 && ((uintptr_t)x & 7) == 0 && ((uintptr_t)w & 7) == 0 && !flip_taps;
#ifdef YH_WGS_TUNE
        if (getenv("YH_NO_FIRST_BF16")) fast = false;
#endif
        if (fast) hipLaunchKernelGGL(narrow_first_bf16_kernel, dim3(grid), dim3(256), 0, st, g);
        else hipLaunchKernelGGL((narrow_conv_kernel<4, 16, 2, T>), dim3(grid), dim3(256), 0, st, g);
    }
    else {
        bool fast = false;
        if constexpr (std::is_same<T, nbf16>::value)      // the bf16-MFMA form: 16-byte pieces of pixels and of the filter pack
            fast = ldx % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && kpad == 16 && !flip_taps;
        if (fast) hipLaunchKernelGGL(narrow_s2_16x32_bf16_kernel, dim3(grid), dim3(256), 0, st, g);
        else hipLaunchKernelGGL((narrow_conv_kernel<16, 32, 2, T>), dim3(grid), dim3(256), 0, st, g);
    }
    YH_CHECK_LAUNCH("conv_narrow");
    return 0;
}
extern "C" int yh_conv_narrow(const float *x, int ldx, const float *w, int ldw, const float *bias, float *y, int ldy,
                              float *bn_partials, int B, int Hi, int Wi, int Cin, int Cout, int s, int flip_taps, int accumulate,
                              void *stream) {
    return narrow_conv_t<float>(x, ldx, nullptr, 0, w, ldw, 0, bias, y, ldy, bn_partials, B, Hi, Wi, Cin, Cout, s, flip_taps, accumulate, stream);
}
extern "C" int yh_conv_narrow_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *w, int ldw, const float *bias,
                                  float *y, int ldy, float *bn_partials, int B, int Hi, int Wi, int Cin, int Cout, int s, void *stream) {
    return narrow_conv_t<float>(x, ldx, icoef, icoef_ld, w, ldw, 0, bias, y, ldy, bn_partials, B, Hi, Wi, Cin, Cout, s, 0, 0, stream);
}
extern "C" int yh_bf16_conv_narrow(const void *x, int ldx, const void *w, int ldw, int kpad, const float *bias, void *y, int ldy,
                                   float *bn_partials, int B, int Hi, int Wi, int Cin, int Cout, int s, int flip_taps, int accumulate,
                                   void *stream) {
    YH_REQUIRE(kpad >= Cin && kpad % 8 == 0, "bf16_conv_narrow: bad pack padding");
    return narrow_conv_t<nbf16>((const nbf16 *)x, ldx, nullptr, 0, (const nbf16 *)w, ldw, kpad, bias, (nbf16 *)y, ldy, bn_partials, B, Hi, Wi, Cin,
                                Cout, s, flip_taps, accumulate, stream);
}

extern "C" int yh_conv_narrow_bwd_weight_ok(int Cin, int cin_real, int Cout, int k, int s) {
    if (k != 3) return 0;
    if (Cin == 16 && cin_real == 16 && Cout == 16 && s == 1) return 1;
    if (Cin == 16 && cin_real == 16 && Cout == 32 && s == 2) return 1;
    if (Cin == 4 && cin_real >= 1 && cin_real <= 4 && Cout == 16 && s == 2) return 1;
    return 0;
}

extern "C" int64_t yh_conv_narrow_bwd_weight_ws(int B, int Hi, int Wi, int Cin, int Cout, int s) {
    const int Ho = (Hi - 1) / s + 1, Wo = (Wi - 1) / s + 1;
    const int tw = s == 1 ? 32 : (Cin == 16 ? 16 : 32);
    const int64_t npatch = (int64_t)B * cdiv(Wo, tw) * cdiv(Ho, 8);
    const int64_t slab = (int64_t)(Cin == 16 ? 9 : 3) * (Cout / 16) * 256 + Cout;      // + the column sums of dY
    return (int64_t)narrow_wgrad_grid((int)(npatch < (1 << 30) ? npatch : (1 << 30)), Cin) * slab;
}

template <typename T>
static int narrow_bwd_weight_t(const T *x, int ldx, const float *icoef, int icoef_ld, const T *dy, int lddy, float *dw, float *dbias, float *ws,
                               int64_t ws_floats, int B, int Hi, int Wi, int Cin, int cin_real, int Cout, int s, void *stream) {
    YH_REQUIRE(x && dy && dw && ws && B > 0 && Hi > 0 && Wi > 0, "conv_narrow_bwd_weight: bad argument");
    YH_REQUIRE(!icoef || (Cin == 16 && ((uintptr_t)icoef & 15) == 0 && icoef_ld % 4 == 0 && icoef_ld >= Cin),
               "conv_narrow_bwd_weight: the input prologue needs a 16-channel layer and a 16-byte aligned table");
    YH_REQUIRE(yh_conv_narrow_bwd_weight_ok(Cin, cin_real, Cout, 3, s), "conv_narrow_bwd_weight: unsupported shape %d(%d) -> %d stride %d",
               Cin, cin_real, Cout, s);
    YH_REQUIRE(ldx >= Cin && ldx % 4 == 0 && ((uintptr_t)x & (4 * sizeof(T) - 1)) == 0 && lddy >= Cout && lddy % 4 == 0 &&
                   ((uintptr_t)dy & (4 * sizeof(T) - 1)) == 0,
               "conv_narrow_bwd_weight: views must be addressable in 4-channel pieces");
    YH_REQUIRE(ws_floats >= yh_conv_narrow_bwd_weight_ws(B, Hi, Wi, Cin, Cout, s), "conv_narrow_bwd_weight: workspace too small");
    NarrowW g{};
    g.x = x; g.dy = dy; g.ws = ws; g.ldx = ldx; g.lddy = lddy; g.B = B; g.Hi = Hi; g.Wi = Wi; g.icoef = icoef; g.icoef_ld = icoef_ld;
    g.Ho = (Hi - 1) / s + 1; g.Wo = (Wi - 1) / s + 1;
    hipStream_t st = (hipStream_t)stream;
    if (Cin == 16 && s == 1) return narrow_wgrad_launch<16, 16, 1, T>(g, dw, dbias, cin_real, st);
    if (Cin == 16) return narrow_wgrad_launch<16, 32, 2, T>(g, dw, dbias, cin_real, st);
    return narrow_wgrad_launch<4, 16, 2, T>(g, dw, dbias, cin_real, st);
}
extern "C" int yh_conv_narrow_bwd_weight(const float *x, int ldx, const float *dy, int lddy, float *dw, float *dbias, float *ws,
                                         int64_t ws_floats, int B, int Hi, int Wi, int Cin, int cin_real, int Cout, int s, void *stream) {
    return narrow_bwd_weight_t<float>(x, ldx, nullptr, 0, dy, lddy, dw, dbias, ws, ws_floats, B, Hi, Wi, Cin, cin_real, Cout, s, stream);
}
extern "C" int yh_conv_narrow_bwd_weight_act(const float *x, int ldx, const float *icoef, int icoef_ld, const float *dy, int lddy, float *dw,
                                             float *dbias, float *ws, int64_t ws_floats, int B, int Hi, int Wi, int Cin, int cin_real, int Cout,
                                             int s, void *stream) {
    return narrow_bwd_weight_t<float>(x, ldx, icoef, icoef_ld, dy, lddy, dw, dbias, ws, ws_floats, B, Hi, Wi, Cin, cin_real, Cout, s, stream);
}
extern "C" int yh_bf16_conv_narrow_bwd_weight(const void *x, int ldx, const void *dy, int lddy, float *dw, float *dbias, float *ws,
                                              int64_t ws_floats, int B, int Hi, int Wi, int Cin, int cin_real, int Cout, int s,
                                              void *stream) {
    return narrow_bwd_weight_t<nbf16>((const nbf16 *)x, ldx, nullptr, 0, (const nbf16 *)dy, lddy, dw, dbias, ws, ws_floats, B, Hi, Wi, Cin, cin_real,
                                      Cout, s, stream);
}
