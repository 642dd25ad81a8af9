// k3_common.h -- what K3 (k3_encode.hip) and the order-search kernel (k3s_search.hip) share: the
// transposed LDS sample image of the fast-path geometries, the exact FIRs over it
// (optimize.c:70-122) and the partition-order clamp (rice.c:148-155).  Internal.
#pragma once

#include "device_util.h"

namespace fhip {
namespace {

constexpr int ENC_WWORDS = 2048;        // 64 Kbit emit window (K3)

// rice.c:148-155 limit_max_partition_order
__device__ __forceinline__ int clamp_porder(int porder, int n, int order)
{
    int lim = ilog2_dev((uint32_t)(n ^ (n - 1)));
    porder = min(porder, lim);
    if (order > 0) porder = min(porder, ilog2_dev((uint32_t)(n / order)));
    return porder;
}

constexpr int HIST = 32;                 // zeroed samples in front of the block

// The block sits in LDS transposed: thread t's run of C samples is column t.
//  * C % 4 != 0: sample i (>= -HIST) is at row (i mod C), column (i div C) + COL0 of
//    a [C][S] int32 image; lanes of a wave touch consecutive words.
//  * C % 4 == 0 (V4): the rows are groups of four samples, [C/4][S] of int4 -- a
//    thread stages its run with C/4 16-byte stores and the FIR fetches its window
//    with 16-byte loads (a quarter of the LDS instructions; lanes touch
//    consecutive 16-byte slots, conflict-free).
// Either way every sample a thread needs at offset c from its run start is at the
// thread's base + a compile-time offset: the window loads carry no address
// arithmetic.  (An fp64 image saves the int->double conversions but its 33 KB cost
// a workgroup per CU: measured 108 vs 96 us.)
template <int C, int T>
struct SmpImg {
    static constexpr bool V4 = (C % 4 == 0);
    static constexpr int CS = V4 ? 4 : 1;             // int32 per column step
    // columns of zeros in front: the FIR looks back 32 samples in tap blocks
    // of 16 (C | 16) or 36 in tap blocks of 9 (C = 3, 9, 18)
    static constexpr int COL0 = (16 % C == 0) ? HIST / C : (36 + C - 1) / C;
    static constexpr int ROWS = V4 ? C / 4 : C;
    static constexpr int S = T + COL0 + (V4 ? 1 : 2);  // row stride in columns
    static constexpr int SIZE = ROWS * S * CS;         // int32
    // int32 index of sample r (0 <= r < C) of column col
    __host__ __device__ static constexpr int at(int col, int r)
    {
        return V4 ? ((r / 4) * S + col) * 4 + (r % 4) : r * S + col;
    }
    // offset of sample (run start of thread t) + c, relative to &img[t * CS]
    __host__ __device__ static constexpr int off(int c)
    {
        return at((c - (((c % C) + C) % C)) / C + COL0, ((c % C) + C) % C);
    }
};

struct FastLds {
    int32_t *smp;                        // SmpImg<C,T>: samples, HIST zeros in front
    unsigned long long *sums;            // [511] heap order
    int32_t *kpar;                       // [511]
    double *coefd;                       // [32] coefficients of the candidate as fp64
    unsigned long long *wtot;            // [16] per-wave totals
    uint32_t *lvl_bits, *lvl_meth;       // [9]
    int32_t *coef;                       // [32]
    int32_t *misc;                       // [16]
    uint32_t *trial;                     // [32]
    uint32_t *bits;                      // [ENC_WWORDS]
};

// Emit window of the fast path, in words: a section of typical density fits one
// window (n/2 words = 16 bits per sample, rounded up to a power of two); denser
// sections take more passes.  Small blocks thus leave LDS for more workgroups.
// wide: 32 bits per sample up to 128 Kbit, for the instance that runs four workgroups per CU
// anyway (MODE 2, VGPRs) on samples wider than 16 bits -- a 24-bit section of ~19 bits per sample
// then takes one pass instead of two.  (MODE 3 had it too while it ran four waves per SIMD; at
// five the 8 KB it costs are a workgroup per CU: configs[3] K3 420 -> 390 us without it.)
__host__ __device__ inline int fast_window_words(int n, bool wide = false)
{
    int w = 256;
    if (wide) { while (w < 2 * ENC_WWORDS && w < n) w <<= 1; return w; }
    while (w < ENC_WWORDS && 2 * w < n) w <<= 1;
    return w;
}
__host__ __device__ inline bool fast_wide_window(int mode, int bps) { return mode == 2 && bps > 16; }

__host__ __device__ inline size_t fast_lds_layout(int n, size_t img_doubles, size_t off[11], bool wide = false)
{
    size_t o = 0;
    off[0] = o; o += 8 * 512;                                   // sums
    off[1] = o; o += 8 * 48;                                    // coefd (zero-padded past 32)
    off[2] = o; o += 8 * 16;                                    // wtot
    off[3] = o; o += 4 * img_doubles;                           // smp image (ints)
    off[4] = o; o += 4 * 512;                                   // kpar
    off[5] = o; o += 4 * 12;                                    // lvl_bits
    off[6] = o; o += 4 * 12;                                    // lvl_meth
    off[7] = o; o += 4 * 32;                                    // coef
    off[8] = o; o += 4 * 16;                                    // misc
    off[9] = o; o += 4 * 32;                                    // trial
    o = (o + 15) & ~(size_t)15;
    off[10] = o; o += 4 * fast_window_words(n, wide);           // bits
    return o;
}

constexpr int clog2(int v) { return v <= 1 ? 0 : 1 + clog2(v >> 1); }          // floor(log2 v)
constexpr int clog2_up(int v) { return clog2(v) + ((v & (v - 1)) ? 1 : 0); }    // ceil(log2 v): runs of 3, 9, 18

template <int C, int T>
struct FastCtx {
    FastLds l;
    int n, i0, tid, lane, wv;
    int obits, precision, pmin_req, pmax_req;
};

// FIR residual of this thread's C samples x[] for an LPC candidate
// (optimize.c:70-122).  l.coefd holds the coefficients as doubles, zero past
// `order`, so the tap loop runs in whole blocks of 8.
template <int C, int T, int OBMAX = 8>
__device__ __forceinline__ void fir_lpc(const FastCtx<C, T> &e, int32_t (&r)[C], int order, int shift)
{
    using Img = SmpImg<C, T>;
    const FastLds &l = e.l;
    const double inv = __builtin_ldexp(1.0, -shift);
    // outputs per register block: a divisor of C (OBMAX = 4: the caller is short of registers)
    constexpr int OB = (C % 8 == 0 && OBMAX >= 8) ? 8 : (C % 4 == 0) ? 4 : (C % 3 == 0) ? 3 : (C % 5 == 0) ? 5 : (C % 7 == 0) ? 7 : 1;
    // taps per block: a multiple of C (going back C*k samples is going back k
    // columns of the image, so every block sees the same immediate offsets)
    constexpr int TB = (16 % C == 0) ? 16 : C * ((8 + C - 1) / C);
    const int32_t *mine = l.smp + e.tid * Img::CS;   // column of this thread's run
#pragma unroll
    for (int ob = 0; ob < C; ob += OB) {
        // keep the register blocks apart: interleaving them only costs VGPRs
        __builtin_amdgcn_sched_barrier(0);
        double acc[OB];
#pragma unroll
        for (int o = 0; o < OB; o++) acc[o] = 0.0;
#pragma unroll 1
        for (int tb = 0; tb < order; tb += TB) {
            const int32_t *base = mine - (tb / C) * Img::CS;
#pragma unroll
            for (int sb = 0; sb < TB; sb += 8) {
                constexpr int dummy = 0; (void)dummy;
                if (order > tb + sb) {
                    // taps tb+sb+1 .. tb+sb+NT_ : samples c = ob+o-(sb+jj+1)
                    const int NT_ = (TB - sb < 8) ? TB - sb : 8;
                    double W[OB + 7];
                    // V4: the window always starts eight samples in front of the block's first
                    // output (a group of four), also where a tap block's last piece is shorter
                    // (runs of 12, 20, 28: TB = C)
                    constexpr int WTOP = Img::V4 ? 7 : -1;       // W index of the sample tap 1 of output 0 reads
                    if constexpr (Img::V4) {
                        // the window starts on a group of four: 16-byte loads
                        static_assert(!Img::V4 || OB % 4 == 0, "aligned windows");
#pragma unroll
                        for (int m4 = 0; m4 < OB + 7; m4 += 4) {
                            const int4 v = *reinterpret_cast<const int4 *>(base + Img::off(ob - sb - 8 + m4));
                            W[m4] = (double)v.x;
                            if (m4 + 1 < OB + 7) W[m4 + 1] = (double)v.y;
                            if (m4 + 2 < OB + 7) W[m4 + 2] = (double)v.z;
                            if (m4 + 3 < OB + 7) W[m4 + 3] = (double)v.w;
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < OB + 7; m++)
                            if (m < OB + NT_ - 1) W[m] = (double)base[Img::off(ob - sb - NT_ + m)];
                    }
                    // taps past the order multiply zeros (the rows are zero-padded).  Stopping the
                    // last piece at the order (a scalar test per tap; 21 % fewer FMAs and coefficient
                    // reads over orders 1 .. 32) measured SLOWER: configs[2]'s search 1.271 -> 1.299 ms
                    // (round 3; the branches break up the block's schedule)
#ifdef FHIP_FIR_TRIM64
                    const int left = order - (tb + sb);
#else
                    constexpr int left = 8;
#endif
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) {
                        if (jj < NT_ && jj < left) {
                            const double cd = l.coefd[tb + sb + jj];
#pragma unroll
                            for (int o = 0; o < OB; o++)
                                acc[o] = __builtin_fma(cd, W[o + (WTOP >= 0 ? WTOP : NT_ - 1) - jj], acc[o]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < OB; o++) {
            // pred >> shift == floor(pred * 2^-shift).  The kernel runs with the
            // fp64 rounding mode "toward -inf" (set_round_down): acc * 2^-shift is
            // exact, |.| < 2^51, so the one rounding of fma(acc, 2^-shift, 1.5 * 2^52)
            // is that floor, and the low mantissa word is the floor's low 32 bits in
            // two's complement.  (int32)(x - (pred >> shift)) only needs those.
            const double z = __builtin_fma(acc[o], inv, 6755399441055744.0);
            const uint32_t qlo = (uint32_t)__double2loint(z);
            uint32_t x;
            if constexpr (Img::V4) {
                const int4 v = *reinterpret_cast<const int4 *>(mine + Img::off(ob + (o & ~3)));   // one load per four
                x = (uint32_t)((o & 3) == 0 ? v.x : (o & 3) == 1 ? v.y : (o & 3) == 2 ? v.z : v.w);
            } else {
                x = (uint32_t)mine[Img::off(ob + o)];
            }
            r[ob + o] = (int32_t)(x - qlo);
        }
    }
    // warm-up samples pass through (optimize.c:84-86): only the first threads
    if (e.i0 < order) {
#pragma unroll
        for (int o = 0; o < C; o++)
            if (e.i0 + o < order) r[o] = mine[Img::off(o)];
    }
}

// The same FIR for orders <= 8 with the coefficients as wave-uniform doubles read
// from K2's compact row by scalar loads (MODE 0): no coefficient traffic through
// LDS, no vector registers for them.
template <int C, int T>
__device__ __forceinline__ void fir_lpc_o8(const FastCtx<C, T> &e, int32_t (&r)[C], int order, int shift,
                                           const double *__restrict__ cd)
{
    using Img = SmpImg<C, T>;
    const double inv = __builtin_ldexp(1.0, -shift);
    constexpr int OB = (C % 8 == 0) ? 8 : (C % 4 == 0) ? 4 : (C % 3 == 0) ? 3 : (C % 5 == 0) ? 5 : (C % 7 == 0) ? 7 : 1;
    const int32_t *mine = e.l.smp + e.tid * Img::CS;
    double cf[8];
#pragma unroll
    for (int jj = 0; jj < 8; jj++) cf[jj] = cd[jj];
#pragma unroll
    for (int ob = 0; ob < C; ob += OB) {
        __builtin_amdgcn_sched_barrier(0);
        double acc[OB];
#pragma unroll
        for (int o = 0; o < OB; o++) acc[o] = 0.0;
        double W[OB + 7];
        if constexpr (Img::V4) {
#pragma unroll
            for (int m4 = 0; m4 < OB + 7; m4 += 4) {
                const int4 v = *reinterpret_cast<const int4 *>(mine + Img::off(ob - 8 + m4));
                W[m4] = (double)v.x;
                if (m4 + 1 < OB + 7) W[m4 + 1] = (double)v.y;
                if (m4 + 2 < OB + 7) W[m4 + 2] = (double)v.z;
                if (m4 + 3 < OB + 7) W[m4 + 3] = (double)v.w;
            }
        } else {
#pragma unroll
            for (int m = 0; m < OB + 7; m++) W[m] = (double)mine[Img::off(ob - 8 + m)];
        }
#pragma unroll
        for (int jj = 0; jj < 8; jj++)
#pragma unroll
            for (int o = 0; o < OB; o++)
                acc[o] = __builtin_fma(cf[jj], W[o + 7 - jj], acc[o]);
#pragma unroll
        for (int o = 0; o < OB; o++) {
            const double z = __builtin_fma(acc[o], inv, 6755399441055744.0);   // floor under round-down, see fir_lpc
            const uint32_t qlo = (uint32_t)__double2loint(z);
            uint32_t x;
            if constexpr (Img::V4) {
                const int4 v = *reinterpret_cast<const int4 *>(mine + Img::off(ob + (o & ~3)));
                x = (uint32_t)((o & 3) == 0 ? v.x : (o & 3) == 1 ? v.y : (o & 3) == 2 ? v.z : v.w);
            } else {
                x = (uint32_t)mine[Img::off(ob + o)];
            }
            r[ob + o] = (int32_t)(x - qlo);
        }
    }
    if (e.i0 < order) {
#pragma unroll
        for (int o = 0; o < C; o++)
            if (e.i0 + o < order) r[o] = mine[Img::off(o)];
    }
}

// Orders 9 .. 16 on a row known up front (MODE 3), the coefficients again as wave-uniform
// doubles from K2's compact row (scalar loads; TAPS = 12 or 16 of them, zero past the order): no
// coefficient reads from the LDS (one per tap and block of eight outputs in fir_lpc), and ONE
// window of TAPS + 7 samples per block of eight outputs instead of one per eight taps.
template <int C, int T, int TAPS>
__device__ __forceinline__ void fir_lpc_o16(const FastCtx<C, T> &e, int32_t (&r)[C], int order, int shift,
                                            const double *__restrict__ cd)
{
    using Img = SmpImg<C, T>;
    static_assert(Img::V4 && C % 8 == 0 && (TAPS == 12 || TAPS == 16), "fir_lpc_o16: runs of 8 or 16");
    // outputs per register block: eight at 12 taps; four at 16 (a window of 24 doubles beside eight accumulators
    // and the run's residuals does not fit the 96 registers of five waves per SIMD: two were spilled)
    constexpr int OBW = (TAPS == 16) ? 4 : 8;
    const double inv = __builtin_ldexp(1.0, -shift);
    const int32_t *mine = e.l.smp + e.tid * Img::CS;
    double cf[TAPS];
#pragma unroll
    for (int jj = 0; jj < TAPS; jj++) cf[jj] = cd[jj];
#pragma unroll
    for (int ob = 0; ob < C; ob += OBW) {
        __builtin_amdgcn_sched_barrier(0);
        double acc[OBW];
#pragma unroll
        for (int o = 0; o < OBW; o++) acc[o] = 0.0;
        double W[TAPS + OBW];                              // samples ob-TAPS .. ob+OBW-1 (the last one unused)
#pragma unroll
        for (int m4 = 0; m4 < TAPS + OBW; m4 += 4) {
            const int4 v = *reinterpret_cast<const int4 *>(mine + Img::off(ob - TAPS + m4));
            W[m4] = (double)v.x; W[m4 + 1] = (double)v.y; W[m4 + 2] = (double)v.z; W[m4 + 3] = (double)v.w;
        }
#pragma unroll
        for (int jj = 0; jj < TAPS; jj++)
#pragma unroll
            for (int o = 0; o < OBW; o++)
                acc[o] = __builtin_fma(cf[jj], W[o + TAPS - 1 - jj], acc[o]);      // tap jj+1: sample ob+o-(jj+1)
#pragma unroll
        for (int o = 0; o < OBW; o++) {
            const double z = __builtin_fma(acc[o], inv, 6755399441055744.0);   // floor under round-down, see fir_lpc
            const uint32_t qlo = (uint32_t)__double2loint(z);
            const int4 v = *reinterpret_cast<const int4 *>(mine + Img::off(ob + (o & ~3)));
            const uint32_t x = (uint32_t)((o & 3) == 0 ? v.x : (o & 3) == 1 ? v.y : (o & 3) == 2 ? v.z : v.w);
            r[ob + o] = (int32_t)(x - qlo);
        }
    }
    if (e.i0 < order) {
#pragma unroll
        for (int o = 0; o < C; o++)
            if (e.i0 + o < order) r[o] = mine[Img::off(o)];
    }
}

template <int C, int T, int NP>
__device__ __forceinline__ void fir_lpc_dotn(const FastCtx<C, T> &e, int32_t (&r)[C], int order, int shift,
                                             const int32_t *__restrict__ cp);
template <int C, int T>
__device__ __forceinline__ void fir_lpc_dot8(const FastCtx<C, T> &e, int32_t (&r)[C], int order, int shift,
                                             const int32_t *__restrict__ cp)
{
    fir_lpc_dotn<C, T, 4>(e, r, order, shift, cp);
}

// Orders <= 8 on a channel whose samples fit 16 bits (K0's narrow rows), when the
// prediction provably stays inside int32 (sum|coef| * 2^magbits < 2^31, checked by
// the caller): v_dot2_i32_i16 does two taps per instruction on int16 pairs and
// costs about what one fp64 FMA does, with no int -> fp64 conversions in front.
// Sample pairs R(k) = (lo: x[k], hi: x[k+1]) are packed from the int32 window;
// cp[j] = (lo: coef of tap 2j+2, hi: coef of tap 2j+1) comes from K2 (scalars).
// NP = int16 pairs per output: 4 for orders <= 8, 8 for orders <= 16 (order searches).
template <int C, int T, int NP>
__device__ __forceinline__ void fir_lpc_dotn(const FastCtx<C, T> &e, int32_t (&r)[C], int order, int shift,
                                             const int32_t *__restrict__ cp)
{
    using Img = SmpImg<C, T>;
    static_assert(Img::V4 && (NP == 4 || NP == 8), "fir_lpc_dotn: runs of whole groups of four");
    typedef short s2 __attribute__((ext_vector_type(2)));
    constexpr int H = 2 * NP;                              // samples of history an output reaches back
    constexpr int OBW = (C % 8 == 0) ? 8 : 4;              // outputs per register block
    const int32_t *mine = e.l.smp + e.tid * Img::CS;
    s2 q[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) q[j] = __builtin_bit_cast(s2, cp[j]);
#pragma unroll
    for (int ob = 0; ob < C; ob += OBW) {
        __builtin_amdgcn_sched_barrier(0);
        int32_t W[H + OBW];                                // samples ob-H .. ob+OBW-1
#pragma unroll
        for (int m4 = 0; m4 < H + OBW; m4 += 4) {
            const int4 v = *reinterpret_cast<const int4 *>(mine + Img::off(ob - H + m4));
            W[m4] = v.x; W[m4 + 1] = v.y; W[m4 + 2] = v.z; W[m4 + 3] = v.w;
        }
        s2 R[H + OBW - 2];                                 // R[m] = (x[ob-H+m], x[ob-H+1+m])
#pragma unroll
        for (int m = 0; m < H + OBW - 2; m++)
            R[m] = __builtin_bit_cast(s2, (int32_t)__builtin_amdgcn_perm((uint32_t)W[m + 1], (uint32_t)W[m], 0x05040100u));
        // taps (2j+1, 2j+2) use x[o-2j-2], x[o-2j-1] = R at window index o + H - 2 - 2j; pairs past
        // the order hold zeros and are skipped (a scalar test per pair)
        int32_t acc[OBW];
#pragma unroll
        for (int o = 0; o < OBW; o++) acc[o] = 0;
#pragma unroll
        for (int j = 0; j < NP; j++) {
#ifndef FHIP_FIR_NO_TRIM
            if (2 * j < order)
#endif
            {
#pragma unroll
                for (int o = 0; o < OBW; o++) acc[o] = __builtin_amdgcn_sdot2(R[o + H - 2 - 2 * j], q[j], acc[o], false);
            }
        }
#pragma unroll
        for (int o = 0; o < OBW; o++) r[ob + o] = (int32_t)((uint32_t)W[H + o] - (uint32_t)(acc[o] >> shift));
    }
    if (e.i0 < order) {
#pragma unroll
        for (int o = 0; o < C; o++)
            if (e.i0 + o < order) r[o] = mine[Img::off(o)];
    }
}

}  // namespace
}  // namespace fhip
