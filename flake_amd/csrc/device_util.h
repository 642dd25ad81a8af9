// device_util.h -- what every kernel file of libflakehip.so shares: the phase-stamp
// macros of the diagnostic build, workgroup constants and the small wave-level
// device helpers (folds, Rice counts, DPP scans).  Internal; included by the
// k*.hip files only.  Compile everything with -ffp-contract=off (see k1_autocorr.hip).
#pragma once

#include "kernels.h"

#include <cstdlib>
#include <type_traits>

#ifdef FHIP_STAMPS
// Diagnostic build only (tools/stamps.py): phase time stamps of workgroup 0.
static __device__ long long g_fhip_stamps[64];   // one per kernel file; api.hip merges them
#define STAMP(i)                                                                       \
    do {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                     \
            unsigned long long t_;                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");  \
            g_fhip_stamps[i] = (long long)t_;                                          \
        }                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                             \
    } while (0)
// every stamped kernel file has its own array; it exports a reader under its own name
#define FHIP_DEFINE_STAMP_READER(fn_)                                                  \
    extern "C" __attribute__((visibility("default"))) int fn_(long long *out)          \
    {                                                                                  \
        return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fhip_stamps), sizeof(long long) * 64); \
    }
// accumulating timers (tools/stamps_k1.py): TICK reads the clock, ACCUM adds an
// interval to slot i for the first wave pair of workgroup 0
#define TICK(v_)                                                                       \
    unsigned long long v_;                                                             \
    do {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v_)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                             \
    } while (0)
#define ACCUM(i, a_, b_)                                                               \
    do {                                                                               \
        if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) g_fhip_stamps[i] += (long long)((b_) - (a_)); \
    } while (0)
#define ACC_RESET(lo, hi)                                                              \
    do {                                                                               \
        if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) for (int z_ = lo; z_ < hi; z_++) g_fhip_stamps[z_] = 0; \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#define TICK(v_) do { } while (0)
#define ACCUM(i, a_, b_) do { } while (0)
#define ACC_RESET(lo, hi) do { } while (0)
#endif

namespace fhip {
namespace {

constexpr int NT = 256;        // threads per workgroup (4 waves)
constexpr int WAVE = 64;

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------

// How many units (frames or subframes) a launch really has: the host's number, or -- for the
// ragged batches of a variable-block-size stream, whose piece counts only the device knows
// (k_vbs_plan) -- the word at `dev`; the grid is then sized for the bin's capacity and
// workgroups past the count leave at once.  Wave-uniform (a scalar load).
__device__ __forceinline__ int dev_count(const int32_t *__restrict__ dev, int host)
{
    return dev ? __builtin_amdgcn_readfirstlane(*dev) : host;
}

// kernels.h MultiBin: is unit s (an index into the handle's unit-indexed workspaces) a live unit of
// its bin?  (per lane)
__device__ __forceinline__ bool bin_unit_live(const MultiBin &mb, int s)
{
    // (entries may list the bins in any order -- K1 lists the longest first --: a range test per entry)
    bool live = false;
#pragma unroll
    for (int q = 0; q < 8; q++)
        if (q < mb.nbins && s >= mb.unit0[q] && s - mb.unit0[q] < mb.cap[q]) live = (s - mb.unit0[q]) < mb.cnt[mb.cnt_ix[q]];
    return live;
}

// kernels.h MultiBin: which bin a workgroup belongs to (wave-uniform)
__device__ __forceinline__ int find_bin(const MultiBin &mb, int blk)
{
    int k = 0;
#pragma unroll
    for (int q = 1; q < 8; q++) if (q < mb.nbins && blk >= mb.wg0[q]) k = q;
    return k;
}

__device__ __forceinline__ uint32_t zigzag32(int32_t x)
{
    // rice.c:122 (search side) and bitio.h:128-129 (emit side): same map
    return ((uint32_t)x << 1) ^ (uint32_t)(x >> 31);
}

// bitio.h:128-129, the emit-side fold (v = -2*val-1; v ^= v>>31 in int):
// equal to zigzag32 only for |x| < 2^30, so the emit uses this one.
__device__ __forceinline__ uint32_t emit_fold32(int32_t x)
{
    int32_t v = (int32_t)(0u - 2u * (uint32_t)x - 1u);
    v ^= (v >> 31);
    return (uint32_t)v;
}

// (a << sh) + b and (a ^ b) + c as the one instruction each they are on gfx950; written out where the
// compiler's own association of a longer expression costs instructions (k_order_search's matrix epilogue)
__device__ __forceinline__ uint32_t lshl_add_u32(uint32_t a, uint32_t sh, uint32_t b)
{
    uint32_t d;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(sh), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t add_u32(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t xad_u32(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ int32_t wrap_abs(int32_t a)
{
    return a < 0 ? (int32_t)(0u - (uint32_t)a) : a;
}

// rice.h:48 rice_encode_count evaluated in uint64 like the C macro:
// n*(k+1) is an int, sum-(n>>1) wraps, the shift is logical.
__device__ __forceinline__ uint64_t rice_count64(uint64_t sum, int n, int k)
{
    return (uint64_t)(int64_t)(n * (k + 1)) + ((sum - (uint64_t)(int64_t)(n >> 1)) >> k);
}

// rice.c:30-45 find_optimal_rice_param: first strict minimum over k=0..30 of
// the count truncated to uint32.
__device__ __forceinline__ int rice_best_k(uint64_t sum, int n, uint32_t *bits_out)
{
    const uint64_t s = sum - (uint64_t)(int64_t)(n >> 1);
    uint32_t best = (uint32_t)((uint64_t)(int64_t)n + s);
    int kb = 0;
#pragma unroll 1
    for (int k = 1; k <= 30; k++) {
        uint32_t b = (uint32_t)((uint64_t)(int64_t)(n * (k + 1)) + (s >> k));
        if (b < best) { best = b; kb = k; }
    }
    *bits_out = best;
    return kb;
}

// (wave_sum_u64 / wave_or_u32, below the DPP helpers: reductions without the LDS crossbar)

// inclusive scan of a u64 across the wave
__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long v, int lane)
{
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        unsigned long long t = __shfl_up(v, off, WAVE);
        if (lane >= off) v += t;
    }
    return v;
}

// DPP controls (gfx9): row_shl:n = 0x100+n, row_shr:n = 0x110+n,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    // lanes without a valid source (or outside ROW_MASK) receive 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
}

// wave64 inclusive scan of a u32: 4 in-row steps + 2 row broadcasts, no LDS
__device__ __forceinline__ uint32_t wave_incl_scan_u32_dpp(uint32_t x)
{
    x += dpp_u32<0x111>(x);
    x += dpp_u32<0x112>(x);
    x += dpp_u32<0x114>(x);
    x += dpp_u32<0x118>(x);
    x += dpp_u32<0x142, 0xA>(x);
    x += dpp_u32<0x143, 0xC>(x);
    return x;
}

// lane i receives lane i+D of the same 16-lane row (0 past the row end)
template <int D>
__device__ __forceinline__ unsigned long long row_shl_u64(unsigned long long v)
{
    const uint32_t lo = dpp_u32<0x100 + D>((uint32_t)v);
    const uint32_t hi = dpp_u32<0x100 + D>((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// Wave reductions, valid in lane 0, by DPP inside the 16-lane rows and three lane reads across them.  (Rounds 1-3 used
// shuffles -- ds_bpermute, six dependent trips through the CU's one LDS crossbar per value: K0's four 64-bit sums and
// four ORs were 72 of them per wave, 4600 per CU and launch, and K0 ran at 41 us where a kernel that only moves its bytes
// takes 29, tools/hbm_probe.hip.)
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    v += row_shl_u64<1>(v);
    v += row_shl_u64<2>(v);
    v += row_shl_u64<4>(v);
    v += row_shl_u64<8>(v);
    const int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
#define ROWSUM_(R_) (((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(hi, R_) << 32) | (uint32_t)__builtin_amdgcn_readlane(lo, R_))
    return v + ROWSUM_(16) + ROWSUM_(32) + ROWSUM_(48);      // lane 0: rows 0 + 1 + 2 + 3
#undef ROWSUM_
}

__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v)
{
    v |= dpp_u32<0x101>(v);
    v |= dpp_u32<0x102>(v);
    v |= dpp_u32<0x104>(v);
    v |= dpp_u32<0x108>(v);
    return v | (uint32_t)__builtin_amdgcn_readlane((int)v, 16) | (uint32_t)__builtin_amdgcn_readlane((int)v, 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)v, 48);      // lane 0
}

__device__ __forceinline__ uint32_t wave_xor_u32(uint32_t v)
{
    v ^= dpp_u32<0x101>(v);
    v ^= dpp_u32<0x102>(v);
    v ^= dpp_u32<0x104>(v);
    v ^= dpp_u32<0x108>(v);
    return v ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 16) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 32) ^
           (uint32_t)__builtin_amdgcn_readlane((int)v, 48);      // lane 0
}

// x86-64 cvttsd2si semantics for (int)double: out-of-range and NaN give
// INT_MIN (the reference's `q = error + 0.5`, lpc.c:211).
__device__ __forceinline__ int c_double_to_int(double x)
{
    if (!(x > -2147483649.0 && x < 2147483648.0)) return (int)0x80000000;
    return (int)x;
}

// floor(log2 v), 0 for v = 0 (the reference's log2i)
__device__ __forceinline__ int ilog2_dev(uint32_t v) { return v ? 31 - __clz((int)v) : 0; }

// rice.c:30-45 find_optimal_rice_param without the scan.  With
// S = sum - (n>>1):  f(k) = n(k+1) + (S>>k).
//  * sum < n>>1: S wraps; f(k) = n(k+1) - ceil(d/2^k) (mod 2^32) with
//    d = (n>>1)-sum <= n/2 is increasing, so k = 0.
//  * no wrap and f < 2^32 for all k: f is convex in k (its increment
//    n - ceil((S>>k)/2) never decreases), so the first minimum is the smallest
//    k with (S>>k) <= 2n, capped at 30.
//  * otherwise (sums near 2^32): the reference scan.
__device__ __forceinline__ int rice_k_fast(uint64_t sum, int n, uint32_t *bits_out)
{
    const uint64_t half = (uint64_t)(n >> 1);
    if (sum < half) {
        *bits_out = (uint32_t)n - (uint32_t)(half - sum);
        return 0;
    }
    const uint64_t S = sum - half;
    if (n <= 0 || S >= 0xFFE00000ull) return rice_best_k(sum, n, bits_out);
    const uint32_t two = 2u * (uint32_t)n;
    int k = 0;
    if (S > two) {
        k = (64 - __clzll((long long)S)) - (32 - __clz((int)two));
        if ((S >> k) > two) k++;
        if (k > 30) k = 30;
    }
    *bits_out = (uint32_t)(n * (k + 1)) + (uint32_t)(S >> k);
    return k;
}

// rice_k_fast for a sum that fits 32 bits (every sum of 24-bit audio does): the same three
// cases in 32-bit arithmetic -- the 64-bit shifts and leading-zero counts of the general
// form are several instructions each.  Sums near 2^32 take the general form.
__device__ __forceinline__ int rice_k_fast_u32(uint32_t sum, int n, uint32_t *bits_out)
{
    const uint32_t half = (uint32_t)(n >> 1);
    if (sum < half) {
        *bits_out = (uint32_t)n - (half - sum);
        return 0;
    }
    const uint32_t S = sum - half;
    if (n <= 0 || S >= 0xFFE00000u) return rice_k_fast((uint64_t)sum, n, bits_out);
    const uint32_t two = 2u * (uint32_t)n;
    int k = 0;
    if (S > two) {
        k = __clz((int)two) - __clz((int)S);
        if ((S >> k) > two) k++;
        if (k > 30) k = 30;
    }
    *bits_out = (uint32_t)(n * (k + 1)) + (S >> k);
    return k;
}

// rice_k_fast_u32 without divergent branches, for callers that have ruled out its slow corners wave-wide
// (n <= 0; sum - n/2 >= 0xFFE00000): the three cases as selects.  __clz(0) = 32.
__device__ __forceinline__ int rice_k_u32_nb(uint32_t sum, uint32_t n, uint32_t *bits_out)
{
    const uint32_t half = n >> 1;
    const uint32_t S = sum - half;                   // wraps when sum < half: then nothing below is used
    const uint32_t two = 2u * n;
    int k = (int)__clz((int)two) - (int)__clz((int)S);
    k = max(k, 0);                                   // S <= two: no shift
    k += ((S >> k) > two) ? 1 : 0;
    k = min(k, 30);
    const uint32_t hi = n * (uint32_t)(k + 1) + (S >> k);
    const bool low = sum < half;
    *bits_out = low ? n - (half - sum) : hi;
    return low ? 0 : k;
}

}  // namespace
}  // namespace fhip
