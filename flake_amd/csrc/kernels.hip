// kernels.hip -- gfx950 (CDNA4, wave64) kernels for the FLAC prediction /
// entropy path of libflake.  Compile with -ffp-contract=off: the fp64 stages
// reproduce the reference's rounding sequence exactly (one rounding per
// multiply and per add, in the reference's order), which is what makes the
// quantised coefficients -- and with them every residual -- bit-exact.
//
// Stage map (reference file:line -> kernel):
//   K0 k_prepare   encode.c:541-553 copy_samples, :598-694 stereo estimate +
//                  decorrelation, :558-593 remove_wasted_bits
//   K1 k_autocorr  lpc.c:28-40 apply_welch_window, :46-71 compute_autocorr
//   K2 k_lpc       lpc.c:77-117 Levinson, :125-162 Schur estimate,
//                  :167-219 quantiser, :224-257 driver
//   K3 k_encode    optimize.c:34-276 residuals + order decision tree,
//                  rice.c:30-187 Rice search, encode.c:766-798 +
//                  bitio.h:120-141 residual-section emit
#include "kernels.h"

#include <cstdlib>
#include <type_traits>

#ifdef FHIP_STAMPS
// Diagnostic build only (tools/stamps.py): phase time stamps of workgroup 0.
__device__ long long g_fhip_stamps[64];
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
extern "C" __attribute__((visibility("default"))) int fhip_debug_read_stamps(long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fhip_stamps), sizeof(long long) * 64);
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

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;   // valid in lane 0
}

__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v |= (uint32_t)__shfl_down((int)v, off, WAVE);
    return v;   // valid in lane 0
}

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

// x86-64 cvttsd2si semantics for (int)double: out-of-range and NaN give
// INT_MIN (the reference's `q = error + 0.5`, lpc.c:211).
__device__ __forceinline__ int c_double_to_int(double x)
{
    if (!(x > -2147483649.0 && x < 2147483648.0)) return (int)0x80000000;
    return (int)x;
}

// ---------------------------------------------------------------------------
// K0  k_prepare
// ---------------------------------------------------------------------------
// Stereo frames that do not qualify for the register path (n > 8192 or n % 4):
// one workgroup per frame, both channels resident in LDS (int32[2n]).  Other
// channel counts go to k_prepare_multi.
__global__ __launch_bounds__(NT)
void k_prepare(const int32_t *__restrict__ pcm, int32_t *__restrict__ smp,
               fhip_subframe_info *__restrict__ info, int n, int nch, int bps, int estimate)
{
    extern __shared__ int32_t lds_i32[];
    __shared__ unsigned long long s_sum[4][4];
    __shared__ uint32_t s_or[4][2];
    __shared__ int s_mode;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    if (nch == 2) {
        const int f = blockIdx.x;
        const int32_t *src = pcm + (size_t)f * n * 2;
        int32_t *L = lds_i32, *R = lds_i32 + n;
        const int2 *src2 = reinterpret_cast<const int2 *>(src);
        for (int i = tid; i < n; i += NT) {
            int2 v = src2[i];
            L[i] = v.x;
            R[i] = v.y;
        }
        __syncthreads();

        int mode = FHIP_CH_LEFT_RIGHT;
        if (estimate && n > 32) {
            // encode.c:598-643 calc_decorr_scores
            unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int i = tid + 2; i < n; i += NT) {
                int32_t lt = (int32_t)((uint32_t)L[i] - 2u * (uint32_t)L[i - 1] + (uint32_t)L[i - 2]);
                int32_t rt = (int32_t)((uint32_t)R[i] - 2u * (uint32_t)R[i - 1] + (uint32_t)R[i - 2]);
                int32_t m = (int32_t)((uint32_t)lt + (uint32_t)rt) >> 1;
                int32_t s = (int32_t)((uint32_t)lt - (uint32_t)rt);
                a0 += (unsigned long long)(long long)wrap_abs(lt);
                a1 += (unsigned long long)(long long)wrap_abs(rt);
                a2 += (unsigned long long)(long long)wrap_abs(m);
                a3 += (unsigned long long)(long long)wrap_abs(s);
            }
            a0 = wave_sum_u64(a0); a1 = wave_sum_u64(a1);
            a2 = wave_sum_u64(a2); a3 = wave_sum_u64(a3);
            if (lane == 0) { s_sum[wv][0] = a0; s_sum[wv][1] = a1; s_sum[wv][2] = a2; s_sum[wv][3] = a3; }
            __syncthreads();
            if (tid == 0) {
                unsigned long long cost[4];
                for (int q = 0; q < 4; q++) {
                    unsigned long long sm = s_sum[0][q] + s_sum[1][q] + s_sum[2][q] + s_sum[3][q];
                    uint32_t dummy;
                    int k = rice_best_k(2 * sm, n, &dummy);
                    cost[q] = rice_count64(2 * sm, n, k);     // no 32-bit truncation here (encode.c:620)
                }
                unsigned long long sc[4] = {cost[0] + cost[1], cost[0] + cost[3],
                                            cost[1] + cost[3], cost[2] + cost[3]};
                int best = 0;
                for (int q = 1; q < 4; q++) if (sc[q] < sc[best]) best = q;
                const int modes[4] = {FHIP_CH_LEFT_RIGHT, FHIP_CH_LEFT_SIDE,
                                      FHIP_CH_RIGHT_SIDE, FHIP_CH_MID_SIDE};
                s_mode = modes[best];
            }
            __syncthreads();
            mode = s_mode;
        }

        // encode.c:668-693 apply, then OR of every sample per channel
        uint32_t or0 = 0, or1 = 0;
        for (int i = tid; i < n; i += NT) {
            int32_t a = L[i], b = R[i];
            if (mode == FHIP_CH_MID_SIDE) {
                int32_t mid = (int32_t)((uint32_t)a + (uint32_t)b) >> 1;
                int32_t sd = (int32_t)((uint32_t)a - (uint32_t)b);
                a = mid; b = sd;
            } else if (mode == FHIP_CH_LEFT_SIDE) {
                b = (int32_t)((uint32_t)a - (uint32_t)b);
            } else if (mode == FHIP_CH_RIGHT_SIDE) {
                a = (int32_t)((uint32_t)a - (uint32_t)b);
            }
            L[i] = a; R[i] = b;
            or0 |= (uint32_t)a; or1 |= (uint32_t)b;
        }
        or0 = wave_or_u32(or0); or1 = wave_or_u32(or1);
        if (lane == 0) { s_or[wv][0] = or0; s_or[wv][1] = or1; }
        __syncthreads();

        int wasted[2], obits[2];
        for (int c = 0; c < 2; c++) {
            // encode.c:558-593: min(bps-1, trailing zeros over non-zero samples);
            // bps-1 (also the all-zero case) is reset to 0
            uint32_t o = s_or[0][c] | s_or[1][c] | s_or[2][c] | s_or[3][c];
            int w = o ? min(__ffs((int)o) - 1, bps - 1) : bps - 1;
            if (w == bps - 1) w = 0;
            wasted[c] = w;
            obits[c] = bps - w;
        }
        if (mode == FHIP_CH_MID_SIDE || mode == FHIP_CH_LEFT_SIDE) obits[1]++;
        if (mode == FHIP_CH_RIGHT_SIDE) obits[0]++;

        int32_t *dst = smp + (size_t)f * 2 * n;
        for (int i = tid; i < n; i += NT) {
            dst[i] = L[i] >> wasted[0];
            dst[n + i] = R[i] >> wasted[1];
        }
        if (tid < 2) {
            fhip_subframe_info *o = &info[(size_t)f * 2 + tid];
            o->obits = obits[tid];
            o->wasted = wasted[tid];
            o->ch_mode = mode;
        }
    }
}

// K0 for 1 or 3..8 channels (no decorrelation, encode.c:660-663): one
// workgroup per frame walks it in tiles of 256 sample-frames.  A tile is read
// with coalesced dword loads, transposed through a padded LDS tile, and each
// thread then owns one sample-frame with all its channels in registers.  Pass 1
// ORs every sample per channel (wasted bits, encode.c:558-593), pass 2 re-reads
// (L2), shifts and writes channel rows coalesced.
__global__ __launch_bounds__(NT)
void k_prepare_multi(const int32_t *__restrict__ pcm, int32_t *__restrict__ smp,
                     fhip_subframe_info *__restrict__ info, int n, int nch, int bps)
{
    __shared__ int32_t s_tile[NT * (FHIP_MAX_CH + 1)];
    __shared__ uint32_t s_orr[4][FHIP_MAX_CH];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int f = blockIdx.x;
    const int32_t *src = pcm + (size_t)f * n * nch;
    const int stride = nch + 1;
    const int total = n * nch;

    uint32_t orv[FHIP_MAX_CH];
#pragma unroll
    for (int c = 0; c < FHIP_MAX_CH; c++) orv[c] = 0;
    int wasted[FHIP_MAX_CH];

    for (int pass = 0; pass < 2; pass++) {
        for (int t0 = 0; t0 < n; t0 += NT) {
            const int base = t0 * nch;
            __syncthreads();
            for (int j = 0; j < nch; j++) {
                const int e = base + j * NT + tid;               // linear element of the tile
                const int le = j * NT + tid;
                const int32_t v = src[min(e, total - 1)];
                s_tile[(le / nch) * stride + (le % nch)] = v;
            }
            __syncthreads();
            const int i = t0 + tid;
#pragma unroll
            for (int c = 0; c < FHIP_MAX_CH; c++) {
                if (c < nch) {
                    const int32_t v = s_tile[tid * stride + c];
                    if (pass == 0) orv[c] |= (i < n) ? (uint32_t)v : 0u;
                    else if (i < n) smp[((size_t)f * nch + c) * n + i] = v >> wasted[c];
                }
            }
        }
        if (pass == 0) {
#pragma unroll
            for (int c = 0; c < FHIP_MAX_CH; c++) {
                const uint32_t o = wave_or_u32(orv[c]);
                if (lane == 0) s_orr[wv][c] = o;
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < FHIP_MAX_CH; c++) {
                const uint32_t o = s_orr[0][c] | s_orr[1][c] | s_orr[2][c] | s_orr[3][c];
                int w = o ? min(__ffs((int)o) - 1, bps - 1) : bps - 1;
                if (w == bps - 1) w = 0;
                wasted[c] = w;
            }
            if (tid < nch) {
                fhip_subframe_info *oi = &info[(size_t)f * nch + tid];
                const uint32_t o = s_orr[0][tid] | s_orr[1][tid] | s_orr[2][tid] | s_orr[3][tid];
                int w = o ? min(__ffs((int)o) - 1, bps - 1) : bps - 1;
                if (w == bps - 1) w = 0;
                oi->obits = bps - w;
                oi->wasted = w;
                oi->ch_mode = FHIP_CH_NOT_STEREO;
            }
        }
    }
}

// K0 fast path for stereo frames with n % 4 == 0 and n <= 8192: the frame never
// touches LDS.  Thread t owns the sample-frame quads 4(t + 256m) .. +3,
// m < M: two 16-byte loads per quad (coalesced 32 B per lane), both channels
// stay in registers through the estimate, the decorrelation and the wasted-bits
// shift, and leave as one 16-byte store per channel and quad.  The two
// sample-frames in front of a quad (for the 2nd-order residual) are one more
// 16-byte load that hits L1/L2.
// APPLY = false is the decision pass of the fused pipeline: it writes only
// obits / wasted / ch_mode; the K1 producers then apply them to the PCM they
// load anyway and write smp (k_autocorr_wt<NCH, true>).
template <int M, bool APPLY>
__global__ __launch_bounds__(NT)
// allow_narrow: a channel whose samples (after the shift) all fit 16 bits is stored
// as int16[n] at the start of its row (info.reserved = 1 tells K1's producers and
// K3's staging; K3 resets the field) -- half the bytes written here and read there.
void k_prepare_stereo(const int32_t *__restrict__ pcm, int32_t *__restrict__ smp,
                      fhip_subframe_info *__restrict__ info, int n, int bps, int estimate,
                      int allow_narrow)
{
    __shared__ unsigned long long s_sum[4][4];
    __shared__ uint32_t s_or[4][4];
    __shared__ int s_mode;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int f = blockIdx.x;
    const int4 *src = reinterpret_cast<const int4 *>(pcm + (size_t)f * n * 2);
    const int nquads = n >> 2;

    int32_t L[M][4], R[M][4];
    int4 prev[M];
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int g = tid + NT * m;
        const int gc = min(g, nquads - 1);                 // clamped: loads stay unconditional
        const int4 a = src[2 * gc], b = src[2 * gc + 1];   // (l0 r0 l1 r1) (l2 r2 l3 r3)
        prev[m] = src[max(2 * gc - 1, 0)];                 // (l-2 r-2 l-1 r-1)
        L[m][0] = a.x; R[m][0] = a.y; L[m][1] = a.z; R[m][1] = a.w;
        L[m][2] = b.x; R[m][2] = b.y; L[m][3] = b.z; R[m][3] = b.w;
    }

    int mode = FHIP_CH_LEFT_RIGHT;
    if (estimate && n > 32) {
        // encode.c:598-643 calc_decorr_scores
        unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        if (bps <= 24) {
            // |2nd-order residual| < 2^(bps+2) and never INT_MIN: the 4*M values of a
            // thread add up in 32 bits, no wrap_abs corner, no per-sample predicate
            uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
            for (int m = 0; m < M; m++) {
                const int g = tid + NT * m;
                if (g < nquads) {
                    int32_t l2 = prev[m].x, r2 = prev[m].y, l1 = prev[m].z, r1 = prev[m].w;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int32_t l0 = L[m][q], r0 = R[m][q];
                        int32_t lt = l0 - 2 * l1 + l2;
                        int32_t rt = r0 - 2 * r1 + r2;
                        if (q < 2 && g == 0) { lt = 0; rt = 0; }        // no history for the first two (encode.c:607)
                        const int32_t mm = (lt + rt) >> 1;
                        const int32_t ss = lt - rt;
                        s0 += (uint32_t)max(lt, -lt);
                        s1 += (uint32_t)max(rt, -rt);
                        s2 += (uint32_t)max(mm, -mm);
                        s3 += (uint32_t)max(ss, -ss);
                        l2 = l1; r2 = r1; l1 = l0; r1 = r0;
                    }
                }
            }
            a0 = s0; a1 = s1; a2 = s2; a3 = s3;
        } else {
#pragma unroll
        for (int m = 0; m < M; m++) {
            const int g = tid + NT * m;
            uint32_t l2 = (uint32_t)prev[m].x, r2 = (uint32_t)prev[m].y;
            uint32_t l1 = (uint32_t)prev[m].z, r1 = (uint32_t)prev[m].w;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t l0 = (uint32_t)L[m][q], r0 = (uint32_t)R[m][q];
                const int32_t lt = (int32_t)(l0 - 2u * l1 + l2);
                const int32_t rt = (int32_t)(r0 - 2u * r1 + r2);
                const int32_t mm = (int32_t)((uint32_t)lt + (uint32_t)rt) >> 1;
                const int32_t ss = (int32_t)((uint32_t)lt - (uint32_t)rt);
                const bool on = (g < nquads) && (4 * g + q >= 2);
                a0 += on ? (unsigned long long)(long long)wrap_abs(lt) : 0ull;
                a1 += on ? (unsigned long long)(long long)wrap_abs(rt) : 0ull;
                a2 += on ? (unsigned long long)(long long)wrap_abs(mm) : 0ull;
                a3 += on ? (unsigned long long)(long long)wrap_abs(ss) : 0ull;
                l2 = l1; r2 = r1; l1 = l0; r1 = r0;
            }
        }
        }
        a0 = wave_sum_u64(a0); a1 = wave_sum_u64(a1);
        a2 = wave_sum_u64(a2); a3 = wave_sum_u64(a3);
        if (lane == 0) { s_sum[wv][0] = a0; s_sum[wv][1] = a1; s_sum[wv][2] = a2; s_sum[wv][3] = a3; }
        __syncthreads();
        if (tid < 4) {
            const unsigned long long sm = s_sum[0][tid] + s_sum[1][tid] + s_sum[2][tid] + s_sum[3][tid];
            uint32_t dummy;
            const int k = rice_best_k(2 * sm, n, &dummy);
            s_sum[0][tid] = rice_count64(2 * sm, n, k);       // no 32-bit truncation (encode.c:620)
        }
        __syncthreads();
        {
            const unsigned long long c0 = s_sum[0][0], c1 = s_sum[0][1], c2 = s_sum[0][2], c3 = s_sum[0][3];
            const unsigned long long sc[4] = {c0 + c1, c0 + c3, c1 + c3, c2 + c3};
            int best = 0;
#pragma unroll
            for (int q = 1; q < 4; q++) if (sc[q] < sc[best]) best = q;
            mode = (best == 0) ? FHIP_CH_LEFT_RIGHT : (best == 1) ? FHIP_CH_LEFT_SIDE
                 : (best == 2) ? FHIP_CH_RIGHT_SIDE : FHIP_CH_MID_SIDE;
        }
    }

    // encode.c:668-693 apply, then OR of every sample per channel
    uint32_t or0 = 0, or1 = 0;
    uint32_t mg0 = 0, mg1 = 0;            // OR of x ^ (x >> 31): the magnitude bits in use
#pragma unroll
    for (int m = 0; m < M; m++) {
        const bool on = (tid + NT * m) < nquads;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int32_t a = L[m][q], b = R[m][q];
            if (mode == FHIP_CH_MID_SIDE) {
                const int32_t mid = (int32_t)((uint32_t)a + (uint32_t)b) >> 1;
                const int32_t sd = (int32_t)((uint32_t)a - (uint32_t)b);
                a = mid; b = sd;
            } else if (mode == FHIP_CH_LEFT_SIDE) {
                b = (int32_t)((uint32_t)a - (uint32_t)b);
            } else if (mode == FHIP_CH_RIGHT_SIDE) {
                a = (int32_t)((uint32_t)a - (uint32_t)b);
            }
            L[m][q] = a; R[m][q] = b;
            or0 |= on ? (uint32_t)a : 0u;
            or1 |= on ? (uint32_t)b : 0u;
            mg0 |= on ? (uint32_t)(a ^ (a >> 31)) : 0u;
            mg1 |= on ? (uint32_t)(b ^ (b >> 31)) : 0u;
        }
    }
    or0 = wave_or_u32(or0); or1 = wave_or_u32(or1);
    mg0 = wave_or_u32(mg0); mg1 = wave_or_u32(mg1);
    if (lane == 0) { s_or[wv][0] = or0; s_or[wv][1] = or1; s_or[wv][2] = mg0; s_or[wv][3] = mg1; }
    __syncthreads();

    int wasted[2], obits[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        // encode.c:558-593
        const uint32_t o = s_or[0][c] | s_or[1][c] | s_or[2][c] | s_or[3][c];
        int w = o ? min(__ffs((int)o) - 1, bps - 1) : bps - 1;
        if (w == bps - 1) w = 0;
        wasted[c] = w;
        obits[c] = bps - w;
    }
    if (mode == FHIP_CH_MID_SIDE || mode == FHIP_CH_LEFT_SIDE) obits[1]++;
    if (mode == FHIP_CH_RIGHT_SIDE) obits[0]++;
    bool narrow[2];
    int magbits[2];                       // |x| < 2^magbits for every shifted sample of the channel
#pragma unroll
    for (int c = 0; c < 2; c++) {
        // (x >> w) ^ sign == (x ^ sign) >> w: every shifted sample within int16
        const uint32_t mg = (s_or[0][2 + c] | s_or[1][2 + c] | s_or[2][2 + c] | s_or[3][2 + c]) >> wasted[c];
        narrow[c] = allow_narrow && (mg < 32768u);
        magbits[c] = 32 - __clz((int)mg);
    }

    int4 *dl = reinterpret_cast<int4 *>(smp + (size_t)f * 2 * n);
    int4 *dr = reinterpret_cast<int4 *>(smp + (size_t)f * 2 * n + n);
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int g = tid + NT * m;
        if (APPLY && g < nquads) {
            const int4 vl = make_int4(L[m][0] >> wasted[0], L[m][1] >> wasted[0], L[m][2] >> wasted[0], L[m][3] >> wasted[0]);
            const int4 vr = make_int4(R[m][0] >> wasted[1], R[m][1] >> wasted[1], R[m][2] >> wasted[1], R[m][3] >> wasted[1]);
            if (narrow[0]) reinterpret_cast<int2 *>(dl)[g] = make_int2((vl.x & 0xFFFF) | (vl.y << 16), (vl.z & 0xFFFF) | (vl.w << 16));
            else dl[g] = vl;
            if (narrow[1]) reinterpret_cast<int2 *>(dr)[g] = make_int2((vr.x & 0xFFFF) | (vr.y << 16), (vr.z & 0xFFFF) | (vr.w << 16));
            else dr[g] = vr;
        }
    }
    if (tid < 2) {
        fhip_subframe_info *o = &info[(size_t)f * 2 + tid];
        o->obits = obits[tid];
        o->wasted = wasted[tid];
        o->ch_mode = mode;
        o->reserved = narrow[tid] ? 1 + magbits[tid] : 0;     // 1..16: narrow row, |x| <= 2^(reserved-1)
    }
}

// ---------------------------------------------------------------------------
// K1  k_autocorr
// ---------------------------------------------------------------------------
// The only order-sensitive stage: every autoc[lag] is two running fp64 sums
// that must receive their products one at a time, in position order (SURVEY
// 8-Q1).  Parallelism is across chains only, so the kernel is bound by the
// length of one chain walk (n positions x 4 fp64 ops), not by HBM.
//
// lane = (subframe g, lag pair {2j, 2j+1}): the lane walks the block front to
// back holding four sums (even/odd position x two lags).  Per position it reads
// a = d[p] and b = d[p-2j] from LDS; the odd lag's operand d[p-2j-1] is the
// previous position's b, carried in a register (half the LDS traffic per
// product).  The steady-state loop is nothing but ds_read_b128 / v_mul_f64 /
// v_add_f64 with immediate LDS offsets, software-pipelined by hand.
//
// A workgroup is four INDEPENDENT waves (one per SIMD, so that no two chain
// walks share an issue port); each wave streams its own G subframes through a
// private LDS tile of windowed fp64 samples: 128 new positions per pass behind
// a 32-entry halo.  Waves never exchange data, so ordering is wave-local.
constexpr int AC_TILE = 128;
constexpr int AC_HALO = 32;               // >= FHIP_MAX_ORDER
constexpr int AC_STRIDE = 170;            // row stride in doubles: >= HALO+TILE, 2*S mod 64 = 20
constexpr int AC_GMAX = 12;
constexpr int AC_PER_LANE = AC_TILE / WAVE;
constexpr int AC_CH = 8;                  // positions per software-pipeline stage
constexpr int AC_WAVES = 4;

// LDS is only shared inside one wave here: order its accesses without a
// workgroup barrier.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(AC_WAVES * WAVE)
void k_autocorr(const int32_t *__restrict__ smp, double *__restrict__ autoc,
                int nsub, int n, int maxlag, int G, int nl2, double c)
{
    __shared__ double s_buf[AC_WAVES][AC_GMAX * AC_STRIDE];

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    double *buf = s_buf[wv];
    const int s0 = (blockIdx.x * AC_WAVES + wv) * G;
    if (s0 >= nsub) return;                       // whole wave idle (uniform)
    const int half = n >> 1;
    const int ntiles = (n + AC_TILE - 1) / AC_TILE;

    // Unconditional, clamped loads (a load under a per-element condition makes
    // hipcc branch around it and wait vmcnt(0) each time); out-of-range rows and
    // positions read a valid address and are zeroed when the tile is written.
    const int32_t *rowp[AC_GMAX];
#pragma unroll
    for (int q = 0; q < AC_GMAX; q++) rowp[q] = smp + (size_t)min(s0 + q, nsub - 1) * n;
    int32_t cur[AC_GMAX][AC_PER_LANE];

    const int g = lane / nl2, j = lane - g * nl2;
    const int L0 = 2 * j, L1 = 2 * j + 1;
    const bool chain = (g < G) && (s0 + g < nsub) && (L0 <= maxlag);
    const double *rowa = buf + (chain ? g : 0) * AC_STRIDE + AC_HALO;    // &d[tile base]
    const double *pb = rowa - (chain ? L0 : 0);
    double accE0 = 1.0, accO0 = 1.0, accE1 = 1.0, accO1 = 1.0;   // lpc.c:58-59
    double b1 = 0.0;                                              // d[p-1-2j], carried

    auto issue_loads = [&](int tb) {
#pragma unroll
        for (int q = 0; q < AC_GMAX; q++)
#pragma unroll
            for (int u = 0; u < AC_PER_LANE; u++)
                cur[q][u] = rowp[q][min(tb + u * WAVE + lane, n - 1)];
    };

    for (int idx = lane; idx < AC_GMAX * AC_HALO; idx += WAVE)
        buf[(idx >> 5) * AC_STRIDE + (idx & 31)] = 0.0;
    issue_loads(0);

    for (int t = 0; t < ntiles; t++) {
        const int tb = t * AC_TILE;
        // ---- window the tile into LDS (lpc.c:28-40: weight of positions i and
        //      n-1-i is 1-(c-i)^2), then fetch the next one -------------------
#pragma unroll
        for (int u = 0; u < AC_PER_LANE; u++) {
            const int p = tb + u * WAVE + lane;
            const int ii = (p < half) ? p : (n - 1 - p);
            const bool valid = (p < n) && (ii < half);
            const double tt = c - (double)ii;
            const double w = valid ? (1.0 - (tt * tt)) : 0.0;
#pragma unroll
            for (int q = 0; q < AC_GMAX; q++)
                buf[q * AC_STRIDE + AC_HALO + u * WAVE + lane] = (double)cur[q][u] * w;
        }
        issue_loads(tb + AC_TILE);
        wave_lds_fence();

        const int kend = min(AC_TILE, n - tb);
        if (t > 0 && kend == AC_TILE) {
            // steady state: tb is even, so even k <-> even position.  Three-deep
            // software pipeline, written out by hand: while stage c's 16 products
            // are added to the four sums, stage c+1's products are formed and
            // stage c+2's operands are read from LDS.  Adds and multiplies
            // alternate so that consecutive adds into the same sum are 8
            // instructions apart (a dependent fp64 add issues every ~8 cycles, an
            // independent one every ~4).
            constexpr int NS = AC_TILE / AC_CH;
            double A[AC_CH], B[AC_CH], P[2 * AC_CH];
            double b1n;                               // b1 for the stage in A/B
            auto rd = [&](int st, double (&a)[AC_CH], double (&b)[AC_CH]) {
#pragma unroll
                for (int u = 0; u < AC_CH; u++) { a[u] = rowa[st * AC_CH + u]; b[u] = pb[st * AC_CH + u]; }
            };
            // products of one stage: P[2u] = a[u]*b[u] (even lag), P[2u+1] = a[u]*prev b (odd lag)
            rd(0, A, B);
            {
                double pbv = b1;
#pragma unroll
                for (int u = 0; u < AC_CH; u++) { P[2 * u] = A[u] * B[u]; P[2 * u + 1] = A[u] * pbv; pbv = B[u]; }
                b1n = pbv;
            }
            rd(1, A, B);
#pragma unroll
            for (int cidx = 0; cidx < NS; cidx++) {
                double An[AC_CH], Bn[AC_CH], Pn[2 * AC_CH];
                if (cidx + 2 < NS) rd(cidx + 2, An, Bn);
                __builtin_amdgcn_sched_barrier(0);
                double pbv = b1n;
#pragma unroll
                for (int u = 0; u < AC_CH; u += 2) {
                    // position u is even, u+1 odd
                    accE0 = accE0 + P[2 * u];
                    if (cidx + 1 < NS) Pn[2 * u] = A[u] * B[u];
                    accE1 = accE1 + P[2 * u + 1];
                    if (cidx + 1 < NS) Pn[2 * u + 1] = A[u] * pbv;
                    accO0 = accO0 + P[2 * u + 2];
                    if (cidx + 1 < NS) Pn[2 * u + 2] = A[u + 1] * B[u + 1];
                    accO1 = accO1 + P[2 * u + 3];
                    if (cidx + 1 < NS) Pn[2 * u + 3] = A[u + 1] * B[u];
                    pbv = B[u + 1];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (cidx + 1 < NS) {
                    b1n = pbv;
#pragma unroll
                    for (int u = 0; u < 2 * AC_CH; u++) P[u] = Pn[u];
                }
                if (cidx + 2 < NS) {
#pragma unroll
                    for (int u = 0; u < AC_CH; u++) { A[u] = An[u]; B[u] = Bn[u]; }
                }
            }
            b1 = pb[AC_TILE - 1];
        } else {
            // first tile (head rule, lpc.c:60-61) and a ragged last tile
            double accH0 = 1.0, accH1 = 1.0;
            for (int k = 0; k < kend; k++) {
                const int p = tb + k;
                const double a = rowa[k], b0 = pb[k];
                const double p0 = a * b0, p1 = a * b1;
                if (p <= maxlag) {
                    // positions lag..maxlag all go to the first sum
                    if (p >= L0) accH0 = accH0 + p0;
                    if (p >= L1) accH1 = accH1 + p1;
                    if (p == maxlag) {
                        // ... which then continues at position maxlag+1 (lpc.c:63-66)
                        if ((maxlag + 1) & 1) { accO0 = accH0; accO1 = accH1; }
                        else { accE0 = accH0; accE1 = accH1; }
                    }
                } else if (p & 1) {
                    accO0 = accO0 + p0;
                    accO1 = accO1 + p1;
                } else {
                    accE0 = accE0 + p0;
                    accE1 = accE1 + p1;
                }
                b1 = b0;
            }
        }
        wave_lds_fence();
        // the last 32 entries of this tile become the halo of the next
        {
            double hv[(AC_GMAX * AC_HALO) / WAVE];
#pragma unroll
            for (int r = 0; r < (AC_GMAX * AC_HALO) / WAVE; r++) {
                const int idx = lane + r * WAVE;
                hv[r] = buf[(idx >> 5) * AC_STRIDE + AC_TILE + (idx & 31)];
            }
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < (AC_GMAX * AC_HALO) / WAVE; r++) {
                const int idx = lane + r * WAVE;
                buf[(idx >> 5) * AC_STRIDE + (idx & 31)] = hv[r];
            }
        }
    }
    // lpc.c:68: autoc = temp + temp2.  The reference's padded product with
    // d[len] = 0 adds +-0.0 to a sum that is never -0.0, so it is skipped.
    if (chain) {
        double *dst = autoc + (size_t)(s0 + g) * FHIP_MAX_LAGS;
        dst[L0] = accE0 + accO0;
        if (L1 <= maxlag) dst[L1] = accE1 + accO1;
    }
}

// ---------------------------------------------------------------------------
// K1 (small batches)  k_autocorr_ps -- parity-split chains
// ---------------------------------------------------------------------------
// A chain walk in k_autocorr is n positions long whatever the batch size, and
// with few subframes most SIMDs hold one wave or none.  Here each running sum
// gets its own lane: lane = (subframe g, lag group {l0, l0+2, l0+4}, parity pi)
// walks only the positions of parity pi, so a walk is n/2 steps of three
// products.  The operands of the two higher lags are the b of the previous two
// steps (d[p-2-l0], d[p-4-l0]), carried in registers.  The LDS tile is stored
// de-interleaved (even positions / odd positions) so that consecutive steps of a
// lane read consecutive doubles.  The head rule (lpc.c:60-61) sends positions
// lag..maxlag of every parity to the sum of parity (maxlag+1)&1, in order, before
// that lane's own walk starts.  autoc = sum(pi=0) + sum(pi=1) (lpc.c:68) joins
// the two lanes at the end.  launch_autocorr picks this kernel when it needs
// fewer fp64 issue slots per SIMD than k_autocorr (e.g. LPC-8 at 4096 frames:
// 1024 waves x 12288 ops instead of 683..1024 x 16384).
constexpr int PS_HALF = AC_TILE / 2;      // steps per tile and parity
constexpr int PS_HH = AC_HALO / 2;        // halo entries per parity array
constexpr int PS_ROW = PS_HH + PS_HALF;   // 80 doubles per parity array
constexpr int PS_STRIDE = 2 * PS_ROW + 10;   // per subframe (two arrays + bank spread)
constexpr int PS_GMAX = 8;
constexpr int PS_CH = 8;                  // steps per software-pipeline stage

__global__ __launch_bounds__(AC_WAVES * WAVE)
void k_autocorr_ps(const int32_t *__restrict__ smp, double *__restrict__ autoc,
                   int nsub, int n, int maxlag, int G, int lps, int ge, double c)
{
    __shared__ double s_buf[AC_WAVES][PS_GMAX * PS_STRIDE];

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    double *buf = s_buf[wv];
    const int s0 = (blockIdx.x * AC_WAVES + wv) * G;
    if (s0 >= nsub) return;
    const int half = n >> 1;
    const int ntiles = (n + AC_TILE - 1) / AC_TILE;

    const int32_t *rowp[PS_GMAX];
#pragma unroll
    for (int q = 0; q < PS_GMAX; q++) rowp[q] = smp + (size_t)min(s0 + q, nsub - 1) * n;
    int32_t cur[PS_GMAX][AC_PER_LANE];

    // lane -> (subframe, lag group, parity)
    const int g = lane / lps, ql = lane - g * lps;
    const int pi = ql & 1, grp = ql >> 1;
    const int l0 = (grp < ge) ? 6 * grp : 1 + 6 * (grp - ge);
    const bool chain = (g < G) && (s0 + g < nsub) && (l0 <= maxlag);
    const bool ok1 = l0 + 2 <= maxlag, ok2 = l0 + 4 <= maxlag;
    const int pib = pi ^ (l0 & 1);                      // parity array that holds d[p - l0]
    const int sft = (l0 + pib - pi) / 2;                // index shift inside that array
    const double *rowA = buf + (chain ? g : 0) * PS_STRIDE + pi * PS_ROW + PS_HH;        // a = rowA[t]
    const double *rowB = buf + (chain ? g : 0) * PS_STRIDE + pib * PS_ROW + PS_HH - (chain ? sft : 0);
    const int pih = (maxlag + 1) & 1;                   // parity whose sum owns the head
    double S0 = 1.0, S1 = 1.0, S2 = 1.0;                // lpc.c:58-59
    double b1 = 0.0, b2 = 0.0;                          // d[p-2-l0], d[p-4-l0], carried

    auto issue_loads = [&](int tb) {
#pragma unroll
        for (int q = 0; q < PS_GMAX; q++)
#pragma unroll
            for (int u = 0; u < AC_PER_LANE; u++)
                cur[q][u] = rowp[q][min(tb + u * WAVE + lane, n - 1)];
    };
    // position tb + x sits in parity array (x & 1) at index PS_HH + x/2
    auto slot = [&](int q, int x) { return q * PS_STRIDE + (x & 1) * PS_ROW + PS_HH + (x >> 1); };

    for (int idx = lane; idx < PS_GMAX * 2 * PS_HH; idx += WAVE) {
        const int q = idx / (2 * PS_HH), r = idx - q * 2 * PS_HH;
        buf[q * PS_STRIDE + (r / PS_HH) * PS_ROW + (r % PS_HH)] = 0.0;
    }
    issue_loads(0);

    for (int t = 0; t < ntiles; t++) {
        const int tb = t * AC_TILE;
        // ---- window the tile into LDS (lpc.c:28-40), de-interleaved by parity ----
#pragma unroll
        for (int u = 0; u < AC_PER_LANE; u++) {
            const int x = u * WAVE + lane;
            const int p = tb + x;
            const int ii = (p < half) ? p : (n - 1 - p);
            const bool valid = (p < n) && (ii < half);
            const double tt = c - (double)ii;
            const double w = valid ? (1.0 - (tt * tt)) : 0.0;
#pragma unroll
            for (int q = 0; q < PS_GMAX; q++) buf[slot(q, x)] = (double)cur[q][u] * w;
        }
        issue_loads(tb + AC_TILE);
        wave_lds_fence();

        const int kend = min(AC_TILE, n - tb);            // positions in this tile
        if (t > 0 && kend == AC_TILE) {
            // steady state: PS_HALF steps, 3 products each, software-pipelined
            constexpr int NS = PS_HALF / PS_CH;
            double A[PS_CH], B[PS_CH];
#pragma unroll
            for (int u = 0; u < PS_CH; u++) { A[u] = rowA[u]; B[u] = rowB[u]; }
#pragma unroll
            for (int st = 0; st < NS; st++) {
                double An[PS_CH], Bn[PS_CH];
                if (st + 1 < NS) {
#pragma unroll
                    for (int u = 0; u < PS_CH; u++) {
                        An[u] = rowA[(st + 1) * PS_CH + u];
                        Bn[u] = rowB[(st + 1) * PS_CH + u];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < PS_CH; u++) {
                    const double p0 = A[u] * B[u], p1 = A[u] * b1, p2 = A[u] * b2;
                    S0 = S0 + p0;
                    S1 = S1 + p1;
                    S2 = S2 + p2;
                    b2 = b1;
                    b1 = B[u];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (st + 1 < NS) {
#pragma unroll
                    for (int u = 0; u < PS_CH; u++) { A[u] = An[u]; B[u] = Bn[u]; }
                }
            }
        } else {
            if (t == 0 && pi == pih) {
                // head: positions lag..maxlag of BOTH parities, in order, into this
                // lane's sums (tile 0 holds them all: maxlag <= 32 < AC_TILE)
                const int hend = min(maxlag, kend - 1);
                for (int x = 0; x <= hend; x++) {
                    const double a = buf[slot(chain ? g : 0, x)];
                    if (x >= l0) {
                        const double p0 = a * buf[slot(chain ? g : 0, x - l0)];
                        S0 = S0 + p0;
                    }
                    if (ok1 && x >= l0 + 2) {
                        const double p1 = a * buf[slot(chain ? g : 0, x - l0 - 2)];
                        S1 = S1 + p1;
                    }
                    if (ok2 && x >= l0 + 4) {
                        const double p2 = a * buf[slot(chain ? g : 0, x - l0 - 4)];
                        S2 = S2 + p2;
                    }
                }
            }
            // own walk through this tile: positions of parity pi above maxlag
            for (int st = 0; st < PS_HALF; st++) {
                const int x = 2 * st + pi;
                const int p = tb + x;
                if (x >= kend) break;
                const double a = rowA[st], b0 = rowB[st];
                if (p > maxlag) {
                    // operands straight from LDS in this slow path (p - l0 - 4 >= tb - 32)
                    const double c1 = (x - l0 - 2 >= -AC_HALO) ? buf[slot(chain ? g : 0, x - l0 - 2 + AC_HALO) - PS_HH] : 0.0;
                    const double c2 = (x - l0 - 4 >= -AC_HALO) ? buf[slot(chain ? g : 0, x - l0 - 4 + AC_HALO) - PS_HH] : 0.0;
                    const double p0 = a * b0, p1 = a * c1, p2 = a * c2;
                    S0 = S0 + p0;
                    S1 = S1 + p1;
                    S2 = S2 + p2;
                }
                b2 = b1;
                b1 = b0;
            }
        }
        wave_lds_fence();
        // the last PS_HH entries of both parity arrays become the halo of the next tile
        {
            constexpr int NH = (PS_GMAX * 2 * PS_HH) / WAVE;
            double hv[NH];
#pragma unroll
            for (int r = 0; r < NH; r++) {
                const int idx = lane + r * WAVE;
                const int q = idx / (2 * PS_HH), rr = idx - q * 2 * PS_HH;
                hv[r] = buf[q * PS_STRIDE + (rr / PS_HH) * PS_ROW + PS_HALF + (rr % PS_HH)];
            }
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < NH; r++) {
                const int idx = lane + r * WAVE;
                const int q = idx / (2 * PS_HH), rr = idx - q * 2 * PS_HH;
                buf[q * PS_STRIDE + (rr / PS_HH) * PS_ROW + (rr % PS_HH)] = hv[r];
            }
        }
    }
    // lpc.c:68: autoc = temp + temp2 -- the two parities of a lag group are
    // neighbouring lanes
    const double o0 = __shfl_xor(S0, 1, WAVE), o1 = __shfl_xor(S1, 1, WAVE), o2 = __shfl_xor(S2, 1, WAVE);
    if (chain && pi == 0) {
        double *dst = autoc + (size_t)(s0 + g) * FHIP_MAX_LAGS;
        dst[l0] = S0 + o0;
        if (ok1) dst[l0 + 2] = S1 + o1;
        if (ok2) dst[l0 + 4] = S2 + o2;
    }
}

// ---------------------------------------------------------------------------
// K1 (main)  k_autocorr_wt<NCH> -- wave-typed producer / consumer
// ---------------------------------------------------------------------------
// PMC and in-kernel stamps on k_autocorr_ps showed the chain walk bound by the LDS
// (two 8-byte operand reads per lane and step, bank conflicts between the lag
// groups of a subframe, and one wave per SIMD that stalls on its own staging),
// not by the fp64 chains.  This kernel keeps the arithmetic and its order and
// changes who does what:
//  * a workgroup owns 32 subframes and has four consumer waves (one per SIMD) and
//    four producer waves (again one per SIMD);
//  * consumer wave w owns one lag group {l0, l0+2, .., l0+2(nch-1)} (same-parity
//    lags) for all 32 subframes: lane = (parity pi, subframe).  The lag shift is
//    wave-uniform, 32 lanes of one parity read 32 consecutive-stride addresses
//    (stride odd: conflict-free ds_read_b64), the group with l0 = 0 needs no
//    second operand stream at all (d[p - 0] is `a`), and the operands of the
//    higher lags of a group are the previous steps' values, carried in registers;
//  * producer wave p loads rows 8p..8p+7 three tiles ahead (counted waits),
//    windows them (lpc.c:28-40; two weights per lane and tile serve all rows)
//    and writes fp64 tiles, de-interleaved by parity, into a ring of three LDS
//    buffers: while the consumers walk buffer t%3 and its halo, the producers
//    fill buffer (t+1)%3 and the halo of buffer (t+2)%3 (the last 32 positions
//    of a tile are the next tile's halo).  One barrier per tile.
// Per step a consumer issues 2*NCH fp64 operations and one or two LDS reads.
// Measured (tools/ubench_walk.hip, s_memtime = core cycles at ~2.1 GHz): a lone
// wave issues an fp64 multiply or add every 3.9 cycles, and every double it
// takes from the LDS costs it another ~8.5 cycles of issue time (17 per
// ds_read_b128, whatever the prefetch depth: the return path, not the latency) --
// about as much as a multiply and an add.  The walk of a 3-chain group is
// therefore 6 x 3.9 + 8.5 = 32 cycles per step in isolation and 41 in the kernel
// (four consumers and the producers share the LDS), of which the kernel's other
// phases add 8 us (barriers, K2 tail).  Producers alone need 24 us per launch,
// consumers alone 41: they overlap to 56.  Two waves per SIMD doing fp64 pull
// the clock to ~1.3 GHz (power), so more fp64 waves per SIMD would not help.
// Requires n % AC_TILE == 0 (launch_autocorr falls back otherwise).
constexpr int WT_SUB = 32;                           // subframes per workgroup
#ifndef FHIP_WT_ROWS0
#define FHIP_WT_ROWS0 8
#endif
constexpr int WT_ROWS0 = FHIP_WT_ROWS0;              // rows staged by the producer next to consumer 0
constexpr int WT_ROWS1 = (WT_SUB - WT_ROWS0) / 3;    // ... by each of the other three
static_assert(WT_ROWS0 + 3 * WT_ROWS1 == WT_SUB, "producer row split");
// LDS geometry: the `a` stream is read 16 bytes (two steps) at a time, so arrays
// start on even doubles; ds_read_b128 serves 16 lanes per LDS cycle and is
// conflict-free when their 16-byte slots differ mod 16: stride/2 odd (83).  The
// shifted `b0` stream stays on single 8-byte reads (its alignment depends on the
// lag); with this stride they are 2-way conflicted, which the walk -- bound by
// instruction issue, not by the LDS -- does not feel.
constexpr int WT_ROW = PS_HH + PS_HALF + 2;          // doubles per parity array (82)
constexpr int WT_STRIDE = 2 * WT_ROW + 2;            // per subframe (166)
static_assert(WT_ROW % 2 == 0 && WT_STRIDE % 4 == 2, "16-byte aligned arrays, odd slot stride");
constexpr int WT_BUF = WT_SUB * WT_STRIDE;           // doubles per tile buffer
constexpr int WT_NBUF = 3;
#ifndef FHIP_WT_AHEAD
#define FHIP_WT_AHEAD 3
#endif
constexpr int WT_AHEAD = FHIP_WT_AHEAD;              // tiles of loads in flight per producer

struct wt_groups { int l0[4]; int nch[4]; };

// FUSED: the producers read the interleaved stereo PCM instead of smp, apply the
// channel mode and wasted-bits shift that the decision pass of K0 left in info[]
// (encode.c:668-693, :586-590), write smp for K3 and window the same values.
#ifdef FHIP_PROBE_NOB
constexpr bool wt_probe_nob = true;         // timing probes only: results are wrong
#else
constexpr bool wt_probe_nob = false;
#endif
#ifdef FHIP_PROBE_NOPROD
constexpr bool wt_probe_noprod = true;
#else
constexpr bool wt_probe_noprod = false;
#endif
#ifdef FHIP_PROBE_NOWALK
constexpr bool wt_probe_nowalk = true;
#else
constexpr bool wt_probe_nowalk = false;
#endif
#ifdef FHIP_PROBE_NOHALO
constexpr bool wt_probe_nohalo = true;      // timing probe only: results are wrong
#else
constexpr bool wt_probe_nohalo = false;
#endif
// LPCMO > 0: K2 as the kernel's tail.  The consumers leave their sums in LDS as
// well, and after one more barrier the first 32 lanes of wave 0 run Levinson /
// Schur and the quantiser for the workgroup's 32 subframes (max order <= LPCMO,
// everything in registers): what a separate launch does in 8 us -- it is latency
// bound, 128 waves on the whole chip -- costs about half of that here.
struct wt_lpc_args { int precision, omethod; int32_t *coefs, *shift, *opt_order, *fin; };
template <int MO>
__device__ __forceinline__ void lpc_reg_one(const double (&ac)[MO + 1], int s, int max_order, int precision,
                                            int omethod, int32_t *__restrict__ coefs,
                                            int32_t *__restrict__ shift, int32_t *__restrict__ opt_order,
                                            int32_t *__restrict__ fin);

template <int NCH, bool FUSED, int LPCMO>
__global__ __launch_bounds__(8 * WAVE)
void k_autocorr_wt(const int32_t *__restrict__ smp, double *__restrict__ autoc,
                   int nsub, int n, int maxlag, wt_groups grp, double c,
                   const int32_t *__restrict__ pcm, int32_t *__restrict__ smp_out,
                   const fhip_subframe_info *__restrict__ info, wt_lpc_args lpc, int narrow_ok)
{
    extern __shared__ __attribute__((aligned(16))) double wt_lds[];
    double *acbuf = wt_lds + WT_NBUF * WT_BUF;          // [32][FHIP_MAX_LAGS], LPCMO > 0 only

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int sub0 = blockIdx.x * WT_SUB;
    const int half = n >> 1;
    const int ntiles = n / AC_TILE;
    const int ntiles_pad = ((ntiles + WT_AHEAD - 1) / WT_AHEAD) * WT_AHEAD;   // the producers' unroll

    if (wv >= 4) {
        // ------------------------------ producer ------------------------------
        // lane = positions 2*lane, 2*lane+1 of the tile: one 8-byte load per row,
        // 512 contiguous bytes per row and instruction.  The loop is unrolled by
        // the prefetch depth so every tile has its own registers, and it has no
        // branches, so the waits stay counted (vmcnt(N), never 0).
        // A producer shares its SIMD with consumer wv-4.  The split of the 32 rows
        // between the producer next to consumer 0 (largest lag group) and the other
        // three is a build constant; measured 2/10/10/10: 60.3 us, 5/9/9/9 and
        // 8/8/8/8: 58.2 -- the walk of consumer 0, not the staging, sets the time.
        // ALLNAR: every row of this wave is a 16-bit row (the usual case for 16-bit input):
        // 4-byte loads and no per-row width select
        auto produce = [&](auto nrows_c, int q0, auto allnar_c) {
            constexpr int NR = decltype(nrows_c)::value;
            constexpr bool ALLNAR = decltype(allnar_c)::value;
            constexpr int NL = FUSED ? (NR + 1) / 2 : NR;      // loads per tile: one per frame when fused
            // (FUSED needs even row counts: rows come in channel pairs; the launcher checks)
            auto uni64 = [](unsigned long long v) {            // wave-uniform value -> SGPR pair
                return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                       (unsigned)__builtin_amdgcn_readfirstlane((int)v);
            };
            // row bases are wave-uniform: SGPR base + lane offset addressing
            unsigned long long rowb[NL], outb[NR];
            int mode[NL], w0s[NL], w1s[NL];
            int nar[NL];                                       // row stored as int16 (K0: info.reserved)
#pragma unroll
            for (int r = 0; r < NL; r++) {
                if (FUSED) {
                    const int sub = min(sub0 + q0 + 2 * r, nsub - 2);      // even: channel 0 of a frame
                    rowb[r] = uni64((unsigned long long)(pcm + (size_t)sub * n));     // frame sub/2: [n][2]
                    outb[2 * r] = uni64((unsigned long long)(smp_out + (size_t)sub * n));
                    outb[2 * r + 1] = uni64((unsigned long long)(smp_out + (size_t)(sub + 1) * n));
                    mode[r] = __builtin_amdgcn_readfirstlane(info[sub].ch_mode);
                    w0s[r] = __builtin_amdgcn_readfirstlane(info[sub].wasted);
                    w1s[r] = __builtin_amdgcn_readfirstlane(info[sub + 1].wasted);
                } else {
                    const int sub = min(sub0 + q0 + r, nsub - 1);
                    rowb[r] = uni64((unsigned long long)(smp + (size_t)sub * n));
                    nar[r] = narrow_ok ? __builtin_amdgcn_readfirstlane(info[sub].reserved) : 0;
                }
            }
            typedef typename std::conditional<FUSED, int4, int2>::type ld_t;
            ld_t pre[WT_AHEAD][NL];
            // pinned: keep the loads in program order.  The waits in the loop are counted, and a
            // prologue whose loads the scheduler shuffled makes the compiler merge both ways into
            // the loop to the smaller count (vmcnt(2) instead of 16: no prefetch left).
            auto issue_loads = [&](ld_t (&dst)[NL], int tb, bool pinned = false) {
                const int p = min(tb + 2 * lane, n - 2);       // past the block: clamped, weight 0
#pragma unroll
                for (int r = 0; r < NL; r++) {
                    // a narrow row holds the pair (2 lane, 2 lane + 1) in ONE dword, at int index p / 2;
                    // the same 8-byte load serves both widths (the second dword is then unused)
                    const int idx = FUSED ? 2 * p : ((ALLNAR || nar[r]) ? (p >> 1) : p);
                    // address space 1 spelled out: from an integer the pointer would be generic, the
                    // loads flat_load, and every wait on them vmcnt(0) lgkmcnt(0) -- no prefetch left
                    typedef const int32_t __attribute__((address_space(1))) *g_i32;
                    typedef int ldv_t __attribute__((ext_vector_type(FUSED ? 4 : 2)));
                    typedef const ldv_t __attribute__((address_space(1))) *g_ld;
                    if constexpr (ALLNAR) {
                        dst[r].x = *((g_i32)rowb[r] + idx);
                    } else {
                        const ldv_t v = *(g_ld)((g_i32)rowb[r] + idx);
                        __builtin_memcpy(&dst[r], &v, sizeof(ld_t));
                    }
                    if (pinned) __builtin_amdgcn_sched_barrier(0);
                }
            };
#pragma unroll
            for (int a = 0; a < WT_AHEAD; a++) issue_loads(pre[a], a * AC_TILE, true);
            // the halo of the first tile is zeros (positions -32 .. -1)
            for (int idx = lane; idx < NR * 2 * PS_HH; idx += WAVE) {
                const int r = idx / (2 * PS_HH), k = idx - r * 2 * PS_HH;
                wt_lds[(q0 + r) * WT_STRIDE + (k / PS_HH) * WT_ROW + (k % PS_HH)] = 0.0;
            }
            // lpc.c:34-39, 0 beyond the block
            auto weight = [&](int p) {
                const int ii = (p < half) ? p : (n - 1 - p);
                const bool valid = (p < n) && (ii < half);
                const double tt = c - (double)ii;
                return valid ? (1.0 - (tt * tt)) : 0.0;
            };
            int bi = 0;                                        // t % 3
            ACC_RESET(44, 48);
            for (int t0 = 0; t0 < ntiles_pad; t0 += WT_AHEAD) {
#pragma unroll
                for (int a = 0; a < WT_AHEAD; a++) {
                    TICK(tp0);
                    const int tb = (t0 + a) * AC_TILE;
                    const int bnx = (bi == WT_NBUF - 1) ? 0 : bi + 1;
                    double *bw = wt_lds + bi * WT_BUF + q0 * WT_STRIDE + PS_HH + lane;
                    double *bn = wt_lds + bnx * WT_BUF + q0 * WT_STRIDE + PS_HH + lane - PS_HALF;
                    const double w0 = weight(tb + 2 * lane), w1 = weight(tb + 2 * lane + 1);
                    const bool tail = lane >= PS_HALF - PS_HH;        // positions 96..127
                    const int pst = min(tb + 2 * lane, n - 2);        // padding tiles rewrite the last pair
                    // all rows in one basic block (the scheduler interleaves their dependent
                    // chains: a producer is alone with its latencies), the halo copies after it
                    double hv0[NR], hv1[NR];
#pragma unroll
                    for (int r = 0; r < (wt_probe_noprod ? 0 : NR); r++) {
                        int32_t x0, x1;                               // samples 2*lane, 2*lane+1 of row r
                        if (FUSED) {
                            const int4 v = *reinterpret_cast<const int4 *>(&pre[a][r / 2]);   // l0 r0 l1 r1
                            const int md = mode[r / 2];
                            // encode.c:668-693: channel 0 is mid / left / side(RS), channel 1 side / right
                            const int32_t s0 = (int32_t)((uint32_t)v.x - (uint32_t)v.y);
                            const int32_t s1 = (int32_t)((uint32_t)v.z - (uint32_t)v.w);
                            if ((r & 1) == 0) {
                                const int32_t m0 = (int32_t)((uint32_t)v.x + (uint32_t)v.y) >> 1;
                                const int32_t m1 = (int32_t)((uint32_t)v.z + (uint32_t)v.w) >> 1;
                                x0 = (md == FHIP_CH_MID_SIDE) ? m0 : (md == FHIP_CH_RIGHT_SIDE) ? s0 : v.x;
                                x1 = (md == FHIP_CH_MID_SIDE) ? m1 : (md == FHIP_CH_RIGHT_SIDE) ? s1 : v.z;
                                x0 >>= w0s[r / 2]; x1 >>= w0s[r / 2];
                            } else {
                                x0 = (md == FHIP_CH_MID_SIDE || md == FHIP_CH_LEFT_SIDE) ? s0 : v.y;
                                x1 = (md == FHIP_CH_MID_SIDE || md == FHIP_CH_LEFT_SIDE) ? s1 : v.w;
                                x0 >>= w1s[r / 2]; x1 >>= w1s[r / 2];
                            }
                            *reinterpret_cast<int2 *>(reinterpret_cast<int32_t *>(outb[r]) + pst) = make_int2(x0, x1);
                        } else {
                            const int2 v = *reinterpret_cast<const int2 *>(&pre[a][r]);
                            const bool nr = ALLNAR || nar[FUSED ? 0 : r] != 0;        // wave-uniform
                            x0 = nr ? (int32_t)(int16_t)v.x : v.x;
                            x1 = nr ? (v.x >> 16) : v.y;
                        }
                        const double v0 = (double)x0 * w0;
                        const double v1 = (double)x1 * w1;
#ifdef FHIP_PROBE_NOLDSW
                        if (r == 0) { hv0[0] = 0; hv1[0] = 0; }
                        hv0[0] += v0; hv1[0] += v1;
                        if (r == NR - 1) { bw[0] = hv0[0]; bw[WT_ROW] = hv1[0]; }
#else
                        bw[r * WT_STRIDE] = v0;
                        bw[r * WT_STRIDE + WT_ROW] = v1;
                        hv0[r] = v0; hv1[r] = v1;
#endif
                    }
                    if (tail && !wt_probe_nohalo && !wt_probe_noprod) {   // = positions -32..-1 of the next tile
#pragma unroll
                        for (int r = 0; r < NR; r++) {
                            bn[r * WT_STRIDE] = hv0[r];
                            bn[r * WT_STRIDE + WT_ROW] = hv1[r];
                        }
                    }
                    TICK(tp1);
                    issue_loads(pre[a], tb + WT_AHEAD * AC_TILE);
                    TICK(tp2);
                    __syncthreads();                               // tile handed over
                    TICK(tp3);
                    ACCUM(44, tp0, tp1); ACCUM(45, tp1, tp2); ACCUM(46, tp2, tp3);
                    // keep the next tile's conversions below this point: hoisted, they
                    // would wait for loads that still have two tiles of time
                    __builtin_amdgcn_sched_barrier(0);
                    bi = bnx;
                }
            }
        };
        const int q0w = (wv == 4) ? 0 : WT_ROWS0 + (wv - 5) * WT_ROWS1;
        const int nrw = (wv == 4) ? WT_ROWS0 : WT_ROWS1;
        bool alln = !FUSED && narrow_ok != 0;
        for (int r = 0; alln && r < nrw; r++) alln = info[min(sub0 + q0w + r, nsub - 1)].reserved != 0;
        alln = __builtin_amdgcn_readfirstlane((int)alln) != 0;
        if (wv == 4) {
            if (alln) produce(std::integral_constant<int, WT_ROWS0>{}, q0w, std::true_type{});
            else produce(std::integral_constant<int, WT_ROWS0>{}, q0w, std::false_type{});
        } else {
            if (alln) produce(std::integral_constant<int, WT_ROWS1>{}, q0w, std::true_type{});
            else produce(std::integral_constant<int, WT_ROWS1>{}, q0w, std::false_type{});
        }
        if (LPCMO > 0) __syncthreads();                    // the tail's barrier (below)
        return;
    }

    // -------------------------------- consumer --------------------------------
    const int pi = lane >> 5, sl = lane & 31;
    const int l0 = grp.l0[wv], nch = grp.nch[wv];       // wave-uniform
    const bool live = (sub0 + sl < nsub) && nch > 0;
    const int pib = pi ^ (l0 & 1);                      // parity array that holds d[p - l0]
    const int sft = (l0 + pib - pi) / 2;                // index shift inside that array
    const int offA = sl * WT_STRIDE + pi * WT_ROW + PS_HH;             // a  = buf[offA + step]
    const int offB = sl * WT_STRIDE + pib * WT_ROW + PS_HH - sft;      // b0 = buf[offB + step]
    const int pih = (maxlag + 1) & 1;                   // parity whose sum owns the head
    auto slotc = [&](int x) { return sl * WT_STRIDE + (x & 1) * WT_ROW + PS_HH + (x >> 1); };
    double S[NCH], cy[NCH];                             // running sums (lpc.c:58-59); cy[j] = d[p - l0 - 2j] carried
#pragma unroll
    for (int j = 0; j < NCH; j++) { S[j] = 1.0; cy[j] = 0.0; }

    // One tile: PS_HALF steps of NCH products, operands read two stages ahead.
    // FIRST is the tile that starts the block: products of positions <= maxlag
    // belong to the head (below), so their `a` is replaced by 0 -- a (+-0) product
    // leaves a running sum, which is never -0, bit for bit as it was.  SAME: the
    // group starts at lag 0, so b0 is a.
    // K = chains of this wave's group (NCH or NCH-1: the groups differ by at most one).
    auto walk_tile = [&](const double *rowA_, const double *rowB_, auto first, auto same, auto kc) {
        constexpr bool FIRST = decltype(first)::value;
        constexpr bool SAME = decltype(same)::value;
        constexpr int K = decltype(kc)::value;
        constexpr int NS = PS_HALF / PS_CH;
        // stages of operands in flight ahead of their use: two while a stage is short
        constexpr int DEPTH = (K <= 3) ? 2 : 1;
        constexpr int NSET = DEPTH + 1;
        // volatile: keeps the reads as written -- ds_read_b128 for `a`, single
        // ds_read_b64 for b0 (merged into ds_read2_b64 they run at half rate)
        typedef const volatile double __attribute__((address_space(3))) lds_cvd;
        typedef double dbl2 __attribute__((ext_vector_type(2)));
        typedef const volatile dbl2 __attribute__((address_space(3))) lds_cvd2;
        lds_cvd2 *rowA = (lds_cvd2 *)rowA_;
        lds_cvd *rowB = (lds_cvd *)rowB_;
        double A[NSET][PS_CH], B[NSET][PS_CH];
        auto fetch = [&](int set, int stage) {
#pragma unroll
            for (int u = 0; u < PS_CH; u += 2) {
                const dbl2 v = rowA[(stage * PS_CH + u) / 2];
                A[set][u] = v.x; A[set][u + 1] = v.y;
            }
            if (!SAME && !wt_probe_nob) {
#pragma unroll
                for (int u = 0; u < PS_CH; u++) B[set][u] = rowB[stage * PS_CH + u];
            }
        };
        constexpr int PER_STAGE = PS_CH / 2 + ((SAME || wt_probe_nob) ? 0 : PS_CH);     // LDS reads per stage
#pragma unroll
        for (int k = 0; k < DEPTH; k++) fetch(k, k);
#pragma unroll
        for (int st = 0; st < NS; st++) {
            if (st + DEPTH < NS) fetch((st + DEPTH) % NSET, st + DEPTH);
            // one wait per stage: everything but the reads just issued (and, two
            // stages deep, the stage before them) has arrived
            {
                constexpr int w1 = PER_STAGE > 15 ? 15 : PER_STAGE;              // one newer stage in flight
                constexpr int w2 = 2 * PER_STAGE > 15 ? 15 : 2 * PER_STAGE;      // two
                constexpr int enc1 = (3 << 14) | (w1 << 8) | (7 << 4) | 0xF;
                constexpr int enc2 = (3 << 14) | (w2 << 8) | (7 << 4) | 0xF;
                constexpr int enc0 = (3 << 14) | (0 << 8) | (7 << 4) | 0xF;
                const int newer = (st + DEPTH < NS ? 1 : 0) + ((DEPTH == 2 && st + 1 < NS) ? 1 : 0);
                if (newer == 2) __builtin_amdgcn_s_waitcnt(enc2);
                else if (newer == 1) __builtin_amdgcn_s_waitcnt(enc1);
                else __builtin_amdgcn_s_waitcnt(enc0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < PS_CH; u++) {
                const double x = (SAME || wt_probe_nob) ? A[st % NSET][u] : B[st % NSET][u];
                double a = A[st % NSET][u];
                if (FIRST && 2 * (st * PS_CH + u) <= FHIP_MAX_ORDER)            // steps that can hold p <= maxlag
                    a = (2 * (st * PS_CH + u) + pi > maxlag) ? a : 0.0;
                double pr[K];
                pr[0] = a * x;
#pragma unroll
                for (int j = 1; j < K; j++) pr[j] = a * cy[j];
#pragma unroll
                for (int j = 0; j < K; j++) S[j] = S[j] + pr[j];
#pragma unroll
                for (int j = K - 1; j >= 2; j--) cy[j] = cy[j - 1];
                if constexpr (K > 1) cy[1] = x;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int bi = 0;
    ACC_RESET(40, 44);
    for (int t = 0; t < ntiles_pad; t++) {
        TICK(tc0);
        __syncthreads();                                   // tile t is in buffer bi
        TICK(tc1);
        ACCUM(40, tc0, tc1);
        if (t >= ntiles || wt_probe_nowalk) continue;      // padding of the producers' unroll
        const double *buf = wt_lds + bi * WT_BUF;
        if (t == 0 && pi == pih) {
            // head (lpc.c:60-61): positions lag..maxlag of BOTH parities, in order,
            // into this lane's sums (tile 0 holds them all: maxlag <= 32 < AC_TILE)
            for (int x = 0; x <= maxlag; x++) {
                const double a = buf[slotc(x)];
#pragma unroll
                for (int j = 0; j < NCH; j++) {
                    const int lag = l0 + 2 * j;
                    if (j < nch && x >= lag) {
                        const double pr = a * buf[slotc(x - lag)];
                        S[j] = S[j] + pr;
                    }
                }
            }
        }
        using KF = std::integral_constant<int, NCH>;
        using KL = std::integral_constant<int, (NCH > 1) ? NCH - 1 : 1>;
        if (l0 == 0) {                                     // group 0 always has NCH chains
            if (t == 0) walk_tile(buf + offA, buf + offB, std::true_type{}, std::true_type{}, KF{});
            else walk_tile(buf + offA, buf + offB, std::false_type{}, std::true_type{}, KF{});
        } else if (nch == NCH) {
            if (t == 0) walk_tile(buf + offA, buf + offB, std::true_type{}, std::false_type{}, KF{});
            else walk_tile(buf + offA, buf + offB, std::false_type{}, std::false_type{}, KF{});
        } else if (nch > 0) {
            if (t == 0) walk_tile(buf + offA, buf + offB, std::true_type{}, std::false_type{}, KL{});
            else walk_tile(buf + offA, buf + offB, std::false_type{}, std::false_type{}, KL{});
        }
        bi = (bi == WT_NBUF - 1) ? 0 : bi + 1;
        TICK(tc2);
        ACCUM(t == 0 ? 42 : 41, tc1, tc2);
    }
    // lpc.c:68: autoc = temp + temp2 -- the two parities of a lag are lanes l, l+32
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        const double o = __shfl_xor(S[j], 32, WAVE);
        if (live && pi == 0 && j < nch) {
            const double v = S[j] + o;
            autoc[(size_t)(sub0 + sl) * FHIP_MAX_LAGS + l0 + 2 * j] = v;
            if (LPCMO > 0) acbuf[sl * FHIP_MAX_LAGS + l0 + 2 * j] = v;
        }
    }
    if constexpr (LPCMO > 0) {
        __syncthreads();                                   // all lags of the 32 subframes are in acbuf
        if (wv == 0 && lane < WT_SUB && sub0 + lane < nsub) {
            double ac[LPCMO + 1];
#pragma unroll
            for (int i = 0; i <= LPCMO; i++) ac[i] = (i <= maxlag) ? acbuf[lane * FHIP_MAX_LAGS + i] : 0.0;
            lpc_reg_one<LPCMO>(ac, sub0 + lane, maxlag, lpc.precision, lpc.omethod, lpc.coefs, lpc.shift,
                               lpc.opt_order, lpc.fin);
        }
    }
}

// ---------------------------------------------------------------------------
// K2  k_lpc
// ---------------------------------------------------------------------------
// One lane per subframe; per-lane work arrays live in LDS, laid out
// [index][lane] so that lanes never collide on a bank.
constexpr int LPC_NT = 64;

struct LaneArr {
    double *base;
    __device__ __forceinline__ double &operator[](int i) const { return base[i * LPC_NT]; }
};

// lpc.c:167-219 quantize_lpc_coefs applied to row = -a[0..order)
__device__ void quantize_row(const LaneArr a, int order, int precision,
                             int32_t *__restrict__ out, int32_t *__restrict__ shift_out)
{
    const int qmax = (1 << (precision - 1)) - 1;
    double cmax = 0.0;
    for (int j = 0; j < order; j++) {
        double m = fabs(a[j]);
        if (m > cmax) cmax = m;
    }
    if (cmax * 32768.0 < 1.0) {
        *shift_out = 0;
        for (int j = 0; j < order; j++) out[j] = 0;
        return;
    }
    int sh = 15;
    while (sh > 0 && cmax * (double)(1 << sh) > (double)qmax) sh--;
    const bool rescale = (sh == 0) && (cmax > (double)qmax);
    const double scale = rescale ? ((double)qmax / cmax) : 1.0;
    const double mul = (double)(1 << sh);
    double carry = 0.0;
    for (int j = 0; j < order; j++) {
        double v = -a[j];
        if (rescale) v = v * scale;
        double t = v * mul;
        carry = carry + t;
        int q = c_double_to_int(carry + 0.5);
        if (q <= -qmax) q = -qmax + 1;
        if (q > qmax) q = qmax;
        carry = carry - (double)q;
        out[j] = q;
    }
    *shift_out = sh;
}

// LDS doubles per lane: R0[33] R1[32] R2[32].
//   Levinson path: R0 = autoc, R1 = lpc_tmp.
//   Schur path:    R0 = autoc, whose tail doubles as gen[0] (gen[0][j] starts
//                  as autoc[j+1]); R1 = gen[1]; R2 = ref; afterwards R0 is
//                  reused as lpc_tmp.
constexpr int LPC_DBL = 33 + 32 + 32;

__global__ __launch_bounds__(LPC_NT)
void k_lpc(const double *__restrict__ autoc_all, int nsub, int max_order, int precision,
           int omethod, int32_t *__restrict__ coefs, int32_t *__restrict__ shift,
           int32_t *__restrict__ opt_order, int32_t *__restrict__ fin)
{
    __shared__ double s_mem[LPC_DBL * LPC_NT];
    const int lane = threadIdx.x;
    const int s = blockIdx.x * LPC_NT + lane;
    if (s >= nsub) return;

    LaneArr R0{s_mem + lane};
    LaneArr R1{s_mem + 33 * LPC_NT + lane};
    LaneArr R2{s_mem + 65 * LPC_NT + lane};

    for (int i = 0; i <= max_order; i++) R0[i] = autoc_all[(size_t)s * FHIP_MAX_LAGS + i];

    int32_t *crow = coefs + (size_t)s * FHIP_MAX_ORDER * FHIP_MAX_ORDER;
    int32_t *srow = shift + (size_t)s * FHIP_MAX_ORDER;

    int levinson_order = max_order;
    const bool use_ref = (omethod == 1 /* FLAKE_ORDER_METHOD_EST */);
    LaneArr ac = R0, a = R1, ref = R2;
    if (use_ref) {
        // lpc.c:125-162: Schur recursion
        LaneArr g0{R0.base + LPC_NT}, g1 = R1;
        for (int i = 0; i < max_order; i++) g1[i] = g0[i];
        double e = R0[0];
        {
            double r0 = -g1[0] / e;
            ref[0] = r0;
            double t = g1[0] * r0;
            e = e + t;
        }
        for (int i = 1; i < max_order; i++) {
            const double k = ref[i - 1];
            for (int j = 0; j < max_order - i; j++) {
                double up = g1[j + 1];
                double lo = g0[j];
                double t0 = k * lo;
                g1[j] = up + t0;
                double t1 = up * k;
                g0[j] = t1 + lo;
            }
            double ri = -g1[0] / e;
            ref[i] = ri;
            double t = g1[0] * ri;
            e = e + t;
        }
        int est = 1;
        for (int i = max_order - 1; i >= 0; i--) {
            if (fabs(ref[i]) > 0.10) { est = i + 1; break; }
        }
        levinson_order = est;
        a = R0;
    }

    // lpc.c:77-117 Levinson-Durbin; a[] is lpc_tmp
    double err = use_ref ? 1.0 : ac[0];
    for (int i = 0; i < FHIP_MAX_ORDER; i++) a[i] = 0.0;
    const bool all_rows = !(omethod == 0 || omethod == 1);
    for (int i = 0; i < levinson_order; i++) {
        double r;
        if (use_ref) {
            r = ref[i];
        } else {
            r = -ac[i + 1];
            for (int j = 0; j < i; j++) {
                double t = a[j] * ac[i - j];
                r = r - t;
            }
            r = r / err;
            double rr = r * r;
            double om = 1.0 - rr;
            err = err * om;
        }
        a[i] = r;
        const int h = i >> 1;
        for (int j = 0; j < h; j++) {
            double lo = a[j];
            double hi = a[i - 1 - j];
            double t0 = r * hi;
            a[j] = lo + t0;
            double t1 = r * lo;
            a[i - 1 - j] = hi + t1;
        }
        if (i & 1) {
            double m = a[h];
            double t = m * r;
            a[h] = m + t;
        }
        // lpc.c:243-254: one row for MAX/EST, every row for the search methods
        if (all_rows || i == levinson_order - 1)
            quantize_row(a, i + 1, precision, crow + i * FHIP_MAX_ORDER, srow + i);
    }
    opt_order[s] = levinson_order;
    if (!all_rows) {
        // compact copy of the single quantised row for K3's prefetch
        int32_t *f = fin + (size_t)s * FIN_STRIDE;
        const int32_t *src = crow + (levinson_order - 1) * FHIP_MAX_ORDER;
        for (int j = 0; j < max_order; j++) f[j] = (j < levinson_order) ? src[j] : 0;
        f[32] = srow[levinson_order - 1];
        f[33] = levinson_order;
        double *fd = reinterpret_cast<double *>(f + FIN_DBL);     // the first 8 as doubles (K3 reads them as scalars)
        for (int j = 0; j < 8; j++) fd[j] = (j < levinson_order && j < max_order) ? (double)src[j] : 0.0;
        int32_t cabs = 0, c8[8];
        for (int j = 0; j < 8; j++) c8[j] = (j < levinson_order && j < max_order) ? src[j] : 0;
        for (int j = 0; j < levinson_order; j++) cabs += (src[j] < 0) ? -src[j] : src[j];
        f[34] = cabs;
        for (int j = 0; j < 4; j++) f[FIN_PAIRS + j] = (c8[2 * j + 1] & 0xFFFF) | (int32_t)((uint32_t)c8[2 * j] << 16);
    }
}

// K2 for max_order <= MO (8 / 12): the same recursions with every array in
// registers.  The LDS version above pays an LDS round trip (~64+ cycles) for each
// of its ~300 dependent accesses per subframe; with one wave per SIMD nothing
// hides that.  All loops are unrolled to compile-time bounds with run-time
// guards, so no array is indexed dynamically.
template <int MO>
__device__ __forceinline__ void quantize_row_reg(const double (&a)[MO], int order, int precision,
                                                 int32_t *__restrict__ out, int32_t *__restrict__ shift_out,
                                                 int32_t *__restrict__ fin_out = nullptr, int max_order = 0)
{
    // fin_out: also the compact copy K3 prefetches (coefs, zeros up to max_order, shift, order)
    // lpc.c:167-219 on row = -a[0..order)
    const int qmax = (1 << (precision - 1)) - 1;
    double cmax = 0.0;
#pragma unroll
    for (int j = 0; j < MO; j++) {
        const double m = fabs(a[j]);
        if (j < order && m > cmax) cmax = m;
    }
    if (cmax * 32768.0 < 1.0) {
        *shift_out = 0;
#pragma unroll
        for (int j = 0; j < MO; j++) if (j < order) out[j] = 0;
        if (fin_out) {
#pragma unroll
            for (int j = 0; j < MO; j++) if (j < max_order) fin_out[j] = 0;
            fin_out[32] = 0;
            fin_out[33] = order;
            double *fd = reinterpret_cast<double *>(fin_out + FIN_DBL);
#pragma unroll
            for (int j = 0; j < 8; j++) fd[j] = 0.0;
            fin_out[34] = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) fin_out[FIN_PAIRS + j] = 0;
        }
        return;
    }
    int sh = 15;
    while (sh > 0 && cmax * (double)(1 << sh) > (double)qmax) sh--;
    const bool rescale = (sh == 0) && (cmax > (double)qmax);
    const double scale = rescale ? ((double)qmax / cmax) : 1.0;
    const double mul = (double)(1 << sh);
    double carry = 0.0;
    int32_t cabs = 0, c8[8];
#pragma unroll
    for (int j = 0; j < 8; j++) c8[j] = 0;
#pragma unroll
    for (int j = 0; j < MO; j++) {
        if (j < order) {
            double v = -a[j];
            if (rescale) v = v * scale;
            const double t = v * mul;
            carry = carry + t;
            int q = c_double_to_int(carry + 0.5);
            if (q <= -qmax) q = -qmax + 1;
            if (q > qmax) q = qmax;
            carry = carry - (double)q;
            out[j] = q;
            cabs += (q < 0) ? -q : q;
            if (j < 8) c8[j] = q;
            if (fin_out) {
                fin_out[j] = q;
                if (j < 8) reinterpret_cast<double *>(fin_out + FIN_DBL)[j] = (double)q;
            }
        } else if (fin_out) {
            if (j < max_order) fin_out[j] = 0;
            if (j < 8) reinterpret_cast<double *>(fin_out + FIN_DBL)[j] = 0.0;
        }
    }
    *shift_out = sh;
    if (fin_out) {
        fin_out[32] = sh;
        fin_out[33] = order;
        // for K3's 16-bit dot-product FIR: sum |coef| and the first 8 coefficients as
        // int16 pairs (lo: tap 2j+2, hi: tap 2j+1)
        fin_out[34] = cabs;
#pragma unroll
        for (int j = 0; j < 4; j++)
            fin_out[FIN_PAIRS + j] = (c8[2 * j + 1] & 0xFFFF) | (int32_t)((uint32_t)c8[2 * j] << 16);
    }
}

// K2 for one subframe with every array in registers (max_order <= MO): Levinson /
// Schur, quantiser, outputs.  Called by k_lpc_reg (one lane per subframe) and by
// the tail of k_autocorr_wt.
template <int MO>
__device__ __forceinline__ void lpc_reg_one(const double (&ac)[MO + 1], int s, int max_order, int precision,
                                            int omethod, int32_t *__restrict__ coefs,
                                            int32_t *__restrict__ shift, int32_t *__restrict__ opt_order,
                                            int32_t *__restrict__ fin)
{
    int32_t *crow = coefs + (size_t)s * FHIP_MAX_ORDER * FHIP_MAX_ORDER;
    int32_t *srow = shift + (size_t)s * FHIP_MAX_ORDER;
    int levinson_order = max_order;
    const bool use_ref = (omethod == 1);
    double ref[MO];
#pragma unroll
    for (int i = 0; i < MO; i++) ref[i] = 0.0;
    if (use_ref) {
        // lpc.c:125-162 Schur recursion
        double g0[MO], g1[MO];
#pragma unroll
        for (int i = 0; i < MO; i++) { g0[i] = ac[i + 1]; g1[i] = ac[i + 1]; }
        double e = ac[0];
        {
            const double r0 = -g1[0] / e;
            ref[0] = r0;
            const double t = g1[0] * r0;
            e = e + t;
        }
#pragma unroll
        for (int i = 1; i < MO; i++) {
            if (i < max_order) {
                const double k = ref[i - 1];
#pragma unroll
                for (int j = 0; j < MO - 1; j++) {
                    if (j < max_order - i) {
                        const double up = g1[j + 1];
                        const double lo = g0[j];
                        const double t0 = k * lo;
                        g1[j] = up + t0;
                        const double t1 = up * k;
                        g0[j] = t1 + lo;
                    }
                }
                const double ri = -g1[0] / e;
                ref[i] = ri;
                const double t = g1[0] * ri;
                e = e + t;
            }
        }
        int est = 1;
        bool found = false;
#pragma unroll
        for (int i = MO - 1; i >= 0; i--) {
            if (!found && i < max_order && fabs(ref[i]) > 0.10) { est = i + 1; found = true; }
        }
        levinson_order = est;
    }

    // lpc.c:77-117 Levinson-Durbin
    double a[MO];
#pragma unroll
    for (int i = 0; i < MO; i++) a[i] = 0.0;
    double err = use_ref ? 1.0 : ac[0];
    const bool all_rows = !(omethod == 0 || omethod == 1);
#pragma unroll
    for (int i = 0; i < MO; i++) {
        if (i < levinson_order) {
            double r;
            if (use_ref) {
                r = ref[i];
            } else {
                r = -ac[i + 1];
#pragma unroll
                for (int j = 0; j < i; j++) {
                    const double t = a[j] * ac[i - j];
                    r = r - t;
                }
                r = r / err;
                const double rr = r * r;
                const double om = 1.0 - rr;
                err = err * om;
            }
            a[i] = r;
            constexpr int dummy = 0; (void)dummy;
            const int h = i >> 1;
#pragma unroll
            for (int j = 0; j < (i >> 1); j++) {
                const double lo = a[j];
                const double hi = a[i - 1 - j];
                const double t0 = r * hi;
                a[j] = lo + t0;
                const double t1 = r * lo;
                a[i - 1 - j] = hi + t1;
            }
            if (i & 1) {
                const double m = a[h];
                const double t = m * r;
                a[h] = m + t;
            }
            if (all_rows)
                quantize_row_reg<MO>(a, i + 1, precision, crow + i * FHIP_MAX_ORDER, srow + i);
            else if (i == levinson_order - 1)       // the one row of MAX / EST, and its compact copy
                quantize_row_reg<MO>(a, i + 1, precision, crow + i * FHIP_MAX_ORDER, srow + i,
                                     fin + (size_t)s * FIN_STRIDE, max_order);
        }
    }
    opt_order[s] = levinson_order;
}

template <int MO>
__global__ __launch_bounds__(LPC_NT)
void k_lpc_reg(const double *__restrict__ autoc_all, int nsub, int max_order, int precision,
               int omethod, int32_t *__restrict__ coefs, int32_t *__restrict__ shift,
               int32_t *__restrict__ opt_order, int32_t *__restrict__ fin)
{
    const int s = blockIdx.x * LPC_NT + threadIdx.x;
    if (s >= nsub) return;
    double ac[MO + 1];
#pragma unroll
    for (int i = 0; i <= MO; i++) ac[i] = (i <= max_order) ? autoc_all[(size_t)s * FHIP_MAX_LAGS + i] : 0.0;
    lpc_reg_one<MO>(ac, s, max_order, precision, omethod, coefs, shift, opt_order, fin);
}

// ---------------------------------------------------------------------------
// K3  k_encode
// ---------------------------------------------------------------------------
// One workgroup per subframe.  Thread t owns the contiguous run of `chunk`
// samples starting at t*chunk: its residuals stay in registers from the FIR
// through the partition sums to the bit emit.  Samples sit in LDS behind one
// pad word per 16 so that lane-strided reads of x[16*t + d] spread over all
// banks.
struct EncLds {
    int32_t *smp;                       // padded samples
    unsigned long long *sums;           // [511] partition sums, heap order
    int32_t *kpar;                      // [511] Rice parameter per node
    uint32_t *lvl_bits;                 // [9]
    uint32_t *lvl_meth;                 // [9]
    int32_t *coef;                      // [32]
    int32_t *misc;                      // [16]
    uint32_t *trial;                    // [32] bits[] table of the log search
    unsigned long long *scan;           // [8]
    uint32_t *bits;                     // [ENC_WWORDS] emit window
};
constexpr int ENC_WWORDS = 2048;        // 64 Kbit emit window
enum { M_PORDER = 0, M_METHOD = 1, M_BITS = 2, M_FLAG = 3 };

__device__ __forceinline__ int padidx(int i) { return i + (i >> 4); }

__host__ __device__ inline size_t enc_lds_layout(int n, size_t off[10])
{
    size_t o = 0;
    off[0] = o; o += 8 * 511;                                   // sums
    off[1] = o; o += 8 * 8;                                     // scan
    off[2] = o; o += 4 * (size_t)(n + (n >> 4) + 1);            // smp
    off[3] = o; o += 4 * 511;                                   // kpar
    off[4] = o; o += 4 * 9;                                     // lvl_bits
    off[5] = o; o += 4 * 9;                                     // lvl_meth
    off[6] = o; o += 4 * 32;                                    // coef
    off[7] = o; o += 4 * 16;                                    // misc
    off[9] = o; o += 4 * 32;                                    // trial
    o = (o + 15) & ~(size_t)15;
    off[8] = o; o += 4 * ENC_WWORDS;                            // bits
    return o;
}

struct EncCtx {
    EncLds l;
    int n, chunk, i0, tid;
    int obits, precision;
    int pmin_req, pmax_req;
};

__device__ __forceinline__ int ilog2_dev(uint32_t v) { return v ? 31 - __clz((int)v) : 0; }

// rice.c:148-155 limit_max_partition_order
__device__ __forceinline__ int clamp_porder(int porder, int n, int order)
{
    int lim = ilog2_dev((uint32_t)(n ^ (n - 1)));
    porder = min(porder, lim);
    if (order > 0) porder = min(porder, ilog2_dev((uint32_t)(n / order)));
    return porder;
}

// Rice search over the residuals held in r[] (rice.c:105-187).  Leaves the
// per-node parameters in l.kpar, the chosen order/method in l.misc and
// returns the subframe bit estimate.  All threads must call it.
template <int C>
__device__ __forceinline__ uint32_t rice_search(const EncCtx &e, const int32_t (&r)[C], int order, bool lpc)
{
    const EncLds &l = e.l;
    const int n = e.n, tid = e.tid;
    const int pmin = clamp_porder(e.pmin_req, n, order);
    const int pmax = clamp_porder(e.pmax_req, n, order);
    const int psize = n >> pmax;

    for (int q = tid; q < 511; q += NT) l.sums[q] = 0;
    if (tid < 9) { l.lvl_bits[tid] = 0; l.lvl_meth[tid] = 0; }
    __syncthreads();

    // rice.c:76-94 finest-level sums: partition 0 starts at `order`
    {
        const int heap0 = (1 << pmax) - 1;
        unsigned long long run = 0;
        int part = -1, bound = 0;
#pragma unroll
        for (int o = 0; o < C; o++) {
            const int i = e.i0 + o;
            if (o < e.chunk && i < n && i >= order) {
                if (part < 0) { part = i / psize; bound = (part + 1) * psize; }
                if (i == bound) {
                    atomicAdd(&l.sums[heap0 + part], run);
                    run = 0; part++; bound += psize;
                }
                run += zigzag32(r[o]);
            }
        }
        if (part >= 0) atomicAdd(&l.sums[heap0 + part], run);
    }
    __syncthreads();
    // rice.c:96-102 pyramid
    for (int p = pmax - 1; p >= pmin; p--) {
        const int np = 1 << p;
        for (int j = tid; j < np; j += NT)
            l.sums[np - 1 + j] = l.sums[2 * np - 1 + 2 * j] + l.sums[2 * np - 1 + 2 * j + 1];
        __syncthreads();
    }
    // rice.c:47-74 per level, per partition: best k and its cost
    {
        const int first = (1 << pmin) - 1, last = (2 << pmax) - 2;
        for (int q = first + tid; q <= last; q += NT) {
            const int p = ilog2_dev((uint32_t)(q + 1));
            const int j = q + 1 - (1 << p);
            const int cnt = (n >> p) - (j == 0 ? order : 0);
            uint32_t b;
            const int k = rice_best_k(l.sums[q], cnt, &b);
            l.kpar[q] = k;
            atomicAdd(&l.lvl_bits[p], b);
            if (k > 14) atomicOr(&l.lvl_meth[p], 1u);
        }
    }
    __syncthreads();
    if (tid == 0) {
        // rice.c:127-138: ties go to the higher partition order
        int bp = pmin;
        uint32_t best = l.lvl_bits[pmin] + 4u * (1u << pmin);
        for (int p = pmin + 1; p <= pmax; p++) {
            uint32_t b = l.lvl_bits[p] + 4u * (1u << p);
            if (b <= best) { best = b; bp = p; }
        }
        const uint32_t method = l.lvl_meth[bp];
        // rice.c:157-171
        uint32_t bits = (uint32_t)(order * e.obits + 2);
        if (lpc) bits += (uint32_t)(4 + 5 + order * e.precision);
        bits += best;
        bits += method + 4u;
        l.misc[M_PORDER] = bp;
        l.misc[M_METHOD] = (int)method;
        l.misc[M_BITS] = (int)bits;
    }
    __syncthreads();
    return (uint32_t)l.misc[M_BITS];
}

// optimize.c:70-122 encode_residual_lpc for this thread's run
template <int C>
__device__ __forceinline__ void residual_lpc(const EncCtx &e, int32_t (&r)[C], int order,
                                             const int32_t *__restrict__ coefs_row, int shift)
{
    const EncLds &l = e.l;
    __syncthreads();                       // previous readers of l.coef are done
    if (e.tid < order) l.coef[e.tid] = coefs_row[e.tid];
    __syncthreads();
#pragma unroll
    for (int o = 0; o < C; o++) {
        const int i = e.i0 + o;
        int32_t v = 0;
        if (o < e.chunk && i < e.n) {
            const int32_t x = l.smp[padidx(i)];
            if (i < order) {
                v = x;
            } else {
                long long pred = 0;
                for (int j = order; j >= 1; j--)
                    pred += (long long)l.coef[j - 1] * (long long)l.smp[padidx(i - j)];
                v = (int32_t)((long long)x - (pred >> shift));
            }
        }
        r[o] = v;
    }
}

// optimize.c:34-68 encode_residual_fixed
template <int C>
__device__ __forceinline__ void residual_fixed(const EncCtx &e, int32_t (&r)[C], int order)
{
    const EncLds &l = e.l;
#pragma unroll
    for (int o = 0; o < C; o++) {
        const int i = e.i0 + o;
        int32_t v = 0;
        if (o < e.chunk && i < e.n) {
            const long long x0 = l.smp[padidx(i)];
            if (i < order || order == 0) {
                v = (int32_t)x0;
            } else {
                const long long x1 = l.smp[padidx(i - 1)];
                long long acc;
                if (order == 1) {
                    acc = x0 - x1;
                } else {
                    const long long x2 = l.smp[padidx(i - 2)];
                    if (order == 2) {
                        acc = x0 - 2 * x1 + x2;
                    } else {
                        const long long x3 = l.smp[padidx(i - 3)];
                        if (order == 3) {
                            acc = x0 - 3 * x1 + 3 * x2 - x3;
                        } else {
                            const long long x4 = l.smp[padidx(i - 4)];
                            acc = x0 - 4 * x1 + 6 * x2 - 4 * x3 + x4;
                        }
                    }
                }
                v = (int32_t)acc;
            }
        }
        r[o] = v;
    }
}

// OR a value of `len` (<= 31) bits into the MSB-first bit string at absolute
// bit position pos, clipped to the LDS window [wlo, wlo + ENC_WWORDS) words.
__device__ __forceinline__ void put_bits(uint32_t *win, long long wlo, long long pos, int len, uint32_t val)
{
    const long long wi = (pos >> 5) - wlo;
    if (wi < -1 || wi >= ENC_WWORDS) return;
    const int sh = 64 - len - (int)(pos & 31);
    const unsigned long long x = (unsigned long long)val << sh;
    const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
    if (wi >= 0 && hi) atomicOr(&win[wi], hi);
    if (lo && wi + 1 < ENC_WWORDS) atomicOr(&win[wi + 1], lo);
}

template <int C>
__global__ __launch_bounds__(NT)
void k_encode(fhip_params P, int n, const int32_t *__restrict__ smp_all,
              const int32_t *__restrict__ coefs_all, const int32_t *__restrict__ shift_all,
              const int32_t *__restrict__ opt_all, fhip_subframe_info *__restrict__ info,
              int32_t *__restrict__ res_out, uint8_t *__restrict__ bits_out, long long slot_bytes,
              int raw_order, int raw_lpc)
{
    // raw_order >= 0: the input already IS a residual; only calc_rice_params_*
    // (rice.c:173-187) with that prediction order and the emit run.
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    size_t off[10];
    enc_lds_layout(n, off);
    EncCtx e;
    e.l.sums = reinterpret_cast<unsigned long long *>(lds_raw + off[0]);
    e.l.scan = reinterpret_cast<unsigned long long *>(lds_raw + off[1]);
    e.l.smp = reinterpret_cast<int32_t *>(lds_raw + off[2]);
    e.l.kpar = reinterpret_cast<int32_t *>(lds_raw + off[3]);
    e.l.lvl_bits = reinterpret_cast<uint32_t *>(lds_raw + off[4]);
    e.l.lvl_meth = reinterpret_cast<uint32_t *>(lds_raw + off[5]);
    e.l.coef = reinterpret_cast<int32_t *>(lds_raw + off[6]);
    e.l.misc = reinterpret_cast<int32_t *>(lds_raw + off[7]);
    e.l.bits = reinterpret_cast<uint32_t *>(lds_raw + off[8]);
    e.l.trial = reinterpret_cast<uint32_t *>(lds_raw + off[9]);
    const EncLds &l = e.l;

    const int s = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    fhip_subframe_info *out = &info[s];
    e.n = n;
    e.tid = tid;
    e.chunk = (n + NT - 1) / NT;
    e.i0 = tid * e.chunk;
    e.obits = out->obits;
    e.precision = P.lpc_precision;
    e.pmin_req = P.min_partition_order;
    e.pmax_req = P.max_partition_order;

    const int32_t *src = smp_all + (size_t)s * n;
    if (tid == 0) l.misc[M_FLAG] = 0;
    __syncthreads();
    {
        const int32_t first = src[0];
        int differs = 0;
        for (int i = tid; i < n; i += NT) {
            int32_t v = src[i];
            l.smp[padidx(i)] = v;
            differs |= (v != first);
        }
        if (differs) atomicOr(&l.misc[M_FLAG], 1);
    }
    __syncthreads();
    const bool constant = (l.misc[M_FLAG] == 0);

    int32_t r[C];
    int type, type_code, order = 0, shift = 0;
    uint32_t est_bits = 0;
    bool has_rice = false;
    const int32_t *crow_base = coefs_all + (size_t)s * FHIP_MAX_ORDER * FHIP_MAX_ORDER;
    const int32_t *srow = shift_all + (size_t)s * FHIP_MAX_ORDER;

    // The decision tree of encode_residual() (optimize.c:124-276) as one
    // candidate loop: `pick` walks the orders the reference would try, in its
    // sequence; the last pass re-encodes the winner exactly as the reference
    // does (optimize.c:184-188, :266-275).  Every variable below is uniform
    // across the workgroup.
    enum { T_CONST, T_VERB, T_FIXED, T_LPC, T_RAW } tree;
    if (raw_order >= 0) tree = T_RAW;
    else if (constant) tree = T_CONST;                                   // optimize.c:143-151
    else if (n < 5 || P.prediction_type == 0) tree = T_VERB;             // optimize.c:153-158
    else if (P.prediction_type == 1 || n <= P.max_prediction_order) tree = T_FIXED;
    else tree = T_LPC;

    const int omethod = P.order_method;
    const int min_order = P.min_prediction_order;
    const int max_order = (tree == T_FIXED) ? min(P.max_prediction_order, 4) : P.max_prediction_order;

    // iteration state
    int it = 0;                 // FIXED: order; LEVEL: index; SEARCH: row
    int best = 0;               // FIXED: order; LPC: zero-based row
    uint32_t best_bits = 0, last_bits = 0;
    bool have_best = false;
    int lg_step = 16, lg_last = 0, lg_pos = 3;      // log search
    bool final_pass = false;

    if (tree == T_FIXED) { it = min_order; best = min_order; }
    if (tree == T_LPC) {
        if (omethod == 0) { best = max_order - 1; final_pass = true; }
        else if (omethod == 1) { best = opt_all[s] - 1; final_pass = true; }
        else if (omethod <= 4) { it = (1 << (omethod - 1)) - 1; best = max_order - 1; }
        else if (omethod == 5) { it = 0; best = 0; }
        else {
            best = min_order - 1 + (max_order - min_order) / 3;
            if (tid < FHIP_MAX_ORDER) l.trial[tid] = 0xFFFFFFFFu;
            __syncthreads();
            lg_step = 32;       // first pick halves it to 16
        }
    }

    if (tree == T_RAW) {
        residual_fixed<C>(e, r, 0);
        est_bits = rice_search<C>(e, r, raw_order, raw_lpc != 0);
        order = raw_order;
        type = raw_lpc ? FHIP_SUB_LPC : FHIP_SUB_FIXED;
        type_code = type;
        has_rice = true;
    } else if (tree == T_CONST || tree == T_VERB) {
        type = type_code = (tree == T_CONST) ? FHIP_SUB_CONSTANT : FHIP_SUB_VERBATIM;
        est_bits = (uint32_t)(tree == T_CONST ? e.obits : e.obits * n);
        residual_fixed<C>(e, r, 0);
    } else {
        for (;;) {
            // ---- pick the next candidate (cand: FIXED order / LPC row) ----
            int cand = -1;
            if (!final_pass) {
                if (tree == T_FIXED) {
                    if (it <= max_order) cand = it;
                } else if (omethod <= 4) {
                    // optimize.c:202-223: level indices high -> low
                    if (it >= 0) {
                        const int levels = 1 << (omethod - 1);
                        cand = min_order + (((max_order - min_order + 1) * (it + 1)) / levels) - 2;
                        if (cand < 0) cand = 0;
                    }
                } else if (omethod == 5) {
                    // optimize.c:224-238: rows 0..max-1, min_order ignored
                    if (it < max_order) cand = it;
                } else {
                    // optimize.c:239-261 log search, bits[] in l.trial
                    for (;;) {
                        if (lg_pos == 3) {
                            lg_step >>= 1;
                            if (lg_step == 0) break;
                            lg_last = best;
                            lg_pos = 0;
                        }
                        const int i = lg_last + (lg_pos - 1) * lg_step;
                        lg_pos++;
                        if (i < min_order - 1 || i >= max_order || l.trial[i] < 0xFFFFFFFFu) continue;
                        cand = i;
                        break;
                    }
                }
                if (cand < 0) {
                    // candidates exhausted: FIXED keeps the last residual when
                    // the winner was tried last (optimize.c:184-189)
                    if (tree == T_FIXED && best == max_order) { est_bits = last_bits; break; }
                    final_pass = true;
                }
            }
            if (final_pass) cand = best;

            // ---- evaluate it ----
            uint32_t b;
            if (tree == T_FIXED) {
                residual_fixed<C>(e, r, cand);
                b = rice_search<C>(e, r, cand, false);
            } else {
                residual_lpc<C>(e, r, cand + 1, crow_base + cand * FHIP_MAX_ORDER, srow[cand]);
                b = rice_search<C>(e, r, cand + 1, true);
            }
            if (final_pass) { est_bits = b; break; }

            // ---- fold it into the running decision ----
            last_bits = b;
            if (tree == T_FIXED) {
                if (!have_best || b < best_bits) { best_bits = b; best = cand; }   // strict '<', optimize.c:177
                it++;
            } else if (omethod <= 4) {
                if (!have_best) best_bits = b;              // index levels-1: opt_order stays max_order-1
                else if (b < best_bits) { best_bits = b; best = cand; }
                it--;
            } else if (omethod == 5) {
                if (!have_best || b < best_bits) { best_bits = b; best = cand; }
                it++;
            } else {
                if (tid == 0) l.trial[cand] = b;
                __syncthreads();
                if (b < l.trial[best]) best = cand;         // optimize.c:256
            }
            have_best = true;
        }
        if (tree == T_FIXED) {
            order = best;
            type = FHIP_SUB_FIXED;
            type_code = FHIP_SUB_FIXED | order;
        } else {
            order = best + 1;
            shift = srow[best];
            type = FHIP_SUB_LPC;
            type_code = FHIP_SUB_LPC | (order - 1);
        }
        has_rice = true;
    }

    const int porder = has_rice ? l.misc[M_PORDER] : 0;
    const int method = has_rice ? l.misc[M_METHOD] : 0;

    // FlacSubframe.residual
    if (res_out) {
        int32_t *dst = res_out + (size_t)s * n;
#pragma unroll
        for (int o = 0; o < C; o++) {
            const int i = e.i0 + o;
            if (o < e.chunk && i < n) dst[i] = r[o];
        }
    }

    // encode.c:766-798 output_residual, all partitions and codewords at once
    long long total_bits = 0;
    if (has_rice) {
        const int psz = n >> porder;
        const int pbits = 4 + method;
        const int heap0 = (1 << porder) - 1;
        unsigned long long mine = 0;
#pragma unroll
        for (int o = 0; o < C; o++) {
            const int i = e.i0 + o;
            if (o < e.chunk && i < n && i >= order) {
                const int part = i / psz;
                const int k = l.kpar[heap0 + part];
                if (part > 0 && i == part * psz) mine += pbits;
                mine += (unsigned long long)(emit_fold32(r[o]) >> k) + 1 + k;
            }
        }
        unsigned long long incl = wave_incl_scan_u64(mine, lane);
        if (lane == 63) l.scan[wv] = incl;
        __syncthreads();
        unsigned long long base = 6 + pbits;           // section header + partition 0 parameter
        for (int w = 0; w < wv; w++) base += l.scan[w];
        const unsigned long long tot = 6 + pbits + l.scan[0] + l.scan[1] + l.scan[2] + l.scan[3];
        const unsigned long long my_off = base + incl - mine;
        total_bits = (tot > 0x7FFFFFFFull) ? 0x7FFFFFFFll : (long long)tot;

        if (bits_out) {
            if (tot > (unsigned long long)slot_bytes * 8ull) {
                total_bits = -1;
            } else {
                uint32_t *dst32 = reinterpret_cast<uint32_t *>(bits_out + (size_t)s * slot_bytes);
                const long long nwords = (long long)((tot + 31) >> 5);
                for (long long wlo = 0; wlo < nwords; wlo += ENC_WWORDS) {
                    __syncthreads();
                    for (int q = tid; q < ENC_WWORDS; q += NT) l.bits[q] = 0;
                    __syncthreads();
                    if (tid == 0) {
                        put_bits(l.bits, wlo, 0, 2, (uint32_t)method);
                        put_bits(l.bits, wlo, 2, 4, (uint32_t)porder);
                        put_bits(l.bits, wlo, 6, pbits, (uint32_t)l.kpar[heap0]);
                    }
                    long long pos = (long long)my_off;
#pragma unroll
                    for (int o = 0; o < C; o++) {
                        const int i = e.i0 + o;
                        if (o < e.chunk && i < n && i >= order) {
                            const int part = i / psz;
                            const int k = l.kpar[heap0 + part];
                            if (part > 0 && i == part * psz) {
                                put_bits(l.bits, wlo, pos, pbits, (uint32_t)k);
                                pos += pbits;
                            }
                            // bitio.h:120-141: q zeros, a one, k low bits
                            const uint32_t u = emit_fold32(r[o]);
                            const uint32_t q = u >> k;
                            put_bits(l.bits, wlo, pos + q, k + 1, (1u << k) | (u & ((1u << k) - 1u)));
                            pos += (long long)q + 1 + k;
                        }
                    }
                    __syncthreads();
                    const long long cnt = (nwords - wlo < ENC_WWORDS) ? (nwords - wlo) : (long long)ENC_WWORDS;
                    for (int q = tid; q < cnt; q += NT)
                        dst32[wlo + q] = __builtin_bswap32(l.bits[q]);
                }
            }
        }
    }

    // FlacSubframe / RiceContext fields
    if (tid == 0) {
        out->type = type;
        out->type_code = type_code;
        out->order = order;
        out->shift = shift;
        out->rice_method = method;
        out->porder = porder;
        out->est_bits = est_bits;
        out->rice_nbits = (int32_t)total_bits;
        out->reserved = 0;
    }
    if (tid < FHIP_MAX_ORDER)
        out->coefs[tid] = (type == FHIP_SUB_LPC && tid < order && raw_order < 0) ? l.coef[tid] : 0;
    {
        const int np = has_rice ? (1 << porder) : 0;
        out->rparams[tid] = (tid < np) ? l.kpar[np - 1 + tid] : 0;
    }
    // warm-up samples (= residual[0..order)); [0] carries a CONSTANT's value
    if (tid < FHIP_MAX_ORDER) {
        const int nw = (type == FHIP_SUB_CONSTANT) ? 1 : order;
        out->warmup[tid] = (tid < nw && tid < n) ? l.smp[padidx(tid)] : 0;
    }
}


// ---------------------------------------------------------------------------
// K3 fast path  k_encode_pow2<C, T>
// ---------------------------------------------------------------------------
// Same contract as k_encode, for block sizes n = C*T with T (threads) a power
// of two >= 64, C in {3, 4, 8, 9, 16, 18} samples per thread and every partition at
// least one thread wide ((n >> pmax) >= C): all of FLAC's standard block sizes
// (192, 576, 1152, 2304, 4608 = 3 or 9 times a power of two; 256 .. 16384).  Then
//   * no lane ever needs a bounds or partition-boundary test per sample: a
//     thread's run lies inside one partition of every level;
//   * the FIR runs as exact fp64 FMAs (|coef| < 2^14, |sample| < 2^31, <= 32
//     taps: every partial sum is an integer below 2^50 < 2^53), in register
//     blocks of 8 taps -- v_fma_f64 issues 3-4x faster than v_mad_i64_i32;
//   * partition sums are a wave shuffle pyramid (thread = finest level);
//   * the best Rice parameter comes from a closed form, with the reference's
//     31-step scan only where its modular arithmetic can bite (see rice_k_fast).
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
__host__ __device__ inline int fast_window_words(int n)
{
    int w = 256;
    while (w < ENC_WWORDS && 2 * w < n) w <<= 1;
    return w;
}

__host__ __device__ inline size_t fast_lds_layout(int n, size_t img_doubles, size_t off[11])
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
    off[10] = o; o += 4 * fast_window_words(n);                 // bits
    return o;
}

constexpr int clog2(int v) { return v <= 1 ? 0 : 1 + clog2(v >> 1); }

template <int C, int T>
struct FastCtx {
    FastLds l;
    int n, i0, tid, lane, wv;
    int obits, precision, pmin_req, pmax_req;
};

// FIR residual of this thread's C samples x[] for an LPC candidate
// (optimize.c:70-122).  l.coefd holds the coefficients as doubles, zero past
// `order`, so the tap loop runs in whole blocks of 8.
template <int C, int T>
__device__ __forceinline__ void fir_lpc(const FastCtx<C, T> &e, int32_t (&r)[C], int order, int shift)
{
    using Img = SmpImg<C, T>;
    const FastLds &l = e.l;
    const double inv = __builtin_ldexp(1.0, -shift);
    // outputs per register block: a divisor of C
    constexpr int OB = (C % 8 == 0) ? 8 : (C % 4 == 0) ? 4 : (C % 3 == 0) ? 3 : 1;
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
                    if constexpr (Img::V4) {
                        // the window starts on a group of four: 16-byte loads
                        static_assert(!Img::V4 || (TB == 16 && OB % 4 == 0), "aligned windows");
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
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) {
                        if (jj < NT_) {
                            const double cd = l.coefd[tb + sb + jj];
#pragma unroll
                            for (int o = 0; o < OB; o++)
                                acc[o] = __builtin_fma(cd, W[o + NT_ - 1 - jj], acc[o]);
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
    constexpr int OB = (C % 8 == 0) ? 8 : (C % 4 == 0) ? 4 : (C % 3 == 0) ? 3 : 1;
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

// Orders <= 8 on a channel whose samples fit 16 bits (K0's narrow rows), when the
// prediction provably stays inside int32 (sum|coef| * 2^magbits < 2^31, checked by
// the caller): v_dot2_i32_i16 does two taps per instruction on int16 pairs and
// costs about what one fp64 FMA does, with no int -> fp64 conversions in front.
// Sample pairs R(k) = (lo: x[k], hi: x[k+1]) are packed from the int32 window;
// cp[j] = (lo: coef of tap 2j+2, hi: coef of tap 2j+1) comes from K2 (scalars).
template <int C, int T>
__device__ __forceinline__ void fir_lpc_dot8(const FastCtx<C, T> &e, int32_t (&r)[C], int order, int shift,
                                             const int32_t *__restrict__ cp)
{
    using Img = SmpImg<C, T>;
    static_assert(Img::V4 && C % 8 == 0, "fir_lpc_dot8: runs of 8 or 16");
    typedef short s2 __attribute__((ext_vector_type(2)));
    const int32_t *mine = e.l.smp + e.tid * Img::CS;
    const s2 q0 = __builtin_bit_cast(s2, cp[0]), q1 = __builtin_bit_cast(s2, cp[1]);
    const s2 q2 = __builtin_bit_cast(s2, cp[2]), q3 = __builtin_bit_cast(s2, cp[3]);
#pragma unroll
    for (int ob = 0; ob < C; ob += 8) {
        __builtin_amdgcn_sched_barrier(0);
        int32_t W[16];                                     // samples ob-8 .. ob+7
#pragma unroll
        for (int m4 = 0; m4 < 16; m4 += 4) {
            const int4 v = *reinterpret_cast<const int4 *>(mine + Img::off(ob - 8 + m4));
            W[m4] = v.x; W[m4 + 1] = v.y; W[m4 + 2] = v.z; W[m4 + 3] = v.w;
        }
        s2 R[14];                                          // R[m] = (x[ob-8+m], x[ob-7+m])
#pragma unroll
        for (int m = 0; m < 14; m++)
            R[m] = __builtin_bit_cast(s2, (int32_t)__builtin_amdgcn_perm((uint32_t)W[m + 1], (uint32_t)W[m], 0x05040100u));
#pragma unroll
        for (int o = 0; o < 8; o++) {
            // taps (1,2) use x[o-2], x[o-1] = R at window index o+6; (3,4): o+4; (5,6): o+2; (7,8): o
            int32_t acc = __builtin_amdgcn_sdot2(R[o + 6], q0, 0, false);
            acc = __builtin_amdgcn_sdot2(R[o + 4], q1, acc, false);
            acc = __builtin_amdgcn_sdot2(R[o + 2], q2, acc, false);
            acc = __builtin_amdgcn_sdot2(R[o], q3, acc, false);
            r[ob + o] = (int32_t)((uint32_t)W[8 + o] - (uint32_t)(acc >> shift));
        }
    }
    if (e.i0 < order) {
#pragma unroll
        for (int o = 0; o < C; o++)
            if (e.i0 + o < order) r[o] = mine[Img::off(o)];
    }
}

// optimize.c:34-68 encode_residual_fixed on the thread's run.  The reference
// computes in long long and stores to int32: the low 32 bits, which wrapping
// 32-bit arithmetic yields directly.
template <int C, int T>
__device__ __forceinline__ void fir_fixed(const FastCtx<C, T> &e, int32_t (&r)[C], int order)
{
    using Img = SmpImg<C, T>;
    const FastLds &l = e.l;
    const int32_t *mine = l.smp + e.tid * Img::CS;
    uint32_t h[4];
    uint32_t xs[C];
    if constexpr (Img::V4) {
        const int4 p = *reinterpret_cast<const int4 *>(mine + Img::off(-4));      // samples -4 .. -1
        h[0] = (uint32_t)p.w; h[1] = (uint32_t)p.z; h[2] = (uint32_t)p.y; h[3] = (uint32_t)p.x;
#pragma unroll
        for (int o = 0; o < C; o += 4) {
            const int4 v = *reinterpret_cast<const int4 *>(mine + Img::off(o));
            xs[o] = (uint32_t)v.x; xs[o + 1] = (uint32_t)v.y; xs[o + 2] = (uint32_t)v.z; xs[o + 3] = (uint32_t)v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) h[k] = (uint32_t)mine[Img::off(-1 - k)];
#pragma unroll
        for (int o = 0; o < C; o++) xs[o] = (uint32_t)mine[Img::off(o)];
    }
#pragma unroll
    for (int o = 0; o < C; o++) {
        const uint32_t x0 = xs[o];
        uint32_t acc;
        if (order == 0) acc = x0;
        else if (order == 1) acc = x0 - h[0];
        else if (order == 2) acc = x0 - 2u * h[0] + h[1];
        else if (order == 3) acc = x0 - 3u * h[0] + 3u * h[1] - h[2];
        else acc = x0 - 4u * h[0] + 6u * h[1] - 4u * h[2] + h[3];
        r[o] = (int32_t)acc;
        h[3] = h[2]; h[2] = h[1]; h[1] = h[0]; h[0] = x0;
    }
    if (e.i0 < order) {
#pragma unroll
        for (int o = 0; o < C; o++)
            if (e.i0 + o < order) r[o] = mine[Img::off(o)];
    }
}

// rice.c:122 folded residuals.  They stay in registers for the emit; the warm-up
// samples (partition 0 of every level starts at `order`, rice.c:85-94) are
// zeroed, which only the first threads have to do.
template <int C, int T>
__device__ __forceinline__ void fold_residuals(const FastCtx<C, T> &e, const int32_t (&r)[C],
                                               uint32_t (&u)[C], int order)
{
#pragma unroll
    for (int o = 0; o < C; o++) u[o] = zigzag32(r[o]);
    if (e.i0 < order) {
#pragma unroll
        for (int o = 0; o < C; o++)
            if (e.i0 + o < order) u[o] = 0u;
    }
}

// rice.c:105-187 on the residuals in r[]; all threads call it.
template <int C, int T>
__device__ __forceinline__ uint32_t rice_search_fast(const FastCtx<C, T> &e, const int32_t (&r)[C],
                                                     uint32_t (&u)[C], int order, bool lpc,
                                                     int *porder_out, int *method_out)
{
    constexpr int LT = clog2(T);                      // the thread level
    const FastLds &l = e.l;
    const int n = e.n, tid = e.tid, lane = e.lane;
    const int pmin = clamp_porder(e.pmin_req, n, order);
    const int pmax = clamp_porder(e.pmax_req, n, order);

    fold_residuals<C, T>(e, r, u, order);
    // thread-level sum
    unsigned long long v;
    if (e.obits <= 31 - clog2(C)) {
        // C folded values below 2^(32 - log2 C) each: the thread's sum fits 32 bits
        uint32_t v32 = 0;
#pragma unroll
        for (int o = 0; o < C; o++) v32 += u[o];
        v = v32;
    } else {
        v = 0;
#pragma unroll
        for (int o = 0; o < C; o++) v += u[o];
    }

    // (callers guarantee a barrier between the previous search's reads of
    // lvl_bits/lvl_meth and this reset)
    if (tid < 12) { l.lvl_bits[tid] = 0; l.lvl_meth[tid] = 0; }
    // levels LT .. LT-6 inside the wave: after s steps, lanes with the low s
    // bits clear hold the sums of level LT-s.  Steps 1,2,4,8 stay inside a
    // 16-lane row (DPP row_shl, no LDS); 16 and 32 cross rows.
#define PYR_STORE(S_)                                                                       \
    do {                                                                                    \
        const int lev_ = LT - (S_);                                                         \
        if (lev_ <= pmax && lev_ >= pmin && lev_ <= 8 && (lane & ((1 << (S_)) - 1)) == 0)   \
            l.sums[(1 << lev_) - 1 + (tid >> (S_))] = v;                                    \
    } while (0)
    PYR_STORE(0); v += row_shl_u64<1>(v);
    PYR_STORE(1); v += row_shl_u64<2>(v);
    PYR_STORE(2); v += row_shl_u64<4>(v);
    PYR_STORE(3); v += row_shl_u64<8>(v);
    PYR_STORE(4); v += __shfl_down(v, 16, WAVE);
    PYR_STORE(5); v += __shfl_down(v, 32, WAVE);
    PYR_STORE(6);
#undef PYR_STORE
    if (lane == 0) l.wtot[e.wv] = v;                 // level LT-6 node
    __syncthreads();
    STAMP(4);
    STAMP(5);
    {
        // one thread per (level, partition) node.  Levels above the waves
        // (LT-7 .. 0) are summed here from the per-wave totals.
        constexpr int NW = T / WAVE;
        const int first = (1 << pmin) - 1, last = (2 << pmax) - 2;
        for (int q = first + tid; q <= last; q += T) {
            const int p = ilog2_dev((uint32_t)(q + 1));
            const int jn = q + 1 - (1 << p);
            const int cnt = (n >> p) - (jn == 0 ? order : 0);
            unsigned long long sum;
            if (p <= LT - 7) {
                const int span = NW >> p;              // waves per node
                sum = 0;
                for (int w = 0; w < span; w++) sum += l.wtot[jn * span + w];
            } else {
                sum = l.sums[q];
            }
            uint32_t b;
            const int k = rice_k_fast(sum, cnt, &b);
            l.kpar[q] = k;
            atomicAdd(&l.lvl_bits[p], b);
            if (k > 14) atomicOr(&l.lvl_meth[0], 1u << p);      // one flag word: bit p = level p needs RICE2
        }
    }
    __syncthreads();
    STAMP(6);
    // rice.c:127-138, evaluated redundantly by every wave (no broadcast
    // barrier).  The inputs are workgroup-uniform: readfirstlane moves them to
    // SGPRs so that the comparison chain runs on the scalar unit.
    uint32_t lb[9];
#pragma unroll
    for (int p = 0; p < 9; p++) lb[p] = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.lvl_bits[p]);
    const uint32_t lmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.lvl_meth[0]);
    int bp = pmin;
    uint32_t best = 0, method = 0;
#pragma unroll
    for (int p = 0; p < 9; p++) {
        const uint32_t b = lb[p] + 4u * (1u << p);
        if (p >= pmin && p <= pmax && (p == pmin || b <= best)) { best = b; bp = p; method = (lmask >> p) & 1u; }
    }
    // rice.c:157-171
    uint32_t bits = (uint32_t)(order * e.obits + 2);
    if (lpc) bits += (uint32_t)(4 + 5 + order * e.precision);
    bits += best;
    bits += method + 4u;
    *porder_out = bp;
    *method_out = (int)method;
    STAMP(7);
    return bits;
}

// optimize.c:170-181 for the fixed predictors: the size estimates of all orders
// min_order..max_order (<= 4) from ONE pass over the samples.  The residual of order
// k+1 is the first difference of the residual of order k (optimize.c:34-68 written
// out), so a thread forms all five from its run and four samples of history, folds
// them (rice.c:122) and keeps five sums; one in-wave pyramid, one node pass and one
// level selection then serve all orders (five separate searches cost five times the
// barriers and LDS traffic).  Returns the order the reference picks (first strict
// minimum from min_order upward) together with its Rice result: the caller only has
// to form that order's residuals once more.  Needs pmax_req <= 5 (64 heap nodes per
// order).  LDS use: l.sums[k * 64 + node], wave totals l.sums[320 + k * 16 + wave],
// level bits l.kpar[k * 9 + p], RICE2 flags l.kpar[48 + k], parameters of every
// node l.kpar[64 + k * 64 + node] (the winner's move to l.kpar[node] at the end).
template <int C, int T>
__device__ __forceinline__ int fixed_search5(const FastCtx<C, T> &e, int min_order, int max_order,
                                             uint32_t *bits_out, int *porder_out, int *method_out)
{
    using Img = SmpImg<C, T>;
    constexpr int LT = clog2(T);
    constexpr int NW = T / WAVE;
    const FastLds &l = e.l;
    const int n = e.n, tid = e.tid, lane = e.lane;
    const int32_t *mine = l.smp + tid * Img::CS;

    uint32_t xs[C], h[4];
    if constexpr (Img::V4) {
        const int4 p = *reinterpret_cast<const int4 *>(mine + Img::off(-4));      // samples -4 .. -1
        h[0] = (uint32_t)p.w; h[1] = (uint32_t)p.z; h[2] = (uint32_t)p.y; h[3] = (uint32_t)p.x;
#pragma unroll
        for (int o = 0; o < C; o += 4) {
            const int4 v = *reinterpret_cast<const int4 *>(mine + Img::off(o));
            xs[o] = (uint32_t)v.x; xs[o + 1] = (uint32_t)v.y; xs[o + 2] = (uint32_t)v.z; xs[o + 3] = (uint32_t)v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) h[k] = (uint32_t)mine[Img::off(-1 - k)];
#pragma unroll
        for (int o = 0; o < C; o++) xs[o] = (uint32_t)mine[Img::off(o)];
    }
    // differences of orders 1..3 at the sample in front of the run
    uint32_t p1 = h[0] - h[1];
    uint32_t p2 = h[0] - 2u * h[1] + h[2];
    uint32_t p3 = h[0] - 3u * h[1] + 3u * h[2] - h[3];
    uint32_t p0 = h[0];

    // folded sums per order; a folded value is below 2^(obits+4)
    const bool sum32 = e.obits + 4 + clog2(C) <= 32;
    unsigned long long v[5];
    uint32_t a32[5] = {0, 0, 0, 0, 0};
    unsigned long long a64[5] = {0, 0, 0, 0, 0};
    const bool head = e.i0 < 4;                      // this run holds warm-up samples of some order
#pragma unroll
    for (int o = 0; o < C; o++) {
        const uint32_t d0 = xs[o];
        const uint32_t d1 = d0 - p0;
        const uint32_t d2 = d1 - p1;
        const uint32_t d3 = d2 - p2;
        const uint32_t d4 = d3 - p3;
        p0 = d0; p1 = d1; p2 = d2; p3 = d3;
        uint32_t z[5] = {zigzag32((int32_t)d0), zigzag32((int32_t)d1), zigzag32((int32_t)d2),
                         zigzag32((int32_t)d3), zigzag32((int32_t)d4)};
        if (head) {
#pragma unroll
            for (int k = 1; k < 5; k++)
                if (e.i0 + o < k) z[k] = 0u;         // rice.c:85-94: partition 0 starts at `order`
        }
        if (sum32) {
#pragma unroll
            for (int k = 0; k < 5; k++) a32[k] += z[k];
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) a64[k] += z[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) v[k] = sum32 ? (unsigned long long)a32[k] : a64[k];

    // partition-order window over all orders: order 0 has the loosest clamp, the
    // highest order the tightest (rice.c:148-155)
    const int pmin_lo = clamp_porder(e.pmin_req, n, max_order);
    const int pmax_hi = clamp_porder(e.pmax_req, n, min_order);

    if (tid < 64) l.kpar[tid] = 0;                   // level bits and RICE2 flags
#define PYR5_STORE(S_)                                                                      \
    do {                                                                                    \
        const int lev_ = LT - (S_);                                                         \
        if (lev_ <= pmax_hi && lev_ >= pmin_lo && (lane & ((1 << (S_)) - 1)) == 0) {        \
            _Pragma("unroll") for (int k = 0; k < 5; k++)                                   \
                l.sums[k * 64 + (1 << lev_) - 1 + (tid >> (S_))] = v[k];                    \
        }                                                                                   \
    } while (0)
    // a wave's total is below 2^(obits + 4 + log2(64 C)): in 32 bits one DPP add per step
    const bool wave32 = e.obits + 4 + clog2(C) + 6 <= 32;
    if (wave32) {
        uint32_t w[5];
#pragma unroll
        for (int k = 0; k < 5; k++) w[k] = (uint32_t)v[k];
#define PYR5_STEP32(S_, CTRL_)                                                              \
    do {                                                                                    \
        PYR5_STORE(S_);                                                                     \
        _Pragma("unroll") for (int k = 0; k < 5; k++) { w[k] += dpp_u32<CTRL_>(w[k]); v[k] = w[k]; } \
    } while (0)
        PYR5_STEP32(0, 0x101);
        PYR5_STEP32(1, 0x102);
        PYR5_STEP32(2, 0x104);
        PYR5_STEP32(3, 0x108);
#undef PYR5_STEP32
        PYR5_STORE(4);
#pragma unroll
        for (int k = 0; k < 5; k++) { w[k] += (uint32_t)__shfl_down((int)w[k], 16, WAVE); v[k] = w[k]; }
        PYR5_STORE(5);
#pragma unroll
        for (int k = 0; k < 5; k++) { w[k] += (uint32_t)__shfl_down((int)w[k], 32, WAVE); v[k] = w[k]; }
        PYR5_STORE(6);
    } else {
        PYR5_STORE(0);
#pragma unroll
        for (int k = 0; k < 5; k++) v[k] += row_shl_u64<1>(v[k]);
        PYR5_STORE(1);
#pragma unroll
        for (int k = 0; k < 5; k++) v[k] += row_shl_u64<2>(v[k]);
        PYR5_STORE(2);
#pragma unroll
        for (int k = 0; k < 5; k++) v[k] += row_shl_u64<4>(v[k]);
        PYR5_STORE(3);
#pragma unroll
        for (int k = 0; k < 5; k++) v[k] += row_shl_u64<8>(v[k]);
        PYR5_STORE(4);
#pragma unroll
        for (int k = 0; k < 5; k++) v[k] += __shfl_down(v[k], 16, WAVE);
        PYR5_STORE(5);
#pragma unroll
        for (int k = 0; k < 5; k++) v[k] += __shfl_down(v[k], 32, WAVE);
        PYR5_STORE(6);
    }
#undef PYR5_STORE
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; k++) l.sums[320 + k * 16 + e.wv] = v[k];
    }
    __syncthreads();

    {
        // one thread per (order, level, partition) node
        const int first = (1 << pmin_lo) - 1, last = (2 << pmax_hi) - 2;
        const int nn = last - first + 1;
        for (int it = tid; it < 5 * nn; it += T) {
            const int k = it / nn;
            const int q = first + (it - k * nn);
            const int p = ilog2_dev((uint32_t)(q + 1));
            const int jn = q + 1 - (1 << p);
            const int cnt = (n >> p) - (jn == 0 ? k : 0);
            unsigned long long sum;
            if (p <= LT - 7) {
                const int span = NW >> p;              // waves per node
                sum = 0;
                for (int w = 0; w < span; w++) sum += l.sums[320 + k * 16 + jn * span + w];
            } else {
                sum = l.sums[k * 64 + q];
            }
            uint32_t b;
            const int kk = rice_k_fast(sum, cnt, &b);
            l.kpar[64 + k * 64 + q] = kk;
            atomicAdd(reinterpret_cast<uint32_t *>(&l.kpar[k * 9 + p]), b);
            if (kk > 14) atomicOr(reinterpret_cast<uint32_t *>(&l.kpar[48 + k]), 1u << p);
        }
    }
    __syncthreads();

    // rice.c:127-138 and :157-171 per order, optimize.c:171-180 across them;
    // evaluated by every thread from workgroup-uniform LDS words
    int best = min_order, best_p = 0, best_m = 0;
    uint32_t best_bits = 0;
    for (int k = min_order; k <= max_order; k++) {
        const int pmin = clamp_porder(e.pmin_req, n, k);
        const int pmax = clamp_porder(e.pmax_req, n, k);
        const uint32_t lmask = (uint32_t)__builtin_amdgcn_readfirstlane(l.kpar[48 + k]);
        uint32_t lb = 0, method = 0;
        int bp = pmin;
        for (int p = pmin; p <= pmax; p++) {
            const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane(l.kpar[k * 9 + p]) + 4u * (1u << p);
            if (p == pmin || b <= lb) { lb = b; bp = p; method = (lmask >> p) & 1u; }
        }
        const uint32_t bits = (uint32_t)(k * e.obits + 2) + lb + method + 4u;
        if (k == min_order || bits < best_bits) { best_bits = bits; best = k; best_p = bp; best_m = (int)method; }
    }
    // the winner's parameters to where the emit and the info record read them
    __syncthreads();                                 // level words (l.kpar[0..52]) fully read
    if (tid < 64) l.kpar[tid] = l.kpar[64 + best * 64 + tid];
    __syncthreads();
    *bits_out = best_bits;
    *porder_out = best_p;
    *method_out = best_m;
    return best;
}

// OR `len` (<= 31) bits of val into the MSB-first bit string at bit `pos` of a
// zeroed LDS window, 32-bit arithmetic only; words outside [0, nw) are skipped.
__device__ __forceinline__ void put_bits32(uint32_t *win, int nw, long long pos, int len, uint32_t val)
{
    const long long wi = pos >> 5;
    const int off = (int)(pos & 31);
    const int room = 32 - off;
    if (len <= room) {
        if (wi >= 0 && wi < nw) atomicOr(&win[wi], val << (room - len));
    } else {
        const int spill = len - room;
        if (wi >= 0 && wi < nw) atomicOr(&win[wi], val >> spill);
        if (wi + 1 >= 0 && wi + 1 < nw) atomicOr(&win[wi + 1], val << (32 - spill));
    }
}

// MODE 0: the MAX / EST order methods (one quantised row, known before the kernel
// starts) -- the lean instance the headline workload runs; MODE 1: fixed predictors
// only (prediction_type FIXED: no LPC code, no fp64); MODE 2: everything --
// FIXED / NONE prediction and the order-search methods.
template <int C, int T, int MODE>
// At least 4 waves per SIMD (<= 128 VGPRs).  The kernel is bound by vector-ALU
// issue (PMC: ~910 VALU instructions per wave, > 80 % of the issue slots), so what
// pays is fewer instructions, not more waves: MODE 0 needs 96 VGPRs and runs five
// workgroups per CU (-3 %); forcing MODE 2 to 96 spills and is slower.
// Geometry for n = 4096, measured: (C,T) = (16,256) 94 us, (8,512) 137, (4,1024)
// 256, (32,128) 115 (206 VGPRs): cross-wave phases grow with T, serial ones with C.
__global__ __launch_bounds__(T, (MODE == 2 || C >= 16) ? 4 : 5)   // VGPR cap per waves/SIMD: 4 -> 128, 5 -> 96
void k_encode_pow2(fhip_params P, int n, int nsub, const int32_t *__restrict__ smp_all,
                   const int32_t *__restrict__ coefs_all, const int32_t *__restrict__ shift_all,
                   const int32_t *__restrict__ opt_all, const int32_t *__restrict__ fin_all,
                   fhip_subframe_info *__restrict__ info,
                   int32_t *__restrict__ res_out, uint8_t *__restrict__ bits_out, long long slot_bytes,
                   int narrow_ok)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    size_t off[12];
    fast_lds_layout(n, SmpImg<C, T>::SIZE, off);
    FastCtx<C, T> e;
    e.l.sums = reinterpret_cast<unsigned long long *>(lds_raw + off[0]);
    e.l.coefd = reinterpret_cast<double *>(lds_raw + off[1]);
    e.l.wtot = reinterpret_cast<unsigned long long *>(lds_raw + off[2]);
    e.l.smp = reinterpret_cast<int32_t *>(lds_raw + off[3]);
    e.l.kpar = reinterpret_cast<int32_t *>(lds_raw + off[4]);
    e.l.lvl_bits = reinterpret_cast<uint32_t *>(lds_raw + off[5]);
    e.l.lvl_meth = reinterpret_cast<uint32_t *>(lds_raw + off[6]);
    e.l.coef = reinterpret_cast<int32_t *>(lds_raw + off[7]);
    e.l.misc = reinterpret_cast<int32_t *>(lds_raw + off[8]);
    e.l.trial = reinterpret_cast<uint32_t *>(lds_raw + off[9]);
    e.l.bits = reinterpret_cast<uint32_t *>(lds_raw + off[10]);
    const FastLds &l = e.l;

    // fp64 rounding toward -inf for the whole kernel (MODE[3:2] = 2): every fp64
    // operation in here is exact except the one fma in fir_lpc that wants a floor.
    // As inline asm: after the builtin the compiler re-asserts the default mode in
    // front of the next fp64 instruction.
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2" ::: "memory");

    const int tid = threadIdx.x;
    e.n = n; e.tid = tid; e.lane = tid & 63; e.wv = tid >> 6;
    e.i0 = tid * C;
    e.precision = P.lpc_precision;
    e.pmin_req = P.min_partition_order;
    e.pmax_req = P.max_partition_order;

    // MAX / EST: the one row the reference quantises is known before the
    // search starts and comes compact from K2
    constexpr bool MULTI = MODE != 0;
    constexpr bool HAS_LPC = MODE != 1;
    constexpr bool pre_row = !MULTI;     // launcher: prediction_type == 2, n > max order, order method <= 1

    // One workgroup per subframe (a persistent variant that prefetched the next
    // subframe into registers measured slower: the hardware's own dispatch of a
    // fresh workgroup per subframe balances better and costs no VGPRs).
    const int s = blockIdx.x;
    int32_t xn[C];
    int32_t first_n, obits_n, fcoef_n = 0, fshift_n = 0, forder_n = 0, fcabs_n = 0, magbits_n = -1;
    {
        const int32_t *srcp = smp_all + (size_t)s * n;
        // K0 may have stored this row as int16 (info.reserved, honoured only when the
        // launcher says the flag is K0's): half the loads, one sign extension per sample
        const int nflag = (C % 8 == 0 && narrow_ok) ? info[s].reserved : 0;     // 0, or 1 + bit length of max |x|
        const bool narrow = nflag != 0;
        magbits_n = nflag - 1;
        if (C % 8 == 0 && narrow) {
            const int4 *src4 = reinterpret_cast<const int4 *>(reinterpret_cast<const int16_t *>(srcp) + e.i0);
#pragma unroll
            for (int q = 0; q < C / 8; q++) {
                const int4 t4 = src4[q];
                xn[8 * q] = (int32_t)(int16_t)t4.x;     xn[8 * q + 1] = t4.x >> 16;
                xn[8 * q + 2] = (int32_t)(int16_t)t4.y; xn[8 * q + 3] = t4.y >> 16;
                xn[8 * q + 4] = (int32_t)(int16_t)t4.z; xn[8 * q + 5] = t4.z >> 16;
                xn[8 * q + 6] = (int32_t)(int16_t)t4.w; xn[8 * q + 7] = t4.w >> 16;
            }
            first_n = (int32_t)*reinterpret_cast<const int16_t *>(srcp);
        } else if (C % 4 == 0) {
            // 16-byte lane accesses of the thread's own run
            const int4 *src4 = reinterpret_cast<const int4 *>(srcp + e.i0);
#pragma unroll
            for (int q = 0; q < C / 4; q++) {
                const int4 t4 = src4[q];
                xn[4 * q] = t4.x; xn[4 * q + 1] = t4.y; xn[4 * q + 2] = t4.z; xn[4 * q + 3] = t4.w;
            }
        } else {
            // runs of 3 or 9 samples: coalesced dword loads, element tid + T*q
#pragma unroll
            for (int q = 0; q < C; q++) xn[q] = srcp[tid + T * q];
        }
        if (!narrow) first_n = srcp[0];
        obits_n = info[s].obits;
        if (pre_row) {
            const int32_t *f = fin_all + (size_t)s * FIN_STRIDE;
            fcoef_n = f[tid & 31];
            fshift_n = f[32];
            forder_n = f[33];
            fcabs_n = f[34];
        }
    }
  {
    fhip_subframe_info *out = &info[s];
    e.obits = obits_n;
    const int fshift = fshift_n, forder = forder_n;

    STAMP(0);
    // ---- stage this subframe in LDS ------------------------------------------
    // "all samples equal the first" (CONSTANT, optimize.c:143-151) is max == min
    // == first: running max / min cost one three-input instruction per two samples
    int differs = 0;
    {
        const int32_t first = first_n;
        int32_t mx = first, mn = first;
#pragma unroll
        for (int o = 0; o < C; o++) {
            const int32_t v = xn[o];
            // own-run mapping: sample i0 + o; coalesced mapping: sample tid + T*o
            if (C % 4 != 0)       // coalesced mapping: this register holds sample tid + T*o of the block
                l.smp[SmpImg<C, T>::at((tid + T * o) / C + SmpImg<C, T>::COL0, (tid + T * o) % C)] = v;
            mx = max(mx, v);
            mn = min(mn, v);
        }
        differs = (mx != mn);
        if constexpr (C % 4 == 0) {
            // own-run mapping: the registers are samples i0 .. i0+C-1: 16-byte stores
#pragma unroll
            for (int g4 = 0; g4 < C; g4 += 4)
                *reinterpret_cast<int4 *>(l.smp + tid * 4 + SmpImg<C, T>::off(g4)) =
                    make_int4(xn[g4], xn[g4 + 1], xn[g4 + 2], xn[g4 + 3]);
        }
    }
    // zeros in front: columns 0 .. COL0-1 of every row
    if (tid < SmpImg<C, T>::COL0 * C) l.smp[SmpImg<C, T>::at(tid / C, tid % C)] = 0;
    // the first emit window is cleared here, under the shadow of the loads above
    const int wwords = fast_window_words(n);
    if (bits_out) for (int q = tid; q < wwords / 4; q += T) reinterpret_cast<uint4 *>(l.bits)[q] = make_uint4(0, 0, 0, 0);
    if (tid < 16) l.coefd[32 + tid] = 0.0;
    if (pre_row && tid < FHIP_MAX_ORDER) {
        l.coef[tid] = fcoef_n;
        l.coefd[tid] = (double)fcoef_n;
    }
    const bool constant = (__syncthreads_or(differs) == 0);
    STAMP(1);

    int32_t r[C];                        // residuals of the current candidate
    uint32_t u[C];                       // ... folded (rice.c:122), warm-up zeroed: what the emit reads
    int type, type_code, order = 0, shift = 0;
    uint32_t est_bits = 0;
    bool has_rice = false;
    const int32_t *crow_base = coefs_all + (size_t)s * FHIP_MAX_ORDER * FHIP_MAX_ORDER;
    const int32_t *srow = shift_all + (size_t)s * FHIP_MAX_ORDER;

    // decision tree of encode_residual() (optimize.c:124-276): same candidate
    // loop as k_encode, every variable workgroup-uniform
    enum { T_CONST, T_VERB, T_FIXED, T_LPC } tree;
    if (constant) tree = T_CONST;
    else if (!MULTI) tree = T_LPC;
    else if (MODE == 1) tree = T_FIXED;                     // launcher: prediction_type == 1, n >= 5
    else if (n < 5 || P.prediction_type == 0) tree = T_VERB;
    else if (P.prediction_type == 1 || n <= P.max_prediction_order) tree = T_FIXED;
    else tree = T_LPC;

    const int omethod = MULTI ? P.order_method : 0;
    const int min_order = P.min_prediction_order;
    const int max_order = (tree == T_FIXED) ? min(P.max_prediction_order, 4) : P.max_prediction_order;

    int it = 0, best = 0;
    uint32_t best_bits = 0, last_bits = 0;
    bool have_best = false;
    int lg_step = 16, lg_last = 0, lg_pos = 3;
    bool final_pass = false;             // MAX / EST: the one row is the result
    int porder = 0, method = 0;          // of the most recent Rice search
    // (Keeping the winner's Rice result instead of searching it again after the
    // order search -- optimize.c:183-187, :265-274 -- was measured: the extra live
    // state costs a wave of occupancy and the kernel ends up 15 % slower.)

    if (tree == T_FIXED) { it = min_order; best = min_order; }
    bool five_wide = false;
    if constexpr (MODE == 1) {
        // all fixed orders from one pass; the winner is then encoded like a single candidate
        if (tree == T_FIXED && max_order > min_order && max_order <= 4 && min_order >= 0 &&
            P.max_partition_order <= 5) {
            five_wide = true;
        }
    }
    if (HAS_LPC && tree == T_LPC) {
        if (omethod <= 1) { best = forder - 1; final_pass = true; }     // MAX: max_order, EST: est
        else if (omethod <= 4) { it = (1 << ((omethod - 1) & 7)) - 1; best = max_order - 1; }
        else if (omethod == 5) { it = 0; best = 0; }
        else {
            best = min_order - 1 + (max_order - min_order) / 3;
            if (tid < FHIP_MAX_ORDER) l.trial[tid] = 0xFFFFFFFFu;
            __syncthreads();
            lg_step = 32;
        }
    }

    if (tree == T_CONST || tree == T_VERB) {
        type = type_code = (tree == T_CONST) ? FHIP_SUB_CONSTANT : FHIP_SUB_VERBATIM;
        est_bits = (uint32_t)(tree == T_CONST ? e.obits : e.obits * n);
    } else {
        if constexpr (MODE == 1) {
            if (five_wide) {
                best = fixed_search5<C, T>(e, min_order, max_order, &est_bits, &porder, &method);
                fir_fixed<C, T>(e, r, best);
                fold_residuals<C, T>(e, r, u, best);
            }
        }
        if (!five_wide) for (;;) {
            int cand = -1;
            if (!final_pass) {
                if (tree == T_FIXED) {
                    if (it <= max_order) cand = it;
                } else if (omethod <= 4) {
                    if (it >= 0) {
                        const int levels = 1 << ((omethod - 1) & 7);
                        cand = min_order + (((max_order - min_order + 1) * (it + 1)) / levels) - 2;
                        if (cand < 0) cand = 0;
                    }
                } else if (omethod == 5) {
                    if (it < max_order) cand = it;
                } else {
                    for (;;) {
                        if (lg_pos == 3) {
                            lg_step >>= 1;
                            if (lg_step == 0) break;
                            lg_last = best;
                            lg_pos = 0;
                        }
                        const int i = lg_last + (lg_pos - 1) * lg_step;
                        lg_pos++;
                        if (i < min_order - 1 || i >= max_order || l.trial[i] < 0xFFFFFFFFu) continue;
                        cand = i;
                        break;
                    }
                }
                if (cand < 0) {
                    if (tree == T_FIXED && best == max_order) { est_bits = last_bits; break; }
                    final_pass = true;
                }
            }
            if (final_pass) cand = best;

            uint32_t b = 0;
            if (MULTI && (!HAS_LPC || tree == T_FIXED)) {
                fir_fixed<C, T>(e, r, cand);
                __syncthreads();                      // previous search fully read
                b = rice_search_fast<C, T>(e, r, u, cand, false, &porder, &method);
            } else if constexpr (HAS_LPC) {
                const int ord = cand + 1;
                int cshift;
                if (pre_row) {
                    cshift = fshift;                  // coef/coefd were staged with the samples
                } else {
                    __syncthreads();                  // readers of coef/coefd/lvl_* are done
                    if (tid < FHIP_MAX_ORDER) {
                        const int32_t cv = (tid < ord) ? crow_base[cand * FHIP_MAX_ORDER + tid] : 0;
                        l.coef[tid] = cv;
                        l.coefd[tid] = (double)cv;
                    }
                    cshift = srow[cand];
                    __syncthreads();
                }
                STAMP(2);
                bool done = false;
                if constexpr (C % 8 == 0) {
                    // 16-bit samples and a prediction that cannot leave int32: packed dot products
                    if (pre_row && ord <= 8 && magbits_n >= 0 &&
                        ((unsigned long long)(uint32_t)fcabs_n << magbits_n) < (1ull << 31)) {
                        fir_lpc_dot8<C, T>(e, r, ord, cshift, fin_all + (size_t)s * FIN_STRIDE + FIN_PAIRS);
                        done = true;
                    }
                }
                if (done) {
                } else if (pre_row && ord <= 8)
                    fir_lpc_o8<C, T>(e, r, ord, cshift,
                                     reinterpret_cast<const double *>(fin_all + (size_t)s * FIN_STRIDE + FIN_DBL));
                else
                    fir_lpc<C, T>(e, r, ord, cshift);
                STAMP(3);
                b = rice_search_fast<C, T>(e, r, u, ord, true, &porder, &method);
                STAMP(8);
            }
            if (final_pass) { est_bits = b; break; }

            last_bits = b;
            if (tree == T_FIXED) {
                if (!have_best || b < best_bits) { best_bits = b; best = cand; }
                it++;
            } else if (omethod <= 4) {
                if (!have_best) best_bits = b;
                else if (b < best_bits) { best_bits = b; best = cand; }
                it--;
            } else if (omethod == 5) {
                if (!have_best || b < best_bits) { best_bits = b; best = cand; }
                it++;
            } else {
                if (tid == 0) l.trial[cand] = b;
                __syncthreads();
                if (b < l.trial[best]) best = cand;
            }
            have_best = true;

        }
        if (tree == T_FIXED) {
            order = best;
            type = FHIP_SUB_FIXED;
            type_code = FHIP_SUB_FIXED | order;
        } else {
            order = best + 1;
            shift = pre_row ? fshift : srow[best];
            type = FHIP_SUB_LPC;
            type_code = FHIP_SUB_LPC | (order - 1);
        }
        has_rice = true;
    }

    if (!has_rice) { porder = 0; method = 0; }

    STAMP(9);
    if (res_out) {
        // FlacSubframe.residual: the samples themselves for CONSTANT / VERBATIM and
        // for warm-up positions, else the fold undone (a bijection on 32 bits)
        const int32_t *mine_s = l.smp + tid * SmpImg<C, T>::CS;
#pragma unroll
        for (int o = 0; o < C; o++) {
            const int32_t back = (int32_t)((u[o] >> 1) ^ (0u - (u[o] & 1u)));
            r[o] = (!has_rice || e.i0 + o < order) ? mine_s[SmpImg<C, T>::off(o)] : back;
        }
        if (C % 4 == 0) {
            int4 *dst4 = reinterpret_cast<int4 *>(res_out + (size_t)s * n + e.i0);
#pragma unroll
            for (int q = 0; q < C / 4; q++)
                dst4[q] = make_int4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
        } else {
            int32_t *dst = res_out + (size_t)s * n + e.i0;
#pragma unroll
            for (int o = 0; o < C; o++) dst[o] = r[o];
        }
    }

    // ---- encode.c:766-798 output_residual -----------------------------------
    long long total_bits = 0;
    if (has_rice) {
        constexpr int LT = clog2(T);
        const int pbits = 4 + method;
        const int heap0 = (1 << porder) - 1;
        const int tpp = LT - porder;                       // log2(threads per partition)
        const int part = tid >> tpp;
        const int k = l.kpar[heap0 + part];
        const int k1 = k + 1;
        const bool part_head = (part > 0) && ((tid & ((1 << tpp) - 1)) == 0);
        // warm-up samples at the front of this thread's run (first threads only)
        const int nwarm = min(max(order - e.i0, 0), C);
        // the emit-side fold (bitio.h:128) differs from rice.c's for |x| >= 2^30
        if (e.obits > 30) {
#pragma unroll
            for (int o = 0; o < C; o++)
                u[o] = emit_fold32((int32_t)((u[o] >> 1) ^ (0u - (u[o] & 1u)))) & ((e.i0 + o < order) ? 0u : ~0u);
        }
        // codeword lengths of the run; in 32 bits unless a quotient is huge.
        // A zeroed warm-up entry counts k+1 bits here, taken off again below.
        uint32_t umax = 0;
#pragma unroll
        for (int o = 0; o < C; o++) umax = max(umax, u[o]);
        const uint32_t longest = umax >> k;
        // every codeword of the wave at most 32 bits: one flush test per codeword
        const bool short_codes = !__any(longest + (uint32_t)k1 > 32u);
        const bool tiny_codes = (C % 2 == 0) && !__any(longest + (uint32_t)k1 > 16u);
        unsigned long long mine = (part_head ? pbits : 0) + (unsigned long long)((C - nwarm) * k1);
        if (short_codes) {
            uint32_t m32 = 0;
#pragma unroll
            for (int o = 0; o < C; o++) m32 += u[o] >> k;
            mine += m32;
        } else {
#pragma unroll
            for (int o = 0; o < C; o++) mine += (unsigned long long)(u[o] >> k);
        }
        // in-wave offsets: DPP scan in 32 bits unless some lane of the wave
        // holds an absurdly long run (then the exact 64-bit shuffle scan)
        unsigned long long incl;
        if (__any(mine >> 24)) incl = wave_incl_scan_u64(mine, e.lane);
        else incl = wave_incl_scan_u32_dpp((uint32_t)mine);
        if (e.lane == 63) l.wtot[e.wv] = incl;
        __syncthreads();
        unsigned long long base = 6 + pbits;
        unsigned long long tot = 6 + pbits;
#pragma unroll
        for (int w = 0; w < T / WAVE; w++) {
            const unsigned long long wt = l.wtot[w];
            if (w < e.wv) base += wt;
            tot += wt;
        }
        const unsigned long long my_off = base + incl - mine;
        total_bits = (tot > 0x7FFFFFFFull) ? 0x7FFFFFFFll : (long long)tot;
        STAMP(10);

        if (bits_out) {
            if (tot > (unsigned long long)slot_bytes * 8ull) {
                total_bits = -1;
            } else {
                uint32_t *dst32 = reinterpret_cast<uint32_t *>(bits_out + (size_t)s * slot_bytes);
                const int nwords = (int)((tot + 31) >> 5);
                // Every thread writes all C codewords, unconditionally.  A zeroed
                // warm-up entry is the k+1-bit code of 0; the run of a thread that
                // has some starts that many bits early, so those land in front of
                // the thread's first real codeword -- i.e. in bits [.., 6+pbits) of
                // the section (only threads at the start of partition 0 have warm-up
                // samples), which thread 0 overwrites with the section header after
                // the barrier.
                const long long start = (long long)my_off - (long long)(nwarm * k1);
                for (int wlo = 0; wlo < nwords; wlo += wwords) {
                    const int nw = min(wwords, nwords - wlo);
                    if (wlo > 0) {
                        // later windows reuse the buffer (the first was cleared at the top)
                        __syncthreads();
                        for (int q = tid; q < (nw + 3) / 4; q += T) reinterpret_cast<uint4 *>(l.bits)[q] = make_uint4(0, 0, 0, 0);
                        __syncthreads();
                    }
                    const long long rel = start - (long long)wlo * 32;
                    // The thread's codewords form one contiguous bit run.  It is
                    // assembled MSB-first in a 32-bit register and leaves a word at
                    // a time by LDS OR (the run's first and last word are shared
                    // with the neighbours; OR-ing the interior ones too costs the
                    // same LDS issue slot as a store and needs no bookkeeping).
                    uint32_t hi = 0;
                    int nacc = (int)(rel & 31);
                    int w = (int)(rel >> 5);
                    // append a field of len <= 32 bits (val < 2^len; len 0 => val 0);
                    // at most one word leaves
                    auto field = [&](int len, uint32_t val) {
                        const uint32_t a = val << ((32 - len) & 31);        // left-aligned
                        const uint32_t head = a >> nacc;
                        // a << (32 - nacc), and 0 for nacc == 0
                        const uint32_t tail = __builtin_amdgcn_alignbit(a, 0u, (uint32_t)nacc);
                        const int t = nacc + len;
                        const uint32_t word = hi | head;
                        const bool full = t >= 32;
                        if (full && (unsigned)w < (unsigned)nw) atomicOr(&l.bits[w], word);
                        hi = full ? tail : word;
                        w += full ? 1 : 0;
                        nacc = t & 31;
                    };
                    if (part_head) field(pbits, (uint32_t)k);
                    const uint32_t kmask = (1u << k) - 1u, kbit = 1u << k;
                    bool packed = false;
                    if constexpr (C % 2 == 0) { if (tiny_codes) {
                        packed = true;
                        // every codeword of the wave <= 16 bits: two codewords are
                        // one field of <= 32 bits (half the append/flush work)
#pragma unroll
                        for (int o = 0; o < C; o += 2) {
                            const int l1 = (int)(u[o + 1] >> k) + k1;
                            const uint32_t v0 = (u[o] & kmask) | kbit, v1 = (u[o + 1] & kmask) | kbit;
                            field((int)(u[o] >> k) + k1 + l1, (v0 << l1) | v1);
                        }
                    } }
                    if (packed) {
                    } else if (short_codes) {
                        // bitio.h:120-141: q zeros, a one, k low bits -- as one field
                        // of q+k+1 <= 32 bits
#pragma unroll
                        for (int o = 0; o < C; o++)
                            field((int)(u[o] >> k) + k1, (u[o] & kmask) | kbit);
                    } else {
#pragma unroll 2
                        for (int o = 0; o < C; o++) {
                            const uint32_t q = u[o] >> k;
                            if (q >= 32u) {
                                // long unary run: the pending word leaves, whole zero
                                // words are skipped (the window is zero-filled)
                                const long long adv = (long long)nacc + q;
                                if (hi && (unsigned)w < (unsigned)nw) atomicOr(&l.bits[w], hi);
                                hi = 0;
                                w += (int)(adv >> 5);
                                nacc = (int)(adv & 31);
                            } else {
                                field((int)q, 0u);
                            }
                            field(k1, (u[o] & kmask) | kbit);
                        }
                    }
                    if (nacc > 0 && hi && (unsigned)w < (unsigned)nw) atomicOr(&l.bits[w], hi);
                    __syncthreads();
                    if (tid == 0 && wlo == 0) {
                        // section header (encode.c:771-776): method, partition order,
                        // first parameter; replaces whatever warm-up filler landed there
                        const int hb = 6 + pbits;
                        const uint32_t hdr = ((uint32_t)method << (4 + pbits)) | ((uint32_t)porder << pbits) |
                                             (uint32_t)l.kpar[heap0];
                        l.bits[0] = (l.bits[0] & (0xFFFFFFFFu >> hb)) | (hdr << (32 - hb));
                    }
                    STAMP(11);
                    // 16 bytes per lane where whole quads of words are left (the LDS window
                    // is 16-byte aligned; a slot need only be dword aligned, which is all a
                    // global dwordx4 store asks for), single words for the last 1..3
                    const int nq = nw >> 2;
                    for (int q = tid; q < nq; q += T) {
                        const uint4 v = reinterpret_cast<const uint4 *>(l.bits)[q];
                        reinterpret_cast<uint4 *>(dst32 + wlo)[q] =
                            make_uint4(__builtin_bswap32(v.x), __builtin_bswap32(v.y),
                                       __builtin_bswap32(v.z), __builtin_bswap32(v.w));
                    }
                    if (tid < (nw & 3)) dst32[wlo + 4 * nq + tid] = __builtin_bswap32(l.bits[4 * nq + tid]);
                }
            }
        }
    }
    STAMP(12);

    if (tid == 0) {
        out->type = type;
        out->type_code = type_code;
        out->order = order;
        out->shift = shift;
        out->rice_method = method;
        out->porder = porder;
        out->est_bits = est_bits;
        out->rice_nbits = (int32_t)total_bits;
        out->reserved = 0;
    }
    if (tid < FHIP_MAX_ORDER) {
        out->coefs[tid] = (type == FHIP_SUB_LPC && tid < order) ? l.coef[tid] : 0;
        const int nw = (type == FHIP_SUB_CONSTANT) ? 1 : order;
        out->warmup[tid] = (tid < nw) ? l.smp[SmpImg<C, T>::at(tid / C + SmpImg<C, T>::COL0, tid % C)] : 0;
    }
    {
        const int np = has_rice ? (1 << porder) : 0;
        for (int q = tid; q < FHIP_MAX_PARTS; q += T) out->rparams[q] = (q < np) ? l.kpar[np - 1 + q] : 0;
    }
  }
}


// ---------------------------------------------------------------------------
// K4  k_assemble -- whole FLAC frames on the device
// ---------------------------------------------------------------------------
// One workgroup per frame turns the side information of K0..K3 and the packed
// residual sections into the finished frame: frame header + CRC-8
// (encode.c:718-764), subframe headers, warm-up samples, LPC header
// (encode.c:800-905), the residual sections appended bit for bit, byte
// alignment, CRC-16 (encode.c:907-917), and the verbatim fallback of
// encode.c:949-964 when the frame would exceed its verbatim size or a section
// did not fit its slot.
//
// The frame is a concatenation of a few bit strings ("segments"): the frame
// header and one small prefix per subframe are built serially in LDS by one
// thread each; the bodies are the residual slots in HBM (or, for VERBATIM
// subframes, the samples themselves).  Every output dword is then produced by
// exactly one thread from the segments that overlap it, so stores are plain and
// coalesced.  CRC-16 is linear: each thread takes the CRC of a contiguous byte
// chunk and the partial CRCs are merged in a log-step tree with the constants
// x^(8*chunk*2^j) mod P.
constexpr int ASM_MAX_SEG = 2 * FHIP_MAX_CH + 2;
constexpr int ASM_PREFIX_BYTES = 224;     // 8+33 header bits, 32 warm-ups of <= 32 bits, 9 + 32*15 coef bits

struct AsmSeg { int kind; int ch; long long nbits; long long dst; };   // kind: 0 LDS bytes, 1 rice slot, 2 verbatim samples

struct MiniSink {                          // MSB-first writer into LDS bytes (serial, one thread)
    uint8_t *buf; int nbits;
    __device__ void put(int nb, uint32_t v)
    {
        for (int b = nb - 1; b >= 0; b--) {
            const int pos = nbits++;
            const uint32_t bit = (b < 32) ? ((v >> b) & 1u) : 0u;   // fields wider than 32 bits are zero-extended
            if ((pos & 7) == 0) buf[pos >> 3] = 0;
            buf[pos >> 3] |= (uint8_t)(bit << (7 - (pos & 7)));
        }
    }
};

__device__ __forceinline__ uint16_t crc16_mulmod(uint16_t a, uint16_t b)
{
    // a * b mod x^16 + x^15 + x^2 + 1 over GF(2)
    uint32_t r = 0;
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        r <<= 1;
        if (r & 0x10000u) r ^= 0x18005u;
        if ((b >> i) & 1u) r ^= a;
    }
    return (uint16_t)r;
}

__device__ __forceinline__ int32_t asm_sample(const int32_t *pcm_frame, int nch, int ch, int t,
                                             int ch_mode, int wasted)
{
    // FlacSubframe.samples recomputed (encode.c:648-694, :558-593)
    int32_t v;
    if (nch == 2 && ch_mode != FHIP_CH_LEFT_RIGHT) {
        const int32_t l = pcm_frame[2 * t], r = pcm_frame[2 * t + 1];
        const int32_t side = (int32_t)((uint32_t)l - (uint32_t)r);
        if (ch_mode == FHIP_CH_MID_SIDE) v = ch ? side : ((int32_t)((uint32_t)l + (uint32_t)r) >> 1);
        else if (ch_mode == FHIP_CH_LEFT_SIDE) v = ch ? side : l;
        else v = ch ? r : side;
    } else {
        v = pcm_frame[(size_t)t * nch + ch];
    }
    return v >> wasted;
}

__global__ __launch_bounds__(NT)
void k_assemble(fhip_params P, int n, const int32_t *__restrict__ pcm,
                const fhip_subframe_info *__restrict__ info, const uint8_t *__restrict__ rice,
                long long slot_bytes, uint8_t *__restrict__ frames, long long frame_stride,
                int32_t *__restrict__ frame_bytes, uint32_t number_base, uint32_t number_step,
                const uint32_t *__restrict__ numbers,
                int sr_code0, int sr_code1, int bps_code, int verbatim_size)
{
    __shared__ uint8_t s_hdr[32];
    __shared__ uint8_t s_prefix[FHIP_MAX_CH][ASM_PREFIX_BYTES];
    __shared__ AsmSeg s_seg[ASM_MAX_SEG];
    __shared__ int s_nseg, s_verbatim, s_hdr_bits;
    __shared__ long long s_total_bits;
    __shared__ uint16_t s_crc_tab[256];
    __shared__ uint16_t s_part[NT];
    __shared__ int s_info[FHIP_MAX_CH][8];       // type, type_code, order, shift, obits, wasted, rice_nbits, ch_mode

    const int tid = threadIdx.x;
    const int f = blockIdx.x;
    const int nch = P.channels;
    const fhip_subframe_info *fi = info + (size_t)f * nch;
    const int32_t *pcm_frame = pcm + (size_t)f * n * nch;
    uint8_t *out = frames + (size_t)f * frame_stride;
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out);

    // CRC-16 table (crc.c:24-44), one entry per thread
    {
        uint16_t c = (uint16_t)(tid << 8);
#pragma unroll
        for (int b = 0; b < 8; b++) c = (uint16_t)((c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1));
        s_crc_tab[tid] = c;
    }
    if (tid < nch) {
        const fhip_subframe_info *i = &fi[tid];
        s_info[tid][0] = i->type; s_info[tid][1] = i->type_code; s_info[tid][2] = i->order;
        s_info[tid][3] = i->shift; s_info[tid][4] = i->obits; s_info[tid][5] = i->wasted;
        s_info[tid][6] = i->rice_nbits; s_info[tid][7] = i->ch_mode;
    }
    __syncthreads();

    // ---- does the frame take the verbatim fallback? (encode.c:949) -----------
    if (tid == 0) {
        // frame header (encode.c:718-764) + CRC-8
        MiniSink hs{s_hdr, 0};
        int bs0 = -1, bs1 = -1;
        const int bs_tab[15] = {0, 192, 576, 1152, 2304, 4608, 0, 0, 256, 512, 1024, 2048, 4096, 8192, 16384};
        for (int q = 0; q < 15; q++) if (n == bs_tab[q]) { bs0 = q; break; }
        if (bs0 < 0) { bs0 = (n <= 256) ? 6 : 7; bs1 = n - 1; }
        const int ch_mode = s_info[0][7];
        const uint32_t number = numbers ? numbers[f] : number_base + (uint32_t)f * number_step;
        hs.put(15, 0x7FFC);
        hs.put(1, (uint32_t)P.allow_vbs);
        hs.put(4, (uint32_t)bs0);
        hs.put(4, (uint32_t)sr_code0);
        hs.put(4, (uint32_t)(ch_mode == FHIP_CH_NOT_STEREO ? nch - 1 : ch_mode));
        hs.put(3, (uint32_t)bps_code);
        hs.put(1, 0);
        if (number < 0x80) {
            hs.put(8, number);
        } else {
            const int bytes = (ilog2_dev(number) + 4) / 5;          // encode.c:696-716
            int sh = (bytes - 1) * 6;
            hs.put(8, ((256u - (256u >> bytes)) | (number >> sh)) & 0xFFu);
            while (sh >= 6) { sh -= 6; hs.put(8, 0x80u | ((number >> sh) & 0x3Fu)); }
        }
        if (bs1 >= 0) hs.put(bs1 < 256 ? 8 : 16, (uint32_t)bs1);
        if (sr_code1 > 0) hs.put(sr_code1 < 256 ? 8 : 16, (uint32_t)sr_code1);
        uint8_t c8 = 0;
        for (int q = 0; q < (hs.nbits >> 3); q++) {
            c8 ^= s_hdr[q];
            for (int b = 0; b < 8; b++) c8 = (uint8_t)((c8 & 0x80) ? ((c8 << 1) ^ 0x07) : (c8 << 1));
        }
        hs.put(8, c8);
        s_hdr_bits = hs.nbits;

        long long bits = hs.nbits;
        int verb = 0;
        for (int c = 0; c < nch; c++) {
            const int type = s_info[c][0], order = s_info[c][2], obits = s_info[c][4];
            const int wasted = s_info[c][5], rn = s_info[c][6];
            bits += 8 + (wasted ? wasted : 0);
            if (type == FHIP_SUB_CONSTANT) bits += obits;
            else if (type == FHIP_SUB_VERBATIM) bits += (long long)n * obits;
            else {
                if (rn < 0) verb = 1;
                bits += (long long)order * obits + rn;
                if (type == FHIP_SUB_LPC) bits += 9 + order * P.lpc_precision;
            }
        }
        const long long bytes = ((bits + 7) >> 3) + 2;
        if (bytes > verbatim_size) verb = 1;
        s_verbatim = verb;
    }
    __syncthreads();
    const int verbatim = s_verbatim;

    // ---- per-subframe prefixes (encode.c:871-905, 800-869), one thread each --
    if (tid < nch) {
        const fhip_subframe_info *i = &fi[tid];
        const int type = verbatim ? FHIP_SUB_VERBATIM : s_info[tid][0];
        const int order = s_info[tid][2], obits = s_info[tid][4], wasted = s_info[tid][5];
        MiniSink ps{s_prefix[tid], 0};
        ps.put(1, 0);
        ps.put(6, (uint32_t)(verbatim ? FHIP_SUB_VERBATIM : s_info[tid][1]));
        if (wasted) { ps.put(1, 1); ps.put(wasted - 1, 0); ps.put(1, 1); }
        else ps.put(1, 0);
        const uint32_t omask = (obits >= 32) ? 0xFFFFFFFFu : ((1u << obits) - 1u);
        if (type == FHIP_SUB_CONSTANT) {
            ps.put(obits, (uint32_t)i->warmup[0] & omask);
        } else if (type == FHIP_SUB_FIXED || type == FHIP_SUB_LPC) {
            for (int t = 0; t < order; t++) ps.put(obits, (uint32_t)i->warmup[t] & omask);
            if (type == FHIP_SUB_LPC) {
                ps.put(4, (uint32_t)(P.lpc_precision - 1));
                ps.put(5, (uint32_t)s_info[tid][3] & 31u);
                const uint32_t cmask = (1u << P.lpc_precision) - 1u;
                for (int t = 0; t < order; t++) ps.put(P.lpc_precision, (uint32_t)i->coefs[t] & cmask);
            }
        }
        s_info[tid][0] = type;
        s_info[tid][3] = ps.nbits;              // reuse: prefix length
    }
    __syncthreads();
    if (tid == 0) {
        int ns = 0;
        long long pos = 0;
        s_seg[ns++] = AsmSeg{0, -1, s_hdr_bits, pos}; pos += s_hdr_bits;
        for (int c = 0; c < nch; c++) {
            s_seg[ns++] = AsmSeg{0, c, s_info[c][3], pos}; pos += s_info[c][3];
            const int type = s_info[c][0];
            if (type == FHIP_SUB_VERBATIM) {
                const long long nb = (long long)n * s_info[c][4];
                s_seg[ns++] = AsmSeg{2, c, nb, pos}; pos += nb;
            } else if (type == FHIP_SUB_FIXED || type == FHIP_SUB_LPC) {
                s_seg[ns++] = AsmSeg{1, c, s_info[c][6], pos}; pos += s_info[c][6];
            }
        }
        s_nseg = ns;
        s_total_bits = pos;
    }
    __syncthreads();

    const long long total_bits = s_total_bits;
    const int body_bytes = (int)((total_bits + 7) >> 3);          // before the CRC-16
    const int nwords = (body_bytes + 3) >> 2;
    const int nseg = s_nseg;

    // ---- every output dword from the segments that overlap it ----------------
    for (int w = tid; w < nwords; w += NT) {
        const long long w0 = (long long)w * 32, w1 = w0 + 32;
        uint32_t word = 0;
        for (int q = 0; q < nseg; q++) {
            const AsmSeg sg = s_seg[q];
            const long long a = max(sg.dst, w0), b = min(sg.dst + sg.nbits, w1);
            if (a >= b) continue;
            const int cnt = (int)(b - a);
            const long long sp = a - sg.dst;                      // bit offset inside the segment
            uint32_t bitsv;                                       // cnt bits, right aligned
            if (sg.kind == 2) {
                const int c = sg.ch, obits = s_info[c][4], wasted = s_info[c][5], cm = s_info[c][7];
                const uint32_t omask = (obits >= 32) ? 0xFFFFFFFFu : ((1u << obits) - 1u);
                int i = (int)(sp / obits), offb = (int)(sp % obits), got = 0;
                unsigned long long acc = 0;
                while (got < cnt) {
                    const uint32_t v = (uint32_t)asm_sample(pcm_frame, nch, c, i, cm, wasted) & omask;
                    const int take = min(obits - offb, cnt - got);
                    const uint32_t piece = (take >= 32) ? v : ((v >> (obits - offb - take)) & ((1u << take) - 1u));
                    acc = (acc << take) | piece;
                    got += take; offb = 0; i++;
                }
                bitsv = (uint32_t)acc;
            } else {
                // 64 source bits that start at the dword holding bit sp
                const long long sw = sp >> 5;
                uint32_t hi, lo;
                if (sg.kind == 0) {
                    const uint8_t *src = (sg.ch < 0) ? s_hdr : s_prefix[sg.ch];
                    const int lim = (sg.ch < 0) ? 32 : ASM_PREFIX_BYTES;
                    uint32_t bv[8];
#pragma unroll
                    for (int z = 0; z < 8; z++) {
                        const long long bi = sw * 4 + z;
                        bv[z] = (bi < lim) ? src[bi] : 0u;
                    }
                    hi = (bv[0] << 24) | (bv[1] << 16) | (bv[2] << 8) | bv[3];
                    lo = (bv[4] << 24) | (bv[5] << 16) | (bv[6] << 8) | bv[7];
                } else {
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(
                        rice + ((size_t)f * nch + sg.ch) * (size_t)slot_bytes);
                    hi = __builtin_bswap32(src[sw]);
                    lo = ((sw + 1) * 4 < slot_bytes) ? __builtin_bswap32(src[sw + 1]) : 0u;
                }
                const unsigned long long x = ((unsigned long long)hi << 32) | lo;
                const int sh = (int)(sp & 31);
                bitsv = (uint32_t)((x << sh) >> (64 - cnt));
            }
            word |= bitsv << (32 - (int)(a - w0) - cnt);
        }
        out32[w] = __builtin_bswap32(word);
    }
    __syncthreads();                       // the frame body is in memory (same CU)

    // ---- CRC-16 (crc.c:59-94) in parallel -------------------------------------
    // chunk c covers message bytes [c*L - pad, (c+1)*L - pad): the message is
    // thought of as left-padded with zero bytes, which leaves a CRC with initial
    // value 0 unchanged.
    const int L = (body_bytes + NT - 1) / NT;
    const int pad = L * NT - body_bytes;
    {
        uint16_t c = 0;
        const int b0 = tid * L - pad;
        for (int q = 0; q < L; q++) {
            const int bi = b0 + q;
            if (bi >= 0) {
                // the bytes were stored by other lanes of this workgroup: read past L1
                const uint32_t wv = __hip_atomic_load(&out32[bi >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t byte = (wv >> (8 * (bi & 3))) & 0xFFu;
                c = (uint16_t)((c << 8) ^ s_crc_tab[((c >> 8) ^ byte) & 0xFFu]);
            }
        }
        s_part[tid] = c;
    }
    // m = x^(8L) mod P by square-and-multiply on x^8
    uint16_t m;
    {
        uint16_t base = 0x100, r = 1;
        int ex = L;
        while (ex) { if (ex & 1) r = crc16_mulmod(r, base); base = crc16_mulmod(base, base); ex >>= 1; }
        m = r;
    }
    __syncthreads();
    for (int step = 1; step < NT; step <<= 1) {
        // left node (tid) absorbs its right neighbour: crc(A||B) = crc(A)*x^(8|B|) + crc(B)
        uint16_t v = 0;
        const bool act = (tid % (2 * step)) == 0;
        if (act) v = (uint16_t)(crc16_mulmod(s_part[tid], m) ^ s_part[tid + step]);
        __syncthreads();
        if (act) s_part[tid] = v;
        __syncthreads();
        m = crc16_mulmod(m, m);
    }
    if (tid == 0) {
        const uint16_t crc = s_part[0];
        out[body_bytes] = (uint8_t)(crc >> 8);
        out[body_bytes + 1] = (uint8_t)crc;
        frame_bytes[f] = body_bytes + 2;
    }
}


// ---------------------------------------------------------------------------
// K-vbs  k_vbs_split -- vbs.c:36-83 split_frame_v1
// ---------------------------------------------------------------------------
// One workgroup per block: eight sections of n/8 sample-frames, for each the
// sum over channels of |x[j] - 2x[j-1] + x[j-2]| on the raw interleaved input
// (int32 wrap, then abs), divided by the channel count, plus one; neighbours
// are merged unless the score changes by more than 25 % -- evaluated with the
// reference's int abs() and 32-bit multiply.
__global__ __launch_bounds__(NT)
void k_vbs_split(const int32_t *__restrict__ pcm, int nblocks, int block_size, int nch,
                 int32_t *__restrict__ nframes_out, int32_t *__restrict__ sizes_out)
{
    __shared__ long long s_score[8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x;
    const int n = block_size / 8;
    const int32_t *base = pcm + (size_t)b * block_size * nch;
    for (int sec = wv; sec < 8; sec += 4) {
        const int32_t *sp = base + (size_t)sec * n * nch;
        long long acc = 0;
        // element e of the section = (j, ch) interleaved; rows j >= 2 only
        const int total = (n - 2) * nch;
        for (int e = lane; e < total; e += WAVE) {
            const int idx = e + 2 * nch;
            const uint32_t x0 = (uint32_t)sp[idx], x1 = (uint32_t)sp[idx - nch], x2 = (uint32_t)sp[idx - 2 * nch];
            const int32_t d = (int32_t)(x0 - 2u * x1 + x2);
            acc += (long long)wrap_abs(d);
        }
        acc = (long long)wave_sum_u64((unsigned long long)acc);
        if (lane == 0) s_score[sec] = acc / nch + 1;
    }
    __syncthreads();
    if (tid == 0) {
        int sizes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int nf = 0;
        for (int p = 0; p < 8; p++) {
            bool cut = (p == 0);
            if (p > 0) {
                int32_t diff = (int32_t)(uint32_t)(unsigned long long)(s_score[p - 1] - s_score[p]);
                diff = wrap_abs(diff);
                const int32_t scaled = (int32_t)((uint32_t)diff * 200u);
                cut = ((long long)scaled / s_score[p - 1]) > 50;
            }
            if (cut) nf++;
            sizes[nf - 1] += n;
        }
        nframes_out[b] = nf;
        for (int p = 0; p < 8; p++) sizes_out[(size_t)b * 8 + p] = sizes[p];
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------

hipError_t launch_prepare(hipStream_t st, const fhip_params &p, const int32_t *pcm,
                          int nframes, int n, int32_t *smp, fhip_subframe_info *info, bool decide_only,
                          bool allow_narrow)
{
    const int nch = p.channels;
    if (nframes == 0) return hipSuccess;
    if (nch == 2 && (n & 3) == 0 && n <= 8192) {
        const int est = p.stereo_method == 1 ? 1 : 0;
        const int quads = n >> 2;
        const int nar = (allow_narrow && !decide_only) ? 1 : 0;
#define LAUNCH_PS(M_, A_) hipLaunchKernelGGL((k_prepare_stereo<M_, A_>), dim3(nframes), dim3(NT), 0, st, pcm, smp, info, n, p.bits_per_sample, est, nar)
        if (decide_only) {
            if (quads <= NT) LAUNCH_PS(1, false); else if (quads <= 2 * NT) LAUNCH_PS(2, false); else LAUNCH_PS(4, false);
        } else {
            if (quads <= NT) LAUNCH_PS(1, true);
            else if (quads <= 2 * NT) LAUNCH_PS(2, true);
            else if (quads <= 4 * NT) LAUNCH_PS(4, true);
            else if (quads <= 5 * NT) LAUNCH_PS(5, true);       // 4608
            else LAUNCH_PS(8, true);                            // 8192
        }
#undef LAUNCH_PS
        return hipGetLastError();
    }
    if (decide_only || allow_narrow) return hipErrorInvalidValue;
    if (nch != 2) {
        hipLaunchKernelGGL(k_prepare_multi, dim3(nframes), dim3(NT), 0, st, pcm, smp, info, n, nch,
                           p.bits_per_sample);
        return hipGetLastError();
    }
    const int blocks = nframes;
    const size_t lds = sizeof(int32_t) * (size_t)n * (nch == 2 ? 2 : 1);
    if (blocks == 0) return hipSuccess;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prepare),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (er != hipSuccess) return er;
    hipLaunchKernelGGL(k_prepare, dim3(blocks), dim3(NT), lds, st, pcm, smp, info, n, nch,
                       p.bits_per_sample, p.stereo_method == 1 ? 1 : 0);
    return hipGetLastError();
}

namespace {
// Which K1 kernel serves a batch: a measured time model in ns (MI355X; rounds =
// workgroup waves over the chip, step = one walk step):
//   wt : rounds x (n/2 x max(30, 6.2 NCH) + 3200)      32 subframes per workgroup, whole tiles only
//   ps : rounds x (n/2 x 39 + 1000)                      Gp subframes per wave
//   cur: rounds x (n x 20 + 1000)                        G subframes per wave
struct ac_choice { int kernel; int G, nl2, Gp, lps, ge, ne, no; };   // kernel: 0 cur, 1 ps, 2 wt
ac_choice pick_autocorr(int nsub, int n, int max_order)
{
    ac_choice ch{};
    const int simds = 1024;
    // k_autocorr: lag pairs, both parities in one lane: n positions x 4 fp64 ops
    ch.nl2 = (max_order + 2) / 2;                 // lag pairs {0,1},{2,3},...
    ch.G = WAVE / ch.nl2;
    if (ch.G > AC_GMAX) ch.G = AC_GMAX;
    if (ch.G < 1) ch.G = 1;
    const long waves_cur = (nsub + ch.G - 1) / ch.G;
    // k_autocorr_ps: lag triples, one parity per lane: n/2 steps x 6 fp64 ops
    ch.ne = max_order / 2 + 1; ch.no = (max_order + 1) / 2;
    ch.ge = (ch.ne + 2) / 3;
    const int go = (ch.no + 2) / 3;
    ch.lps = 2 * (ch.ge + go);
    ch.Gp = WAVE / ch.lps;
    if (ch.Gp > PS_GMAX) ch.Gp = PS_GMAX;
    const double t_cur = (double)((waves_cur + simds - 1) / simds) * (n * 20.0 + 1000.0);
    const double t_ps = (ch.Gp >= 1) ? (double)(((nsub + ch.Gp - 1) / ch.Gp + simds - 1) / simds) * (0.5 * n * 39.0 + 1000.0) : 1e30;
    double t_wt = 1e30;
    if ((n % AC_TILE) == 0) {
        const int e0 = (ch.ne + 1) / 2;
        const double per_step = (6.2 * e0 > 30.0) ? 6.2 * e0 : 30.0;
        t_wt = (double)(((nsub + WT_SUB - 1) / WT_SUB + 255) / 256) * (0.5 * n * per_step + 3200.0);
    }
    ch.kernel = (t_wt <= t_ps && t_wt <= t_cur) ? 2 : (t_ps < t_cur) ? 1 : 0;
    if (const char *force = getenv("FHIP_AC_KERNEL")) {       // "cur" / "ps" / "wt": measurements only
        if (force[0] == 'c') ch.kernel = 0;
        if (force[0] == 'p' && ch.Gp >= 1) ch.kernel = 1;
        if (force[0] == 'w' && (n % AC_TILE) == 0) ch.kernel = 2;
    }
    return ch;
}
}  // namespace

// True when K1 will also run K2 (launch_autocorr with lpc outputs): the wave-typed
// kernel and a maximum order the register version of K2 covers.
bool autocorr_does_lpc(int nsub, int n, int max_order)
{
    static const bool off = getenv("FHIP_NO_LPC_TAIL") != nullptr;      // measurements only
    return !off && max_order <= 12 && pick_autocorr(nsub, n, max_order).kernel == 2;
}

bool autocorr_fuses_prepare(const fhip_params &p, int nsub, int n)
{
    // Off by default: measured on configs[1] the decision-only K0 saves 18 us and 134 MB
    // of HBM writes, but the producers' extra work costs K1 11 us on the SIMDs that
    // are its bottleneck, and the step ends up 2 % slower (0.2106 vs 0.2057 ms).
    static const bool on = getenv("FHIP_FUSE") != nullptr && (WT_ROWS0 % 2) == 0 && (WT_ROWS1 % 2) == 0;
    if (!on || p.channels != 2 || (n & 3) != 0 || n > 4096 || (nsub & 1)) return false;
    return pick_autocorr(nsub, n, p.max_prediction_order).kernel == 2;
}

hipError_t launch_autocorr(hipStream_t st, const int32_t *smp, int nsub, int n,
                           int max_order, double *autoc, const int32_t *pcm_fused,
                           int32_t *smp_out, const fhip_subframe_info *info,
                           const autocorr_lpc_out *lpc_out, bool narrow_ok)
{
    if (nsub == 0) return hipSuccess;
    // the window constant is computed on the host exactly as lpc.c:34 does
    const double c = (2.0 / (n - 1.0)) - 1.0;
    const ac_choice ch = pick_autocorr(nsub, n, max_order);
    const int ne = ch.ne, no = ch.no, Gp = ch.Gp, lps = ch.lps, ge = ch.ge, nl2 = ch.nl2;
    int G = ch.G;
    const bool use_wt = ch.kernel == 2, use_ps = ch.kernel == 1;
    if ((pcm_fused || lpc_out || narrow_ok) && !use_wt) return hipErrorInvalidValue;
    if (narrow_ok && (!info || pcm_fused)) return hipErrorInvalidValue;
    const int e0 = (ne + 1) / 2, e1 = ne - e0, o0 = (no + 1) / 2, o1 = no - o0;
    if (use_wt) {
        wt_groups gr;
        gr.l0[0] = 0;          gr.nch[0] = e0;
        gr.l0[1] = 2 * e0;     gr.nch[1] = e1;
        gr.l0[2] = 1;          gr.nch[2] = o0;
        gr.l0[3] = 1 + 2 * o0; gr.nch[3] = o1;
        const int nch = e0;                                    // e0 >= e1, o0, o1
        const int blocks = (nsub + WT_SUB - 1) / WT_SUB;
        const size_t lds = sizeof(double) * (size_t)WT_NBUF * WT_BUF;
        wt_lpc_args la{};
        int lpcmo = 0;
        if (lpc_out) {
            if (max_order > 12 || pcm_fused) return hipErrorInvalidValue;
            la.precision = lpc_out->precision; la.omethod = lpc_out->omethod;
            la.coefs = lpc_out->coefs; la.shift = lpc_out->shift; la.opt_order = lpc_out->opt_order;
            la.fin = lpc_out->fin;
            lpcmo = (max_order <= 8) ? 8 : 12;
        }
        const size_t lds_all = lds + (lpcmo ? sizeof(double) * (size_t)WT_SUB * FHIP_MAX_LAGS : 0);
#define LAUNCH_WT3(N_, F_, L_)                                                               \
    do {                                                                                     \
        hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_autocorr_wt<N_, F_, L_>), \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_all); \
        if (er != hipSuccess) return er;                                                     \
        hipLaunchKernelGGL((k_autocorr_wt<N_, F_, L_>), dim3(blocks), dim3(8 * WAVE), lds_all, st, smp, \
                           autoc, nsub, n, max_order, gr, c, pcm_fused, smp_out, info, la, narrow_ok ? 1 : 0); \
    } while (0)
#define LAUNCH_WT(N_)                                                                        \
    case N_:                                                                                 \
        if (pcm_fused) LAUNCH_WT3(N_, true, 0); else LAUNCH_WT3(N_, false, 0);               \
        break;
        if (lpcmo == 8) {                      // max_order <= 8: NCH <= 3
            switch (nch) {
            case 1: LAUNCH_WT3(1, false, 8); break;
            case 2: LAUNCH_WT3(2, false, 8); break;
            case 3: LAUNCH_WT3(3, false, 8); break;
            default: return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
        if (lpcmo == 12) {                     // max_order 9..12: NCH 3 or 4
            switch (nch) {
            case 3: LAUNCH_WT3(3, false, 12); break;
            case 4: LAUNCH_WT3(4, false, 12); break;
            default: return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
        switch (nch) {
            LAUNCH_WT(1) LAUNCH_WT(2) LAUNCH_WT(3) LAUNCH_WT(4) LAUNCH_WT(5)
            LAUNCH_WT(6) LAUNCH_WT(7) LAUNCH_WT(8) LAUNCH_WT(9)
        default: return hipErrorInvalidValue;
        }
#undef LAUNCH_WT
#undef LAUNCH_WT3
        return hipGetLastError();
    }
    if (use_ps) {
        const int per_block = Gp * AC_WAVES;
        const int blocks = (nsub + per_block - 1) / per_block;
        hipLaunchKernelGGL(k_autocorr_ps, dim3(blocks), dim3(AC_WAVES * WAVE), 0, st, smp, autoc,
                           nsub, n, max_order, Gp, lps, ge, c);
        return hipGetLastError();
    }
    // spread over all CUs when the batch is small: fewer subframes per wave
    // cost nothing (a wave's time is its chain length, not its lane count)
    while (G > 1 && (nsub + G * AC_WAVES - 1) / (G * AC_WAVES) < 256) G--;
    if (const char *dbg = getenv("FHIP_AC_G")) { int v = atoi(dbg); if (v >= 1 && v <= AC_GMAX && v * nl2 <= WAVE) G = v; }
    const int per_block = G * AC_WAVES;
    const int blocks = (nsub + per_block - 1) / per_block;
    hipLaunchKernelGGL(k_autocorr, dim3(blocks), dim3(AC_WAVES * WAVE), 0, st, smp, autoc,
                       nsub, n, max_order, G, nl2, c);
    return hipGetLastError();
}

hipError_t launch_lpc(hipStream_t st, const double *autoc, int nsub, int max_order,
                      int precision, int omethod, int32_t *coefs, int32_t *shift,
                      int32_t *opt_order, int32_t *fin)
{
    if (nsub == 0) return hipSuccess;
    const int blocks = (nsub + LPC_NT - 1) / LPC_NT;
    if (max_order <= 8)
        hipLaunchKernelGGL(k_lpc_reg<8>, dim3(blocks), dim3(LPC_NT), 0, st, autoc, nsub, max_order,
                           precision, omethod, coefs, shift, opt_order, fin);
    else if (max_order <= 12)
        hipLaunchKernelGGL(k_lpc_reg<12>, dim3(blocks), dim3(LPC_NT), 0, st, autoc, nsub, max_order,
                           precision, omethod, coefs, shift, opt_order, fin);
    else
        hipLaunchKernelGGL(k_lpc, dim3(blocks), dim3(LPC_NT), 0, st, autoc, nsub, max_order,
                           precision, omethod, coefs, shift, opt_order, fin);
    return hipGetLastError();
}

size_t encode_lds_bytes(int n)
{
    if (n < 1 || n > FHIP_MAX_BLOCK) return 0;
    size_t off[10];
    return enc_lds_layout(n, off);
}

// Fast-path geometry for a block size: C samples per thread, T threads,
// n = C*T, T a power of two >= 64.
static bool fast_geometry(const fhip_params &p, int n, int *C, int *T)
{
    if (n < 192 || n > FHIP_MAX_BLOCK) return false;
    int odd = n, lg = 0;
    while ((odd & 1) == 0) { odd >>= 1; lg++; }
    int c, t;
    if (odd == 1) {                       // 256 .. 16384
        if (n >= 4096) { c = 16; t = n / 16; }
        else if (n >= 2048) { c = 8; t = 256; }
        else if (n >= 1024) { c = 4; t = 256; }
        else if (n == 512) { c = 8; t = 64; }
        else if (n == 256) { c = 4; t = 64; }
        else return false;
    } else if (odd == 9) {                // 576, 1152, 2304, 4608, 9216
        // 256 threads where the block allows (measured at 4608: (18,256) 84 us, (9,512) 100)
        c = 9; t = 1 << lg;
        if (t >= 512) { c = 18; t >>= 1; }
    } else if (odd == 3) {                // 192, 384, 768, 1536, ...
        c = 3; t = 1 << lg;
        if (t > 1024) { c = 0; }
    } else {
        return false;
    }
    if (const char *g = getenv("FHIP_K3_GEOM")) {     // measurements only: "C,T" of an instantiated pair
        int gc = 0, gt = 0;
        if (sscanf(g, "%d,%d", &gc, &gt) == 2 && gc * gt == n) { c = gc; t = gt; }
    }
    if (c == 0 || t < 64 || t > 1024) return false;
    // every partition at least one thread wide at the finest level that can occur
    if ((n >> p.max_partition_order) < c && odd == 1) return false;
    *C = c; *T = t;
    return true;
}

// True when every kernel of the pipeline that touches the sample rows understands
// 16-bit rows for such a batch: the register K0 for stereo, the wave-typed K1 (or
// no K1 at all) and a K3 fast-path geometry with runs of 8 or 16 samples.
bool narrow_rows_ok(const fhip_params &p, int nsub, int n, bool lpc_path)
{
    static const bool off = getenv("FHIP_NO_NARROW") != nullptr;        // measurements only
    if (off || p.channels != 2 || (n & 3) != 0 || n > 8192) return false;
    int fc = 0, ft = 0;
    if (!fast_geometry(p, n, &fc, &ft) || (fc % 8) != 0) return false;
    if (lpc_path && pick_autocorr(nsub, n, p.max_prediction_order).kernel != 2) return false;
    return true;
}

hipError_t launch_encode(hipStream_t st, const fhip_params &p, const int32_t *smp,
                         int nsub, int n, const int32_t *coefs, const int32_t *shift,
                         const int32_t *opt_order, const int32_t *fin,
                         fhip_subframe_info *info,
                         int32_t *residual, uint8_t *bits, int64_t slot_bytes,
                         int raw_order, int raw_lpc, bool narrow_ok)
{
    if (nsub == 0) return hipSuccess;
    int fc = 0, ft = 0;
    static const bool force_generic = getenv("FHIP_K3_GENERIC") != nullptr;    // measurements only
    if (raw_order < 0 && !force_generic && fast_geometry(p, n, &fc, &ft)) {
        size_t off[12];
        size_t lds = 0;
#define LAUNCH_FAST2(CC, TT, MM)                                                             \
    do {                                                                                     \
        lds = fast_lds_layout(n, (size_t)SmpImg<CC, TT>::SIZE, off);                         \
        hipError_t er = hipFuncSetAttribute(                                                 \
            reinterpret_cast<const void *>(&k_encode_pow2<CC, TT, MM>),                      \
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        if (er != hipSuccess) return er;                                                     \
        hipLaunchKernelGGL((k_encode_pow2<CC, TT, MM>), dim3(nsub), dim3(TT), lds, st, p, n, \
                           nsub, smp, coefs, shift, opt_order, fin, info, residual, bits,    \
                           (long long)slot_bytes, narrow_ok ? 1 : 0);                        \
    } while (0)
        // one quantised row known up front (MAX / EST): the lean instance
        const bool single_row = (p.prediction_type == 2) && (n > p.max_prediction_order) && (p.order_method <= 1);
        const bool fixed_only = (p.prediction_type == 1) && n >= 5;
#define LAUNCH_FAST(CC, TT)                                                                  \
    do {                                                                                     \
        if (single_row) LAUNCH_FAST2(CC, TT, 0);                                             \
        else if (fixed_only) LAUNCH_FAST2(CC, TT, 1);                                        \
        else LAUNCH_FAST2(CC, TT, 2);                                                        \
    } while (0)
        const int key = fc * 10000 + ft;
        switch (key) {
        case 160256: LAUNCH_FAST(16, 256); break;
        case 160512: LAUNCH_FAST(16, 512); break;
        case 161024: LAUNCH_FAST(16, 1024); break;
        case 80256: LAUNCH_FAST(8, 256); break;
        case 40256: LAUNCH_FAST(4, 256); break;
        case 80064: LAUNCH_FAST(8, 64); break;
        case 40064: LAUNCH_FAST(4, 64); break;
        case 90064: LAUNCH_FAST(9, 64); break;
        case 180256: LAUNCH_FAST(18, 256); break;
        case 180512: LAUNCH_FAST(18, 512); break;
        case 90128: LAUNCH_FAST(9, 128); break;
        case 90256: LAUNCH_FAST(9, 256); break;
        case 30064: LAUNCH_FAST(3, 64); break;
        case 30128: LAUNCH_FAST(3, 128); break;
        case 30256: LAUNCH_FAST(3, 256); break;
        case 30512: LAUNCH_FAST(3, 512); break;
        case 31024: LAUNCH_FAST(3, 1024); break;
        default: return hipErrorInvalidValue;
        }
#undef LAUNCH_FAST
#undef LAUNCH_FAST2
        return hipGetLastError();
    }
    const size_t lds = encode_lds_bytes(n);
    if (lds == 0) return hipErrorInvalidValue;
    const int chunk = (n + NT - 1) / NT;
#define LAUNCH_ENC(CC)                                                                       \
    do {                                                                                     \
        hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_encode<CC>),   \
                                            hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                            (int)lds);                                       \
        if (er != hipSuccess) return er;                                                     \
        hipLaunchKernelGGL(k_encode<CC>, dim3(nsub), dim3(NT), lds, st, p, n, smp, coefs,    \
                           shift, opt_order, info, residual, bits, (long long)slot_bytes,    \
                           raw_order, raw_lpc);                                              \
    } while (0)
    if (chunk <= 16) LAUNCH_ENC(16);
    else if (chunk <= 32) LAUNCH_ENC(32);
    else LAUNCH_ENC(64);
#undef LAUNCH_ENC
    return hipGetLastError();
}

hipError_t launch_vbs_split(hipStream_t st, const int32_t *pcm, int nblocks, int block_size,
                            int nch, int32_t *nframes_out, int32_t *sizes_out)
{
    if (nblocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_vbs_split, dim3(nblocks), dim3(NT), 0, st, pcm, nblocks, block_size, nch,
                       nframes_out, sizes_out);
    return hipGetLastError();
}

hipError_t launch_assemble(hipStream_t st, const fhip_params &p, const int32_t *pcm, int nframes,
                           int n, const fhip_subframe_info *info, const uint8_t *rice,
                           int64_t slot_bytes, uint8_t *frames, int64_t frame_stride,
                           int32_t *frame_bytes, uint32_t number_base, uint32_t number_step,
                           const uint32_t *numbers)
{
    if (nframes == 0) return hipSuccess;
    // sample-rate / bit-depth codes of flake_encode_init() (encode.c:400-438)
    static const int sr_table[16] = {0, 0, 0, 0, 8000, 16000, 22050, 24000, 32000, 44100, 48000,
                                     96000, 0, 0, 0, 0};
    static const int bd_table[8] = {0, 8, 12, 0, 16, 20, 24, 0};
    int sr0 = 0, sr1 = 0, bpsc = 0;
    for (int i = 4; i < 12; i++) if (p.sample_rate == sr_table[i]) { sr0 = i; break; }
    if (!sr0) {
        const int sr = p.sample_rate;
        if (sr % 1000 == 0 && sr <= 255000) { sr0 = 12; sr1 = sr / 1000; }
        else if (sr % 10 == 0 && sr <= 655350) { sr0 = 14; sr1 = sr / 10; }
        else if (sr < 65535) { sr0 = 13; sr1 = sr; }
    }
    for (int i = 1; i < 8; i++) if (p.bits_per_sample == bd_table[i]) { bpsc = i; break; }
    const int bps = p.bits_per_sample;
    const int vsize = (p.channels == 2) ? 16 + ((n * (bps + bps + 1) + 7) >> 3)
                                        : 16 + ((n * p.channels * bps + 7) >> 3);
    hipLaunchKernelGGL(k_assemble, dim3(nframes), dim3(NT), 0, st, p, n, pcm, info, rice,
                       (long long)slot_bytes, frames, (long long)frame_stride, frame_bytes,
                       number_base, number_step, numbers, sr0, sr1, bpsc, vsize);
    return hipGetLastError();
}

}  // namespace fhip
