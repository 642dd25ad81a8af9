// k0_prepare.hip -- K0: copy_samples + stereo estimate / decorrelation + wasted bits
// (encode.c:541-694).  pcm [nframes][n][ch] -> smp [nframes][ch][n].
#include "device_util.h"

namespace fhip {
namespace {

// ---------------------------------------------------------------------------
// K0  k_prepare
// ---------------------------------------------------------------------------
// Stereo frames that do not qualify for the register path (n > 8192 or n % 4):
// one workgroup per frame, both channels resident in LDS (int32[2n]).  Other
// channel counts go to k_prepare_multi.
// RESIDENT = false: frames too long for LDS (n > 20 k, up to FLAC's 65535) are
// streamed from global memory in each of the three passes instead (L2 serves the
// re-reads: a frame is 512 KB at most).
template <bool RESIDENT>
__global__ __launch_bounds__(NT)
void k_prepare(const int32_t *__restrict__ pcm, int32_t *__restrict__ smp,
               fhip_subframe_info *__restrict__ info, int n, int nch, int bps, int estimate,
               const long long *__restrict__ frame_src, const int32_t *__restrict__ dev_frames)
{
    extern __shared__ int32_t lds_i32[];
    if (dev_frames && (int)blockIdx.x >= dev_count(dev_frames, 0)) return;
    __shared__ unsigned long long s_sum[4][4];
    __shared__ uint32_t s_or[4][2];
    __shared__ int s_mode;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    if (nch == 2) {
        const int f = blockIdx.x;
        const int32_t *src = pcm + (frame_src ? (size_t)frame_src[f] : (size_t)f * n * 2);
        int32_t *L = lds_i32, *R = lds_i32 + n;
        const int2 *src2 = reinterpret_cast<const int2 *>(src);
        if (RESIDENT) {
            for (int i = tid; i < n; i += NT) {
                int2 v = src2[i];
                L[i] = v.x;
                R[i] = v.y;
            }
            __syncthreads();
        }
        auto left = [&](int i) { return RESIDENT ? L[i] : src2[i].x; };
        auto right = [&](int i) { return RESIDENT ? R[i] : src2[i].y; };
        // encode.c:668-693 for one sample-frame of the input
        auto decorrelate = [&](int32_t &a, int32_t &b, int md) {
            if (md == FHIP_CH_MID_SIDE) {
                const int32_t mid = (int32_t)((uint32_t)a + (uint32_t)b) >> 1;
                const int32_t sd = (int32_t)((uint32_t)a - (uint32_t)b);
                a = mid; b = sd;
            } else if (md == FHIP_CH_LEFT_SIDE) {
                b = (int32_t)((uint32_t)a - (uint32_t)b);
            } else if (md == FHIP_CH_RIGHT_SIDE) {
                a = (int32_t)((uint32_t)a - (uint32_t)b);
            }
        };

        int mode = FHIP_CH_LEFT_RIGHT;
        if (estimate && n > 32) {
            // encode.c:598-643 calc_decorr_scores
            unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int i = tid + 2; i < n; i += NT) {
                int32_t lt = (int32_t)((uint32_t)left(i) - 2u * (uint32_t)left(i - 1) + (uint32_t)left(i - 2));
                int32_t rt = (int32_t)((uint32_t)right(i) - 2u * (uint32_t)right(i - 1) + (uint32_t)right(i - 2));
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
            int32_t a = left(i), b = right(i);
            decorrelate(a, b, mode);
            if (RESIDENT) { L[i] = a; R[i] = b; }
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
            int32_t a = left(i), b = right(i);
            if (!RESIDENT) decorrelate(a, b, mode);          // resident rows were decorrelated in place
            dst[i] = a >> wasted[0];
            dst[n + i] = b >> wasted[1];
        }
        if (tid < 2) {
            fhip_subframe_info *o = &info[(size_t)f * 2 + tid];
            o->obits = obits[tid];
            o->wasted = wasted[tid];
            o->ch_mode = mode;
            o->reserved = 0;               // no 16-bit rows, magnitude not known (see k_prepare_stereo)
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
                     fhip_subframe_info *__restrict__ info, int n, int nch, int bps,
                     const long long *__restrict__ frame_src, const int32_t *__restrict__ dev_frames)
{
    __shared__ int32_t s_tile[NT * (FHIP_MAX_CH + 1)];
    __shared__ uint32_t s_orr[4][FHIP_MAX_CH];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int f = blockIdx.x;
    if (dev_frames && f >= dev_count(dev_frames, 0)) return;
    const int32_t *src = pcm + (frame_src ? (size_t)frame_src[f] : (size_t)f * n * nch);
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
                oi->reserved = 0;
            }
        }
    }
}

// K0 for 1, 3 .. 8 channels with the frame in REGISTERS (round 2): k_prepare_multi above reads
// the interleaved block twice (the OR pass, then the shift-and-write pass) through an LDS
// transpose tile; here thread t owns the sample-frame quads 4(t + RT*m) .. +3 -- 4*NCH
// consecutive ints, i.e. NCH 16-byte loads per quad -- keeps them through the OR reduction
// (remove_wasted_bits, encode.c:558-593: no decorrelation without exactly two channels,
// encode.c:660-663) and the shift, and stores four consecutive samples of one channel at a time
// (16 bytes).  One read of the PCM, no LDS beyond the cross-wave OR.  n % 4 == 0 and
// n <= 4 * RT * M.
constexpr int RT = 1024;       // threads per frame
template <int NCH, int M>
__global__ __launch_bounds__(RT)
void k_prepare_multi_reg(const int32_t *__restrict__ pcm, int32_t *__restrict__ smp,
                         fhip_subframe_info *__restrict__ info, int n, int bps,
                         const long long *__restrict__ frame_src, const int32_t *__restrict__ dev_frames)
{
    __shared__ uint32_t s_orr[RT / 64][8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int f = blockIdx.x;
    if (dev_frames && f >= dev_count(dev_frames, 0)) return;
    const int quads = n >> 2;
    const int32_t *src = pcm + (frame_src ? (size_t)frame_src[f] : (size_t)f * n * NCH);
    int32_t v[M][4 * NCH];                 // v[m][4*NCH]: element e = sample-frame (e / NCH), channel (e % NCH)
    uint32_t orv[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) orv[c] = 0;
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int q = tid + RT * m;
        if (q < quads) {
            const int4 *p4 = reinterpret_cast<const int4 *>(src + (size_t)q * 4 * NCH);   // 16 * NCH bytes: aligned
#pragma unroll
            for (int k = 0; k < NCH; k++) {
                const int4 t4 = p4[k];
                v[m][4 * k] = t4.x; v[m][4 * k + 1] = t4.y; v[m][4 * k + 2] = t4.z; v[m][4 * k + 3] = t4.w;
            }
#pragma unroll
            for (int e = 0; e < 4 * NCH; e++) orv[e % NCH] |= (uint32_t)v[m][e];
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const uint32_t o = wave_or_u32(orv[c]);
        if (lane == 0) s_orr[wv][c] = o;
    }
    __syncthreads();
    int wasted[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        uint32_t o = 0;
#pragma unroll
        for (int w = 0; w < RT / 64; w++) o |= s_orr[w][c];
        int w = o ? min(__ffs((int)o) - 1, bps - 1) : bps - 1;
        if (w == bps - 1) w = 0;                                   // encode.c:583-584
        wasted[c] = w;
    }
    if (tid < NCH) {
        fhip_subframe_info *oi = &info[(size_t)f * NCH + tid];
        int w = 0;
#pragma unroll
        for (int c = 0; c < NCH; c++) if (c == tid) w = wasted[c];
        oi->obits = bps - w;
        oi->wasted = w;
        oi->ch_mode = FHIP_CH_NOT_STEREO;
        oi->reserved = 0;
    }
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int q = tid + RT * m;
        if (q < quads) {
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int4 o4 = make_int4(v[m][c] >> wasted[c], v[m][NCH + c] >> wasted[c],
                                          v[m][2 * NCH + c] >> wasted[c], v[m][3 * NCH + c] >> wasted[c]);
                *reinterpret_cast<int4 *>(smp + ((size_t)f * NCH + c) * n + 4 * q) = o4;
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
// WPF = waves per frame (1, 2 or 4): short blocks put 4 / WPF frames in a workgroup so
// that its 256 threads have quads to work on (n = 256: one wave per frame; at four
// waves per frame three quarters of the threads idled and K0 took 5.7x its time per
// sample).  Thread t of a frame owns quads t + 64 WPF m.
template <int M, int WPF, bool APPLY>
// allow_narrow: a channel whose samples (after the shift) all fit 16 bits is stored
// as int16[n] at the start of its row (info.reserved = 1 tells K1's producers and
// K3's staging; K3 resets the field) -- half the bytes written here and read there.
// blk: the workgroup's index within its batch (or within its bin of a ragged batch).
__device__ __forceinline__
void prepare_stereo_body(const int32_t *__restrict__ pcm, int32_t *__restrict__ smp,
                         fhip_subframe_info *__restrict__ info, int n, int bps, int estimate,
                         int allow_narrow, int nframes, const long long *__restrict__ frame_src, int blk)
{
    __shared__ unsigned long long s_sum[4][4];
    __shared__ uint32_t s_or[4][4];
    if (blk * (4 / WPF) >= nframes) return;                      // (a ragged batch's grid is its bin's capacity)

    constexpr int TF = WAVE * WPF;                       // threads per frame
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wb = (wv / WPF) * WPF;                     // first wave of this thread's frame
    const int tid = threadIdx.x - wb * WAVE;             // thread index inside the frame
    const int fq = blk * (4 / WPF) + wv / WPF;
    const bool fvalid = fq < nframes;                    // a partial last workgroup computes a copy of the last frame
    const int f = min(fq, nframes - 1);
    const int4 *src = reinterpret_cast<const int4 *>(pcm + (frame_src ? (size_t)frame_src[f] : (size_t)f * n * 2));
    const int nquads = n >> 2;

    int32_t L[M][4], R[M][4];
    int4 prev[M];
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int g = tid + TF * m;
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
                const int g = tid + TF * m;
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
            const int g = tid + TF * m;
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
        // every wave prices the four sums itself (lane & 3 picks the sum, four lanes hold the
        // results): no second barrier, no four threads the workgroup waits for
        unsigned long long c0, c1, c2, c3;
        {
            unsigned long long sm = 0;
#pragma unroll
            for (int w = 0; w < WPF; w++) sm += s_sum[wb + w][lane & 3];
            uint32_t dummy;
            const int k = rice_k_fast(2 * sm, n, &dummy);     // closed form of the 31-step scan
            const unsigned long long cnt = rice_count64(2 * sm, n, k);      // no 32-bit truncation (encode.c:620)
            const int lo = (int)(uint32_t)cnt, hi = (int)(uint32_t)(cnt >> 32);
#define LANE64(q_) (((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(hi, q_) << 32) | (uint32_t)__builtin_amdgcn_readlane(lo, q_))
            c0 = LANE64(0); c1 = LANE64(1); c2 = LANE64(2); c3 = LANE64(3);
#undef LANE64
        }
        {
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
        const bool on = (tid + TF * m) < nquads;
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
        uint32_t o = 0;
#pragma unroll
        for (int w = 0; w < WPF; w++) o |= s_or[wb + w][c];
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
        uint32_t mgo = 0;
#pragma unroll
        for (int w = 0; w < WPF; w++) mgo |= s_or[wb + w][2 + c];
        const uint32_t mg = mgo >> wasted[c];
        narrow[c] = allow_narrow && (mg < 32768u);
        magbits[c] = 32 - __clz((int)mg);
    }

    int4 *dl = reinterpret_cast<int4 *>(smp + (size_t)f * 2 * n);
    int4 *dr = reinterpret_cast<int4 *>(smp + (size_t)f * 2 * n + n);
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int g = tid + TF * m;
        if (APPLY && g < nquads && fvalid) {
            const int4 vl = make_int4(L[m][0] >> wasted[0], L[m][1] >> wasted[0], L[m][2] >> wasted[0], L[m][3] >> wasted[0]);
            const int4 vr = make_int4(R[m][0] >> wasted[1], R[m][1] >> wasted[1], R[m][2] >> wasted[1], R[m][3] >> wasted[1]);
            if (narrow[0]) reinterpret_cast<int2 *>(dl)[g] = make_int2((vl.x & 0xFFFF) | (vl.y << 16), (vl.z & 0xFFFF) | (vl.w << 16));
            else dl[g] = vl;
            if (narrow[1]) reinterpret_cast<int2 *>(dr)[g] = make_int2((vr.x & 0xFFFF) | (vr.y << 16), (vr.z & 0xFFFF) | (vr.w << 16));
            else dr[g] = vr;
        }
    }
    if (tid < 2 && fvalid) {
        fhip_subframe_info *o = &info[(size_t)f * 2 + tid];
        o->obits = obits[tid];
        o->wasted = wasted[tid];
        o->ch_mode = mode;
        // low byte 1..16: narrow row, |x| < 2^(low byte - 1); bits 8..15: 1 + magbits whatever the row's
        // width (the order-search kernel's packed FIR asks for it on int32 rows too)
        o->reserved = (narrow[tid] ? 1 + magbits[tid] : 0) | ((1 + magbits[tid]) << 8);
    }
}

// The same for blocks of more than 4096 samples inside a ragged batch, in two passes over the block (the second finds it
// in L2): five to eight quads per thread held through the decision are 120-181 registers -- two waves per SIMD for every
// bin of the launch (k_prepare_stereo_bins) and 355 us for the blocks of a level-12 batch whose bytes take 80.  Pass 1
// takes the sums of the estimate and, for all four candidates of the two output channels (left, right, mid, side), the OR
// of the samples and of their magnitude bits -- none of them depends on the decision --; pass 2 applies the decision.
__device__ __forceinline__
void prepare_stereo_body_2p(const int32_t *__restrict__ pcm, int32_t *__restrict__ smp,
                            fhip_subframe_info *__restrict__ info, int n, int bps, int estimate,
                            int allow_narrow, int nframes, const long long *__restrict__ frame_src, int blk)
{
    __shared__ unsigned long long s_sum2[4][4];
    __shared__ uint32_t s_or2[4][8];
    if (blk >= nframes) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int f = blk;
    const int4 *src = reinterpret_cast<const int4 *>(pcm + (frame_src ? (size_t)frame_src[f] : (size_t)f * n * 2));
    const int nquads = n >> 2;
    const bool est = estimate && n > 32;

    unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    uint32_t o[4] = {0, 0, 0, 0}, mg[4] = {0, 0, 0, 0};          // left, right, mid, side
#pragma unroll 2
    for (int g = tid; g < nquads; g += NT) {
        const int4 a = src[2 * g], b = src[2 * g + 1];           // (l0 r0 l1 r1) (l2 r2 l3 r3)
        const int4 pv = src[max(2 * g - 1, 0)];                  // (l-2 r-2 l-1 r-1)
        const uint32_t l[4] = {(uint32_t)a.x, (uint32_t)a.z, (uint32_t)b.x, (uint32_t)b.z};
        const uint32_t r[4] = {(uint32_t)a.y, (uint32_t)a.w, (uint32_t)b.y, (uint32_t)b.w};
        uint32_t l2 = (uint32_t)pv.x, r2 = (uint32_t)pv.y, l1 = (uint32_t)pv.z, r1 = (uint32_t)pv.w;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            // encode.c:598-643 calc_decorr_scores (no history for the block's first two samples, encode.c:607)
            const int32_t lt = (int32_t)(l[q] - 2u * l1 + l2);
            const int32_t rt = (int32_t)(r[q] - 2u * r1 + r2);
            const int32_t mm = (int32_t)((uint32_t)lt + (uint32_t)rt) >> 1;
            const int32_t ss = (int32_t)((uint32_t)lt - (uint32_t)rt);
            const bool on = est && (4 * g + q >= 2);
            a0 += on ? (unsigned long long)(long long)wrap_abs(lt) : 0ull;
            a1 += on ? (unsigned long long)(long long)wrap_abs(rt) : 0ull;
            a2 += on ? (unsigned long long)(long long)wrap_abs(mm) : 0ull;
            a3 += on ? (unsigned long long)(long long)wrap_abs(ss) : 0ull;
            l2 = l1; r2 = r1; l1 = l[q]; r1 = r[q];
            // encode.c:668-693: what either output channel can be
            const int32_t c4[4] = {(int32_t)l[q], (int32_t)r[q], (int32_t)(l[q] + r[q]) >> 1, (int32_t)(l[q] - r[q])};
#pragma unroll
            for (int z = 0; z < 4; z++) { o[z] |= (uint32_t)c4[z]; mg[z] |= (uint32_t)(c4[z] ^ (c4[z] >> 31)); }
        }
    }
    if (est) {
        a0 = wave_sum_u64(a0); a1 = wave_sum_u64(a1); a2 = wave_sum_u64(a2); a3 = wave_sum_u64(a3);
        if (lane == 0) { s_sum2[wv][0] = a0; s_sum2[wv][1] = a1; s_sum2[wv][2] = a2; s_sum2[wv][3] = a3; }
    }
#pragma unroll
    for (int z = 0; z < 4; z++) {
        const uint32_t x = wave_or_u32(o[z]), y = wave_or_u32(mg[z]);
        if (lane == 0) { s_or2[wv][z] = x; s_or2[wv][4 + z] = y; }
    }
    __syncthreads();
    int mode = FHIP_CH_LEFT_RIGHT;
    if (est) {
        // every wave prices the four sums itself (prepare_stereo_body)
        const unsigned long long sm = s_sum2[0][lane & 3] + s_sum2[1][lane & 3] + s_sum2[2][lane & 3] + s_sum2[3][lane & 3];
        uint32_t dummy;
        const int k = rice_k_fast(2 * sm, n, &dummy);
        const unsigned long long cnt = rice_count64(2 * sm, n, k);      // no 32-bit truncation (encode.c:620)
        const int lo = (int)(uint32_t)cnt, hi = (int)(uint32_t)(cnt >> 32);
#define LANE64(q_) (((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(hi, q_) << 32) | (uint32_t)__builtin_amdgcn_readlane(lo, q_))
        const unsigned long long c0 = LANE64(0), c1 = LANE64(1), c2 = LANE64(2), c3 = LANE64(3);
#undef LANE64
        const unsigned long long sc[4] = {c0 + c1, c0 + c3, c1 + c3, c2 + c3};
        int best = 0;
#pragma unroll
        for (int q = 1; q < 4; q++) if (sc[q] < sc[best]) best = q;
        mode = (best == 0) ? FHIP_CH_LEFT_RIGHT : (best == 1) ? FHIP_CH_LEFT_SIDE
             : (best == 2) ? FHIP_CH_RIGHT_SIDE : FHIP_CH_MID_SIDE;
    }
    // which candidates the two channels are: (left, right), (left, side), (side, right), (mid, side)
    const int cand[2] = {(mode == FHIP_CH_MID_SIDE) ? 2 : (mode == FHIP_CH_RIGHT_SIDE) ? 3 : 0,
                         (mode == FHIP_CH_LEFT_RIGHT || mode == FHIP_CH_RIGHT_SIDE) ? 1 : 3};
    int wasted[2], obits[2], magbits[2];
    bool narrow[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        // encode.c:558-593
        uint32_t ov = 0, mv = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
#pragma unroll
            for (int z = 0; z < 4; z++) if (z == cand[c]) { ov |= s_or2[w][z]; mv |= s_or2[w][4 + z]; }
        }
        int w = ov ? min(__ffs((int)ov) - 1, bps - 1) : bps - 1;
        if (w == bps - 1) w = 0;
        wasted[c] = w;
        obits[c] = bps - w;
        const uint32_t m = mv >> w;
        narrow[c] = allow_narrow && (m < 32768u);
        magbits[c] = 32 - __clz((int)m);
    }
    if (mode == FHIP_CH_MID_SIDE || mode == FHIP_CH_LEFT_SIDE) obits[1]++;
    if (mode == FHIP_CH_RIGHT_SIDE) obits[0]++;

    int4 *dl = reinterpret_cast<int4 *>(smp + (size_t)f * 2 * n);
    int4 *dr = reinterpret_cast<int4 *>(smp + (size_t)f * 2 * n + n);
#pragma unroll 2
    for (int g = tid; g < nquads; g += NT) {
        const int4 a = src[2 * g], b = src[2 * g + 1];
        int32_t L[4] = {a.x, a.z, b.x, b.z}, R[4] = {a.y, a.w, b.y, b.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int32_t l = L[q], r = R[q];
            const int32_t mid = (int32_t)((uint32_t)l + (uint32_t)r) >> 1, sd = (int32_t)((uint32_t)l - (uint32_t)r);
            L[q] = ((mode == FHIP_CH_MID_SIDE) ? mid : (mode == FHIP_CH_RIGHT_SIDE) ? sd : l) >> wasted[0];
            R[q] = ((mode == FHIP_CH_LEFT_RIGHT || mode == FHIP_CH_RIGHT_SIDE) ? r : sd) >> wasted[1];
        }
        if (narrow[0]) reinterpret_cast<int2 *>(dl)[g] = make_int2((L[0] & 0xFFFF) | (L[1] << 16), (L[2] & 0xFFFF) | (L[3] << 16));
        else dl[g] = make_int4(L[0], L[1], L[2], L[3]);
        if (narrow[1]) reinterpret_cast<int2 *>(dr)[g] = make_int2((R[0] & 0xFFFF) | (R[1] << 16), (R[2] & 0xFFFF) | (R[3] << 16));
        else dr[g] = make_int4(R[0], R[1], R[2], R[3]);
    }
    if (tid < 2) {
        fhip_subframe_info *oi = &info[(size_t)f * 2 + tid];
        oi->obits = obits[tid];
        oi->wasted = wasted[tid];
        oi->ch_mode = mode;
        oi->reserved = (narrow[tid] ? 1 + magbits[tid] : 0) | ((1 + magbits[tid]) << 8);      // (prepare_stereo_body)
    }
}

template <int M, int WPF, bool APPLY>
__global__ __launch_bounds__(NT)
void k_prepare_stereo(const int32_t *__restrict__ pcm, int32_t *__restrict__ smp,
                      fhip_subframe_info *__restrict__ info, int n, int bps, int estimate,
                      int allow_narrow, int nframes, const long long *__restrict__ frame_src,
                      const int32_t *__restrict__ dev_frames)
{
    prepare_stereo_body<M, WPF, APPLY>(pcm, smp, info, n, bps, estimate, allow_narrow,
                                       dev_count(dev_frames, nframes), frame_src, (int)blockIdx.x);
}

// quads per thread M and waves per frame WPF of a stereo block of n samples (n % 4 == 0, n <= 8192): the
// fullest threads win for short blocks.  One rule for the launcher and for the kernel of a ragged batch.
#define FHIP_STEREO_GEOM(QUADS_, DO_)                                                       \
    do {                                                                                    \
        if ((QUADS_) <= 64) DO_(1, 1);              /* n <= 256: one wave per frame */      \
        else if ((QUADS_) <= 128) DO_(1, 2);        /* 512 */                               \
        else if ((QUADS_) <= 192) DO_(3, 1);        /* 576, 768 */                          \
        else if ((QUADS_) <= NT) DO_(1, 4);         /* 1024 */                              \
        else if ((QUADS_) <= 320) DO_(5, 1);        /* 1152 */                              \
        else if ((QUADS_) <= 2 * NT) DO_(2, 4);                                             \
        else if ((QUADS_) <= 3 * NT) DO_(3, 4);     /* 2304, 3072 */                        \
        else if ((QUADS_) <= 4 * NT) DO_(4, 4);                                             \
        else if ((QUADS_) <= 5 * NT) DO_(5, 4);     /* 4608 */                              \
        else DO_(8, 4);                             /* 8192 */                              \
    } while (0)

// K0 for every bin of a ragged (variable-block-size) stereo batch in ONE launch (kernels.h: MultiBin;
// unit0 = the bin's first frame slot, cnt = live frames per bin on the device): eight launches of
// ~10 us each were 55-80 us of a batch whose other stages already run once over all bins.
// QLIM: the longest bin's quads in units of NT.  A kernel's registers are those of its widest path: with the
// variants of five and eight quads per thread compiled in, every bin ran at 256 VGPRs (two waves per SIMD) and a
// level-12 batch of 8192 blocks took 355 us for bytes that take 80; blocks of up to 4096 samples (QLIM 4)
// need four quads per thread at most.
template <int QLIM>
__global__ __launch_bounds__(NT, 4)
void k_prepare_stereo_bins(const int32_t *__restrict__ pcm, int32_t *__restrict__ smp,
                           fhip_subframe_info *__restrict__ info, int bps, int estimate,
                           const long long *__restrict__ frame_src, MultiBin mb)
{
    int blk = blockIdx.x;
    const int k = find_bin(mb, blk);
    blk -= mb.wg0[k];
    const int n = mb.n[k];
    const int nframes = __builtin_amdgcn_readfirstlane(mb.cnt[mb.cnt_ix[k]]);
    const size_t f0 = (size_t)mb.unit0[k];
    int32_t *smp_k = smp + mb.smp_off[k];
    fhip_subframe_info *info_k = info + f0 * 2;
    const long long *src_k = frame_src + f0;
    const int nar = mb.narrow[k];
    const int quads = n >> 2;
#define BODY_(M_, W_) prepare_stereo_body<M_, W_, true>(pcm, smp_k, info_k, n, bps, estimate, nar, nframes, src_k, blk)
    if constexpr (QLIM <= 4) {
        // FHIP_STEREO_GEOM's rule up to 4 NT quads
        if (quads <= 64) BODY_(1, 1);
        else if (quads <= 128) BODY_(1, 2);
        else if (quads <= 192) BODY_(3, 1);
        else if (quads <= NT) BODY_(1, 4);
        else if (quads <= 320) BODY_(2, 4);         // (five quads per thread would set this kernel's registers)
        else if (quads <= 2 * NT) BODY_(2, 4);
        else if (quads <= 3 * NT) BODY_(3, 4);
        else BODY_(4, 4);
    } else {
        // up to 4 NT quads as above; beyond, the block in two passes
        if (quads <= 64) BODY_(1, 1);
        else if (quads <= 128) BODY_(1, 2);
        else if (quads <= 192) BODY_(3, 1);
        else if (quads <= NT) BODY_(1, 4);
        else if (quads <= 320) BODY_(5, 1);
        else if (quads <= 2 * NT) BODY_(2, 4);
        else if (quads <= 3 * NT) BODY_(3, 4);
        else if (quads <= 4 * NT) BODY_(4, 4);
        else prepare_stereo_body_2p(pcm, smp_k, info_k, n, bps, estimate, nar, nframes, src_k, blk);
    }
#undef BODY_
}

}  // namespace

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------

hipError_t launch_prepare(hipStream_t st, const fhip_params &p, const int32_t *pcm,
                          int nframes, int n, int32_t *smp, fhip_subframe_info *info, bool decide_only,
                          bool allow_narrow, const long long *frame_src, const int32_t *dev_frames)
{
    const int nch = p.channels;
    if (nframes == 0) return hipSuccess;
    if (nch == 2 && (n & 3) == 0 && n <= 8192) {
        const int est = p.stereo_method == 1 ? 1 : 0;
        const int quads = n >> 2;
        const int nar = (allow_narrow && !decide_only) ? 1 : 0;
#define LAUNCH_PS(M_, W_, A_) hipLaunchKernelGGL((k_prepare_stereo<M_, W_, A_>), dim3((nframes + 4 / W_ - 1) / (4 / W_)), dim3(NT), 0, st, pcm, smp, info, n, p.bits_per_sample, est, nar, nframes, frame_src, dev_frames)
        if (decide_only) {
            if (quads <= NT) LAUNCH_PS(1, 4, false); else if (quads <= 2 * NT) LAUNCH_PS(2, 4, false); else LAUNCH_PS(4, 4, false);
        } else {
#define LAUNCH_PSA(M_, W_) LAUNCH_PS(M_, W_, true)
            FHIP_STEREO_GEOM(quads, LAUNCH_PSA);
#undef LAUNCH_PSA
        }
#undef LAUNCH_PS
        return hipGetLastError();
    }
    if (decide_only || allow_narrow) return hipErrorInvalidValue;
    if (nch != 2) {
        static const bool two_pass = getenv("FHIP_K0_MULTI_TWO_PASS") != nullptr;      // measurements only
        if (!two_pass && (n & 3) == 0 && n <= 8192 && n >= 256) {
            // the frame in registers: one read of the PCM
#define LAUNCH_MR(C_, M_) hipLaunchKernelGGL((k_prepare_multi_reg<C_, M_>), dim3(nframes), dim3(RT), 0, st, pcm, smp, info, n, p.bits_per_sample, frame_src, dev_frames)
#define LAUNCH_MRC(C_) do { if (n <= 4 * RT) LAUNCH_MR(C_, 1); else LAUNCH_MR(C_, 2); } while (0)
            switch (nch) {
            case 1: LAUNCH_MRC(1); break;
            case 3: LAUNCH_MRC(3); break;
            case 4: LAUNCH_MRC(4); break;
            case 5: LAUNCH_MRC(5); break;
            case 6: LAUNCH_MRC(6); break;
            case 7: LAUNCH_MRC(7); break;
            default: LAUNCH_MRC(8); break;
            }
#undef LAUNCH_MRC
#undef LAUNCH_MR
            return hipGetLastError();
        }
        hipLaunchKernelGGL(k_prepare_multi, dim3(nframes), dim3(NT), 0, st, pcm, smp, info, n, nch,
                           p.bits_per_sample, frame_src, dev_frames);
        return hipGetLastError();
    }
    const int blocks = nframes;
    const size_t lds = sizeof(int32_t) * (size_t)n * (nch == 2 ? 2 : 1);
    if (blocks == 0) return hipSuccess;
    if (lds > 150 * 1024) {
        // frames of more than ~19 k sample-frames: streamed from global memory
        hipLaunchKernelGGL(k_prepare<false>, dim3(blocks), dim3(NT), 0, st, pcm, smp, info, n, nch,
                           p.bits_per_sample, p.stereo_method == 1 ? 1 : 0, frame_src, dev_frames);
        return hipGetLastError();
    }
    hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_prepare<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (er != hipSuccess) return er;
    hipLaunchKernelGGL(k_prepare<true>, dim3(blocks), dim3(NT), lds, st, pcm, smp, info, n, nch,
                       p.bits_per_sample, p.stereo_method == 1 ? 1 : 0, frame_src, dev_frames);
    return hipGetLastError();
}

bool prepare_bins_supported(const fhip_params &p, const int *n, int nbins)
{
    if (p.channels != 2) return false;
    for (int k = 0; k < nbins; k++) if ((n[k] & 3) != 0 || n[k] > 8192 || n[k] < 4) return false;
    return true;
}

int prepare_bins_workgroups(int n, int cap, int nmax)
{
    int wpf = 4;
#define WPF_(M_, W_) wpf = W_
    FHIP_STEREO_GEOM(n >> 2, WPF_);
#undef WPF_
    if ((nmax >> 2) <= 4 * NT && (n >> 2) > NT && (n >> 2) <= 320) wpf = 4;       // k_prepare_stereo_bins<4>'s rule
    const int fpw = 4 / wpf;                       // frames per workgroup
    return (cap + fpw - 1) / fpw;
}

hipError_t launch_prepare_bins(hipStream_t st, const fhip_params &p, const int32_t *pcm, const MultiBin &mb,
                               int32_t *smp, fhip_subframe_info *info, const long long *frame_src)
{
    if (!prepare_bins_supported(p, mb.n, mb.nbins) || !frame_src) return hipErrorInvalidValue;
    const int blocks = mb.wg0[mb.nbins];
    if (blocks == 0) return hipSuccess;
    int nmax = 0;
    for (int k = 0; k < mb.nbins; k++) nmax = mb.n[k] > nmax ? mb.n[k] : nmax;
    if ((nmax >> 2) <= 4 * NT)
        hipLaunchKernelGGL(k_prepare_stereo_bins<4>, dim3(blocks), dim3(NT), 0, st, pcm, smp, info, p.bits_per_sample,
                           p.stereo_method == 1 ? 1 : 0, frame_src, mb);
    else
        hipLaunchKernelGGL(k_prepare_stereo_bins<8>, dim3(blocks), dim3(NT), 0, st, pcm, smp, info, p.bits_per_sample,
                           p.stereo_method == 1 ? 1 : 0, frame_src, mb);
    return hipGetLastError();
}

}  // namespace fhip
