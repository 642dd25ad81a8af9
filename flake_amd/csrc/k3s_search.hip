// k3s_search.hip -- K3-S: the LPC order searches of encode_residual() (optimize.c:201-261:
// 2/4/8-LEVEL, SEARCH, LOG) as a kernel of their own; K3's lean instance then encodes the winner.
#include "device_util.h"
#include "k3_common.h"

#ifdef FHIP_STAMPS
FHIP_DEFINE_STAMP_READER(fhip_debug_read_stamps_k3s)     // tools/stamps_k3s.py: phases of workgroup 0's rounds
#endif

namespace fhip {
namespace {

// ---------------------------------------------------------------------------
// K3-S  k_order_search<C, T, G>: the order searches of encode_residual()
// (optimize.c:201-261: 2/4/8-LEVEL, SEARCH, LOG) as a kernel of their own.
// ---------------------------------------------------------------------------
// bits[order] depends on nothing but the order, so the reference's sequential search is a
// walk over a table.  This kernel fills the table -- FIR (optimize.c:70-122), fold, partition
// sums, Rice parameter / partition-order search (rice.c:105-187) for every candidate order
// the method can visit -- G candidates per round: every wave runs the FIR and the fold of every
// candidate over its part of the block, the T thread sums go to LDS as 32-bit leaves, and then
// ONE WAVE PER CANDIDATE does the whole Rice search from them (wave_candidate_bits); nothing of
// the emit's state (residuals, parameters, windows) is carried.  (Round 2 also had an instance
// whose FIRs ran as fp64 matrix products, v_mfma_f64_16x16x4_f64 over 16 candidates x 16
// samples, exact because every product and sum is an integer below 2^53: once the search
// behind the FIR went wave-per-candidate in this instance too, the two measured the same --
// SEARCH 1-32 on 24-bit samples 1.33 against 1.41 ms, 1-24: 1.30 against 0.98, 1-12: 0.54
// against 0.43 -- and the matrix instance was dropped)
// It then replays the method's decision on the table and leaves the winner the way K2 leaves
// the single row of the MAX / EST methods: opt_order[s] and the compact row fin[s].  The lean
// K3 instance (MODE 0 / 3) encodes that row -- the reference's own final call
// (optimize.c:266-275) -- so est_bits, parameters and bits come from the same code as before.
template <int G>
struct SrchLds {
    unsigned long long *sums;   // [G][512] heap order per candidate slot; in leaf mode G per-wave heaps of 128
                                // and behind them the G x T thread sums as 32-bit leaves
    unsigned long long *wtot;   // [G][16]  per-wave totals
    double *coefd;              // [G][srch_crow(C)] candidate rows as doubles, zero past the order -- and past 32:
                                // fir_lpc's tap blocks of 20 / 28 (runs of 20, 28) read up to tap 36
    int32_t *smp;               // SmpImg<C, T>
    uint32_t *lvl_bits;         // [2][G][12] (double-buffered by round parity)
    uint32_t *lvl_meth;         // [2][G]
    int32_t *pairs;             // [G][8]   taps as int16 pairs (packed FIR)
    int32_t *rowi;              // [G][4]   order index, shift, sum |coef|, spare
    uint32_t *trial;            // [32]     bits[order index], 0xFFFFFFFF = not evaluated
    int32_t *list;              // [32]     candidate order indices of this subframe
    int32_t *misc;              // [32]: 0 count, 1 winner, 4.. a flag word per wave (<= 16), 24/25 overflow flags
};

// MM (the SEARCH method's FIRs on the int8 matrix pipe, k_order_search<.., true>): three planes of
// sample limbs (bytes, MM_HIST zeros in front), the coefficient limbs of all 32 candidate rows, four
// a heap per wave for the Rice search and two small tables; the leaves of the 16 candidates of a
// pass lie over the general way's sums (16 KB; 32 KB with 512 leaves).
// doubles per candidate row in LDS: fir_lpc's tap blocks of 20 / 28 (runs of 20, 28) read up to tap 36
__host__ __device__ constexpr int srch_crow(int c) { return (c % 8 == 4 && c > 16) ? 40 : 32; }
constexpr int MM_HIST = 32;
__host__ __device__ constexpr int mm_plane_bytes(int n) { return n + MM_HIST + 16; }

template <int G>
__host__ __device__ inline size_t srch_lds_layout(size_t img_ints, size_t off[16], int leaves, int mm_n = 0, int crow = 32)
{
    size_t o = 0;
    // sums: G x 4 KB for the general way; leaf mode overlays G heaps of 1 KB and G x NL leaves
    const size_t general = 8 * 512 * (size_t)G, leafy = 8 * 128 * (size_t)G + 4 * (size_t)G * (size_t)leaves;
    const size_t mm_leaves = mm_n ? 4 * 16 * (size_t)leaves : 0;      // the 16 candidates of a matrix pass
    const size_t s0 = general > leafy ? general : leafy;
    off[0] = o; o += s0 > mm_leaves ? s0 : mm_leaves;
    off[1] = o; o += 8 * 16 * G;
    off[2] = o; o += 8 * (size_t)crow * G;
    off[3] = o; o += 4 * img_ints;
    o = (o + 15) & ~(size_t)15;
    off[4] = o; o += 4 * 2 * G * 12;
    off[5] = o; o += 4 * 2 * G;
    off[6] = o; o += 4 * 8 * G;
    off[7] = o; o += 4 * 4 * G;
    off[8] = o; o += 4 * 32;
    off[9] = o; o += 4 * 32;
    off[10] = o; o += 4 * 32;
    o = (o + 15) & ~(size_t)15;
    off[11] = o; o += mm_n ? 0 : 2 * 32 * 32 + 4 * 64;          // every candidate row as int16, shifts, partition-order windows (the
                                                                // matrix instances keep their LDS for three workgroups per CU)
    o = (o + 15) & ~(size_t)15;
    off[12] = off[13] = off[14] = off[15] = 0;
    if (mm_n) {
        off[12] = o; o += 3 * (size_t)mm_plane_bytes(mm_n);      // sample limb planes
        off[13] = o; o += 2 * 32 * 32;                          // coefficient limbs [limb][candidate][32]
        off[14] = o; o += (size_t)(leaves / 64) * 128 * 8;       // a heap per wave
        off[15] = o; o += 4 * 64;                               // shift[32], pmin | pmax << 8 [32]
    }
    return (o + 15) & ~(size_t)15;
}


// The first round of a LOG walk (optimize.c:240-261) depends on nothing but the order range: the host plans
// it once per launch (log_plan_round0 below) instead of every workgroup on its critical path.
struct LogPlan0 { uint32_t pack; int ng, merged; };

// rice.c:105-187 for ONE candidate by ONE wave, from the finest-level sums a round of
// k_order_search left in LDS (leaf[t] = the sum of thread t's run): no barrier, no atomics.  A lane
// takes T/64 consecutive leaves and owns the nodes above them down to level 6 (always seven
// that can be asked for: four at level 8, two at 7, one at 6); levels 5..0 are 63 nodes built
// through a small per-wave heap in LDS and evaluated one per lane.  Level totals: a wave
// reduction for the three lane-local levels, a wave scan over the heap lanes for the rest.
// Returns the subframe estimate of calc_rice_params_lpc (rice.c:180-187).
template <int NL>
__device__ __forceinline__ uint32_t wave_candidate_bits(const uint32_t *__restrict__ leaf,
                                                        unsigned long long *__restrict__ heap, int n, int ord,
                                                        int pmin, int pmax, int obits, int precision, int lane)
{
    constexpr int LPL = NL / 64;                // leaves per lane: 4 (256 leaves), 8, 16
    constexpr int L8 = LPL / 4;                 // leaves per level-8 node
    static_assert(LPL >= 4 && LPL <= 16, "wave_candidate_bits: 256 .. 1024 tiles");
    uint32_t lf[LPL];
#pragma unroll
    for (int q = 0; q < LPL; q += 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(leaf + lane * LPL + q);
        lf[q] = v.x; lf[q + 1] = v.y; lf[q + 2] = v.z; lf[q + 3] = v.w;
    }
    unsigned long long s8[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        s8[i] = 0;
#pragma unroll
        for (int q = 0; q < L8; q++) s8[i] += lf[i * L8 + q];
    }
    const unsigned long long s7[2] = {s8[0] + s8[1], s8[2] + s8[3]};
    const unsigned long long s6 = s7[0] + s7[1];

    // The usual case -- every sum below 0xFFE00000 and no empty first partition -- takes the node
    // evaluation as straight-line 32-bit code (rice_k_u32_nb); the corners (32-bit noise summed over
    // four leaves, n >> p == order) keep the general form.  The choice is wave-uniform.
#ifndef FHIP_NODE_FAST
#define FHIP_NODE_FAST 1
#endif
    const bool corners = !FHIP_NODE_FAST || __any(s6 >= 0xFFE00000ull) || ((n >> pmax) - ord) <= 0;
    uint32_t lb[9];
#pragma unroll
    for (int p = 0; p < 9; p++) lb[p] = 0;
    uint32_t rice2 = 0;                         // bit p: some parameter of level p is above 14
    auto levels = [&](auto fast_c) {
        constexpr bool FAST = decltype(fast_c)::value;
        auto node = [&](unsigned long long sum, int p, int jn, uint32_t *b) -> int {
            const int cnt = (n >> p) - (jn == 0 ? ord : 0);
            if constexpr (FAST) return rice_k_u32_nb((uint32_t)sum, (uint32_t)cnt, b);
            else return (sum >> 32) ? rice_k_fast(sum, cnt, b) : rice_k_fast_u32((uint32_t)sum, cnt, b);
        };
        // ---- lane-local levels 8, 7, 6 ----
        {
            uint32_t b8 = 0, b7 = 0, b6 = 0;
            bool k8 = false, k7 = false, k6 = false;
            if (pmax >= 8 && pmin <= 8) {
#pragma unroll
                for (int i = 0; i < 4; i++) { uint32_t b; k8 |= node(s8[i], 8, 4 * lane + i, &b) > 14; b8 += b; }
            }
            if (pmax >= 7 && pmin <= 7) {
#pragma unroll
                for (int i = 0; i < 2; i++) { uint32_t b; k7 |= node(s7[i], 7, 2 * lane + i, &b) > 14; b7 += b; }
            }
            if (pmax >= 6 && pmin <= 6) { uint32_t b; k6 = node(s6, 6, lane, &b) > 14; b6 = b; }
            uint32_t t8 = b8, t7 = b7, t6 = b6;
#define WSUM(X_) do { X_ += dpp_u32<0x111>(X_); X_ += dpp_u32<0x112>(X_); X_ += dpp_u32<0x114>(X_);       \
                      X_ += dpp_u32<0x118>(X_); X_ += dpp_u32<0x142, 0xA>(X_); X_ += dpp_u32<0x143, 0xC>(X_); } while (0)
            WSUM(t8); WSUM(t7); WSUM(t6);
#undef WSUM
            lb[8] = (uint32_t)__builtin_amdgcn_readlane((int)t8, 63);
            lb[7] = (uint32_t)__builtin_amdgcn_readlane((int)t7, 63);
            lb[6] = (uint32_t)__builtin_amdgcn_readlane((int)t6, 63);
            if (__any(k8)) rice2 |= 1u << 8;
            if (__any(k7)) rice2 |= 1u << 7;
            if (__any(k6)) rice2 |= 1u << 6;
        }
        // ---- levels 5 .. 0: a 127-entry heap of this wave (entry 2^p - 1 + j = node j of level p) ----
        if (pmin <= 5) {
            // the sums of levels 5 .. 0 by a pyramid in registers (after step s, lanes = 0 mod 2^s hold level
            // 6 - s), ONE trip through the heap to hand node q to lane q (round 3; before: six dependent
            // store / load steps through LDS on the round's critical path)
            unsigned long long v = s6;
#define HEAP_STORE(S_) do { if ((lane & ((1 << (S_)) - 1)) == 0) heap[(1 << (6 - (S_))) - 1 + (lane >> (S_))] = v; } while (0)
            v += row_shl_u64<1>(v); HEAP_STORE(1);
            v += row_shl_u64<2>(v); HEAP_STORE(2);
            v += row_shl_u64<4>(v); HEAP_STORE(3);
            v += row_shl_u64<8>(v); HEAP_STORE(4);
            v += __shfl_down(v, 16, WAVE); HEAP_STORE(5);
            v += __shfl_down(v, 32, WAVE); HEAP_STORE(6);
#undef HEAP_STORE
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int p = ilog2_dev((uint32_t)(lane + 1));           // lanes 0..62: node `lane`, level p
            uint32_t b = 0;
            bool big = false;
            const unsigned long long hs = heap[min(lane, 62)];
            const unsigned long long total = heap[0];       // the block's total bounds every node of the heap
            if (lane < 63 && p >= pmin && p <= pmax) {
                const int jn = lane + 1 - (1 << p);
                const int cnt = (n >> p) - (jn == 0 ? ord : 0);
                if (FAST && total < 0xFFE00000ull) big = rice_k_u32_nb((uint32_t)hs, (uint32_t)cnt, &b) > 14;
                else big = ((hs >> 32) ? rice_k_fast(hs, cnt, &b) : rice_k_fast_u32((uint32_t)hs, cnt, &b)) > 14;
            }
            const uint32_t sc = wave_incl_scan_u32_dpp(b);
            const unsigned long long bigm = __ballot(big);
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)sc, (2 << q) - 2);
                const uint32_t lo = q ? (uint32_t)__builtin_amdgcn_readlane((int)sc, (1 << q) - 2) : 0u;
                lb[q] = hi - lo;
                const unsigned long long lvl = ((1ull << ((2 << q) - 1)) - 1) & ~((1ull << ((1 << q) - 1)) - 1);
                if (bigm & lvl) rice2 |= 1u << q;
            }
            __builtin_amdgcn_wave_barrier();                        // the heap is reused by the next candidate
        }
    };
    if (!corners) levels(std::true_type{});
    else levels(std::false_type{});
    // rice.c:127-138, :157-171, :180-187
    uint32_t best = 0, method = 0;
#pragma unroll
    for (int p = 0; p < 9; p++) {
        const uint32_t b = lb[p] + 4u * (1u << p);
        if (p >= pmin && p <= pmax && (p == pmin || b <= best)) { best = b; method = (rice2 >> p) & 1u; }
    }
    uint32_t bits = (uint32_t)(ord * obits + 2) + (uint32_t)(4 + 5 + ord * precision);
    bits += best;
    bits += method + 4u;
    return bits;
}


// The same for TWO candidates by one wave, a 32-lane half each (round 4): a lane takes NL / 32 consecutive leaves of its
// half's candidate and owns the nodes above them down to level 5 (fifteen that can be asked for: eight at level 8, four at 7,
// two at 6, one at 5); levels 4 .. 0 are 31 nodes built through a 32-entry heap of the half.  A wave per candidate spends
// ~350 of its ~510 instructions on the reductions, the heap trip and the level choice, whatever the number of nodes a lane
// evaluates: a 128-thread workgroup (two waves, four candidates a round) ran two such passes per wave in sequence.
// leaf / heap / ord / pmin / pmax are the HALF's (per lane); the result is the half's, in every lane of it.  The node forms
// are chosen wave-uniformly (both halves small, or the general form for both).
template <int NL>
__device__ __forceinline__ uint32_t wave_candidate_bits_x2(const uint32_t *__restrict__ leaf,
                                                           unsigned long long *__restrict__ heap, int n, int ord,
                                                           int pmin, int pmax, int obits, int precision, int lane)
{
    constexpr int LPL = NL / 32;                // leaves per lane: 8 (256 leaves), 16
    constexpr int L8 = NL / 256;                // leaves per level-8 node
    static_assert(NL == 256 || NL == 512, "wave_candidate_bits_x2: 256 or 512 leaves");
    const int hl = lane & 31, h32 = lane & 32;
    uint32_t lf[LPL];
#pragma unroll
    for (int q = 0; q < LPL; q += 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(leaf + hl * LPL + q);
        lf[q] = v.x; lf[q + 1] = v.y; lf[q + 2] = v.z; lf[q + 3] = v.w;
    }
    unsigned long long s8[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        s8[i] = 0;
#pragma unroll
        for (int q = 0; q < L8; q++) s8[i] += lf[i * L8 + q];
    }
    const unsigned long long s7[4] = {s8[0] + s8[1], s8[2] + s8[3], s8[4] + s8[5], s8[6] + s8[7]};
    const unsigned long long s6[2] = {s7[0] + s7[1], s7[2] + s7[3]};
    const unsigned long long s5 = s6[0] + s6[1];
    const bool corners = !FHIP_NODE_FAST || __any(s5 >= 0xFFE00000ull) || __any(((n >> pmax) - ord) <= 0);
    uint32_t lb[9];
#pragma unroll
    for (int p = 0; p < 9; p++) lb[p] = 0;
    uint32_t rice2 = 0;                         // bit p: some parameter of level p is above 14
    // inclusive scan inside each 32-lane half: four steps inside the rows, row 0 -> 1 and 2 -> 3
#define HSCAN(X_) do { X_ += dpp_u32<0x111>(X_); X_ += dpp_u32<0x112>(X_); X_ += dpp_u32<0x114>(X_);       \
                       X_ += dpp_u32<0x118>(X_); X_ += dpp_u32<0x142, 0xA>(X_); } while (0)
    auto levels = [&](auto fast_c) {
        constexpr bool FAST = decltype(fast_c)::value;
        auto node = [&](unsigned long long sum, int p, int jn, uint32_t *b) -> bool {
            const int cnt = (n >> p) - (jn == 0 ? ord : 0);
            if constexpr (FAST) return rice_k_u32_nb((uint32_t)sum, (uint32_t)cnt, b) > 14;
            else return ((sum >> 32) ? rice_k_fast(sum, cnt, b) : rice_k_fast_u32((uint32_t)sum, cnt, b)) > 14;
        };
        // ---- lane-local levels 8, 7, 6, 5 ----
        {
            uint32_t b8 = 0, b7 = 0, b6 = 0, b5 = 0;
            bool k8 = false, k7 = false, k6 = false, k5 = false;
            if (pmax >= 8 && pmin <= 8) {
#pragma unroll
                for (int i = 0; i < 8; i++) { uint32_t b; k8 |= node(s8[i], 8, 8 * hl + i, &b); b8 += b; }
            }
            if (pmax >= 7 && pmin <= 7) {
#pragma unroll
                for (int i = 0; i < 4; i++) { uint32_t b; k7 |= node(s7[i], 7, 4 * hl + i, &b); b7 += b; }
            }
            if (pmax >= 6 && pmin <= 6) {
#pragma unroll
                for (int i = 0; i < 2; i++) { uint32_t b; k6 |= node(s6[i], 6, 2 * hl + i, &b); b6 += b; }
            }
            if (pmax >= 5 && pmin <= 5) k5 = node(s5, 5, hl, &b5);
            HSCAN(b8); HSCAN(b7); HSCAN(b6); HSCAN(b5);
            lb[8] = (uint32_t)__shfl((int)b8, h32 | 31, WAVE);
            lb[7] = (uint32_t)__shfl((int)b7, h32 | 31, WAVE);
            lb[6] = (uint32_t)__shfl((int)b6, h32 | 31, WAVE);
            lb[5] = (uint32_t)__shfl((int)b5, h32 | 31, WAVE);
            if ((uint32_t)(__ballot(k8) >> h32)) rice2 |= 1u << 8;
            if ((uint32_t)(__ballot(k7) >> h32)) rice2 |= 1u << 7;
            if ((uint32_t)(__ballot(k6) >> h32)) rice2 |= 1u << 6;
            if ((uint32_t)(__ballot(k5) >> h32)) rice2 |= 1u << 5;
        }
        // ---- levels 4 .. 0: a 31-entry heap of the half (entry 2^p - 1 + j = node j of level p) ----
        {
            unsigned long long v = s5;
#define HEAP_STORE(S_) do { if ((hl & ((1 << (S_)) - 1)) == 0) heap[(1 << (5 - (S_))) - 1 + (hl >> (S_))] = v; } while (0)
            v += row_shl_u64<1>(v); HEAP_STORE(1);
            v += row_shl_u64<2>(v); HEAP_STORE(2);
            v += row_shl_u64<4>(v); HEAP_STORE(3);
            v += row_shl_u64<8>(v); HEAP_STORE(4);
            v += __shfl_down(v, 16, WAVE); HEAP_STORE(5);
#undef HEAP_STORE
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int p = ilog2_dev((uint32_t)(hl + 1));             // lanes 0 .. 30 of the half: node hl, level p
            uint32_t b = 0;
            bool big = false;
            const unsigned long long hs = heap[min(hl, 30)];
            // (a half's total bounds every node of its heap: the 32-bit form only where both halves' totals allow it)
            const bool small = FAST && __all(heap[0] < 0xFFE00000ull) != 0;
            if (hl < 31 && p >= pmin && p <= pmax) {
                const int jn = hl + 1 - (1 << p);
                const int cnt = (n >> p) - (jn == 0 ? ord : 0);
                if (small) big = rice_k_u32_nb((uint32_t)hs, (uint32_t)cnt, &b) > 14;
                else big = ((hs >> 32) ? rice_k_fast(hs, cnt, &b) : rice_k_fast_u32((uint32_t)hs, cnt, &b)) > 14;
            }
            HSCAN(b);
            const uint32_t bigm = (uint32_t)(__ballot(big) >> h32);
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const uint32_t hi = (uint32_t)__shfl((int)b, h32 | ((2 << q) - 2), WAVE);
                const uint32_t lo = q ? (uint32_t)__shfl((int)b, h32 | ((1 << q) - 2), WAVE) : 0u;
                lb[q] = hi - lo;
                const uint32_t lvl = ((1u << ((2 << q) - 1)) - 1u) & ~((1u << ((1 << q) - 1)) - 1u);
                if (bigm & lvl) rice2 |= 1u << q;
            }
            __builtin_amdgcn_wave_barrier();                        // the heaps are reused by the next candidates
        }
    };
    if (!corners) levels(std::true_type{});
    else levels(std::false_type{});
#undef HSCAN
    // rice.c:127-138, :157-171, :180-187
    uint32_t best = 0, method = 0;
#pragma unroll
    for (int p = 0; p < 9; p++) {
        const uint32_t b = lb[p] + 4u * (1u << p);
        if (p >= pmin && p <= pmax && (p == pmin || b <= best)) { best = b; method = (rice2 >> p) & 1u; }
    }
    uint32_t bits = (uint32_t)(ord * obits + 2) + (uint32_t)(4 + 5 + ord * precision);
    bits += best;
    bits += method + 4u;
    return bits;
}

// ---------------------------------------------------------------------------
// The FIRs of an order SEARCH on the int8 matrix pipe (round 3; tools/mfma_i8_probe.hip priced it)
// ---------------------------------------------------------------------------
// pred[c][i] = sum_t coef[c][t] x[i - t] (optimize.c:95-106) for 16 candidate rows at once is a matrix
// product, and an EXACT one on v_mfma_i32_16x16x64_i8 once both sides are cut into balanced int8
// limbs: x = x0 + 2^8 x1 + 2^16 x2 (|x| < 2^23), c = c0 + 2^8 c1 (|c| < 2^14).  The limb products of
// equal weight 2^(8w) share one instruction along K = 64 (chunks 0,1: c0 . x_w, taps 1..32; chunks
// 2,3: c1 . x_{w-1}): four instructions per tile of 16 samples x 16 candidates, int32 sums below 2^21.
// pred = lo + 2^16 hi with lo = P0 + 2^8 P1, hi = P2 + 2^8 P3 (exact in int32), and for shift <= 16
// pred >> shift = (hi << (16 - shift)) + (lo >> shift) in the low 32 bits -- all the reference's
// (int32)(x - (pred >> shift)) needs (optimize.c:108).
//
// A tile's 16 rows are the samples of equal index o in the 16 leaves (runs of 16) of a block of 256:
// sample 16 (16 B + m) + o for row m.  Every lane of a tile then needs its 16 operand bytes at the
// same misalignment o behind a 16-byte aligned address that does not depend on o: two aligned reads
// per plane and BLOCK, a funnel shift by a compile-time amount per tile; and after the block's 16
// tiles a lane holds the folded sums of its own four leaves (rows 4 g + r) for its candidate (column
// n): no cross-lane reduction.  Lane maps (probed): A[m = l & 15][k = 16 (l >> 4) + j],
// B[k = 16 (l >> 4) + j][n = l & 15], D[m = 4 (l >> 4) + r][n = l & 15].
template <int C, int T>
__device__ __forceinline__ void mm_search(const unsigned char *__restrict__ planes, const signed char *__restrict__ cl,
                                          const int32_t *__restrict__ tab, const int32_t *__restrict__ img,
                                          uint32_t *__restrict__ leaf, unsigned long long *__restrict__ heaps,
                                          uint32_t *__restrict__ trial, int n, int max_order, int obits, int precision,
                                          int tid)
{
    using Img = SmpImg<C, T>;
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef const v4i __attribute__((address_space(3))) *lds_v4;
    typedef const v2i __attribute__((address_space(3))) *lds_v2;
    typedef const int __attribute__((address_space(3))) *lds_i;
    constexpr int PB = mm_plane_bytes(C * T);
    constexpr int NW = T / WAVE;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the first block's branch below is wave-uniform
    const int g = lane >> 4, nn = lane & 15, h = g & 1;
    const unsigned lbase = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)planes;
    const unsigned xbase = (unsigned)(size_t)(const __attribute__((address_space(3))) int32_t *)img;

#pragma unroll 1
    for (int ct = 0; ct * 16 < max_order; ct++) {
        // ---- the pass's 16 rows as B operands: weight w pairs (c0, x_w) with (c1, x_{w-1}) ----
        v4i Bop[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int limb = g >> 1;
            const bool on = (limb == 0) ? (w <= 2) : (w >= 1);
            v4i v = *reinterpret_cast<const v4i *>(cl + (limb * 32 + ct * 16 + nn) * 32 + 16 * h);
            if (!on) v = v4i{0, 0, 0, 0};
            Bop[w] = v;
        }
        const int cand = ct * 16 + nn;                       // this lane's candidate: order cand + 1
        const int sh = tab[cand], sh16 = 16 - sh, ord = cand + 1;
        // One block of 16 leaves (256 / 16 per wave and pass).  FIRST: the subframe's first block, the only one
        // with warm-up samples -- its own copy of the code behind a wave-uniform branch, so that the other
        // blocks carry no per-element masks (round 4: as selects in one body they were four vector
        // instructions per element, 16 of the 76 a tile cost).
        auto block = [&](auto first_c, const int blk) {
            constexpr bool FIRST = decltype(first_c)::value;
            uint32_t acc[4] = {0, 0, 0, 0};
            // this lane's operand bytes of tile o: 16 bytes from byte rowoff + o of its plane -- a dword-aligned
            // start (C is a multiple of four), so the misalignment of a tile is the compile-time o & 3
            const unsigned rowoff = (unsigned)(MM_HIST + C * (16 * blk + nn) - 16 * (h + 1));
            unsigned ad[4];
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int pl = (g >> 1) == 0 ? w : w - 1;        // chunks 0,1: x_w; chunks 2,3: x_{w-1}
                const int plc = pl < 0 ? 0 : (pl > 2 ? 2 : pl);
                ad[w] = lbase + (unsigned)plc * PB + rowoff;
            }
            // up to 16 tiles at a time: their windows are NT + 15 bytes behind ad + OC, read once
            auto chunk = [&](auto oc_c) {
                constexpr int OC = decltype(oc_c)::value;
                constexpr int NT = (C - OC < 16) ? C - OC : 16;
                constexpr int ND = NT / 4 + 4;
                // The window's dwords are read INSIDE the tile loop, at compile-time offsets from ad[w]: the
                // same addresses for every tile, the compiler keeps one copy.  (Preloaded into an array ahead of
                // the loop they cost the kernel 50-76 spilled registers and configs[2] 0.2 ms.)
#pragma unroll
                for (int o4 = 0; o4 < NT; o4 += 4) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int oo = o4; oo < o4 + 4; oo++) {
                        const int o = OC + oo;
                        const int aa = oo >> 2, bb = oo & 3;
                        v4i Aop[4];
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            int W[8];
                            if constexpr (C % 16 == 0) {
                                const v4i R0 = *(lds_v4)(size_t)(ad[w] + OC);
                                v4i R1 = R0;
                                if (oo) R1 = *(lds_v4)(size_t)(ad[w] + OC + 16);
                                W[0] = R0.x; W[1] = R0.y; W[2] = R0.z; W[3] = R0.w;
                                W[4] = R1.x; W[5] = R1.y; W[6] = R1.z; W[7] = R1.w;
                            } else if constexpr (C % 8 == 0) {
                                // dwords aa .. aa + 4 of the window: three 8-byte reads at even dword offsets
                                const int e0 = aa & ~1;
#pragma unroll
                                for (int q = 0; q < 6; q += 2) {
                                    v2i R = v2i{0, 0};
                                    if (e0 + q < ND) R = *(lds_v2)(size_t)(ad[w] + OC + 4 * (e0 + q));
                                    W[q] = R.x; W[q + 1] = R.y;
                                }
                                W[6] = W[7] = 0;
                                if (aa & 1) {
#pragma unroll
                                    for (int q = 0; q < 5; q++) W[q] = W[q + 1];
                                }
                            } else {
#pragma unroll
                                for (int q = 0; q < 8; q++)
                                    W[q] = (q < 5 && aa + q < ND) ? *(lds_i)(size_t)(ad[w] + OC + 4 * (aa + q)) : 0;
                            }
                            const int a0 = (C % 16 == 0) ? aa : 0;
                            if (bb == 0) Aop[w] = v4i{W[a0], W[a0 + 1], W[a0 + 2], W[a0 + 3]};
                            else Aop[w] = v4i{(int)__builtin_amdgcn_alignbyte(W[a0 + 1], W[a0], bb),
                                              (int)__builtin_amdgcn_alignbyte(W[a0 + 2], W[a0 + 1], bb),
                                              (int)__builtin_amdgcn_alignbyte(W[a0 + 3], W[a0 + 2], bb),
                                              (int)__builtin_amdgcn_alignbyte(W[a0 + 4], W[a0 + 3], bb)};
                        }
                        // the samples of this lane's four rows: element o of the runs of threads 16 blk + 4 g + r
                        int32_t xs[4];
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            xs[r] = *(lds_i)(size_t)(xbase + 4u * (unsigned)((16 * blk + 4 * g + r) * 4 + Img::off(o & ~3) + (o & 3)));
                        v4i P[4];
#pragma unroll
                        for (int w = 0; w < 4; w++)
                            P[w] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aop[w], Bop[w], v4i{0, 0, 0, 0}, 0, 0, 0);
                        // eight vector instructions per element: lo, hi (shift-adds), lo >> sh, (hi << (16 - sh)) + that
                        // (one shift-add: written as such, the compiler's own order negates and shifts separately), the
                        // subtraction, the fold's two shifts, and xor-and-accumulate in one
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int32_t lo = P[0][r] + (P[1][r] << 8), hi = P[2][r] + (P[3][r] << 8);
                            const uint32_t q = lshl_add_u32((uint32_t)hi, (uint32_t)sh16, (uint32_t)(lo >> sh));
                            const int32_t res = (int32_t)((uint32_t)xs[r] - q);
                            // rice.c:85-94: partition 0 of every level starts at the order
                            if constexpr (FIRST) {
                                // o < ord - C (4 g + r).  The limit goes through an opaque move so that the 16 x 4 compares of
                                // the block stay where they are used: hoisted out of the tile loop they were 64 scalar pairs
                                // (150 scalar spills into vector lanes in this instance)
                                int lim = ord - C * (4 * g + r);
                                asm volatile("" : "+v"(lim));
                                const uint32_t u = zigzag32(res);
                                acc[r] += (o < lim) ? 0u : u;
                            } else {
                                acc[r] = xad_u32((uint32_t)res << 1, (uint32_t)(res >> 31), acc[r]);       // += zigzag32(res)
                            }
                        }
                    }
                }
            };
            chunk(std::integral_constant<int, 0>{});
            if constexpr (C > 16) chunk(std::integral_constant<int, 16>{});
            // leaves 16 blk + 4 g + r of this lane's candidate
            *reinterpret_cast<uint4 *>(leaf + nn * T + 16 * blk + 4 * g) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
        };
#pragma unroll 1
        for (int bq = 0; bq < 4; bq++) {
            const int blk = wv * 4 + bq;                     // block of 16 leaves: threads 16 blk .. 16 blk + 15 (wave-uniform)
            if (blk == 0) block(std::true_type{}, 0);
            else block(std::false_type{}, blk);
        }
        __syncthreads();
        // ---- rice.c:105-187 per candidate, a wave each ----
#ifndef FHIP_CAND_X2
#define FHIP_CAND_X2 1
#endif
        if constexpr (FHIP_CAND_X2 && (T == 256 || T == 512)) {
            // two candidates a wave (wave_candidate_bits_x2): a half whose candidate does not exist repeats the last one
            const int lastm = min(16, max_order - ct * 16) - 1;
            for (int m0 = 2 * wv; m0 <= lastm; m0 += 2 * NW) {
                const int m = m0 + (lane >> 5);
                const int mc = min(m, lastm);
                const int o1 = ct * 16 + mc + 1;
                const int pmm = tab[32 + ct * 16 + mc];
                const uint32_t b = wave_candidate_bits_x2<T>(leaf + mc * T, heaps + wv * 128 + (lane >> 5) * 32, n, o1, pmm & 0xFF,
                                                             pmm >> 8, obits, precision, lane);
                if ((lane & 31) == 0 && m <= lastm) trial[o1 - 1] = b;
            }
        } else {
            for (int m = wv; m < 16 && ct * 16 + m < max_order; m += NW) {
                const int o1 = ct * 16 + m + 1;
                const int pmm = tab[32 + ct * 16 + m];
                const uint32_t b = wave_candidate_bits<T>(leaf + m * T, heaps + wv * 128, n, o1, pmm & 0xFF, pmm >> 8,
                                                          obits, precision, lane);
                if (lane == 0) trial[o1 - 1] = b;
            }
        }
        __syncthreads();
    }
}

// rice.c:105-187 for ONE candidate by ONE wave from what mm_search32 left in LDS: the 64 sums of level 6 (s6, one per
// lane) and, per level 8 / 7 / 6, the bit totals and RICE2 flags of the eight slots (wave, lane half) that evaluated
// those levels' nodes beside the FIRs.  Levels 5 .. 0 as in wave_candidate_bits: a register pyramid, one trip through the
// wave's heap, node q on lane q.
__device__ __forceinline__ uint32_t wave_candidate_bits_s6(const uint32_t *__restrict__ s6, const uint4 *__restrict__ part,
                                                           unsigned long long *__restrict__ heap, int n, int ord,
                                                           int pmin, int pmax, int obits, int precision, int lane)
{
    uint32_t lb[9];
#pragma unroll
    for (int p = 0; p < 9; p++) lb[p] = 0;
    uint32_t rice2 = 0;
    {
        // levels 8, 7, 6: eight partial totals (x: level 8, y: 7, z: 6, w: RICE2 flags by level bit)
        uint4 v = make_uint4(0, 0, 0, 0);
        if (lane < 8) v = part[lane];
        uint32_t a = v.x, b = v.y, c = v.z, f = v.w;
#define RSUM(X_) do { X_ += dpp_u32<0x111>(X_); X_ += dpp_u32<0x112>(X_); X_ += dpp_u32<0x114>(X_); } while (0)
        RSUM(a); RSUM(b); RSUM(c);
#undef RSUM
        f |= dpp_u32<0x111>(f); f |= dpp_u32<0x112>(f); f |= dpp_u32<0x114>(f);
        lb[8] = (uint32_t)__builtin_amdgcn_readlane((int)a, 7);
        lb[7] = (uint32_t)__builtin_amdgcn_readlane((int)b, 7);
        lb[6] = (uint32_t)__builtin_amdgcn_readlane((int)c, 7);
        rice2 = (uint32_t)__builtin_amdgcn_readlane((int)f, 7) & 0x1C0u;
    }
    if (pmin <= 5) {
        unsigned long long v = s6[lane];
#define HEAP_STORE(S_) do { if ((lane & ((1 << (S_)) - 1)) == 0) heap[(1 << (6 - (S_))) - 1 + (lane >> (S_))] = v; } while (0)
        v += row_shl_u64<1>(v); HEAP_STORE(1);
        v += row_shl_u64<2>(v); HEAP_STORE(2);
        v += row_shl_u64<4>(v); HEAP_STORE(3);
        v += row_shl_u64<8>(v); HEAP_STORE(4);
        v += __shfl_down(v, 16, WAVE); HEAP_STORE(5);
        v += __shfl_down(v, 32, WAVE); HEAP_STORE(6);
#undef HEAP_STORE
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int p = ilog2_dev((uint32_t)(lane + 1));           // lanes 0..62: node `lane`, level p
        uint32_t b = 0;
        bool big = false;
        const unsigned long long hs = heap[min(lane, 62)];
        const unsigned long long total = heap[0];
        if (lane < 63 && p >= pmin && p <= pmax) {
            const int jn = lane + 1 - (1 << p);
            const int cnt = (n >> p) - (jn == 0 ? ord : 0);
            if (total < 0xFFE00000ull) big = rice_k_u32_nb((uint32_t)hs, (uint32_t)cnt, &b) > 14;
            else big = ((hs >> 32) ? rice_k_fast(hs, cnt, &b) : rice_k_fast_u32((uint32_t)hs, cnt, &b)) > 14;
        }
        const uint32_t sc = wave_incl_scan_u32_dpp(b);
        const unsigned long long bigm = __ballot(big);
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)sc, (2 << q) - 2);
            const uint32_t lo = q ? (uint32_t)__builtin_amdgcn_readlane((int)sc, (1 << q) - 2) : 0u;
            lb[q] = hi - lo;
            const unsigned long long lvl = ((1ull << ((2 << q) - 1)) - 1) & ~((1ull << ((1 << q) - 1)) - 1);
            if (bigm & lvl) rice2 |= 1u << q;
        }
        __builtin_amdgcn_wave_barrier();                        // the heap is reused by the next candidate
    }
    uint32_t best = 0, method = 0;
#pragma unroll
    for (int p = 0; p < 9; p++) {
        const uint32_t b = lb[p] + 4u * (1u << p);
        if (p >= pmin && p <= pmax && (p == pmin || b <= best)) { best = b; method = (rice2 >> p) & 1u; }
    }
    uint32_t bits = (uint32_t)(ord * obits + 2) + (uint32_t)(4 + 5 + ord * precision);
    bits += best;
    bits += method + 4u;
    return bits;
}

// The same for FOUR candidates by one wave, a 16-lane row each (candidate m0 + row): a lane takes four consecutive level-6
// sums and evaluates its two nodes of level 5 and its node of level 4; levels 3 .. 0 are 15 nodes built by a pyramid inside
// the row and handed to the row's lanes through a 16-entry heap.  (A wave per candidate spent ~240 instructions on 63
// nodes, one per lane, most of them reductions and the level choice: eight candidates a wave in sequence behind
// mm_search32's single pass -- a quarter of that kernel's instructions.)  Level totals are row sums; the level choice runs
// per lane on the row's values.
__device__ __forceinline__ void wave_candidates_s6x4(const uint32_t *__restrict__ s6all, const uint4 *__restrict__ part,
                                                     unsigned long long *__restrict__ heap, const int32_t *__restrict__ tab,
                                                     uint32_t *__restrict__ trial, int n, int m0, int max_order, int obits,
                                                     int precision, int lane)
{
    const int row16 = lane & 48, ll = lane & 15;
    const int m = m0 + (lane >> 4);
    const bool live = m < max_order;
    const int mc = live ? m : max_order - 1;
    const int pm = tab[32 + mc];
    const int pmin = pm & 0xFF, pmax = pm >> 8, ord = mc + 1;
    uint32_t lb[9];
#pragma unroll
    for (int p = 0; p < 9; p++) lb[p] = 0;
    uint32_t rice2;
#define ROWSCAN(X_) do { X_ += dpp_u32<0x111>(X_); X_ += dpp_u32<0x112>(X_); X_ += dpp_u32<0x114>(X_); X_ += dpp_u32<0x118>(X_); } while (0)
    {
        // levels 8, 7, 6: eight partial totals (x: level 8, y: 7, z: 6, w: RICE2 flags by level bit)
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ll < 8) v = part[mc * 8 + ll];
        uint32_t a = v.x, b = v.y, c = v.z, f = v.w;
        ROWSCAN(a); ROWSCAN(b); ROWSCAN(c);
        f |= dpp_u32<0x111>(f); f |= dpp_u32<0x112>(f); f |= dpp_u32<0x114>(f); f |= dpp_u32<0x118>(f);
        lb[8] = (uint32_t)__shfl((int)a, row16 | 15, WAVE);
        lb[7] = (uint32_t)__shfl((int)b, row16 | 15, WAVE);
        lb[6] = (uint32_t)__shfl((int)c, row16 | 15, WAVE);
        rice2 = (uint32_t)__shfl((int)f, row16 | 15, WAVE) & 0x1C0u;
    }
    {
        const uint4 q = *reinterpret_cast<const uint4 *>(s6all + mc * 64 + 4 * ll);
        const unsigned long long s5a = (unsigned long long)q.x + q.y, s5b = (unsigned long long)q.z + q.w;
        const unsigned long long s4 = s5a + s5b;
        // levels 3 .. 0 of the row: after step s, lanes = 0 mod 2^s hold node (ll >> s) of level 4 - s
        unsigned long long v = s4;
        unsigned long long *hp = heap + (row16 >> 4) * 16;
#define HEAP_STORE(S_) do { if ((ll & ((1 << (S_)) - 1)) == 0) hp[(1 << (4 - (S_))) - 1 + (ll >> (S_))] = v; } while (0)
        v += row_shl_u64<1>(v); HEAP_STORE(1);
        v += row_shl_u64<2>(v); HEAP_STORE(2);
        v += row_shl_u64<4>(v); HEAP_STORE(3);
        v += row_shl_u64<8>(v); HEAP_STORE(4);
#undef HEAP_STORE
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long hs = hp[min(ll, 14)];
        const unsigned long long total = hp[0];
        // (the straight-line 32-bit node form where every row of the wave allows it: a wave-uniform choice)
        const bool small = __all(total < 0xFFE00000ull) != 0;
        auto node = [&](unsigned long long sum, int p, int jn, uint32_t *b) -> bool {
            const int cnt = (n >> p) - (jn == 0 ? ord : 0);
            if (small) return rice_k_u32_nb((uint32_t)sum, (uint32_t)cnt, b) > 14;
            return ((sum >> 32) ? rice_k_fast(sum, cnt, b) : rice_k_fast_u32((uint32_t)sum, cnt, b)) > 14;
        };
        uint32_t b5 = 0, b4 = 0, bh = 0;
        bool big5 = false, big4 = false, bigh = false;
        if (pmin <= 5 && pmax >= 5) {
            uint32_t x, y;
            big5 = node(s5a, 5, 2 * ll, &x);
            big5 |= node(s5b, 5, 2 * ll + 1, &y);
            b5 = x + y;
        }
        if (pmin <= 4 && pmax >= 4) big4 = node(s4, 4, ll, &b4);
        const int p = ilog2_dev((uint32_t)(ll + 1));              // lanes 0 .. 14 of the row: node ll of the heap, level p
        if (ll < 15 && p >= pmin && p <= pmax) bigh = node(hs, p, ll + 1 - (1 << p), &bh);
        ROWSCAN(b5); ROWSCAN(b4); ROWSCAN(bh);
        lb[5] = (uint32_t)__shfl((int)b5, row16 | 15, WAVE);
        lb[4] = (uint32_t)__shfl((int)b4, row16 | 15, WAVE);
#pragma unroll
        for (int q2 = 0; q2 < 4; q2++) {
            const uint32_t hi = (uint32_t)__shfl((int)bh, row16 | ((2 << q2) - 2), WAVE);
            const uint32_t lo = q2 ? (uint32_t)__shfl((int)bh, row16 | ((1 << q2) - 2), WAVE) : 0u;
            lb[q2] = hi - lo;
        }
        const uint32_t m5 = (uint32_t)(__ballot(big5) >> row16) & 0xFFFFu, m4 = (uint32_t)(__ballot(big4) >> row16) & 0xFFFFu;
        const uint32_t mh = (uint32_t)(__ballot(bigh) >> row16) & 0xFFFFu;
        if (m5) rice2 |= 1u << 5;
        if (m4) rice2 |= 1u << 4;
#pragma unroll
        for (int q2 = 0; q2 < 4; q2++) {
            const uint32_t lvl = ((1u << ((2 << q2) - 1)) - 1u) & ~((1u << ((1 << q2) - 1)) - 1u);
            if (mh & lvl) rice2 |= 1u << q2;
        }
        __builtin_amdgcn_wave_barrier();                        // the heaps are reused by the wave's next four candidates
    }
#undef ROWSCAN
    // rice.c:127-138, :157-171, :180-187
    uint32_t best = 0, method = 0;
#pragma unroll
    for (int p = 0; p < 9; p++) {
        const uint32_t b = lb[p] + 4u * (1u << p);
        if (p >= pmin && p <= pmax && (p == pmin || b <= best)) { best = b; method = (rice2 >> p) & 1u; }
    }
    uint32_t bits = (uint32_t)(ord * obits + 2) + (uint32_t)(4 + 5 + ord * precision);
    bits += best;
    bits += method + 4u;
    if (live && ll == 0) trial[m] = bits;
}

// ---------------------------------------------------------------------------
// Round 4: the same FIRs for ALL 32 candidates in one pass on v_mfma_i32_32x32x32_i8 (256-thread instances)
// ---------------------------------------------------------------------------
// A tile is 32 samples x 32 candidates x 32 taps: rows = the samples of equal index o in the 32 leaves (runs of C) of a
// block, columns = the candidate orders 1 .. 32, K = the taps of ONE limb pair.  The six limb products (c0 x0 | c0 x1 + c1 x0 |
// c0 x2 + c1 x1 | c1 x2) are six instructions, equal weights chained through the accumulator input, so lo and hi come out as
// before.  Against the 16x16x64 form (mm_search): the operand windows' funnel shifts serve 1024 elements instead of 256, both
// halves of the candidates share them, the coefficient operands of all 32 rows are eight registers, and there is one pass, not
// two.  Lane maps (tools/mfma_i8_32_probe.hip): A[m = l & 31][k = 16 (l >> 5) + j], B[k][n = l & 31], D[m = (r & 3) + 8 (r >> 2) +
// 4 (l >> 5)][n = l & 31] in register r: a lane ends a block with the folded sums of 16 leaves of ITS candidate -- four aligned
// groups of four consecutive leaves -- and evaluates the nodes of levels 8, 7 and 6 above them right there (rice.c:105-187: the
// 448 of a candidate's 511 nodes that wave_candidate_bits spent a wave per candidate on); what goes to LDS is a sum per level-6
// node and three bit totals per (wave, lane half).  Levels 5 .. 0: wave_candidate_bits_s6, a wave per candidate as before.
// Every folded value must stay below 2^27 and every leaf below 2^29 (checked; true of any signal whose residuals are not many
// times full scale): otherwise the subframe takes the general way (returns false).
template <int C, int T>
__device__ __forceinline__ bool mm_search32(const unsigned char *__restrict__ planes, const signed char *__restrict__ cl,
                                            const int32_t *__restrict__ tab, const int32_t *__restrict__ img,
                                            uint32_t *__restrict__ s6all, unsigned long long *__restrict__ heaps,
                                            uint32_t *__restrict__ trial, int32_t *__restrict__ flagw, int n, int max_order,
                                            int obits, int precision, int tid, bool force_bad)
{
    static_assert(T == 256, "mm_search32: 256 leaves of one run each");
    using Img = SmpImg<C, T>;
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef int v16i __attribute__((ext_vector_type(16)));
    typedef const v4i __attribute__((address_space(3))) *lds_v4;
    typedef const v2i __attribute__((address_space(3))) *lds_v2;
    typedef const int __attribute__((address_space(3))) *lds_i;
    constexpr int PB = mm_plane_bytes(C * T);
    constexpr int NW = T / WAVE;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, cc = lane & 31;                  // K half / row half; this lane's candidate: order cc + 1
    const unsigned lbase = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)planes;
    const unsigned xbase = (unsigned)(size_t)(const __attribute__((address_space(3))) int32_t *)img;
    uint4 *part = reinterpret_cast<uint4 *>(s6all + 32 * 64);   // [32 candidates][8 slots]

    // the candidates' rows as B operands: limb 0 and limb 1, this lane's 16 taps of its K half
    const v4i B0 = *reinterpret_cast<const v4i *>(cl + (0 * 32 + cc) * 32 + 16 * h);
    const v4i B1 = *reinterpret_cast<const v4i *>(cl + (1 * 32 + cc) * 32 + 16 * h);
    const int sh = tab[cc], sh16 = 16 - sh, ord = cc + 1;
    const int pmm = tab[32 + cc];
    const int pmin = pmm & 0xFF, pmax = pmm >> 8;
    uint32_t bt8 = 0, bt7 = 0, bt6 = 0, r2 = 0;               // bit totals of this lane's nodes, RICE2 flags (bit = level)
    uint32_t bad = force_bad ? 1u : 0u;                       // a folded value of 2^27 or a leaf of 2^29 and more

    auto block = [&](auto first_c, const int blk) {
        constexpr bool FIRST = decltype(first_c)::value;
        uint32_t acc[16];
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0;
        uint32_t uor = 0;
        // this lane's operand bytes of tile o: 16 bytes from byte rowoff + o of a plane (a dword-aligned start);
        // the samples of its rows: element o of the runs of leaves 32 blk + 4 h + (r & 3) + 8 (r >> 2)
        const unsigned ad0 = lbase + (unsigned)(MM_HIST + C * (32 * blk + cc) - 16 * (h + 1));
        const unsigned xr0 = xbase + 4u * (unsigned)((32 * blk + 4 * h) * 4 + Img::off(0));
        const int lim0 = ord - C * 4 * h;                     // FIRST: row m's warm-up samples are o < lim0 - C (m - 4 h)
        // Tiles in groups of four (o = 4 aa + bb): the misalignment bb is a compile-time constant, the group a real
        // loop -- fully unrolled, the compiler hoisted per-tile addresses and masks out of it and spilled 140 .. 2400
        // registers.  A group reads five dwords per plane (its window) and, per element, the sample itself.
#pragma unroll 1
        for (int aa = 0; aa < C / 4; aa++) {
            const unsigned adg = ad0 + 4u * (unsigned)aa;
            const unsigned xrg = xr0 + (unsigned)aa * (unsigned)(4 * (Img::off(4) - Img::off(0)));
            int W[3][5];
#pragma unroll
            for (int pl = 0; pl < 3; pl++)
#pragma unroll
                for (int q = 0; q < 5; q++) W[pl][q] = *(lds_i)(size_t)(adg + (unsigned)pl * PB + 4u * q);
#pragma unroll
            for (int bb = 0; bb < 4; bb++) {
                __builtin_amdgcn_sched_barrier(0);           // a tile at a time: tiles interleaved hold 64 result registers each
                v4i Aop[3];
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    if (bb == 0) Aop[pl] = v4i{W[pl][0], W[pl][1], W[pl][2], W[pl][3]};
                    else Aop[pl] = v4i{(int)__builtin_amdgcn_alignbyte(W[pl][1], W[pl][0], bb),
                                       (int)__builtin_amdgcn_alignbyte(W[pl][2], W[pl][1], bb),
                                       (int)__builtin_amdgcn_alignbyte(W[pl][3], W[pl][2], bb),
                                       (int)__builtin_amdgcn_alignbyte(W[pl][4], W[pl][3], bb)};
                }
                const v16i Z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                const v16i P0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Aop[0], B0, Z, 0, 0, 0);
                v16i P1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Aop[0], B1, Z, 0, 0, 0);
                v16i P2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Aop[1], B1, Z, 0, 0, 0);
                const v16i P3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Aop[2], B1, Z, 0, 0, 0);
                P1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Aop[1], B0, P1, 0, 0, 0);
                P2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Aop[2], B0, P2, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int mr = (r & 3) + 8 * (r >> 2);                 // the row: leaf 32 blk + 4 h + mr of the subframe
                    const int32_t xs = *(lds_i)(size_t)(xrg + 4u * (unsigned)(mr * 4 + bb));
                    const int32_t lo = P0[r] + (P1[r] << 8), hi = P2[r] + (P3[r] << 8);
                    const uint32_t q = lshl_add_u32((uint32_t)hi, (uint32_t)sh16, (uint32_t)(lo >> sh));
                    const int32_t res = (int32_t)((uint32_t)xs - q);
                    // rice.c:122; 2 res as a full-rate add (the compiler's own choice is the half-rate shift)
                    uint32_t u = add_u32((uint32_t)res, (uint32_t)res) ^ (uint32_t)(res >> 31);
                    if constexpr (FIRST) {
                        // rice.c:85-94: partition 0 of every level starts at the order
                        u = (4 * aa + bb < lim0 - C * mr) ? 0u : u;
                    }
                    acc[r] += u;
                    uor |= u;
                }
            }
        }

        // ---- the nodes of levels 8, 7, 6 above this lane's 16 leaves ----
        uint32_t lor = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) lor |= acc[r];
        bad |= (uor >> 27) | (lor >> 29);
        const uint32_t n8 = (uint32_t)(n >> 8), n7 = (uint32_t)(n >> 7), n6 = (uint32_t)(n >> 6);
        const bool on8 = pmin <= 8 && pmax >= 8, on7 = pmin <= 7 && pmax >= 7, on6 = pmin <= 6 && pmax >= 6;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t a0 = acc[4 * g], a1 = acc[4 * g + 1], a2 = acc[4 * g + 2], a3 = acc[4 * g + 3];
            const uint32_t s70 = a0 + a1, s71 = a2 + a3, s60 = s70 + s71;
            // node j of a level owns the subframe's first samples when it holds leaf 0: block 0, group 0, lane half 0
            const bool head = FIRST && g == 0 && h == 0;
            uint32_t b;
            int k;
            // (the subframe's first node of a level has n >> p - order samples, possibly none: the form with every corner,
            // on the lanes that count it only)
            auto node = [&](uint32_t sum, uint32_t cnt, bool first_node, bool on) {
                if (FIRST && first_node) { b = 0; k = 0; if (on) k = rice_k_fast_u32(sum, (int)cnt - ord, &b); }
                else k = rice_k_u32_nb(sum, cnt, &b);
            };
            if (__any(on8)) {
                node(a0, n8, head, on8); bt8 += on8 ? b : 0u; r2 |= (on8 && k > 14) ? 0x100u : 0u;
                node(a1, n8, false, on8); bt8 += on8 ? b : 0u; r2 |= (on8 && k > 14) ? 0x100u : 0u;
                node(a2, n8, false, on8); bt8 += on8 ? b : 0u; r2 |= (on8 && k > 14) ? 0x100u : 0u;
                node(a3, n8, false, on8); bt8 += on8 ? b : 0u; r2 |= (on8 && k > 14) ? 0x100u : 0u;
            }
            if (__any(on7)) {
                node(s70, n7, head, on7); bt7 += on7 ? b : 0u; r2 |= (on7 && k > 14) ? 0x80u : 0u;
                node(s71, n7, false, on7); bt7 += on7 ? b : 0u; r2 |= (on7 && k > 14) ? 0x80u : 0u;
            }
            if (__any(on6)) {
                node(s60, n6, head, on6); bt6 += on6 ? b : 0u; r2 |= (on6 && k > 14) ? 0x40u : 0u;
            }
            s6all[cc * 64 + 8 * blk + 2 * g + h] = s60;
        }
    };
#pragma unroll 1
    for (int bq = 0; bq < 8 / NW; bq++) {
        const int blk = wv * (8 / NW) + bq;                   // block of 32 leaves (wave-uniform)
        if (blk == 0) block(std::true_type{}, 0);
        else block(std::false_type{}, blk);
    }
    part[cc * 8 + 2 * wv + h] = make_uint4(bt8, bt7, bt6, r2);
    if (__any(bad != 0u) && lane == 0) atomicOr(reinterpret_cast<uint32_t *>(flagw), 1u);
    __syncthreads();
    if (*flagw != 0) return false;                            // workgroup-uniform: the general way does this subframe
    // ---- levels 5 .. 0 and the level choice per candidate, a wave each ----
#ifndef FHIP_S6X4
#define FHIP_S6X4 1
#endif
    if (FHIP_S6X4) {
        for (int m0 = 4 * wv; m0 < max_order; m0 += 4 * NW)
            wave_candidates_s6x4(s6all, part, heaps + wv * 128, tab, trial, n, m0, max_order, obits, precision, lane);
    } else {
        for (int m = wv; m < max_order; m += NW) {
            const int pm = tab[32 + m];
            const uint32_t b = wave_candidate_bits_s6(s6all + m * 64, part + m * 8, heaps + wv * 128, n, m + 1, pm & 0xFF,
                                                      pm >> 8, obits, precision, lane);
            if (lane == 0) trial[m] = b;
        }
    }
    __syncthreads();
    return true;
}

// Geometries: runs of C samples (whole groups of four) in T threads, n = C * T.  The finest
// partition sums there are -- the LEAVES -- are the 256 partitions of level 8 (T <= 256: one or
// two per thread) or the T thread sums; every piece of a variable-block-size stream (k eighths of
// a 4096 or 8192 block) is 256 leaves of 2k or 4k samples: T = 256 with runs of 4k, T = 128 with
// runs of 4k for the odd eighths of a 4096 block.
template <int C, int T, int G, bool MM = false>
// (runs of 20 .. 28 samples at three waves per SIMD, 168 VGPRs: at four they spill up to 200 bytes per lane)
#ifndef FHIP_SRCH_WLONG
#define FHIP_SRCH_WLONG 3
#endif
// (the matrix instances of runs >= 20 hold two workgroups per CU by their LDS -- image, limb planes, leaves: 59-72 KB)
// The search of subframe s (an index into the subframe-indexed workspaces; its samples at smp_all + s n): the body of
// k_order_search (one geometry: a batch, or one bin of a ragged batch) and of k_order_search_bins (several thinly
// filled bins of a ragged batch in one launch).
__device__ __forceinline__
void order_search_body(const fhip_params &P, const int n, const int32_t *__restrict__ smp_all,
                       const int32_t *__restrict__ coefs_all, const int32_t *__restrict__ shift_all,
                       int32_t *__restrict__ opt_all, int32_t *__restrict__ fin_all,
                       const fhip_subframe_info *__restrict__ prep, const int narrow_ok,
                       uint32_t *__restrict__ table_out, const LogPlan0 &lg0, const int s)
{
    static_assert(C % 4 == 0 && T >= 128 && (T & (T - 1)) == 0, "k_order_search: runs of whole groups of four");
    using Img = SmpImg<C, T>;
    constexpr int LT = clog2(T);
    constexpr int NW = T / WAVE;
    constexpr int NL = (T < 256) ? 256 : T;            // leaves
    constexpr int LPT = NL / T;                        // ... per thread: 1, or 2 (T = 128: half runs)
    static_assert(LPT == 1 || (C / 2) % 2 == 0, "half runs of whole sample pairs");
#ifndef LOG_MERGE
#define LOG_MERGE 1
#endif
    static_assert(!MM || (T >= 256 && C <= 32), "the matrix path: one leaf per thread, runs of up to 32");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    size_t off[16];
    constexpr int SRCH_CROW = srch_crow(C);
    srch_lds_layout<G>((size_t)Img::SIZE, off, NL, MM ? C * T : 0, SRCH_CROW);
    SrchLds<G> l;
    l.sums = reinterpret_cast<unsigned long long *>(lds_raw + off[0]);
    l.wtot = reinterpret_cast<unsigned long long *>(lds_raw + off[1]);
    l.coefd = reinterpret_cast<double *>(lds_raw + off[2]);
    l.smp = reinterpret_cast<int32_t *>(lds_raw + off[3]);
    l.lvl_bits = reinterpret_cast<uint32_t *>(lds_raw + off[4]);
    l.lvl_meth = reinterpret_cast<uint32_t *>(lds_raw + off[5]);
    l.pairs = reinterpret_cast<int32_t *>(lds_raw + off[6]);
    l.rowi = reinterpret_cast<int32_t *>(lds_raw + off[7]);
    l.trial = reinterpret_cast<uint32_t *>(lds_raw + off[8]);
    l.list = reinterpret_cast<int32_t *>(lds_raw + off[9]);
    l.misc = reinterpret_cast<int32_t *>(lds_raw + off[10]);
    int16_t *crows = reinterpret_cast<int16_t *>(lds_raw + off[11]);            // [32][32] K2's rows (|coef| < 2^15)
    int32_t *cshifts = reinterpret_cast<int32_t *>(lds_raw + off[11] + 2 * 32 * 32);

    // fp64 rounding toward -inf: fir_lpc's floor (see k_encode_pow2)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2" ::: "memory");

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef FHIP_WV_VECTOR
    const int wv = tid >> 6;
#else
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: what only wave 0 does stays scalar code
#endif
    FastCtx<C, T> e;
    e.l.smp = l.smp; e.l.sums = nullptr; e.l.kpar = nullptr; e.l.coefd = l.coefd; e.l.wtot = nullptr;
    e.l.lvl_bits = nullptr; e.l.lvl_meth = nullptr; e.l.coef = nullptr; e.l.misc = nullptr;
    e.l.trial = nullptr; e.l.bits = nullptr;
    e.n = n; e.tid = tid; e.lane = lane; e.wv = wv; e.i0 = tid * C;
    e.obits = prep[s].obits;
    e.precision = P.lpc_precision;
    e.pmin_req = P.min_partition_order;
    e.pmax_req = P.max_partition_order;

    // ---- stage the samples (the same image k_encode_pow2 builds) ----------------------
    int magbits = -1;
    int differs;
    bool mm_fits = false;
    {
        const int32_t *srcp = smp_all + (size_t)s * n;
        const int rflag = prep[s].reserved;                 // K0's: low byte 16-bit row, bits 8.. 1 + magbits
        const int nflag = narrow_ok ? (rflag & 0xFF) : 0;
        magbits = (rflag >> 8) - 1;                          // -1: not known (K0 variants other than the stereo one)
        int32_t xn[C];
        int32_t first;
        if (nflag != 0 && C % 8 != 0) {
            // runs of 4, 12, 20, 28 samples: eight bytes at a time
            const int2 *src2 = reinterpret_cast<const int2 *>(reinterpret_cast<const int16_t *>(srcp) + e.i0);
#pragma unroll
            for (int q = 0; q < C / 4; q++) {
                const int2 t2 = src2[q];
                xn[4 * q] = (int32_t)(int16_t)t2.x;     xn[4 * q + 1] = t2.x >> 16;
                xn[4 * q + 2] = (int32_t)(int16_t)t2.y; xn[4 * q + 3] = t2.y >> 16;
            }
            first = (int32_t)*reinterpret_cast<const int16_t *>(srcp);
        } else if (nflag != 0) {
            const int4 *src4 = reinterpret_cast<const int4 *>(reinterpret_cast<const int16_t *>(srcp) + e.i0);
#pragma unroll
            for (int q = 0; q < C / 8; q++) {
                const int4 t4 = src4[q];
                xn[8 * q] = (int32_t)(int16_t)t4.x;     xn[8 * q + 1] = t4.x >> 16;
                xn[8 * q + 2] = (int32_t)(int16_t)t4.y; xn[8 * q + 3] = t4.y >> 16;
                xn[8 * q + 4] = (int32_t)(int16_t)t4.z; xn[8 * q + 5] = t4.z >> 16;
                xn[8 * q + 6] = (int32_t)(int16_t)t4.w; xn[8 * q + 7] = t4.w >> 16;
            }
            first = (int32_t)*reinterpret_cast<const int16_t *>(srcp);
        } else {
            const int4 *src4 = reinterpret_cast<const int4 *>(srcp + e.i0);
#pragma unroll
            for (int q = 0; q < C / 4; q++) {
                const int4 t4 = src4[q];
                xn[4 * q] = t4.x; xn[4 * q + 1] = t4.y; xn[4 * q + 2] = t4.z; xn[4 * q + 3] = t4.w;
            }
            first = srcp[0];
        }
        int32_t mx = first, mn = first;
#pragma unroll
        for (int o = 0; o < C; o++) { mx = max(mx, xn[o]); mn = min(mn, xn[o]); }
        differs = (mx != mn);
        if constexpr (MM) {
            // three balanced int8 limbs per sample, x = x0 + 2^8 x1 + 2^16 x2, as byte planes in sample
            // order; the top limb holds samples from -2^23 up to 2^23 - 32897 (beyond: the vector way)
            mm_fits = (mx <= 8355711) && (mn >= -8388608);
            uint32_t b0[C / 4], b1[C / 4], b2[C / 4];
#pragma unroll
            for (int q = 0; q < C / 4; q++) { b0[q] = 0; b1[q] = 0; b2[q] = 0; }
#pragma unroll
            for (int o = 0; o < C; o++) {
                const int32_t v = xn[o];
                const int32_t x0 = (int32_t)(int8_t)v;
                const int32_t r1 = (v - x0) >> 8;
                const int32_t x1 = (int32_t)(int8_t)r1;
                const int32_t x2 = (r1 - x1) >> 8;
                b0[o >> 2] |= ((uint32_t)x0 & 0xFFu) << (8 * (o & 3));
                b1[o >> 2] |= ((uint32_t)x1 & 0xFFu) << (8 * (o & 3));
                b2[o >> 2] |= ((uint32_t)x2 & 0xFFu) << (8 * (o & 3));
            }
            unsigned char *pl = lds_raw + off[12] + MM_HIST + C * tid;
            constexpr int PB = mm_plane_bytes(C * T);
            if constexpr (C % 16 == 0) {
#pragma unroll
                for (int q = 0; q < C / 4; q += 4) {
                    *reinterpret_cast<uint4 *>(pl + 4 * q) = make_uint4(b0[q], b0[q + 1], b0[q + 2], b0[q + 3]);
                    *reinterpret_cast<uint4 *>(pl + PB + 4 * q) = make_uint4(b1[q], b1[q + 1], b1[q + 2], b1[q + 3]);
                    *reinterpret_cast<uint4 *>(pl + 2 * PB + 4 * q) = make_uint4(b2[q], b2[q + 1], b2[q + 2], b2[q + 3]);
                }
            } else if constexpr (C % 8 == 0) {
#pragma unroll
                for (int q = 0; q < C / 4; q += 2) {
                    *reinterpret_cast<uint2 *>(pl + 4 * q) = make_uint2(b0[q], b0[q + 1]);
                    *reinterpret_cast<uint2 *>(pl + PB + 4 * q) = make_uint2(b1[q], b1[q + 1]);
                    *reinterpret_cast<uint2 *>(pl + 2 * PB + 4 * q) = make_uint2(b2[q], b2[q + 1]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < C / 4; q++) {
                    *reinterpret_cast<uint32_t *>(pl + 4 * q) = b0[q];
                    *reinterpret_cast<uint32_t *>(pl + PB + 4 * q) = b1[q];
                    *reinterpret_cast<uint32_t *>(pl + 2 * PB + 4 * q) = b2[q];
                }
            }
        }
#pragma unroll
        for (int g4 = 0; g4 < C; g4 += 4)
            *reinterpret_cast<int4 *>(l.smp + tid * 4 + Img::off(g4)) =
                make_int4(xn[g4], xn[g4 + 1], xn[g4 + 2], xn[g4 + 3]);
    }
    if (tid < Img::COL0 * C) l.smp[Img::at(tid / C, tid % C)] = 0;
    if (tid < 32) l.trial[tid] = 0xFFFFFFFFu;
    if constexpr (!MM) {
        // K2's 32 x 32 rows and their shifts into LDS once, beside the samples: a round then stages its
        // candidates from LDS -- a global load per round sat on the round's critical path
        const int32_t *cb = coefs_all + (size_t)s * FHIP_MAX_ORDER * FHIP_MAX_ORDER;
        for (int q = tid; q < FHIP_MAX_ORDER * FHIP_MAX_ORDER / 4; q += T) {
            const int4 v = reinterpret_cast<const int4 *>(cb)[q];
            *reinterpret_cast<uint2 *>(crows + 4 * q) =
                make_uint2(((uint32_t)v.x & 0xFFFFu) | ((uint32_t)v.y << 16), ((uint32_t)v.z & 0xFFFFu) | ((uint32_t)v.w << 16));
        }
        if (tid < FHIP_MAX_ORDER) {
            cshifts[tid] = shift_all[(size_t)s * FHIP_MAX_ORDER + tid];
            // rice.c:148-155 for every order, once (an integer division each: off the rounds' path)
            cshifts[32 + tid] = clamp_porder(P.min_partition_order, n, tid + 1) | (clamp_porder(P.max_partition_order, n, tid + 1) << 8);
        }
    }
    if (tid < 3) l.misc[24 + tid] = 0;                 // rounds: "a thread sum left 32 bits", per round parity; 26: mm_search32's

    STAMP(0);
    const int omethod = P.order_method;
    const int min_order = P.min_prediction_order, max_order = P.max_prediction_order;
    const int32_t *crow_base = coefs_all + (size_t)s * FHIP_MAX_ORDER * FHIP_MAX_ORDER;
    const int32_t *srow = shift_all + (size_t)s * FHIP_MAX_ORDER;

    // the orders the method can visit (indices = order - 1), each once
    if (tid == 0) {
        int nc = 0;
        if (omethod >= 2 && omethod <= 4) {
            const int levels = 1 << (omethod - 1);
            uint32_t seen = 0;
            for (int i = levels - 1; i >= 0; i--) {
                int o = min_order + (((max_order - min_order + 1) * (i + 1)) / levels) - 2;
                if (o < 0) o = 0;
                if (!((seen >> o) & 1u)) { seen |= 1u << o; l.list[nc++] = o; }
            }
        } else if (omethod == 5) {
            for (int i = 0; i < max_order; i++) l.list[nc++] = i;
        }
        // (LOG, order method 6: the candidates of a step depend on the winner so far; its
        // rounds are the steps of optimize.c:244-261, see below)
        l.misc[0] = nc;
    }
    signed char *mm_cl = reinterpret_cast<signed char *>(lds_raw + off[13]);
    int32_t *mm_tab = reinterpret_cast<int32_t *>(lds_raw + off[15]);
    if constexpr (MM) {
        // the limbs c = c0 + 2^8 c1 of every candidate row (zero past its order), 16 taps to a K
        // chunk, bytes in the order of the samples they meet: chunk h holds taps 16 h + 16 .. 16 h + 1
        for (int q = tid; q < 32 * 32; q += T) {
            const int c = q >> 5, t = q & 31;                // tap t + 1 of the row of order c + 1
            const int32_t v = (t <= c && c < max_order) ? crow_base[c * FHIP_MAX_ORDER + t] : 0;
            const int32_t c0 = (int32_t)(int8_t)v, c1 = (v - c0) >> 8;
            const int h = t >> 4, j = 15 - (t & 15);
            mm_cl[(0 * 32 + c) * 32 + 16 * h + j] = (signed char)c0;
            mm_cl[(1 * 32 + c) * 32 + 16 * h + j] = (signed char)c1;
        }
        if (tid < 32) {
            mm_tab[tid] = (tid < max_order) ? srow[tid] : 0;
            mm_tab[32 + tid] = clamp_porder(e.pmin_req, n, tid + 1) | (clamp_porder(e.pmax_req, n, tid + 1) << 8);
        }
        if (tid < 6) {                                       // MM_HIST zeros in front of every plane
            unsigned char *z = lds_raw + off[12] + (tid >> 1) * mm_plane_bytes(C * T) + 16 * (tid & 1);
            *reinterpret_cast<uint4 *>(z) = make_uint4(0, 0, 0, 0);
        }
    }
    // per wave: bit 0 "some sample differs from the first", bit 1 (matrix instances) "the limbs hold every sample"
    const int wave_flags = ((__ballot(differs) != 0ull) ? 1 : 0) | ((MM && __all(mm_fits)) ? 2 : 0);
    if (lane == 0) l.misc[4 + wv] = wave_flags;
    __syncthreads();
    int any_differs = 0, all_fit = 2;
#pragma unroll
    for (int w = 0; w < NW; w++) { any_differs |= l.misc[4 + w] & 1; all_fit &= l.misc[4 + w]; }
    const bool constant = (__builtin_amdgcn_readfirstlane(any_differs) == 0);
    STAMP(1);
    const int nc = __builtin_amdgcn_readfirstlane(l.misc[0]);
    bool mm_done = false;
    if constexpr (MM) {
        const int fits = all_fit;
        // SEARCH (optimize.c:224-238) on samples the limbs hold: bits[order] of every order from the
        // matrix pipe, 16 candidates to a pass
        if (__builtin_amdgcn_readfirstlane(fits) && !constant && omethod == 5) {
            // (16-bit blocks too: leaving their orders 1..16 to the packed dot products of the rounds below and
            // only the orders above to the matrix pipe measured SLOWER -- SEARCH 1-32 at 16 bits 0.99 against
            // 0.89 ms, level 12 1.43 against 1.30 -- and a run-time choice between the two cost the 24-bit
            // path 6 %: the compiler must then keep the rounds' state alive across the matrix passes)
#ifndef FHIP_MM32
#define FHIP_MM32 1
#endif
            // (runs of four -- the 1024-sample pieces of a variable-block-size stream -- keep the 16x16x64 form: a block
            // is a single group of four tiles there and a wave has nothing to put beside the six dependent matrix
            // instructions of a tile: 184 against 161 us for the level-12 batch's 16384 such pieces)
            if constexpr (T == 256 && C >= 8 && FHIP_MM32) {
                // all 32 candidates in one pass (32x32x32 tiles); false: residuals many times full scale, the general way
                mm_done = mm_search32<C, T>(lds_raw + off[12], mm_cl, mm_tab, l.smp, reinterpret_cast<uint32_t *>(l.sums),
                                            reinterpret_cast<unsigned long long *>(lds_raw + off[14]), l.trial,
                                            &l.misc[26], n, max_order, e.obits, e.precision, tid, lg0.merged < 0);
            } else {
                mm_search<C, T>(lds_raw + off[12], mm_cl, mm_tab, l.smp, reinterpret_cast<uint32_t *>(l.sums),
                                reinterpret_cast<unsigned long long *>(lds_raw + off[14]), l.trial, n, max_order,
                                e.obits, e.precision, tid);
                mm_done = true;
            }
        }
    }

    // partition-order window over all candidates (rice.c:148-155): the highest order has the
    // tightest clamp, order 1 the loosest
    const int pmin_lo = clamp_porder(e.pmin_req, n, max_order);
    const int pmax_hi = clamp_porder(e.pmax_req, n, 1);

    // LOG walk (optimize.c:240-261), carried by every thread alike: the winner so far, the
    // orders evaluated, the step
    const bool is_log = (omethod == 6);
    int lg_best = min_order - 1 + (max_order - min_order) / 3;
    uint32_t lg_seen = 0;
    int lg_step = 16;
    static_assert(G >= 3, "a LOG step has up to three new candidates");

    for (int g0 = 0, round = 0;; g0 += G, round++) {
        int ng;
        uint32_t lg_pack = 0;             // the round's LOG candidates, five bits each, in visiting order
        int lg_merged = 0;                // ... and the steps of optimize.c:247 they belong to
        if (constant || mm_done) break;
        if (!is_log) {
            ng = min(G, nc - g0);
            if (ng <= 0) break;
        } else {
            // The next steps' orders not yet evaluated (last - step, last, last + step).  What a step
            // visits depends on the winner so far -- but at the start often not at all: from order index 3
            // of 1..12 the walk visits 3, then 11, then 7 whoever wins; log_plan_round0 (below, on the host)
            // merges such steps into round 0.  The replay behind every round then visits exactly the
            // reference's candidates in the reference's order.
            // Round 0 comes planned from the host (lg0).  Later rounds take ONE step -- whether further steps could
            // be merged depends on this round's winner, and in practice they cannot --: the orders
            // last - step, last, last + step not yet evaluated, after the steps that have none.  Scalar, ~30
            // instructions (round 3: the general merging planner cost every round 1.5-3 k cycles of its path).
            ng = 0;
            if (round == 0) {
                ng = lg0.ng; lg_pack = lg0.pack; lg_merged = lg0.merged;
            } else {
                int st = lg_step;
                while (st > 0 && ng == 0) {
                    for (int i = lg_best - st; i <= lg_best + st; i += st)
                        if (i >= min_order - 1 && i < max_order && !((lg_seen >> i) & 1u)) lg_pack |= (uint32_t)i << (5 * ng++);
                    lg_merged++;
                    st >>= 1;
                }
            }
            if (ng == 0) break;                            // every step consumed, nothing left to visit
        }
        const int par = round & 1;
        STAMP(2 + 6 * (round & 3));
        // ---- the round's rows: doubles, int16 pairs, sum |coef|, shift ----
        // (the LOG walk planned and replayed by wave 0 alone, with the rows staged by that wave, saved
        // a third of the kernel's scalar instructions and made it SLOWER -- level 8's search 264 -> 276 us:
        // the kernel waits on its per-round critical path, not on instruction issue)
        if (tid < G * 32) {
            const int g = tid >> 5, j = tid & 31;
            const int cand = (g >= ng) ? 0 : is_log ? (int)((lg_pack >> (5 * g)) & 31u) : l.list[g0 + g];
            const int ord = cand + 1;
            const int32_t cv = !(g < ng && j < ord) ? 0 : MM ? crow_base[cand * FHIP_MAX_ORDER + j]
                                                             : (int32_t)crows[cand * FHIP_MAX_ORDER + j];
            l.coefd[g * SRCH_CROW + j] = (double)cv;
            if (j < SRCH_CROW - 32) l.coefd[g * SRCH_CROW + 32 + j] = 0.0;
            const int32_t nb = __shfl_xor(cv, 1, WAVE);
            if (j < 16 && (j & 1) == 0) l.pairs[g * 8 + (j >> 1)] = (nb & 0xFFFF) | (cv << 16);
            int32_t sa = cv < 0 ? -cv : cv;
            sa += __shfl_xor(sa, 1, WAVE); sa += __shfl_xor(sa, 2, WAVE);
            sa += __shfl_xor(sa, 4, WAVE); sa += __shfl_xor(sa, 8, WAVE);
            sa += __shfl_xor(sa, 16, WAVE);
            if (j == 0) {
                l.rowi[g * 4 + 0] = cand;
                l.rowi[g * 4 + 1] = !(g < ng) ? 0 : MM ? srow[cand] : cshifts[cand];
                l.rowi[g * 4 + 2] = sa;
                // rice.c:148-155 for this order
                if constexpr (MM) l.rowi[g * 4 + 3] = clamp_porder(e.pmin_req, n, ord) | (clamp_porder(e.pmax_req, n, ord) << 8);
                else l.rowi[g * 4 + 3] = cshifts[32 + cand];
            }
        }
        if (tid < G * 12) l.lvl_bits[par * G * 12 + tid] = 0;
        if (tid < G) l.lvl_meth[par * G + tid] = 0;
        if (tid == 0) l.misc[24 + (par ^ 1)] = 0;
        __syncthreads();
        STAMP(3 + 6 * (round & 3));

        // The search behind the thread sums, two ways.  Leaf mode (T <= 512): the thread sums
        // -- the finest partition sums there are, T of them -- go to LDS as 32-bit leaves and
        // ONE WAVE PER CANDIDATE does rice.c:105-187 from them (wave_candidate_bits: no barrier,
        // no atomics, a third of the instructions of the pyramid + node pass below, which every
        // wave runs for every candidate).  A thread sum that leaves 32 bits (32-bit noise)
        // flags the round, which is then run again the general way: 64-bit pyramid in every
        // wave, one thread per (candidate, level, partition) node.
        bool leaf_mode = true;
        uint32_t *rleaf = reinterpret_cast<uint32_t *>(l.sums + G * 128);      // behind the G heaps
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
        // ---- per candidate: FIR, fold, thread sum, in-wave pyramid ----
#pragma unroll 1
        for (int g = 0; g < ng; g++) {
            const int cand = __builtin_amdgcn_readfirstlane(l.rowi[g * 4 + 0]);
            const int cshift = __builtin_amdgcn_readfirstlane(l.rowi[g * 4 + 1]);
            const uint32_t cabs = (uint32_t)__builtin_amdgcn_readfirstlane(l.rowi[g * 4 + 2]);
            const int ord = cand + 1;
            unsigned long long v = 0;
            unsigned long long vh1 = 0;      // LPT == 2: the second half of the run (a level-8 partition of its own)
            {
            int32_t r[C];
            FastCtx<C, T> eg = e;
            eg.l.coefd = l.coefd + g * SRCH_CROW;
#if defined(FHIP_SRCH_PROBE) && FHIP_SRCH_PROBE == 1      // timing probe: no FIR
#pragma unroll
            for (int o = 0; o < C; o++) r[o] = l.smp[tid * 4 + Img::off(o)] + cshift;
#else
            // packed 16-bit dot products: samples within int16 and a prediction that cannot leave int32
            if (ord <= 16 && magbits >= 0 && magbits <= 15 && ((unsigned long long)cabs << magbits) < (1ull << 31)) {
                if (ord <= 8) fir_lpc_dotn<C, T, 4>(eg, r, ord, cshift, l.pairs + g * 8);
                else fir_lpc_dotn<C, T, 8>(eg, r, ord, cshift, l.pairs + g * 8);
            } else {
                fir_lpc<C, T>(eg, r, ord, cshift);
            }
#endif
            // rice.c:120-123 fold; partition 0 of every level starts at `ord` (rice.c:85-94)
            constexpr int CH = (LPT == 2) ? C / 2 : C;       // samples of the first leaf
            if (e.i0 < ord) {
#pragma unroll
                for (int o = 0; o < C; o++) {
                    const uint32_t z = (e.i0 + o < ord) ? 0u : zigzag32(r[o]);
                    if (o < CH) v += z; else vh1 += z;
                }
            } else {
#pragma unroll
                for (int o = 0; o < C; o++) { if (o < CH) v += zigzag32(r[o]); else vh1 += zigzag32(r[o]); }
            }
            }
            if (leaf_mode) {
                if constexpr (LPT == 2) {
                    *reinterpret_cast<uint2 *>(rleaf + g * NL + 2 * tid) = make_uint2((uint32_t)v, (uint32_t)vh1);
                } else {
                    rleaf[g * NL + tid] = (uint32_t)v;
                }
                if (__any(((v | vh1) >> 32) != 0ull) && lane == 0) atomicOr(reinterpret_cast<uint32_t *>(&l.misc[24 + par]), 1u);
                continue;
            }
            const int pmm = __builtin_amdgcn_readfirstlane(l.rowi[g * 4 + 3]);
            const int pmin = pmm & 0xFF, pmax = pmm >> 8;
            unsigned long long *sums = l.sums + g * 512;
            if constexpr (LPT == 2) {
                // the half runs are the level-8 partitions (LT = 7); the thread sum starts the pyramid
                if (pmax >= 8 && pmin <= 8) { sums[255 + 2 * tid] = v; sums[255 + 2 * tid + 1] = vh1; }
                v += vh1;
            }
#define SPYR_STORE(S_, V_)                                                                  \
    do {                                                                                    \
        const int lev_ = LT - (S_);                                                         \
        if (lev_ <= pmax && lev_ >= pmin && lev_ <= 8 && (lane & ((1 << (S_)) - 1)) == 0)   \
            sums[(1 << lev_) - 1 + (tid >> (S_))] = (V_);                                   \
    } while (0)
            if (!__any((v >> 26) != 0ull)) {
                // every thread sum below 2^26: the wave's total fits 32 bits, one DPP add per step
                uint32_t w = (uint32_t)v;
                SPYR_STORE(0, (unsigned long long)w); w += dpp_u32<0x101>(w);
                SPYR_STORE(1, (unsigned long long)w); w += dpp_u32<0x102>(w);
                SPYR_STORE(2, (unsigned long long)w); w += dpp_u32<0x104>(w);
                SPYR_STORE(3, (unsigned long long)w); w += dpp_u32<0x108>(w);
                SPYR_STORE(4, (unsigned long long)w); w += (uint32_t)__shfl_down((int)w, 16, WAVE);
                SPYR_STORE(5, (unsigned long long)w); w += (uint32_t)__shfl_down((int)w, 32, WAVE);
                SPYR_STORE(6, (unsigned long long)w);
                v = w;
            } else {
                SPYR_STORE(0, v); v += row_shl_u64<1>(v);
                SPYR_STORE(1, v); v += row_shl_u64<2>(v);
                SPYR_STORE(2, v); v += row_shl_u64<4>(v);
                SPYR_STORE(3, v); v += row_shl_u64<8>(v);
                SPYR_STORE(4, v); v += __shfl_down(v, 16, WAVE);
                SPYR_STORE(5, v); v += __shfl_down(v, 32, WAVE);
                SPYR_STORE(6, v);
            }
#undef SPYR_STORE
            if (lane == 0) l.wtot[g * 16 + wv] = v;
        }
        STAMP(4 + 6 * (round & 3));
        __syncthreads();
        STAMP(5 + 6 * (round & 3));
        if (leaf_mode) {
            if (l.misc[24 + par] != 0) { leaf_mode = false; continue; }      // workgroup-uniform
            // one wave per candidate (two candidates per wave where the workgroup has only two waves)
            if constexpr (FHIP_CAND_X2 && NW == 2) {
                // 128 threads: two candidates a wave (wave_candidate_bits_x2; a half whose candidate does not exist repeats the
                // last one) -- with four waves the candidates of a round have a wave each already, and a LOG round of three
                // took 11 % longer by halves (level 8)
                for (int m0 = 2 * wv; m0 < ng; m0 += 2 * NW) {
                    const int m = m0 + (lane >> 5);
                    const int mc = min(m, ng - 1);
                    const int ord = l.rowi[mc * 4 + 0] + 1;
                    const int pmm = l.rowi[mc * 4 + 3];
                    const uint32_t b = wave_candidate_bits_x2<NL>(rleaf + mc * NL, l.sums + mc * 128, n, ord, pmm & 0xFF,
                                                                  pmm >> 8, e.obits, e.precision, lane);
                    if ((lane & 31) == 0 && m < ng) l.trial[ord - 1] = b;
                }
            } else {
                for (int m = wv; m < ng; m += NW) {
                    const int ord = l.rowi[m * 4 + 0] + 1;
                    const int pmm = l.rowi[m * 4 + 3];
                    const uint32_t b = wave_candidate_bits<NL>(rleaf + m * NL, l.sums + m * 128, n, ord, pmm & 0xFF,
                                                               pmm >> 8, e.obits, e.precision, lane);
                    if (lane == 0) l.trial[ord - 1] = b;
                }
            }
            STAMP(6 + 6 * (round & 3));
            break;
        }

        // ---- one thread per (candidate, level, partition) node ----
        {
            const int first = (1 << pmin_lo) - 1, last = (2 << pmax_hi) - 2;
#pragma unroll 1
            for (int g = 0; g < ng; g++) {
                const int ord = __builtin_amdgcn_readfirstlane(l.rowi[g * 4 + 0]) + 1;
                const int pmm = __builtin_amdgcn_readfirstlane(l.rowi[g * 4 + 3]);
                const int pmin = pmm & 0xFF, pmax = pmm >> 8;
                for (int q0 = first; q0 <= last; q0 += T) {
                    const int q = q0 + tid;
                    const int p = ilog2_dev((uint32_t)(q + 1));
                    const bool live = q <= last && p >= pmin && p <= pmax;
                    uint32_t b = 0;
                    int k = 0;
                    if (live) {
                        const int jn = q + 1 - (1 << p);
                        const int cnt = (n >> p) - (jn == 0 ? ord : 0);
                        unsigned long long sum;
                        if (p <= LT - 7) {
                            const int span = NW >> p;
                            sum = 0;
                            for (int w = 0; w < span; w++) sum += l.wtot[g * 16 + jn * span + w];
                        } else {
                            sum = l.sums[g * 512 + q];
                        }
                        k = (sum >> 32) ? rice_k_fast(sum, cnt, &b) : rice_k_fast_u32((uint32_t)sum, cnt, &b);
                    }
                    // a wave whose 64 nodes lie on one level (every wave past the first node
                    // row) adds them up in registers: one LDS atomic per wave instead of 64
                    const int p_first = __builtin_amdgcn_readfirstlane(p);
                    const int p_last = ilog2_dev((uint32_t)(q0 + (tid | 63) + 1));
                    if (p_first == p_last) {
                        uint32_t t = b;
                        t += dpp_u32<0x111>(t); t += dpp_u32<0x112>(t);
                        t += dpp_u32<0x114>(t); t += dpp_u32<0x118>(t);
                        t += dpp_u32<0x142, 0xA>(t); t += dpp_u32<0x143, 0xC>(t);
                        const bool rice2 = __any(live && k > 14);
                        if (lane == 63 && p_first >= pmin && p_first <= pmax) {
                            atomicAdd(&l.lvl_bits[(par * G + g) * 12 + p_first], t);
                            if (rice2) atomicOr(&l.lvl_meth[par * G + g], 1u << p_first);
                        }
                    } else if (live) {
                        atomicAdd(&l.lvl_bits[(par * G + g) * 12 + p], b);
                        if (k > 14) atomicOr(&l.lvl_meth[par * G + g], 1u << p);
                    }
                }
            }
        }
        __syncthreads();

        // ---- rice.c:127-138 and :157-171 per candidate ----
        if (tid < ng) {
            const int ord = l.rowi[tid * 4 + 0] + 1;
            const int pmin = l.rowi[tid * 4 + 3] & 0xFF, pmax = l.rowi[tid * 4 + 3] >> 8;
            const uint32_t lmask = l.lvl_meth[par * G + tid];
            uint32_t best = 0, method = 0;
            for (int p = pmin; p <= pmax; p++) {
                const uint32_t b = l.lvl_bits[(par * G + tid) * 12 + p] + 4u * (1u << p);
                if (p == pmin || b <= best) { best = b; method = (lmask >> p) & 1u; }
            }
            uint32_t bits = (uint32_t)(ord * e.obits + 2) + (uint32_t)(4 + 5 + ord * e.precision);
            bits += best;
            bits += method + 4u;
            l.trial[ord - 1] = bits;
        }
        break;
        }   // pass
        // The round's rows (l.rowi), leaves, heaps and level words are read until the last wave is
        // through -- in the general way too: its per-candidate block above reads l.rowi after the
        // node pass's barrier, and the next round's row staging would overwrite it (one barrier per
        // round; the general way is the rare fall-back).
        __syncthreads();
        STAMP(7 + 6 * (round & 3));
        if (is_log) {
            // optimize.c:249-259: the step's orders in ascending order against the winner so far.  The table
            // is read once (lane i holds bits[i]); its entries come by read-lane, not by LDS round trips.
            const int tr = (int)l.trial[lane & 31];
            // bits[] of the winner so far (optimize.c:243: UINT32_MAX until it has been evaluated itself)
            uint32_t cur = ((lg_seen >> lg_best) & 1u) ? (uint32_t)__builtin_amdgcn_readlane(tr, lg_best) : 0xFFFFFFFFu;
            for (int sidx = 0; sidx < lg_merged; sidx++) {
                const int last = lg_best;
                for (int i = last - lg_step; i <= last + lg_step; i += lg_step) {
                    if (i < min_order - 1 || i >= max_order || ((lg_seen >> i) & 1u)) continue;
                    lg_seen |= 1u << i;
                    const uint32_t bi = (uint32_t)__builtin_amdgcn_readlane(tr, i);
                    if (bi < cur) { lg_best = i; cur = bi; }
                }
                lg_step >>= 1;
            }
            lg_best = __builtin_amdgcn_readfirstlane(lg_best);      // keeps the walk in scalar registers
            lg_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)lg_seen);
        }
    }
    __syncthreads();

    STAMP(30);
    // ---- the method's walk over the table (optimize.c:201-261) ----
    if (tid == 0) {
        int best = 0;
        if (constant) {
            best = 0;
        } else if (omethod >= 2 && omethod <= 4) {
            const int levels = 1 << (omethod - 1);
            uint32_t best_bits = 0;
            best = max_order - 1;
            for (int i = levels - 1; i >= 0; i--) {
                int o = min_order + (((max_order - min_order + 1) * (i + 1)) / levels) - 2;
                if (o < 0) o = 0;
                const uint32_t b = l.trial[o];
                if (i == levels - 1) best_bits = b;
                else if (b < best_bits) { best_bits = b; best = o; }
            }
        } else if (omethod == 5) {
            uint32_t best_bits = l.trial[0];
            for (int i = 1; i < max_order; i++) {
                const uint32_t b = l.trial[i];
                if (b < best_bits) { best_bits = b; best = i; }
            }
        } else {
            best = lg_best;
        }
        l.misc[1] = best;
    }
    __syncthreads();
    // (stage entry fhip_order_search_bits: the table itself, 0xFFFFFFFF = order not visited)
    if (table_out && tid < 32) table_out[(size_t)s * 32 + tid] = constant ? 0xFFFFFFFFu : l.trial[tid];
    if (tid < 32) {
        // the winner as K2's compact row (kernels.h: FIN_STRIDE / FIN_DBL / FIN_PAIRS)
        const int best = l.misc[1];
        const int order = best + 1;
        int32_t *f = fin_all + (size_t)s * FIN_STRIDE;
        // (the row from LDS where it lies there: no global load at the end of the workgroup's path)
        const int32_t cv = !(tid < order) ? 0 : MM ? crow_base[best * FHIP_MAX_ORDER + tid]
                                                     : (int32_t)crows[best * FHIP_MAX_ORDER + tid];
        f[tid] = cv;
        int32_t sa = cv < 0 ? -cv : cv;
        sa += __shfl_xor(sa, 1, WAVE); sa += __shfl_xor(sa, 2, WAVE);
        sa += __shfl_xor(sa, 4, WAVE); sa += __shfl_xor(sa, 8, WAVE);
        sa += __shfl_xor(sa, 16, WAVE);
        const int32_t nb = __shfl_xor(cv, 1, WAVE);
        if (tid < 16) reinterpret_cast<double *>(f + FIN_DBL)[tid] = (double)cv;
        if (tid < 8 && (tid & 1) == 0) f[FIN_PAIRS + (tid >> 1)] = (nb & 0xFFFF) | (int32_t)((uint32_t)cv << 16);
        if (tid == 0) {
            f[32] = MM ? srow[best] : cshifts[best];
            f[33] = order;
            f[34] = sa;
            opt_all[s] = order;
        }
    }
    STAMP(31);
}

template <int C, int T, int G, bool MM = false>
__global__ __launch_bounds__(T, (T <= 256) ? ((MM && C >= 20) ? 2 : (C >= 20 || MM) ? FHIP_SRCH_WLONG : 4) : (T <= 512) ? 2 : 1)
void k_order_search(fhip_params P, int n, const int32_t *__restrict__ smp_all,
                    const int32_t *__restrict__ coefs_all, const int32_t *__restrict__ shift_all,
                    int32_t *__restrict__ opt_all, int32_t *__restrict__ fin_all,
                    const fhip_subframe_info *__restrict__ prep, int narrow_ok,
                    const int32_t *__restrict__ dev_sub, uint32_t *__restrict__ table_out, LogPlan0 lg0)
{
    if (dev_sub && (int)blockIdx.x >= dev_count(dev_sub, 0)) return;
    order_search_body<C, T, G, MM>(P, n, smp_all, coefs_all, shift_all, opt_all, fin_all, prep, narrow_ok, table_out, lg0,
                                   (int)blockIdx.x);
}

// Several bins of a ragged batch in ONE launch (kernels.h: MultiBin, units = subframes): bins whose geometries share the
// workgroup size T and the register class -- CSET 0: runs of 4 .. 16 samples, 1: runs of 20 .. 28 -- for the THINLY filled
// bins of a small batch (the long pieces: a few hundred subframes each), whose launches are latency one after another on
// the handle's three lanes (`profiles/r04_vbs_timeline.txt`).  Every bin of the launch runs with the largest one's LDS and
// registers: for well-filled bins a launch each is faster, and the matrix instances are not built this way.
template <int T, int G, int CSET>
__global__ __launch_bounds__(T, CSET ? FHIP_SRCH_WLONG : 4)
void k_order_search_bins(fhip_params P, MultiBin mb, const int32_t *__restrict__ smp,
                         const int32_t *__restrict__ coefs_all, const int32_t *__restrict__ shift_all,
                         int32_t *__restrict__ opt_all, int32_t *__restrict__ fin_all,
                         const fhip_subframe_info *__restrict__ prep, LogPlan0 lg0)
{
    const int blk = blockIdx.x;
    const int k = find_bin(mb, blk);
    const int local = blk - mb.wg0[k];
    if (local >= __builtin_amdgcn_readfirstlane(mb.cnt[mb.cnt_ix[k]])) return;
    const int n = mb.n[k];
    const int s = mb.unit0[k] + local;
    const int32_t *smp_k = smp + mb.smp_off[k] - (long long)mb.unit0[k] * n;          // row s: smp_k + s n
    const int nar = mb.narrow[k];
#define BODY_(CC) order_search_body<CC, T, G, false>(P, n, smp_k, coefs_all, shift_all, opt_all, fin_all, prep, nar, nullptr, lg0, s)
    const int c = n / T;
    if constexpr (T == 128) {
        if constexpr (CSET == 0) { if (c == 4) BODY_(4); else BODY_(12); }
        else { if (c == 20) BODY_(20); else BODY_(28); }
    } else {
        if constexpr (CSET == 0) { if (c == 4) BODY_(4); else if (c == 8) BODY_(8); else if (c == 12) BODY_(12); else BODY_(16); }
        else { if (c == 20) BODY_(20); else if (c == 24) BODY_(24); else BODY_(28); }
    }
#undef BODY_
}

}  // namespace

// The order-search kernel serves the LPC order searches (order methods 2..6) of block sizes that
// are 256 leaves of a whole number of sample pairs: the power-of-two blocks from 2048 up (runs
// of 8 or 16) and every piece the VBS splitter makes of a 4096 or 8192 block (k eighths: 256
// level-8 partitions of 2k or 4k samples; vbs.c:36-83).
// Round 0 of the LOG walk: the first step that has candidates and then every following step whose
// candidate set is the same for EVERY order that could be the winner by then (the winner on entry or any
// candidate of the round), while they fit G rows.
static LogPlan0 log_plan_round0(int min_order, int max_order, int G)
{
    LogPlan0 r{0u, 0, 0};
    uint32_t poss = 1u << (min_order - 1 + (max_order - min_order) / 3), seen = 0;
    for (int st = 16; st > 0; st >>= 1) {
        uint32_t set0 = 0;
        bool first = true, same = true;
        for (uint32_t pm = poss; pm; pm &= pm - 1) {
            const int b = __builtin_ctz(pm);
            uint32_t sb = 0;
            for (int i = b - st; i <= b + st; i += st)
                if (i >= min_order - 1 && i < max_order && !((seen >> i) & 1u)) sb |= 1u << i;
            if (first) { set0 = sb; first = false; }
            else if (sb != set0) same = false;
        }
        const int cnt = __builtin_popcount(set0);
        if (!same || r.ng + cnt > G) break;           // (never on the first step: one winner, <= 3 orders)
        for (uint32_t m = set0; m; m &= m - 1) r.pack |= (uint32_t)__builtin_ctz(m) << (5 * r.ng++);
        seen |= set0;
        poss |= set0;
        r.merged++;
    }
    return r;
}

static bool search_geometry(int n, int *C, int *T)
{
    if (n == 16384) { *C = 16; *T = 1024; return true; }
    if (n == 8192) { *C = 16; *T = 512; return true; }
    if (n % 1024 == 0 && n >= 1024 && n <= 7168) { *C = n / 256; *T = 256; return true; }    // runs of 4 .. 28
    if (n % 512 == 0 && n >= 512 && n <= 3584) { *C = n / 128; *T = 128; return true; }      // 512, 1536, 2560, 3584
    return false;
}

bool order_search_supported(const fhip_params &p, int n)
{
    static const bool off = getenv("FHIP_NO_ORDER_SEARCH") != nullptr;      // measurements only
    if (off) return false;
    if (p.prediction_type != 2 || n <= p.max_prediction_order || n < 5) return false;
    // (LOG, order method 6, walks step by step: a round per step of optimize.c:244-261, up to
    // three candidates each.  A table of all orders measured slower: 342 + 74 us at level 8.)
    static const bool no_log = getenv("FHIP_ORDER_SEARCH_NO_LOG") != nullptr;     // measurements only
    if (p.order_method < 2 || p.order_method > (no_log ? 5 : 6)) return false;
    int fc = 0, ft = 0;
    return search_geometry(n, &fc, &ft);
}

hipError_t launch_order_search(hipStream_t st, const fhip_params &p, const int32_t *smp, int nsub,
                               int n, const int32_t *coefs, const int32_t *shift,
                               int32_t *opt_order, int32_t *fin, const fhip_subframe_info *prep,
                               bool narrow_ok, const int32_t *dev_sub, uint32_t *table_out)
{
    if (nsub == 0) return hipSuccess;
    int fc = 0, ft = 0;
    if (!order_search_supported(p, n) || !search_geometry(n, &fc, &ft)) return hipErrorInvalidValue;
    constexpr int G = 4;
    size_t off[16];
    LogPlan0 lg0 = (p.order_method == 6) ? log_plan_round0(p.min_prediction_order, p.max_prediction_order, G) : LogPlan0{0u, 0, 0};
    // tests only: FHIP_MM32_FORCE_FALLBACK=1 makes mm_search32 report "residuals too large" for every subframe, so that
    // the general way runs behind a finished matrix pass (the path a signal has to be pathological to reach)
    static const bool force_fb = getenv("FHIP_MM32_FORCE_FALLBACK") != nullptr;
    if (force_fb && p.order_method == 5) lg0.merged = -1;
    // SEARCH on 4096-sample blocks: the FIRs on the int8 matrix pipe (samples beyond 24 bits, constant
    // subframes and the other methods take the vector way inside the same kernel)
    static const bool no_mm = getenv("FHIP_NO_MM") != nullptr;              // measurements only
    // (16-bit samples at orders <= 16 keep the packed dot products: SEARCH 1-12 at 16 bits 0.28 ms that
    // way, 0.43 on the matrix pipe, whose passes always cost 16 candidates x 32 taps)
    if (!no_mm && ft >= 256 && p.order_method == 5 && p.bits_per_sample <= 24 &&
        (p.bits_per_sample > 16 || p.max_prediction_order > 16)) {
#define LAUNCH_MM(CC, TT)                                                                    \
    do {                                                                                     \
        const size_t lds = srch_lds_layout<G>((size_t)SmpImg<CC, TT>::SIZE, off, TT, CC * TT, srch_crow(CC)); \
        hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_order_search<CC, TT, G, true>), \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (er != hipSuccess) return er;                                                     \
        hipLaunchKernelGGL((k_order_search<CC, TT, G, true>), dim3(nsub), dim3(TT), lds, st, p, n, smp, coefs, \
                           shift, opt_order, fin, prep, narrow_ok ? 1 : 0, dev_sub, table_out, lg0); \
        return hipGetLastError();                                                            \
    } while (0)
        switch (fc * 10000 + ft) {
        case 160256: LAUNCH_MM(16, 256);
        case 160512: LAUNCH_MM(16, 512);
        case 40256: LAUNCH_MM(4, 256);
        case 80256: LAUNCH_MM(8, 256);
        case 120256: LAUNCH_MM(12, 256);
        case 200256: LAUNCH_MM(20, 256);
        case 240256: LAUNCH_MM(24, 256);
        case 280256: LAUNCH_MM(28, 256);
        default: break;                                      // (n = 16384, the T = 128 pieces: the vector way)
        }
#undef LAUNCH_MM
    }
#define LAUNCH_SRCH(CC, TT)                                                                  \
    do {                                                                                     \
        const size_t lds = srch_lds_layout<G>((size_t)SmpImg<CC, TT>::SIZE, off, TT < 256 ? 256 : TT, 0, srch_crow(CC)); \
        hipError_t er = hipFuncSetAttribute(                                                 \
            reinterpret_cast<const void *>(&k_order_search<CC, TT, G>),                      \
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        if (er != hipSuccess) return er;                                                     \
        hipLaunchKernelGGL((k_order_search<CC, TT, G>), dim3(nsub), dim3(TT), lds, st, p, n, \
                           smp, coefs, shift, opt_order, fin, prep, narrow_ok ? 1 : 0, dev_sub, table_out, lg0); \
    } while (0)
    switch (fc * 10000 + ft) {
    case 160256: LAUNCH_SRCH(16, 256); break;
    case 160512: LAUNCH_SRCH(16, 512); break;
    case 161024: LAUNCH_SRCH(16, 1024); break;
    case 80256: LAUNCH_SRCH(8, 256); break;
    case 40256: LAUNCH_SRCH(4, 256); break;
    case 120256: LAUNCH_SRCH(12, 256); break;
    case 200256: LAUNCH_SRCH(20, 256); break;
    case 240256: LAUNCH_SRCH(24, 256); break;
    case 280256: LAUNCH_SRCH(28, 256); break;
    case 40128: LAUNCH_SRCH(4, 128); break;
    case 120128: LAUNCH_SRCH(12, 128); break;
    case 200128: LAUNCH_SRCH(20, 128); break;
    case 280128: LAUNCH_SRCH(28, 128); break;
    default: return hipErrorInvalidValue;
    }
#undef LAUNCH_SRCH
    return hipGetLastError();
}

static bool order_search_is_matrix(const fhip_params &p, int ft)
{
    // (launch_order_search's rule)
    static const bool no_mm = getenv("FHIP_NO_MM") != nullptr;              // measurements only
    return !no_mm && ft >= 256 && p.order_method == 5 && p.bits_per_sample <= 24 &&
           (p.bits_per_sample > 16 || p.max_prediction_order > 16);
}

static size_t smp_img_size(int fc, int ft)
{
#define SZ_(CC, TT) if (fc == CC && ft == TT) return (size_t)SmpImg<CC, TT>::SIZE
    SZ_(4, 128); SZ_(12, 128); SZ_(20, 128); SZ_(28, 128);
    SZ_(4, 256); SZ_(8, 256); SZ_(12, 256); SZ_(16, 256); SZ_(20, 256); SZ_(24, 256); SZ_(28, 256);
#undef SZ_
    return 0;
}

// the launch group of a bin of a ragged batch: bins of one group can share a k_order_search_bins launch; -1: the bin has
// its own launch (launch_order_search: the matrix instances, the 512-thread ones)
int order_search_group(const fhip_params &p, int n)
{
    int fc = 0, ft = 0;
    if (!order_search_supported(p, n) || !search_geometry(n, &fc, &ft) || ft > 256 || order_search_is_matrix(p, ft)) return -1;
    return ft | ((fc >= 20) ? 0x1000 : 0);
}

hipError_t launch_order_search_bins(hipStream_t st, const fhip_params &p, const MultiBin &mb, const int32_t *smp,
                                    const int32_t *coefs, const int32_t *shift, int32_t *opt_order, int32_t *fin,
                                    const fhip_subframe_info *prep)
{
    if (mb.nbins < 1) return hipSuccess;
    const int grid = mb.wg0[mb.nbins];
    if (grid == 0) return hipSuccess;
    const int key = order_search_group(p, mb.n[0]);
    if (key < 0) return hipErrorInvalidValue;
    constexpr int G = 4;
    size_t lds = 0;
    for (int k = 0; k < mb.nbins; k++) {
        if (order_search_group(p, mb.n[k]) != key) return hipErrorInvalidValue;
        int fc = 0, ft = 0;
        search_geometry(mb.n[k], &fc, &ft);
        size_t off[16];
        const size_t l = srch_lds_layout<G>(smp_img_size(fc, ft), off, ft < 256 ? 256 : ft, 0, srch_crow(fc));
        lds = l > lds ? l : lds;
    }
    const LogPlan0 lg0 = (p.order_method == 6) ? log_plan_round0(p.min_prediction_order, p.max_prediction_order, G) : LogPlan0{0u, 0, 0};
#define LAUNCH_BINS(TT, CS)                                                                  \
    do {                                                                                     \
        hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_order_search_bins<TT, G, CS>), \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (er != hipSuccess) return er;                                                     \
        hipLaunchKernelGGL((k_order_search_bins<TT, G, CS>), dim3(grid), dim3(TT), lds, st, p, mb, smp, coefs, \
                           shift, opt_order, fin, prep, lg0);                                \
        return hipGetLastError();                                                            \
    } while (0)
    switch (key) {
    case 128: LAUNCH_BINS(128, 0);
    case 128 | 0x1000: LAUNCH_BINS(128, 1);
    case 256: LAUNCH_BINS(256, 0);
    case 256 | 0x1000: LAUNCH_BINS(256, 1);
    default: break;
    }
#undef LAUNCH_BINS
    return hipErrorInvalidValue;
}

}  // namespace fhip
