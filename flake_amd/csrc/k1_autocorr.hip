// k1_autocorr.hip -- K1: apply_welch_window + compute_autocorr (lpc.c:28-71), with K2 as
// the tail of the wave-typed kernel.  -ffp-contract=off is load-bearing here: the fp64
// stages reproduce the reference's rounding sequence exactly (one rounding per multiply
// and per add, in the reference's order), which is what makes the quantised
// coefficients -- and with them every residual -- bit-exact.
#include "device_util.h"
#include "lpc_reg.h"

#ifdef FHIP_STAMPS
FHIP_DEFINE_STAMP_READER(fhip_debug_read_stamps_k1)      // slots 40..63 (tools/stamps_k1.py)
#endif

namespace fhip {
namespace {

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
                int nsub, int n, int maxlag, int G, int nl2, double c, const int32_t *__restrict__ dev_sub)
{
    __shared__ double s_buf[AC_WAVES][AC_GMAX * AC_STRIDE];
    nsub = dev_count(dev_sub, nsub);

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
                   int nsub, int n, int maxlag, int G, int lps, int ge, double c,
                   const int32_t *__restrict__ dev_sub)
{
    __shared__ double s_buf[AC_WAVES][PS_GMAX * PS_STRIDE];
    nsub = dev_count(dev_sub, nsub);

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
// consumers alone 41: they overlap to 56.  A second consumer wave per SIMD would not
// help: a SIMD serves its waves oldest first and the LDS return blocks the SIMD's
// vector issue, not just the reading wave's (ubench_walk, two waves per SIMD: 1.8x
// the wall time of one; the clock stays at 2.1-2.4 GHz, tools/fp64_clock.hip).
// Fatter groups were priced too: all even lags of order 8 in one wave (5 chains, one
// stream) 49.8 cycles per step, all odd ones (4 chains, two streams) 43.5 -- 93 cycles
// of SIMD time per 32 subframes and step against 32 + 3 x 25.7 = 109 for today's four
// groups, but on two SIMDs: it pays only with 64 subframes per CU, which the LDS does
// not hold at this tile size.
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
// lag).  An 8-byte read serves 32 lanes per LDS cycle; with an even stride (2 x 83
// doubles = 332 dwords = 12 mod 64) subframes sl and sl + 16 fall on the same bank
// pair: every b0 read is 2-way conflicted -- ALL of the kernel's bank-conflict cycles
// (SQ pass over the probe builds, profiles/r03_k1_lds_counters.txt: producers alone 0,
// consumers 1546 per wave, none without the b0 reads).  A stride that clears them is
// odd and breaks the `a` stream's 16-byte alignment; dropping EVERY b0 read (probe
// NOB, wrong results) shortens the launch by 5 us of 55, so the conflicts are worth
// 2-3 us at most and the layout stays.
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
struct wt_lpc_args { int precision, omethod; int32_t *coefs, *shift, *opt_order, *fin; int32_t *tile_ctr; };
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
                   const fhip_subframe_info *__restrict__ info, wt_lpc_args lpc, int narrow_ok,
                   const int32_t *__restrict__ dev_sub, MultiBin mb, wt_groups grp1, int lsplit)
{
    extern __shared__ __attribute__((aligned(16))) double wt_lds[];
    double *acbuf = wt_lds + WT_NBUF * WT_BUF;          // [32][FHIP_MAX_LAGS], LPCMO > 0 only

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    int blk = blockIdx.x;
    if (mb.nbins) {
        // every bin of a ragged batch in one launch (kernels.h: MultiBin): this workgroup's bin,
        // its block size and window constant, its rows and its range of the subframe-indexed arrays
        const int k = find_bin(mb, blk);
        blk -= mb.wg0[k];
        n = mb.n[k];
        c = mb.c[k];
        narrow_ok = mb.narrow[k];
        nsub = __builtin_amdgcn_readfirstlane(mb.cnt[mb.cnt_ix[k]]);
        const size_t u0 = (size_t)mb.unit0[k];
        smp += mb.smp_off[k];
        autoc += u0 * FHIP_MAX_LAGS;
        info += u0;
        lpc.coefs += u0 * FHIP_MAX_ORDER * FHIP_MAX_ORDER;
        lpc.shift += u0 * FHIP_MAX_ORDER;
        lpc.opt_order += u0;
        lpc.fin += u0 * FIN_STRIDE;
    } else {
        nsub = dev_count(dev_sub, nsub);
    }
    // lsplit = 2 (small batches: fewer workgroups than CUs): two workgroups share a tile of 32
    // subframes, one walks the even lags (grp), the other the odd ones (grp1) -- a wave then carries
    // one or two chains instead of three, and the walk, which is what a small batch waits for, is
    // that much shorter.  Both stage the same rows.  K2 is then the tail of whichever of the two finishes
    // second (round 4; a launch of its own before: 8.8 us of a 75 us step at 512 frames).
    if (lsplit == 2) {
        if (blk & 1) grp = grp1;
        blk >>= 1;
    }
    const int sub0 = blk * WT_SUB;
    if (sub0 >= nsub) return;                           // (a ragged batch's grid is its bin's capacity)
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
                    nar[r] = narrow_ok ? (__builtin_amdgcn_readfirstlane(info[sub].reserved) & 0xFF) : 0;
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
        for (int r = 0; alln && r < nrw; r++) alln = (info[min(sub0 + q0w + r, nsub - 1)].reserved & 0xFF) != 0;
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
    // (s_setprio for the walk -- consumers ahead of the staging wave on their SIMD -- changed nothing:
    // 55.6 vs 56.1 us, round 3)
    const int pi = lane >> 5, sl = lane & 31;
    // (selects, not grp.l0[wv]: a kernel argument indexed at run time is copied to scratch first)
    const int l0 = wv == 0 ? grp.l0[0] : wv == 1 ? grp.l0[1] : wv == 2 ? grp.l0[2] : grp.l0[3];       // wave-uniform
    const int nch = wv == 0 ? grp.nch[0] : wv == 1 ? grp.nch[1] : wv == 2 ? grp.nch[2] : grp.nch[3];
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
        bool tail_here = true;
        if (lsplit == 2) {
            // The tile's sums are complete when BOTH of its workgroups have stored theirs: an arrival counter per
            // tile (cdna_hip_programming.md, split-K recipe).  Every storing wave drains its stores, the workgroup
            // meets, one lane releases at agent scope and draws a ticket; the second arrival (odd ticket: the
            // counters are never reset, two arrivals per tile and launch) acquires and runs the tail on sums read
            // back from global memory.  Correct wherever the two workgroups were placed.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int last = 0;
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int ticket = __hip_atomic_fetch_add(&lpc.tile_ctr[blk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = ticket & 1;
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            // (only wave 0 runs the tail: its own lane 0 made the acquire, the wave's loads below follow it)
            tail_here = wv == 0 && __builtin_amdgcn_readfirstlane(last) != 0;
        } else {
            __syncthreads();                               // all lags of the 32 subframes are in acbuf
        }
        if (tail_here && wv == 0 && lane < WT_SUB && sub0 + lane < nsub) {
            double ac[LPCMO + 1];
            if (lsplit == 2) {
                const double *row = autoc + (size_t)(sub0 + lane) * FHIP_MAX_LAGS;
#pragma unroll
                for (int i = 0; i <= LPCMO; i++) ac[i] = (i <= maxlag) ? row[i] : 0.0;
            } else {
#pragma unroll
                for (int i = 0; i <= LPCMO; i++) ac[i] = (i <= maxlag) ? acbuf[lane * FHIP_MAX_LAGS + i] : 0.0;
            }
            lpc_reg_one<LPCMO>(ac, sub0 + lane, maxlag, lpc.precision, lpc.omethod, lpc.coefs, lpc.shift,
                               lpc.opt_order, lpc.fin);
        }
    }
}

}  // namespace

namespace {
// Which K1 kernel serves a batch: a measured time model in ns (MI355X; rounds =
// workgroup waves over the chip, step = one walk step):
//   wt : rounds x (n/2 x max(23, 4.8 NCH) + 8000)      32 subframes per workgroup, whole tiles only, K2 included
//   ps : rounds x (n/2 x 39 + 1000)                      Gp subframes per wave
//   cur: passes of 2048 waves x (n x 33..41 + 1000)      G subframes per wave
struct ac_choice { int kernel; int G, nl2, Gp, lps, ge, ne, no, split; };   // kernel: 0 cur, 1 ps, 2 wt; split: wt's lag split
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
    // k_autocorr keeps two waves per SIMD resident (LDS): 2048 walk at once, at 33 ns per
    // position alone on a SIMD and 41 when two share it (measured at n = 1536 .. 3584)
    const double t_cur = (double)((waves_cur + 2 * simds - 1) / (2 * simds)) *
                         (n * (waves_cur > simds ? 41.0 : 33.0) + 1000.0);
    const double t_ps = (ch.Gp >= 1) ? (double)(((nsub + ch.Gp - 1) / ch.Gp + simds - 1) / simds) * (0.5 * n * 39.0 + 1000.0) : 1e30;
    double t_wt = 1e30;
    ch.split = 1;
    if ((n % AC_TILE) == 0) {
        const int e0 = (ch.ne + 1) / 2;
        // re-measured after this round's changes: 23 ns per step up to three chains per
        // group, 8 us per launch for barriers, head and the K2 tail (n = 2560 .. 7168)
        const double per_step = (4.8 * e0 > 23.0) ? 4.8 * e0 : 23.0;
        const int tiles = (nsub + WT_SUB - 1) / WT_SUB;
        t_wt = (double)((tiles + 255) / 256) * (0.5 * n * per_step + 8000.0);
        // a small batch leaves CUs idle: split the lags over two workgroups per tile (round 3)
        static const bool no_split = getenv("FHIP_NO_LAG_SPLIT") != nullptr;      // measurements only
        if (!no_split && 2 * tiles <= 256 && max_order >= 2) {
            ch.split = 2;
            const int m = (ch.ne + 3) / 4;                 // chains of the largest group after the split
            const double ps2 = (4.8 * m > 14.0) ? 4.8 * m : 14.0;
            t_wt = 0.5 * n * ps2 + 8000.0;
        }
    }
    // only the wave-typed kernel runs K2 as its tail; the others pay its launch (~9 us)
    const double k2 = (max_order <= 12) ? 9000.0 : 0.0;
    ch.kernel = (t_wt <= t_ps + k2 && t_wt <= t_cur + k2) ? 2 : (t_ps < t_cur) ? 1 : 0;
    if (const char *force = getenv("FHIP_AC_KERNEL")) {       // "cur" / "ps" / "wt": measurements only
        if (force[0] == 'c') ch.kernel = 0;
        if (force[0] == 'p' && ch.Gp >= 1) ch.kernel = 1;
        if (force[0] == 'w' && (n % AC_TILE) == 0) ch.kernel = 2;
    }
    return ch;
}
}  // namespace

bool autocorr_is_wave_typed(int nsub, int n, int max_order)
{
    return pick_autocorr(nsub, n, max_order).kernel == 2;
}


// True when K1 will also run K2 (launch_autocorr with lpc outputs): the wave-typed
// kernel and a maximum order the register version of K2 covers.
bool autocorr_does_lpc(int nsub, int n, int max_order, bool have_tile_counters)
{
    static const bool off = getenv("FHIP_NO_LPC_TAIL") != nullptr;      // measurements only
    // K2 as the tail of a LAG-SPLIT launch (the second of a tile's two workgroups to arrive runs it; round 4) is
    // built and tested but off: measured on MI355X the step gains 1 us at 512 stereo frames (K1 44.9 + K2 8.7 ->
    // 50.6 us, step 0.0747 -> 0.0737 ms -- the K2 launch mostly ran in the shadow of K3's launch latency already)
    // and LOSES where the chip is full: 2048 frames 0.1048 -> 0.1079 ms, configs[3] at 512 frames 0.138 -> 0.144.
    static const bool split_tail = getenv("FHIP_SPLIT_TAIL") != nullptr;
    if (off || max_order > 12) return false;
    const ac_choice ch = pick_autocorr(nsub, n, max_order);
    // (a lag-split launch needs the per-tile arrival counters for its tail, else K2 runs as its own kernel)
    return ch.kernel == 2 && (ch.split == 1 || (have_tile_counters && split_tail));
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

bool autocorr_bins_supported(int max_order, const int *n, int nbins)
{
    static const bool off = getenv("FHIP_NO_MULTIBIN") != nullptr;      // measurements only
    if (off || nbins < 1 || nbins > 8 || max_order < 1 || max_order > FHIP_MAX_ORDER) return false;
    for (int k = 0; k < nbins; k++) if (n[k] < AC_TILE || (n[k] % AC_TILE) != 0 || n[k] <= max_order) return false;
    return true;
}

// K1 (wave-typed kernel, K2 as its tail where the order fits registers) for all bins of a ragged
// batch in one launch: the launch lasts as long as the longest chain walk, not the sum of eight.
hipError_t launch_autocorr_bins(hipStream_t st, const MultiBin &mb, const int32_t *smp, int max_order,
                                double *autoc, const fhip_subframe_info *info,
                                const autocorr_lpc_out *lpc_out)
{
    if (!autocorr_bins_supported(max_order, mb.n, mb.nbins)) return hipErrorInvalidValue;
    const int ne = max_order / 2 + 1, no = (max_order + 1) / 2;
    const int e0 = (ne + 1) / 2, e1 = ne - e0, o0 = (no + 1) / 2, o1 = no - o0;
    wt_groups gr;
    gr.l0[0] = 0;          gr.nch[0] = e0;
    gr.l0[1] = 2 * e0;     gr.nch[1] = e1;
    gr.l0[2] = 1;          gr.nch[2] = o0;
    gr.l0[3] = 1 + 2 * o0; gr.nch[3] = o1;
    const int nch = e0;
    const int blocks = mb.wg0[mb.nbins];
    if (blocks == 0) return hipSuccess;
    wt_lpc_args la{};
    int lpcmo = 0;
    if (lpc_out) {
        if (max_order > 12) return hipErrorInvalidValue;
        la.precision = lpc_out->precision; la.omethod = lpc_out->omethod;
        la.coefs = lpc_out->coefs; la.shift = lpc_out->shift; la.opt_order = lpc_out->opt_order;
        la.fin = lpc_out->fin;
        lpcmo = (max_order <= 8) ? 8 : 12;
    }
    const size_t lds = sizeof(double) * (size_t)WT_NBUF * WT_BUF +
                       (lpcmo ? sizeof(double) * (size_t)WT_SUB * FHIP_MAX_LAGS : 0);
#define LAUNCH_WTB(N_, L_)                                                                   \
    do {                                                                                     \
        hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_autocorr_wt<N_, false, L_>), \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (er != hipSuccess) return er;                                                     \
        hipLaunchKernelGGL((k_autocorr_wt<N_, false, L_>), dim3(blocks), dim3(8 * WAVE), lds, st, smp, \
                           autoc, 0, 0, max_order, gr, 0.0, (const int32_t *)nullptr, (int32_t *)nullptr, info, la, 0, \
                           (const int32_t *)nullptr, mb, wt_groups{}, 1);                    \
    } while (0)
    if (lpcmo == 8) {
        switch (nch) {
        case 1: LAUNCH_WTB(1, 8); break;
        case 2: LAUNCH_WTB(2, 8); break;
        case 3: LAUNCH_WTB(3, 8); break;
        default: return hipErrorInvalidValue;
        }
    } else if (lpcmo == 12) {
        switch (nch) {
        case 3: LAUNCH_WTB(3, 12); break;
        case 4: LAUNCH_WTB(4, 12); break;
        default: return hipErrorInvalidValue;
        }
    } else {
        switch (nch) {
        case 1: LAUNCH_WTB(1, 0); break;
        case 2: LAUNCH_WTB(2, 0); break;
        case 3: LAUNCH_WTB(3, 0); break;
        case 4: LAUNCH_WTB(4, 0); break;
        case 5: LAUNCH_WTB(5, 0); break;
        case 6: LAUNCH_WTB(6, 0); break;
        case 7: LAUNCH_WTB(7, 0); break;
        case 8: LAUNCH_WTB(8, 0); break;
        case 9: LAUNCH_WTB(9, 0); break;
        default: return hipErrorInvalidValue;
        }
    }
#undef LAUNCH_WTB
    return hipGetLastError();
}

hipError_t launch_autocorr(hipStream_t st, const int32_t *smp, int nsub, int n,
                           int max_order, double *autoc, const int32_t *pcm_fused,
                           int32_t *smp_out, const fhip_subframe_info *info,
                           const autocorr_lpc_out *lpc_out, bool narrow_ok, const int32_t *dev_sub,
                           int nsub_hint)
{
    if (nsub == 0) return hipSuccess;
    // the window constant is computed on the host exactly as lpc.c:34 does
    const double c = (2.0 / (n - 1.0)) - 1.0;
    // (nsub_hint: the count the kernel choice is made for when the real one is only known on the
    // device; the caller made its other choices -- K2 as the tail, 16-bit rows -- with the same)
    const ac_choice ch = pick_autocorr(nsub_hint > 0 ? nsub_hint : nsub, n, max_order);
    const int ne = ch.ne, no = ch.no, Gp = ch.Gp, lps = ch.lps, ge = ch.ge, nl2 = ch.nl2;
    int G = ch.G;
    const bool use_wt = ch.kernel == 2, use_ps = ch.kernel == 1;
    if ((pcm_fused || lpc_out || narrow_ok) && !use_wt) return hipErrorInvalidValue;
    if (narrow_ok && (!info || pcm_fused)) return hipErrorInvalidValue;
    const int e0 = (ne + 1) / 2, e1 = ne - e0, o0 = (no + 1) / 2, o1 = no - o0;
    if (use_wt) {
        wt_groups gr, gr1{};
        const int split = (pcm_fused || (lpc_out && !lpc_out->tile_ctr)) ? 1 : ch.split;
        int nch;
        if (split == 2) {
            // even lags over the four consumer waves of one workgroup, odd lags over those of its
            // partner: groups of consecutive same-parity lags, sizes differing by at most one,
            // the largest first
            auto spread = [](wt_groups &g, int first, int count) {
                int l = first;
                for (int w = 0; w < 4; w++) {
                    const int k = count / 4 + (w < count % 4 ? 1 : 0);
                    g.l0[w] = l; g.nch[w] = k;
                    l += 2 * k;
                }
            };
            spread(gr, 0, ne);
            spread(gr1, 1, no);
            nch = (ne + 3) / 4;                                // ne >= no
        } else {
            gr.l0[0] = 0;          gr.nch[0] = e0;
            gr.l0[1] = 2 * e0;     gr.nch[1] = e1;
            gr.l0[2] = 1;          gr.nch[2] = o0;
            gr.l0[3] = 1 + 2 * o0; gr.nch[3] = o1;
            nch = e0;                                          // e0 >= e1, o0, o1
        }
        const int blocks = split * ((nsub + WT_SUB - 1) / WT_SUB);
        const size_t lds = sizeof(double) * (size_t)WT_NBUF * WT_BUF;
        wt_lpc_args la{};
        int lpcmo = 0;
        if (lpc_out) {
            if (max_order > 12 || pcm_fused) return hipErrorInvalidValue;
            la.precision = lpc_out->precision; la.omethod = lpc_out->omethod;
            la.coefs = lpc_out->coefs; la.shift = lpc_out->shift; la.opt_order = lpc_out->opt_order;
            la.fin = lpc_out->fin; la.tile_ctr = lpc_out->tile_ctr;
            lpcmo = (max_order <= 8) ? 8 : 12;
        }
        const size_t lds_all = lds + (lpcmo ? sizeof(double) * (size_t)WT_SUB * FHIP_MAX_LAGS : 0);
#define LAUNCH_WT3(N_, F_, L_)                                                               \
    do {                                                                                     \
        hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_autocorr_wt<N_, F_, L_>), \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_all); \
        if (er != hipSuccess) return er;                                                     \
        hipLaunchKernelGGL((k_autocorr_wt<N_, F_, L_>), dim3(blocks), dim3(8 * WAVE), lds_all, st, smp, \
                           autoc, nsub, n, max_order, gr, c, pcm_fused, smp_out, info, la, narrow_ok ? 1 : 0, dev_sub, MultiBin{}, gr1, split); \
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
        if (lpcmo == 12) {                     // max_order 9..12: NCH 3 or 4 (2 in a lag-split launch)
            switch (nch) {
            case 2: LAUNCH_WT3(2, false, 12); break;
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
                           nsub, n, max_order, Gp, lps, ge, c, dev_sub);
        return hipGetLastError();
    }
    // spread over all CUs when the batch is small: fewer subframes per wave
    // cost nothing (a wave's time is its chain length, not its lane count)
    while (G > 1 && (nsub + G * AC_WAVES - 1) / (G * AC_WAVES) < 256) G--;
    if (const char *dbg = getenv("FHIP_AC_G")) { int v = atoi(dbg); if (v >= 1 && v <= AC_GMAX && v * nl2 <= WAVE) G = v; }
    const int per_block = G * AC_WAVES;
    const int blocks = (nsub + per_block - 1) / per_block;
    hipLaunchKernelGGL(k_autocorr, dim3(blocks), dim3(AC_WAVES * WAVE), 0, st, smp, autoc,
                       nsub, n, max_order, G, nl2, c, dev_sub);
    return hipGetLastError();
}

}  // namespace fhip
