// k3_encode.hip -- K3: encode_residual (optimize.c:34-276) incl. the Rice search
// (rice.c:30-187) and the residual section of output_residual (encode.c:766-798,
// bitio.h:120-141): k_encode<C> for any block size, k_encode_pow2<C,T,MODE> for
// n = C*T.
#include "device_util.h"
#include "k3_common.h"

#ifdef FHIP_STAMPS
FHIP_DEFINE_STAMP_READER(fhip_debug_read_stamps_k3)      // slots 0..39 (tools/stamps.py)
extern "C" int fhip_debug_read_stamps_k1(long long *out);
// the tools' entry point: K3's slots, with K1's accumulators in 40..63
extern "C" __attribute__((visibility("default"))) int fhip_debug_read_stamps(long long *out)
{
    long long k1[64];
    int rc = fhip_debug_read_stamps_k3(out);
    if (rc == 0) rc = fhip_debug_read_stamps_k1(k1);
    for (int i = 40; i < 64 && rc == 0; i++) out[i] = k1[i];
    return rc;
}
#endif

namespace fhip {
namespace {

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


// Rice search over the residuals held in r[] (rice.c:105-187).  Leaves the
// per-node parameters in l.kpar, the chosen order/method in l.misc and
// returns the subframe bit estimate.  All threads must call it.
// The search in three steps, so that a run too long for the registers can be fed
// piece by piece (k_encode_big): clear, add the finest-level sums of a piece, finish.
__device__ __forceinline__ void rice_zero(const EncCtx &e)
{
    const EncLds &l = e.l;
    for (int q = e.tid; q < 511; q += NT) l.sums[q] = 0;
    if (e.tid < 9) { l.lvl_bits[e.tid] = 0; l.lvl_meth[e.tid] = 0; }
    __syncthreads();
}

// rice.c:76-94 finest-level sums of the samples i0 .. i0+cnt-1 held in r[]: partition 0
// starts at `order`
template <int C>
__device__ __forceinline__ void rice_accumulate(const EncCtx &e, const int32_t (&r)[C], int order,
                                                int i0, int cnt)
{
    const EncLds &l = e.l;
    const int n = e.n;
    const int pmax = clamp_porder(e.pmax_req, n, order);
    const int psize = n >> pmax;
    const int heap0 = (1 << pmax) - 1;
    unsigned long long run = 0;
    int part = -1, bound = 0;
#pragma unroll
    for (int o = 0; o < C; o++) {
        const int i = i0 + o;
        if (o < cnt && i < n && i >= order) {
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

__device__ __forceinline__ uint32_t rice_finish(const EncCtx &e, int order, bool lpc);

template <int C>
__device__ __forceinline__ uint32_t rice_search(const EncCtx &e, const int32_t (&r)[C], int order, bool lpc)
{
    rice_zero(e);
    rice_accumulate<C>(e, r, order, e.i0, e.chunk);
    return rice_finish(e, order, lpc);
}

__device__ __forceinline__ uint32_t rice_finish(const EncCtx &e, int order, bool lpc)
{
    const EncLds &l = e.l;
    const int n = e.n, tid = e.tid;
    const int pmin = clamp_porder(e.pmin_req, n, order);
    const int pmax = clamp_porder(e.pmax_req, n, order);
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
              const fhip_subframe_info *__restrict__ prep,
              int32_t *__restrict__ res_out, uint8_t *__restrict__ bits_out, long long slot_bytes,
              int raw_order, int raw_lpc, const int32_t *__restrict__ dev_sub)
{
    if (dev_sub && (int)blockIdx.x >= dev_count(dev_sub, 0)) return;
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
    e.obits = prep[s].obits;
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
        out->obits = prep[s].obits;          // K0's fields: from the handle's prepare record
        out->wasted = prep[s].wasted;        // (the caller's when the stages run in one call)
        out->ch_mode = prep[s].ch_mode;
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
// K3 for long blocks  k_encode_big -- 16384 < n <= 65535
// ---------------------------------------------------------------------------
// Same contract and the same decision tree as k_encode, for blocks whose samples fit
// neither LDS nor a thread's registers: a thread's run (up to 256 samples) is walked
// in pieces of BIG_C, the samples are read from global memory (the subframe is at
// most 256 KB: L2), and residuals are formed again wherever they are needed -- once
// per candidate for the partition sums, once for the code lengths, once per emit
// window that the piece's bits touch.  Correctness first: this path serves block
// sizes outside FLAC's subset that the reference accepts (encode.c:288 ff.).
constexpr int BIG_C = 64;

// residuals of samples i0 .. i0+cnt-1 (optimize.c:34-122); LPC coefficients in l.coef
__device__ __forceinline__ void residual_big(const EncCtx &e, const int32_t *__restrict__ x,
                                             int32_t (&r)[BIG_C], int i0, int cnt, bool lpc, int order, int shift)
{
    const EncLds &l = e.l;
#pragma unroll 4
    for (int o = 0; o < BIG_C; o++) {
        const int i = i0 + o;
        int32_t v = 0;
        if (o < cnt && i < e.n) {
            const long long x0 = x[i];
            if (i < order || order == 0) {
                v = (int32_t)x0;
            } else if (lpc) {
                long long pred = 0;
                for (int j = order; j >= 1; j--) pred += (long long)l.coef[j - 1] * (long long)x[i - j];
                v = (int32_t)(x0 - (pred >> shift));
            } else {
                const long long x1 = x[i - 1];
                long long acc = x0 - x1;
                if (order >= 2) {
                    const long long x2 = x[i - 2];
                    acc = x0 - 2 * x1 + x2;
                    if (order >= 3) {
                        const long long x3 = x[i - 3];
                        acc = x0 - 3 * x1 + 3 * x2 - x3;
                        if (order >= 4) acc = x0 - 4 * x1 + 6 * x2 - 4 * x3 + (long long)x[i - 4];
                    }
                }
                v = (int32_t)acc;
            }
        }
        r[o] = v;
    }
}

__global__ __launch_bounds__(NT)
void k_encode_big(fhip_params P, int n, const int32_t *__restrict__ smp_all,
                  const int32_t *__restrict__ coefs_all, const int32_t *__restrict__ shift_all,
                  const int32_t *__restrict__ opt_all, fhip_subframe_info *__restrict__ info,
                  const fhip_subframe_info *__restrict__ prep,
                  int32_t *__restrict__ res_out, uint8_t *__restrict__ bits_out, long long slot_bytes,
                  int raw_order, int raw_lpc, const int32_t *__restrict__ dev_sub)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (dev_sub && (int)blockIdx.x >= dev_count(dev_sub, 0)) return;
    size_t off[10];
    enc_lds_layout(0, off);                              // no sample image
    EncCtx e;
    e.l.sums = reinterpret_cast<unsigned long long *>(lds_raw + off[0]);
    e.l.scan = reinterpret_cast<unsigned long long *>(lds_raw + off[1]);
    e.l.smp = nullptr;
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
    e.chunk = (n + NT - 1) / NT;                         // <= 256
    e.i0 = tid * e.chunk;
    e.obits = prep[s].obits;
    e.precision = P.lpc_precision;
    e.pmin_req = P.min_partition_order;
    e.pmax_req = P.max_partition_order;
    const int npieces = (e.chunk + BIG_C - 1) / BIG_C;   // <= 4

    const int32_t *src = smp_all + (size_t)s * n;
    if (tid == 0) l.misc[M_FLAG] = 0;
    __syncthreads();
    {
        const int32_t first = src[0];
        int differs = 0;
        for (int i = tid; i < n; i += NT) differs |= (src[i] != first);
        if (differs) atomicOr(&l.misc[M_FLAG], 1);
    }
    __syncthreads();
    const bool constant = (l.misc[M_FLAG] == 0);

    int32_t r[BIG_C];
    int type, type_code, order = 0, shift = 0;
    uint32_t est_bits = 0;
    bool has_rice = false, res_lpc = false;
    int res_order = 0;                                   // predictor of the residual that is kept
    const int32_t *crow_base = coefs_all + (size_t)s * FHIP_MAX_ORDER * FHIP_MAX_ORDER;
    const int32_t *srow = shift_all + (size_t)s * FHIP_MAX_ORDER;

    // one candidate: its partition sums piece by piece, then the search (rice.c:105-187)
    auto evaluate = [&](bool lpc, int pred_order, int rice_order, int sh) -> uint32_t {
        if (lpc) {
            __syncthreads();                             // previous readers of l.coef are done
            if (tid < pred_order) l.coef[tid] = crow_base[(pred_order - 1) * FHIP_MAX_ORDER + tid];
            __syncthreads();
        }
        rice_zero(e);
        for (int pc = 0; pc < npieces; pc++) {
            const int i0 = e.i0 + pc * BIG_C, cnt = min(BIG_C, e.chunk - pc * BIG_C);
            residual_big(e, src, r, i0, cnt, lpc, pred_order, sh);
            rice_accumulate<BIG_C>(e, r, rice_order, i0, cnt);
        }
        return rice_finish(e, rice_order, lpc);
    };

    enum { T_CONST, T_VERB, T_FIXED, T_LPC, T_RAW } tree;
    if (raw_order >= 0) tree = T_RAW;
    else if (constant) tree = T_CONST;                                   // optimize.c:143-151
    else if (n < 5 || P.prediction_type == 0) tree = T_VERB;             // optimize.c:153-158
    else if (P.prediction_type == 1 || n <= P.max_prediction_order) tree = T_FIXED;
    else tree = T_LPC;

    const int omethod = P.order_method;
    const int min_order = P.min_prediction_order;
    const int max_order = (tree == T_FIXED) ? min(P.max_prediction_order, 4) : P.max_prediction_order;

    int it = 0, best = 0;
    uint32_t best_bits = 0, last_bits = 0;
    bool have_best = false;
    int lg_step = 16, lg_last = 0, lg_pos = 3;
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
            lg_step = 32;
        }
    }

    if (tree == T_RAW) {
        // the input already is a residual: only calc_rice_params_* with that order
        rice_zero(e);
        for (int pc = 0; pc < npieces; pc++) {
            const int i0 = e.i0 + pc * BIG_C, cnt = min(BIG_C, e.chunk - pc * BIG_C);
            residual_big(e, src, r, i0, cnt, false, 0, 0);
            rice_accumulate<BIG_C>(e, r, raw_order, i0, cnt);
        }
        est_bits = rice_finish(e, raw_order, raw_lpc != 0);
        order = raw_order;
        type = raw_lpc ? FHIP_SUB_LPC : FHIP_SUB_FIXED;
        type_code = type;
        has_rice = true;
        res_order = 0; res_lpc = false;
    } else if (tree == T_CONST || tree == T_VERB) {
        type = type_code = (tree == T_CONST) ? FHIP_SUB_CONSTANT : FHIP_SUB_VERBATIM;
        est_bits = (uint32_t)(tree == T_CONST ? e.obits : e.obits * n);
    } else {
        for (;;) {
            int cand = -1;
            if (!final_pass) {
                if (tree == T_FIXED) {
                    if (it <= max_order) cand = it;
                } else if (omethod <= 4) {
                    if (it >= 0) {
                        const int levels = 1 << (omethod - 1);
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

            uint32_t b;
            if (tree == T_FIXED) b = evaluate(false, cand, cand, 0);
            else b = evaluate(true, cand + 1, cand + 1, srow[cand]);
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
            res_order = order; res_lpc = false;
        } else {
            order = best + 1;
            shift = srow[best];
            type = FHIP_SUB_LPC;
            type_code = FHIP_SUB_LPC | (order - 1);
            res_order = order; res_lpc = true;           // l.coef holds this row (last one staged)
        }
        has_rice = true;
    }

    const int porder = has_rice ? l.misc[M_PORDER] : 0;
    const int method = has_rice ? l.misc[M_METHOD] : 0;

    // FlacSubframe.residual (the samples themselves for CONSTANT / VERBATIM)
    if (res_out) {
        int32_t *dst = res_out + (size_t)s * n;
        for (int pc = 0; pc < npieces; pc++) {
            const int i0 = e.i0 + pc * BIG_C, cnt = min(BIG_C, e.chunk - pc * BIG_C);
            residual_big(e, src, r, i0, cnt, has_rice && res_lpc, has_rice ? res_order : 0, shift);
            for (int o = 0; o < BIG_C; o++)
                if (o < cnt && i0 + o < n) dst[i0 + o] = r[o];
        }
    }

    // encode.c:766-798 output_residual
    long long total_bits = 0;
    if (has_rice) {
        const int psz = n >> porder;
        const int pbits = 4 + method;
        const int heap0 = (1 << porder) - 1;
        unsigned long long piece_off[5];                 // bits in front of each piece of this run
        unsigned long long mine = 0;
        for (int pc = 0; pc < 4; pc++) {
            piece_off[pc] = mine;
            if (pc < npieces) {
                const int i0 = e.i0 + pc * BIG_C, cnt = min(BIG_C, e.chunk - pc * BIG_C);
                residual_big(e, src, r, i0, cnt, res_lpc, res_order, shift);
                for (int o = 0; o < BIG_C; o++) {
                    const int i = i0 + o;
                    if (o < cnt && i < n && i >= order) {
                        const int part = i / psz;
                        const int k = l.kpar[heap0 + part];
                        if (part > 0 && i == part * psz) mine += pbits;
                        mine += (unsigned long long)(emit_fold32(r[o]) >> k) + 1 + k;
                    }
                }
            }
        }
        piece_off[4] = mine;
        unsigned long long incl = wave_incl_scan_u64(mine, lane);
        if (lane == 63) l.scan[wv] = incl;
        __syncthreads();
        unsigned long long base = 6 + pbits;
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
                    const long long win_lo = wlo * 32, win_hi = (wlo + ENC_WWORDS) * 32;
                    for (int pc = 0; pc < npieces; pc++) {
                        // only the pieces whose bits reach into this window are formed again
                        const long long p_lo = (long long)(my_off + piece_off[pc]);
                        const long long p_hi = (long long)(my_off + piece_off[pc + 1]);
                        if (p_hi <= win_lo - 64 || p_lo >= win_hi) continue;
                        const int i0 = e.i0 + pc * BIG_C, cnt = min(BIG_C, e.chunk - pc * BIG_C);
                        residual_big(e, src, r, i0, cnt, res_lpc, res_order, shift);
                        long long pos = p_lo;
                        for (int o = 0; o < BIG_C; o++) {
                            const int i = i0 + o;
                            if (o < cnt && i < n && i >= order) {
                                const int part = i / psz;
                                const int k = l.kpar[heap0 + part];
                                if (part > 0 && i == part * psz) {
                                    put_bits(l.bits, wlo, pos, pbits, (uint32_t)k);
                                    pos += pbits;
                                }
                                const uint32_t u = emit_fold32(r[o]);
                                const uint32_t q = u >> k;
                                put_bits(l.bits, wlo, pos + q, k + 1, (1u << k) | (u & ((1u << k) - 1u)));
                                pos += (long long)q + 1 + k;
                            }
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
        out->obits = prep[s].obits;          // K0's fields: from the handle's prepare record
        out->wasted = prep[s].wasted;        // (the caller's when the stages run in one call)
        out->ch_mode = prep[s].ch_mode;
    }
    if (tid < FHIP_MAX_ORDER)
        out->coefs[tid] = (type == FHIP_SUB_LPC && tid < order && raw_order < 0) ? l.coef[tid] : 0;
    {
        const int np = has_rice ? (1 << porder) : 0;
        out->rparams[tid] = (tid < np) ? l.kpar[np - 1 + tid] : 0;
    }
    if (tid < FHIP_MAX_ORDER) {
        const int nw = (type == FHIP_SUB_CONSTANT) ? 1 : order;
        out->warmup[tid] = (tid < nw && tid < n) ? src[tid] : 0;
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

// rice.c:105-187 by ONE wave from the workgroup's thread sums as 32-bit leaves in LDS -- the order-search kernel's
// wave_candidate_bits with the outputs the emit needs: every node's parameter at its heap index in kpar[], the level
// chosen and its method.  A lane takes NL / 64 consecutive leaves and owns the nodes above them down to level 6 (four at
// level 8, two at 7, one at 6); levels 5 .. 0 are 63 nodes built through a heap in LDS and evaluated one per lane; level
// totals by a wave reduction / a wave scan; no barrier, no atomics (round 4: the workgroup-wide form below it took 4.7 k of a
// subframe's 14.6 k cycles -- a 64-bit pyramid stored level by level, a thread per node with an LDS atomic each, a
// serial choice over nine level words in every wave).
template <int NL>
__device__ __forceinline__ void wave_rice_leaves(const uint32_t *__restrict__ leaf, unsigned long long *__restrict__ heap,
                                                 int32_t *__restrict__ kpar, int n, int ord, int pmin, int pmax, int lane,
                                                 uint32_t *best_out, int *bp_out, uint32_t *method_out)
{
    constexpr int LPL = NL / 64;                // leaves per lane: 4 (256 leaves), 8
    constexpr int L8 = LPL / 4;                 // leaves per level-8 node
    static_assert(LPL >= 4 && LPL <= 16, "wave_rice_leaves: 256 .. 1024 leaves");
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
    // the usual case -- every sum below 0xFFE00000, no empty first partition -- in straight-line 32-bit code
    const bool corners = __any(s6 >= 0xFFE00000ull) || ((n >> pmax) - ord) <= 0;
    uint32_t lb[9];
#pragma unroll
    for (int p = 0; p < 9; p++) lb[p] = 0;
    uint32_t rice2 = 0;                         // bit p: some parameter of level p is above 14
    auto levels = [&](auto fast_c) {
        constexpr bool FAST = decltype(fast_c)::value;
        auto node = [&](unsigned long long sum, int p, int jn, uint32_t *b) -> int {
            const int cnt = (n >> p) - (jn == 0 ? ord : 0);
            int k;
            if constexpr (FAST) k = rice_k_u32_nb((uint32_t)sum, (uint32_t)cnt, b);
            else k = (sum >> 32) ? rice_k_fast(sum, cnt, b) : rice_k_fast_u32((uint32_t)sum, cnt, b);
            kpar[(1 << p) - 1 + jn] = k;
            return k;
        };
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
            if (pmax >= 6) {
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
        }
        if (pmin <= 5) {
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
                int k;
                if (FAST && total < 0xFFE00000ull) k = rice_k_u32_nb((uint32_t)hs, (uint32_t)cnt, &b);
                else k = (hs >> 32) ? rice_k_fast(hs, cnt, &b) : rice_k_fast_u32((uint32_t)hs, cnt, &b);
                kpar[lane] = k;
                big = k > 14;
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
        }
    };
    if (!corners) levels(std::true_type{});
    else levels(std::false_type{});
    // rice.c:127-138
    uint32_t best = 0, method = 0;
    int bp = pmin;
#pragma unroll
    for (int p = 0; p < 9; p++) {
        const uint32_t b = lb[p] + 4u * (1u << p);
        if (p >= pmin && p <= pmax && (p == pmin || b <= best)) { best = b; bp = p; method = (rice2 >> p) & 1u; }
    }
    *best_out = best;
    *bp_out = bp;
    *method_out = method;
}

// rice.c:105-187 on the residuals in r[]; all threads call it.
// ONE_WAVE: the search by one wave from the thread sums (wave_rice_leaves) -- the instances whose searches are large
// (MODE 3: orders 9 .. 16 and every order-search winner, partition orders up to 8; MODE 2) take it: configs[2]'s K3
// 112 -> 82 us, configs[3]'s 322 -> 297.  The lean instance (MODE 0, partition orders <= 5 at the headline: 63 nodes) measured
// the same either way (58.5 us) and keeps the workgroup-wide form, which fits its 96 registers without a spill.
template <int C, int T, bool ONE_WAVE = false>
__device__ __forceinline__ uint32_t rice_search_fast(const FastCtx<C, T> &e, const int32_t (&r)[C],
                                                     uint32_t (&u)[C], int order, bool lpc,
                                                     int *porder_out, int *method_out, uint32_t *umax_out)
{
    constexpr int LT = clog2(T);                      // the thread level
    const FastLds &l = e.l;
    const int n = e.n, tid = e.tid, lane = e.lane;
    const int pmin = clamp_porder(e.pmin_req, n, order);
    const int pmax = clamp_porder(e.pmax_req, n, order);

    fold_residuals<C, T>(e, r, u, order);
    // thread-level sum.  The residuals are not bounded by the sample width (a fixed
    // predictor of order 4 gains four bits, an LPC row with shift 0 more, and the
    // reference wraps at 32), so the narrow sum is chosen by the data: the largest
    // folded value of the run -- which the emit needs anyway -- decides per wave.
    uint32_t umax = 0;
#pragma unroll
    for (int o = 0; o < C; o++) umax = max(umax, u[o]);
    *umax_out = umax;
    unsigned long long v;
    if (!__any((umax >> (32 - clog2_up(C))) != 0u)) {
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

    if constexpr (ONE_WAVE && T >= 256 && T <= 512) {
        // One wave does the whole search from the thread sums (wave_rice_leaves); the others meet it at the second
        // barrier.  Thread sums beyond 32 bits (residuals of 32-bit noise) keep the workgroup-wide form below.
        uint32_t *leaf = reinterpret_cast<uint32_t *>(l.sums);          // [T] (2 KB at most), the heap behind them
        leaf[tid] = (uint32_t)v;
        const int wide = __any((v >> 32) != 0ull) ? 1 : 0;
        if (lane == 0) l.wtot[e.wv] = (unsigned long long)wide;
        __syncthreads();
        int any_wide = 0;
#pragma unroll
        for (int w = 0; w < T / WAVE; w++) any_wide |= (int)l.wtot[w];
        if (__builtin_amdgcn_readfirstlane(any_wide) == 0) {
            if (e.wv == 0) {
                uint32_t best;
                int bp;
                uint32_t method;
                wave_rice_leaves<T>(leaf, l.sums + 256, l.kpar, n, order, pmin, pmax, lane, &best, &bp, &method);
                if (lane == 0) { l.lvl_bits[0] = best; l.lvl_bits[1] = (uint32_t)bp; l.lvl_bits[2] = method; }
            }
            __syncthreads();
            const uint32_t best = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.lvl_bits[0]);
            const int bp = __builtin_amdgcn_readfirstlane((int)l.lvl_bits[1]);
            const uint32_t method = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.lvl_bits[2]);
            uint32_t bits = (uint32_t)(order * e.obits + 2);            // rice.c:157-171
            if (lpc) bits += (uint32_t)(4 + 5 + order * e.precision);
            bits += best;
            bits += method + 4u;
            *porder_out = bp;
            *method_out = (int)method;
            return bits;
        }
        __syncthreads();             // (leaf[] and the wave flags are read; the pyramid below stores over them)
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
// Round 4: ONE WAVE PER ORDER behind the thread sums -- the form the order-search kernel's wave_candidate_bits has.  The
// stamps of workgroup 0 (tools/stamps.py level2 / fixed) put rounds 2-3's version at 9.8-10.8 k cycles of the 18.7 k (n = 1152)
// / 25.6 k (mono n = 4096) a subframe took: five 64-bit pyramids stored level by level, one thread per (order, level,
// partition) node with an integer division to find its order, 64-bit parameter searches, LDS atomics per node, a serial
// selection over 5 x 6 level words.  Here the waves reduce their thread sums to the 32 sums of level 5 (one to five DPP
// steps, in 32 bits where a wave's total fits), one barrier, and wave k - min_order does rice.c:105-187 for order k alone:
// levels 4 .. 0 by a register pyramid and one trip through its heap, node q on lane q, level totals by a wave scan, the level
// choice on scalars; a second barrier and every thread compares the five results.
template <int C, int T>
__device__ __forceinline__ int fixed_search5(const FastCtx<C, T> &e, int min_order, int max_order,
                                             uint32_t *bits_out, int *porder_out, int *method_out)
{
    using Img = SmpImg<C, T>;
    constexpr int LT = clog2(T);
    constexpr int NW = T / WAVE;
    static_assert(T >= 64, "fixed_search5: whole waves");
    const bool narrow = e.obits + 4 + clog2_up(C) + 6 <= 32;      // a wave's total fits 32 bits
    const bool sum32 = e.obits + 4 + clog2_up(C) <= 32;           // a thread's sum does (a folded value is below 2^(obits+4))
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
    uint32_t p1 = h[0] - h[1];
    uint32_t p2 = h[0] - 2u * h[1] + h[2];
    uint32_t p3 = h[0] - 3u * h[1] + 3u * h[2] - h[3];
    uint32_t p0 = h[0];
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
    // ---- the 32 sums of level 5: T / 32 threads each ----
    unsigned long long *s5 = l.sums;                                        // [5][32]
    unsigned long long *heaps = l.sums + 160;                               // [5][64] behind them
    constexpr int ST = LT - 5;                                              // log2(threads per level-5 node): 1 .. 5
    const bool writer = (lane & ((1 << ST) - 1)) == 0;
    if (narrow) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            uint32_t v = a32[k];
            if (ST >= 1) v += dpp_u32<0x101>(v);
            if (ST >= 2) v += dpp_u32<0x102>(v);
            if (ST >= 3) v += dpp_u32<0x104>(v);
            if (ST >= 4) v += dpp_u32<0x108>(v);
            if (ST >= 5) v += (uint32_t)__shfl_down((int)v, 16, WAVE);
            if (writer) s5[k * 32 + (tid >> ST)] = v;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            unsigned long long v = sum32 ? (unsigned long long)a32[k] : a64[k];
            if (ST >= 1) v += row_shl_u64<1>(v);
            if (ST >= 2) v += row_shl_u64<2>(v);
            if (ST >= 3) v += row_shl_u64<4>(v);
            if (ST >= 4) v += row_shl_u64<8>(v);
            if (ST >= 5) v += __shfl_down(v, 16, WAVE);
            if (writer) s5[k * 32 + (tid >> ST)] = v;
        }
    }
    __syncthreads();
    // ---- rice.c:105-187 per order ----
    // Partition orders 0 .. 3 (levels 0-2's own setting): an order's 15 nodes fit 16 lanes, so FOUR orders share a wave
    // (a DPP row each) and the fifth takes the next wave, all at once -- a wave per order ran two (256 threads) or three
    // (128 threads) orders deep in wave 0.  Everything below is the wave-per-order code with lane -> (row, local lane) and
    // compile-time widths; the level choice is made per row by its first lanes, not on scalars.
    const int pmax_all = clamp_porder(e.pmax_req, n, min_order);             // order min_order has the loosest clamp
    if (pmax_all <= 3) {
        const int row = lane >> 4, ll = lane & 15;
        for (int slot0 = e.wv * 4; min_order + slot0 <= max_order; slot0 += NW * 4) {
            const int k = min_order + slot0 + row;
            const bool valid = k <= max_order;
            const int kc = min(k, max_order);
            unsigned long long *heap = heaps + (kc - min_order) * 64;
            // the 8 sums of level 3: four level-5 sums each
            unsigned long long v = 0;
            if (valid && ll < 8) v = s5[kc * 32 + 4 * ll] + s5[kc * 32 + 4 * ll + 1] + s5[kc * 32 + 4 * ll + 2] + s5[kc * 32 + 4 * ll + 3];
            const bool leaf = valid && ll < 8;
            if (leaf) heap[7 + ll] = v;
            v += row_shl_u64<1>(v); if (leaf && (ll & 1) == 0) heap[3 + (ll >> 1)] = v;
            v += row_shl_u64<2>(v); if (leaf && (ll & 3) == 0) heap[1 + (ll >> 2)] = v;
            v += row_shl_u64<4>(v); if (leaf && ll == 0) heap[0] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // pmin / pmax of this row's order: rice.c:148-155 without the division (orders 0 .. 4)
            const int L = ilog2_dev((uint32_t)n);
            const int lq = (kc == 0) ? 31 : (kc == 1) ? L : (kc == 2) ? L - 1 : (kc == 3) ? ((n >= (3 << (L - 1))) ? L - 1 : L - 2) : L - 2;
            const int lim = min(ilog2_dev((uint32_t)(n ^ (n - 1))), lq);
            const int pmin = min(e.pmin_req, lim), pmax = min(e.pmax_req, lim);
            const int p = ilog2_dev((uint32_t)(ll + 1));             // local lanes 0 .. 14: node ll, level p
            uint32_t b = 0;
            int kk = 0;
            const unsigned long long hs = heap[min(ll, 14)];
            const unsigned long long total = heap[0];
            if (valid && ll < 15 && p >= pmin && p <= pmax) {
                const int jn = ll + 1 - (1 << p);
                const int cnt = (n >> p) - (jn == 0 ? kc : 0);
                if (total < 0xFFE00000ull && cnt > 0) kk = rice_k_u32_nb((uint32_t)hs, (uint32_t)cnt, &b);
                else kk = (hs >> 32) ? rice_k_fast(hs, cnt, &b) : rice_k_fast_u32((uint32_t)hs, cnt, &b);
            }
            if (valid) l.kpar[64 + kc * 64 + ll] = kk;                // every node's parameter (the winner's move below)
            // level totals inside the row: an inclusive row scan, then level q = sc[2^(q+1) - 2] - sc[2^q - 2]
            uint32_t sc = b;
            sc += dpp_u32<0x111>(sc); sc += dpp_u32<0x112>(sc); sc += dpp_u32<0x114>(sc); sc += dpp_u32<0x118>(sc);
            uint32_t big = kk > 14 ? 1u : 0u;                         // ... and "some parameter above 14" per level, the same way
            uint32_t bs = big;
            bs += dpp_u32<0x111>(bs); bs += dpp_u32<0x112>(bs); bs += dpp_u32<0x114>(bs); bs += dpp_u32<0x118>(bs);
            const int base = lane & ~15;
            uint32_t best = 0, method = 0;
            int bp = pmin;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t hi = (uint32_t)__shfl((int)sc, base + (2 << q) - 2, WAVE);
                const uint32_t lo = q ? (uint32_t)__shfl((int)sc, base + (1 << q) - 2, WAVE) : 0u;
                const uint32_t bh = (uint32_t)__shfl((int)bs, base + (2 << q) - 2, WAVE);
                const uint32_t bl = q ? (uint32_t)__shfl((int)bs, base + (1 << q) - 2, WAVE) : 0u;
                const uint32_t bb = hi - lo + 4u * (1u << q);
                if (q >= pmin && q <= pmax && (q == pmin || bb <= best)) { best = bb; bp = q; method = (bh != bl) ? 1u : 0u; }
            }
            if (valid && ll == 0) {
                l.kpar[4 * kc + 0] = (int32_t)((uint32_t)(kc * e.obits + 2) + best + method + 4u);      // rice.c:157-171
                l.kpar[4 * kc + 1] = bp;
                l.kpar[4 * kc + 2] = (int32_t)method;
            }
        }
    } else
    for (int k = min_order + e.wv; k <= max_order; k += NW) {
        const int pmin = clamp_porder(e.pmin_req, n, k);
        const int pmax = clamp_porder(e.pmax_req, n, k);
        unsigned long long *heap = heaps + (k - min_order) * 64;
        unsigned long long v = (lane < 32) ? s5[k * 32 + lane] : 0ull;
#define HEAP_STORE(S_) do { if (lane < 32 && (lane & ((1 << (S_)) - 1)) == 0) heap[(1 << (5 - (S_))) - 1 + (lane >> (S_))] = v; } while (0)
        HEAP_STORE(0);
        v += row_shl_u64<1>(v); HEAP_STORE(1);
        v += row_shl_u64<2>(v); HEAP_STORE(2);
        v += row_shl_u64<4>(v); HEAP_STORE(3);
        v += row_shl_u64<8>(v); HEAP_STORE(4);
        v += __shfl_down(v, 16, WAVE); HEAP_STORE(5);
#undef HEAP_STORE
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int p = ilog2_dev((uint32_t)(lane + 1));           // lanes 0..62: node `lane`, level p
        uint32_t b = 0;
        int kk = 0;
        const unsigned long long hs = heap[min(lane, 62)];
        const unsigned long long total = heap[0];
        if (lane < 63 && p >= pmin && p <= pmax) {
            const int jn = lane + 1 - (1 << p);
            const int cnt = (n >> p) - (jn == 0 ? k : 0);
            if (total < 0xFFE00000ull && cnt > 0) kk = rice_k_u32_nb((uint32_t)hs, (uint32_t)cnt, &b);
            else kk = (hs >> 32) ? rice_k_fast(hs, cnt, &b) : rice_k_fast_u32((uint32_t)hs, cnt, &b);
        }
        if (lane < 64) l.kpar[64 + k * 64 + lane] = kk;           // every node's parameter (the winner's move below)
        const uint32_t sc = wave_incl_scan_u32_dpp(b);
        const unsigned long long bigm = __ballot(kk > 14);
        uint32_t best = 0, method = 0;
        int bp = pmin;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)sc, (2 << q) - 2);
            const uint32_t lo = q ? (uint32_t)__builtin_amdgcn_readlane((int)sc, (1 << q) - 2) : 0u;
            const unsigned long long lvl = ((1ull << ((2 << q) - 1)) - 1) & ~((1ull << ((1 << q) - 1)) - 1);
            const uint32_t bb = hi - lo + 4u * (1u << q);
            if (q >= pmin && q <= pmax && (q == pmin || bb <= best)) { best = bb; bp = q; method = (bigm & lvl) ? 1u : 0u; }
        }
        if (lane == 0) {
            l.kpar[4 * k + 0] = (int32_t)((uint32_t)(k * e.obits + 2) + best + method + 4u);      // rice.c:157-171
            l.kpar[4 * k + 1] = bp;
            l.kpar[4 * k + 2] = (int32_t)method;
        }
    }
    __syncthreads();
    // optimize.c:171-180: first strict minimum from min_order upward
    int best = min_order;
    uint32_t best_bits = (uint32_t)__builtin_amdgcn_readfirstlane(l.kpar[4 * min_order]);
    for (int k = min_order + 1; k <= max_order; k++) {
        const uint32_t bits = (uint32_t)__builtin_amdgcn_readfirstlane(l.kpar[4 * k]);
        if (bits < best_bits) { best_bits = bits; best = k; }
    }
    const int best_p = __builtin_amdgcn_readfirstlane(l.kpar[4 * best + 1]);
    const int best_m = __builtin_amdgcn_readfirstlane(l.kpar[4 * best + 2]);
    // the winner's parameters to where the emit and the info record read them
    __syncthreads();                                 // result words (l.kpar[0..19]) fully read
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
// Waves per SIMD (the second launch bound; VGPR cap 4 -> 128, 5 -> 96).  The kernel is bound by
// vector-ALU issue (PMC: ~850 VALU instructions per wave, > 80 % of the issue slots) with a
// barrier after every phase, and a fifth workgroup per CU fills what the barriers leave idle:
// the lean 256-thread instance with runs of 16 (MODE 0, the headline) fits 96 VGPRs without
// scratch (the compiler takes 106 when allowed 128) and its LDS (31 KB at n = 4096) fits five
// times: configs[1] 63.1 -> 57.1 us (round 2; round 1 had backed off to four at C >= 14 when the
// packed FIR first spilled).  MODE 3 / MODE 1 at (16, 256) do spill a little at five (6 / 1
// registers, 20 / 8 bytes of scratch: profiles/r03_kernel_resources.txt) and are still faster there
// (configs[3] K3 342 -> 320 us, round 2).
// Four where five would spill (the order-search instance MODE 2; runs of 18) or where the LDS
// of a 512- / 1024-thread workgroup stops at four waves per SIMD anyway.
// Geometry for n = 4096, measured: (C,T) = (16,256) 94 us, (8,512) 137, (4,1024)
// 256, (32,128) 115 (206 VGPRs): cross-wave phases grow with T, serial ones with C.
// Subframe s (an index into the subframe-indexed workspaces; its samples at smp_all + s n, its section at bits_out +
// s slot_bytes): the body of k_encode_pow2 (one geometry: a batch, or one bin of a ragged batch) and of k_encode_bins
// (several thinly filled bins of a ragged batch in one launch).
__device__ __forceinline__
void encode_pow2_body(const fhip_params &P, const int n, const int32_t *__restrict__ smp_all,
                      const int32_t *__restrict__ coefs_all, const int32_t *__restrict__ shift_all,
                      const int32_t *__restrict__ opt_all, const int32_t *__restrict__ fin_all,
                      fhip_subframe_info *__restrict__ info, const fhip_subframe_info *__restrict__ prep,
                      int32_t *__restrict__ res_out, uint8_t *__restrict__ bits_out, const long long slot_bytes,
                      const int narrow_ok, const int s)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    size_t off[12];
    fast_lds_layout(n, SmpImg<C, T>::SIZE, off, fast_wide_window(MODE, P.bits_per_sample));
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
    e.n = n; e.tid = tid; e.lane = tid & 63;
    e.wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: a scalar, not a vector register
    e.i0 = tid * C;
    e.precision = P.lpc_precision;
    e.pmin_req = P.min_partition_order;
    e.pmax_req = P.max_partition_order;

    // MAX / EST: the one row the reference quantises is known before the
    // search starts and comes compact from K2
    constexpr bool MULTI = MODE != 0 && MODE != 3;      // MODE 3: MODE 0 with the packed FIR for orders 9..16
    constexpr bool HAS_LPC = MODE != 1;
    constexpr bool pre_row = !MULTI;     // launcher: prediction_type == 2, n > max order, order method <= 1

    // One workgroup per subframe (a persistent variant that prefetched the next
    // subframe into registers measured slower: the hardware's own dispatch of a
    // fresh workgroup per subframe balances better and costs no VGPRs).
    int32_t xn[C];
    int32_t first_n, obits_n, fcoef_n = 0, fshift_n = 0, forder_n = 0, fcabs_n = 0, magbits_n = -1;
    {
        const int32_t *srcp = smp_all + (size_t)s * n;
        // K0 may have stored this row as int16 (info.reserved, honoured only when the
        // launcher says the flag is K0's): half the loads, one sign extension per sample
        const int nflag = (C % 8 == 0 && narrow_ok) ? (prep[s].reserved & 0xFF) : 0;     // 0, or 1 + bit length of max |x|
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
        obits_n = prep[s].obits;
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
    const int wwords = fast_window_words(n, fast_wide_window(MODE, P.bits_per_sample));
    if (bits_out) for (int q = tid; q < wwords / 4; q += T) reinterpret_cast<uint4 *>(l.bits)[q] = make_uint4(0, 0, 0, 0);
    if (tid < 16) l.coefd[32 + tid] = 0.0;
    if (pre_row && tid < FHIP_MAX_ORDER) {
        l.coef[tid] = fcoef_n;
        l.coefd[tid] = (double)fcoef_n;
    }
    if constexpr (MODE == 3 && C % 8 == 0) {
        // orders 9..16 on 16-bit rows: the taps as eight int16 pairs for the packed FIR (K2's
        // compact row carries the first four; l.trial is idle in this instance)
        const int32_t other = __shfl_xor(fcoef_n, 1, WAVE);
        if (tid < 16 && (tid & 1) == 0) l.trial[tid >> 1] = (uint32_t)((other & 0xFFFF) | (fcoef_n << 16));
    }
    // one barrier (the library's __syncthreads_or is three): per-wave flags, then everyone ORs them
    const int wave_differs = (__ballot(differs) != 0ull) ? 1 : 0;       // all lanes vote
    if (e.lane == 0) l.misc[e.wv] = wave_differs;
    __syncthreads();
    int any_differs = 0;
#pragma unroll
    for (int w = 0; w < T / WAVE; w++) any_differs |= l.misc[w];
    const bool constant = (__builtin_amdgcn_readfirstlane(any_differs) == 0);
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
    uint32_t umax_run = 0;               // largest folded value of this thread's run in that search
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
                STAMP(2);
                best = fixed_search5<C, T>(e, min_order, max_order, &est_bits, &porder, &method);
                STAMP(3);
                fir_fixed<C, T>(e, r, best);
                fold_residuals<C, T>(e, r, u, best);
                umax_run = 0;
#pragma unroll
                for (int o = 0; o < C; o++) umax_run = max(umax_run, u[o]);
                STAMP(8);
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
                b = rice_search_fast<C, T>(e, r, u, cand, false, &porder, &method, &umax_run);
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
                        if constexpr (C % 8 == 0) {
                            // what the packed FIR wants of a candidate row (orders <= 16): the taps as
                            // int16 pairs (lo: tap 2j+2, hi: tap 2j+1) and the sum of their magnitudes
                            const int32_t nb = __shfl_xor(cv, 1, WAVE);            // the pair's other tap
                            if (tid < 16 && (tid & 1) == 0) l.misc[tid >> 1] = (nb & 0xFFFF) | (cv << 16);
                            int32_t sa = cv < 0 ? -cv : cv;
                            sa += __shfl_xor(sa, 1, WAVE); sa += __shfl_xor(sa, 2, WAVE);
                            sa += __shfl_xor(sa, 4, WAVE); sa += __shfl_xor(sa, 8, WAVE);
                            sa += __shfl_xor(sa, 16, WAVE);
                            if (tid == 0) l.misc[8] = sa;
                        }
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
                    if constexpr (MODE == 3) {
                        if (!done && ord <= 16 && magbits_n >= 0 &&
                            ((unsigned long long)(uint32_t)fcabs_n << magbits_n) < (1ull << 31)) {
                            fir_lpc_dotn<C, T, 8>(e, r, ord, cshift, reinterpret_cast<const int32_t *>(l.trial));
                            done = true;
                        }
                    }
                    if (!pre_row && ord <= 16 && magbits_n >= 0 &&
                        ((unsigned long long)(uint32_t)l.misc[8] << magbits_n) < (1ull << 31)) {
                        if (ord <= 8) fir_lpc_dotn<C, T, 4>(e, r, ord, cshift, l.misc);
                        else fir_lpc_dotn<C, T, 8>(e, r, ord, cshift, l.misc);
                        done = true;
                    }
                }
                if (done) {
                } else if (pre_row && ord <= 8)
                    fir_lpc_o8<C, T>(e, r, ord, cshift,
                                     reinterpret_cast<const double *>(fin_all + (size_t)s * FIN_STRIDE + FIN_DBL));
                else if constexpr (MODE != 0) {
                    bool wide_done = false;
                    if constexpr (MODE == 3 && C % 8 == 0) {
                        const double *cd = reinterpret_cast<const double *>(fin_all + (size_t)s * FIN_STRIDE + FIN_DBL);
                        if (ord <= 12) { fir_lpc_o16<C, T, 12>(e, r, ord, cshift, cd); wide_done = true; }
                        else if (ord <= 16) { fir_lpc_o16<C, T, 16>(e, r, ord, cshift, cd); wide_done = true; }
                    }
                    // (MODE 0 is launched for maximum orders <= 8 only; MODE 3 -- five waves per SIMD, 96 registers --
                    // takes the general FIR in register blocks of four outputs: eight spilled six registers)
                    if (!wide_done) fir_lpc<C, T, (MODE == 3) ? 4 : 8>(e, r, ord, cshift);
                }
                STAMP(3);
                b = rice_search_fast<C, T, (MODE == 2 || MODE == 3)>(e, r, u, ord, true, &porder, &method, &umax_run);
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
        // the emit-side fold (bitio.h:128) differs from rice.c's for |x| >= 2^30, i.e. for
        // folded values from 2^31 (any sample width: it is the residual that counts)
        uint32_t umax = umax_run;
        if (__any((umax >> 31) != 0u)) {
            umax = 0;
#pragma unroll
            for (int o = 0; o < C; o++) {
                u[o] = emit_fold32((int32_t)((u[o] >> 1) ^ (0u - (u[o] & 1u)))) & ((e.i0 + o < order) ? 0u : ~0u);
                umax = max(umax, u[o]);
            }
        }
        // codeword lengths of the run; in 32 bits unless a quotient is huge.
        // A zeroed warm-up entry counts k+1 bits here, taken off again below.
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
                    // a section longer than the window (dense or long blocks, wide samples) takes
                    // several passes: a thread only walks its codewords in the passes its bits
                    // fall into, not in every one
                    const bool touches = (nwords <= wwords) ||
                                         (rel + (long long)(nwarm * k1) + (long long)mine > 0 && rel < (long long)nw * 32);
                    if (touches) {
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
                    }
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
        out->obits = prep[s].obits;          // K0's fields: from the handle's prepare record
        out->wasted = prep[s].wasted;        // (the caller's when the stages run in one call)
        out->ch_mode = prep[s].ch_mode;
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

template <int C, int T, int MODE>
__global__ __launch_bounds__(T, (MODE == 2 || C >= 18 || (C >= 14 && T > 256)) ? 4 : 5)
void k_encode_pow2(fhip_params P, int n, int nsub, const int32_t *__restrict__ smp_all,
                   const int32_t *__restrict__ coefs_all, const int32_t *__restrict__ shift_all,
                   const int32_t *__restrict__ opt_all, const int32_t *__restrict__ fin_all,
                   fhip_subframe_info *__restrict__ info, const fhip_subframe_info *__restrict__ prep,
                   int32_t *__restrict__ res_out, uint8_t *__restrict__ bits_out, long long slot_bytes,
                   int narrow_ok, const int32_t *__restrict__ dev_sub)
{
    if (dev_sub && (int)blockIdx.x >= dev_count(dev_sub, 0)) return;      // (a ragged batch's grid is its bin's capacity)
    encode_pow2_body<C, T, MODE>(P, n, smp_all, coefs_all, shift_all, opt_all, fin_all, info, prep, res_out, bits_out,
                                 slot_bytes, narrow_ok, (int)blockIdx.x);
}

// Several thinly filled bins of a ragged batch in ONE launch (kernels.h: MultiBin, units = subframes; k_order_search_bins
// has the why): 256-thread geometries with runs of 2, 4, 8, 10, 12, 14, 16 samples -- every piece of a 4096 block but three
// eighths, the first four of an 8192 block -- whose row is known when the kernel starts (MODE 0 / 3).
template <int MODE>
__global__ __launch_bounds__(256, 5)
void k_encode_bins(fhip_params P, MultiBin mb, const int32_t *__restrict__ smp,
                   const int32_t *__restrict__ coefs_all, const int32_t *__restrict__ shift_all,
                   const int32_t *__restrict__ opt_all, const int32_t *__restrict__ fin_all,
                   fhip_subframe_info *__restrict__ info, const fhip_subframe_info *__restrict__ prep,
                   uint8_t *__restrict__ bits)
{
    constexpr int T = 256;
    const int blk = blockIdx.x;
    const int k = find_bin(mb, blk);
    const int local = blk - mb.wg0[k];
    if (local >= __builtin_amdgcn_readfirstlane(mb.cnt[mb.cnt_ix[k]])) return;
    const int n = mb.n[k];
    const int s = mb.unit0[k] + local;
    const int32_t *smp_k = smp + mb.smp_off[k] - (long long)mb.unit0[k] * n;          // row s: smp_k + s n
    const long long slot = mb.slot[k];
    uint8_t *bits_k = bits + mb.bits_off[k] - (long long)mb.unit0[k] * slot;           // section s: bits_k + s slot
    const int nar = mb.narrow[k];
#define BODY_(CC) encode_pow2_body<CC, T, MODE>(P, n, smp_k, coefs_all, shift_all, opt_all, fin_all, info, prep, nullptr, bits_k, slot, nar, s)
    const int c = n / T;
    if (c == 2) BODY_(2); else if (c == 4) BODY_(4); else if (c == 8) BODY_(8); else if (c == 10) BODY_(10);
    else if (c == 12) BODY_(12); else if (c == 14) BODY_(14); else BODY_(16);
#undef BODY_
}

}  // namespace

size_t encode_lds_bytes(int n)
{
    if (n < 1 || n > FHIP_MAX_BLOCK) return 0;
    size_t off[10];
    return enc_lds_layout(n > FHIP_MAX_RESIDENT_BLOCK ? 0 : n, off);      // long blocks: no sample image (k_encode_big)
}

// Fast-path geometry for a block size: C samples per thread, T threads,
// n = C*T, T a power of two >= 64.
bool fast_geometry(const fhip_params &p, int n, int *C, int *T)
{
    if (n < 192 || n > FHIP_MAX_BLOCK) return false;
    int odd = n, lg = 0;
    while ((odd & 1) == 0) { odd >>= 1; lg++; }
    int c, t;
    if (odd == 1) {                       // 256 .. 16384
        if (n >= 4096) { c = 16; t = n / 16; }
        else if (n >= 2048) { c = 8; t = 256; }
        else if (n >= 1024) { c = 4; t = 256; }
        else if (n == 512) {
            c = 8; t = 64;
            // partitions finer than a run of 8 (partition orders 7, 8: the eighths of a 4096 block
            // under the VBS presets): a thread per level-8 partition instead of the generic kernel
            if ((n >> p.max_partition_order) < 8 && (n >> p.max_partition_order) >= 2) { c = 2; t = 256; }
        }
        else if (n == 256) { c = 4; t = 64; }
        else return false;
    } else if (odd == 9) {                // 576, 1152, 2304, 4608, 9216
        // 256 threads where the block allows (measured at 4608: (18,256) 84 us, (9,512) 100)
        c = 9; t = 1 << lg;
        if (t >= 512) { c = 18; t >>= 1; }
    } else if (odd == 3) {                // 192, 384, 768, 1536, 3072; 6144 (six eighths of an 8192 block)
        c = 3; t = 1 << lg;
        if (t > 1024) { c = 6; t >>= 1; }
        if (t > 1024) { c = 0; }
        // 3072, 6144 (three / six eighths of a 4096 / 8192 block): runs of 12 in 256 / 512 threads -- a
        // workgroup of 1024 threads per subframe left the CU to one or two subframes at a time
        // (round 3: k_encode_pow2<3,1024> 84 us, <6,1024> 131 us per VBS batch of 1024 blocks)
        if (lg == 10 || lg == 11) { c = 12; t = 1 << (lg - 2); }
        // 1536 (three eighths of a 4096 block): runs of 6 in 256 threads -- a thread per finest partition of the VBS presets;
        // (3, 512) took 11.0 ns per subframe where 1024 / 2048 samples take 4.8 / 5.8 (tools/piece_cost.py)
        if (lg == 9) { c = 6; t = 256; }
    } else if ((odd == 5 || odd == 7) && (lg == 9 || lg == 10)) {
        // 2560, 3584, 5120, 7168: five or seven eighths of a 4096 / 8192 block, the pieces
        // the VBS splitter (vbs.c:36-83) makes most often besides the plain power-of-two ones
        c = 2 * odd; t = 1 << (lg - 1);
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
bool narrow_rows_ok(const fhip_params &p, int nsub, int n, bool lpc_path, bool wave_typed_k1)
{
    // measurements only (the generic K3 forced onto a fast-path geometry reads int32 rows)
    static const bool off = getenv("FHIP_NO_NARROW") != nullptr || getenv("FHIP_K3_GENERIC") != nullptr;
    if (off || p.channels != 2 || (n & 3) != 0 || n > 8192) return false;
    int fc = 0, ft = 0;
    if (!fast_geometry(p, n, &fc, &ft) || (fc % 8) != 0) return false;
    if (lpc_path && !wave_typed_k1 && !autocorr_is_wave_typed(nsub, n, p.max_prediction_order)) return false;
    return true;
}

// the K3 launch group of a bin of a ragged batch: 256 (k_encode_bins), or -1: launch_encode for the bin alone
int encode_group(const fhip_params &p, int n, bool order_known)
{
    static const bool generic = getenv("FHIP_K3_GENERIC") != nullptr;           // measurements only
    int fc = 0, ft = 0;
    if (generic || !fast_geometry(p, n, &fc, &ft)) return -1;
    const bool single_row = (p.prediction_type == 2) && (n > p.max_prediction_order) && (p.order_method <= 1 || order_known);
    if (!single_row) return -1;
    if (ft == 256 && (fc == 2 || fc == 4 || fc == 8 || fc == 10 || fc == 12 || fc == 14 || fc == 16)) return 256;
    return -1;
}

hipError_t launch_encode_bins(hipStream_t st, const fhip_params &p, const MultiBin &mb, const int32_t *smp,
                              const int32_t *coefs, const int32_t *shift, const int32_t *opt_order, const int32_t *fin,
                              fhip_subframe_info *info, const fhip_subframe_info *prep, uint8_t *bits)
{
    if (mb.nbins < 1) return hipSuccess;
    const int grid = mb.wg0[mb.nbins];
    if (grid == 0) return hipSuccess;
    if (!prep || prep == info) return hipErrorInvalidValue;
    const int mode = (p.max_prediction_order > 8) ? 3 : 0;           // launch_encode's rule for a single row
    size_t lds = 0;
    for (int k = 0; k < mb.nbins; k++) {
        int fc = 0, ft = 0;
        if (encode_group(p, mb.n[k], true) != 256 || !fast_geometry(p, mb.n[k], &fc, &ft)) return hipErrorInvalidValue;
        size_t off[12], img = 0;
#define SZ_(CC) if (fc == CC) img = (size_t)SmpImg<CC, 256>::SIZE
        SZ_(2); SZ_(4); SZ_(8); SZ_(10); SZ_(12); SZ_(14); SZ_(16);
#undef SZ_
        const size_t l = fast_lds_layout(mb.n[k], img, off, fast_wide_window(mode, p.bits_per_sample));
        lds = l > lds ? l : lds;
    }
#define LAUNCH_EB(MM)                                                                        \
    do {                                                                                     \
        hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_encode_bins<MM>), \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (er != hipSuccess) return er;                                                     \
        hipLaunchKernelGGL((k_encode_bins<MM>), dim3(grid), dim3(256), lds, st, p, mb, smp, coefs, shift, opt_order, \
                           fin, info, prep, bits);                                           \
        return hipGetLastError();                                                            \
    } while (0)
    if (mode == 3) LAUNCH_EB(3);
    LAUNCH_EB(0);
#undef LAUNCH_EB
}

hipError_t launch_encode(hipStream_t st, const fhip_params &p, const int32_t *smp,
                         int nsub, int n, const int32_t *coefs, const int32_t *shift,
                         const int32_t *opt_order, const int32_t *fin,
                         fhip_subframe_info *info,
                         int32_t *residual, uint8_t *bits, int64_t slot_bytes,
                         int raw_order, int raw_lpc, bool narrow_ok, const fhip_subframe_info *prep,
                         bool order_known, const int32_t *dev_sub)
{
    if (nsub == 0) return hipSuccess;
    if (!prep || prep == info) return hipErrorInvalidValue;      // K3 reads K0's records beside the ones it writes
    int fc = 0, ft = 0;
    static const bool force_generic = getenv("FHIP_K3_GENERIC") != nullptr;    // measurements only
    if (raw_order < 0 && !force_generic && fast_geometry(p, n, &fc, &ft)) {
        size_t off[12];
        size_t lds = 0;
#define LAUNCH_FAST2(CC, TT, MM)                                                             \
    do {                                                                                     \
        lds = fast_lds_layout(n, (size_t)SmpImg<CC, TT>::SIZE, off,                          \
                              fast_wide_window(MM, p.bits_per_sample));                      \
        hipError_t er = hipFuncSetAttribute(                                                 \
            reinterpret_cast<const void *>(&k_encode_pow2<CC, TT, MM>),                      \
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        if (er != hipSuccess) return er;                                                     \
        hipLaunchKernelGGL((k_encode_pow2<CC, TT, MM>), dim3(nsub), dim3(TT), lds, st, p, n, \
                           nsub, smp, coefs, shift, opt_order, fin, info, prep, residual,    \
                           bits, (long long)slot_bytes, narrow_ok ? 1 : 0, dev_sub);         \
    } while (0)
        // one quantised row known up front (MAX / EST): the lean instance
        // ... or chosen by the order-search kernel, which leaves the same compact row
        const bool single_row = (p.prediction_type == 2) && (n > p.max_prediction_order) &&
                                (p.order_method <= 1 || order_known);
        const bool fixed_only = (p.prediction_type == 1) && n >= 5;
        // orders 9..16 on 16-bit rows: their own instance, so that the lean one (orders <= 8,
        // the default presets) does not carry the second packed FIR (it cost it 3 %)
        const bool wide_rows = p.max_prediction_order > 8;
#define LAUNCH_FAST(CC, TT)                                                                  \
    do {                                                                                     \
        if (single_row && wide_rows) LAUNCH_FAST2(CC, TT, 3);                                \
        else if (single_row) LAUNCH_FAST2(CC, TT, 0);                                        \
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
        case 20256: LAUNCH_FAST(2, 256); break;
        case 90064: LAUNCH_FAST(9, 64); break;
        case 100256: LAUNCH_FAST(10, 256); break;
        case 100512: LAUNCH_FAST(10, 512); break;
        case 140256: LAUNCH_FAST(14, 256); break;
        case 140512: LAUNCH_FAST(14, 512); break;
        case 180256: LAUNCH_FAST(18, 256); break;
        case 180512: LAUNCH_FAST(18, 512); break;
        case 90128: LAUNCH_FAST(9, 128); break;
        case 90256: LAUNCH_FAST(9, 256); break;
        case 30064: LAUNCH_FAST(3, 64); break;
        case 30128: LAUNCH_FAST(3, 128); break;
        case 30256: LAUNCH_FAST(3, 256); break;
        case 30512: LAUNCH_FAST(3, 512); break;
        case 31024: LAUNCH_FAST(3, 1024); break;
        case 61024: LAUNCH_FAST(6, 1024); break;
        case 60256: LAUNCH_FAST(6, 256); break;
        case 120256: LAUNCH_FAST(12, 256); break;
        case 120512: LAUNCH_FAST(12, 512); break;
        default: return hipErrorInvalidValue;
        }
#undef LAUNCH_FAST
#undef LAUNCH_FAST2
        return hipGetLastError();
    }
    const size_t lds = encode_lds_bytes(n);
    if (lds == 0) return hipErrorInvalidValue;
    if (n > FHIP_MAX_RESIDENT_BLOCK) {
        hipLaunchKernelGGL(k_encode_big, dim3(nsub), dim3(NT), lds, st, p, n, smp, coefs, shift,
                           opt_order, info, prep, residual, bits, (long long)slot_bytes, raw_order,
                           raw_lpc, dev_sub);
        return hipGetLastError();
    }
    const int chunk = (n + NT - 1) / NT;
#define LAUNCH_ENC(CC)                                                                       \
    do {                                                                                     \
        hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_encode<CC>),   \
                                            hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                            (int)lds);                                       \
        if (er != hipSuccess) return er;                                                     \
        hipLaunchKernelGGL(k_encode<CC>, dim3(nsub), dim3(NT), lds, st, p, n, smp, coefs,    \
                           shift, opt_order, info, prep, residual, bits,                     \
                           (long long)slot_bytes,                                            \
                           raw_order, raw_lpc, dev_sub);                                     \
    } while (0)
    if (chunk <= 16) LAUNCH_ENC(16);
    else if (chunk <= 32) LAUNCH_ENC(32);
    else LAUNCH_ENC(64);
#undef LAUNCH_ENC
    return hipGetLastError();
}

}  // namespace fhip
