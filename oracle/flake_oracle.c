/*
 * flake_oracle.c -- CPU restatement of libflake's prediction/entropy path.
 *
 * TEST INFRASTRUCTURE ONLY (see flake_oracle.h for the rules and the pinning
 * status of each part).  Written from the behaviour of the reference, not
 * from its text; every function cites the file:line it follows (paths are
 * relative to /root/reference).  Build with -ffp-contract=off: the fp64
 * stages must round after every multiply and every add, in the order given.
 */
#include "flake_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* small helpers                                                      */
/* ------------------------------------------------------------------ */

/* common.h:53-65 log2i(): floor(log2(v)), 0 for v == 0 */
static int ilog2_u32(uint32_t v)
{
    int r = 0;
    while (v > 1) { v >>= 1; r++; }
    return r;
}

/* two's-complement wrap-around int32 arithmetic (what gcc/x86-64 produces
 * for the reference's signed expressions at the 32-bit edge) */
static int32_t wrap_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t wrap_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static int32_t wrap_abs(int32_t a) { return a < 0 ? (int32_t)(0u - (uint32_t)a) : a; }

/* bitio.h:128-129, the EMIT-side sign fold: v = -2*val-1; v ^= v>>31 in int.
 * Equal to the search-side fold (rice.c:122) for |val| < 2^30 only (8-Q7). */
static uint32_t emit_fold(int32_t val)
{
    int32_t v = (int32_t)(0u - 2u * (uint32_t)val - 1u);
    v ^= (v >> 31);
    return (uint32_t)v;
}

/* ------------------------------------------------------------------ */
/* parameters: encode.c:158-266 flake_set_defaults                    */
/* ------------------------------------------------------------------ */
void fo_set_defaults(fo_params *p, int level)
{
    /* level 5 is the base row; the table lists differences from it */
    p->order_method = FO_OM_EST;
    p->stereo_method = FO_STEREO_ESTIMATE;
    p->block_size = 4096;
    p->prediction_type = FO_PRED_LEVINSON;
    p->min_prediction_order = 1;
    p->max_prediction_order = 8;
    p->min_partition_order = 0;
    p->max_partition_order = 5;
    p->variable_block_size = 0;
    p->allow_vbs = 0;
    p->lpc_precision = 15;            /* encode.c:443 */
    if (level <= 2) {
        static const int lo[3] = {2, 2, 0}, hi[3] = {2, 4, 4};
        p->block_size = 1152;
        p->prediction_type = FO_PRED_FIXED;
        p->min_prediction_order = lo[level];
        p->max_prediction_order = hi[level];
        p->max_partition_order = 3;
        if (level == 0) p->stereo_method = FO_STEREO_INDEPENDENT;
    } else if (level == 3) {
        p->stereo_method = FO_STEREO_INDEPENDENT;
        p->max_prediction_order = 6;
        p->max_partition_order = 4;
    } else if (level == 4) {
        p->max_partition_order = 4;
    } else if (level == 6) {
        p->max_partition_order = 6;
    } else if (level == 7) {
        p->order_method = FO_OM_4LEVEL;
        p->max_partition_order = 6;
    } else if (level >= 8) {
        int search = (level == 10 || level == 12);
        int big = (level >= 11);
        p->order_method = search ? FO_OM_SEARCH : FO_OM_LOG;
        p->max_prediction_order = big ? 32 : 12;
        p->max_partition_order = (level == 8) ? 6 : 8;
        if (big) p->block_size = 8192;
        if (level >= 9) { p->allow_vbs = 1; p->variable_block_size = 1; }
    }
}

/* ------------------------------------------------------------------ */
/* lpc.c                                                              */
/* ------------------------------------------------------------------ */

/*
 * lpc.c:28-40 apply_welch_window + lpc.c:46-71 compute_autocorr.
 * Window weight for the pair (i, n-1-i) is 1-(c-i)^2 with c = 2/(n-1)-1,
 * exactly as written there.  For each lag the products
 * d[p]*d[p-lag], p = lag..n-1, are added one at a time: p <= max_lag into
 * acc0; after that p alternates acc0, acc1, starting with acc0 at
 * p = max_lag+1.  Both accumulators start at 1.0 (8-Q1).
 * Odd n: the reference leaves the centre element of its malloc'd buffer
 * uninitialised; here it is defined as 0.0.
 */
void fo_window_autocorr(const int32_t *smp, int n, int lag, double *autoc)
{
    double *d = (double *)calloc((size_t)n + 16, sizeof(double));
    double c = (2.0 / (n - 1.0)) - 1.0;
    int half = n >> 1;
    for (int i = 0; i < half; i++) {
        double t = c - i;
        double w = 1.0 - (t * t);
        d[i] = smp[i] * w;
        d[n - 1 - i] = smp[n - 1 - i] * w;
    }
    d[n] = 0.0;

    for (int L = 0; L <= lag; L++) {
        double acc0 = 1.0, acc1 = 1.0;
        int p = L;
        for (; p <= lag; p++) {
            double prod = d[p] * d[p - L];
            acc0 = acc0 + prod;
        }
        for (; p <= n - 1; p += 2) {
            double pa = d[p] * d[p - L];
            acc0 = acc0 + pa;
            double pb = d[p + 1] * d[p + 1 - L];   /* d[n] == 0 pads the tail */
            acc1 = acc1 + pb;
        }
        autoc[L] = acc0 + acc1;
    }
    free(d);
}

/*
 * lpc.c:77-117 compute_lpc_coefs.  lpc is [32][32] row-major; row i holds
 * the order-(i+1) predictor.  With ref != NULL the reflection coefficients
 * are taken from ref and autoc is not read.
 */
void fo_levinson(const double *autoc, int max_order, const double *ref, double *lpc)
{
    double a[FO_MAX_ORDER];
    double err = autoc ? autoc[0] : 1.0;
    memset(a, 0, sizeof(a));

    for (int i = 0; i < max_order; i++) {
        double r;
        if (ref) {
            r = ref[i];
        } else {
            r = -autoc[i + 1];
            for (int j = 0; j < i; j++) {
                double t = a[j] * autoc[i - j];
                r = r - t;
            }
            r = r / err;
            double rr = r * r;
            double om = 1.0 - rr;
            err = err * om;
        }
        a[i] = r;
        int h = i >> 1;
        for (int j = 0; j < h; j++) {
            double lo = a[j];
            double hi = a[i - 1 - j];
            double t0 = r * hi;
            a[j] = lo + t0;
            double t1 = r * lo;
            a[i - 1 - j] = hi + t1;
        }
        if (i & 1) {                       /* middle element, lpc.c:109-111 */
            double t = a[h] * r;
            a[h] = a[h] + t;
        }
        for (int j = 0; j <= i; j++)
            lpc[i * FO_MAX_ORDER + j] = -a[j];
    }
}

/*
 * lpc.c:125-162 compute_lpc_coefs_est: Schur recursion for the reflection
 * coefficients, order estimate = highest index with |ref| > 0.10 (+1, at
 * least 1), then Levinson driven by ref for that order only.
 */
int fo_schur_order_est(const double *autoc, int max_order, double *lpc)
{
    double g0[FO_MAX_ORDER], g1[FO_MAX_ORDER], ref[FO_MAX_ORDER];
    for (int i = 0; i < max_order; i++) g0[i] = g1[i] = autoc[i + 1];
    double e = autoc[0];
    ref[0] = -g1[0] / e;
    { double t = g1[0] * ref[0]; e = e + t; }
    for (int i = 1; i < max_order; i++) {
        double k = ref[i - 1];
        for (int j = 0; j < max_order - i; j++) {
            double up = g1[j + 1];
            double t0 = k * g0[j];
            g1[j] = up + t0;
            double t1 = up * k;
            g0[j] = t1 + g0[j];
        }
        ref[i] = -g1[0] / e;
        double t = g1[0] * ref[i];
        e = e + t;
    }
    int est = 1;
    for (int i = max_order - 1; i >= 0; i--) {
        if (fabs(ref[i]) > 0.10) { est = i + 1; break; }
    }
    fo_levinson(NULL, est, ref, lpc);
    return est;
}

/*
 * lpc.c:167-219 quantize_lpc_coefs.  Error-feedback rounding with C
 * truncation; lpc_row is scaled in place when the shift bottoms out (8-Q4).
 */
void fo_quantize_coefs(double *lpc_row, int order, int precision, int32_t *out, int *shift)
{
    int32_t qmax = (1 << (precision - 1)) - 1;
    double cmax = 0.0;
    for (int i = 0; i < order; i++) {
        double m = fabs(lpc_row[i]);
        if (m > cmax) cmax = m;
    }
    if (cmax * (double)(1 << 15) < 1.0) {
        *shift = 0;
        for (int i = 0; i < order; i++) out[i] = 0;
        return;
    }
    int sh = 15;
    while (sh > 0 && cmax * (double)(1 << sh) > (double)qmax) sh--;
    if (sh == 0 && cmax > (double)qmax) {
        double scale = ((double)qmax) / cmax;
        for (int i = 0; i < order; i++) lpc_row[i] = lpc_row[i] * scale;
    }
    double carry = 0.0;
    double mul = (double)(1 << sh);
    for (int i = 0; i < order; i++) {
        double t = lpc_row[i] * mul;
        carry = carry + t;
        int q = (int)(carry + 0.5);
        if (q <= -qmax) q = -qmax + 1;
        if (q > qmax) q = qmax;
        carry = carry - (double)q;
        out[i] = q;
    }
    *shift = sh;
}

/*
 * lpc.c:224-257 lpc_calc_coefs.  Only row opt_order-1 of coefs/shift is
 * written for MAX/EST; every row for the search methods.
 */
int fo_lpc_calc_coefs(const int32_t *smp, int n, int max_order, int precision,
                      int omethod, int32_t *coefs, int *shift)
{
    double autoc[FO_MAX_ORDER + 1];
    double *lpc = (double *)calloc(FO_MAX_ORDER * FO_MAX_ORDER, sizeof(double));
    int opt = max_order;

    fo_window_autocorr(smp, n, max_order, autoc);
    if (omethod == FO_OM_EST)
        opt = fo_schur_order_est(autoc, max_order, lpc);
    else
        fo_levinson(autoc, max_order, NULL, lpc);

    if (omethod == FO_OM_MAX || omethod == FO_OM_EST) {
        int r = opt - 1;
        fo_quantize_coefs(&lpc[r * FO_MAX_ORDER], r + 1, precision,
                          &coefs[r * FO_MAX_ORDER], &shift[r]);
    } else {
        for (int r = 0; r < max_order; r++)
            fo_quantize_coefs(&lpc[r * FO_MAX_ORDER], r + 1, precision,
                              &coefs[r * FO_MAX_ORDER], &shift[r]);
    }
    free(lpc);
    return opt;
}

/* ------------------------------------------------------------------ */
/* rice.c                                                             */
/* ------------------------------------------------------------------ */

/* rice.h:48 rice_encode_count, evaluated in uint64 as C does (8-Q5) */
static uint64_t rice_count64(uint64_t sum, int n, int k)
{
    uint64_t lin = (uint64_t)(int64_t)(n * (k + 1));
    uint64_t rest = (sum - (uint64_t)(int64_t)(n >> 1)) >> k;
    return lin + rest;
}

/* rice.c:30-45 find_optimal_rice_param: first strict minimum over k = 0..30
 * of the count truncated to 32 bits */
int fo_rice_best_k(uint64_t sum, int n)
{
    int best = 0;
    uint32_t best_bits = (uint32_t)rice_count64(sum, n, 0);
    for (int k = 1; k <= 30; k++) {
        uint32_t b = (uint32_t)rice_count64(sum, n, k);
        if (b < best_bits) { best_bits = b; best = k; }
    }
    return best;
}

/*
 * rice.c:105-139 calc_rice_params with calc_sums (:76-103) and
 * calc_optimal_rice_params (:47-74) folded in.  Partition 0 of level p
 * holds (n>>p)-pred_order residuals starting at pred_order; the zig-zag
 * value of every sample is (2*x) ^ (x>>31) in uint32 (8-Q6, 8-Q7).
 * Partition-order ties go to the higher order.
 */
uint32_t fo_rice_search(fo_subframe *sf, int pmin, int pmax,
                        const int32_t *res, int n, int pred_order)
{
    static const int top = 8;
    uint64_t (*sums)[FO_MAX_PARTS] = malloc(sizeof(uint64_t) * (top + 1) * FO_MAX_PARTS);
    int parts = 1 << pmax;
    int psize = n >> pmax;

    for (int i = 0; i < parts; i++) {
        int beg = (i == 0) ? pred_order : i * psize;
        int end = (i == 0) ? psize : beg + psize;
        uint64_t s = 0;
        for (int j = beg; j < end; j++) {
            uint32_t x = (uint32_t)res[j];
            uint32_t u = (x << 1) ^ (uint32_t)(res[j] >> 31);
            s += u;
        }
        sums[pmax][i] = s;
    }
    for (int p = pmax - 1; p >= pmin; p--)
        for (int j = 0; j < (1 << p); j++)
            sums[p][j] = sums[p + 1][2 * j] + sums[p + 1][2 * j + 1];

    uint32_t best_bits = 0;
    int have = 0;
    int32_t trial[FO_MAX_PARTS];
    for (int p = pmin; p <= pmax; p++) {
        int np = 1 << p;
        uint32_t bits = 0;
        int method = 0;
        for (int i = 0; i < np; i++) {
            int cnt = (n >> p) - (i == 0 ? pred_order : 0);
            int k = fo_rice_best_k(sums[p][i], cnt);
            trial[i] = k;
            if (k > 14) method = 1;
            bits += (uint32_t)rice_count64(sums[p][i], cnt, k);
        }
        bits += 4u * (uint32_t)np;
        if (!have || bits <= best_bits) {
            have = 1;
            best_bits = bits;
            sf->rice_method = method;
            sf->porder = p;
            memcpy(sf->rparams, trial, sizeof(int32_t) * np);
        }
    }
    free(sums);
    return best_bits;
}

/* rice.c:148-155 limit_max_partition_order */
static int clamp_porder(int porder, int n, int order)
{
    int lim = ilog2_u32((uint32_t)(n ^ (n - 1)));
    if (porder > lim) porder = lim;
    if (order > 0) {
        int l2 = ilog2_u32((uint32_t)(n / order));
        if (porder > l2) porder = l2;
    }
    return porder;
}

/* rice.c:157-187 calc_rice_params_common/_fixed/_lpc */
uint32_t fo_subframe_bits(fo_subframe *sf, int pmin, int pmax, const int32_t *res,
                          int n, int pred_order, int bps, int precision, int lpc)
{
    pmin = clamp_porder(pmin, n, pred_order);
    pmax = clamp_porder(pmax, n, pred_order);
    uint32_t bits = (uint32_t)(pred_order * bps + 2);
    if (lpc) bits += (uint32_t)(4 + 5 + pred_order * precision);
    bits += fo_rice_search(sf, pmin, pmax, res, n, pred_order);
    bits += (uint32_t)(sf->rice_method + 4);
    return bits;
}

/* ------------------------------------------------------------------ */
/* optimize.c                                                         */
/* ------------------------------------------------------------------ */

/* optimize.c:34-68 encode_residual_fixed: finite differences of order 0..4
 * with 64-bit intermediates, truncated to int32 */
void fo_residual_fixed(int32_t *res, const int32_t *smp, int n, int order)
{
    static const int64_t binom[5][5] = {
        {1, 0, 0, 0, 0}, {1, -1, 0, 0, 0}, {1, -2, 1, 0, 0},
        {1, -3, 3, -1, 0}, {1, -4, 6, -4, 1}};
    if (order < 0 || order > 4) return;
    for (int i = 0; i < order && i < n; i++) res[i] = smp[i];
    for (int i = order; i < n; i++) {
        int64_t v = 0;
        for (int j = 0; j <= order; j++) v += binom[order][j] * (int64_t)smp[i - j];
        res[i] = (int32_t)v;
    }
}

/* optimize.c:70-122 encode_residual_lpc: 64-bit prediction, arithmetic
 * shift, truncation to int32 */
void fo_residual_lpc(int32_t *res, const int32_t *smp, int n, int order,
                     const int32_t *coefs, int shift)
{
    for (int i = 0; i < order && i < n; i++) res[i] = smp[i];
    for (int i = order; i < n; i++) {
        int64_t pred = 0;
        for (int j = order; j >= 1; j--)
            pred += (int64_t)coefs[j - 1] * (int64_t)smp[i - j];
        res[i] = (int32_t)((int64_t)smp[i] - (pred >> shift));
    }
}

/* exact length in bits of what output_residual() (encode.c:766-798) writes */
int64_t fo_residual_section_bits(const fo_subframe *sf, const int32_t *res, int n)
{
    int64_t bits = 2 + 4;
    int np = 1 << sf->porder;
    int psize = n >> sf->porder;
    int pbits = 4 + sf->rice_method;
    int j = sf->order;
    for (int p = 0; p < np; p++) {
        int k = sf->rparams[p];
        int end = (p + 1) * psize;
        bits += pbits;
        for (; j < end && j < n; j++) {
            uint32_t u = emit_fold(res[j]);
            bits += (int64_t)(u >> k) + 1 + k;
        }
    }
    return bits;
}

/*
 * optimize.c:124-276 encode_residual.  sf->obits must be set on entry.
 * Candidate evaluation order and tie rules follow the reference (8-Q8);
 * the winning order is re-encoded at the end exactly as there.
 */
static int encode_residual_core(const fo_params *p, fo_subframe *sf,
                                const int32_t *smp, int32_t *res, int n)
{
    int i;
    sf->order = 0; sf->shift = 0; sf->rice_method = 0; sf->porder = 0;
    sf->rice_nbits = 0;

    /* CONSTANT, optimize.c:143-151 */
    for (i = 1; i < n; i++) if (smp[i] != smp[0]) break;
    if (i == n) {
        sf->type = sf->type_code = FO_SUB_CONSTANT;
        res[0] = smp[0];
        sf->est_bits = (uint32_t)sf->obits;
        return sf->obits;
    }
    /* VERBATIM, optimize.c:153-158 */
    if (n < 5 || p->prediction_type == FO_PRED_NONE) {
        sf->type = sf->type_code = FO_SUB_VERBATIM;
        memcpy(res, smp, sizeof(int32_t) * (size_t)n);
        sf->est_bits = (uint32_t)(sf->obits * n);
        return sf->obits * n;
    }

    int omethod = p->order_method;
    int min_order = p->min_prediction_order;
    int max_order = p->max_prediction_order;
    int pmin = p->min_partition_order, pmax = p->max_partition_order;
    uint32_t ret;

    /* FIXED, optimize.c:167-190 */
    if (p->prediction_type == FO_PRED_FIXED || n <= max_order) {
        uint32_t bits[5];
        if (max_order > 4) max_order = 4;
        int best = min_order;
        for (i = min_order; i <= max_order; i++) {
            fo_residual_fixed(res, smp, n, i);
            bits[i] = fo_subframe_bits(sf, pmin, pmax, res, n, i, sf->obits, 0, 0);
            if (i > min_order && bits[i] < bits[best]) best = i;
        }
        sf->order = best;
        sf->type = FO_SUB_FIXED;
        sf->type_code = FO_SUB_FIXED | best;
        if (best != max_order) {
            fo_residual_fixed(res, smp, n, best);
            ret = fo_subframe_bits(sf, pmin, pmax, res, n, best, sf->obits, 0, 0);
        } else {
            ret = bits[best];
        }
        sf->est_bits = ret;
        return (int)ret;
    }

    /* LPC, optimize.c:192-275 */
    int32_t *coefs = (int32_t *)calloc(FO_MAX_ORDER * FO_MAX_ORDER, sizeof(int32_t));
    int shift[FO_MAX_ORDER];
    memset(shift, 0, sizeof(shift));
    int prec = p->lpc_precision;
    int est = fo_lpc_calc_coefs(smp, n, max_order, prec, omethod, coefs, shift);
    int opt;   /* zero-based row until the ++ below */

#define TRY_ORDER(row) ( \
        fo_residual_lpc(res, smp, n, (row) + 1, &coefs[(row) * FO_MAX_ORDER], shift[row]), \
        fo_subframe_bits(sf, pmin, pmax, res, n, (row) + 1, sf->obits, prec, 1))

    if (omethod == FO_OM_MAX) {
        opt = max_order - 1;
    } else if (omethod == FO_OM_EST) {
        opt = est - 1;
    } else if (omethod >= FO_OM_2LEVEL && omethod <= FO_OM_8LEVEL) {
        /* optimize.c:202-223: indices high -> low, strict '<' */
        int levels = 1 << (omethod - 1);
        uint32_t best_bits = 0;
        int have = 0;
        opt = max_order - 1;
        for (i = levels - 1; i >= 0; i--) {
            int row = min_order + (((max_order - min_order + 1) * (i + 1)) / levels) - 2;
            if (row < 0) row = 0;
            uint32_t b = TRY_ORDER(row);
            if (!have) { have = 1; best_bits = b; /* opt stays max_order-1 */ }
            else if (b < best_bits) { best_bits = b; opt = row; }
        }
        /* The reference keeps opt_order = max_order-1 when index levels-1 wins,
         * and that index always maps to row max_order-1. */
    } else if (omethod == FO_OM_SEARCH) {
        /* optimize.c:224-238: rows 0..max-1 regardless of min_order */
        uint32_t best_bits = 0;
        opt = 0;
        for (i = 0; i < max_order; i++) {
            uint32_t b = TRY_ORDER(i);
            if (i == 0 || b < best_bits) { best_bits = b; opt = i; }
        }
    } else if (omethod == FO_OM_LOG) {
        /* optimize.c:239-261 */
        uint32_t bits[FO_MAX_ORDER];
        memset(bits, 0xFF, sizeof(bits));
        opt = min_order - 1 + (max_order - min_order) / 3;
        for (int step = 16; step > 0; step >>= 1) {
            int last = opt;
            for (i = last - step; i <= last + step; i += step) {
                if (i < min_order - 1 || i >= max_order || bits[i] < UINT32_MAX) continue;
                bits[i] = TRY_ORDER(i);
                if (bits[i] < bits[opt]) opt = i;
            }
        }
    } else {
        free(coefs);
        return -1;
    }
#undef TRY_ORDER
    opt++;

    sf->order = opt;
    sf->type = FO_SUB_LPC;
    sf->type_code = FO_SUB_LPC | (opt - 1);
    sf->shift = shift[opt - 1];
    for (i = 0; i < opt; i++) sf->coefs[i] = coefs[(opt - 1) * FO_MAX_ORDER + i];
    fo_residual_lpc(res, smp, n, opt, sf->coefs, sf->shift);
    ret = fo_subframe_bits(sf, pmin, pmax, res, n, opt, sf->obits, prec, 1);
    sf->est_bits = ret;
    free(coefs);
    return (int)ret;
}

int fo_encode_residual(const fo_params *p, fo_subframe *sf,
                       const int32_t *smp, int32_t *res, int n)
{
    int rc = encode_residual_core(p, sf, smp, res, n);
    memset(sf->warmup, 0, sizeof(sf->warmup));
    if (rc >= 0) {
        if (sf->type == FO_SUB_CONSTANT) sf->warmup[0] = res[0];
        for (int w = 0; w < sf->order && w < n; w++) sf->warmup[w] = res[w];
    }
    return rc;
}

/* ------------------------------------------------------------------ */
/* encode.c feeder stages                                             */
/* ------------------------------------------------------------------ */

/*
 * encode.c:598-643 calc_decorr_scores.  Sums of |2nd-order residual| for
 * L, R, (L+R)>>1, L-R from sample 2 on; each doubled sum is priced by the
 * Rice estimator WITHOUT the 32-bit truncation (8-Q5, 8-Q10); first
 * minimum of {LR, LS, RS, MS}.
 */
int fo_stereo_mode(const int32_t *left, const int32_t *right, int n)
{
    uint64_t sum[4] = {0, 0, 0, 0};
    for (int i = 2; i < n; i++) {
        int32_t lt = wrap_add(wrap_sub(left[i], (int32_t)(2u * (uint32_t)left[i - 1])), left[i - 2]);
        int32_t rt = wrap_add(wrap_sub(right[i], (int32_t)(2u * (uint32_t)right[i - 1])), right[i - 2]);
        sum[2] += (uint64_t)(int64_t)wrap_abs(wrap_add(lt, rt) >> 1);
        sum[3] += (uint64_t)(int64_t)wrap_abs(wrap_sub(lt, rt));
        sum[0] += (uint64_t)(int64_t)wrap_abs(lt);
        sum[1] += (uint64_t)(int64_t)wrap_abs(rt);
    }
    uint64_t cost[4];
    for (int i = 0; i < 4; i++) {
        int k = fo_rice_best_k(2 * sum[i], n);
        cost[i] = rice_count64(2 * sum[i], n, k);
    }
    uint64_t score[4] = {cost[0] + cost[1], cost[0] + cost[3],
                         cost[1] + cost[3], cost[2] + cost[3]};
    int best = 0;
    for (int i = 1; i < 4; i++) if (score[i] < score[best]) best = i;
    static const int modes[4] = {FO_CH_LEFT_RIGHT, FO_CH_LEFT_SIDE,
                                 FO_CH_RIGHT_SIDE, FO_CH_MID_SIDE};
    return modes[best];
}

/*
 * encode.c:490-536 init_frame (obits), :541-553 copy_samples, :648-694
 * channel_decorrelation, :558-593 remove_wasted_bits, in that order
 * (8-Q10, 8-Q11).
 */
int fo_prepare_frame(const fo_params *p, const int32_t *pcm, int n,
                     int32_t *smp, fo_subframe *sf)
{
    int ch, nch = p->channels, bps = p->bits_per_sample;
    for (ch = 0; ch < nch; ch++) {
        int32_t *dst = smp + (size_t)ch * n;
        for (int i = 0; i < n; i++) dst[i] = pcm[(size_t)i * nch + ch];
        sf[ch].obits = bps;
        sf[ch].wasted = 0;
    }
    int mode;
    if (nch != 2) {
        mode = FO_CH_NOT_STEREO;
    } else if (n <= 32 || p->stereo_method == FO_STEREO_INDEPENDENT) {
        mode = FO_CH_LEFT_RIGHT;
    } else {
        int32_t *l = smp, *r = smp + n;
        mode = fo_stereo_mode(l, r, n);
        if (mode == FO_CH_MID_SIDE) {
            for (int i = 0; i < n; i++) {
                int32_t a = l[i], b = r[i];
                l[i] = wrap_add(a, b) >> 1;
                r[i] = wrap_sub(a, b);
            }
            sf[1].obits++;
        } else if (mode == FO_CH_LEFT_SIDE) {
            for (int i = 0; i < n; i++) r[i] = wrap_sub(l[i], r[i]);
            sf[1].obits++;
        } else if (mode == FO_CH_RIGHT_SIDE) {
            for (int i = 0; i < n; i++) l[i] = wrap_sub(l[i], r[i]);
            sf[0].obits++;
        }
    }
    for (ch = 0; ch < nch; ch++) {
        int32_t *s = smp + (size_t)ch * n;
        int wasted = bps - 1;
        for (int i = 0; i < n && wasted; i++) {
            uint32_t v = (uint32_t)s[i];
            if (v) {
                int tz = 0;
                while (!(v & 1u)) { v >>= 1; tz++; }
                if (tz < wasted) wasted = tz;
            }
        }
        if (wasted == bps - 1) {
            wasted = 0;
        } else if (wasted) {
            for (int i = 0; i < n; i++) s[i] >>= wasted;
            sf[ch].obits -= wasted;
        }
        sf[ch].wasted = wasted;
        sf[ch].ch_mode = mode;
    }
    return mode;
}

/* ------------------------------------------------------------------ */
/* bit writer and emit                                                */
/* ------------------------------------------------------------------ */

/* MSB-first writer producing the same byte stream as bitio.h:83-141 when
 * that writer does not overflow; 'over' latches once cap is exceeded.
 * Bits collect in a 64-bit accumulator and leave a byte at a time. */
typedef struct {
    uint8_t *buf; int64_t cap; int64_t nbits; int over;
    uint64_t acc; int nacc; int64_t wpos;
} bitsink;

static void sink_init(bitsink *s, uint8_t *buf, int64_t cap)
{
    s->buf = buf; s->cap = cap; s->nbits = 0; s->over = 0;
    s->acc = 0; s->nacc = 0; s->wpos = 0;
}

static void sink_put(bitsink *s, int nb, uint32_t val)
{
    if (nb == 0) return;
    s->nbits += nb;
    if (s->over) return;
    if ((s->nbits + 7) / 8 > s->cap) { s->over = 1; return; }
    s->acc = (s->acc << nb) | (uint64_t)val;
    s->nacc += nb;
    while (s->nacc >= 8) {
        s->nacc -= 8;
        s->buf[s->wpos++] = (uint8_t)(s->acc >> s->nacc);
    }
}

/* write the pending partial byte (zero-padded) without consuming it */
static void sink_sync(bitsink *s)
{
    if (!s->over && s->nacc > 0)
        s->buf[s->wpos] = (uint8_t)((s->acc << (8 - s->nacc)) & 0xFF);
}

static void sink_put_signed(bitsink *s, int nb, int32_t val)
{
    uint32_t m = (nb >= 32) ? 0xFFFFFFFFu : ((1u << nb) - 1u);
    sink_put(s, nb, (uint32_t)val & m);
}

/* bitio.h:120-141 bitwriter_write_rice_signed: q = (u>>k) zeros, a one,
 * then the k low bits of u */
static void sink_put_rice(bitsink *s, int k, int32_t v)
{
    uint32_t u = emit_fold(v);
    uint32_t q = u >> k;
    if (s->over || (s->nbits + (int64_t)q + 1 + k + 7) / 8 > s->cap) {
        s->over = 1; s->nbits += (int64_t)q + 1 + k; return;
    }
    while (q >= 31) { sink_put(s, 31, 0); q -= 31; }
    sink_put(s, (int)q + 1, 1);
    if (k) sink_put(s, k, u & ((1u << k) - 1u));
}

static void sink_pad_to_byte(bitsink *s)
{
    if (s->nbits & 7) sink_put(s, (int)(8 - (s->nbits & 7)), 0);
}

/* encode.c:766-798 output_residual */
static void emit_residual(bitsink *s, const fo_subframe *sf, const int32_t *res, int n)
{
    int np = 1 << sf->porder, psize = n >> sf->porder;
    int pbits = 4 + sf->rice_method;
    int j = sf->order;
    sink_put(s, 2, (uint32_t)sf->rice_method);
    sink_put(s, 4, (uint32_t)sf->porder);
    for (int p = 0; p < np; p++) {
        int k = sf->rparams[p];
        int end = (p + 1) * psize;
        sink_put(s, pbits, (uint32_t)k);
        for (; j < end && j < n; j++) sink_put_rice(s, k, res[j]);
    }
}

int64_t fo_emit_residual(const fo_subframe *sf, const int32_t *res, int n,
                         uint8_t *out, int64_t cap_bytes)
{
    bitsink s;
    sink_init(&s, out, cap_bytes);
    emit_residual(&s, sf, res, n);
    if (s.over) return -1;
    sink_sync(&s);
    return s.nbits;
}

/* crc.c:24-94: CRC-8 poly 0x07 and CRC-16 poly 0x8005, MSB-first, init 0 */
uint8_t fo_crc8(const uint8_t *d, uint32_t len)
{
    uint8_t c = 0;
    for (uint32_t i = 0; i < len; i++) {
        c ^= d[i];
        for (int b = 0; b < 8; b++) c = (uint8_t)((c & 0x80) ? ((c << 1) ^ 0x07) : (c << 1));
    }
    return c;
}

uint16_t fo_crc16(const uint8_t *d, uint32_t len)
{
    uint16_t c = 0;
    for (uint32_t i = 0; i < len; i++) {
        c ^= (uint16_t)((uint16_t)d[i] << 8);
        for (int b = 0; b < 8; b++) c = (uint16_t)((c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1));
    }
    return c;
}

/* ------------------------------------------------------------------ */
/* frame assembly: encode.c:696-977                                   */
/* ------------------------------------------------------------------ */

static const int sr_table[16] = {0, 0, 0, 0, 8000, 16000, 22050, 24000, 32000,
                                 44100, 48000, 96000, 0, 0, 0, 0};   /* encode.c:33-37 */
static const int bd_table[8] = {0, 8, 12, 0, 16, 20, 24, 0};         /* encode.c:39-41 */
static const int bs_table[15] = {0, 192, 576, 1152, 2304, 4608, 0, 0, 256, 512,
                                 1024, 2048, 4096, 8192, 16384};      /* encode.c:43-49 */

/* encode.c:696-716 write_utf8 */
static void put_utf8(bitsink *s, uint32_t v)
{
    if (v < 0x80) { sink_put(s, 8, v); return; }
    int bytes = (ilog2_u32(v) + 4) / 5;
    int sh = (bytes - 1) * 6;
    sink_put(s, 8, (uint32_t)((256 - (256 >> bytes)) | (v >> sh)) & 0xFFu);
    while (sh >= 6) {
        sh -= 6;
        sink_put(s, 8, 0x80u | ((v >> sh) & 0x3Fu));
    }
}

/* encode.c:718-764 output_frame_header (+ the code lookups of
 * flake_encode_init :400-438 and init_frame :502-519) */
static void put_frame_header(bitsink *s, const fo_params *p, uint32_t frame_number,
                             int n, int ch_mode)
{
    int sr0 = -1, sr1 = 0, bps_code = 0, bs0 = -1, bs1 = -1;
    for (int i = 4; i < 12; i++) if (p->sample_rate == sr_table[i]) { sr0 = i; break; }
    if (sr0 < 0) {
        int sr = p->sample_rate;
        sr0 = 0;
        if (sr % 1000 == 0 && sr <= 255000) { sr0 = 12; sr1 = sr / 1000; }
        else if (sr % 10 == 0 && sr <= 655350) { sr0 = 14; sr1 = sr / 10; }
        else if (sr < 65535) { sr0 = 13; sr1 = sr; }
    }
    for (int i = 1; i < 8; i++) if (p->bits_per_sample == bd_table[i]) { bps_code = i; break; }
    for (int i = 0; i < 15; i++) if (n == bs_table[i]) { bs0 = i; break; }
    if (bs0 < 0) { bs0 = (n <= 256) ? 6 : 7; bs1 = n - 1; }

    int64_t start = s->nbits;
    sink_put(s, 15, 0x7FFC);
    sink_put(s, 1, (uint32_t)p->allow_vbs);
    sink_put(s, 4, (uint32_t)bs0);
    sink_put(s, 4, (uint32_t)sr0);
    sink_put(s, 4, (uint32_t)(ch_mode == FO_CH_NOT_STEREO ? p->channels - 1 : ch_mode));
    sink_put(s, 3, (uint32_t)bps_code);
    sink_put(s, 1, 0);
    put_utf8(s, frame_number);
    if (bs1 >= 0) sink_put(s, bs1 < 256 ? 8 : 16, (uint32_t)bs1);
    if (sr1 > 0) sink_put(s, sr1 < 256 ? 8 : 16, (uint32_t)sr1);
    if (!s->over) {
        uint8_t c = fo_crc8(s->buf + (start >> 3), (uint32_t)((s->nbits - start) >> 3));
        sink_put(s, 8, c);
    } else {
        sink_put(s, 8, 0);
    }
}

/* encode.c:800-905 output_subframes */
static void put_subframe(bitsink *s, const fo_params *p, const fo_subframe *sf,
                         const int32_t *res, int n)
{
    sink_put(s, 1, 0);
    sink_put(s, 6, (uint32_t)sf->type_code);
    if (sf->wasted) {
        sink_put(s, 1, 1);
        sink_put(s, sf->wasted - 1, 0);
        sink_put(s, 1, 1);
    } else {
        sink_put(s, 1, 0);
    }
    switch (sf->type) {
    case FO_SUB_CONSTANT:
        sink_put_signed(s, sf->obits, res[0]);
        break;
    case FO_SUB_VERBATIM:
        for (int i = 0; i < n; i++) sink_put_signed(s, sf->obits, res[i]);
        break;
    case FO_SUB_FIXED:
        for (int i = 0; i < sf->order; i++) sink_put_signed(s, sf->obits, res[i]);
        emit_residual(s, sf, res, n);
        break;
    case FO_SUB_LPC:
        for (int i = 0; i < sf->order; i++) sink_put_signed(s, sf->obits, res[i]);
        sink_put(s, 4, (uint32_t)(p->lpc_precision - 1));
        sink_put_signed(s, 5, sf->shift);
        for (int i = 0; i < sf->order; i++) sink_put_signed(s, p->lpc_precision, sf->coefs[i]);
        emit_residual(s, sf, res, n);
        break;
    }
}

static int frame_verbatim_size(const fo_params *p, int n)    /* encode.c:521-527 */
{
    int bps = p->bits_per_sample;
    if (p->channels == 2) return 16 + ((n * (bps + bps + 1) + 7) >> 3);
    return 16 + ((n * p->channels * bps + 7) >> 3);
}

/* header + subframes + footer (encode.c:944-947); returns bytes or -1 if
 * the reference writer would have hit eof in a buffer of buf_size bytes. */
static int assemble(const fo_params *p, uint32_t frame_number, int n, int ch_mode,
                    const fo_subframe *sf, const int32_t *res,
                    uint8_t *out, int buf_size)
{
    bitsink s;
    /* bitio.h:90-93: a write is refused once fewer than 4 bytes remain */
    sink_init(&s, out, (int64_t)buf_size - 3);
    put_frame_header(&s, p, frame_number, n, ch_mode);
    for (int ch = 0; ch < p->channels; ch++)
        put_subframe(&s, p, &sf[ch], res + (size_t)ch * n, n);
    sink_pad_to_byte(&s);
    if (s.over) return -1;
    uint16_t c = fo_crc16(out, (uint32_t)(s.nbits >> 3));
    sink_put(&s, 16, c);
    if (s.over) return -1;
    return (int)(s.nbits >> 3);
}

/* encode.c:919-977 encode_frame */
int fo_encode_frame(const fo_params *p, uint32_t frame_number, const int32_t *pcm,
                    int n, uint8_t *out, int buf_size,
                    fo_subframe *sf_out, int32_t *res_out, int *was_verbatim)
{
    if (!pcm || buf_size <= 0 || n < 1 || n > FO_MAX_BLOCK) return -1;
    int nch = p->channels;
    fo_subframe *sf = (fo_subframe *)calloc((size_t)nch, sizeof(fo_subframe));
    int32_t *smp = (int32_t *)malloc(sizeof(int32_t) * (size_t)nch * n);
    int32_t *res = (int32_t *)malloc(sizeof(int32_t) * (size_t)nch * n);
    int rc = -1;
    if (was_verbatim) *was_verbatim = 0;

    int mode = fo_prepare_frame(p, pcm, n, smp, sf);
    for (int ch = 0; ch < nch; ch++) {
        if (fo_encode_residual(p, &sf[ch], smp + (size_t)ch * n, res + (size_t)ch * n, n) < 0)
            goto done;
    }
    rc = assemble(p, frame_number, n, mode, sf, res, out, buf_size);
    if (rc < 0 || rc > frame_verbatim_size(p, n)) {
        /* encode.c:949-964 + optimize.c:278-289 */
        for (int ch = 0; ch < nch; ch++) {
            sf[ch].type = sf[ch].type_code = FO_SUB_VERBATIM;
            memcpy(res + (size_t)ch * n, smp + (size_t)ch * n, sizeof(int32_t) * (size_t)n);
        }
        if (was_verbatim) *was_verbatim = 1;
        rc = assemble(p, frame_number, n, mode, sf, res, out, buf_size);
    }
done:
    if (sf_out) memcpy(sf_out, sf, sizeof(fo_subframe) * (size_t)nch);
    if (res_out) memcpy(res_out, res, sizeof(int32_t) * (size_t)nch * n);
    free(sf); free(smp); free(res);
    return rc;
}

/* ------------------------------------------------------------------ */
/* vbs.c                                                              */
/* ------------------------------------------------------------------ */

/* vbs.c:36-83 split_frame_v1, including the 32-bit abs/imul wrap of the
 * threshold test (8-Q9) */
void fo_vbs_split(const int32_t *pcm, int channels, int block_size,
                  int *frames, int sizes[8])
{
    int n = block_size / 8;
    int64_t score[8];
    for (int s = 0; s < 8; s++) {
        const int32_t *base = pcm + (size_t)s * n * channels;
        int64_t acc = 0;
        for (int ch = 0; ch < channels; ch++) {
            for (int j = 2; j < n; j++) {
                int32_t a = base[(size_t)j * channels + ch];
                int32_t b = base[(size_t)(j - 1) * channels + ch];
                int32_t c = base[(size_t)(j - 2) * channels + ch];
                int32_t d2 = wrap_add(wrap_sub(a, (int32_t)(2u * (uint32_t)b)), c);
                acc += (int64_t)wrap_abs(d2);
            }
        }
        score[s] = acc / channels + 1;
    }
    int cut[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 1; s < 8; s++) {
        int32_t diff = wrap_abs((int32_t)(uint32_t)(uint64_t)(score[s - 1] - score[s]));
        int32_t scaled = (int32_t)((uint32_t)diff * 200u);
        if ((int64_t)scaled / score[s - 1] > 50) cut[s] = 1;
    }
    int nf = 0;
    for (int s = 0; s < 8; s++) sizes[s] = 0;
    for (int s = 0; s < 8; s++) {
        if (cut[s]) nf++;
        sizes[nf - 1] += n;
    }
    *frames = nf;
}

/* encode.c:979-1008 flake_encode_frame (without MD5) + vbs.c:85-119 */
int fo_encode_block(const fo_params *p, uint32_t *frame_count, const int32_t *pcm,
                    int block_size, uint8_t *out, int buf_size)
{
    int fs = -1;
    if (block_size < 1 || block_size > p->block_size) return -1;
    if (p->variable_block_size > 0 && (block_size % 8) == 0 && block_size >= 8 * 16) {
        int nf, sizes[8];
        uint32_t fc0 = *frame_count;
        fo_vbs_split(pcm, p->channels, block_size, &nf, sizes);
        if (nf > 1) {
            int fpos = 0, spos = 0, ok = 1;
            for (int i = 0; i < nf; i++) {
                int r = fo_encode_frame(p, *frame_count, pcm + (size_t)spos * p->channels,
                                        sizes[i], out + fpos, buf_size - fpos, NULL, NULL, NULL);
                if (r < 0) { ok = 0; break; }
                *frame_count += p->allow_vbs ? (uint32_t)sizes[i] : 1u;
                fpos += r;
                spos += sizes[i];
            }
            if (ok) return fpos;
            *frame_count = fc0;
        }
    }
    fs = fo_encode_frame(p, *frame_count, pcm, block_size, out, buf_size, NULL, NULL, NULL);
    if (fs >= 0) *frame_count += p->allow_vbs ? (uint32_t)block_size : 1u;
    return fs;
}

/* ------------------------------------------------------------------ */
/* hot path over a batch (the cpu_baseline unit of work)              */
/* ------------------------------------------------------------------ */
int fo_encode_subframes_batch(const fo_params *p, const int32_t *pcm, int nframes,
                              int n, fo_subframe *sf, int32_t *res,
                              uint8_t *bits, int64_t slot_bytes)
{
    int nch = p->channels;
    int32_t *smp = (int32_t *)malloc(sizeof(int32_t) * (size_t)nch * n);
    int32_t *tmp = res ? NULL : (int32_t *)malloc(sizeof(int32_t) * (size_t)nch * n);
    for (int f = 0; f < nframes; f++) {
        fo_subframe *fsf = sf + (size_t)f * nch;
        int32_t *fres = res ? res + (size_t)f * nch * n : tmp;
        memset(fsf, 0, sizeof(fo_subframe) * (size_t)nch);
        fo_prepare_frame(p, pcm + (size_t)f * n * nch, n, smp, fsf);
        for (int ch = 0; ch < nch; ch++) {
            const int32_t *r = fres + (size_t)ch * n;
            if (fo_encode_residual(p, &fsf[ch], smp + (size_t)ch * n,
                                   fres + (size_t)ch * n, n) < 0) {
                free(smp); free(tmp);
                return -1;
            }
            if (fsf[ch].type != FO_SUB_FIXED && fsf[ch].type != FO_SUB_LPC) continue;
            if (bits) {
                int64_t nb = fo_emit_residual(&fsf[ch], r, n,
                        bits + ((size_t)f * nch + ch) * (size_t)slot_bytes, slot_bytes);
                fsf[ch].rice_nbits = (int32_t)nb;   /* -1: slot too small */
            } else {
                fsf[ch].rice_nbits = (int32_t)fo_residual_section_bits(&fsf[ch], r, n);
            }
        }
    }
    free(smp); free(tmp);
    return 0;
}
