/*
 * ref_harness.c -- thin exports around the REAL reference sources.
 *
 * TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it
 * #includes the reference translation units where they lie under
 * /root/reference (lpc.c, rice.c, crc.c, bitio.h) so that their `static`
 * functions become reachable, and re-exports them under ref_* names with
 * plain-pointer signatures for ctypes.  Built only by oracle/Makefile into
 * oracle/_ref/ (git-ignored) and only when /root/reference exists.
 *
 * Not buildable here (and therefore not wrapped): optimize.c, encode.c,
 * vbs.c -- they include encode.h, which includes the CMake-generated
 * config.h.  rice.c includes encode.h too but uses nothing from it except
 * what flake.h declares; the Makefile predefines encode.h's include guard
 * (-DFLAC_H) and force-includes the reference's own flake.h instead.
 */
#include <stdint.h>
#include <string.h>

#include "lpc.c"      /* /root/reference/libflake/lpc.c  */
#include "rice.c"     /* /root/reference/libflake/rice.c */
#include "crc.c"      /* /root/reference/libflake/crc.c  */
#include "bitio.h"    /* /root/reference/libflake/bitio.h */

#define EXPORT __attribute__((visibility("default")))

EXPORT void ref_compute_autocorr(const int32_t *data, int len, int lag, double *autoc)
{
    compute_autocorr(data, len, lag, autoc);
}

EXPORT void ref_compute_lpc_coefs(const double *autoc, int max_order, double *ref,
                                  double *lpc /* [32][32] */)
{
    compute_lpc_coefs(autoc, max_order, ref, (double (*)[MAX_LPC_ORDER])lpc);
}

EXPORT int ref_compute_lpc_coefs_est(const double *autoc, int max_order, double *lpc)
{
    return compute_lpc_coefs_est(autoc, max_order, (double (*)[MAX_LPC_ORDER])lpc);
}

EXPORT void ref_quantize_lpc_coefs(double *lpc_in, int order, int precision,
                                   int32_t *lpc_out, int *shift)
{
    quantize_lpc_coefs(lpc_in, order, precision, lpc_out, shift);
}

EXPORT int ref_lpc_calc_coefs(const int32_t *samples, int blocksize, int max_order,
                              int precision, int omethod, int32_t *coefs, int *shift)
{
    return lpc_calc_coefs(samples, blocksize, max_order, precision, omethod,
                          (int32_t (*)[MAX_LPC_ORDER])coefs, shift);
}

EXPORT int ref_find_optimal_rice_param(uint64_t sum, int n)
{
    return find_optimal_rice_param(sum, n);
}

/* out: method, porder, params[256] */
EXPORT uint32_t ref_calc_rice_params(int lpc, int pmin, int pmax, int32_t *data, int n,
                                     int pred_order, int bps, int precision,
                                     int *method, int *porder, int *params)
{
    RiceContext rc;
    uint32_t bits;
    memset(&rc, 0, sizeof(rc));
    if (lpc)
        bits = calc_rice_params_lpc(&rc, pmin, pmax, data, n, pred_order, bps, precision);
    else
        bits = calc_rice_params_fixed(&rc, pmin, pmax, data, n, pred_order, bps);
    *method = rc.method;
    *porder = rc.porder;
    memcpy(params, rc.params, sizeof(int) * MAX_PARTITIONS);
    return bits;
}

EXPORT uint64_t ref_rice_encode_count(uint64_t sum, int n, int k)
{
    return rice_encode_count(sum, n, k);
}

EXPORT int ref_limit_max_partition_order(int max_porder, int n, int order)
{
    return limit_max_partition_order(max_porder, n, order);
}

EXPORT int ref_log2i(uint32_t v) { return log2i(v); }

/* Residual section driven through the reference BitWriter.  The loop shape is
 * that of encode.c:766-798 (not buildable here); every bit goes through the
 * real bitwriter_writebits / bitwriter_write_rice_signed.  Returns bytes
 * (bitwriter_count after flush) or -1 on eof; *nbits gets the bit length. */
EXPORT int ref_emit_residual(int method, int porder, const int *params, int order,
                             const int32_t *residual, int n, uint8_t *out, int cap,
                             int64_t *nbits)
{
    BitWriter bw;
    int p, i, j, psize, res_cnt;
    int64_t bits = 6;
    bitwriter_init(&bw, out, cap);
    bitwriter_writebits(&bw, 2, method);
    bitwriter_writebits(&bw, 4, porder);
    psize = n >> porder;
    res_cnt = psize - order;
    j = order;
    for (p = 0; p < (1 << porder); p++) {
        int k = params[p];
        bitwriter_writebits(&bw, 4 + method, k);
        bits += 4 + method;
        for (i = 0; i < res_cnt && j < n; i++, j++) {
            int v = -2 * residual[j] - 1;
            v ^= (v >> 31);
            bits += (v >> k) + 1 + k;
            bitwriter_write_rice_signed(&bw, k, residual[j]);
        }
        res_cnt = psize;
    }
    bitwriter_flush(&bw);
    if (nbits) *nbits = bits;
    if (bw.eof) return -1;
    return bitwriter_count(&bw);
}

/* generic writer exercise: ops[i] = {nbits, value}; signed_mask bit i set ->
 * bitwriter_writebits_signed */
EXPORT int ref_bitwriter_run(const int *nbits, const int32_t *vals, const uint8_t *is_signed,
                             int nops, uint8_t *out, int cap)
{
    BitWriter bw;
    int i;
    bitwriter_init(&bw, out, cap);
    for (i = 0; i < nops; i++) {
        if (is_signed[i]) bitwriter_writebits_signed(&bw, nbits[i], vals[i]);
        else bitwriter_writebits(&bw, nbits[i], (uint32_t)vals[i]);
    }
    bitwriter_flush(&bw);
    if (bw.eof) return -1;
    return bitwriter_count(&bw);
}

EXPORT int ref_crc8(const uint8_t *d, uint32_t len)  { crc_init(); return calc_crc8(d, len); }
EXPORT int ref_crc16(const uint8_t *d, uint32_t len) { crc_init(); return calc_crc16(d, len); }

/* ------------------------------------------------------------------------
 * Timing leg for bench.py's cpu_baseline: the hot path of one prepared subframe
 * batch (LPC, order method MAX) with every stage that CAN be compiled here
 * running as the reference's own code -- lpc_calc_coefs (lpc.c:224),
 * calc_rice_params_lpc (rice.c:180), the BitWriter + bitwriter_write_rice_signed
 * (bitio.h), calc_crc16 (crc.c) -- and the one stage that cannot (optimize.c needs
 * the generated config.h) as a plain loop of this file: the integer FIR.
 * t[0..4] receive seconds spent in lpc / fir / rice / emit / crc16.
 * Returns the total number of residual-section bits (cross-checked by the caller
 * against the HIP path's rice_nbits), or -1. */
#include <time.h>

static volatile unsigned crc_sink;      /* keeps the CRC pass from being optimised away */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

EXPORT int64_t ref_time_hotpath(const int32_t *smp, const int *obits, int nsub, int n,
                                int max_order, int precision, int pmin, int pmax,
                                int32_t *res, uint8_t *out, int cap, double *t)
{
    int32_t coefs[MAX_LPC_ORDER][MAX_LPC_ORDER];
    int shift[MAX_LPC_ORDER];
    int64_t total = 0;
    int s, i, j;
    crc_init();
    for (i = 0; i < 5; i++) t[i] = 0.0;
    for (s = 0; s < nsub; s++) {
        const int32_t *x = smp + (size_t)s * n;
        RiceContext rc;
        int64_t nbits;
        int bytes;
        double t0 = now_s(), t1;
        lpc_calc_coefs(x, n, max_order, precision, FLAKE_ORDER_METHOD_MAX, coefs, shift);
        t1 = now_s(); t[0] += t1 - t0; t0 = t1;
        {
            const int32_t *c = coefs[max_order - 1];
            int sh = shift[max_order - 1];
            for (i = 0; i < max_order; i++) res[i] = x[i];
            for (i = max_order; i < n; i++) {
                int64_t acc = 0;
                for (j = max_order; j >= 1; j--) acc += (int64_t)c[j - 1] * x[i - j];
                res[i] = (int32_t)(x[i] - (acc >> sh));
            }
        }
        t1 = now_s(); t[1] += t1 - t0; t0 = t1;
        memset(&rc, 0, sizeof(rc));
        calc_rice_params_lpc(&rc, pmin, pmax, res, n, max_order, obits[s], precision);
        t1 = now_s(); t[2] += t1 - t0; t0 = t1;
        bytes = ref_emit_residual(rc.method, rc.porder, rc.params, max_order, res, n, out, cap, &nbits);
        t1 = now_s(); t[3] += t1 - t0; t0 = t1;
        if (bytes < 0) return -1;
        crc_sink ^= calc_crc16(out, bytes);
        t1 = now_s(); t[4] += t1 - t0;
        total += nbits;
    }
    return total;
}
