// lpc_reg.h -- K2 for one subframe with every array in registers (lpc.c:77-257),
// shared by k_lpc_reg (k2_lpc.hip) and the tail of k_autocorr_wt (k1_autocorr.hip).
#pragma once

#include "device_util.h"

namespace fhip {
namespace {

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
            for (int j = 0; j < 16; j++) fd[j] = 0.0;
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
                if (j < 16) reinterpret_cast<double *>(fin_out + FIN_DBL)[j] = (double)q;
            }
        } else if (fin_out) {
            if (j < max_order) fin_out[j] = 0;
            if (j < 16) reinterpret_cast<double *>(fin_out + FIN_DBL)[j] = 0.0;
        }
    }
    if (fin_out) {
#pragma unroll
        for (int j = MO; j < 16; j++) reinterpret_cast<double *>(fin_out + FIN_DBL)[j] = 0.0;
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

}  // namespace
}  // namespace fhip
