// k2_lpc.hip -- K2: compute_lpc_coefs / _est + quantize_lpc_coefs (lpc.c:77-257) as its
// own launch (order searches, orders above 12, batches the wave-typed K1 does not serve).
#include "device_util.h"
#include "lpc_reg.h"

namespace fhip {
namespace {

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
           int32_t *__restrict__ opt_order, int32_t *__restrict__ fin, const int32_t *__restrict__ dev_sub,
           MultiBin mb)
{
    __shared__ double s_mem[LPC_DBL * LPC_NT];
    const int lane = threadIdx.x;
    const int s = blockIdx.x * LPC_NT + lane;
    if (mb.nbins) { if (s >= nsub || !bin_unit_live(mb, s)) return; }       // every bin of a ragged batch at once
    else { nsub = dev_count(dev_sub, nsub); if (s >= nsub) return; }

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
        double *fd = reinterpret_cast<double *>(f + FIN_DBL);     // the first 16 as doubles (K3 reads them as scalars)
        for (int j = 0; j < 16; j++) fd[j] = (j < levinson_order && j < max_order) ? (double)src[j] : 0.0;
        int32_t cabs = 0, c8[8];
        for (int j = 0; j < 8; j++) c8[j] = (j < levinson_order && j < max_order) ? src[j] : 0;
        for (int j = 0; j < levinson_order; j++) cabs += (src[j] < 0) ? -src[j] : src[j];
        f[34] = cabs;
        for (int j = 0; j < 4; j++) f[FIN_PAIRS + j] = (c8[2 * j + 1] & 0xFFFF) | (int32_t)((uint32_t)c8[2 * j] << 16);
    }
}
template <int MO>
__global__ __launch_bounds__(LPC_NT)
void k_lpc_reg(const double *__restrict__ autoc_all, int nsub, int max_order, int precision,
               int omethod, int32_t *__restrict__ coefs, int32_t *__restrict__ shift,
               int32_t *__restrict__ opt_order, int32_t *__restrict__ fin, const int32_t *__restrict__ dev_sub,
               MultiBin mb)
{
    const int s = blockIdx.x * LPC_NT + threadIdx.x;
    if (mb.nbins) { if (s >= nsub || !bin_unit_live(mb, s)) return; }
    else { nsub = dev_count(dev_sub, nsub); if (s >= nsub) return; }
    double ac[MO + 1];
#pragma unroll
    for (int i = 0; i <= MO; i++) ac[i] = (i <= max_order) ? autoc_all[(size_t)s * FHIP_MAX_LAGS + i] : 0.0;
    lpc_reg_one<MO>(ac, s, max_order, precision, omethod, coefs, shift, opt_order, fin);
}

// ---------------------------------------------------------------------------
// K2  k_lpc_rows: every row of a large-order search (orders 13 .. 32)
// ---------------------------------------------------------------------------
// The order searches quantise all max_order rows (lpc.c:249-254): 528 error-feedback steps
// for order 32, each a dependent chain, on top of ~800 dependent Levinson steps -- with one
// lane per subframe and the arrays in LDS (k_lpc) that is 113 us of pure latency for 8192
// subframes.  Here a workgroup owns 32 subframes:
//   phase A  lanes 0..31 of wave 0 run Levinson-Durbin (lpc.c:77-117) with autoc[] and
//            lpc_tmp[] in registers (loops unrolled to compile-time indices) and leave every
//            row lpc[i][0..i] in LDS;
//   phase B  the rows are independent of each other: 512 threads quantise the 32 x 32 rows
//            (lpc.c:167-219), two each.
constexpr int LR_NT = 512;
constexpr int LR_SUB = 32;
constexpr int LR_STRIDE = 529;          // 528 doubles of rows per subframe, odd: lanes on distinct banks

__device__ __forceinline__ void quantize_row_lds(const double *__restrict__ a, int order, int precision,
                                                 int32_t *__restrict__ out, int32_t *__restrict__ shift_out)
{
    const int qmax = (1 << (precision - 1)) - 1;
    double cmax = 0.0;
    for (int j = 0; j < order; j++) {
        const double m = fabs(a[j]);
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
        const double t = v * mul;
        carry = carry + t;
        int q = c_double_to_int(carry + 0.5);
        if (q <= -qmax) q = -qmax + 1;
        if (q > qmax) q = qmax;
        carry = carry - (double)q;
        out[j] = q;
    }
    *shift_out = sh;
}

__global__ __launch_bounds__(LR_NT)
void k_lpc_rows(const double *__restrict__ autoc_all, int nsub, int max_order, int precision,
                int32_t *__restrict__ coefs, int32_t *__restrict__ shift,
                int32_t *__restrict__ opt_order, const int32_t *__restrict__ dev_sub, MultiBin mb)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double *rows = reinterpret_cast<double *>(lds_raw);              // [LR_SUB][LR_STRIDE]
    constexpr int MO = FHIP_MAX_ORDER;
    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * LR_SUB;
    if (!mb.nbins) nsub = dev_count(dev_sub, nsub);
    if (s0 >= nsub) return;
    // (every bin of a ragged batch at once: nsub is the workspaces' capacity, a unit is live
    // when its bin's count says so)
    auto live = [&](int s) { return s < nsub && (!mb.nbins || bin_unit_live(mb, s)); };

    if (tid < LR_SUB && live(s0 + tid)) {
        const int s = s0 + tid;
        double ac[MO + 1];
#pragma unroll
        for (int i = 0; i <= MO; i++) ac[i] = (i <= max_order) ? autoc_all[(size_t)s * FHIP_MAX_LAGS + i] : 0.0;
        double a[MO];
#pragma unroll
        for (int i = 0; i < MO; i++) a[i] = 0.0;
        double err = ac[0];
        double *mine = rows + tid * LR_STRIDE;
#pragma unroll
        for (int i = 0; i < MO; i++) {
            if (i < max_order) {
                double r = -ac[i + 1];
#pragma unroll
                for (int j = 0; j < i; j++) {
                    const double t = a[j] * ac[i - j];
                    r = r - t;
                }
                r = r / err;
                const double rr = r * r;
                const double om = 1.0 - rr;
                err = err * om;
                a[i] = r;
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
                    const double m = a[i >> 1];
                    const double t = m * r;
                    a[i >> 1] = m + t;
                }
#pragma unroll
                for (int j = 0; j <= i; j++) mine[i * (i + 1) / 2 + j] = a[j];
            }
        }
        opt_order[s] = max_order;
    }
    __syncthreads();

    for (int item = tid; item < LR_SUB * MO; item += LR_NT) {
        const int sub = item >> 5, row = item & 31;
        const int s = s0 + sub;
        if (!live(s) || row >= max_order) continue;
        quantize_row_lds(rows + sub * LR_STRIDE + row * (row + 1) / 2, row + 1, precision,
                         coefs + ((size_t)s * FHIP_MAX_ORDER + row) * FHIP_MAX_ORDER,
                         shift + (size_t)s * FHIP_MAX_ORDER + row);
    }
}

}  // namespace

static hipError_t launch_lpc_any(hipStream_t st, const double *autoc, int nsub, int max_order,
                                 int precision, int omethod, int32_t *coefs, int32_t *shift,
                                 int32_t *opt_order, int32_t *fin, const int32_t *dev_sub, const MultiBin &mb);

hipError_t launch_lpc(hipStream_t st, const double *autoc, int nsub, int max_order,
                      int precision, int omethod, int32_t *coefs, int32_t *shift,
                      int32_t *opt_order, int32_t *fin, const int32_t *dev_sub)
{
    return launch_lpc_any(st, autoc, nsub, max_order, precision, omethod, coefs, shift, opt_order, fin, dev_sub,
                          MultiBin{});
}

// K2 for every bin of a ragged batch at once (the recursion does not depend on the block size):
// the grid covers the workspaces' whole unit range, a unit is live when its bin's count says so.
hipError_t launch_lpc_bins(hipStream_t st, const MultiBin &mb, const double *autoc, int max_order,
                           int precision, int omethod, int32_t *coefs, int32_t *shift,
                           int32_t *opt_order, int32_t *fin)
{
    if (mb.nbins < 1) return hipErrorInvalidValue;
    const int units = mb.unit0[mb.nbins - 1] + mb.cap[mb.nbins - 1];
    return launch_lpc_any(st, autoc, units, max_order, precision, omethod, coefs, shift, opt_order, fin, nullptr, mb);
}

static hipError_t launch_lpc_any(hipStream_t st, const double *autoc, int nsub, int max_order,
                                 int precision, int omethod, int32_t *coefs, int32_t *shift,
                                 int32_t *opt_order, int32_t *fin, const int32_t *dev_sub, const MultiBin &mb)
{
    if (nsub == 0) return hipSuccess;
    const int blocks = (nsub + LPC_NT - 1) / LPC_NT;
    if (max_order <= 8)
        hipLaunchKernelGGL(k_lpc_reg<8>, dim3(blocks), dim3(LPC_NT), 0, st, autoc, nsub, max_order,
                           precision, omethod, coefs, shift, opt_order, fin, dev_sub, mb);
    else if (max_order <= 12)
        hipLaunchKernelGGL(k_lpc_reg<12>, dim3(blocks), dim3(LPC_NT), 0, st, autoc, nsub, max_order,
                           precision, omethod, coefs, shift, opt_order, fin, dev_sub, mb);
    else if (omethod >= 2 && getenv("FHIP_K2_ONE_LANE") == nullptr) {
        // every row wanted (lpc.c:249-254): Levinson in registers, rows quantised side by side
        const size_t lds = (size_t)LR_SUB * LR_STRIDE * sizeof(double);
        hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lpc_rows),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (er != hipSuccess) return er;
        hipLaunchKernelGGL(k_lpc_rows, dim3((nsub + LR_SUB - 1) / LR_SUB), dim3(LR_NT), lds, st, autoc, nsub,
                           max_order, precision, coefs, shift, opt_order, dev_sub, mb);
    } else
        hipLaunchKernelGGL(k_lpc, dim3(blocks), dim3(LPC_NT), 0, st, autoc, nsub, max_order,
                           precision, omethod, coefs, shift, opt_order, fin, dev_sub, mb);
    return hipGetLastError();
}

}  // namespace fhip
