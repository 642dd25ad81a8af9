// Diagnostic (not part of the product): the order search's FIRs on the int8 matrix pipe, priced
// before built.  pred[cand][i] = sum_t coef[cand][t] * x[i - t] for 32 candidate
// orders at once as exact integer matrix products: samples (< 2^23) as three balanced int8 limbs,
// coefficients (< 2^14) as two, the limb-pair products of equal weight 2^(8w) paired along K = 64
// of v_mfma_i32_16x16x64_i8 (M = 16 samples, N = 16 candidates), int32 sums exact.
//   1. checks the operand / result lane maps of the instruction with random data
//   2. checks the whole arithmetic (limbs, weights, floor shift, fold, leaf sums) against the CPU
//   3. times one workgroup-shaped pass (4 waves x 64 tiles, 32 candidates) with s_memtime and the
//      wall time of a chip-filling launch
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_i8_probe.hip -o tools/bin/mfma_i8_probe && tools/bin/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------- 1. lane maps
__global__ void k_layout(const signed char *A, const signed char *B, int *D)   // A[16][64], B[64][16], D[16][16]
{
    const int l = threadIdx.x;
    // assumed: lane l holds A[m = l & 15][k = 16 (l >> 4) + j], B[k = 16 (l >> 4) + j][n = l & 15], j = 0..15
    v4i a, b, c = {0, 0, 0, 0};
    signed char ab[16], bb[16];
    for (int j = 0; j < 16; j++) { ab[j] = A[(l & 15) * 64 + 16 * (l >> 4) + j]; bb[j] = B[(16 * (l >> 4) + j) * 16 + (l & 15)]; }
    memcpy(&a, ab, 16); memcpy(&b, bb, 16);
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    // assumed: D[row = 4 (l >> 4) + r][col = l & 15]
    for (int r = 0; r < 4; r++) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

// ---------------------------------------------------------------- 2./3. the FIR of one subframe
constexpr int N = 4096, HIST = 32, NC = 32, T = 256;
struct Args { const int *x; const int *coef; const int *shift; const int *order; unsigned *leaf; long long *cyc; };

// limb planes: bytes, plane b at lds + b * PLANE, sample i at [HIST + i]; HIST zeros in front
constexpr int PLANE = N + HIST + 16;

typedef const v4i __attribute__((address_space(3))) *lds_v4;
typedef const int __attribute__((address_space(3))) *lds_i;

// Tile = 16 samples SPACED 16 APART inside a block of 256 (rows m: samples i0 + o + 16 m, o = 0 .. 15 the
// tile's offset): every lane of a tile then reads its 16 operand bytes at the same misalignment o -- two
// aligned 16-byte reads and a funnel shift by a compile-time amount -- and the 16 tiles of a block leave
// each lane with the sums of its own four LEAVES (runs of 16 consecutive samples, rows 4 g + r) per
// candidate: no cross-lane reduction.
template <bool TIMING>
__global__ __launch_bounds__(T, 3) void k_fir(Args a, int nsub)
{
    extern __shared__ __attribute__((aligned(16))) signed char lds[];
    signed char *plane = lds;                                   // [3][PLANE]
    int *ximg = (int *)(lds + 3 * PLANE);                       // [N] int32 (the product has its transposed image)
    signed char *cl = (signed char *)(ximg + N);                // coefficient limbs [2 limbs][NC][32 taps], reversed in 16-groups
    int *shs = (int *)(cl + 2 * NC * 32);                       // [NC] shift
    int *ord = shs + NC;                                        // [NC]
    unsigned *leafs = (unsigned *)(ord + NC);                   // [NC][T] leaf sums (what the Rice search reads)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int s = blockIdx.x % nsub;
    const int *x = a.x + (size_t)s * N;
    // ---- stage: int32 image + three balanced limbs per sample ----
    for (int i = tid; i < N; i += T) {
        const int v = x[i];
        ximg[i] = v;
        const int x0 = (int)(signed char)v;
        const int r1 = (v - x0) >> 8;
        const int x1 = (int)(signed char)r1;
        const int x2 = (r1 - x1) >> 8;
        plane[0 * PLANE + HIST + i] = (signed char)x0;
        plane[1 * PLANE + HIST + i] = (signed char)x1;
        plane[2 * PLANE + HIST + i] = (signed char)x2;
    }
    if (tid < HIST) for (int b = 0; b < 3; b++) plane[b * PLANE + tid] = 0;
    for (int q = tid; q < NC * 32; q += T) {
        const int c = q >> 5, t = q & 31;                       // tap t + 1 of candidate c
        const int v = (t < a.order[c]) ? a.coef[c * 32 + t] : 0;
        const int c0 = (int)(signed char)v, c1 = (v - c0) >> 8;
        // K chunk h = t >> 4 holds taps 16 h + 16 .. 16 h + 1 at bytes 0 .. 15 (sample bytes ascend)
        const int h = t >> 4, j = 15 - (t & 15);
        cl[(0 * NC + c) * 32 + 16 * h + j] = (signed char)c0;
        cl[(1 * NC + c) * 32 + 16 * h + j] = (signed char)c1;
    }
    if (tid < NC) { shs[tid] = a.shift[tid]; ord[tid] = a.order[tid]; }
    __syncthreads();
    // ---- B operands (coefficients), resident: [cand tile][weight] ----
    // weight w pairs (c0, x_w) on K chunks 0,1 with (c1, x_{w-1}) on chunks 2,3
    const int g = lane >> 4, n = lane & 15;
    v4i Bop[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int limb = g >> 1;                             // chunks 0,1: c0; 2,3: c1
            const bool on = (limb == 0) ? (w <= 2) : (w >= 1);   // (c0, x_w) exists for w <= 2, (c1, x_{w-1}) for w >= 1
            v4i v = *(const v4i *)(cl + (limb * NC + ct * 16 + n) * 32 + 16 * (g & 1));
            if (!on) v = v4i{0, 0, 0, 0};
            Bop[ct][w] = v;
        }
    int mysh[2], mysh16[2], myord[2];
#pragma unroll
    for (int ct = 0; ct < 2; ct++) { mysh[ct] = shs[ct * 16 + n]; mysh16[ct] = 16 - mysh[ct]; myord[ct] = ord[ct * 16 + n]; }

    long long t0 = 0;
    if (TIMING) t0 = __builtin_amdgcn_s_memtime();
    const int h = g & 1, m = n;
    // byte offset (LDS address) of this lane's operand rows: plane pl at + pl * PLANE
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) signed char *)plane;
    const unsigned xbase = (unsigned)(size_t)(__attribute__((address_space(3))) int *)ximg;
#pragma unroll 1
    for (int blk = 0; blk < 4; blk++) {
        const int i0 = wv * 1024 + blk * 256;
        const bool first = (i0 == 0);                            // only the subframe's first block holds warm-up samples
        unsigned acc[2][4];
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[ct][r] = 0;
        const unsigned rowoff = (unsigned)(HIST + i0 + 16 * m - 16 * (h + 1));      // 16-byte aligned
#pragma unroll
        for (int o4 = 0; o4 < 16; o4 += 4) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int o = o4; o < o4 + 4; o++) {
            // A operands: bytes [o, o + 16) behind rowoff of plane x_w (chunks 0,1) / x_{w-1} (chunks 2,3)
            v4i Aop[4];
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int pl = (g >> 1) == 0 ? w : w - 1;
                const int plc = pl < 0 ? 0 : (pl > 2 ? 2 : pl);
                const unsigned ad = lbase + (unsigned)plc * PLANE + rowoff;
                const v4i R0 = *(lds_v4)(size_t)ad;
                v4i R1 = R0;
                if (o) R1 = *(lds_v4)(size_t)(ad + 16);
                const int W[8] = {R0.x, R0.y, R0.z, R0.w, R1.x, R1.y, R1.z, R1.w};
                const int aa = o >> 2, bb = o & 3;
                if (bb == 0) Aop[w] = v4i{W[aa], W[aa + 1], W[aa + 2], W[aa + 3]};
                else Aop[w] = v4i{(int)__builtin_amdgcn_alignbyte(W[aa + 1], W[aa], bb), (int)__builtin_amdgcn_alignbyte(W[aa + 2], W[aa + 1], bb),
                                  (int)__builtin_amdgcn_alignbyte(W[aa + 3], W[aa + 2], bb), (int)__builtin_amdgcn_alignbyte(W[aa + 4], W[aa + 3], bb)};
            }
            int xs[4];
#pragma unroll
            for (int r = 0; r < 4; r++) xs[r] = *(lds_i)(size_t)(xbase + 4u * (unsigned)(i0 + o + 64 * g + 16 * r));
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                v4i P[4];
#pragma unroll
                for (int w = 0; w < 4; w++) P[w] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aop[w], Bop[ct][w], v4i{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    // pred = lo + 2^16 hi with lo = P0 + 2^8 P1, hi = P2 + 2^8 P3 (both exact in int32);
                    // pred >> shift (shift <= 16) = (hi << (16 - shift)) + (lo >> shift), low 32 bits
                    const int lo = P[0][r] + (P[1][r] << 8), hi = P[2][r] + (P[3][r] << 8);
                    const unsigned q = ((unsigned)hi << mysh16[ct]) + (unsigned)(lo >> mysh[ct]);
                    const int res = (int)((unsigned)xs[r] - q);
                    const unsigned u = ((unsigned)res << 1) ^ (unsigned)(res >> 31);
                    if (first) acc[ct][r] += (i0 + o + 64 * g + 16 * r < myord[ct]) ? 0u : u;    // warm-up samples are not part of partition 0
                    else acc[ct][r] += u;
                }
            }
        }
        }
        // leaves 16 (wv * 4 + blk) + 4 g + r of candidate ct * 16 + n
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
            *(uint4 *)(leafs + (ct * 16 + n) * T + 16 * (wv * 4 + blk) + 4 * g) = make_uint4(acc[ct][0], acc[ct][1], acc[ct][2], acc[ct][3]);
    }
    if (TIMING && tid == 0) a.cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    __syncthreads();
    for (int q = tid; q < NC * T; q += T) a.leaf[(size_t)blockIdx.x * NC * T + q] = leafs[q];
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    srand(7);
    // ---- 1. lane maps ----
    {
        std::vector<signed char> A(16 * 64), B(64 * 16);
        for (auto &v : A) v = (signed char)(rand() % 255 - 127);
        for (auto &v : B) v = (signed char)(rand() % 255 - 127);
        signed char *dA, *dB; int *dD;
        CK(hipMalloc(&dA, A.size())); CK(hipMalloc(&dB, B.size())); CK(hipMalloc(&dD, 256 * 4));
        CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        int D[256]; CK(hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
            int e = 0; for (int k = 0; k < 64; k++) e += (int)A[i * 64 + k] * (int)B[k * 16 + j];
            bad += e != D[i * 16 + j];
        }
        printf("1. v_mfma_i32_16x16x64_i8 lane maps (A[l&15][16(l>>4)+j], B[16(l>>4)+j][l&15], D[4(l>>4)+r][l&15]): %s (%d wrong)\n",
               bad ? "WRONG" : "confirmed", bad);
    }
    // ---- 2. arithmetic against the CPU ----
    const int nsub = 8;
    std::vector<int> x((size_t)nsub * N), coef(NC * 32), shift(NC), order(NC);
    for (int s = 0; s < nsub; s++) {
        long long y1 = 0, y2 = 0;
        for (int i = 0; i < N; i++) {
            long long e = (rand() % 65536 - 32768) * 8;
            long long y = ((30000 * y1 - 16100 * y2) >> 14) + e;
            if (y > 8355711) y = 8355711; if (y < -8388608) y = -8388608;     // (the balanced top limb: samples above 2^23 - 32897 need a fourth)
            y2 = y1; y1 = y; x[(size_t)s * N + i] = (int)y;
        }
    }
    for (int c = 0; c < NC; c++) {
        order[c] = c + 1; shift[c] = c % 16;
        for (int t = 0; t < 32; t++) coef[c * 32 + t] = rand() % 32767 - 16383;
    }
    int *dx, *dc, *ds, *dor; unsigned *dleaf; long long *dcyc;
    const int big = 256 * 3 * 8;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dc, coef.size() * 4)); CK(hipMalloc(&ds, NC * 4)); CK(hipMalloc(&dor, NC * 4));
    CK(hipMalloc(&dleaf, (size_t)big * NC * T * 4)); CK(hipMalloc(&dcyc, big * 8));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, shift.data(), NC * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dor, order.data(), NC * 4, hipMemcpyHostToDevice));
    Args a{dx, dc, ds, dor, dleaf, dcyc};
    const size_t ldsb = 3 * PLANE + N * 4 + 2 * NC * 32 + NC * 4 + NC * 4 + NC * T * 4 + 64;
    CK(hipFuncSetAttribute((const void *)&k_fir<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    CK(hipFuncSetAttribute((const void *)&k_fir<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipLaunchKernelGGL(k_fir<false>, dim3(nsub), dim3(T), ldsb, 0, a, nsub);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> leaf((size_t)nsub * NC * T);
    CK(hipMemcpy(leaf.data(), dleaf, leaf.size() * 4, hipMemcpyDeviceToHost));
    long long bad = 0;
    for (int s = 0; s < nsub; s++) for (int c = 0; c < NC; c++) for (int tl = 0; tl < T; tl++) {
        unsigned e = 0;
        for (int i = tl * 16; i < tl * 16 + 16; i++) {
            if (i < order[c]) continue;
            long long pred = 0;
            for (int t = 1; t <= order[c]; t++) pred += (long long)coef[c * 32 + t - 1] * (i - t >= 0 ? x[(size_t)s * N + i - t] : 0);
            const int res = (int)((long long)x[(size_t)s * N + i] - (pred >> shift[c]));
            e += ((unsigned)res << 1) ^ (unsigned)(res >> 31);
        }
        bad += e != leaf[((size_t)s * NC + c) * T + tl];
    }
    printf("2. leaf sums of 32 candidate orders x %d subframes of 24-bit samples against the CPU: %s (%lld wrong of %zu)\n",
           nsub, bad ? "WRONG" : "bit-exact", bad, leaf.size());
    // ---- 3. timing ----
    hipLaunchKernelGGL(k_fir<true>, dim3(1), dim3(T), ldsb, 0, a, nsub);
    long long c1; CK(hipMemcpy(&c1, dcyc, 8, hipMemcpyDeviceToHost));
    printf("3. one workgroup alone: %lld memtime ticks for 64 tiles per wave = %.0f per tile (16 samples x 32 candidates x 32 taps)\n", c1, c1 / 64.0);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {256, 256 * 3, 256 * 3 * 8}) {
        hipLaunchKernelGGL(k_fir<true>, dim3(blocks), dim3(T), ldsb, 0, a, nsub);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_fir<true>, dim3(blocks), dim3(T), ldsb, 0, a, nsub);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> cy(blocks); CK(hipMemcpy(cy.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost));
        double av = 0; for (auto v : cy) av += v; av /= blocks;
        printf("   %5d workgroups: %.3f ms wall = %.1f us per 256 subframes-worth (8192 subframes: %.3f ms); tile loop %.0f ticks per workgroup\n",
               blocks, ms, ms * 1e3 * 256 / blocks, ms * 8192.0 / blocks, av);
    }
    return 0;
}
