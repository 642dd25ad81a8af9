// ubench_walk.hip -- the consumer step of k_autocorr_wt in isolation (not part of the
// product): K chains of (multiply, add into a running sum) per step, operands in registers
// or read from LDS as the kernel does.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define STEPS 65536

// MODE 0: operands in registers, source order mul,mul,mul,add,add,add
// MODE 1: same, products of step s+1 issued before the adds of step s
// MODE 2: MODE 0 with the a-stream from LDS (ds_read_b128 per two steps, waits counted)
template <int K, int MODE>
__global__ void walk(double *out, long long *cyc, double seed)
{
    __shared__ __attribute__((aligned(16))) double buf[64 * 70];
    for (int i = threadIdx.x; i < 64 * 70; i += blockDim.x) buf[i] = seed + i * 1e-3;
    __syncthreads();
    double S[K], cy[K];
    for (int j = 0; j < K; j++) { S[j] = 1.0; cy[j] = seed * (j + 2); }
    double xs[8];
    for (int u = 0; u < 8; u++) xs[u] = seed + u + threadIdx.x;
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    typedef const volatile dbl2 __attribute__((address_space(3))) lds_cvd2;
    lds_cvd2 *row = (lds_cvd2 *)(buf + (threadIdx.x & 63) * 70);
    long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 0 || MODE == 2) {
#pragma unroll 1
        for (int st = 0; st < STEPS; st += 8) {
            if (MODE == 2) {
#pragma unroll
                for (int u = 0; u < 8; u += 2) { const dbl2 v = row[(st & 63) / 2 + u / 2]; xs[u] = v.x; xs[u + 1] = v.y; }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const double a = xs[u];
                double pr[K];
                pr[0] = a * a;
#pragma unroll
                for (int j = 1; j < K; j++) pr[j] = a * cy[j];
#pragma unroll
                for (int j = 0; j < K; j++) S[j] = S[j] + pr[j];
#pragma unroll
                for (int j = K - 1; j >= 2; j--) cy[j] = cy[j - 1];
                if (K > 1) cy[1] = a;
            }
        }
    } else if (MODE == 5) {
        // two operand streams (a lane of an odd-lag class: d[p] and the other parity's d[p-1],
        // the higher odd lags from carried values), one stage ahead
        constexpr int PS_CH = 8, NS = 8;
        lds_cvd2 *rowb = (lds_cvd2 *)(buf + ((threadIdx.x + 17) & 63) * 70);
#pragma unroll 1
        for (int tile = 0; tile < STEPS / 64; tile++) {
            double A[2][PS_CH], B[2][PS_CH];
            auto fetch = [&](int set, int stage) {
#pragma unroll
                for (int u = 0; u < PS_CH; u += 2) {
                    const dbl2 v = row[(stage * PS_CH + u) / 2];
                    A[set][u] = v.x; A[set][u + 1] = v.y;
                    const dbl2 w = rowb[(stage * PS_CH + u) / 2];
                    B[set][u] = w.x; B[set][u + 1] = w.y;
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int st = 0; st < NS; st++) {
                if (st + 1 < NS) fetch((st + 1) & 1, st + 1);
                if (st + 1 < NS) __builtin_amdgcn_s_waitcnt((3 << 14) | (8 << 8) | (7 << 4) | 0xF);
                else __builtin_amdgcn_s_waitcnt((3 << 14) | (0 << 8) | (7 << 4) | 0xF);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < PS_CH; u++) {
                    const double a = A[st & 1][u], b = B[st & 1][u];
                    double pr[K];
                    pr[0] = a * b;
#pragma unroll
                    for (int j = 1; j < K; j++) pr[j] = a * cy[j];
#pragma unroll
                    for (int j = 0; j < K; j++) S[j] = S[j] + pr[j];
#pragma unroll
                    for (int j = K - 1; j >= 2; j--) cy[j] = cy[j - 1];
                    if (K > 1) cy[1] = b;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (MODE == 3 || MODE == 4) {
        constexpr int PS_CH = 8, NS = 8, DEPTH = (MODE == 3) ? 2 : 1, NSET = DEPTH + 1;
#pragma unroll 1
        for (int tile = 0; tile < STEPS / 64; tile++) {
            double A[NSET][PS_CH];
            auto fetch = [&](int set, int stage) {
#pragma unroll
                for (int u = 0; u < PS_CH; u += 2) {
                    const dbl2 v = row[(stage * PS_CH + u) / 2];
                    A[set][u] = v.x; A[set][u + 1] = v.y;
                }
            };
#pragma unroll
            for (int k = 0; k < DEPTH; k++) fetch(k, k);
#pragma unroll
            for (int st = 0; st < NS; st++) {
                if (st + DEPTH < NS) fetch((st + DEPTH) % NSET, st + DEPTH);
                const int newer = (st + DEPTH < NS ? 1 : 0) + ((DEPTH == 2 && st + 1 < NS) ? 1 : 0);
                if (newer == 2) __builtin_amdgcn_s_waitcnt((3 << 14) | (8 << 8) | (7 << 4) | 0xF);
                else if (newer == 1) __builtin_amdgcn_s_waitcnt((3 << 14) | (4 << 8) | (7 << 4) | 0xF);
                else __builtin_amdgcn_s_waitcnt((3 << 14) | (0 << 8) | (7 << 4) | 0xF);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < PS_CH; u++) {
                    const double a = A[st % NSET][u];
                    double pr[K];
                    pr[0] = a * a;
#pragma unroll
                    for (int j = 1; j < K; j++) pr[j] = a * cy[j];
#pragma unroll
                    for (int j = 0; j < K; j++) S[j] = S[j] + pr[j];
#pragma unroll
                    for (int j = K - 1; j >= 2; j--) cy[j] = cy[j - 1];
                    if (K > 1) cy[1] = a;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        double pr[K];
        pr[0] = xs[0] * xs[0];
#pragma unroll
        for (int j = 1; j < K; j++) pr[j] = xs[0] * cy[j];
#pragma unroll 1
        for (int st = 0; st < STEPS; st += 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                // rotate the carries for the NEXT step, form its products, then add this step's
#pragma unroll
                for (int j = K - 1; j >= 2; j--) cy[j] = cy[j - 1];
                if (K > 1) cy[1] = xs[u];
                const double an = xs[(u + 1) & 7];
                double pn[K];
                pn[0] = an * an;
#pragma unroll
                for (int j = 1; j < K; j++) pn[j] = an * cy[j];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < K; j++) S[j] = S[j] + pr[j];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < K; j++) pr[j] = pn[j];
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double s = 0;
    for (int j = 0; j < K; j++) s += S[j] + cy[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int MODE>
void run(const char *name, int blocks, int threads)
{
    double *out; long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) walk<K, MODE><<<blocks, threads>>>(out, cyc, 1.5);
    hipEventRecord(e0);
    walk<K, MODE><<<blocks, threads>>>(out, cyc, 1.5);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    printf("%-44s K=%d  cycles/step=%7.2f  wall ns/step=%7.2f  => %.2f GHz\n", name, K, avg / STEPS, ms * 1e6 / STEPS,
           (avg / STEPS) / (ms * 1e6 / STEPS));
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<3, 0>("regs, 1 wave/SIMD, chip", 256, 256);
    run<3, 1>("regs pipelined, 1 wave/SIMD, chip", 256, 256);
    run<3, 2>("LDS a-stream, 1 wave/SIMD, chip", 256, 256);
    run<3, 3>("LDS a-stream, 2 stages ahead, 1 w/SIMD", 256, 256);
    run<3, 4>("LDS a-stream, 1 stage ahead, 1 w/SIMD", 256, 256);
    run<3, 3>("LDS a-stream, 2 stages ahead, 1 wave on chip", 1, 64);
    run<2, 0>("regs, 1 wave/SIMD, chip", 256, 256);
    run<2, 1>("regs pipelined, 1 wave/SIMD, chip", 256, 256);
    run<3, 0>("regs, 2 waves/SIMD, chip", 256, 512);
    // fatter lag classes: all even lags of order 8 / 12 (one stream), all odd lags (two streams)
    run<5, 3>("LDS a-stream, K=5 (even lags of order 8)", 256, 256);
    run<7, 3>("LDS a-stream, K=7 (even lags of order 12)", 256, 256);
    run<4, 5>("LDS a+b streams, K=4 (odd lags of order 8)", 256, 256);
    run<6, 5>("LDS a+b streams, K=6 (odd lags of order 12)", 256, 256);
    run<2, 5>("LDS a+b streams, K=2 (today's groups)", 256, 256);
    run<5, 3>("LDS a-stream, K=5, 2 waves/SIMD", 256, 512);
    run<4, 5>("LDS a+b streams, K=4, 2 waves/SIMD", 256, 512);
    // does a second wave on the SIMD fill the issue time the first one loses to its LDS reads?
    // (waves are served oldest first: a per-wave cycle count says nothing here, wall time does)
    run<3, 3>("LDS a-stream, 2 ahead, 2 waves/SIMD, chip", 256, 512);
    run<3, 3>("LDS a-stream, 2 ahead, 2 waves/SIMD, 1 CU", 1, 512);
    run<3, 3>("LDS a-stream, 2 ahead, 1 wave/SIMD, 1 CU", 1, 256);
    run<3, 3>("LDS a-stream, 2 ahead, 3 waves/SIMD, chip", 256, 768);
    run<3, 0>("regs, 1 wave on the chip", 1, 64);
    run<3, 1>("regs pipelined, 1 wave on the chip", 1, 64);
    return 0;
}
