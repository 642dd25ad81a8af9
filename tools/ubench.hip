// ubench.hip -- instruction latency / issue-rate probes for gfx950 (not part
// of the product; numbers feed DESIGN.md).  hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define N_IT 2048

template <int OP, int CHAINS>
__global__ void k(double *out, long long *cyc, double seed, int iseed)
{
    double a[CHAINS];
    long long ia[CHAINS];
    for (int i = 0; i < CHAINS; i++) { a[i] = seed + i + threadIdx.x; ia[i] = iseed + i; }
    double m = seed * 1.0000001;
    int im = iseed | 3;
    long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int it = 0; it < N_IT; it += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (OP == 0) a[c] = a[c] + m;
                else if (OP == 1) a[c] = a[c] * m;
                else if (OP == 2) a[c] = __builtin_fma(a[c], m, m);
                else if (OP == 3) ia[c] = ia[c] + (long long)im * (long long)(int)ia[c];     // v_mad_i64_i32
                else if (OP == 4) ia[c] = (int)ia[c] * im + 1;                                // 32-bit mul lo
                else if (OP == 5) ia[c] = __mul24((int)ia[c], im) + 7;       // mul_i24
                else if (OP == 6) a[c] = (double)(int)ia[c] + a[c], ia[c] += 1;               // cvt_f64_i32 + add
                else if (OP == 8) ia[c] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short __attribute__((ext_vector_type(2))), (int)ia[c]), __builtin_bit_cast(short __attribute__((ext_vector_type(2))), im), (int)ia[c], false);   // v_dot2_i32_i16
                else if (OP == 9) ia[c] = __builtin_amdgcn_sdot4((int)ia[c], im, (int)ia[c], false);            // v_dot4_i32_i8
                else if (OP == 7) { const double pr = m * (m + (double)(u + c)); a[c] = a[c] + pr; }   // mul + dependent add (autocorr step)
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double s = 0; long long is = 0;
    for (int i = 0; i < CHAINS; i++) { s += a[i]; is += ia[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (double)is;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP, int CHAINS>
void run(const char *name, int blocks, int threads)
{
    double *out; long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP, CHAINS><<<blocks, threads>>>(out, cyc, 1.5, 12345);
    hipEventRecord(e0);
    k<OP, CHAINS><<<blocks, threads>>>(out, cyc, 1.5, 12345);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    // s_memtime ticks at 100 MHz? report both ticks and wall-derived ns per op
    double ops = (double)N_IT * CHAINS;
    printf("%-28s chains=%d blocks=%4d thr=%4d  memtime/op=%7.3f  wall ns/op(per wave)=%7.3f\n",
           name, CHAINS, blocks, threads, avg / ops, ms * 1e6 / ops);
    hipFree(out); hipFree(cyc);
}

int main()
{
    // one wave per block, few blocks: pure latency / single-wave issue
    run<0, 1>("add_f64 dep", 64, 64);
    run<0, 4>("add_f64 4ch", 64, 64);
    run<0, 8>("add_f64 8ch", 64, 64);
    run<1, 1>("mul_f64 dep", 64, 64);
    run<1, 8>("mul_f64 8ch", 64, 64);
    run<2, 1>("fma_f64 dep", 64, 64);
    run<2, 8>("fma_f64 8ch", 64, 64);
    run<3, 1>("mad_i64_i32 dep", 64, 64);
    run<3, 8>("mad_i64_i32 8ch", 64, 64);
    run<4, 8>("mul_lo_i32 8ch", 64, 64);
    run<5, 8>("mul_i24 8ch", 64, 64);
    run<6, 8>("cvt_f64_i32+add 8ch", 64, 64);
    // 4 waves per SIMD on every CU: throughput under full occupancy
    run<0, 8>("add_f64 8ch full", 1024, 256);
    run<2, 8>("fma_f64 8ch full", 1024, 256);
    run<3, 8>("mad_i64_i32 8ch full", 1024, 256);
    run<4, 8>("mul_lo_i32 8ch full", 1024, 256);
    run<8, 8>("dot2_i32_i16 8ch", 64, 64);
    run<9, 8>("dot4_i32_i8 8ch", 64, 64);
    run<8, 8>("dot2_i32_i16 8ch full", 1024, 256);
    // the autocorrelation step: CHAINS x (one multiply + one add into a running sum)
    run<7, 3>("mul+add 3 chains, 1 wave", 1, 64);
    run<7, 3>("mul+add 3ch 1 wave/SIMD, 1 CU", 1, 256);
    run<7, 3>("mul+add 3ch 1 wave/SIMD, chip", 256, 256);
    run<7, 3>("mul+add 3ch 2 waves/SIMD, chip", 256, 512);
    run<7, 4>("mul+add 4ch 1 wave/SIMD, chip", 256, 256);
    run<7, 6>("mul+add 6ch 1 wave/SIMD, chip", 256, 256);
    run<0, 3>("add only 3ch 1 wave/SIMD, chip", 256, 256);
    run<1, 3>("mul only 3ch 1 wave/SIMD, chip", 256, 256);
    return 0;
}
