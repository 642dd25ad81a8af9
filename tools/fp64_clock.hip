// Diagnostic (not part of the product): what the chip clocks at under sustained un-fused fp64
// work -- the load of K1's consumers.  Each lane runs CH chains of (multiply, dependent add);
// the launch lasts ~1 ms and is repeated, shader-clock ticks (s_memtime) and wall time are
// both taken, ticks / wall = the clock the SIMDs really ran at.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/fp64_clock.hip -o tools/bin/fp64_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CH>
__global__ void k(double *out, long long *cyc, long long *rt, double seed, int iters)
{
    double a[CH], x[CH];
    for (int i = 0; i < CH; i++) { a[i] = seed + i + threadIdx.x; x[i] = seed * (1.0 + 1e-9 * i); }
    const double m = seed * 1.0000001;
    const long long t0 = __builtin_amdgcn_s_memtime();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < CH; c++) { const double pr = x[c] * m; a[c] = a[c] + pr; x[c] = pr; }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < CH; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int CH>
void run(const char *name, int blocks, int threads, int iters, int launches)
{
    double *out; long long *cyc, *rt;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipMalloc(&rt, sizeof(long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < launches; i++) k<CH><<<blocks, threads>>>(out, cyc, rt, 1.0000001, iters);   // settle
    hipEventRecord(e0);
    for (int i = 0; i < launches; i++) k<CH><<<blocks, threads>>>(out, cyc, rt, 1.0000001, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks), r(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), rt, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double ticks = 0, real = 0;
    for (int i = 0; i < blocks; i++) { ticks += h[i]; real += r[i]; }
    ticks /= blocks; real /= blocks;
    const double ops = 2.0 * 8 * CH * iters;                 // fp64 instructions per wave per launch
    const double wall_us = ms * 1e3 / launches;
    const double waves = (double)blocks * threads / 64;
    printf("%-34s launch %8.1f us  s_memtime ticks/instr %6.3f  s_memrealtime ticks %9.0f (%6.1f us at 100 MHz)  "
           "memtime/realtime %6.2f  ->  %6.2f T fp64 instr-lanes/s\n",
           name, wall_us, ticks / ops, real, real / 100.0, ticks / real, ops * 64 * waves / (wall_us * 1e-6) / 1e12);
    hipFree(out); hipFree(cyc); hipFree(rt);
}

int main()
{
    run<7>("7 chains, 1 wave, 1 CU", 1, 64, 4000, 20);
    run<7>("7 chains, 1 wave/SIMD, 1 CU", 1, 256, 4000, 20);
    run<7>("7 chains, 1 wave/SIMD, chip", 256, 256, 4000, 200);
    run<7>("7 chains, 2 waves/SIMD, chip", 256, 512, 4000, 200);
    run<3>("3 chains, 2 waves/SIMD, chip", 256, 512, 8000, 200);
    run<3>("3 chains, 1 wave/SIMD, chip", 256, 256, 8000, 200);
    run<7>("7 chains, 1 wave/SIMD, half chip", 128, 256, 4000, 200);
    return 0;
}
