// ubench_walk2.hip -- a K1 consumer step WITHOUT producers (not part of the product): the lane
// reads the raw int16 sample pair (x[2k], x[2k+1]) of its subframe from LDS (one ds_read_b32),
// windows what it needs itself (sign-extend, int -> fp64, multiply by a wave-uniform weight) and
// runs K chains of (multiply, add).  BOTH = 1: an odd-lag lane (needs both samples), 0: an
// even-lag lane (its own parity only).  Question: is (read + windowing) cheaper than the 8.5
// cycles per double an fp64 operand costs through the LDS today?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_walk2.hip -o tools/bin/ubench_walk2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define STEPS 32768

template <int K, int BOTH>
__global__ void walk(double *out, long long *cyc, const double *__restrict__ wtab, int seed)
{
    __shared__ uint32_t buf[64 * 65 * 4];                 // [wave][lane-row stride 65][64 dwords]
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 65 * 4; i += blockDim.x) buf[i] = (uint32_t)(seed * 2654435761u + i * 40503u);
    __syncthreads();
    double S[K], cy[K];
    for (int j = 0; j < K; j++) { S[j] = 1.0; cy[j] = 0.5 * (j + 2); }
    typedef const volatile uint32_t __attribute__((address_space(3))) lds_cu;
    lds_cu *row = (lds_cu *)(buf + wv * 64 * 65 + (lane & 31) * 65);
    const int par = lane >> 5;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int st = 0; st < STEPS; st += 8) {
        uint32_t raw[8];
#pragma unroll
        for (int u = 0; u < 8; u++) raw[u] = row[(st + u) & 63];
        // two wave-uniform weights per step (positions 2k, 2k+1): scalar loads
        double w0[8], w1[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { w0[u] = wtab[2 * ((st + u) & 1023)]; w1[u] = wtab[2 * ((st + u) & 1023) + 1]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int32_t x0 = (int32_t)(int16_t)(raw[u] & 0xFFFF), x1 = (int32_t)raw[u] >> 16;
            const double d0 = (double)x0 * w0[u], d1 = (double)x1 * w1[u];
            double a, b;
            if (BOTH) { a = par ? d1 : d0; b = par ? d0 : d1; }
            else { a = (double)(par ? x1 : x0) * (par ? w1[u] : w0[u]); b = a; }
            double pr[K];
            pr[0] = a * (BOTH ? b : cy[0]);
#pragma unroll
            for (int j = 1; j < K; j++) pr[j] = a * cy[j];
#pragma unroll
            for (int j = 0; j < K; j++) S[j] = S[j] + pr[j];
#pragma unroll
            for (int j = K - 1; j >= 1; j--) cy[j] = cy[j - 1];
            cy[0] = BOTH ? b : a;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int j = 0; j < K; j++) s += S[j] + cy[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int BOTH>
void run(const char *name, int blocks, int threads, const double *wtab)
{
    double *out; long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) walk<K, BOTH><<<blocks, threads>>>(out, cyc, wtab, 3);
    hipEventRecord(e0);
    walk<K, BOTH><<<blocks, threads>>>(out, cyc, wtab, 3);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    printf("%-40s K=%d BOTH=%d cycles/step=%7.2f  wall ns/step=%7.2f => %.2f GHz\n", name, K, BOTH, avg / STEPS,
           ms * 1e6 / STEPS, (avg / STEPS) / (ms * 1e6 / STEPS));
    hipFree(out); hipFree(cyc);
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    double *wtab; hipMalloc(&wtab, sizeof(double) * 2048);
    std::vector<double> h(2048); for (int i = 0; i < 2048; i++) h[i] = 1.0 - 1e-4 * i;
    hipMemcpy(wtab, h.data(), sizeof(double) * 2048, hipMemcpyHostToDevice);
    run<3, 0>("even lane {0,2,4}, 1 wave on chip", 1, 64, wtab);
    run<2, 1>("odd lane {1,3}, 1 wave on chip", 1, 64, wtab);
    run<3, 0>("even lane {0,2,4}, 4 waves/CU, chip", 256, 256, wtab);
    run<2, 0>("even lane {6,8}, 4 waves/CU, chip", 256, 256, wtab);
    run<2, 1>("odd lane {1,3}, 4 waves/CU, chip", 256, 256, wtab);
    run<7, 1>("all 13 lags of order 12 (6 odd + 7 even ~ K=13)", 256, 256, wtab);
    return 0;
}
