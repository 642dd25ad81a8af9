// Diagnostic (not part of the product): layout and issue time of v_mfma_f64_16x16x4_f64 on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o tools/bin/mfma_probe && tools/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k_layout(double *out)   // out[lane*4 + r]
{
    const int l = threadIdx.x;
    // try: A lane l holds A[i = l%16][k = l/16]; B lane l holds B[k = l/16][j = l%16]
    const double a = (double)((l % 16) * 100 + (l / 16));        // A[i][k] = 100 i + k
    const double b = (l / 16 == 0) ? (double)(1 << (l % 16)) : 0.0;   // B[0][j] = 2^j, other k rows 0
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    // D[i][j] = A[i][0] * 2^j = 100 i * 2^j  -> decode (i, j) from value per (lane, r)
    for (int r = 0; r < 4; r++) out[l * 4 + r] = c[r];
}

__global__ void k_time(long long *cyc, double *sink, int reps)
{
    const int l = threadIdx.x;
    double a = 1.0 + l, b = 2.0;
    double4_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * 64 + l] = c0[0] + c1[1] + c2[2] + c3[3];
    // dependent chain
    double4_t d = {0, 0, 0, 0};
    long long t2 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; i++) {
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
    }
    long long t3 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * 64 + l] += d[0];
    if (l == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; }
}

int main()
{
    double *d_out; hipMalloc(&d_out, 256 * sizeof(double));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d_out);
    double h[256]; hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    setvbuf(stdout, nullptr, _IONBF, 0);
    for (int l = 0; l < 64; l += 5) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; r++) {
            double v = h[l * 4 + r];
            // v = 100 i 2^j; find j = trailing power, i
            int j = 0; long long q = (long long)v; 
            if (q == 0) { printf("  r%d: (i=0, j=?)", r); continue; }
            // 100 = 4*25: remove factor 25*i first by finding j such that q / 2^j is 100 i with i<16 odd part
            int found = 0;
            for (int jj = 0; jj < 16 && !found; jj++) for (int ii = 1; ii < 16; ii++) if (q == (100LL * ii) << jj) { printf("  r%d: (i=%d, j=%d)", r, ii, jj); found = 1; break; }
            if (!found) printf("  r%d: %g", r, v);
        }
        printf("\n");
    }
    long long *d_c; hipMalloc(&d_c, 16); double *d_s; hipMalloc(&d_s, (size_t)4096 * 64 * sizeof(double));     // the largest launch below: 4096 blocks of 64 lanes
    for (int blocks : {1, 4096}) {
        hipLaunchKernelGGL(k_time, dim3(blocks), dim3(64), 0, 0, d_c, d_s, 1000);
        long long c[2]; hipMemcpy(c, d_c, 16, hipMemcpyDeviceToHost);
        printf("blocks %d: independent x4: %.1f memtime-ticks per MFMA; dependent: %.1f\n", blocks, c[0] / 4000.0, c[1] / 4000.0);
    }
    // throughput: all SIMDs busy, wall time
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 4096, reps = 20000;
    hipLaunchKernelGGL(k_time, dim3(blocks), dim3(64), 0, 0, d_c, d_s, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_time, dim3(blocks), dim3(64), 0, 0, d_c, d_s, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)blocks * reps * 8.0;      // MFMAs
    printf("throughput: %.3f ms for %.3g MFMA 16x16x4 f64 = %.1f TFLOP/s (2*1024 flop each)\n", ms, mf, mf * 2048 / ms / 1e9);
    return 0;
}
