// Diagnostic (not part of the product): what a kernel that only moves K0's bytes reaches on this box -- read 134 MB of
// int32 PCM, write 67 MB (the 16-bit rows), nothing else -- against k_prepare_stereo's 41 us; plus a plain read and a copy.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o tools/bin/hbm_probe && tools/bin/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
// one workgroup per frame of 4096 stereo sample-frames (32 KB in, 16 KB out), 256 threads, four quads per thread
template <int MODE>
__global__ __launch_bounds__(256) void k_move(const int4 *__restrict__ in, int2 *__restrict__ out, int *__restrict__ sink)
{
    const int f = blockIdx.x, t = threadIdx.x;
    const int4 *src = in + (size_t)f * 2048;
    int4 a[4], b[4];
#pragma unroll
    for (int m = 0; m < 4; m++) { a[m] = src[2 * (t + 256 * m)]; b[m] = src[2 * (t + 256 * m) + 1]; }
    if (MODE == 0) {                      // read only
        int acc = 0;
#pragma unroll
        for (int m = 0; m < 4; m++) acc += a[m].x + a[m].w + b[m].y + b[m].z;
        if (acc == 0x7fffffff) sink[0] = acc;
    } else {                              // read 2, write 1: two 16-bit rows
        int2 *dl = out + (size_t)f * 2048, *dr = dl + 1024;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            dl[t + 256 * m] = make_int2((a[m].x & 0xFFFF) | (a[m].z << 16), (b[m].x & 0xFFFF) | (b[m].z << 16));
            dr[t + 256 * m] = make_int2((a[m].y & 0xFFFF) | (a[m].w << 16), (b[m].y & 0xFFFF) | (b[m].w << 16));
        }
    }
}
int main()
{
    const int frames = 4096;
    int4 *in; int2 *out; int *sink;
    CK(hipMalloc(&in, (size_t)frames * 32768)); CK(hipMalloc(&out, (size_t)frames * 16384)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 1, (size_t)frames * 32768));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++) {
        for (int w = 0; w < 200; w++) { if (mode == 0) k_move<0><<<frames, 256>>>(in, out, sink); else k_move<1><<<frames, 256>>>(in, out, sink); }
        CK(hipEventRecord(e0));
        for (int w = 0; w < 400; w++) { if (mode == 0) k_move<0><<<frames, 256>>>(in, out, sink); else k_move<1><<<frames, 256>>>(in, out, sink); }
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms / 400 * 1e3, bytes = (double)frames * (mode ? 49152 : 32768);
        printf("%s: %.1f us per 4096 frames, %.2f TB/s\n", mode ? "read 134 MB + write 67 MB" : "read 134 MB", us, bytes / us / 1e6);
    }
    return 0;
}
