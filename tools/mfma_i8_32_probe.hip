// Diagnostic (not part of the product): operand / result lane maps of v_mfma_i32_32x32x32_i8 on gfx950, checked with
// random data against the CPU (the matrix order search, k3s_search.hip mm_search32, relies on them).
//   assumed: lane l holds A[m = l & 31][k = 16 (l >> 5) + j], B[k = 16 (l >> 5) + j][n = l & 31], j = 0..15 (4 dwords),
//            D[m = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][n = l & 31] in register r = 0..15; C accumulates.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_i8_32_probe.hip -o tools/bin/mfma_i8_32_probe && tools/bin/mfma_i8_32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void k_layout(const signed char *A, const signed char *B, const int *Cin, int *D)
{
    const int l = threadIdx.x;
    v4i a, b;
    v16i c;
    signed char ab[16], bb[16];
    for (int j = 0; j < 16; j++) { ab[j] = A[(l & 31) * 32 + 16 * (l >> 5) + j]; bb[j] = B[(16 * (l >> 5) + j) * 32 + (l & 31)]; }
    memcpy(&a, ab, 16); memcpy(&b, bb, 16);
    for (int r = 0; r < 16; r++) c[r] = Cin[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main()
{
    std::vector<signed char> A(32 * 32), B(32 * 32);
    std::vector<int> C(32 * 32), D(32 * 32), R(32 * 32);
    srand(7);
    for (auto &v : A) v = (signed char)(rand() % 256 - 128);
    for (auto &v : B) v = (signed char)(rand() % 256 - 128);
    for (auto &v : C) v = rand() % 100000 - 50000;
    for (int m = 0; m < 32; m++) for (int n = 0; n < 32; n++) { int s = C[m * 32 + n]; for (int k = 0; k < 32; k++) s += (int)A[m * 32 + k] * (int)B[k * 32 + n]; R[m * 32 + n] = s; }
    signed char *dA, *dB; int *dC, *dD;
    CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4096)); CK(hipMalloc(&dD, 4096));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice));
    k_layout<<<1, 64>>>(dA, dB, dC, dD);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 1024; i++) bad += D[i] != R[i];
    printf("v_mfma_i32_32x32x32_i8 lane maps: %s (%d of 1024 elements differ)\n", bad ? "WRONG" : "as assumed", bad);
    return bad != 0;
}
