// Operand / result layout of v_mfma_i32_16x16x64_i8 on gfx950, checked against a host product: lane l supplies A[l % 16][16 (l / 16) .. +16)
// and B[16 (l / 16) .. +16)[l % 16] as 16 bytes each; result register i of lane l is D[4 (l / 16) + i][l % 16].  Prints "layout ok".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const signed char* A, const signed char* B, int* D) {
    const int l = threadIdx.x;
    i32x4 a = *(const i32x4*)(A + (l % 16) * 64 + 16 * (l / 16));     // A row-major [16][64]
    i32x4 b = *(const i32x4*)(B + (l % 16) * 64 + 16 * (l / 16));     // B stored column-major: [16 columns][64 k]
    i32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * (l / 16) + i) * 16 + (l % 16)] = acc[i];
}
int main() {
    signed char hA[16 * 64], hB[16 * 64];
    srand(3);
    for (int i = 0; i < 16 * 64; ++i) { hA[i] = (signed char)(rand() % 255 - 127); hB[i] = (signed char)(rand() % 255 - 127); }
    signed char *dA, *dB; int* dD; int hD[256];
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
        int s = 0;
        for (int kk = 0; kk < 64; ++kk) s += (int)hA[r * 64 + kk] * (int)hB[c * 64 + kk];
        if (s != hD[r * 16 + c]) ++bad;
    }
    printf(bad ? "layout WRONG (%d of 256 entries)\n" : "layout ok\n", bad);
    return bad ? 1 : 0;
}
