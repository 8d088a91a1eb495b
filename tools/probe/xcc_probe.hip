// prints the XCC id each workgroup of a 256-block launch reads from HW_REG_XCC_ID (one block per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* o) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) o[blockIdx.x] = x;
}
int main() {
    unsigned* d; hipMalloc(&d, 4 * 512);
    hipLaunchKernelGGL(k, dim3(512), dim3(512), 150 * 1024, 0, d);
    unsigned h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int cnt[16] = {0};
    for (int i = 0; i < 512; ++i) { if (i < 32) printf("%x ", h[i]); cnt[h[i] & 15]++; }
    printf("\n"); for (int i = 0; i < 16; ++i) printf("xcc%d:%d ", i, cnt[i]); printf("\n");
    return 0;
}
