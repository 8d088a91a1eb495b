// Probe: what does it cost 2048 waves to read the same 3 KiB (a query) at kernel start from (a) device memory,
// (b) coherent pinned host memory, (c) non-coherent pinned host memory -- and is (c) fresh at every launch when the host
// rewrites it between launches?  build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/hostq_probe tools/probe/hostq_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void reader(const float* __restrict__ q, float* __restrict__ out, int* __restrict__ stale, float expect) {
    const int lane = threadIdx.x & 63;
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const f32x4 v = *(const f32x4*)(q + 4 * (lane + 64 * u));
        s += v[0] + v[1] + v[2] + v[3];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        if (s != expect) atomicAdd(stale, 1);
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
    }
}
int main() {
    const int D = 768, G = 512, R = 300;
    float *dq, *hq_c, *hq_nc, *out;
    int* stale;
    CK(hipMalloc(&dq, D * 4));
    CK(hipHostMalloc(&hq_c, D * 4, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc(&hq_nc, D * 4, hipHostMallocMapped | hipHostMallocNonCoherent));
    CK(hipMalloc(&out, G * 4 * 4));
    CK(hipMalloc(&stale, 4));
    float *dc, *dnc;
    CK(hipHostGetDevicePointer((void**)&dc, hq_c, 0));
    CK(hipHostGetDevicePointer((void**)&dnc, hq_nc, 0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[3] = {"device memory (hipMemcpyAsync before each launch)", "pinned host, coherent", "pinned host, non-coherent"};
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemset(stale, 0, 4));
        std::vector<float> h(D);
        float tot = 0;
        for (int r = 0; r < R; ++r) {
            float expect = 0.f;
            for (int i = 0; i < D; ++i) { h[i] = (float)((r * 7 + i) % 13); expect += h[i]; }
            const float* src = dq;
            if (mode == 0) CK(hipMemcpyAsync(dq, h.data(), D * 4, hipMemcpyHostToDevice, st));
            if (mode == 1) { for (int i = 0; i < D; ++i) hq_c[i] = h[i]; src = dc; }
            if (mode == 2) { for (int i = 0; i < D; ++i) hq_nc[i] = h[i]; src = dnc; }
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(reader, dim3(G), dim3(256), 0, st, src, out, stale, expect);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 20) tot += ms;
        }
        int hs = 0;
        CK(hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost));
        printf("%-52s kernel %.2f us avg, stale waves %d\n", names[mode], tot / (R - 20) * 1e3, hs);
    }
    return 0;
}
