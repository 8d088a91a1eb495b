// Sustained matrix-pipe rate on RANDOM operands (the chip lowers its clock under load: MI355X_MICROARCH.md, DVFS give-back):
// v_mfma_f32_32x32x16_bf16 against v_mfma_i32_32x32x32_i8, operands in registers, 2 waves per SIMD, 4 accumulators per wave.
// KIND 2: v_mfma_i32_16x16x64_i8 on the same 32 x 128 output tile per wave (2 x 8 accumulators of 16 x 16), the same products per
// iteration -- the guide's DVFS item 7 measured the 16x16 bf16 shape at 1.12-1.15 x the 32x32 one on random data.
// usage: mfma_rate_probe            prints TMAC/s of each and the ratios
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512) void k(const f32x4* __restrict__ src, float* out, int iters) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    f32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = src[(t * 8 + i) & 0xfffff];
        b[i] = src[(t * 8 + 4 + i) & 0xfffff];
    }
    f32x16 accf[4] = {};
    i32x16 acci[4] = {};
    if (KIND == 2) {
        i32x4 b8[8], acc16[2][8] = {};
        for (int i = 0; i < 8; ++i) b8[i] = __builtin_bit_cast(i32x4, src[(t * 16 + 8 + i) & 0xfffff]);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int v = 0; v < 8; ++v)
                        acc16[rg][v] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a[rg * 2 + u]), b8[v], acc16[rg][v], 0, 0, 0);
        }
        float s2 = 0;
        for (int rg = 0; rg < 2; ++rg)
            for (int v = 0; v < 8; ++v)
                for (int r = 0; r < 4; ++r) s2 += (float)acc16[rg][v][r];
        out[t] = s2;
        return;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (KIND == 0) accf[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[u]), __builtin_bit_cast(bf16x8, b[v]), accf[v], 0, 0, 0);
                else acci[v] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a[u]), __builtin_bit_cast(i32x4, b[v]), acci[v], 0, 0, 0);
            }
    }
    float s = 0;
    for (int v = 0; v < 4; ++v)
        for (int r = 0; r < 16; ++r) s += KIND == 0 ? accf[v][r] : (float)acci[v][r];
    out[t] = s;
}

int main() {
    const int n = 1 << 20;
    std::vector<unsigned> h(n * 4);
    srand(1);
    for (auto& x : h) {
        // random bf16 pairs with moderate exponents (as split embeddings have) / random int8 bytes
        unsigned lo = (rand() & 0x807f) | ((120 + (rand() % 8)) << 7), hi = (rand() & 0x807f) | ((120 + (rand() % 8)) << 7);
        x = lo | (hi << 16);
    }
    f32x4* d; float* o;
    hipMalloc(&d, n * 16); hipMalloc(&o, 256 * 512 * 4);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    double rate[3];
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, d, o, iters);
            else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, d, o, iters);
            else hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double macs = 256.0 * 8 * iters * 16.0 * 32 * 32 * (kind == 0 ? 16 : 32);   // (kind 2: 32 instructions of 16 x 16 x 64 = the same products)
            rate[kind] = macs / (ms * 1e-3) / 1e12;
            printf("%s: %.1f ms, %.1f TMAC/s (%.1f TOP/s)\n", kind == 0 ? "bf16 32x32x16" : (kind == 1 ? "i8   32x32x32" : "i8   16x16x64"), ms, rate[kind], 2 * rate[kind]);
        }
    }
    printf("ratio i8 / bf16 = %.2f, i8 16x16x64 / 32x32x32 = %.3f\n", rate[1] / rate[0], rate[2] / rate[1]);
    return 0;
}
