// K2 on the bf16 matrix pipe: the fused X.X^T + streaming k-smallest kernel of as_build.hip with every operand as
// bf16 head + tail (x = xh + xl + r, |r| <= 2^-16 |x|) and x_i.x_j ~ xh_i.xh_j + xh_i.xl_j + xl_i.xh_j -- three
// v_mfma_f32_32x32x16_bf16 per 16 columns instead of eight v_mfma_f32_32x32x2_f32 (the bf16 pipe runs at 16 x the
// fp32 rate on gfx950).  The items are split ONCE (split_rows_bf16: per row and 32-column slab 32 heads then 32
// tails, the bytes and the stride of the fp32 slab row), so the LDS-DMA image, its XOR swizzle and the fragment
// addresses are those of the fp32 kernel: 16-byte chunk c of a slab row holds heads (c < 4) or tails (c >= 4) of
// columns 8 (c & 3) .. 8 (c & 3) + 7 -- exactly one lane's operand of a 16-column k-step.
//
// With the matrix time cut to a fifth, the kernel lives on its staging pipeline: a ring of RING = 3 slab buffers,
// slabs issued two ahead of the MFMAs, one raw s_barrier per slab, counted vmcnt (never 0 in the loop) -- and every
// LDS access of the steady state through inline asm, because the compiler orders any LDS access it can see behind ALL
// outstanding LDS-DMA (vmcnt(0)), which would drain the ring.  Candidate bookkeeping, symmetric mode, collect mode and
// every output are those of knn_mfma_dma8_kernel; what the dropped products cost is in
// err_coef (as_build.hip).  Replaces the distance block of the crate call at /root/reference/src/lib.rs:278-289.
#include <cstdlib>

#include "as_knn.hpp"

namespace as {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int RING = 3;             // slab buffers
constexpr int NPIECE = 6;           // DMA pieces per wave and slab: 4 x 8 A rows of its 32, 2 x 8 B rows of its 16
constexpr int SN = 4;               // column-side norm / threshold lines kept (tiles in flight: see the WAR note below)

// Timing skeletons (make ABLATION=1; ARROWSPACE_K2_DIAG, wrong results): bit 0 no MFMA, 1 no DMA / ring waits, 2 no
// epilogue, 3 no fragment reads, 4 no slab barrier.  The product library compiles none of it.
#define K2_DIAG(bit) ((DIAG & (bit)) != 0)

bool k2_bf16_enabled() {
    const char* e = getenv("ARROWSPACE_K2_FP32");
    return !(e && atoi(e) != 0);
}

// ------------------------------------------------------------------ split image
// 8 consecutive floats -> 8 heads (16 B) at the same place of the slab's first half, 8 tails in its second half.
// A non-finite value keeps its head and gets a zero tail (inf - inf would be NaN).
__global__ void split_bf16_kernel(const float* __restrict__ x32, float* __restrict__ xs, int64_t groups) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        const f32x4 a = *(const f32x4*)(x32 + 8 * g), b = *(const f32x4*)(x32 + 8 * g + 4);
        bf16x8 hi, lo;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            hi[t] = (__bf16)a[t];
            hi[4 + t] = (__bf16)b[t];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float ra = a[t] - (float)hi[t], rb = b[t] - (float)hi[4 + t];
            lo[t] = (__bf16)(ra == ra ? ra : 0.0f);
            lo[4 + t] = (__bf16)(rb == rb ? rb : 0.0f);
        }
        char* slab = (char*)xs + (g >> 2) * 128 + (g & 3) * 16;
        *(bf16x8*)slab = hi;
        *(bf16x8*)(slab + 64) = lo;
    }
}

as_status split_rows_bf16(const float* x32, float* xs, int64_t rows, int64_t dp, hipStream_t st) {
    const int64_t groups = rows * dp / 8;
    if (groups <= 0) return AS_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((groups + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(split_bf16_kernel, dim3(grid), dim3(256), 0, st, x32, xs, groups);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

// ------------------------------------------------------------------ int8 two-digit image
// x ~ s (128 a1 + a2) / 16256 with s = max |x_c| of the row, q = round(x / s * 16256) in [-16256, 16256], a2 = ((q + 64) mod
// 128) - 64 in [-64, 63], a1 = (q - a2) / 128 in [-127, 127].  x_i . x_j ~ s_i s_j (16384 a1.b1 + 128 (a1.b2 + a2.b1)) / 16256^2:
// three int8 products per column on v_mfma_i32_32x32x32_i8 (twice the bf16 rate, half its operand bytes), accumulated
// EXACTLY in int32 (|sum| <= 127^2 dp < 2^31 up to dp = 133 000).  What is lost: the quantisation residues theta (x = s (q +
// theta) / 16256, |theta| <= 1/2) and a2.b2 -- bounded by Cauchy-Schwarz through the rows' ACTUAL |theta|_2 and |a2|_2:
// maxima[0] = max_i s_i |theta_i|_2 / (16256 |x_i|), maxima[1] = max_i s_i |a2_i|_2 / (16256 |x_i|) (err_coef_i8).
// Per row and 64-column slab: 64 bytes of a1, then 64 of a2 -- the 128-byte slab row of the bf16 image, covering 64 columns.
__global__ __launch_bounds__(256) void quant_i8_kernel(const float* __restrict__ x32, const float* __restrict__ n32, signed char* __restrict__ x8,
                                                       float* __restrict__ fa8, int64_t rows, int64_t dp, int64_t dp8, unsigned int* maxima) {
    const int lane = lane_id();
    const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    float wr = 0.0f, wl = 0.0f;
    bool bad = false;
    for (int64_t row = gw; row < rows; row += nw) {
        const float* x = x32 + row * dp;
        float m = 0.0f, l1 = 0.0f;
        for (int64_t c = lane; c < dp; c += 64) {
            const float v = fabsf(x[c]);
            m = fmaxf(m, v);
            l1 += v;
            bad = bad || !(v <= 3.0e38f);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m = fmaxf(m, __shfl_xor(m, o, 64));
            l1 += __shfl_xor(l1, o, 64);
        }
        const float inv = m > 0.0f ? 16256.0f / m : 0.0f;
        signed char* dst = x8 + row * dp8 * 2;
        float st2 = 0.0f, sa2 = 0.0f;
        for (int64_t c = lane; c < dp8; c += 64) {
            const float v = c < dp ? x[c] : 0.0f;
            const float sc = v * inv;
            int q = (int)rintf(sc);
            q = q > 16256 ? 16256 : (q < -16256 ? -16256 : q);
            const int a2 = ((q + 64 + (1 << 20)) & 127) - 64;
            const int a1 = (q - a2) >> 7;
            const int64_t slab = c >> 6, k = c & 63;
            dst[slab * 128 + k] = (signed char)a1;
            dst[slab * 128 + 64 + k] = (signed char)a2;
            // the residue, with the rounding of v * inv (up to 16256 * 2^-23 either way) on top
            const float th = fabsf(sc - (float)q) + 0.004f;
            st2 += th * th;
            sa2 += (float)(a2 * a2);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            st2 += __shfl_xor(st2, o, 64);
            sa2 += __shfl_xor(sa2, o, 64);
        }
        (void)l1;
        if (lane == 0) {
            fa8[row] = m * (11.313708498984761f / 16256.0f);   // s sqrt(128) / 16256
            const float nx = sqrtf(n32[row]);
            if (nx > 0.0f) {   // (fp32 sums of dp squares: rounded up generously)
                wr = fmaxf(wr, m * sqrtf(st2) / (16256.0f * nx) * 1.001f);
                wl = fmaxf(wl, m * sqrtf(sa2) / (16256.0f * nx) * 1.001f);
            }
        }
    }
    if (lane == 0) {   // non-negative floats order like their bit patterns
        if (wr > 0.0f) atomicMax(maxima, __float_as_uint(wr));
        if (wl > 0.0f) atomicMax(maxima + 1, __float_as_uint(wl));
    }
    if (__any(bad) && lane == 0) atomicOr(maxima + 2, 1u);
}

as_status quant_rows_i8(const float* x32, const float* n32, void* x8, float* fa8, int64_t rows, int64_t dp, int64_t dp8, unsigned int* maxima,
                        hipStream_t st) {
    if (rows <= 0) return AS_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((rows + 3) / 4, 256 * 8);
    hipLaunchKernelGGL(quant_i8_kernel, dim3(grid), dim3(256), 0, st, x32, n32, (signed char*)x8, fa8, rows, dp, dp8, maxima);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

typedef int i32x4k __attribute__((ext_vector_type(4)));
// ------------------------------------------------------------------ LDS through inline asm
// Each block waits for its own reads before it ends: no register the compiler may copy or reuse holds data in flight.
__device__ __forceinline__ void lds_frag5(unsigned aa, unsigned ab, f32x4& a, f32x4& b0, f32x4& b1, f32x4& b2, f32x4& b3) {
    asm volatile(
        "ds_read_b128 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %6 offset:4096\n\tds_read_b128 %3, %6 offset:8192\n\t"
        "ds_read_b128 %4, %6 offset:12288\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(a), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
        : "v"(aa), "v"(ab)
        : "memory");
}
__device__ __forceinline__ void lds_read_b128x8(unsigned a0, f32x4 (&v)[8]) {
    // rows 4h + {0..3} + 8g, g = 0..3, of the wave's (bound, norm) pairs: 32 B per group, 64 B between groups
    asm volatile(
        "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:64\n\tds_read_b128 %3, %8 offset:80\n\t"
        "ds_read_b128 %4, %8 offset:128\n\tds_read_b128 %5, %8 offset:144\n\tds_read_b128 %6, %8 offset:192\n\tds_read_b128 %7, %8 offset:208\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(a0)
        : "memory");
}
__device__ __forceinline__ void lds_read_b128x2(unsigned a0, f32x4& v0, f32x4& v1) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a0) : "memory");
}
__device__ __forceinline__ f32x4 lds_read_b128(unsigned a0) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a0) : "memory");
    return v;
}
__device__ __forceinline__ void lds_read_b128x4(unsigned a0, f32x4 (&v)[4]) {
    // one float per row: rows 4h + {0..3} + 8g are 16 contiguous bytes, 32 bytes between groups
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\tds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"(a0)
                 : "memory");
}
__device__ __forceinline__ void lds_read_b32x4(unsigned a0, float& v0, float& v1, float& v2, float& v3) {
    asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:128\n\tds_read_b32 %2, %4 offset:256\n\tds_read_b32 %3, %4 offset:384\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                 : "v"(a0)
                 : "memory");
}
// S16 form: eight values of a column line, one per 16-column group (64 bytes apart)
__device__ __forceinline__ void lds_read_b32x8(unsigned a0, float (&v)[8]) {
    asm volatile(
        "ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:64\n\tds_read_b32 %2, %8 offset:128\n\tds_read_b32 %3, %8 offset:192\n\t"
        "ds_read_b32 %4, %8 offset:256\n\tds_read_b32 %5, %8 offset:320\n\tds_read_b32 %6, %8 offset:384\n\tds_read_b32 %7, %8 offset:448\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(a0)
        : "memory");
}
// S16 form: the A fragments of a slab -- both 16-row groups, high and low digits
__device__ __forceinline__ void lds_frag_a16(unsigned a00, unsigned a10, unsigned a01, unsigned a11, i32x4k& x00, i32x4k& x10, i32x4k& x01, i32x4k& x11) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x00), "=&v"(x10), "=&v"(x01), "=&v"(x11)
                 : "v"(a00), "v"(a10), "v"(a01), "v"(a11)
                 : "memory");
}
// ... and the B fragments of four consecutive 16-column groups (2 KiB apart), high digits from b1, low digits from b2
__device__ __forceinline__ void lds_frag_b16(unsigned b1, unsigned b2, i32x4k (&h)[4], i32x4k (&l)[4]) {
    asm volatile(
        "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:2048\n\tds_read_b128 %2, %8 offset:4096\n\tds_read_b128 %3, %8 offset:6144\n\t"
        "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:2048\n\tds_read_b128 %6, %9 offset:4096\n\tds_read_b128 %7, %9 offset:6144\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3]), "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]), "=&v"(l[3])
        : "v"(b1), "v"(b2)
        : "memory");
}
__device__ __forceinline__ int lds_read_i32(unsigned a) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write_i32(unsigned a, int v) {
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(a), "v"(v) : "memory");
}
// s_waitcnt vmcnt(n) for the wave-uniform counts the ring produces; anything else waits for everything (always correct)
__device__ __forceinline__ void ring_wait(int n) {
    if (n == NPIECE) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n == NPIECE + 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (n == NPIECE + 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == NPIECE + 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// keep the M smallest (key, idx) of the row's cnt <= CAP buffered candidates -- the ranks of compact_row (as_build.hip),
// formed in registers: a lane holds entries lane + 64 u and meets every entry through v_readlane (no LDS scratch:
// the ring takes the room).  Only called with the DMA ring's state irrelevant (it drains it: compiler-visible loads).
__device__ __forceinline__ void compact_row_reg(int M, float* bk, int* bi, int cnt, float* s_thr, int* s_cur_row, int* s_drop_row, float* pub) {
    const int lane = lane_id();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float k[4];
    int id[4], rank[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = lane + 64 * u;
        k[u] = t < cnt ? ld_l2(bk + t) : __int_as_float(0x7f800000);
        id[u] = t < cnt ? ld_l2(bi + t) : 0x7fffffff;
        rank[u] = 0;
    }
#pragma unroll
    for (int us = 0; us < 4; ++us) {
        const int lim = min(64, cnt - 64 * us);
        for (int l = 0; l < lim; ++l) {
            const float ks = bcast_lane(k[us], l);
            const int is = bcast_lane(id[us], l);
#pragma unroll
            for (int u = 0; u < 4; ++u) rank[u] += lex_less<float>(ks, is, k[u], id[u]) ? 1 : 0;
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (lane + 64 * u < cnt && rank[u] < M) {
            bk[rank[u]] = k[u];
            bi[rank[u]] = id[u];
            if (rank[u] == M - 1) {
                *s_thr = k[u];
                // symmetric mode: the row's new bound is published for the blocks that hold this item as a column
                if (pub) atomicMin((int*)pub, __float_as_int(fmaxf(k[u], 0.0f)));
            }
        }
    }
    if (lane == 0) {
        *s_cur_row = M;
        *s_drop_row = 1;
    }
    AS_CBAR();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------ the kernel
// Block = 8 waves (two per SIMD), tile 256 rows x 128 columns; wave w owns rows [32w, 32w+32) as 1x4 accumulators.
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// I8: the same kernel on the int8 two-digit image (quant_i8_kernel): a slab row is 64 columns (64 bytes of a1, 64 of a2), a
// k-step 32 columns of v_mfma_i32_32x32x32_i8, the chunk addressing that of heads and tails; two int32 accumulator sets per
// tile (a1.b1 and the cross terms), turned into the dot by the rows' and columns' scales in the epilogue.
// S16 (int8 image only): the products on v_mfma_i32_16x16x64_i8 -- a wave's 32 x 128 tile as 2 x 8 accumulators of 16 x 16, one
// k-step of 64 columns per slab; the same LDS image, the same bytes read per product, the same cycles per product -- but the
// chip holds a higher clock under this shape on random data (MI355X_MICROARCH.md, DVFS give-back item 7; a bare loop of it
// delivered 1.21 x the 32x32x32 loop's rate: tools/probe/mfma_rate_probe.hip).  Lane l of the wave supplies row (l % 16) and
// columns 16 (l / 16) .. + 15 of a fragment; result register i of accumulator (rg, cg) is row 16 rg + 4 (l / 16) + i, column
// 16 cg + l % 16 (tools/probe/mfma16_layout.hip).  The epilogue appends in ascending column order as the 32x32 form does: the
// candidate buffers, and everything behind them, are the same bit for bit.
template <int METRIC, bool COLLECT, bool SYM, int DIAG = 0, bool I8 = false, bool S16 = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void knn_bf16_kernel(KnnArgs a) {
    static_assert(!S16 || I8, "the 16x16x64 shape is the int8 image's");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Sl = (float*)smem;                     // RING slab buffers: A rows then B rows, 128 B per row
    float2* s_ta = (float2*)(Sl + RING * DSLAB);  // per row: (running bound, n_i or 1/|x_i|)
    int* s_cur = (int*)(s_ta + BM);
    int* s_drop = s_cur + BM;
    int* s_id = s_drop + BM;                      // collect mode: global item id of every A row
    float* s_n = (float*)(s_id + BM);             // [SN][3][BN]: the column items' norms, thresholds and (int8 image) scales of the tiles in flight
    float* s_fa = s_n + SN * 3 * BN;              // int8 image, L2: the rows' scales
    int* s_unit = (int*)(s_fa + BM);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* __restrict__ bkey = a.buf_key + (size_t)blockIdx.x * BM * CAP;
    int* __restrict__ bidx = a.buf_idx + (size_t)blockIdx.x * BM * CAP;
    const int units = a.nrb * a.S;
    const int nslab = a.nslab;
    const float finf = __int_as_float(0x7f800000);
    const int drow = lane >> 3;
    const int csw0 = (lane & 7) ^ ((lane >> 4) & 7);
    const int csw1 = (lane & 7) ^ ((4 + (lane >> 4)) & 7);
    const unsigned lo0 = (unsigned)((drow * a.ld + csw0 * 4) * 4), lo1 = (unsigned)((drow * a.ld + csw1 * 4) * 4);
    // fragment addresses inside a slab buffer: chunk 2 q + h of the lane's row (q = 0, 1: heads of k-step q; q = 2, 3: tails)
    unsigned aoff[4], boff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned sw = (unsigned)(((2 * q + h) ^ ((l31 >> 1) & 7)) << 4);
        aoff[q] = lds0 + (unsigned)((w * 32 + l31) * DROW * 4) + sw;
        boff[q] = lds0 + (unsigned)((BM + l31) * DROW * 4) + sw;
    }
    // S16: row (lane % 16) of a 16-row group, chunk (lane / 16) of its high digits (+ 4: low digits)
    const int l15 = lane & 15, g4 = lane >> 4;
    unsigned a16[2][2], b16[2];
#pragma unroll
    for (int dg = 0; dg < 2; ++dg) {
        const unsigned sw = (unsigned)(((g4 + 4 * dg) ^ ((l15 >> 1) & 7)) << 4);
        a16[0][dg] = lds0 + (unsigned)((w * 32 + l15) * DROW * 4) + sw;
        a16[1][dg] = lds0 + (unsigned)((w * 32 + 16 + l15) * DROW * 4) + sw;
        b16[dg] = lds0 + (unsigned)((BM + l15) * DROW * 4) + sw;
    }
    const unsigned ta16 = lds0 + (unsigned)(RING * DSLAB * 4) + (unsigned)((w * 32 + 4 * g4) * 8);
    const unsigned ta0 = lds0 + (unsigned)(RING * DSLAB * 4) + (unsigned)((w * 32 + 4 * h) * 8);
    const unsigned cur0 = lds0 + (unsigned)(RING * DSLAB * 4 + BM * 8);
    const unsigned sn0 = lds0 + (unsigned)(RING * DSLAB * 4 + BM * 8 + 3 * BM * 4);
    const unsigned sfa0 = sn0 + (unsigned)(SN * 3 * BN * 4) + (unsigned)((w * 32 + 4 * h) * 4);
    const unsigned sfa16 = sn0 + (unsigned)(SN * 3 * BN * 4) + (unsigned)((w * 32 + 4 * g4) * 4);
    const bool late = wu >= 4;   // wave-uniform: the second-dispatched half issues its DMA mid-slab
    int xcc = 0;                 // the XCD this block runs on (L2 affinity of the unit lists: speed only)
    if (SYM) {
        unsigned xr;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xr));
        xcc = (int)(xr & 7u);
    }
    // vector-memory operations a wave adds to the first slab of a tile: the column items' norms (and thresholds)
    const int nextra = wu < 2 ? (SYM ? 2 : 1) + (I8 ? 1 : 0) : 0;
    const float* __restrict__ cnorm = METRIC == AS_METRIC_L2 ? a.n32 : a.inorm32;
    const float* __restrict__ cthr = a.thr_col ? a.thr_col : a.n32;

    for (int u = blockIdx.x;; u += gridDim.x) {
        int rb, cs, t0, t1;
        if (SYM) {
            // longest units first, handed out through atomic cursors (the triangle's units differ in length): the list of
            // this block's XCD first -- its neighbours there work on the same row blocks and column pieces --, then the others'
            if (tid == 0) {
                if (a.xoff) {
                    int got = a.nunits;
                    for (int y = 0; y < 8 && got == a.nunits; ++y) {
                        const int x = (xcc + y) & 7;
                        const int len = a.xoff[x + 1] - a.xoff[x];
                        if (ld_l2(a.xcur + 16 * x) >= len) continue;   // (an empty list's cursor is left alone)
                        const int j = atomicAdd(a.xcur + 16 * x, 1);
                        if (j < len) got = a.xoff[x] + j;
                    }
                    *s_unit = got;
                } else {
                    *s_unit = atomicAdd(a.unit_ctr, 1);
                }
            }
            __syncthreads();   // the previous unit ended with a barrier: nobody still reads the old value
            u = *s_unit;
            if (u >= a.nunits) break;
            const int4 ud = a.units[u];
            rb = ud.x; t0 = ud.y; t1 = ud.z; cs = ud.w;
        } else {
            if (u >= units) break;
            rb = u / a.S;
            cs = u % a.S;
            t0 = (int)((int64_t)a.ntile * cs / a.S);
            t1 = (int)((int64_t)a.ntile * (cs + 1) / a.S);
        }
        const int64_t rowbase = a.r0 + (int64_t)rb * BM;
        if (tid < BM) {
            const int64_t rg = rowbase + tid;
            const bool valid = rg < a.r1;
            // second component: n_i (L2) or 1 / |x_i| (cosine) -- times the row's scale on the int8 image, where the cosine needs nothing else
            const float fai = I8 && valid ? a.a_fa[rg] : 1.0f;
            if (I8) s_fa[tid] = valid ? fai : 0.0f;
            if (COLLECT) {
                s_ta[tid] = make_float2(valid ? a.a_thr[rg] : -finf, valid ? (METRIC == AS_METRIC_L2 ? a.a_n32[rg] : a.a_inorm32[rg] * fai) : 0.0f);
                s_id[tid] = valid ? a.a_ids[rg] : -1;
                s_drop[tid] = 0;
            } else {
                const float ni = valid ? a.a_n32[rg] : 0.0f;
                float bound = METRIC == AS_METRIC_L2 ? a.epskey + a.coef * (ni + a.nmax) : a.epskey + a.coef;
                int dropped = 0;
                if (a.thr0 && valid) {
                    const float t0r = ld_l2(a.thr0 + rg);   // possibly tightened by other units of this row since the threshold pass
                    if (t0r < bound) {   // what the tighter start rejects is beyond the row's M-th smallest key: a drop
                        bound = t0r;
                        dropped = 1;
                    }
                }
                s_ta[tid] = make_float2(valid ? bound : -finf, valid ? (METRIC == AS_METRIC_L2 ? ni : a.a_inorm32[rg] * fai) : 0.0f);
                s_drop[tid] = dropped;
            }
            s_cur[tid] = 0;
        }
        const char* pa0 = (const char*)(a.xa + (size_t)(rowbase + wu * 32) * a.ld);
        // prefetch cursor: (tile, slab) of the next slab to issue, RING - 1 slabs ahead of the MFMAs
        int pct = t0, pks = 0, pbuf = 0, inflight = 0;
        bool young_first = false;   // the youngest slab in flight opens a tile (it carries the wave's extra operations)
        // (a macro, not a lambda: the by-reference closure of a lambda this size is left in scratch memory)
#define K2_ISSUE()                                                                                                          \
    do {                                                                                                                    \
        if (pct < t1) {                                                                                                     \
            const int64_t cb_ = ((int64_t)pct * a.tstride + a.tphase) * BN;                                                 \
            const char* sa_ = pa0 + pks * (BK * 4);                                                                         \
            const char* sb_ = (const char*)(a.x32 + (size_t)(cb_ + wu * 16) * a.ld) + pks * (BK * 4);                       \
            float* dst_ = Sl + pbuf * DSLAB;                                                                                \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                   \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sa_ + (size_t)(8 * j) * a.ld * 4 + ((j & 1) ? lo1 : lo0)), \
                                                 (__attribute__((address_space(3))) void*)(dst_ + (wu * 32 + 8 * j) * DROW), 16, 0, 0);          \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                   \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb_ + (size_t)(8 * j) * a.ld * 4 + ((j & 1) ? lo1 : lo0)), \
                                                 (__attribute__((address_space(3))) void*)(dst_ + BM * DROW + (wu * 16 + 8 * j) * DROW), 16, 0, 0); \
            young_first = pks == 0;                                                                                         \
            if (pks == 0 && wu < 2) {                                                                                       \
                /* always issued, addresses clamped instead of lanes masked: the ring's counts assume the operation */      \
                const int64_t cg_ = cb_ + wu * 64 + lane;                                                                   \
                float* sn_ = s_n + (pct & (SN - 1)) * 3 * BN + wu * 64;                                                     \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cnorm + cg_),              \
                                                 (__attribute__((address_space(3))) void*)sn_, 4, 0, 0);                    \
                if (SYM)                                                                                                    \
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cthr + (cg_ < a.n ? cg_ : a.n - 1)), \
                                                     (__attribute__((address_space(3))) void*)(sn_ + BN), 4, 0, 0);         \
                if (I8)                                                                                                     \
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.fa + cg_),           \
                                                     (__attribute__((address_space(3))) void*)(sn_ + 2 * BN), 4, 0, 0);     \
            }                                                                                                               \
            pbuf = pbuf + 1 == RING ? 0 : pbuf + 1;                                                                         \
            if (++pks == nslab) {                                                                                           \
                pks = 0;                                                                                                    \
                ++pct;                                                                                                      \
            }                                                                                                               \
            ++inflight;                                                                                                     \
        }                                                                                                                   \
    } while (0)
        // the previous unit's last reads of the ring and of s_n are behind its closing barrier
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) K2_ISSUE();
        AS_LDS_FENCE();
        __builtin_amdgcn_s_barrier();   // s_ta / s_cur / s_drop of this unit are in place (no vmcnt(0): the ring is filling)
        unsigned cbuf = 0;   // byte offset of the slab buffer the MFMAs read next
        for (int ct = t0; ct < t1; ++ct) {
            const int64_t colbase = ((int64_t)ct * a.tstride + a.tphase) * BN;
            f32x16 acc[4];
            i32x16 acc1[4], accx[4];   // int8 image: a1.b1 and the cross terms a1.b2 + a2.b1
            i32x4k c1[2][8], cx[2][8];   // ... as 16 x 16 accumulators (S16): [16-row group][16-column group]
            if (S16) {
#pragma unroll
                for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                    for (int cg = 0; cg < 8; ++cg) {
                        c1[rg][cg] = i32x4k{0, 0, 0, 0};
                        cx[rg][cg] = i32x4k{0, 0, 0, 0};
                    }
            }
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (S16) {
                    } else if (I8) {
                        acc1[nn][r] = 0;
                        accx[nn][r] = 0;
                    } else {
                        acc[nn][r] = 0.0f;
                    }
                }
            for (int ks = 0; ks < nslab; ++ks) {
                // the slab has landed once only the younger slab's operations are outstanding (operations retire in
                // issue order; anything else in flight -- appends -- only makes the wait stricter)
                if (!K2_DIAG(2)) ring_wait(inflight >= 2 ? NPIECE + (young_first ? nextra : 0) : 0);
                if (!K2_DIAG(16)) __builtin_amdgcn_s_barrier();   // everybody's pieces are in, everybody is done with the buffer issued into next
                --inflight;
                if (!late && !K2_DIAG(2)) K2_ISSUE();
                if (S16) {
                    // one k-step of 64 columns: 4 A fragments, then the 16-column groups in two batches of four
                    i32x4k fa1[2], fa2[2], hb[4], lb[4];
                    lds_frag_a16(a16[0][0] + cbuf, a16[1][0] + cbuf, a16[0][1] + cbuf, a16[1][1] + cbuf, fa1[0], fa1[1], fa2[0], fa2[1]);
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        lds_frag_b16(b16[0] + cbuf + (unsigned)(hf * 8192), b16[1] + cbuf + (unsigned)(hf * 8192), hb, lb);
                        if (!K2_DIAG(1)) {
#pragma unroll
                            for (int rg = 0; rg < 2; ++rg) {
#pragma unroll
                                for (int j = 0; j < 4; ++j)   // a1 . b1
                                    c1[rg][4 * hf + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa1[rg], hb[j], c1[rg][4 * hf + j], 0, 0, 0);
#pragma unroll
                                for (int j = 0; j < 4; ++j)   // a1 . b2
                                    cx[rg][4 * hf + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa1[rg], lb[j], cx[rg][4 * hf + j], 0, 0, 0);
#pragma unroll
                                for (int j = 0; j < 4; ++j)   // a2 . b1
                                    cx[rg][4 * hf + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa2[rg], hb[j], cx[rg][4 * hf + j], 0, 0, 0);
                            }
                        }
                        if (hf == 0 && late && !K2_DIAG(2)) K2_ISSUE();
                    }
                    cbuf = cbuf + DSLAB * 4 == RING * DSLAB * 4 ? 0 : cbuf + DSLAB * 4;
                    continue;
                }
                f32x4 fa, fb[4], ga, gb[4];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (!K2_DIAG(8)) {
                        lds_frag5(aoff[s] + cbuf, boff[s] + cbuf, fa, fb[0], fb[1], fb[2], fb[3]);          // heads of k-step s
                        lds_frag5(aoff[2 + s] + cbuf, boff[2 + s] + cbuf, ga, gb[0], gb[1], gb[2], gb[3]);  // tails
                    } else {
                        asm volatile("" : "=v"(fa), "=v"(fb[0]), "=v"(fb[1]), "=v"(fb[2]), "=v"(fb[3]));
                        asm volatile("" : "=v"(ga), "=v"(gb[0]), "=v"(gb[1]), "=v"(gb[2]), "=v"(gb[3]));
                    }
                    if (I8 && !K2_DIAG(1)) {
#pragma unroll
                        for (int nn = 0; nn < 4; ++nn)   // a1 . b1
                            acc1[nn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, fa), __builtin_bit_cast(i32x4, fb[nn]), acc1[nn], 0, 0, 0);
#pragma unroll
                        for (int nn = 0; nn < 4; ++nn)   // a1 . b2
                            accx[nn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, fa), __builtin_bit_cast(i32x4, gb[nn]), accx[nn], 0, 0, 0);
#pragma unroll
                        for (int nn = 0; nn < 4; ++nn)   // a2 . b1
                            accx[nn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, ga), __builtin_bit_cast(i32x4, fb[nn]), accx[nn], 0, 0, 0);
                    } else if (!K2_DIAG(1)) {
#pragma unroll
                        for (int nn = 0; nn < 4; ++nn)   // xh . yh
                            acc[nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb[nn]), acc[nn], 0, 0, 0);
#pragma unroll
                        for (int nn = 0; nn < 4; ++nn)   // xh . yl
                            acc[nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, gb[nn]), acc[nn], 0, 0, 0);
#pragma unroll
                        for (int nn = 0; nn < 4; ++nn)   // xl . yh
                            acc[nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ga), __builtin_bit_cast(bf16x8, fb[nn]), acc[nn], 0, 0, 0);
                    } else {
                        acc[0][0] += fa[0] + fb[0][0] + fb[1][1] + fb[2][2] + fb[3][3] + ga[0] + gb[0][0] + gb[1][1] + gb[2][2] + gb[3][3];
                    }
                    if (s == 0 && late && !K2_DIAG(2)) K2_ISSUE();
                }
                cbuf = cbuf + DSLAB * 4 == RING * DSLAB * 4 ? 0 : cbuf + DSLAB * 4;
            }
            if (K2_DIAG(4)) {
                if (S16) {
#pragma unroll
                    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                        for (int cg = 0; cg < 8; ++cg) asm volatile("" ::"v"(c1[rg][cg]), "v"(cx[rg][cg]));
                }
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) {
                    if (S16) continue;
                    if (I8) asm volatile("" ::"v"(acc1[nn]), "v"(accx[nn]));
                    else asm volatile("" ::"v"(acc[nn]));
                }
                continue;
            }
            if (S16) {
                // ---- epilogue of the 16 x 16 accumulators: a lane holds 8 columns (16 cg + lane % 16) of rows 16 rg + 4 (lane / 16) + i;
                // sixteen lanes share a row.  Keys, bound test, transposed appends and appends as below, column order kept.
                float nj8[8], tj8[8], fj8[8], sj8[8], cb8[8];
                int cj8[8];
                {
                    const unsigned sna = sn0 + (unsigned)(((ct & (SN - 1)) * 3 * BN + l15) * 4);
                    lds_read_b32x8(sna, nj8);
                    if (SYM) lds_read_b32x8(sna + BN * 4, tj8);
                    lds_read_b32x8(sna + 2 * BN * 4, fj8);
#pragma unroll
                    for (int cg = 0; cg < 8; ++cg) {
                        cj8[cg] = (int)(colbase + cg * 16 + l15);
                        if (!SYM) tj8[cg] = finf;
                    }
                }
                {
                    const int mycnt = lds_read_i32(cur0 + (unsigned)((w * 32 + l31) * 4));
                    unsigned long long need = __ballot(lane < 32 && mycnt > CAP - BN);
                    if (COLLECT) {
                        if (need) {
                            const int rl = w * 32 + l31;
                            if (lane < 32 && s_cur[rl] > CAP - BN) {
                                s_ta[rl].x = -finf;
                                s_drop[rl] = 2;
                            }
                            AS_LDS_FENCE();
                        }
                        need = 0;
                    }
                    while (need) {
                        const int r = __ffsll((long long)need) - 1;
                        const unsigned rr = w * 32 + r;
                        compact_row_reg(a.M, bkey + rr * CAP, bidx + rr * CAP, s_cur[rr], &s_ta[rr].x, s_cur + rr, s_drop + rr,
                                        SYM && a.thr_pub ? a.thr_pub + (rowbase + rr) : nullptr);
                        need &= need - 1;
                    }
                }
                const int64_t colg = a.col_goff + colbase, rowg = a.row_goff + rowbase;
                const bool edge = COLLECT || colbase + BN > a.n || (colg < rowg + BM && colg + BN > rowg);
                const bool transp = SYM && a.t_cnt && (a.t_all || colbase >= rowbase + BM);
#pragma unroll
                for (int cg = 0; cg < 8; ++cg) {
                    sj8[cg] = METRIC == AS_METRIC_L2 ? fj8[cg] : nj8[cg] * fj8[cg];
                    cb8[cg] = !transp || cj8[cg] >= (int)a.n ? -finf : (METRIC == AS_METRIC_L2 ? a.epskey + a.coef * (nj8[cg] + a.nmax) : a.epskey + a.coef);
                    if (transp && a.thr_col && cj8[cg] < (int)a.n) cb8[cg] = fminf(cb8[cg], tj8[cg]);
                }
#pragma unroll
                for (int rg = 0; rg < 2; ++rg) {
                    // (bound, norm) of rows 16 rg + 4 (lane / 16) + {0..3}: 32 contiguous bytes; their scales: 16
                    f32x4 tg0, tg1, fg = {1, 1, 1, 1};
                    lds_read_b128x2(ta16 + (unsigned)(128 * rg), tg0, tg1);
                    if (METRIC == AS_METRIC_L2) fg = lds_read_b128(sfa16 + (unsigned)(64 * rg));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int rl = w * 32 + rg * 16 + 4 * g4 + i;
                        const f32x4 tq = (i & 2) ? tg1 : tg0;
                        const float thr = (i & 1) ? tq[2] : tq[0], ai = (i & 1) ? tq[3] : tq[1];
                        const float fi = METRIC == AS_METRIC_L2 ? fg[i] : ai;
                        float key[8];
#pragma unroll
                        for (int cg = 0; cg < 8; ++cg) {
                            const float t = fmaf((float)c1[rg][cg][i], 128.0f, (float)cx[rg][cg][i]);
                            const float gg = t * (fi * sj8[cg]);
                            key[cg] = METRIC == AS_METRIC_L2 ? fmaf(-2.0f, gg, ai + nj8[cg]) : 1.0f - fmaxf(0.0f, gg);
                        }
                        const bool rowok = rowbase + rl < a.r1;
                        bool any_t = false;
                        if (transp) {
#pragma unroll
                            for (int cg = 0; cg < 8; ++cg) any_t = any_t || (rowok && key[cg] <= cb8[cg]);
                        }
                        if (transp && __ballot(any_t)) {
                            const int rg_ = (int)(rowg + rl);
#pragma unroll
                            for (int cg = 0; cg < 8; ++cg)
                                if (rowok && key[cg] <= cb8[cg]) {
                                    const int slot = atomicAdd(a.t_cnt + cj8[cg], 1);
                                    if (slot < a.t_cap) {
                                        a.t_key[(size_t)cj8[cg] * a.t_cap + slot] = key[cg];
                                        a.t_idx[(size_t)cj8[cg] * a.t_cap + slot] = rg_;
                                    }
                                }
                        }
                        if (edge) {
                            const float excl = COLLECT ? __int_as_float(0x7fc00000) : finf;
                            const int rg_ = COLLECT ? s_id[rl] : (int)(rowg + rl);
#pragma unroll
                            for (int cg = 0; cg < 8; ++cg)
                                if (cj8[cg] >= (int)a.n || cj8[cg] + (int)a.col_goff == rg_) key[cg] = excl;
                        }
                        float kmin = key[0];
#pragma unroll
                        for (int cg = 1; cg < 8; ++cg) kmin = fminf(kmin, key[cg]);
                        if (__ballot(kmin <= thr)) {
#pragma unroll
                            for (int cg = 0; cg < 8; ++cg) {
                                const bool p = key[cg] <= thr;
                                const unsigned long long mk = __ballot(p);
                                if (!mk) continue;
                                const unsigned gm = (unsigned)(mk >> (16 * g4)) & 0xffffu;   // this row's sixteen lanes
                                const int base = s_cur[rl];
                                if (p) {
                                    const unsigned slot = (unsigned)rl * CAP + base + __popc(gm & ((1u << l15) - 1u));
                                    bkey[slot] = key[cg];
                                    bidx[slot] = cj8[cg] + (int)a.col_goff;
                                }
                                AS_CBAR();
                                s_cur[rl] = base + __popc(gm);
                                AS_CBAR();
                            }
                        }
                    }
                }
                continue;
            }
            // ---- epilogue: keys, bound test, append (32 rows per wave).  The tile's norm line was issued with its first
            // slab and waited for with it; a tile of one slab has not met a later wait yet.
            float nj[4], tj[4] = {finf, finf, finf, finf}, fj[4] = {1.0f, 1.0f, 1.0f, 1.0f};
            int cj[4];
            {
                const unsigned sna = sn0 + (unsigned)(((ct & (SN - 1)) * 3 * BN + l31) * 4);
                lds_read_b32x4(sna, nj[0], nj[1], nj[2], nj[3]);
                if (SYM) lds_read_b32x4(sna + BN * 4, tj[0], tj[1], tj[2], tj[3]);
                if (I8) lds_read_b32x4(sna + 2 * BN * 4, fj[0], fj[1], fj[2], fj[3]);
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) cj[nn] = (int)(colbase + nn * 32 + l31);
            }
            {
                const int mycnt = lds_read_i32(cur0 + (unsigned)((w * 32 + l31) * 4));
                unsigned long long need = __ballot(lane < 32 && mycnt > CAP - BN);
                if (COLLECT) {
                    // nothing may be dropped here: a row whose band does not fit stops collecting and is reported
                    if (need) {
                        const int rl = w * 32 + l31;
                        if (lane < 32 && s_cur[rl] > CAP - BN) {
                            s_ta[rl].x = -finf;
                            s_drop[rl] = 2;
                        }
                        AS_LDS_FENCE();
                    }
                    need = 0;
                }
                while (need) {
                    const int r = __ffsll((long long)need) - 1;
                    const unsigned rr = w * 32 + r;
                    compact_row_reg(a.M, bkey + rr * CAP, bidx + rr * CAP, s_cur[rr], &s_ta[rr].x, s_cur + rr, s_drop + rr,
                                    SYM && a.thr_pub ? a.thr_pub + (rowbase + rr) : nullptr);
                    need &= need - 1;
                }
            }
            float sj[4];    // per column: what the integer sums are multiplied by besides the row's factor
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) sj[nn] = METRIC == AS_METRIC_L2 ? fj[nn] : nj[nn] * fj[nn];
            const int64_t colg = a.col_goff + colbase, rowg = a.row_goff + rowbase;   // global ids of the tile's corner
            const bool edge = COLLECT || colbase + BN > a.n || (colg < rowg + BM && colg + BN > rowg);
            // symmetric mode: tiles strictly above the row block also serve the column items' rows (the diagonal tiles
            // hold both (i, j) and (j, i) themselves)
            const bool transp = SYM && a.t_cnt && (a.t_all || colbase >= rowbase + BM);
            f32x4 tg0 = {0, 0, 0, 0}, tg1 = {0, 0, 0, 0}, fg = {1, 1, 1, 1};
            float cb[4];
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) {
                cb[nn] = !transp || cj[nn] >= (int)a.n ? -finf : (METRIC == AS_METRIC_L2 ? a.epskey + a.coef * (nj[nn] + a.nmax) : a.epskey + a.coef);
                if (transp && a.thr_col && cj[nn] < (int)a.n) cb[nn] = fminf(cb[nn], tj[nn]);   // as of the tile's first slab: a stale bound is only less tight
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = w * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                // (bound, norm) of rows 4 h + {0..3} + 8 g, read per group of four registers (g = r >> 2): the accumulators of
                // the int8 form leave no room to hold all sixteen pairs at once
                if ((r & 3) == 0) {
                    lds_read_b128x2(ta0 + (unsigned)(64 * (r >> 2)), tg0, tg1);
                    if (I8 && METRIC == AS_METRIC_L2) fg = lds_read_b128(sfa0 + (unsigned)(32 * (r >> 2)));
                }
                const f32x4 tq = (r & 2) ? tg1 : tg0;
                const float thr = (r & 1) ? tq[2] : tq[0], ai = (r & 1) ? tq[3] : tq[1];
                float key[4];
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) {
                    if (I8) {
                        // x_i . x_j = s_i s_j (16384 a1.b1 + 128 (a1.b2 + a2.b1)) / 16256^2 = t fa_i fa_j, t = 128 a1.b1 + cross (both sums < 2^24 -- exact floats -- up to 1 040 columns; one rounding each beyond: charged in err_coef_i8)
                        const float t = fmaf((float)acc1[nn][r], 128.0f, (float)accx[nn][r]);
                        const float fi = METRIC == AS_METRIC_L2 ? fg[r & 3] : ai;
                        const float gg = t * (fi * sj[nn]);
                        key[nn] = METRIC == AS_METRIC_L2 ? fmaf(-2.0f, gg, ai + nj[nn]) : 1.0f - fmaxf(0.0f, gg);
                    } else {
                        const float gg = acc[nn][r];
                        key[nn] = METRIC == AS_METRIC_L2 ? fmaf(-2.0f, gg, ai + nj[nn]) : 1.0f - fmaxf(0.0f, gg * (ai * nj[nn]));
                    }
                }
                bool any_t = false;
                if (transp) {
                    const bool rowok = rowbase + rl < a.r1;
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn) any_t = any_t || (rowok && key[nn] <= cb[nn]);
                }
                if (transp && __ballot(any_t)) {
                    const int rg = (int)(rowg + rl);
                    const bool rowok = rowbase + rl < a.r1;
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn)
                        if (rowok && key[nn] <= cb[nn]) {
                            const int slot = atomicAdd(a.t_cnt + cj[nn], 1);
                            if (slot < a.t_cap) {
                                a.t_key[(size_t)cj[nn] * a.t_cap + slot] = key[nn];
                                a.t_idx[(size_t)cj[nn] * a.t_cap + slot] = rg;
                            }
                        }
                }
                if (edge) {  // wave-uniform: only tiles on the diagonal or at the padded tail pay for the exclusions
                    // collect mode: an excluded entry must fail `key <= thr` even against an infinite band: NaN
                    const float excl = COLLECT ? __int_as_float(0x7fc00000) : finf;
                    const int rg = COLLECT ? s_id[rl] : (int)(rowg + rl);
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn)
                        if (cj[nn] >= (int)a.n || cj[nn] + (int)a.col_goff == rg) key[nn] = excl;
                }
                const float kmin = fminf(fminf(key[0], key[1]), fminf(key[2], key[3]));
                if (__ballot(kmin <= thr)) {
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn) {
                        const bool p = key[nn] <= thr;
                        const unsigned long long mk = __ballot(p);
                        if (!mk) continue;
                        const unsigned hm = h ? (unsigned)(mk >> 32) : (unsigned)mk;
                        const int base = s_cur[rl];
                        if (p) {
                            const unsigned slot = (unsigned)rl * CAP + base + __popc(hm & ((1u << l31) - 1u));
                            bkey[slot] = key[nn];
                            bidx[slot] = cj[nn] + (int)a.col_goff;
                        }
                        AS_CBAR();
                        s_cur[rl] = base + __popc(hm);
                        AS_CBAR();
                    }
                }
            }
        }
        // ---- finalize this unit's rows (each wave: its 32 rows); the ring is empty
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        AS_CBAR();
        for (int r = 0; r < 32; ++r) {
            const unsigned rl = w * 32 + r;
            const int64_t rg = rowbase + rl;
            if (rg >= a.r1) break;
            if (!COLLECT && s_cur[rl] > a.M)
                compact_row_reg(a.M, bkey + rl * CAP, bidx + rl * CAP, s_cur[rl], &s_ta[rl].x, s_cur + rl, s_drop + rl,
                                SYM && a.thr_pub ? a.thr_pub + rg : nullptr);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int cnt = s_cur[rl];
            const size_t ob = ((size_t)(rg - a.r0) * a.S + cs) * a.M;
            for (int t = lane; t < cnt; t += 64) {
                a.out_key[ob + t] = ld_l2(bkey + rl * CAP + t);
                a.out_idx[ob + t] = ld_l2(bidx + rl * CAP + t);
            }
            if (lane == 0) {
                a.out_cnt[(size_t)(rg - a.r0) * a.S + cs] = (int)((unsigned)cnt | ((unsigned)s_drop[rl] << 30));
                if (!COLLECT && a.out_thr) a.out_thr[rg] = s_ta[rl].x;   // M-th smallest key seen (or the start bound): threshold pass, S = 1
            }
        }
        __syncthreads();
#undef K2_ISSUE
    }
}

constexpr size_t K2BF_LDS = sizeof(float) * RING * DSLAB + sizeof(float2) * BM + sizeof(int) * 3 * BM + sizeof(float) * SN * 3 * BN + sizeof(float) * BM + 16;

as_status launch_k2_bf16(const KnnArgs& ka, int metric, bool collect, bool sym, int grid, hipStream_t st, bool i8) {
    if (ka.ld <= 0 || ka.nslab <= 0 || (i8 && (!ka.fa || !ka.a_fa))) {
        set_err("launch_k2_bf16: operand image geometry not set");
        return AS_EINVAL;
    }
#ifdef AS_ABLATION
    if (const char* e = getenv("ARROWSPACE_K2_DIAG")) {   // timing skeletons of the symmetric L2 kernel
        const int dg = atoi(e);
        if (dg && sym && metric == AS_METRIC_L2) {
#define AS_K2D(DD)                                                                                                                 \
    case DD:                                                                                                                       \
        AS_HIP(hipFuncSetAttribute((const void*)knn_bf16_kernel<AS_METRIC_L2, false, true, DD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K2BF_LDS)); \
        hipLaunchKernelGGL((knn_bf16_kernel<AS_METRIC_L2, false, true, DD>), dim3(grid), dim3(512), K2BF_LDS, st, ka);             \
        break;
            switch (dg) {
                AS_K2D(4) AS_K2D(5) AS_K2D(6) AS_K2D(12) AS_K2D(13) AS_K2D(14) AS_K2D(15) AS_K2D(30) AS_K2D(22) AS_K2D(20)
                default:
                    set_err("ARROWSPACE_K2_DIAG=%d is not compiled", dg);
                    return AS_EINVAL;
            }
#undef AS_K2D
            AS_HIP(hipGetLastError());
            return AS_OK;
        }
    }
#endif
    // (A/B: ARROWSPACE_K2_MFMA16=0 keeps the 32x32x32 shape on the int8 image)
    static const bool s16 = !(getenv("ARROWSPACE_K2_MFMA16") && atoi(getenv("ARROWSPACE_K2_MFMA16")) == 0);
#define AS_K2B(MM, CC, SS)                                                                                                        \
    do {                                                                                                                          \
        if (i8 && s16) {                                                                                                          \
            AS_HIP(hipFuncSetAttribute((const void*)knn_bf16_kernel<MM, CC, SS, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K2BF_LDS)); \
            hipLaunchKernelGGL((knn_bf16_kernel<MM, CC, SS, 0, true, true>), dim3(grid), dim3(512), K2BF_LDS, st, ka);             \
        } else if (i8) {                                                                                                          \
            AS_HIP(hipFuncSetAttribute((const void*)knn_bf16_kernel<MM, CC, SS, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K2BF_LDS)); \
            hipLaunchKernelGGL((knn_bf16_kernel<MM, CC, SS, 0, true>), dim3(grid), dim3(512), K2BF_LDS, st, ka);                   \
        } else {                                                                                                                  \
            AS_HIP(hipFuncSetAttribute((const void*)knn_bf16_kernel<MM, CC, SS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K2BF_LDS)); \
            hipLaunchKernelGGL((knn_bf16_kernel<MM, CC, SS>), dim3(grid), dim3(512), K2BF_LDS, st, ka);                            \
        }                                                                                                                         \
    } while (0)
    if (metric == AS_METRIC_L2) {
        if (collect) AS_K2B(AS_METRIC_L2, true, false);
        else if (sym) AS_K2B(AS_METRIC_L2, false, true);
        else AS_K2B(AS_METRIC_L2, false, false);
    } else {
        if (collect) AS_K2B(AS_METRIC_COSINE, true, false);
        else if (sym) AS_K2B(AS_METRIC_COSINE, false, true);
        else AS_K2B(AS_METRIC_COSINE, false, false);
    }
#undef AS_K2B
    AS_HIP(hipGetLastError());
    return AS_OK;
}

}  // namespace as
