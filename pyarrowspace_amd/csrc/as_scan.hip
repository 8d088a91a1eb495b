// K7a of the search chain on gfx950: dots[i] = x_i . q over the fp32 item matrix, with the k-NN prefilter fused
// into the epilogue (DESIGN.md sections 5.4, 5.5).  One HBM pass per query (scan_dma_kernel, scan_dots_f32_kernel)
// or per 32 queries (scan_gemm_kernel: bf16 head + tail products on the matrix pipe, or fp32 MFMA).  Replaces the scan inside `search_lambda_aware` and
// `prepare_query_item` (/root/reference/src/lib.rs:154,173).
#include "as_query.hpp"

// ARROWSPACE_SC_DBG (1 no publication, 2 no histogram read, 4 no candidates: the fused tail's cost breakdown, WRONG results)
// is compiled into `make ABLATION=1` builds only; the product library's kernels do not test it.
#ifdef AS_ABLATION
#define AS_SC_DBG(bit) (pre.sc_dbg & (bit))
#else
#define AS_SC_DBG(bit) 0
#endif

namespace as {

#ifdef AS_STAMPS
// diagnostic build only (make STAMPS=1): when every wave of the last tile scan started and ended (100 MHz wall clock): the
// launch's ramp and tail (tools/scan_stamps.py).  No product code reads them.
__device__ unsigned long long g_scan_stamps[2 * 4096];
#endif

// ------------------------------------------------------------------ K7a scan: dots[i] = x_i . q  (+ k-NN prefilter)

// aux = n32[row] (L2) or inorm32[row] (cosine), loaded by the caller together with the row
// full: set once this lane has seen the buffer overflow -- the counter only has to exceed CAND_CAP, and a
// neighbourhood of most of the items would otherwise serialise a million atomics on one address
__device__ __forceinline__ void prefilter_f32(const PreArgs& p, int64_t row, float dot, float aux, float nq32, float inq32, int& full) {
    if (!p.enabled || full || row >= p.n || row == p.exclude) return;
    float key, bound;
    if (p.metric == AS_METRIC_L2) {
        key = fmaf(-2.0f, dot, aux + nq32);
        bound = ((float)p.epskey + (float)p.coef * (aux + nq32)) * 1.000001f;
    } else {
        key = 1.0f - fmaxf(0.0f, dot * aux * inq32);
        bound = ((float)p.epskey + (float)p.coef) * 1.000001f;
    }
    if (key <= bound) {
        const int slot = atomicAdd(&p.infow->knn_cnt, 1);
        if (slot < CAND_CAP) {
            ((float*)p.ckey)[slot] = key;
            p.cidx[slot] = (int)row;
        } else {
            full = 1;
        }
    }
}

// HBM-bound: one wave per row, 16 B per lane per load, query fragment in registers,
// two rows in flight per wave.  NCH = ceil(dp / 256) chunks of 256 floats.
// NT: non-temporal row loads.  rev: walk the rows from the end -- consecutive queries alternate
// direction, so the tail of one scan (still in the 256 MiB Infinity Cache) is the head of the next.
template <int NCH, bool NT>
__global__ __launch_bounds__(256) void scan_dots_f32_kernel(const float* __restrict__ x32, const float* __restrict__ q32,
                                                            int64_t dp, int64_t r0, int64_t r1, float* __restrict__ dots,
                                                            PreArgs pre, int rev) {
    const int lane = lane_id();
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    const float nq32 = pre.info->nq32, inq32 = pre.info->inq32;
    int full = 0;
    const float* __restrict__ auxv = pre.metric == AS_METRIC_L2 ? pre.n32 : pre.inorm32;
    f32x4 qv[NCH];
    bool on[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        const int64_t c = 4 * (lane + 64 * u);
        on[u] = c < dp;
        qv[u] = on[u] ? *(const f32x4*)(q32 + c) : f32x4{0, 0, 0, 0};
    }
    auto ld = [&](const float* p) -> f32x4 { return NT ? __builtin_nontemporal_load((const f32x4*)p) : *(const f32x4*)p; };
    const int64_t last = r0 + r1 - 1;   // rev: logical row t maps to physical row last - t
    int64_t row = r0 + gw;
    for (; row + nw < r1; row += 2 * nw) {
        const int64_t ra = rev ? last - row : row, rb = rev ? last - (row + nw) : row + nw;
        const float* pa = x32 + ra * dp + 4 * lane;
        const float* pb = x32 + rb * dp + 4 * lane;
        f32x4 va[NCH], vb[NCH];
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            va[u] = on[u] ? ld(pa + 256 * u) : f32x4{0, 0, 0, 0};
            vb[u] = on[u] ? ld(pb + 256 * u) : f32x4{0, 0, 0, 0};
        }
        // lanes 0 / 1 own the two results; their aux value is in flight with the row loads
        const int64_t myrow = (lane & 1) ? rb : ra;
        const float aux = auxv[myrow];
        float sa = 0.0f, sb = 0.0f;
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sa = fmaf(va[u][e], qv[u][e], sa);
                sb = fmaf(vb[u][e], qv[u][e], sb);
            }
        }
        sa = wave_sum(sa);
        sb = wave_sum(sb);
        if (lane < 2) {
            const float dot = lane ? sb : sa;
            dots[myrow] = dot;
            prefilter_f32(pre, myrow, dot, aux, nq32, inq32, full);
        }
    }
    if (row < r1) {
        const int64_t ra = rev ? last - row : row;
        const float* pa = x32 + ra * dp + 4 * lane;
        float sa = 0.0f;
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            if (on[u]) {
                const f32x4 v = ld(pa + 256 * u);
#pragma unroll
                for (int e = 0; e < 4; ++e) sa = fmaf(v[e], qv[u][e], sa);
            }
        }
        sa = wave_sum(sa);
        if (lane == 0) {
            dots[ra] = sa;
            prefilter_f32(pre, ra, sa, auxv[ra], nq32, inq32, full);
        }
    }
}

// Batched scan (as_search_batch): QB query fragments live in registers, every row is read from
// HBM once for QB queries.  The QB partial sums per lane are reduced with a halving butterfly
// (4+2+1 exchanges, then 3 on the single survivor): lane L with (L & 7) == 0 ends up owning
// query ((L>>5)&1)*4 + ((L>>4)&1)*2 + ((L>>3)&1).
template <int NCH>
__global__ __launch_bounds__(256) void scan_dots_batch_kernel(const float* __restrict__ x32, const float* __restrict__ q32,
                                                              int64_t dp, int64_t r0, int64_t r1, float* __restrict__ dots,
                                                              int64_t sd, int64_t ts, int rs, int slot0, PreArgs pre) {
    const int lane = lane_id();
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    const float* __restrict__ auxv = pre.metric == AS_METRIC_L2 ? pre.n32 : pre.inorm32;
    f32x4 qv[QB][NCH];
    bool on[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        const int64_t c = 4 * (lane + 64 * u);
        on[u] = c < dp;
#pragma unroll
        for (int b = 0; b < QB; ++b) qv[b][u] = on[u] ? *(const f32x4*)(q32 + (int64_t)b * dp + c) : f32x4{0, 0, 0, 0};
    }
    const int myq = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
    const bool owner = (lane & 7) == 0;
    const float nq32 = pre.info[myq].nq32, inq32 = pre.info[myq].inq32;
    int full = 0;
    PreArgs mine = pre;
    mine.info = pre.info + myq;
    mine.infow = pre.infow + myq;
    mine.ckey = (void*)((float*)pre.ckey + (int64_t)myq * CAND_CAP);
    mine.cidx = pre.cidx + (int64_t)myq * CAND_CAP;
    float* __restrict__ mydots = dots + dots_slot_off(slot0 + myq, sd, rs);   // tile-major (SlotStride)
    for (int64_t row = r0 + gw; row < r1; row += nw) {
        const float* pa = x32 + row * dp + 4 * lane;
        f32x4 v[NCH];
#pragma unroll
        for (int u = 0; u < NCH; ++u) v[u] = on[u] ? __builtin_nontemporal_load((const f32x4*)(pa + 256 * u)) : f32x4{0, 0, 0, 0};
        const float aux = auxv[row];
        float acc[QB];
#pragma unroll
        for (int b = 0; b < QB; ++b) acc[b] = 0.0f;
#pragma unroll
        for (int u = 0; u < NCH; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int b = 0; b < QB; ++b) acc[b] = fmaf(v[u][e], qv[b][u][e], acc[b]);
        // halving butterfly: keep the half selected by this lane's bit, send the other half
        float k4[4], k2[2], k1;
        {
            const bool hi = (lane >> 5) & 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float keep = hi ? acc[j + 4] : acc[j];
                const float send = hi ? acc[j] : acc[j + 4];
                k4[j] = keep + __shfl_xor(send, 32, 64);
            }
        }
        {
            const bool hi = (lane >> 4) & 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float keep = hi ? k4[j + 2] : k4[j];
                const float send = hi ? k4[j] : k4[j + 2];
                k2[j] = keep + __shfl_xor(send, 16, 64);
            }
        }
        {
            const bool hi = (lane >> 3) & 1;
            const float keep = hi ? k2[1] : k2[0];
            const float send = hi ? k2[0] : k2[1];
            k1 = keep + __shfl_xor(send, 8, 64);
        }
        k1 += __shfl_xor(k1, 4, 64);
        k1 += __shfl_xor(k1, 2, 64);
        k1 += __shfl_xor(k1, 1, 64);
        if (owner) {
            mydots[(row >> 5) * ts + (row & 31) * rs] = k1;
            prefilter_f32(mine, row, k1, aux, nq32, inq32, full);
        }
    }
}

// Batched scan as a GEMM (rows up to 768 floats): dots[GQ x rows] = Q . X^T on the matrix pipe -- fp32
// (v_mfma_f32_32x32x2_f32, A = 32 queries, B = 32 item rows) or, by default, bf16 head + tail (BF3, below).  A block is a team of 4 waves
// that splits K: wave w keeps the Q fragments of its quarter of the columns in registers for the
// whole launch (<= 6 slabs of 32 floats -> 96 VGPRs) and streams the matching quarter of every
// 32-row block by LDS-DMA into a private ring of NBUF XOR-swizzled slabs (same image as
// knn_mfma_dma_kernel) -- the K loop has no block barrier, only the wave's own vmcnt, and all
// of LDS is staging (2 blocks per CU, ~100 KB of rows in flight per CU).  The four partial
// 32x32 tiles meet in LDS once per row block; wave w then owns queries [8w, 8w+8) of the
// epilogue (store + fused kNN prefilter).  One HBM pass serves GQ queries: 2*GQ*dp flops per
// row against dp*4 bytes -- still HBM-bound at GQ=32.

// LDS accesses of the MFMA scan go through inline asm: the compiler orders every LDS read it can see after
// *all* outstanding LDS-DMA (s_waitcnt vmcnt(0)), which would drain the prefetch ring at each slab.
// Each asm block waits for its own reads before it ends, so no register the compiler may copy or reuse ever
// holds data that is still in flight.
__device__ __forceinline__ void lds_read4x4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, f32x4& x0, f32x4& x1, f32x4& x2, f32x4& x3) {
    asm volatile(
        "ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
        : "memory");
}
__device__ __forceinline__ void lds_read4x3(unsigned a0, unsigned a1, unsigned a2, f32x4& x0, f32x4& x1, f32x4& x2) {
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x0), "=&v"(x1), "=&v"(x2)
                 : "v"(a0), "v"(a1), "v"(a2)
                 : "memory");
}
__device__ __forceinline__ float lds_read1(unsigned a) {
    float v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
    return v;
}
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate).  Only the counts the
// slab ring produces get an exact wait; anything else waits for everything (always correct).  A 14-way switch
// at every slab of the unrolled K loop pushed the kernel into scratch spills.
__device__ __forceinline__ void wait_vmcnt(int n) {
    if (n == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// always issued (an exec-masked store the compiler may not branch around): the wave's count of outstanding
// operations must never be smaller than the ring's bookkeeping assumes
__device__ __forceinline__ void store_dword_issued(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_x4_issued(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_x2_issued(void* p, u32x2 v) {
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
// two floats -> two fp16 (round to nearest even) in one dword
__device__ __forceinline__ unsigned int pack_half2(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned int, v);
}
// s_nop: the hazard recogniser does not look inside asm, and an MFMA result may be the operand
__device__ __forceinline__ void lds_write4(unsigned addr, f32x4 v) {
    asm volatile("s_nop 15\n\ts_nop 3\n\tds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// AUX: cache policy of the row DMA (2 = nt: rows are read once per launch; measured 7-10 % faster than the default).
// DIAG = 1 (measurement only, wrong results): no MFMA -- the memory side alone (0.59 ms of the kernel's 0.61).
// Measured on that skeleton: fetching the same bytes as 2 rows x 512 B or 1 row x 1 KiB per instruction
// instead of 8 rows x 128 B: no change; without the norm DMA: no change; without the 4 dot stores per row
// block: 0.46 ms (7.0 TB/s); with nt stores: 0.52 ms -- but in the full kernel nt stores were slower (0.63 ms)
// and cost the per-slot selection kernels their cache hits.  The cost is per byte, not per instruction or per
// episode: 16 dword stores per row block, 4 dwordx4 stores, or the stores of 4 row blocks issued together all
// measure the same (0.60 ms): 134 MB of write-backs among 3 GB of streamed reads cost what 0.9 GB of reads would.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// eight fp32 values -> their bf16 head and the bf16 of what the head leaves: v = hi + lo + r, |r| <= 2^-16 |v| (two
// roundings to nearest of 2^-8 each: bf16 carries 8 significant bits; the subtraction is exact).  A non-finite value keeps its head and gets a zero tail (inf - inf would be NaN).
__device__ __forceinline__ void split_bf16(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
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
}

// BF3: the products on the bf16 matrix pipe (16 x the fp32 rate), each operand as head + tail: q.x ~ qh.xh + qh.xl + ql.xh --
// three v_mfma_f32_32x32x16_bf16 per 16 columns instead of eight v_mfma_f32_32x32x2_f32; bf16 x bf16 is exact in the fp32
// accumulator, what is dropped (ql.xl and the two remainders) is at most 3 * 2^-16 |q_k x_k| per term (coef_query).
// I8: items and queries as int8 two-digit images (as_k2bf.hip, quant_i8_kernel; the queries' image by the host,
// q_quant_batch_kernel in as_search.hip): a slab row is 64 columns (64 bytes of a1, 64 of a2) -- HALF the bytes of the fp32 items --, `dp` the image
// row in floats, the chunk addressing that of the fp32 slab; three v_mfma_i32_32x32x32_i8 per 32 columns into two int32
// accumulators (q1.a1 and the cross terms), scaled by the rows' and the slots' factors in the epilogue.
template <int NBUF, int DIAG, int AUX, bool BF3 = false, int NSW = GEMM_NSW, bool I8 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void scan_gemm_kernel(
    const float* __restrict__ x32, const float* __restrict__ q32, int64_t dp, int64_t r0, int64_t r1, float* __restrict__ dots,
    int64_t ts, PreArgs pre, int nb, int half_dots, int64_t kbase_, int64_t kcols_, int raw, int npass, int64_t pstride) {
    // A wave keeps the query fragments of NSW slabs in registers: 4 * NSW * 32 columns per launch (768 at NSW = 6; the
    // bf16 form also exists with NSW = 8: 1024).  Wider rows are scanned in K-CHUNK passes: a launch covers the columns
    // [kbase, kbase + kcols) and, `raw`, leaves its fp32 partial dots in `dots` (tile-major, no epilogue);
    // gemm_combine_kernel adds the passes' partials and does the epilogue.  The npass passes of a wide row share ONE launch
    // (block b works on pass b % npass: one ramp instead of npass, and the passes balance each other's tails): pass p covers
    // the columns [p kcols_, ...) and writes the partial buffer p (pstride floats apart).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    float* St = (float*)smem;   // per wave: NBUF slabs x [32 rows][32 floats]; Ex[owner wave][3 senders][64 lanes][4]; Ax[wave][64]
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ex0 = lds0 + 4 * NBUF * 4096, ax0 = ex0 + 4 * 3 * 64 * 16;
    const unsigned fx0 = ax0 + 4 * 64 * 4;   // I8: the row block's scales, one 256-byte line per wave
    typedef int i32x16 __attribute__((ext_vector_type(16)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pz = npass > 1 ? (int)(blockIdx.x % (unsigned)npass) : 0;
    const int64_t bx = npass > 1 ? blockIdx.x / (unsigned)npass : blockIdx.x, gx = npass > 1 ? gridDim.x / (unsigned)npass : gridDim.x;
    const int64_t kbase = kbase_ + (int64_t)pz * kcols_;
    const int64_t kcols = min(kcols_, dp - kbase);
    dots += (int64_t)pz * pstride;
    const int nslab = (int)(kcols / 32), nsw = (nslab + 3) / 4;
    const int ks0 = wu * nsw;
    const int myns = max(0, min(nsw, nslab - ks0));
    f32x4 qf[NSW][4];
#pragma unroll
    for (int ks = 0; ks < NSW; ++ks)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qf[ks][s] = ks < myns ? *(const f32x4*)(q32 + (int64_t)l31 * dp + kbase + (ks0 + ks) * 32 + (2 * s + h) * 4) : f32x4{0, 0, 0, 0};
    // per-query constants of the prefilter, for the 4 queries this lane finishes: b = e + 8 wu + 4 h
    float nqv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) nqv[e] = pre.metric == AS_METRIC_L2 ? pre.info[e + 8 * wu + 4 * h].nq32 : pre.info[e + 8 * wu + 4 * h].inq32;
    float iqv[4];   // 1/|q| of the same queries: the stored value is the cosine (half_dots)
#pragma unroll
    for (int e = 0; e < 4; ++e) iqv[e] = pre.info[e + 8 * wu + 4 * h].inq32;
    float fqv[4] = {1.0f, 1.0f, 1.0f, 1.0f};   // I8: the same queries' scales s_q sqrt(128) / 16256
    if (I8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) fqv[e] = pre.faqv[e + 8 * wu + 4 * h];
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(fqv[e]));
    }
    // the loads above complete here, once: otherwise the compiler has to assume they are still pending inside
    // the loop and puts a vmcnt(0) -- which also waits for the whole prefetch ring -- in front of their first use
#pragma unroll
    for (int ks = 0; ks < NSW; ++ks)
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[ks][s]));
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(nqv[e]));
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(iqv[e]));
    bf16x8 qh[NSW][2], ql[NSW][2];
    if (BF3) {
#pragma unroll
        for (int ks = 0; ks < NSW; ++ks) {
            split_bf16(qf[ks][0], qf[ks][1], qh[ks][0], ql[ks][0]);
            split_bf16(qf[ks][2], qf[ks][3], qh[ks][1], ql[ks][1]);
        }
    }
    float* my = St + wu * NBUF * 1024;
    const unsigned my0 = lds0 + wu * NBUF * 4096;
    const int drow = lane >> 3;
    const int csw0 = (lane & 7) ^ ((lane >> 4) & 7), csw1 = (lane & 7) ^ ((4 + (lane >> 4)) & 7);
    const unsigned lo0 = (unsigned)((drow * dp + csw0 * 4) * 4), lo1 = (unsigned)((drow * dp + csw1 * 4) * 4);
    unsigned foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = my0 + (unsigned)(l31 * 128 + (((2 * s + h) ^ ((l31 >> 1) & 7)) << 4));
    const int64_t nrb = (r1 - r0 + 31) / 32;
    const float* __restrict__ auxv = pre.metric == AS_METRIC_L2 ? pre.n32 : pre.inorm32;
    // prefetch cursor: the wave's slab sequence (row block, k slab), NBUF-1 slabs ahead of the MFMAs
    int64_t prb = myns > 0 ? bx : nrb;
    int pks = 0, pbuf = 0, inflight = 0;
    // x0/x1/x2: vector-memory operations other than slab DMAs issued after the oldest / 2nd / 3rd slab in
    // flight -- a lower bound: the norm DMA and the 4 dot stores of each row block (rare appends add more and
    // only make the wait stricter).  Operations retire in issue order, so "oldest slab landed" == "at most
    // 4 (inflight - 1) + x0 operations outstanding"; counting the stores keeps them off the critical path
    // (waiting for them too cost 23 % of the launch).
    int x0 = 0, x1 = 0, x2 = 0;
    // (a macro, not a lambda: the by-reference closure of a lambda this size was left in scratch memory)
#define AS_ISSUE_SLAB()                                                                                                   \
    do {                                                                                                                  \
        if (prb < nrb) {                                                                                                  \
            /* 4 x 1 KiB pieces per 32-row slab: piece j = rows [8j, 8j+8) */                                             \
            const char* base_ = (const char*)(x32 + (size_t)(r0 + prb * 32) * dp + kbase + (ks0 + pks) * 32);             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                               \
                const char* src_ = base_ + (size_t)(8 * j) * dp * 4 + ((j & 1) ? lo1 : lo0);                              \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,                     \
                                                 (__attribute__((address_space(3))) void*)(my + pbuf * 1024 + 8 * j * 32), 16, 0, AUX); \
            }                                                                                                             \
            pbuf = pbuf + 1 == NBUF ? 0 : pbuf + 1;                                                                       \
            if (++pks == myns) {                                                                                          \
                pks = 0;                                                                                                  \
                prb += gx;                                                                                                \
            }                                                                                                             \
            if (inflight == 0) x0 = 0;                                                                                    \
            else if (inflight == 1) x1 = 0;                                                                               \
            else x2 = 0;                                                                                                  \
            ++inflight;                                                                                                   \
        }                                                                                                                 \
    } while (0)
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) AS_ISSUE_SLAB();
    unsigned cur = 0;   // byte offset of the slab the MFMAs read next
    int full = 0;       // bit e: query e of this lane has overflowed its candidate buffer (see prefilter_f32)
    for (int64_t rb = bx; rb < nrb; rb += gx) {
        f32x16 acc;
        i32x16 acc1, accx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r] = 0.0f;
            if (I8) {
                acc1[r] = 0;
                accx[r] = 0;
            }
        }
        // the row's norm for the prefilter: issued before this block's slabs, so it is older than every DMA still in
        // flight at the epilogue when the wave has at least NBUF-1 slabs per block
        const int64_t row = r0 + rb * 32 + l31;
        {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(auxv + row),   // padded arrays: readable
                                             (__attribute__((address_space(3))) void*)(St + 4 * NBUF * 1024 + 4 * 3 * 64 * 4 + wu * 64), 4, 0, 0);
            if (I8)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pre.fa8 + row),
                                                 (__attribute__((address_space(3))) void*)(St + 4 * NBUF * 1024 + 4 * 3 * 64 * 4 + 4 * 64 + wu * 64), 4, 0, 0);
            x0 += 1;
            x1 += 1;
            x2 += 1;
        }
#pragma unroll
        for (int ks = 0; ks < NSW; ++ks) {
            if (ks < myns) {
                // slab `cur` has landed once at most inflight-1 newer slabs (4 DMA ops each) are outstanding:
                // loads retire in order, so the count is conservative whatever else is in flight
                wait_vmcnt(4 * (inflight - 1) + (x0 >= 2 ? 2 : 0));
                --inflight;
                x0 = x1;
                x1 = x2;
                AS_ISSUE_SLAB();   // overwrites the slab read in the previous iteration (its ds_reads were consumed by MFMAs)
                f32x4 x0, x1, x2, x3;
                lds_read4x4(foff[0] + cur, foff[1] + cur, foff[2] + cur, foff[3] + cur, x0, x1, x2, x3);
                if (DIAG == 1) {
                    acc[0] += x0[0] + x1[1] + x2[2] + x3[3];
                    cur = cur + 4096 == NBUF * 4096 ? 0 : cur + 4096;
                    continue;
                }
                if (I8) {
                    // qf[ks][0..3] = q1 of k-steps 0, 1, q2 of k-steps 0, 1; x0..x3 = a1, a1, a2, a2 likewise
                    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, qf[ks][0]), __builtin_bit_cast(i32x4, x0), acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, qf[ks][1]), __builtin_bit_cast(i32x4, x1), acc1, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, qf[ks][0]), __builtin_bit_cast(i32x4, x2), accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, qf[ks][1]), __builtin_bit_cast(i32x4, x3), accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, qf[ks][2]), __builtin_bit_cast(i32x4, x0), accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, qf[ks][3]), __builtin_bit_cast(i32x4, x1), accx, 0, 0, 0);
                } else if (BF3) {
                    // lane (row l31, half h) holds the same 8 + 8 columns of its query and of its item: the order of the
                    // columns inside an instruction is free as long as both operands agree
                    bf16x8 bh0, bl0, bh1, bl1;
                    split_bf16(x0, x1, bh0, bl0);
                    split_bf16(x2, x3, bh1, bl1);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[ks][0], bh0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[ks][1], bh1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[ks][0], bl0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[ks][1], bl1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ql[ks][0], bh0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ql[ks][1], bh1, acc, 0, 0, 0);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[ks][0][t], x0[t], acc, 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[ks][1][t], x1[t], acc, 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[ks][2][t], x2[t], acc, 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[ks][3][t], x3[t], acc, 0, 0, 0);
                }
                cur = cur + 4096 == NBUF * 4096 ? 0 : cur + 4096;
            }
        }
        // C[i = query][j = row]: register r <-> query (r&3) + 8 (r>>2) + 4 h, lane <-> row l31.  Wave o owns
        // registers [4o, 4o+4) = queries 8o + {0..3} + 4h; the other three waves send it their partials.
        // Raw barriers: __syncthreads() carries a vmcnt(0) fence that would drain the prefetch ring.
        if (I8) {   // 128 q1.a1 + cross terms: both sums of a wave's quarter of the columns are below 2^24 (exact floats)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaf((float)acc1[r], 128.0f, (float)accx[r]);
        }
        __builtin_amdgcn_s_barrier();   // the previous row block's exchange has been read (its reads were waited for)
        f32x4 mine = {0, 0, 0, 0};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const f32x4 part = {acc[4 * o], acc[4 * o + 1], acc[4 * o + 2], acc[4 * o + 3]};
            if (o != wu) lds_write4(ex0 + (unsigned)(((o * 3 + (wu < o ? wu : wu - 1)) * 64 + lane) * 16), part);
            else mine = part;
        }
        AS_LDS_FENCE();
        __builtin_amdgcn_s_barrier();
        {
            f32x4 p0, p1, p2;
            const unsigned pa = ex0 + (unsigned)((wu * 3 * 64 + lane) * 16);
            lds_read4x3(pa, pa + 1024, pa + 2048, p0, p1, p2);
            mine += p0;
            mine += p1;
            mine += p2;
        }
        // the norms: older than the slabs in flight when those were all issued inside this row block
        wait_vmcnt(myns >= NBUF - 1 ? 4 * inflight : 0);
        const float aux = lds_read1(ax0 + (unsigned)((wu * 64 + lane) * 4));
        if (I8 && !raw) {   // x . q = (128 q1.a1 + cross) fa_row fa_q
            const float far = lds_read1(fx0 + (unsigned)((wu * 64 + lane) * 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) mine[e] *= far * fqv[e];
        }
        if (raw) {   // a K-chunk pass: the partial dots of this chunk, fp32, in the tile layout (gemm_combine_kernel finishes)
            store_x4_issued(dots + (row >> 5) * ts + ((2 * wu + h) * 32 + (row & 31)) * 4, mine);
            x0 += 1;
            x1 += 1;
            x2 += 1;
            continue;
        }
        {
            // the store is issued whatever the lane holds: rows past r1 land in the padding behind the last tile
            // (np + ROW_TILE rows are allocated), idle slots have their places in every tile.  Tile layout
            // [slot quad][32 rows][4 slots]: the lane's four values (slots 8 wu + 4 h + {0..3} of row l31) are 16
            // contiguous bytes, the wave's ONE dwordx4 store fills 1 KiB of the row block's contiguous 4 KiB
            // half_dots: the same places, two bytes each -- the cosines (in [-1, 1]: no range problem whatever the items'
            // magnitudes) rounded to fp16; 67 MB of write-backs per pass instead of 134, and the selection kernels behind
            // read half as much and need no norms.  The neighbour prefilter below keeps the fp32 dots in registers.
            if (half_dots) {
                const float inr = pre.metric == AS_METRIC_L2 ? (aux > 0.0f ? rsqrtf(aux) : 0.0f) : aux;
                u32x2 pk;
                pk[0] = pack_half2(mine[0] * inr * iqv[0], mine[1] * inr * iqv[1]);
                pk[1] = pack_half2(mine[2] * inr * iqv[2], mine[3] * inr * iqv[3]);
                store_x2_issued((char*)dots + ((row >> 5) * ts + ((2 * wu + h) * 32 + (row & 31)) * 4) * 2, pk);
            } else {
                store_x4_issued(dots + (row >> 5) * ts + ((2 * wu + h) * 32 + (row & 31)) * 4, mine);
            }
            x0 += 1;
            x1 += 1;
            x2 += 1;
            const bool pf = pre.enabled && row < r1 && row < pre.n && row != pre.exclude;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int b = e + 8 * wu + 4 * h;
                if (b >= nb) continue;   // idle slot
                const float dot = mine[e];
                if (pf && !((full >> e) & 1)) {
                    float key, bound;
                    if (pre.metric == AS_METRIC_L2) {
                        key = fmaf(-2.0f, dot, aux + nqv[e]);
                        bound = ((float)pre.epskey + (float)pre.coef * (aux + nqv[e])) * 1.000001f;
                    } else {
                        key = 1.0f - fmaxf(0.0f, dot * aux * nqv[e]);
                        bound = ((float)pre.epskey + (float)pre.coef) * 1.000001f;
                    }
                    if (key <= bound) {
                        const int slot = atomicAdd(&pre.infow[b].knn_cnt, 1);
                        if (slot < CAND_CAP) {
                            ((float*)pre.ckey)[(int64_t)b * CAND_CAP + slot] = key;
                            pre.cidx[(int64_t)b * CAND_CAP + slot] = (int)row;
                        } else {
                            full |= 1 << e;
                        }
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef AS_ISSUE_SLAB
}

// The int8 batched pass for TWO workspaces at once: 64 queries per read of the items.  A block's wave keeps the query fragments of its
// quarter of the columns in registers -- at up to 768 columns (image rows of up to 384 floats, 3 slabs per wave) the single-set
// kernel's six slabs of fragments are half empty: the second workspace's 32 queries take the other half.  One read of every slab
// from LDS feeds both sets' products (12 MFMAs per slab instead of 6); the epilogue (exchange of the waves' partial sums, scales,
// fp16 cosines, k-NN prefilter) runs once per set, each on its own workspace's buffers.  Measured before (rocprofv3 trace of the
// 256-query bench call): the two workspaces' scans ran side by side and shared the HBM -- 291 us alone, 458 us overlapped --
// and the first one's selection kernels starved behind the second scan (knn_finish 265 us instead of 30).
struct DualArgs {
    PreArgs p[2];
    const float* q[2];   // the sets' query images ([32 slots][ld floats])
    float* dots[2];
    int nb[2];
};
template <int NBUF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void scan_gemm_dual_kernel(
    const float* __restrict__ x32, int64_t dp, int64_t r0, int64_t r1, int64_t ts, DualArgs da) {
    constexpr int NS = 3;   // slabs per wave and set
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef int i32x16 __attribute__((ext_vector_type(16)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    float* St = (float*)smem;   // the single-set kernel's layout: slabs, exchange, norms, scales
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ex0 = lds0 + 4 * NBUF * 4096, ax0 = ex0 + 4 * 3 * 64 * 16, fx0 = ax0 + 4 * 64 * 4;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t bx = blockIdx.x, gx = gridDim.x;
    const int nslab = (int)(dp / 32), nsw = (nslab + 3) / 4;
    const int ks0 = wu * nsw;
    const int myns = max(0, min(nsw, nslab - ks0));
    f32x4 qf[2][NS][4];
    float nqv[2][4], iqv[2][4], fqv[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int ks = 0; ks < NS; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                qf[s][ks][j] = ks < myns ? *(const f32x4*)(da.q[s] + (int64_t)l31 * dp + (ks0 + ks) * 32 + (2 * j + h) * 4) : f32x4{0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const QInfo* in = da.p[s].info + (e + 8 * wu + 4 * h);
            nqv[s][e] = da.p[s].metric == AS_METRIC_L2 ? in->nq32 : in->inq32;
            iqv[s][e] = in->inq32;
            fqv[s][e] = da.p[s].faqv[e + 8 * wu + 4 * h];
        }
    }
    // (the loads complete here, once: see scan_gemm_kernel)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int ks = 0; ks < NS; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(qf[s][ks][j]));
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(nqv[s][e]), "+v"(iqv[s][e]), "+v"(fqv[s][e]));
    }
    float* my = St + wu * NBUF * 1024;
    const unsigned my0 = lds0 + wu * NBUF * 4096;
    const int drow = lane >> 3;
    const int csw0 = (lane & 7) ^ ((lane >> 4) & 7), csw1 = (lane & 7) ^ ((4 + (lane >> 4)) & 7);
    const unsigned lo0 = (unsigned)((drow * dp + csw0 * 4) * 4), lo1 = (unsigned)((drow * dp + csw1 * 4) * 4);
    unsigned foff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) foff[j] = my0 + (unsigned)(l31 * 128 + (((2 * j + h) ^ ((l31 >> 1) & 7)) << 4));
    const int64_t nrb = (r1 - r0 + 31) / 32;
    const float* __restrict__ auxv = da.p[0].metric == AS_METRIC_L2 ? da.p[0].n32 : da.p[0].inorm32;
    int64_t prb = myns > 0 ? bx : nrb;
    int pks = 0, pbuf = 0, inflight = 0;
    int x0 = 0, x1 = 0, x2 = 0;   // (other vector-memory operations behind the oldest / 2nd / 3rd slab in flight: scan_gemm_kernel)
#define AS_ISSUE_SLAB()                                                                                                   \
    do {                                                                                                                  \
        if (prb < nrb) {                                                                                                  \
            const char* base_ = (const char*)(x32 + (size_t)(r0 + prb * 32) * dp + (ks0 + pks) * 32);                     \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                               \
                const char* src_ = base_ + (size_t)(8 * j) * dp * 4 + ((j & 1) ? lo1 : lo0);                              \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,                     \
                                                 (__attribute__((address_space(3))) void*)(my + pbuf * 1024 + 8 * j * 32), 16, 0, 2); \
            }                                                                                                             \
            pbuf = pbuf + 1 == NBUF ? 0 : pbuf + 1;                                                                       \
            if (++pks == myns) {                                                                                          \
                pks = 0;                                                                                                  \
                prb += gx;                                                                                                \
            }                                                                                                             \
            if (inflight == 0) x0 = 0;                                                                                    \
            else if (inflight == 1) x1 = 0;                                                                               \
            else x2 = 0;                                                                                                  \
            ++inflight;                                                                                                   \
        }                                                                                                                 \
    } while (0)
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) AS_ISSUE_SLAB();
    unsigned cur = 0;
    int full = 0;   // bit 4 s + e: query e of set s (of this lane) has overflowed its candidate buffer
    for (int64_t rb = bx; rb < nrb; rb += gx) {
        i32x16 acc1[2], accx[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc1[s][r] = 0;
                accx[s][r] = 0;
            }
        const int64_t row = r0 + rb * 32 + l31;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(auxv + row),   // padded arrays: readable
                                         (__attribute__((address_space(3))) void*)(St + 4 * NBUF * 1024 + 4 * 3 * 64 * 4 + wu * 64), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(da.p[0].fa8 + row),
                                         (__attribute__((address_space(3))) void*)(St + 4 * NBUF * 1024 + 4 * 3 * 64 * 4 + 4 * 64 + wu * 64), 4, 0, 0);
        x0 += 1;
        x1 += 1;
        x2 += 1;
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
            if (ks < myns) {
                wait_vmcnt(4 * (inflight - 1) + (x0 >= 2 ? 2 : 0));
                --inflight;
                x0 = x1;
                x1 = x2;
                AS_ISSUE_SLAB();
                f32x4 v0, v1, v2, v3;   // a1 of k-steps 0, 1; a2 of k-steps 0, 1
                lds_read4x4(foff[0] + cur, foff[1] + cur, foff[2] + cur, foff[3] + cur, v0, v1, v2, v3);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    acc1[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, qf[s][ks][0]), __builtin_bit_cast(i32x4, v0), acc1[s], 0, 0, 0);
                    acc1[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, qf[s][ks][1]), __builtin_bit_cast(i32x4, v1), acc1[s], 0, 0, 0);
                    accx[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, qf[s][ks][0]), __builtin_bit_cast(i32x4, v2), accx[s], 0, 0, 0);
                    accx[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, qf[s][ks][1]), __builtin_bit_cast(i32x4, v3), accx[s], 0, 0, 0);
                    accx[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, qf[s][ks][2]), __builtin_bit_cast(i32x4, v0), accx[s], 0, 0, 0);
                    accx[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, qf[s][ks][3]), __builtin_bit_cast(i32x4, v1), accx[s], 0, 0, 0);
                }
                cur = cur + 4096 == NBUF * 4096 ? 0 : cur + 4096;
            }
        }
        float aux = 0.0f, far = 0.0f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // the exchange of scan_gemm_kernel, once per set: wave o owns registers [4 o, 4 o + 4) = queries 8 o + {0..3} + 4 h
            __builtin_amdgcn_s_barrier();   // the previous exchange has been read
            f32x4 mine = {0, 0, 0, 0};
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                f32x4 part;
#pragma unroll
                for (int e = 0; e < 4; ++e) part[e] = fmaf((float)acc1[s][4 * o + e], 128.0f, (float)accx[s][4 * o + e]);
                if (o != wu) lds_write4(ex0 + (unsigned)(((o * 3 + (wu < o ? wu : wu - 1)) * 64 + lane) * 16), part);
                else mine = part;
            }
            AS_LDS_FENCE();
            __builtin_amdgcn_s_barrier();
            {
                f32x4 p0, p1, p2;
                const unsigned pa = ex0 + (unsigned)((wu * 3 * 64 + lane) * 16);
                lds_read4x3(pa, pa + 1024, pa + 2048, p0, p1, p2);
                mine += p0;
                mine += p1;
                mine += p2;
            }
            if (s == 0) {
                // the norms and scales: older than the slabs in flight, which were all issued inside this row block (myns = NBUF - 1)
                wait_vmcnt(myns >= NBUF - 1 ? 4 * inflight : 0);
                aux = lds_read1(ax0 + (unsigned)((wu * 64 + lane) * 4));
                far = lds_read1(fx0 + (unsigned)((wu * 64 + lane) * 4));
            }
            const PreArgs& pre = da.p[s];
#pragma unroll
            for (int e = 0; e < 4; ++e) mine[e] *= far * fqv[s][e];
            {
                const float inr = pre.metric == AS_METRIC_L2 ? (aux > 0.0f ? rsqrtf(aux) : 0.0f) : aux;
                u32x2 pk;
                pk[0] = pack_half2(mine[0] * inr * iqv[s][0], mine[1] * inr * iqv[s][1]);
                pk[1] = pack_half2(mine[2] * inr * iqv[s][2], mine[3] * inr * iqv[s][3]);
                store_x2_issued((char*)da.dots[s] + ((row >> 5) * ts + ((2 * wu + h) * 32 + (row & 31)) * 4) * 2, pk);
            }
            x0 += 1;
            x1 += 1;
            x2 += 1;
            const bool pf = pre.enabled && row < r1 && row < pre.n && row != pre.exclude;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int b = e + 8 * wu + 4 * h;
                if (b >= da.nb[s]) continue;   // idle slot
                const float dot = mine[e];
                if (pf && !((full >> (4 * s + e)) & 1)) {
                    float key, bound;
                    if (pre.metric == AS_METRIC_L2) {
                        key = fmaf(-2.0f, dot, aux + nqv[s][e]);
                        bound = ((float)pre.epskey + (float)pre.coef * (aux + nqv[s][e])) * 1.000001f;
                    } else {
                        key = 1.0f - fmaxf(0.0f, dot * aux * nqv[s][e]);
                        bound = ((float)pre.epskey + (float)pre.coef) * 1.000001f;
                    }
                    if (key <= bound) {
                        const int slot = atomicAdd(&pre.infow[b].knn_cnt, 1);
                        if (slot < CAND_CAP) {
                            ((float*)pre.ckey)[(int64_t)b * CAND_CAP + slot] = key;
                            pre.cidx[(int64_t)b * CAND_CAP + slot] = (int)row;
                        } else {
                            full |= 1 << (4 * s + e);
                        }
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef AS_ISSUE_SLAB
}

// Wave-wide sum on the DPP crossbar (6 VALU adds, no LDS round trips: a __shfl_xor butterfly is 6 dependent
// ds_bpermute, ~600 cycles of latency per row): xor-1 and xor-2 inside quads, half-row and row mirrors, then
// row_bcast15 / row_bcast31 carry the row sums up to lane 63.  Returns the total in every lane (readlane 63).
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));  // row_bcast15 -> rows 1, 3
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));  // row_bcast31 -> rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Single-query scan on the LDS-DMA ring (rows up to 1024 floats) -- the default.  The batched kernel's
// skeleton streams at 7.0 TB/s where the register-staged scan above stops at 6.3: every wave keeps NSLOT-1
// whole rows (NCH KiB each) in flight into a private LDS ring by `global_load_lds ... nt`, costs no VGPRs for
// it, and consumes the oldest row behind a counted vmcnt.  Rows are handed out in chunks of 64 consecutive rows
// per wave, round-robin over all waves (a moving window over the items), the remainder split evenly; lane r of
// the wave ends up with the dot of the chunk's row r, so norms arrive by one 256-byte DMA per chunk, the dots
// leave by one coalesced store and the k-NN prefilter runs lane-parallel.
// All LDS reads and the store are inline asm (see scan_gemm_kernel): nothing the compiler can see may force
// a vmcnt(0) inside the loop.
//
// SC (fused tail): the scan also collects the SCORER's candidates, before lambda_q exists.  With lambdas in [0, 1] the
// lambda term of a score lies in [(1 - tau) / 2, 1 - tau], so a row whose cosine is more than W = (1 - tau) / (2 tau)
// below the M-th largest cosine B cannot be among the M best scores.  B is learned on the way: at every chunk end a wave
// adds its chunk's best rows (the bins within two of the chunk's maximum, one atomic per bin) to a 64-bin histogram of
// cosines in QInfo, and reads the histogram back with the next chunk's norms (one more 256-byte DMA, agent scope): the
// lower edge of the highest bin with >= M rows at or above it bounds B from below whatever subset of rows was
// published.  Rows at or above (bound - W) wait in a per-wave LDS list, are re-tested against the bound the wave knows
// at its end and only then go to the candidate buffer -- the first chunks, scanned before any bound exists, leave
// nothing behind.  A dropped row has score < tau (B - W) + (1 - tau) <= the score of each of the >= M kept rows with
// cosine >= B: the M best scores are all in the buffer, and the finish kernel's a-posteriori proof applies unchanged.
constexpr int SC_PEND = 128;   // per-wave list of pending candidates (cosine, row)
constexpr int TILE_PEND = 256;  // ... of the tile scan (scan_chunk_end), GANG_PEND: per query of a gang scan
constexpr int GANG_PEND = 192;
constexpr int SC_BINS = 64;
__device__ __forceinline__ void lds_write2(unsigned addr, float a, int b) {   // 8-byte aligned
    const unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)(unsigned)b << 32);
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_read2(unsigned addr, float& a, int& b) {
    unsigned long long v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    a = __uint_as_float((unsigned)v);
    b = (int)(v >> 32);
}
__device__ __forceinline__ unsigned lds_read1u(unsigned a) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ int sc_bin(float c) {
    const int b = (int)floorf((c + 1.0f) * 32.0f);
    return b < 0 ? 0 : (b > SC_BINS - 1 ? SC_BINS - 1 : b);
}
// ---- what a scanning wave does at the end of a chunk of up to 64 rows (lane r holds the dot of the chunk's row r), shared by the
// row-ring scan (scan_dma_kernel) and the tile scan (scan_tile_kernel): the dots' store, the k-NN prefilter and -- SC -- the
// scorer's candidates: publication into the cosine histogram, the bound read back, the wave's pending list.
struct ScanWave {
    int full = 0;                // the k-NN candidate buffer has overflowed (this lane has seen it)
    int npend = 0;               // SC: entries of the wave's pending list
    float thr_last = -3.0e38f;   // SC: the threshold of the last chunk that read the histogram
    bool thr_known = false;
    int jb_last = -1;
    int sc_over = 0;             // SC: the list overflowed (the wave reports -1 candidates: the host takes the threshold chain)
    float floor = -3.0e38f;      // SC: the largest cosine the wave let go of while it had no bound tight enough (scan_chunk_end)
};
// PEND: entries of the wave's pending list.  Until the wave knows a bound (its third chunk at the earliest: the histogram must hold
// M rows) every row is kept: 128 entries last exactly to that third chunk -- enough when all waves of the launch run side by
// side, not when the launch shares the chip (another query's tail kernels on half the CUs when it starts: the waves that run
// ahead find too few rows published and report an overflow).  The tile kernels keep 256 / 192 entries and read the histogram at
// every chunk until they have a bound.
template <bool SC, int PEND = SC_PEND>
__device__ __forceinline__ void scan_chunk_end(const PreArgs& pre, ScanWave& w, float* __restrict__ dots, int t, int rounds, int64_t gw, int lane,
                                               int64_t base, int cnt, float mydot, float aux, bool hread, unsigned hx0, unsigned px0,
                                               float nq32, float inq32) {
    const int64_t row = base + lane;
    if (lane < cnt) {   // cnt >= 1: the store is always issued (the ring's bookkeeping counts it)
        store_dword_issued(dots + row, mydot);
        prefilter_f32(pre, row, mydot, aux, nq32, inq32, w.full);
    }
    if (SC) {
        int jb = w.jb_last;
        if (hread) {   // (between reads the wave keeps the bound of its last one)
            w.thr_last = sc_bound(lds_read1u(hx0 + lane * 4), pre.sc_m, lane, jb) - pre.sc_w;
            w.jb_last = jb;
            w.thr_known = true;
        }
        const float thr = w.thr_last;
        const bool valid = lane < cnt && row < pre.n;
        const float inr = pre.metric == AS_METRIC_L2 ? (aux > 0.0f ? rsqrtf(aux) : 0.0f) : aux;
        const float c = mydot * inr * inq32;
        // Publish the chunk's best rows: the bins within two of the chunk's maximum that lie above the bound's -- but
        // only rows that stand out of their own chunk (more than 4 standard deviations above its mean), or every
        // 16th chunk unconditionally.  Any subset of the rows bounds B from below; the rule keeps the atomics few:
        // a posted atomic is cheap for its wave, but every wave's next histogram read (past the L2, in order in front
        // of its row DMAs) queues behind the atomics in flight on that line -- 2 048 chunks publishing at once made
        // the whole launch 35 % slower.  A query's near neighbours are outliers of their chunks; smooth cosine
        // distributions are covered by the unconditional chunks.
        const bool fin = valid && c == c;           // (a NaN cosine neither publishes nor bounds)
        const int mybin = fin ? sc_bin(c) : -1;
        // (only a row above the bound's bin can raise the bound: once it stands, the chunk statistics below -- 18
        // cross-lane operations -- are skipped for almost every chunk)
        // (publishing only rows two or more bins above a standing bound was tried: no change in the launch's time -- the candidate
        // bookkeeping's 8 us are not the publications)
        if (__ballot(mybin > jb) && (t < rounds || rounds >= 2) && !AS_SC_DBG(1)) {
            float cm = fin ? c : -2.0f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cm = fmaxf(cm, __shfl_xor(cm, o, 64));
            const float nv = (float)__popcll(__ballot(fin));
            const float mean = wave_sum_dpp(fin ? c : 0.0f) / fmaxf(nv, 1.0f);
            const float var = fmaxf(wave_sum_dpp(fin ? (c - mean) * (c - mean) : 0.0f) / fmaxf(nv, 1.0f), 0.0f);
            const bool every = (((int)gw + t) & 31) == 0;
            const int bfloor = every ? -1 : sc_bin(mean + 4.0f * sqrtf(var));
            const int bmax = cm > -2.0f ? sc_bin(cm) : -1;
            int nb3[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) nb3[u] = __popcll(__ballot(mybin == bmax - u && bmax - u >= 0));
            // lane = copy * 3 + u: bin bmax - u of histogram copy `copy`.  (A wave of one or two chunks does not publish
            // its last one: nobody is left to read it but the waves' final reads, which would queue behind it.)
            const int u = lane % 3, copy = lane / 3;
            const int b = bmax - u;
            const int nb_ = u == 0 ? nb3[0] : (u == 1 ? nb3[1] : nb3[2]);
            if (copy < SC_COPIES && b > jb && b > bfloor && nb_ > 0) atomicAdd(&pre.sc_hist[copy * SC_HSTRIDE + b], (unsigned)nb_);
        }
        // candidates: c >= thr -- a NaN cosine (a poisoned row) never qualifies, as under the plain chain's `key <= thr`
        bool pass = valid && c >= thr && !AS_SC_DBG(4);
        unsigned long long pm = __ballot(pass);
        int np = __popcll(pm);
        if (w.npend + np > PEND) {
            // re-test the list against the bound known now (rows kept before a bound existed); flush what is left
            int keep = 0;
            for (int e0 = 0; e0 < w.npend; e0 += 64) {
                float ce = 0.0f;
                int re = 0;
                if (e0 + lane < w.npend) lds_read2(px0 + (unsigned)(e0 + lane) * 8, ce, re);
                const bool kp = e0 + lane < w.npend && ce >= thr;
                const unsigned long long km = __ballot(kp);
                // in place: the entries written are at or before the entries read by this or an earlier trip
                if (kp) lds_write2(px0 + (unsigned)(keep + __popcll(km & ((1ull << lane) - 1))) * 8, ce, re);
                AS_LDS_FENCE();
                keep += __popcll(km);
            }
            w.npend = keep;
            if (w.npend + np > PEND && (PEND + 63) / 64 <= 4) {
                // More than the list holds under the bound the wave knows -- none at all, when the launch shares the chip and the
                // waves that run ahead find too few rows published.  The wave then keeps the PEND rows of the LARGEST cosines (list
                // and this chunk's together) and remembers the largest cosine it let go (w.floor): lossless as long as the bound
                // the wave ends with lies above it (scan_wave_report says "overflow" otherwise).  The cut is found by bisection
                // over the order-preserving bit patterns of the cosines, entries in registers (up to four per lane); a rare path.
                auto okey = [](float f) -> unsigned { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
                unsigned ek[4];
                int er[4];
                float ec[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = lane + 64 * u;
                    ec[u] = 0.0f;
                    er[u] = 0;
                    if (e < w.npend) lds_read2(px0 + (unsigned)e * 8, ec[u], er[u]);
                    ek[u] = e < w.npend ? okey(ec[u]) : 0xffffffffu;   // (absent entries sort above everything: never counted)
                }
                const unsigned nk = pass ? okey(c) : 0xffffffffu;
                const int excess = w.npend + np - PEND;
                // smallest cut with at least `excess` entries at or below it
                unsigned lo = 0u, hi = 0xfffffffeu;
                while (lo < hi) {   // (wave-uniform: 32 trips)
                    const unsigned mid = lo + ((hi - lo) >> 1);
                    int cntl = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) cntl += __popcll(__ballot(ek[u] <= mid));
                    cntl += __popcll(__ballot(nk <= mid));
                    if (cntl >= excess) hi = mid;
                    else lo = mid + 1u;
                }
                const unsigned cut = lo;
                const unsigned cb_ = (cut & 0x80000000u) ? (cut & 0x7fffffffu) : ~cut;   // back to the float it came from
                w.floor = fmaxf(w.floor, __uint_as_float(cb_));
                int keep2 = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {   // (in place: written positions are at or before the positions read by this or an earlier trip)
                    const bool kp = ek[u] != 0xffffffffu && ek[u] > cut;
                    const unsigned long long km = __ballot(kp);
                    if (kp) lds_write2(px0 + (unsigned)(keep2 + __popcll(km & ((1ull << lane) - 1))) * 8, ec[u], er[u]);
                    AS_LDS_FENCE();
                    keep2 += __popcll(km);
                }
                w.npend = keep2;
                pass = pass && nk > cut;
                pm = __ballot(pass);
                np = __popcll(pm);
            }
            if (w.npend + np > PEND) {   // (a list of more than 256 entries: no selection) this query is not for the fused tail
                w.sc_over = 1;
                w.npend = 0;
            }
        }
        if (pass) lds_write2(px0 + (unsigned)(w.npend + __popcll(pm & ((1ull << lane) - 1))) * 8, c, (int)row);
        AS_LDS_FENCE();
        w.npend += np;
    }
}
// the wave's last word (SC): its surviving candidates into its own region of the report buffer
template <bool SC>
__device__ __forceinline__ void scan_wave_report(const PreArgs& pre, ScanWave& w, int64_t gw, int lane, unsigned px0) {
    if (SC) {
        // The wave's last word: the histogram as it stands now, the list re-tested against it, the survivors into the
        // wave's OWN region of the candidate buffer (SC_WCAP words: the count, then the rows -- one 16-byte load tells the
        // finish kernel the count and the first three) -- plain stores: a returning atomic per wave on one counter, all
        // waves ending together, cost the launch 20 us.
        // Late validation (pre.sc_late: the tail kernels recompute the bound from the FINAL histogram): a wave that ends with a
        // loose bound or none -- it ran ahead of the publications it needed: a launch that shares the chip, a scan of two or three
        // chunks per wave -- does not give up.  It reports its rows of the largest cosines (at most SC_WCAP - 2) and the largest
        // cosine it may have let go (w.floor, the cut of this selection) in the region's last word, flagged in the count; the
        // tail accepts the report when that cosine lies below the final bound, and only otherwise calls it an overflow.
        int keep = 0;
        int* region = pre.sc_idx + gw * SC_WCAP;   // [0] = number of candidates (-1: more than the region holds), [1 ..] = their rows
        const bool late = pre.sc_late != 0;
        float lossy = w.floor;
        bool flagged = false;
        if (w.npend > 0 && !w.sc_over) {
            // (a wave of three or more chunks has read the histogram with its last chunks: that threshold will do -- a
            // fresh read past the L2 at every wave's end is two microseconds of every launch's tail)
            float thr = w.thr_last;
            if (!w.thr_known) {
                int jb;
                const unsigned h = __hip_atomic_load(&pre.sc_hist[(gw & (SC_COPIES - 1)) * SC_HSTRIDE + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                thr = sc_bound(h, pre.sc_m, lane, jb) - pre.sc_w;
            }
            if (!(thr > w.floor)) {   // (rows the wave let go of might pass the bound it ends with)
                if (late) flagged = true;
                else w.sc_over = 1;
            }
            const int cap = late ? SC_WCAP - 2 : SC_WCAP - 1;
            // survivors, in registers (at most four entries per lane: the lists hold up to 256)
            auto okey = [](float f) -> unsigned { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
            unsigned ek[4];
            int er[4];
            int nsurv = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = lane + 64 * u;
                float ce = 0.0f;
                er[u] = 0;
                if (e < w.npend) lds_read2(px0 + (unsigned)e * 8, ce, er[u]);
                const bool kp = e < w.npend && ce >= thr;
                ek[u] = kp ? okey(ce) : 0u;   // (0: not a survivor -- below every real key)
                nsurv += __popcll(__ballot(kp));
            }
            unsigned cut = 0u;   // survivors with a key above the cut are reported
            if (nsurv > cap && late) {
                // the rows of the largest cosines: the smallest cut that leaves at most `cap` survivors above it
                unsigned lo = 1u, hi = 0xfffffffeu;
                while (lo < hi) {   // (wave-uniform: 32 trips; a rare path)
                    const unsigned mid = lo + ((hi - lo) >> 1);
                    int above = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) above += __popcll(__ballot(ek[u] > mid));
                    if (above <= cap) hi = mid;
                    else lo = mid + 1u;
                }
                cut = lo;
                const unsigned cb_ = (cut & 0x80000000u) ? (cut & 0x7fffffffu) : ~cut;
                lossy = fmaxf(lossy, __uint_as_float(cb_));
                flagged = true;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool kp = ek[u] > cut;
                const unsigned long long km = __ballot(kp);
                const int pos = keep + __popcll(km & ((1ull << lane) - 1));
                if (kp && pos < cap) region[1 + pos] = er[u];
                keep += __popcll(km);
            }
            if (keep > cap) w.sc_over = 1;   // (not late: more survivors than the region holds)
        } else if (w.npend == 0 && !w.sc_over && w.floor > -3.0e38f) {
            // (nothing pending, but rows were let go of: the same question against the bound the wave knows)
            float thr = w.thr_last;
            if (!w.thr_known) {
                int jb;
                const unsigned h = __hip_atomic_load(&pre.sc_hist[(gw & (SC_COPIES - 1)) * SC_HSTRIDE + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                thr = sc_bound(h, pre.sc_m, lane, jb) - pre.sc_w;
            }
            if (!(thr > w.floor)) {
                if (late) flagged = true;
                else w.sc_over = 1;
            }
        }
        if (lane == 0) {
            if (flagged && !w.sc_over) region[SC_WCAP - 1] = __float_as_int(lossy);
            region[0] = w.sc_over ? -1 : (flagged ? (keep | SC_REPORT_LOSSY) : keep);
        }
    }
}

// I8: the rows are those of the int8 two-digit image (as_k2bf.hip, quant_i8_kernel: x ~ s (128 a1 + a2) / 16256, per 64-column
// slab 64 bytes of a1 then 64 of a2) -- HALF the bytes of the fp32 items for this HBM-bound kernel; `dp` is then the image row
// in floats (dp8 / 2).  A lane's 16-byte chunk is 16 digits a1 or 16 digits a2 of 16 columns; the query's digits q1, q2 of
// those columns sit in the lane's registers in two roles (pre.q8): qa multiplies into the 16384-weighted sum (q1 for an a1
// chunk, nothing for an a2 chunk), qb into the 128-weighted one (q2 for an a1 chunk, q1 for an a2 chunk) -- eight
// v_dot4_i32_i8 per chunk, exact; the dot is (128 HI + XS) fa_row fa_q.  Like every fp32 dot of this path it is a
// PREFILTER value: what it costs in accuracy is in coef_query (the query's and the items' measured quantisation norms).
typedef int i32x4s __attribute__((ext_vector_type(4)));
template <int NCH, int NSLOT, bool SC = false, bool I8 = false>
__global__ __launch_bounds__(256) void scan_dma_kernel(const float* __restrict__ x32, const float* __restrict__ q32, int64_t dp,
                                                       int64_t r0, int64_t r1, float* __restrict__ dots, PreArgs pre, int rounds,
                                                       int tail_rows, int crows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RING = NSLOT * NCH * 1024;   // bytes per wave
    constexpr int WAVE_LDS = RING + 256 + (SC ? 256 + SC_PEND * 8 : 0) + (I8 ? 256 : 0);   // + the chunk's 64 norms (+ histogram, pending list) (+ the rows' scales)
    constexpr int K1 = NCH * (NSLOT - 2);      // DMA operations younger than the oldest row of a full ring
    constexpr int KB = I8 ? 3 : 2;             // operations of a chunk boundary behind its rows: the dots' store, the norm DMA (, the scale DMA)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* myp = smem + wu * WAVE_LDS;
    const unsigned my0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wu * WAVE_LDS;
    const unsigned ax0 = my0 + RING;
    const unsigned hx0 = ax0 + 256, px0 = hx0 + 256;   // SC: histogram landing area, pending list
    constexpr int FA_OFF = RING + 256 + (SC ? 256 + SC_PEND * 8 : 0);   // I8: the chunk's 64 row scales
    // lanes past the end of a row never receive DMA data: they must read zeros, not stale bits
    for (int i = lane; i < RING / 16; i += 64) *(f32x4*)(myp + i * 16) = f32x4{0, 0, 0, 0};
    f32x4 qv[NCH];
    i32x4s qa[NCH], qb[NCH];
    bool on[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        on[u] = 4 * (lane + 64 * u) < dp;
        if (I8) {
            qa[u] = on[u] ? *(const i32x4s*)(pre.q8 + 8 * (lane + 64 * u)) : i32x4s{0, 0, 0, 0};
            qb[u] = on[u] ? *(const i32x4s*)(pre.q8 + 8 * (lane + 64 * u) + 4) : i32x4s{0, 0, 0, 0};
        } else {
            qv[u] = on[u] ? *(const f32x4*)(q32 + 4 * (lane + 64 * u)) : f32x4{0, 0, 0, 0};
        }
    }
    float nq32 = pre.host_q ? pre.nq32 : pre.info->nq32, inq32 = pre.host_q ? pre.inq32 : pre.info->inq32;
    if (pre.host_q && blockIdx.x == 0 && tid == 0) {   // what q_prepare would have filed: read by the kernels behind the scan
        pre.infow->nq = pre.nq;
        pre.infow->inq = pre.inq;
        pre.infow->nq32 = pre.nq32;
        pre.infow->inq32 = pre.inq32;
        pre.infow->tau = 1.0;
    }
    if (pre.host_q && pre.q64_dev)   // the fp64 query for the kernels behind the scan: pinned host memory -> device (PreArgs)
        for (int g = (int)blockIdx.x * 256 + tid; g < pre.qdp; g += (int)gridDim.x * 256) pre.q64_dev[g] = pre.q64_host[g];
#pragma unroll
    for (int u = 0; u < NCH; ++u) {   // loads complete here, once (see scan_gemm_kernel)
        if (I8) asm volatile("" : "+v"(qa[u]), "+v"(qb[u]));
        else asm volatile("" : "+v"(qv[u]));
    }
    asm volatile("" : "+v"(nq32), "+v"(inq32));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zero fill
    const float* __restrict__ auxv = pre.metric == AS_METRIC_L2 ? pre.n32 : pre.inorm32;
    const int64_t NW = (int64_t)gridDim.x * 4, gw = (int64_t)blockIdx.x * 4 + wu;
    const int64_t tail0 = r0 + (int64_t)rounds * NW * crows;
    // chunk t of this wave: crows rows (64; 32 or 16 when a short scan must still give every wave three chunks -- the scorer's
    // bound is learned from the chunks before the last, launch_scan) for t < rounds, then its share of the remainder
#define AS_CHUNK(t, base, cnt)                                                           \
    do {                                                                                 \
        if ((t) < rounds) {                                                              \
            base = r0 + ((int64_t)(t) * NW + gw) * crows;                                 \
            cnt = crows;                                                                 \
        } else if ((t) == rounds) {                                                      \
            base = tail0 + gw * tail_rows;                                               \
            const int64_t left_ = r1 - base;                                             \
            cnt = (int)(left_ < 0 ? 0 : (left_ < tail_rows ? left_ : tail_rows));        \
        } else {                                                                         \
            base = r1;                                                                   \
            cnt = 0;                                                                     \
        }                                                                                \
    } while (0)
    // prefetch cursor
    int pt = 0, pr = 0, pcnt = 0, pslot = 0, inflight = 0;
    int64_t pbase = 0;
    AS_CHUNK(0, pbase, pcnt);
#define AS_ISSUE_ROW()                                                                                                  \
    do {                                                                                                                \
        if (pcnt > 0) {                                                                                                 \
            const float* src_ = x32 + (size_t)(pbase + pr) * dp + 4 * lane;                                             \
            _Pragma("unroll") for (int u = 0; u < NCH; ++u) {                                                           \
                if (on[u])                                                                                              \
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_ + 256 * u),   \
                                                     (__attribute__((address_space(3))) void*)(myp + (pslot * NCH + u) * 1024), 16, 0, 2); \
            }                                                                                                           \
            pslot = pslot + 1 == NSLOT ? 0 : pslot + 1;                                                                 \
            ++inflight;                                                                                                 \
            if (++pr == pcnt) {                                                                                         \
                pr = 0;                                                                                                 \
                ++pt;                                                                                                   \
                AS_CHUNK(pt, pbase, pcnt);                                                                              \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i) AS_ISSUE_ROW();
    unsigned cur = 0;    // byte offset of the oldest row in the ring
    int marked = 0;      // rows in flight that have a chunk boundary's store + norm DMA behind them in the queue
    bool first = true;
    ScanWave w;
    for (int t = 0; t <= rounds; ++t) {
        int64_t base;
        int cnt;
        AS_CHUNK(t, base, cnt);
        if (cnt <= 0) continue;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(auxv + base + lane),   // padded: readable
                                         (__attribute__((address_space(3))) void*)(myp + RING), 4, 0, 0);
        if (I8)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pre.fa8 + base + lane),
                                             (__attribute__((address_space(3))) void*)(myp + FA_OFF), 4, 0, 0);
        // SC: the histogram as the other waves have left it (sc1: past this XCD's L2), consumed at the chunk's end -- from
        // the wave's third chunk on: a read issued at the start of the second chunk arrives right behind the burst of
        // publications that ends every wave's first chunk, and queues behind it (in order in front of the row DMAs)
        // ... and then ever more rarely (chunks 2, 3, 5, 9, 17, ...): the read comes back later than a row's DMA and every
        // operation behind it retires behind it -- six reads per wave cost the 1M x 768 scan 8 us, and the bound hardly
        // moves once the first quarter of the rows has been seen
        // (... and at every chunk for as long as the wave has no bound at all: a scan that shares the chip with another kernel has only
        // part of its waves resident at first, and the waves that run ahead found an empty histogram at their third chunk, kept
        // every row and reported an overflow -- "candidates did not fit" on every workspace of a 4-thread run)
        const bool hread = SC && t >= 2 && (t == 2 || ((t - 1) & (t - 2)) == 0 || w.jb_last < 0) && !AS_SC_DBG(2);
        if (hread)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pre.sc_hist + (gw & (SC_COPIES - 1)) * SC_HSTRIDE + lane),
                                             (__attribute__((address_space(3))) void*)(myp + RING + 256), 4, 0, 16);
        marked = first ? 0 : inflight;   // the first chunk has only the norm DMA behind its rows: assume nothing
        const bool hmark = hread;        // (the chunk boundary behind the marked rows holds the histogram DMA as well)
        first = false;
        float mydot = 0.0f;
        for (int r = 0; r < cnt; ++r) {
            // operations retire in issue order: the oldest row has landed once at most (rows behind it) * NCH
            // (+ 2 for a chunk boundary behind it) operations are outstanding
            if (inflight == NSLOT - 1) {
                if (marked > 0 && hmark) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K1 + KB + 1) : "memory");
                else if (marked > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K1 + KB) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K1) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            marked = marked > 0 ? marked - 1 : 0;
            --inflight;
            AS_ISSUE_ROW();   // into the slot consumed one row ago
            f32x4 xv[NCH];
            const unsigned a0 = my0 + cur + lane * 16;
            if (NCH == 1) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(xv[0]) : "v"(a0) : "memory");
            if (NCH == 2)
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(xv[0]), "=&v"(xv[NCH > 1 ? 1 : 0]) : "v"(a0) : "memory");
            if (NCH == 3)
                asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(xv[0]), "=&v"(xv[NCH > 1 ? 1 : 0]), "=&v"(xv[NCH > 2 ? 2 : 0]) : "v"(a0) : "memory");
            if (NCH >= 4)
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(xv[0]), "=&v"(xv[NCH > 1 ? 1 : 0]), "=&v"(xv[NCH > 2 ? 2 : 0]), "=&v"(xv[NCH > 3 ? 3 : 0]) : "v"(a0) : "memory");
            // (rows of 1 025 .. 2 048 image floats -- 2 049 .. 4 096 columns of the int8 image: chunks 4 .. 7)
            if (NCH == 5) asm volatile("ds_read_b128 %0, %1 offset:4096\n\ts_waitcnt lgkmcnt(0)" : "=&v"(xv[NCH > 4 ? 4 : 0]) : "v"(a0) : "memory");
            if (NCH == 6)
                asm volatile("ds_read_b128 %0, %2 offset:4096\n\tds_read_b128 %1, %2 offset:5120\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(xv[NCH > 4 ? 4 : 0]), "=&v"(xv[NCH > 5 ? 5 : 0]) : "v"(a0) : "memory");
            if (NCH == 7)
                asm volatile("ds_read_b128 %0, %3 offset:4096\n\tds_read_b128 %1, %3 offset:5120\n\tds_read_b128 %2, %3 offset:6144\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(xv[NCH > 4 ? 4 : 0]), "=&v"(xv[NCH > 5 ? 5 : 0]), "=&v"(xv[NCH > 6 ? 6 : 0]) : "v"(a0) : "memory");
            if (NCH == 8)
                asm volatile("ds_read_b128 %0, %4 offset:4096\n\tds_read_b128 %1, %4 offset:5120\n\tds_read_b128 %2, %4 offset:6144\n\tds_read_b128 %3, %4 offset:7168\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(xv[NCH > 4 ? 4 : 0]), "=&v"(xv[NCH > 5 ? 5 : 0]), "=&v"(xv[NCH > 6 ? 6 : 0]), "=&v"(xv[NCH > 7 ? 7 : 0]) : "v"(a0) : "memory");
            cur = cur + NCH * 1024 == RING ? 0 : cur + NCH * 1024;
            float sacc = 0.0f;
            if (I8) {
                int hi = 0, xs = 0;
#pragma unroll
                for (int u = 0; u < NCH; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi = __builtin_amdgcn_sdot4(__float_as_int(xv[u][e]), qa[u][e], hi, false);
                        xs = __builtin_amdgcn_sdot4(__float_as_int(xv[u][e]), qb[u][e], xs, false);
                    }
                // (the lanes' partial sums stay below 2^24; the wave's totals too up to 1 040 columns -- 127^2 D --, beyond that each of the
                // six additions of a wave sum may round: host_query_digits charges 12 more roundings for rows wider than 1 024 columns)
                const float fh = wave_sum_dpp((float)hi), fx = wave_sum_dpp((float)xs);
                sacc = fmaf(fh, 128.0f, fx);
            } else {
#pragma unroll
                for (int u = 0; u < NCH; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sacc = fmaf(xv[u][e], qv[u][e], sacc);
                sacc = wave_sum_dpp(sacc);
            }
            mydot = lane == r ? sacc : mydot;
        }
        // the norms are older than every row issued inside this chunk; one of those has been consumed once the
        // chunk is at least as long as the ring
        if (cnt < NSLOT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float aux = lds_read1(ax0 + lane * 4);
        if (I8) mydot *= lds_read1(my0 + FA_OFF + lane * 4) * pre.faq;   // x_i . q = (128 HI + XS) fa_i fa_q
        scan_chunk_end<SC>(pre, w, dots, t, rounds, gw, lane, base, cnt, mydot, aux, hread, hx0, px0, nq32, inq32);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    scan_wave_report<SC>(pre, w, gw, lane, px0);
#undef AS_ISSUE_ROW
#undef AS_CHUNK
}

// Tile scan -- the single query's COARSE scan.  The items' high digits lie in TILES: per 64 consecutive rows and
// 16-column chunk c ONE KiB -- row r's sixteen digits at byte 16 r ([tile][chunk][64 rows][16 B]; space_i8h_image).  A wave takes
// a chunk of up to 64 consecutive rows as before, but one LDS-DMA now brings chunk c of ALL its rows: lane r receives row r's
// digits, multiplies them with the query's digits of those 16 columns -- the same for every lane: read from LDS by broadcast --
// and keeps row r's two integer sums in its own registers for the whole chunk.  No cross-lane reduction (the row-ring kernel
// pays two 64-lane DPP sums and a readlane per 768-byte row, with 48 of 64 lanes busy), every DMA a full KiB whatever the row
// width, a quarter of the instructions per byte: per KiB one counted wait, one DMA, three LDS reads, eight v_dot4_i32_i8.
// Chunks need not be aligned with tiles (the remainder's even split, short shards' chunks of 32 / 16 rows): the lanes' source
// addresses are computed per row.  The ring holds NSLOT KiB per wave, NSLOT - 2 in flight; items are issued and consumed in
// pairs (the row width in 16-column chunks is a multiple of 4).  Chunk schedule, chunk end (dots, k-NN prefilter, scorer
// candidates) and the wave's report are those of scan_dma_kernel.
template <int NSLOT, bool SC, int U = 2>
__global__ __launch_bounds__(256) void scan_tile_kernel(const signed char* __restrict__ xt, int C, int64_t r0, int64_t r1, float* __restrict__ dots,
                                                        PreArgs pre, int rounds, int tail_rows, int crows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // U: items (KiB) per step
    constexpr int RING = NSLOT * 1024;            // bytes per wave
    constexpr int WAVE_LDS = RING + 256 + 256 + (SC ? 256 + TILE_PEND * 8 : 0);   // + the chunk's 64 norms, 64 scales (+ histogram, pending list)
    constexpr int K1 = NSLOT - 2 * U;             // DMA operations younger than the oldest pair of a full ring
    constexpr int KB = 3;                         // operations of a chunk boundary: the dots' store, the norm DMA, the scale DMA
    static_assert((U == 2 || U == 4) && NSLOT % U == 0 && NSLOT >= 2 * U, "ring of whole steps");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* myp = smem + wu * WAVE_LDS;
    const unsigned s0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned my0 = s0 + wu * WAVE_LDS;
    const unsigned ax0 = my0 + RING, fx0 = ax0 + 256, hx0 = fx0 + 256, px0 = hx0 + 256;
    const unsigned qx0 = s0 + 4 * WAVE_LDS;       // the query's digits, shared by the block: per chunk 16 bytes q1, 16 bytes q2
    const float* __restrict__ auxv = pre.metric == AS_METRIC_L2 ? pre.n32 : pre.inorm32;
    const int64_t NW = (int64_t)gridDim.x * 4, gw = (int64_t)blockIdx.x * 4 + wu;
    const int64_t tail0 = r0 + (int64_t)rounds * NW * crows;
#ifdef AS_STAMPS
    if (lane == 0 && gw < 4096) g_scan_stamps[gw] = __builtin_amdgcn_s_memrealtime();
#endif
#define AS_CHUNK(t, base, cnt)                                                           \
    do {                                                                                 \
        if ((t) < rounds) {                                                              \
            base = r0 + ((int64_t)(t) * NW + gw) * crows;                                 \
            cnt = crows;                                                                 \
        } else if ((t) == rounds) {                                                      \
            base = tail0 + gw * tail_rows;                                               \
            const int64_t left_ = r1 - base;                                             \
            cnt = (int)(left_ < 0 ? 0 : (left_ < tail_rows ? left_ : tail_rows));        \
        } else {                                                                         \
            base = r1;                                                                   \
            cnt = 0;                                                                     \
        }                                                                                \
    } while (0)
    // prefetch cursor: chunk pt, column chunk pcol; the lanes' byte offsets from the chunk's first tile
    int pt = 0, pcol = 0, pcnt = 0, pslot = 0, inflight = 0;
    int64_t pbase = 0;
    unsigned pvoff = 0;
    const signed char* pub = xt;
    const size_t tile_bytes = (size_t)C * 1024;
#define AS_TILE_ENTER()                                                                                                  \
    do {                                                                                                                 \
        if (pcnt > 0) {                                                                                                  \
            const int rl_ = lane < pcnt ? lane : pcnt - 1;   /* lanes past the chunk's end fetch its last row again */     \
            const int tl_ = (int)(pbase & 63) + rl_;                                                                     \
            pvoff = (unsigned)(tl_ >> 6) * (unsigned)tile_bytes + (unsigned)(tl_ & 63) * 16u;                            \
            pub = xt + (size_t)(pbase >> 6) * tile_bytes;                                                                \
        }                                                                                                                \
    } while (0)
#define AS_TILE_ISSUE()                                                                                                  \
    do {                                                                                                                 \
        if (pcnt > 0) {                                                                                                  \
            _Pragma("unroll") for (int u_ = 0; u_ < U; ++u_)                                                             \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pub + pvoff + u_ * 1024), \
                                                 (__attribute__((address_space(3))) void*)(myp + (pslot + u_) * 1024), 16, 0, 2); \
            pub += U * 1024;                                                                                             \
            pslot = pslot + U == NSLOT ? 0 : pslot + U;                                                                  \
            inflight += U;                                                                                               \
            pcol += U;                                                                                                   \
            if (pcol == C) {                                                                                             \
                pcol = 0;                                                                                                \
                ++pt;                                                                                                    \
                AS_CHUNK(pt, pbase, pcnt);                                                                               \
                AS_TILE_ENTER();                                                                                         \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)
    AS_CHUNK(0, pbase, pcnt);
    AS_TILE_ENTER();
#pragma unroll
    for (int i = 0; i < NSLOT / U - 1; ++i) AS_TILE_ISSUE();
    // (behind the ring's first fill: the query's digits come from the host's pinned memory -- a PCIe round trip every block
    // would otherwise take before its first DMA)
    {
        int* qd = (int*)(smem + 4 * WAVE_LDS);
        for (int i = tid; i < C * 8; i += 256) qd[i] = pre.q8[i];
    }
    float nq32 = pre.host_q ? pre.nq32 : pre.info->nq32, inq32 = pre.host_q ? pre.inq32 : pre.info->inq32;
    if (pre.host_q && blockIdx.x == 0 && tid == 0) {   // what q_prepare would have filed: read by the kernels behind the scan
        pre.infow->nq = pre.nq;
        pre.infow->inq = pre.inq;
        pre.infow->nq32 = pre.nq32;
        pre.infow->inq32 = pre.inq32;
        pre.infow->tau = 1.0;
    }
    if (pre.host_q && pre.q64_dev)   // the fp64 query for the kernels behind the scan: pinned host memory -> device (PreArgs)
        for (int g = (int)blockIdx.x * 256 + tid; g < pre.qdp; g += (int)gridDim.x * 256) pre.q64_dev[g] = pre.q64_host[g];
    asm volatile("" : "+v"(nq32), "+v"(inq32));
    __syncthreads();   // (the only block barrier: the query's digits are in LDS; from here on every LDS access is inline asm)
    unsigned cur = 0;    // byte offset of the oldest pair in the ring
    int marked = 0;      // items in flight that have a chunk boundary's store + norm + scale DMA behind them in the queue
    bool first = true;
    ScanWave w;
    for (int t = 0; t <= rounds; ++t) {
        int64_t base;
        int cnt;
        AS_CHUNK(t, base, cnt);
        if (cnt <= 0) continue;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(auxv + base + lane),   // padded: readable
                                         (__attribute__((address_space(3))) void*)(myp + RING), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pre.fa8 + base + lane),
                                         (__attribute__((address_space(3))) void*)(myp + RING + 256), 4, 0, 0);
        // SC: the histogram as the other waves have left it, consumed at the chunk's end (chunks 2, 3, 5, 9, 17, ...: scan_dma_kernel)
        const bool hread = SC && t >= 2 && (t == 2 || ((t - 1) & (t - 2)) == 0 || w.jb_last < 0) && !AS_SC_DBG(2);
        if (hread)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pre.sc_hist + (gw & (SC_COPIES - 1)) * SC_HSTRIDE + lane),
                                             (__attribute__((address_space(3))) void*)(myp + RING + 512), 4, 0, 16);
        marked = first ? 0 : inflight;   // the first chunk has no store in front of its boundary DMAs: assume nothing
        const bool hmark = hread;
        first = false;
        int hi0 = 0, xs0 = 0, hi1 = 0, xs1 = 0;
        unsigned qa = qx0;
        for (int c = 0; c < C; c += U) {
            // operations retire in issue order: the oldest pair has landed once at most NSLOT - 2 U younger items (+ a chunk
            // boundary's operations behind it) are outstanding
            if (inflight == NSLOT - U) {
                if (marked > 0 && hmark) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K1 + KB + 1) : "memory");
                else if (marked > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K1 + KB) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K1) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            marked = marked > U ? marked - U : 0;
            inflight -= U;
            AS_TILE_ISSUE();   // into the slots consumed one step ago
            i32x4s xv[U], qa_[U], qb_[U];
            const unsigned a0 = my0 + cur + lane * 16;
            if (U == 2)
                asm volatile(
                    "ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:1024\n\tds_read_b128 %2, %7\n\tds_read_b128 %3, %7 offset:16\n\t"
                    "ds_read_b128 %4, %7 offset:32\n\tds_read_b128 %5, %7 offset:48\n\ts_waitcnt lgkmcnt(0)"
                    : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(qa_[0]), "=&v"(qb_[0]), "=&v"(qa_[1]), "=&v"(qb_[1])
                    : "v"(a0), "v"(qa)
                    : "memory");
            if (U == 4)
                asm volatile(
                    "ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:1024\n\tds_read_b128 %2, %12 offset:2048\n\tds_read_b128 %3, %12 offset:3072\n\t"
                    "ds_read_b128 %4, %13\n\tds_read_b128 %5, %13 offset:16\n\tds_read_b128 %6, %13 offset:32\n\tds_read_b128 %7, %13 offset:48\n\t"
                    "ds_read_b128 %8, %13 offset:64\n\tds_read_b128 %9, %13 offset:80\n\tds_read_b128 %10, %13 offset:96\n\tds_read_b128 %11, %13 offset:112\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[U > 2 ? 2 : 0]), "=&v"(xv[U > 3 ? 3 : 0]), "=&v"(qa_[0]), "=&v"(qb_[0]), "=&v"(qa_[1]), "=&v"(qb_[1]),
                      "=&v"(qa_[U > 2 ? 2 : 0]), "=&v"(qb_[U > 2 ? 2 : 0]), "=&v"(qa_[U > 3 ? 3 : 0]), "=&v"(qb_[U > 3 ? 3 : 0])
                    : "v"(a0), "v"(qa)
                    : "memory");
            cur = cur + U * 1024 == RING ? 0 : cur + U * 1024;
            qa += U * 32;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (u & 1) {
                        hi1 = __builtin_amdgcn_sdot4(xv[u][e], qa_[u][e], hi1, false);
                        xs1 = __builtin_amdgcn_sdot4(xv[u][e], qb_[u][e], xs1, false);
                    } else {
                        hi0 = __builtin_amdgcn_sdot4(xv[u][e], qa_[u][e], hi0, false);
                        xs0 = __builtin_amdgcn_sdot4(xv[u][e], qb_[u][e], xs0, false);
                    }
                }
        }
        // the boundary DMAs are older than every item issued inside this chunk; one of those has been consumed once the chunk
        // has more items than the ring keeps in flight
        if (C <= NSLOT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float aux = lds_read1(ax0 + lane * 4);
        // x_i . q = (128 HI + XS) fa_i fa_q: the integer is exact, its conversion the one rounding the row-ring kernel's fused
        // multiply-add makes
        const long long tot = (long long)(hi0 + hi1) * 128 + (long long)(xs0 + xs1);
        const float mydot = (float)tot * (lds_read1(fx0 + lane * 4) * pre.faq);
        scan_chunk_end<SC, TILE_PEND>(pre, w, dots, t, rounds, gw, lane, base, cnt, mydot, aux, hread, hx0, px0, nq32, inq32);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    scan_wave_report<SC>(pre, w, gw, lane, px0);
#ifdef AS_STAMPS
    if (lane == 0 && gw < 4096) g_scan_stamps[4096 + gw] = __builtin_amdgcn_s_memrealtime();
#endif
#undef AS_TILE_ISSUE
#undef AS_TILE_ENTER
#undef AS_CHUNK
}

// The same scan with a DYNAMIC chunk schedule.  Waves do not run at one speed: with equal shares the median wave of the 1M x 768
// scan ended 119 us after the launch, the last one at 131.5 (make STAMPS=1, tools/scan_stamps.py) -- a tail of 12 us at falling
// occupancy.  Here chunk c is rows [r0 + c crows, + crows); a wave's first chunk is its own (c = wave), every further one is
// NW + ng j + g for ticket j of the wave's group g (ng = 16 atomic cursors on a full grid, zero between searches).  The ticket for the chunk after
// next is drawn
// at a chunk's START by a returning atomic issued through inline asm: it is older than every item issued inside that chunk, so
// the chunk's counted waits have seen it retire long before its value is read at the chunk's end -- no wait of its own, and
// nothing the compiler would drain the ring for.  (Rows of fewer than 16 chunks keep the static kernel: the prefetch cursor
// must not cross two chunk boundaries inside one chunk.)
// Measured (1M x 768, tools/tile_geom.py, profiles/r05_tile_dyn.txt): ONE cursor for all 2 048 waves ran at the rate of one
// word's atomics -- 252 us against the static schedule's 138; sixteen cursors 130.7 us against 135.5, the last wave ending at
// 129.5 us instead of 132.5, the waves' ends 115.5 .. 126 (p10 .. p90) instead of 111 .. 128.  Chunks of 32 rows (half the lanes
// of every DMA idle) 133.6 us, of 16 rows 193: the chunk stays 64 rows, a tile.  Half chunks for the waves' LAST tickets only (the
// last 1/16 .. 1/4 of the rows): 134-136 us against 132-133 on the same box -- no gain, not kept.
template <int NSLOT, bool SC>
__global__ __launch_bounds__(256) void scan_tile_kernel_dyn(const signed char* __restrict__ xt, int C, int64_t r0, int64_t r1, float* __restrict__ dots,
                                                            PreArgs pre, int crows, int ng) {
    constexpr int U = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RING = NSLOT * 1024;            // bytes per wave
    constexpr int WAVE_LDS = RING + 256 + 256 + (SC ? 256 + TILE_PEND * 8 : 0);   // + the chunk's 64 norms, 64 scales (+ histogram, pending list)
    constexpr int K1 = NSLOT - 2 * U;             // DMA operations younger than the oldest pair of a full ring
    constexpr int KB = 4;                         // operations of a chunk boundary: the dots' store, the norm DMA, the scale DMA, the ticket
    static_assert((U == 2 || U == 4) && NSLOT % U == 0 && NSLOT >= 2 * U, "ring of whole steps");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* myp = smem + wu * WAVE_LDS;
    const unsigned s0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned my0 = s0 + wu * WAVE_LDS;
    const unsigned ax0 = my0 + RING, fx0 = ax0 + 256, hx0 = fx0 + 256, px0 = hx0 + 256;
    const unsigned qx0 = s0 + 4 * WAVE_LDS;       // the query's digits, shared by the block: per chunk 16 bytes q1, 16 bytes q2
    const float* __restrict__ auxv = pre.metric == AS_METRIC_L2 ? pre.n32 : pre.inorm32;
    const int64_t NW = (int64_t)gridDim.x * 4, gw = (int64_t)blockIdx.x * 4 + wu;
#ifdef AS_STAMPS
    if (lane == 0 && gw < 4096) g_scan_stamps[gw] = __builtin_amdgcn_s_memrealtime();
#endif
#define AS_CHUNK_ID(cid, base, cnt)                                                      \
    do {                                                                                 \
        base = r0 + (int64_t)(cid) * crows;                                              \
        const int64_t left_ = r1 - base;                                                 \
        cnt = (int)(left_ <= 0 ? 0 : (left_ < crows ? left_ : crows));                   \
    } while (0)
    // SC_COPIES cursors, each in the unused tail of a histogram copy's stride (different memory channels; zeroed with the
    // histograms): 2 048 waves x 8 tickets on ONE word ran at the rate of that word's atomics -- 250 us.  A block's group is
    // (b + b / 8) mod 16: a group's blocks cycle through the XCDs (consecutive blocks go to consecutive XCDs), so a slow XCD
    // slows every group alike; group g hands out the chunks NW + ng j + g, j = 0, 1, ...
    // (ng groups, a power of two up to SC_COPIES that the host picks so that EVERY group has a block: a grid of fewer than 32
    // blocks -- a shard of a few thousand rows -- with sixteen groups left some of them, and their chunks, without a wave)
    const int grp = (int)((blockIdx.x + (blockIdx.x >> 3)) & (unsigned)(ng - 1));
    int* ctr = (int*)(pre.tile_ctrs + grp * SC_HSTRIDE + SC_CTR_WORD);
    int tk = 0;                // the ticket in flight (written by the atomic: read only behind the waits described above)
    const int one = 1;
    // (one atomic per WAVE: lane 0 draws, v_readfirstlane reads lane 0's register)
#define AS_TICKET()                                                                                                   \
    do {                                                                                                              \
        if (lane == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "+v"(tk) : "v"(ctr), "v"(one) : "memory"); \
    } while (0)
    AS_TICKET();               // the wave's second chunk
    int64_t ccur = gw, cnxt = 0;   // chunk being consumed; the one the prefetch cursor crosses into
    // prefetch cursor: chunk pt, column chunk pcol; the lanes' byte offsets from the chunk's first tile
    int pcol = 0, pcnt = 0, pslot = 0, inflight = 0;
    int64_t pbase = 0;
    unsigned pvoff = 0;
    const signed char* pub = xt;
    const size_t tile_bytes = (size_t)C * 1024;
#define AS_TILE_ENTER()                                                                                                  \
    do {                                                                                                                 \
        if (pcnt > 0) {                                                                                                  \
            const int rl_ = lane < pcnt ? lane : pcnt - 1;   /* lanes past the chunk's end fetch its last row again */     \
            const int tl_ = (int)(pbase & 63) + rl_;                                                                     \
            pvoff = (unsigned)(tl_ >> 6) * (unsigned)tile_bytes + (unsigned)(tl_ & 63) * 16u;                            \
            pub = xt + (size_t)(pbase >> 6) * tile_bytes;                                                                \
        }                                                                                                                \
    } while (0)
#define AS_TILE_ISSUE()                                                                                                  \
    do {                                                                                                                 \
        if (pcnt > 0) {                                                                                                  \
            _Pragma("unroll") for (int u_ = 0; u_ < U; ++u_)                                                             \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pub + pvoff + u_ * 1024), \
                                                 (__attribute__((address_space(3))) void*)(myp + (pslot + u_) * 1024), 16, 0, 2); \
            pub += U * 1024;                                                                                             \
            pslot = pslot + U == NSLOT ? 0 : pslot + U;                                                                  \
            inflight += U;                                                                                               \
            pcol += U;                                                                                                   \
            if (pcol == C) {                                                                                             \
                pcol = 0;                                                                                                \
                AS_CHUNK_ID(cnxt, pbase, pcnt);                                                                          \
                AS_TILE_ENTER();                                                                                         \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)
    AS_CHUNK_ID(ccur, pbase, pcnt);
    AS_TILE_ENTER();
#pragma unroll
    for (int i = 0; i < NSLOT / U - 1; ++i) AS_TILE_ISSUE();
    // (behind the ring's first fill: the query's digits come from the host's pinned memory -- a PCIe round trip every block
    // would otherwise take before its first DMA)
    {
        int* qd = (int*)(smem + 4 * WAVE_LDS);
        for (int i = tid; i < C * 8; i += 256) qd[i] = pre.q8[i];
    }
    float nq32 = pre.host_q ? pre.nq32 : pre.info->nq32, inq32 = pre.host_q ? pre.inq32 : pre.info->inq32;
    if (pre.host_q && blockIdx.x == 0 && tid == 0) {   // what q_prepare would have filed: read by the kernels behind the scan
        pre.infow->nq = pre.nq;
        pre.infow->inq = pre.inq;
        pre.infow->nq32 = pre.nq32;
        pre.infow->inq32 = pre.inq32;
        pre.infow->tau = 1.0;
    }
    if (pre.host_q && pre.q64_dev)   // the fp64 query for the kernels behind the scan: pinned host memory -> device (PreArgs)
        for (int g = (int)blockIdx.x * 256 + tid; g < pre.qdp; g += (int)gridDim.x * 256) pre.q64_dev[g] = pre.q64_host[g];
    asm volatile("" : "+v"(nq32), "+v"(inq32));
    __syncthreads();   // (the only block barrier: the query's digits are in LDS; from here on every LDS access is inline asm)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(tk) : : "memory");   // (start-up: the ring's first fill and the first ticket)
    cnxt = NW + (int64_t)__builtin_amdgcn_readfirstlane(tk) * ng + grp;
    unsigned cur = 0;    // byte offset of the oldest pair in the ring
    int marked = 0;      // items in flight that have a chunk boundary's store + norm + scale DMA behind them in the queue
    bool first = true;
    ScanWave w;
    for (int t = 0;; ++t) {
        int64_t base;
        int cnt;
        AS_CHUNK_ID(ccur, base, cnt);
        if (cnt <= 0) break;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(auxv + base + lane),   // padded: readable
                                         (__attribute__((address_space(3))) void*)(myp + RING), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pre.fa8 + base + lane),
                                         (__attribute__((address_space(3))) void*)(myp + RING + 256), 4, 0, 0);
        // SC: the histogram as the other waves have left it, consumed at the chunk's end (chunks 2, 3, 5, 9, 17, ...: scan_dma_kernel)
        const bool hread = SC && t >= 2 && (t == 2 || ((t - 1) & (t - 2)) == 0 || w.jb_last < 0) && !AS_SC_DBG(2);
        if (hread)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pre.sc_hist + (gw & (SC_COPIES - 1)) * SC_HSTRIDE + lane),
                                             (__attribute__((address_space(3))) void*)(myp + RING + 512), 4, 0, 16);
        AS_TICKET();                     // the chunk after `cnxt`
        marked = first ? 0 : inflight;   // the first chunk has no store in front of its boundary DMAs: assume nothing
        const bool hmark = hread;
        first = false;
        int hi0 = 0, xs0 = 0, hi1 = 0, xs1 = 0;
        unsigned qa = qx0;
        for (int c = 0; c < C; c += U) {
            // operations retire in issue order: the oldest pair has landed once at most NSLOT - 2 U younger items (+ a chunk
            // boundary's operations behind it) are outstanding
            if (inflight == NSLOT - U) {
                if (marked > 0 && hmark) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K1 + KB + 1) : "memory");
                else if (marked > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K1 + KB) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K1) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            marked = marked > U ? marked - U : 0;
            inflight -= U;
            AS_TILE_ISSUE();   // into the slots consumed one step ago
            i32x4s xv[U], qa_[U], qb_[U];
            const unsigned a0 = my0 + cur + lane * 16;
            if (U == 2)
                asm volatile(
                    "ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:1024\n\tds_read_b128 %2, %7\n\tds_read_b128 %3, %7 offset:16\n\t"
                    "ds_read_b128 %4, %7 offset:32\n\tds_read_b128 %5, %7 offset:48\n\ts_waitcnt lgkmcnt(0)"
                    : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(qa_[0]), "=&v"(qb_[0]), "=&v"(qa_[1]), "=&v"(qb_[1])
                    : "v"(a0), "v"(qa)
                    : "memory");
            if (U == 4)
                asm volatile(
                    "ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:1024\n\tds_read_b128 %2, %12 offset:2048\n\tds_read_b128 %3, %12 offset:3072\n\t"
                    "ds_read_b128 %4, %13\n\tds_read_b128 %5, %13 offset:16\n\tds_read_b128 %6, %13 offset:32\n\tds_read_b128 %7, %13 offset:48\n\t"
                    "ds_read_b128 %8, %13 offset:64\n\tds_read_b128 %9, %13 offset:80\n\tds_read_b128 %10, %13 offset:96\n\tds_read_b128 %11, %13 offset:112\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[U > 2 ? 2 : 0]), "=&v"(xv[U > 3 ? 3 : 0]), "=&v"(qa_[0]), "=&v"(qb_[0]), "=&v"(qa_[1]), "=&v"(qb_[1]),
                      "=&v"(qa_[U > 2 ? 2 : 0]), "=&v"(qb_[U > 2 ? 2 : 0]), "=&v"(qa_[U > 3 ? 3 : 0]), "=&v"(qb_[U > 3 ? 3 : 0])
                    : "v"(a0), "v"(qa)
                    : "memory");
            cur = cur + U * 1024 == RING ? 0 : cur + U * 1024;
            qa += U * 32;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (u & 1) {
                        hi1 = __builtin_amdgcn_sdot4(xv[u][e], qa_[u][e], hi1, false);
                        xs1 = __builtin_amdgcn_sdot4(xv[u][e], qb_[u][e], xs1, false);
                    } else {
                        hi0 = __builtin_amdgcn_sdot4(xv[u][e], qa_[u][e], hi0, false);
                        xs0 = __builtin_amdgcn_sdot4(xv[u][e], qb_[u][e], xs0, false);
                    }
                }
        }
        // the boundary DMAs are older than every item issued inside this chunk; one of those has been consumed once the chunk
        // has more items than the ring keeps in flight
        if (C <= NSLOT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float aux = lds_read1(ax0 + lane * 4);
        // x_i . q = (128 HI + XS) fa_i fa_q: the integer is exact, its conversion the one rounding the row-ring kernel's fused
        // multiply-add makes
        const long long tot = (long long)(hi0 + hi1) * 128 + (long long)(xs0 + xs1);
        const float mydot = (float)tot * (lds_read1(fx0 + lane * 4) * pre.faq);
        scan_chunk_end<SC, TILE_PEND>(pre, w, dots, t, 1 << 30, gw, lane, base, cnt, mydot, aux, hread, hx0, px0, nq32, inq32);
        // (the ticket drawn at this chunk's start has retired: C >= 16 items were issued behind it and all but the ring's youngest waited for)
        asm volatile("" : "+v"(tk));
        ccur = cnxt;
        cnxt = NW + (int64_t)__builtin_amdgcn_readfirstlane(tk) * ng + grp;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    scan_wave_report<SC>(pre, w, gw, lane, px0);
#ifdef AS_STAMPS
    if (lane == 0 && gw < 4096) g_scan_stamps[4096 + gw] = __builtin_amdgcn_s_memrealtime();
#endif
#undef AS_TILE_ISSUE
#undef AS_TILE_ENTER
#undef AS_CHUNK_ID
#undef AS_TICKET
}

// Gang scan: ONE pass over the tiles for up to FOUR single queries of concurrent host threads (as_search is re-entrant: callers
// that arrive together share the read of the image instead of competing for HBM with a scan each; as_search.hip, gang_join).
// The products move to the matrix pipe, which the single-query kernel leaves idle: one v_mfma_i32_16x16x64_i8 per KiB and digit.
// A DMA'd KiB is chunk c of 64 rows, lane l holding row l -- as the MFMA's A operand that reads "rows 0..15, four k-chunks", the
// k-chunk g = l / 16 being row group g.  The B operand's 16 columns are SLOTS (query qq, row group g') = 4 qq + g': slot (qq, g')
// carries query qq's digits of chunk c in k-chunk g' and zeros elsewhere, so D[r][4 qq + g'] = row (16 g' + r) . q_qq -- 64 rows x 4
// queries per instruction.  A lane's B fragment is either its query's 16 digits or zeros: lanes of the second kind read a zero
// line of LDS (their address does not move), no select in the loop.  Per KiB: one counted wait, one DMA, three LDS reads, two
// MFMAs -- whatever the number of queries.  At the chunk's end the accumulators (result register i of lane l: row 16 (l % 4) +
// 4 (l / 16) + i of query (l % 16) / 4) pass through LDS into "lane r holds row r", query by query, and every query runs the
// chunk end of the single-query kernels on its own buffers.  Bit-identical dots: the integer sums are exact.
struct GangArgs {
    PreArgs p[4];
    float* dots[4];
};
typedef int i32x4g __attribute__((ext_vector_type(4)));
// DYN: the chunks by tickets as in scan_tile_kernel_dyn (the cursors are the first member's: ga.p[0].tile_ctrs, `ng` groups) --
// a gang's scan shares the chip with its members' previous tail kernels more often than not, and equal shares cannot adapt to
// CUs that are busy with something else.
template <int NSLOT, int NQ, bool DYN = false>
__global__ __launch_bounds__(256) void scan_tile_gang_kernel(const signed char* __restrict__ xt, int C, int64_t r0, int64_t r1, GangArgs ga, int rounds,
                                                             int tail_rows, int crows, int ng) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int U = 2;
    constexpr int RING = NSLOT * 1024;
    constexpr int QAREA = 256 + GANG_PEND * 8;                     // per query and wave: histogram landing area, pending list
    constexpr int TR = NQ * 64 * 4 * 2;                            // per wave: the accumulators on their way to "lane r = row r" (HI and XS)
    constexpr int WAVE_LDS = RING + 512 + NQ * QAREA + TR;
    constexpr int K1 = NSLOT - 2 * U;
    constexpr int KB = NQ + 2 + (DYN ? 1 : 0);                     // a chunk boundary: NQ dot stores, the norm DMA, the scale DMA (DYN: the ticket)
    static_assert(NSLOT % U == 0 && NSLOT >= 2 * U && NQ >= 2 && NQ <= 4, "ring of whole steps, two to four queries");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* myp = smem + wu * WAVE_LDS;
    const unsigned s0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned my0 = s0 + wu * WAVE_LDS;
    const unsigned ax0 = my0 + RING, fx0 = ax0 + 256, qa0 = fx0 + 256, tr0 = qa0 + NQ * QAREA;
    // block: the queries' digits (per query C x 32 bytes: 16 of q1, 16 of q2 per chunk), then one zero line of 32 bytes
    const int QS = C * 32;
    const unsigned qx0 = s0 + 4 * WAVE_LDS, zx0 = qx0 + NQ * QS;
    const signed char* pub = xt;
    int pt = 0, pcol = 0, pcnt = 0, pslot = 0, inflight = 0;
    int64_t pbase = 0;
    unsigned pvoff = 0;
    const size_t tile_bytes = (size_t)C * 1024;
    const int64_t NW = (int64_t)gridDim.x * 4, gw = (int64_t)blockIdx.x * 4 + wu;
    const int64_t tail0 = r0 + (int64_t)rounds * NW * crows;
    // DYN: chunk c = rows [r0 + c crows, + crows); the wave's first chunk is its own, the others NW + ng j + g for ticket j of its group g
    const int grp = (int)((blockIdx.x + (blockIdx.x >> 3)) & (unsigned)(ng - 1));
    int* ctr = DYN ? (int*)(ga.p[0].tile_ctrs + grp * SC_HSTRIDE + SC_CTR_WORD) : nullptr;
    int tk = 0;
    const int one = 1;
    int64_t ccur = gw, cnxt = 0;
#define AS_CHUNK_ID(cid, base, cnt)                                                      \
    do {                                                                                 \
        base = r0 + (int64_t)(cid) * crows;                                              \
        const int64_t left_ = r1 - base;                                                 \
        cnt = (int)(left_ <= 0 ? 0 : (left_ < crows ? left_ : crows));                   \
    } while (0)
#define AS_TICKET()                                                                                                   \
    do {                                                                                                              \
        if (DYN && lane == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "+v"(tk) : "v"(ctr), "v"(one) : "memory"); \
    } while (0)
    AS_TICKET();
#define AS_CHUNK(t, base, cnt)                                                           \
    do {                                                                                 \
        if ((t) < rounds) {                                                              \
            base = r0 + ((int64_t)(t) * NW + gw) * crows;                                 \
            cnt = crows;                                                                 \
        } else if ((t) == rounds) {                                                      \
            base = tail0 + gw * tail_rows;                                               \
            const int64_t left_ = r1 - base;                                             \
            cnt = (int)(left_ < 0 ? 0 : (left_ < tail_rows ? left_ : tail_rows));        \
        } else {                                                                         \
            base = r1;                                                                   \
            cnt = 0;                                                                     \
        }                                                                                \
    } while (0)
#define AS_TILE_ENTER()                                                                                                  \
    do {                                                                                                                 \
        if (pcnt > 0) {                                                                                                  \
            const int rl_ = lane < pcnt ? lane : pcnt - 1;                                                               \
            const int tl_ = (int)(pbase & 63) + rl_;                                                                     \
            pvoff = (unsigned)(tl_ >> 6) * (unsigned)tile_bytes + (unsigned)(tl_ & 63) * 16u;                            \
            pub = xt + (size_t)(pbase >> 6) * tile_bytes;                                                                \
        }                                                                                                                \
    } while (0)
#define AS_TILE_ISSUE()                                                                                                  \
    do {                                                                                                                 \
        if (pcnt > 0) {                                                                                                  \
            _Pragma("unroll") for (int u_ = 0; u_ < U; ++u_)                                                             \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pub + pvoff + u_ * 1024), \
                                                 (__attribute__((address_space(3))) void*)(myp + (pslot + u_) * 1024), 16, 0, 2); \
            pub += U * 1024;                                                                                             \
            pslot = pslot + U == NSLOT ? 0 : pslot + U;                                                                  \
            inflight += U;                                                                                               \
            pcol += U;                                                                                                   \
            if (pcol == C) {                                                                                             \
                pcol = 0;                                                                                                \
                if (DYN) {                                                                                               \
                    AS_CHUNK_ID(cnxt, pbase, pcnt);                                                                      \
                } else {                                                                                                 \
                    ++pt;                                                                                                \
                    AS_CHUNK(pt, pbase, pcnt);                                                                           \
                }                                                                                                        \
                AS_TILE_ENTER();                                                                                         \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)
    if (DYN) {
        AS_CHUNK_ID(ccur, pbase, pcnt);
    } else {
        AS_CHUNK(0, pbase, pcnt);
    }
    AS_TILE_ENTER();
#pragma unroll
    for (int i = 0; i < NSLOT / U - 1; ++i) AS_TILE_ISSUE();
    {
        int* qd = (int*)(smem + 4 * WAVE_LDS);
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq)
            for (int i = tid; i < C * 8; i += 256) qd[qq * C * 8 + i] = ga.p[qq].q8[i];
        if (tid < 8) qd[NQ * C * 8 + tid] = 0;
    }
    float nq32[NQ], inq32[NQ];
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) {
        const PreArgs& pre = ga.p[qq];
        nq32[qq] = pre.nq32;
        inq32[qq] = pre.inq32;
        if (blockIdx.x == 0 && tid == 0) {   // (every member's query is host-prepared: what q_prepare would have filed)
            pre.infow->nq = pre.nq;
            pre.infow->inq = pre.inq;
            pre.infow->nq32 = pre.nq32;
            pre.infow->inq32 = pre.inq32;
            pre.infow->tau = 1.0;
        }
        if (pre.q64_dev)
            for (int g = (int)blockIdx.x * 256 + tid; g < pre.qdp; g += (int)gridDim.x * 256) pre.q64_dev[g] = pre.q64_host[g];
    }
    __syncthreads();
    if (DYN) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(tk) : : "memory");   // (start-up: the ring's first fill and the first ticket)
        cnxt = NW + (int64_t)__builtin_amdgcn_readfirstlane(tk) * ng + grp;
    }
    const float* __restrict__ auxv = ga.p[0].metric == AS_METRIC_L2 ? ga.p[0].n32 : ga.p[0].inorm32;
    // this lane's B fragment: slot lane % 16 = (query qq, row group g'); its k-chunk is lane / 16: digits where g' == lane / 16
    const int myq = (lane & 15) >> 2;
    const bool bon = ((lane & 3) == (lane >> 4)) && myq < NQ;
    const unsigned qstep = bon ? (unsigned)(U * 32) : 0u;
    unsigned qa = bon ? qx0 + (unsigned)(myq * QS) : zx0;
    const unsigned qa_start = qa;
    unsigned cur = 0;
    int marked = 0;
    bool first = true;
    ScanWave w[NQ];
    for (int t = 0; DYN || t <= rounds; ++t) {
        int64_t base;
        int cnt;
        if (DYN) {
            AS_CHUNK_ID(ccur, base, cnt);
            if (cnt <= 0) break;
        } else {
            AS_CHUNK(t, base, cnt);
            if (cnt <= 0) continue;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(auxv + base + lane),
                                         (__attribute__((address_space(3))) void*)(myp + RING), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga.p[0].fa8 + base + lane),
                                         (__attribute__((address_space(3))) void*)(myp + RING + 256), 4, 0, 0);
        bool nobound = false;   // (every member reads or none: the ring's counts assume NQ reads)
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) nobound = nobound || w[qq].jb_last < 0;
        const bool hread = t >= 2 && (t == 2 || ((t - 1) & (t - 2)) == 0 || nobound);
        if (hread) {
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga.p[qq].sc_hist + (gw & (SC_COPIES - 1)) * SC_HSTRIDE + lane),
                                                 (__attribute__((address_space(3))) void*)(myp + RING + 512 + qq * QAREA), 4, 0, 16);
        }
        AS_TICKET();                     // (DYN: the chunk after `cnxt`)
        marked = first ? 0 : inflight;
        const bool hmark = hread;
        first = false;
        i32x4g hi = {0, 0, 0, 0}, xs = {0, 0, 0, 0};
        qa = qa_start;
        for (int c = 0; c < C; c += U) {
            if (inflight == NSLOT - U) {
                if (marked > 0 && hmark) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K1 + KB + NQ) : "memory");
                else if (marked > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K1 + KB) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K1) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            marked = marked > U ? marked - U : 0;
            inflight -= U;
            AS_TILE_ISSUE();
            i32x4g xv0, xv1, b10, b20, b11, b21;
            const unsigned a0 = my0 + cur + lane * 16;
            asm volatile(
                "ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:1024\n\tds_read_b128 %2, %7\n\tds_read_b128 %3, %7 offset:16\n\t"
                "ds_read_b128 %4, %8\n\tds_read_b128 %5, %8 offset:16\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(xv0), "=&v"(xv1), "=&v"(b10), "=&v"(b20), "=&v"(b11), "=&v"(b21)
                : "v"(a0), "v"(qa), "v"(qa + (qstep >> 1))
                : "memory");
            cur = cur + U * 1024 == RING ? 0 : cur + U * 1024;
            qa += qstep;
            hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(xv0, b10, hi, 0, 0, 0);
            xs = __builtin_amdgcn_mfma_i32_16x16x64_i8(xv0, b20, xs, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(xv1, b11, hi, 0, 0, 0);
            xs = __builtin_amdgcn_mfma_i32_16x16x64_i8(xv1, b21, xs, 0, 0, 0);
        }
        if (C <= NSLOT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float aux = lds_read1(ax0 + lane * 4);
        const float fa = lds_read1(fx0 + lane * 4);
        // the accumulators into "lane r holds row r": [HI | XS][query][64 rows]; lane l writes rows 16 (l % 4) + 4 (l / 16) + {0..3} of its query
        if (myq < NQ) {
            const unsigned ta = tr0 + (unsigned)((myq * 64 + 16 * (lane & 3) + 4 * (lane >> 4)) * 4);
            asm volatile("s_nop 15\n\ts_nop 3\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:%3" ::"v"(ta), "v"(hi), "v"(xs), "n"(NQ * 256) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
            const int h_ = (int)lds_read1u(tr0 + (unsigned)((qq * 64 + lane) * 4));
            const int x_ = (int)lds_read1u(tr0 + (unsigned)(NQ * 256 + (qq * 64 + lane) * 4));
            const long long tot = (long long)h_ * 128 + (long long)x_;
            const float mydot = (float)tot * (fa * ga.p[qq].faq);
            scan_chunk_end<true, GANG_PEND>(ga.p[qq], w[qq], ga.dots[qq], t, DYN ? (1 << 30) : rounds, gw, lane, base, cnt, mydot, aux, hread, qa0 + (unsigned)(qq * QAREA),
                                 qa0 + (unsigned)(qq * QAREA + 256), nq32[qq], inq32[qq]);
        }
        if (DYN) {   // (the ticket drawn at this chunk's start has retired: C >= 16 items were issued behind it)
            asm volatile("" : "+v"(tk));
            ccur = cnxt;
            cnxt = NW + (int64_t)__builtin_amdgcn_readfirstlane(tk) * ng + grp;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) scan_wave_report<true>(ga.p[qq], w[qq], gw, lane, qa0 + (unsigned)(qq * QAREA + 256));
#undef AS_TILE_ISSUE
#undef AS_TILE_ENTER
#undef AS_CHUNK
#undef AS_CHUNK_ID
#undef AS_TICKET
}

// generic width (dp > 2048): query re-read from L1 per chunk
__global__ __launch_bounds__(256) void scan_dots_f32_generic_kernel(const float* __restrict__ x32, const float* __restrict__ q32,
                                                                    int64_t dp, int64_t r0, int64_t r1, float* __restrict__ dots,
                                                                    PreArgs pre) {
    const int lane = lane_id();
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    const float nq32 = pre.info->nq32, inq32 = pre.info->inq32;
    int full = 0;
    for (int64_t row = r0 + gw; row < r1; row += nw) {
        const float* pa = x32 + row * dp;
        float s = 0.0f;
        for (int64_t c = 4 * lane; c < dp; c += 256) {
            const f32x4 v = *(const f32x4*)(pa + c);
            const f32x4 q = *(const f32x4*)(q32 + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fmaf(v[e], q[e], s);
        }
        s = wave_sum(s);
        if (lane == 0) {
            dots[row] = s;
            prefilter_f32(pre, row, s, (pre.metric == AS_METRIC_L2 ? pre.n32 : pre.inorm32)[row], nq32, inq32, full);
        }
    }
}

// exact mode: fp64 accumulation over the fp64 items (or the widened fp32 items when lossless)
__global__ __launch_bounds__(256) void scan_dots_f64_kernel(const float* __restrict__ x32, const double* __restrict__ x64,
                                                            const double* __restrict__ q64, int64_t d, int64_t dp, int64_t r0,
                                                            int64_t r1, double* __restrict__ dots, PreArgs pre) {
    const int lane = lane_id();
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    const double nq = pre.info->nq;
    int full = 0;
    for (int64_t row = r0 + gw; row < r1; row += nw) {
        double s = 0.0;
        if (x64) {
            const double* p = x64 + row * d;
            for (int64_t c = lane; c < d; c += 64) s += p[c] * q64[c];
        } else {
            const float* p = x32 + row * dp;
            for (int64_t c = lane; c < d; c += 64) s += (double)p[c] * q64[c];
        }
        s = wave_sum(s);
        if (lane == 0) {
            dots[row] = s;
            if (pre.enabled && !full && row < pre.n && row != pre.exclude) {
                const double ni = pre.n64[row];
                double key, bound;
                if (pre.metric == AS_METRIC_L2) {
                    key = ni + nq - 2.0 * s;
                    bound = pre.epskey + pre.coef * (ni + nq);
                } else {
                    const double den = sqrt(ni * nq);
                    const double c = den > 0.0 ? s / den : 0.0;
                    key = cosine_distance(c);
                    bound = pre.epskey + pre.coef;
                }
                if (key <= bound) {
                    const int slot = atomicAdd(&pre.infow->knn_cnt, 1);
                    if (slot < CAND_CAP) {
                        ((double*)pre.ckey)[slot] = key;
                        pre.cidx[slot] = (int)row;
                    } else {
                        full = 1;
                    }
                }
            }
        }
    }
}

// K-chunk passes of the batched scan (rows wider than 768 floats): the passes' fp32 partial dots -> the epilogue of
// scan_gemm_kernel: cosines as fp16 (or the dots as fp32) in the slot buffers and the fused k-NN prefilter.  A thread takes
// 4 slots of one row (16 contiguous bytes of every partial tile); 32 x N x 4 B per pass read once: 8 % of the item bytes.
__global__ __launch_bounds__(256) void gemm_combine_kernel(const float* __restrict__ part, int npass, int64_t pstride, int64_t r0, int64_t r1,
                                                           float* __restrict__ dots, int64_t ts, PreArgs pre, int nb, int half_dots, int i8) {
    const int64_t tile = blockIdx.x;
    const int quad = threadIdx.x >> 5, r = threadIdx.x & 31;
    const int64_t row = r0 + tile * 32 + r;
    const int64_t off = (row >> 5) * ts + (quad * 32 + (row & 31)) * 4;
    f32x4 mine = *(const f32x4*)(part + off);
    for (int p = 1; p < npass; ++p) mine += *(const f32x4*)(part + (int64_t)p * pstride + off);   // pass order: one summation order per launch geometry
    if (i8) {   // the passes left unscaled integer sums (int8 images): x . q = t fa_row fa_q
        const float far = pre.fa8[row];
#pragma unroll
        for (int e = 0; e < 4; ++e) mine[e] *= far * pre.faqv[4 * quad + e];
    }
    const float aux = pre.metric == AS_METRIC_L2 ? pre.n32[row] : pre.inorm32[row];   // padded arrays: readable past r1
    if (half_dots) {
        const float inr = pre.metric == AS_METRIC_L2 ? (aux > 0.0f ? rsqrtf(aux) : 0.0f) : aux;
        u32x2 pk;
        pk[0] = pack_half2(mine[0] * inr * pre.info[4 * quad].inq32, mine[1] * inr * pre.info[4 * quad + 1].inq32);
        pk[1] = pack_half2(mine[2] * inr * pre.info[4 * quad + 2].inq32, mine[3] * inr * pre.info[4 * quad + 3].inq32);
        *(u32x2*)((char*)dots + off * 2) = pk;
    } else {
        *(f32x4*)(dots + off) = mine;
    }
    if (!(pre.enabled && row < r1 && row < pre.n && row != pre.exclude)) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int b = 4 * quad + e;
        if (b >= nb) continue;   // idle slot
        const float nq = pre.metric == AS_METRIC_L2 ? pre.info[b].nq32 : pre.info[b].inq32;
        float key, bound;
        if (pre.metric == AS_METRIC_L2) {
            key = fmaf(-2.0f, mine[e], aux + nq);
            bound = ((float)pre.epskey + (float)pre.coef * (aux + nq)) * 1.000001f;
        } else {
            key = 1.0f - fmaxf(0.0f, mine[e] * aux * nq);
            bound = ((float)pre.epskey + (float)pre.coef) * 1.000001f;
        }
        // (a full buffer is left alone: the counter only has to exceed the capacity -- a million atomics on one word cost 7 ms)
        if (key <= bound && __hip_atomic_load(&pre.infow[b].knn_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= CAND_CAP) {
            const int slot = atomicAdd(&pre.infow[b].knn_cnt, 1);
            if (slot < CAND_CAP) {
                ((float*)pre.ckey)[(int64_t)b * CAND_CAP + slot] = key;
                pre.cidx[(int64_t)b * CAND_CAP + slot] = (int)row;
            }
        }
    }
}

// ------------------------------------------------------------------ host side
// fp32/fp64 error coefficient of one dot product: (terms in the longest rounding chain + slack) * u.
// Wave-per-row scans sum dp/64 fused terms per lane before a 6-level butterfly; the MFMA pass
// accumulates a quarter of the columns in sequence (two roundings per term, in case the matrix
// core rounds the products) and adds four partials.
// K-chunk passes of the batched MFMA scan: a wave keeps the query fragments of GEMM_NSW slabs in registers, a block of 4
// waves covers 4 * GEMM_NSW * 32 = 768 columns per pass -- 4 * GEMM_NSW_WIDE * 32 = 1024 in the kernel instantiated for wide
// rows (bf16 products only: the fp32 fragments of 8 slabs do not fit) --; wider rows take P = ceil(dp / cap) passes of even
// width (whole slabs).  Returns P; *chunk = columns of a pass (the last one takes what is left).
int gemm_chunks(int64_t dp, int64_t* chunk, bool bf16_products) {
    const int64_t cap = 4 * (dp > 4 * GEMM_NSW * 32 && bf16_products ? GEMM_NSW_WIDE : GEMM_NSW) * 32;
    const int P = (int)((dp + cap - 1) / cap);
    *chunk = ((dp + P - 1) / P + 31) / 32 * 32;
    return P;
}

double coef_query(const as_query* q, bool exact) {
    // single query scanned on the int8 two-digit image: |dot - x.q| <= |x||q| (u_q + U + v_q V) + four fp32 roundings of the
    // scaling -- the query's own measured residue norms with the items' maxima (query_begin forms it per query)
    if (!exact && q->i8_scan) return q->coef_i8;   // (batched workspace: an a-priori bound from the slots' s_q / |q|, host_batch_coef)
    const double u = exact ? 1.1102230246251565e-16 : 5.9604644775390625e-8;
    const int64_t dp = q->sp->dp;
    if (!exact && q->cap > 1) {
        // the batched MFMA pass.  fp32 pipe: two roundings per term of a wave's quarter of the columns.  bf16 pipe (BF3, with
        // the fp16 cosines): three exact products per column -- three times the accumulated terms -- and the operands' dropped
        // remainders, 3 * 2^-16 of |q_k x_k| per term, at most that of |q| |x| in the sum (Cauchy-Schwarz)
        // Rows wider than 768 floats take P K-chunk passes (gemm_chunks): every pass's error is that of its own columns --
        // sum_p |q_p||x_p| <= |q||x| -- and gemm_combine_kernel's P - 1 fp32 additions of the partials come on top.
        int64_t chunk = dp;
        const int P = gemm_chunks(dp, &chunk, q->half_enabled && q->ss.dots_rs == 4);
        const int64_t kw = ((chunk / 32 + 3) / 4) * 32;
        if (q->half_enabled && q->ss.dots_rs == 4) return (double)(6 * kw + 3 + 24 + P) * u + 3.0 * 1.52587890625e-5 * 1.01;   // 3 * 2^-16
        return (double)(2 * kw + 3 + 24 + P) * u;
    }
    return (double)(dp / 64 + 24) * u;
}

PreArgs make_pre(as_query* q, double eps, int64_t exclude, bool enabled) {
    const as_space* sp = q->sp;
    PreArgs p;
    p.n32 = sp->n32; p.inorm32 = sp->inorm32; p.n64 = sp->n64; p.info = q->info; p.infow = q->info;
    p.ckey = q->ckey_k; p.cidx = q->cidx_k;
    p.metric = sp->opts.metric;
    p.epskey = p.metric == AS_METRIC_L2 ? eps * eps : eps;
    p.coef = coef_query(q, q->exact != 0);
    p.n = sp->n; p.exclude = exclude; p.enabled = enabled ? 1 : 0;
    p.host_q = q->host_q;
    if (q->i8_scan && q->cap == 1) {
        p.fa8 = sp->fa8;
        p.q8 = q->coarse ? q->hq8h_dev : q->hq8_dev;
        p.faq = q->h_faq;
    }
    p.tile_ctrs = q->sc_hist;   // (the tile scan's chunk cursors live in the histogram copies' padding)
    if (q->host_q) {
        p.nq = q->h_nq; p.inq = q->h_inq;
        p.nq32 = (float)q->h_nq; p.inq32 = q->h_nq > 0.0 ? (float)(1.0 / sqrt(q->h_nq)) : 0.0f;
        p.q64_host = q->hq_dev; p.q64_dev = q->q64; p.qdp = (int)sp->dp;
    }
    if (q->fused_tail && enabled) {
        // window of the cosine bound + slack for the fp32 cosine of the scan against the fp64-over-fp32-dot cosine of the
        // finish kernel's keys (a few ulp of fp32 each way)
        p.sc_enabled = 1;
        p.sc_m = q->Ms;
        // ... and for the scan's own error, twice: a cosine off by at most `coef` (the int8 image: 3e-4 .. 2e-3; fp32: 2e-6) both in
        // the rows that set the bound and in the row held against it
        p.sc_w = (float)((1.0 - q->tau_cur) / (2.0 * q->tau_cur) + 2.0 * p.coef * 1.0001 + 1.0e-5);
        p.sc_idx = q->sc_widx;
        p.sc_hist = q->sc_hist;
        p.sc_late = q->sc_late;
        q->last_sc_m = p.sc_m;
        q->last_sc_w = p.sc_w;
#ifdef AS_ABLATION   // measurement switches that return wrong answers exist in `make ABLATION=1` builds only
        static const int dbg = getenv("ARROWSPACE_SC_DBG") ? atoi(getenv("ARROWSPACE_SC_DBG")) : 0;
        p.sc_dbg = dbg;
#endif
    }
    return p;
}

static constexpr size_t gemm_lds(int nbuf, bool i8 = false) { return sizeof(float) * ((size_t)4 * nbuf * 1024 + 4 * 3 * 64 * 4 + 4 * 64 + (i8 ? 4 * 64 : 0)); }
static constexpr size_t dma_lds(int nch, int nslot, bool sc = false, bool i8 = false) {
    return 4 * ((size_t)nslot * nch * 1024 + 256 + (sc ? 256 + SC_PEND * 8 : 0) + (i8 ? 256 : 0));
}

// launch geometry of the tile scan, <blocks per CU><two digits: ring KiB per wave>; ARROWSPACE_TILE_GEOM at load, as_set_tuning("tile_geom", v) later
static std::atomic<int> g_tile_geom{getenv("ARROWSPACE_TILE_GEOM") ? atoi(getenv("ARROWSPACE_TILE_GEOM")) : 208};
void set_tile_geom(int v) { g_tile_geom.store(v, std::memory_order_relaxed); }
// chunk schedule of the tile scan: 1 dynamic (an atomic cursor: scan_tile_kernel_dyn), 0 equal shares; ARROWSPACE_TILE_DYN, as_set_tuning("tile_dyn", v)
static std::atomic<int> g_tile_dyn{getenv("ARROWSPACE_TILE_DYN") ? atoi(getenv("ARROWSPACE_TILE_DYN")) : 1};
void set_tile_dyn(int v) { g_tile_dyn.store(v, std::memory_order_relaxed); }   // (0 static, 1 dynamic, 16 / 32: dynamic with chunks of that many rows)

static constexpr size_t gang_lds(int nslot, int nq, int64_t chunks) {
    return 4 * ((size_t)nslot * 1024 + 512 + (size_t)nq * (256 + GANG_PEND * 8) + (size_t)nq * 512) + (size_t)nq * chunks * 32 + 32;
}

static constexpr size_t tile_lds(int nslot, bool sc, int64_t chunks) {
    return 4 * ((size_t)nslot * 1024 + 512 + (sc ? 256 + TILE_PEND * 8 : 0)) + (size_t)chunks * 32;
}

// The dynamic-LDS opt-in is a per-device attribute of a kernel: set it for every scan kernel on the device a
// workspace is created on (query_alloc), not once per process -- a second device would never be opted in.
as_status set_scan_attrs() {
#define AS_ATTR(KERN, BYTES) AS_HIP(hipFuncSetAttribute((const void*)(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)))
    AS_ATTR((scan_gemm_kernel<3, 0, 2>), gemm_lds(3));
    AS_ATTR((scan_gemm_kernel<4, 0, 0>), gemm_lds(4));
#ifdef AS_ABLATION
    AS_ATTR((scan_gemm_kernel<4, 1, 2>), gemm_lds(4));
#endif
    AS_ATTR((scan_gemm_kernel<4, 0, 2>), gemm_lds(4));
    AS_ATTR((scan_gemm_kernel<3, 0, 2, true>), gemm_lds(3));
    AS_ATTR((scan_gemm_kernel<4, 0, 0, true>), gemm_lds(4));
    AS_ATTR((scan_gemm_kernel<4, 0, 2, true>), gemm_lds(4));
    AS_ATTR((scan_gemm_kernel<4, 0, 2, true, GEMM_NSW_WIDE>), gemm_lds(4));
    AS_ATTR((scan_gemm_kernel<4, 0, 2, false, GEMM_NSW, true>), gemm_lds(4, true));
    AS_ATTR((scan_gemm_kernel<4, 0, 2, false, GEMM_NSW_WIDE, true>), gemm_lds(4, true));
    AS_ATTR((scan_gemm_dual_kernel<4>), gemm_lds(4, true));
    AS_ATTR((scan_tile_kernel<6, true>), tile_lds(6, true, 256));
    AS_ATTR((scan_tile_kernel<8, true>), tile_lds(8, true, 256));
    AS_ATTR((scan_tile_kernel<12, true>), tile_lds(12, true, 256));
    AS_ATTR((scan_tile_kernel<16, true>), tile_lds(16, true, 256));
    AS_ATTR((scan_tile_kernel<8, false>), tile_lds(8, false, 256));
    AS_ATTR((scan_tile_kernel_dyn<8, true>), tile_lds(8, true, 256));
    AS_ATTR((scan_tile_kernel_dyn<8, false>), tile_lds(8, false, 256));
    AS_ATTR((scan_tile_gang_kernel<8, 2>), gang_lds(8, 2, 256));
    AS_ATTR((scan_tile_gang_kernel<8, 3>), gang_lds(8, 3, 256));
    AS_ATTR((scan_tile_gang_kernel<8, 4>), gang_lds(8, 4, 256));
    AS_ATTR((scan_tile_gang_kernel<8, 2, true>), gang_lds(8, 2, 256));
    AS_ATTR((scan_tile_gang_kernel<8, 3, true>), gang_lds(8, 3, 256));
    AS_ATTR((scan_tile_gang_kernel<8, 4, true>), gang_lds(8, 4, 256));
    AS_ATTR((scan_tile_kernel<8, true, 4>), tile_lds(8, true, 256));
    AS_ATTR((scan_tile_kernel<12, true, 4>), tile_lds(12, true, 256));
    AS_ATTR((scan_tile_kernel<16, true, 4>), tile_lds(16, true, 256));
    AS_ATTR((scan_dma_kernel<1, 8>), dma_lds(1, 8));
    AS_ATTR((scan_dma_kernel<2, 8>), dma_lds(2, 8));
    AS_ATTR((scan_dma_kernel<2, 5>), dma_lds(2, 5));
    AS_ATTR((scan_dma_kernel<2, 4>), dma_lds(2, 4));
    AS_ATTR((scan_dma_kernel<3, 5>), dma_lds(3, 5));
    AS_ATTR((scan_dma_kernel<4, 4>), dma_lds(4, 4));
    AS_ATTR((scan_dma_kernel<1, 8, false, true>), dma_lds(1, 8, false, true));
    AS_ATTR((scan_dma_kernel<2, 4, false, true>), dma_lds(2, 4, false, true));
    AS_ATTR((scan_dma_kernel<2, 6, false, true>), dma_lds(2, 6, false, true));
    AS_ATTR((scan_dma_kernel<2, 8, false, true>), dma_lds(2, 8, false, true));
    AS_ATTR((scan_dma_kernel<2, 6, true, true>), dma_lds(2, 6, true, true));
    AS_ATTR((scan_dma_kernel<2, 8, true, true>), dma_lds(2, 8, true, true));
    AS_ATTR((scan_dma_kernel<3, 5, false, true>), dma_lds(3, 5, false, true));
    AS_ATTR((scan_dma_kernel<4, 4, false, true>), dma_lds(4, 4, false, true));
    AS_ATTR((scan_dma_kernel<1, 8, true, true>), dma_lds(1, 8, true, true));
    AS_ATTR((scan_dma_kernel<2, 4, true, true>), dma_lds(2, 4, true, true));
    AS_ATTR((scan_dma_kernel<3, 5, true, true>), dma_lds(3, 5, true, true));
    AS_ATTR((scan_dma_kernel<4, 4, true, true>), dma_lds(4, 4, true, true));
    AS_ATTR((scan_dma_kernel<5, 3, false, true>), dma_lds(5, 3, false, true));
    AS_ATTR((scan_dma_kernel<6, 3, false, true>), dma_lds(6, 3, false, true));
    AS_ATTR((scan_dma_kernel<7, 3, false, true>), dma_lds(7, 3, false, true));
    AS_ATTR((scan_dma_kernel<8, 3, false, true>), dma_lds(8, 3, false, true));
    AS_ATTR((scan_dma_kernel<5, 3, true, true>), dma_lds(5, 3, true, true));
    AS_ATTR((scan_dma_kernel<6, 3, true, true>), dma_lds(6, 3, true, true));
    AS_ATTR((scan_dma_kernel<7, 3, true, true>), dma_lds(7, 3, true, true));
    AS_ATTR((scan_dma_kernel<8, 3, true, true>), dma_lds(8, 3, true, true));
    AS_ATTR((scan_dma_kernel<1, 8, true>), dma_lds(1, 8, true));
    AS_ATTR((scan_dma_kernel<2, 4, true>), dma_lds(2, 4, true));
    AS_ATTR((scan_dma_kernel<3, 5, true>), dma_lds(3, 5, true));
    AS_ATTR((scan_dma_kernel<4, 4, true>), dma_lds(4, 4, true));
#undef AS_ATTR
    return AS_OK;
}

// chunk schedule of a tile scan over `rows` rows that collects the scorer's candidates (launch_scan's, for the coarse operand)
static void tile_schedule(const as_query* q, int64_t rows, int bpc, int64_t* nblk, int* rounds, int* tail_rows, int* crows) {
    static const int sc_rows_per_block = getenv("ARROWSPACE_SC_ROWS_PER_BLOCK") ? atoi(getenv("ARROWSPACE_SC_ROWS_PER_BLOCK")) : 128;
    const int64_t want = std::max<int64_t>(1, (rows + 63) / 64);
    const int64_t want_sc = std::max<int64_t>(1, rows / std::max(sc_rows_per_block, 64));
    *nblk = std::min<int64_t>(std::min(want, want_sc), bpc * (int64_t)q->cus);
    const int64_t NW = *nblk * 4;
    auto chunks = [&](int c) { return rows / (NW * c) + (rows % (NW * c) ? 1 : 0); };
    *crows = chunks(64) >= 2 ? 64 : (chunks(32) >= 2 ? 32 : (chunks(16) >= 2 ? 16 : 64));
    *rounds = (int)(rows / (NW * *crows));
    const int64_t rem = rows - (int64_t)*rounds * NW * *crows;
    *tail_rows = (int)((rem + NW - 1) / NW);
}

// ONE coarse tile scan for the single queries of n = 2 .. 4 workspaces of one space (every member host-prepared, coarse, collecting
// scorer candidates, over all rows), on `st`; pre[i] = make_pre of member i
as_status launch_scan_gang(as_query* const* m, const PreArgs* pre, int n, hipStream_t st) {
    const as_space* sp = m[0]->sp;
    const int64_t rows = m[0]->r1 - m[0]->r0;
    if (n < 2 || n > 4 || rows <= 0 || !sp->x8h) {
        set_err("launch_scan_gang: bad gang");
        return AS_EINVAL;
    }
    int64_t nblk;
    int rounds, tail_rows, crows;
    tile_schedule(m[0], rows, 2, &nblk, &rounds, &tail_rows, &crows);
    GangArgs ga;
    for (int i = 0; i < 4; ++i) {
        ga.p[i] = pre[i < n ? i : 0];
        ga.dots[i] = m[i < n ? i : 0]->dots32;
    }
    for (int i = 0; i < n; ++i) {
        m[i]->sc_nw = (int)(nblk * 4);
        m[i]->dots_half = 0;
    }
    const int C = (int)(sp->dp8 / 16);
    const signed char* xt = (const signed char*)sp->x8h;
    // (the chunks by tickets under the single-query kernel's conditions: launch_scan)
    const int64_t NWg = nblk * 4;
    const int ng = nblk >= 32 ? SC_COPIES : (nblk >= 8 ? 8 : (nblk >= 4 ? 4 : (nblk >= 2 ? 2 : 1)));
    const bool dyn = C >= 16 && ga.p[0].tile_ctrs && g_tile_dyn.load(std::memory_order_relaxed) && rows >= NWg * crows * 3;
#define AS_GANG(NQ_)                                                                                                   \
    do {                                                                                                               \
        if (dyn)                                                                                                       \
            hipLaunchKernelGGL((scan_tile_gang_kernel<8, NQ_, true>), dim3((unsigned)nblk), dim3(256), gang_lds(8, NQ_, C), st, xt, C, m[0]->r0, m[0]->r1, ga, \
                               rounds, tail_rows, crows, ng);                                                           \
        else                                                                                                           \
            hipLaunchKernelGGL((scan_tile_gang_kernel<8, NQ_, false>), dim3((unsigned)nblk), dim3(256), gang_lds(8, NQ_, C), st, xt, C, m[0]->r0, m[0]->r1, ga, \
                               rounds, tail_rows, crows, ng);                                                           \
    } while (0)
    if (n == 2) AS_GANG(2);
    else if (n == 3) AS_GANG(3);
    else AS_GANG(4);
#undef AS_GANG
    AS_HIP(hipGetLastError());
    return AS_OK;
}

// Can the batched passes of workspaces a and b (both begun with defer_scan: their queries staged and quantised, their PreArgs
// made) share one read of the items?  Both on the int8 image with fp16 cosines, the same rows, image rows of up to 384 floats.
bool scan_dual_ok(const as_query* a, const as_query* b) {
    static const bool off = getenv("ARROWSPACE_NO_BATCH_DUAL") != nullptr;
    if (off || a == b || a->sp != b->sp) return false;
    const as_space* sp = a->sp;
    for (const as_query* q : {a, b})
        if (q->cap != QUERY_BATCH || q->exact || q->ss.dots_rs != 4 || !q->i8_scan || !q->q8img_dev || !q->faqv_dev || !q->half_enabled || !q->dots32) return false;
    if (!sp->x8 || !sp->fa8 || a->r0 != b->r0 || a->r1 != b->r1 || a->r1 <= a->r0 || a->ss.dots_ts != b->ss.dots_ts) return false;
    return sp->dp8 / 2 <= 4 * 3 * 32;
}

// ... then ONE launch on a's stream serves both (scan_gemm_dual_kernel); the caller orders b's stream around it
as_status launch_scan_dual(as_query* a, as_query* b, const PreArgs& pa, const PreArgs& pb, hipStream_t st) {
    const as_space* sp = a->sp;
    const int64_t rows = a->r1 - a->r0, ld = sp->dp8 / 2;
    DualArgs da;
    as_query* m[2] = {a, b};
    const PreArgs* pr[2] = {&pa, &pb};
    for (int s = 0; s < 2; ++s) {
        da.p[s] = *pr[s];
        da.p[s].fa8 = sp->fa8;
        da.p[s].faqv = m[s]->faqv_dev;
        da.q[s] = (const float*)m[s]->q8img_dev;
        da.dots[s] = m[s]->dots32;
        da.nb[s] = m[s]->nb;
        m[s]->dots_half = 1;
    }
    const int64_t nrb = (rows + 31) / 32;
    const int64_t grid = std::max<int64_t>(1, std::min<int64_t>(nrb, 2 * a->cus));   // (50 .. 87 % of it, for the other pair's kernels to run beside: 122 000 .. 125 000 against 129 700 queries/s)
    hipLaunchKernelGGL((scan_gemm_dual_kernel<4>), dim3((unsigned)grid), dim3(256), gemm_lds(4, true), st, (const float*)sp->x8, ld, a->r0, a->r1,
                       a->ss.dots_ts, da);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

as_status launch_scan(as_query* q, const PreArgs& pre) {
    const as_space* sp = q->sp;
    const int64_t rows = q->r1 - q->r0;
    if (rows <= 0) return AS_OK;
    hipStream_t st = q->stream;
    q->dots_half = 0;   // (only the batched MFMA pass below writes fp16 cosines)
    if (q->exact) {
        if (!q->dots64) AS_HIP(hipMalloc(&q->dots64, sizeof(double) * (sp->np + ROW_TILE)));
        const unsigned grid = (unsigned)std::min<int64_t>((rows + 3) / 4, 4096);
        hipLaunchKernelGGL(scan_dots_f64_kernel, dim3(grid), dim3(256), 0, st, sp->x32, sp->x64, q->q64, sp->d, sp->dp, q->r0,
                           q->r1, q->dots64, pre);
    } else {
        // (single query on the int8 two-digit image: the image's rows are dp8 / 2 floats long -- half the bytes, half the chunks)
        const bool i8 = q->cap == 1 && q->i8_scan && pre.q8 && pre.fa8 && sp->x8;
        const bool coarse = i8 && q->coarse && sp->x8h;   // (the planar high digits: rows of dp8 bytes)
        const int64_t ldrow = i8 ? (coarse ? sp->dp8 / 4 : sp->dp8 / 2) : sp->dp;
        const float* xrows = i8 ? (const float*)(coarse ? sp->x8h : sp->x8) : sp->x32;
        const int nch = (int)((ldrow + 255) / 256);
        if (q->cap > 1 && q->ss.dots_rs == 4) {
            // batched pass, GEMM-shaped: matrix pipe (bf16 head + tail with the fp16 cosines, else fp32), K split over the 4 waves of a block, 2 blocks per CU
            if (q->i8_scan && q->q8img_dev && q->faqv_dev && sp->x8 && q->half_enabled) {
                // int8 images of items and queries: rows of ld = dp8 / 2 image floats, 768 (6 slabs per wave: 1 536 columns) or
                // 1 024 (8 slabs: 2 048 columns) of them per launch, K-chunk passes side by side beyond
                const int64_t ld = sp->dp8 / 2;
                int64_t ichunk = ld;
                const int ipass = gemm_chunks(ld, &ichunk, true);
                if (ipass > 1 && !q->part32) {
                    set_err("launch_scan: the workspace has no partial buffer for rows of %lld floats", (long long)sp->dp);
                    return AS_EINVAL;
                }
                q->dots_half = 1;
                const int64_t ipstride = (sp->np + ROW_TILE) * (int64_t)q->cap;
                const size_t lds = gemm_lds(4, true);
                const int64_t nrb = (rows + 31) / 32;
                const int64_t per = std::max<int64_t>(1, std::min<int64_t>(nrb, (2 * q->cus) / ipass));
                PreArgs p8 = pre;
                p8.fa8 = sp->fa8;
                p8.faqv = q->faqv_dev;
                if (ichunk > 4 * GEMM_NSW * 32)
                    hipLaunchKernelGGL((scan_gemm_kernel<4, 0, 2, false, GEMM_NSW_WIDE, true>), dim3((unsigned)(per * ipass)), dim3(256), lds, st, (const float*)sp->x8,
                                       (const float*)q->q8img_dev, ld, q->r0, q->r1, ipass > 1 ? q->part32 : q->dots32, q->ss.dots_ts, p8, q->nb, 1, (int64_t)0,
                                       ichunk, ipass > 1 ? 1 : 0, ipass, ipstride);
                else
                    hipLaunchKernelGGL((scan_gemm_kernel<4, 0, 2, false, GEMM_NSW, true>), dim3((unsigned)(per * ipass)), dim3(256), lds, st, (const float*)sp->x8,
                                       (const float*)q->q8img_dev, ld, q->r0, q->r1, ipass > 1 ? q->part32 : q->dots32, q->ss.dots_ts, p8, q->nb, 1, (int64_t)0,
                                       ichunk, ipass > 1 ? 1 : 0, ipass, ipstride);
                AS_HIP(hipGetLastError());
                if (ipass > 1) {
                    hipLaunchKernelGGL(gemm_combine_kernel, dim3((unsigned)nrb), dim3(256), 0, st, (const float*)q->part32, ipass, ipstride, q->r0, q->r1,
                                       q->dots32, q->ss.dots_ts, p8, q->nb, 1, 1);
                    AS_HIP(hipGetLastError());
                }
                return AS_OK;
            }
            int64_t chunk = sp->dp;
            q->dots_half = q->half_enabled && q->ss.dots_rs == 4 ? 1 : 0;
            const int npass = gemm_chunks(sp->dp, &chunk, q->dots_half != 0);
            if (npass > 1 && !q->part32) {
                set_err("launch_scan: the workspace has no partial buffer for rows of %lld floats", (long long)sp->dp);
                return AS_EINVAL;
            }
            const int64_t pstride = (sp->np + ROW_TILE) * (int64_t)q->cap;   // floats between the passes' partial buffers
            int64_t kbase = 0, kcols = sp->dp;
            float* gdst = q->dots32;
            int graw = 0;
#define AS_GSCAN(NB_, DG, AX)                                                                                                \
    do {                                                                                                               \
        const size_t lds = gemm_lds(NB_);                                                                              \
        const int64_t nrb = (rows + 31) / 32;                                                                          \
        const unsigned grid = (unsigned)std::min<int64_t>(nrb, 2 * q->cus);                                            \
        if (q->dots_half && DG == 0)                                                                                   \
            hipLaunchKernelGGL((scan_gemm_kernel<NB_, DG, AX, true>), dim3(grid), dim3(256), lds, st, sp->x32, q->q32, sp->dp, q->r0, \
                               q->r1, gdst, q->ss.dots_ts, pre, q->nb, q->dots_half, kbase, kcols, graw, 1, (int64_t)0);       \
        else                                                                                                           \
            hipLaunchKernelGGL((scan_gemm_kernel<NB_, DG, AX>), dim3(grid), dim3(256), lds, st, sp->x32, q->q32, sp->dp, q->r0, q->r1, \
                               gdst, q->ss.dots_ts, pre, q->nb, q->dots_half, kbase, kcols, graw, 1, (int64_t)0);              \
    } while (0)
            if (chunk > 4 * GEMM_NSW * 32) {
                // rows wider than 768 floats on the bf16 pipe: 1024 columns per pass (8 slabs per wave), the K-chunk passes of
                // wider rows side by side in ONE launch (block b: pass b % npass), then the combine + epilogue launch
                const size_t lds = gemm_lds(4);
                const int64_t nrb = (rows + 31) / 32;
                const int64_t per = std::max<int64_t>(1, std::min<int64_t>(nrb, (2 * q->cus) / npass));
                hipLaunchKernelGGL((scan_gemm_kernel<4, 0, 2, true, GEMM_NSW_WIDE>), dim3((unsigned)(per * npass)), dim3(256), lds, st, sp->x32, q->q32, sp->dp,
                                   q->r0, q->r1, npass > 1 ? q->part32 : q->dots32, q->ss.dots_ts, pre, q->nb, q->dots_half, (int64_t)0, chunk,
                                   npass > 1 ? 1 : 0, npass, pstride);
                AS_HIP(hipGetLastError());
                if (npass > 1) {
                    hipLaunchKernelGGL(gemm_combine_kernel, dim3((unsigned)nrb), dim3(256), 0, st, (const float*)q->part32, npass, pstride, q->r0, q->r1,
                                       q->dots32, q->ss.dots_ts, pre, q->nb, q->dots_half, 0);
                    AS_HIP(hipGetLastError());
                }
                return AS_OK;
            }
            if (npass > 1) {
                // K-chunk passes: raw fp32 partials per pass, then one combine + epilogue launch over 32 x rows partial dots
                graw = 1;
                for (int p = 0; p < npass; ++p) {
                    kbase = (int64_t)p * chunk;
                    kcols = std::min<int64_t>(chunk, sp->dp - kbase);
                    gdst = q->part32 + (int64_t)p * pstride;
                    AS_GSCAN(4, 0, 2);
                    AS_HIP(hipGetLastError());
                }
                const int64_t nrb = (rows + 31) / 32;
                hipLaunchKernelGGL(gemm_combine_kernel, dim3((unsigned)nrb), dim3(256), 0, st, (const float*)q->part32, npass, pstride, q->r0, q->r1,
                                   q->dots32, q->ss.dots_ts, pre, q->nb, q->dots_half, 0);
                AS_HIP(hipGetLastError());
                return AS_OK;
            }
            if (q->gemm_variant == 1) AS_GSCAN(3, 0, 2);
            else if (q->gemm_variant == 2) AS_GSCAN(4, 0, 0);
#ifdef AS_ABLATION   // the no-MFMA skeleton (timing only, wrong results) is not in the product library
            else if (q->gemm_variant == 16) AS_GSCAN(4, 1, 2);
#endif
            else AS_GSCAN(4, 0, 2);
#undef AS_GSCAN
            AS_HIP(hipGetLastError());
            return AS_OK;
        }
        if (q->cap > 1) {
            // batched pass on VALU FMAs: QB queries per launch (query fragments in registers); dp <= 1024 only
#define AS_BSCAN(N)                                                                                                    \
    do {                                                                                                               \
        if (!q->scan_grid) {                                                                                           \
            int nb_ = 0;                                                                                               \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, scan_dots_batch_kernel<N>, 256, 0) != hipSuccess) nb_ = 2; \
            q->scan_grid = q->cus * std::max(1, std::min(nb_, 8));                                                     \
        }                                                                                                              \
        const unsigned grid = (unsigned)std::min<int64_t>((rows + 3) / 4, q->scan_grid);                               \
        for (int j0 = 0; j0 < q->nb; j0 += QB) {                                                                       \
            PreArgs pj = pre;                                                                                          \
            pj.info = pre.info + j0;                                                                                   \
            pj.infow = pre.infow + j0;                                                                                 \
            pj.ckey = (void*)((float*)pre.ckey + (int64_t)j0 * CAND_CAP);                                              \
            pj.cidx = pre.cidx + (int64_t)j0 * CAND_CAP;                                                               \
            hipLaunchKernelGGL(scan_dots_batch_kernel<N>, dim3(grid), dim3(256), 0, st, sp->x32, q->q32 + (int64_t)j0 * sp->dp, \
                               sp->dp, q->r0, q->r1, q->dots32, q->ss.dots, q->ss.dots_ts, q->ss.dots_rs, j0, pj);                     \
        }                                                                                                              \
    } while (0)
            switch (nch) {
                case 1: AS_BSCAN(1); break;
                case 2: AS_BSCAN(2); break;
                case 3: AS_BSCAN(3); break;
                default: AS_BSCAN(4); break;
            }
#undef AS_BSCAN
            AS_HIP(hipGetLastError());
            return AS_OK;
        }
        const int rev = (q->scan_variant & 1) ? (int)(q->scan_count++ & 1) : 0;
        // resident grid: every wave gets the same number of rows and all of them run at once
        // (a grid one block over residency costs a whole extra round at 1/8 occupancy)
#define AS_SCAN(N)                                                                                                     \
    do {                                                                                                               \
        if (!q->scan_grid) {                                                                                           \
            int nb = 0;                                                                                                \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, scan_dots_f32_kernel<N, true>, 256, 0) != hipSuccess) nb = 4; \
            q->scan_grid = q->cus * std::max(1, std::min(nb, 8));                                                      \
        }                                                                                                              \
        const unsigned grid = (unsigned)std::min<int64_t>((rows + 3) / 4, q->scan_grid);                               \
        if (q->scan_variant & 2)                                                                                       \
            hipLaunchKernelGGL((scan_dots_f32_kernel<N, false>), dim3(grid), dim3(256), 0, st, sp->x32, q->q32, sp->dp, \
                               q->r0, q->r1, q->dots32, pre, rev);                                                     \
        else                                                                                                           \
            hipLaunchKernelGGL((scan_dots_f32_kernel<N, true>), dim3(grid), dim3(256), 0, st, sp->x32, q->q32, sp->dp,  \
                               q->r0, q->r1, q->dots32, pre, rev);                                                     \
    } while (0)
        // default for rows up to 1024 floats: the LDS-DMA ring scan (ARROWSPACE_SCAN_VARIANT bit2 = register-staged scan)
        if ((nch <= 4 || (i8 && nch <= 8)) && !(q->scan_variant & 4)) {
            const int64_t want = std::max<int64_t>(1, (rows + 63) / 64);                 // blocks that still get >= 16 rows per wave
            // 2 blocks per CU, ring of 5 rows at 768 columns; rings of 4 or 6 rows and 3 blocks per CU measured the same or slower.
            // Short rows (up to 512 floats) pay the same per-row bookkeeping for half the bytes: 4 blocks per CU there
            // (ring of 4 two-KiB slots, or of 8 one-KiB slots) -- 400k x 384: 105 -> 99 us, x 256: 78.5 -> 75.7, x 512: 141 -> 137.
            // (measurement: ARROWSPACE_SCAN_GEOM=<blocks per CU><ring slots> for rows up to 512 floats, e.g. 28 = the old form)
            static const int geom = getenv("ARROWSPACE_SCAN_GEOM") ? atoi(getenv("ARROWSPACE_SCAN_GEOM")) : 0;
            // The int8 image at 2 chunks per lane (rows of 257 .. 512 image floats = 513 .. 1 024 columns): 2 blocks per CU -- 1M x 768
            // 0.270 -> 0.255 ms, 200k x 768 57.3 -> 55.8 us, 1M x 512 196 -> 190 us; 1 block 0.357 ms, rings of 6 / 8 slots 0.264 / 0.265 ms
            // (tools/scan_geom.sh); at 1 chunk per lane 4 blocks stay ahead (400k x 384: 64 us against 84 us with 2).
            const int bpc_default = i8 && nch == 2 ? 2 : 4;
            // (rows of 2 049 .. 4 096 columns of the image: rings of 3 rows of 5 .. 8 KB per wave -- 2 blocks per CU fit at 5 KB only)
            // The coarse scan reads TILES (scan_tile_kernel): its geometry is <blocks per CU><ring KiB per wave> -- 4 x 6 by default;
            // measurement: ARROWSPACE_TILE_GEOM=<blocks per CU><two digits of ring slots>, e.g. 308, 216)
            const int tgeom = g_tile_geom.load(std::memory_order_relaxed);
            const int tslots = tgeom % 100 >= 16 ? 16 : (tgeom % 100 >= 12 ? 12 : (tgeom % 100 >= 8 ? 8 : 6));
            const int tstep = tgeom / 1000 == 4 && tslots >= 8 ? 4 : 2;   // (thousands digit 4: four KiB per step instead of two)
            const int bpc = coarse ? std::max(1, std::min((tgeom / 100) % 10, 4))
                                   : (nch <= 2 ? std::max(1, std::min(geom ? geom / 10 : bpc_default, 4)) : (nch <= 5 ? 2 : 1));   // (<= 4: the wave reports are sized for 16 waves per CU)
            // (collecting the scorer's candidates: every wave needs a chunk in front of its last to publish from -- at least 32 rows
            // per wave, two chunks of 16; a 30 000-row index on 1 876 waves had 16 rows per wave, no bound, and every row a candidate)
            // (A/B: 256 rows per block instead of 128 -- 200k x 768 11 700 -> 9 000 queries/s, 100k x 768 13 700 -> 10 900, 65536 x 4096 unchanged)
            static const int sc_rows_per_block = getenv("ARROWSPACE_SC_ROWS_PER_BLOCK") ? atoi(getenv("ARROWSPACE_SC_ROWS_PER_BLOCK")) : 128;
            const int64_t want_sc = pre.sc_enabled ? std::max<int64_t>(1, rows / std::max(sc_rows_per_block, 64)) : want;
            const int64_t nblk = std::min<int64_t>(std::min(want, want_sc), bpc * (int64_t)q->cus);
            const int64_t NW = nblk * 4;
            q->sc_nw = (int)NW;
            // Chunks of 64 rows (a row per lane at the chunk's end).  The scan that also collects the scorer's candidates learns
            // its cosine bound from the chunks BEFORE a wave's last (scan_dma_kernel, SC): a shard of fewer than 64 rows per
            // wave (125k rows on 2 048 waves: one chunk each, no bound, every row a candidate -> the fused tail and the
            // one-exchange pass never applied to an 8-GPU shard of a 1M index) runs chunks of 32 or 16 rows instead: two or
            // more chunks per wave, the first publishes, the wave's last word reads the bound (as at 200k rows).
            int crows = 64;
            // (the largest chunk that still gives every wave TWO chunks, the remainder's counted: 262 144 rows on 4 096 waves are ONE
            // chunk of 64 each -- the coarse scan's candidates then never fitted --, two of 32 do; chunks of 16 where 32 would do
            // learn a poorer bound: the rows that stand out of 16 are fewer)
            if (pre.sc_enabled) {
                auto chunks = [&](int c) { return rows / (NW * c) + (rows % (NW * c) ? 1 : 0); };
                crows = chunks(64) >= 2 ? 64 : (chunks(32) >= 2 ? 32 : (chunks(16) >= 2 ? 16 : 64));
            }
            const int rounds = (int)(rows / (NW * crows));
            const int64_t rem = rows - (int64_t)rounds * NW * crows;
            const int tail_rows = (int)((rem + NW - 1) / NW);
            if (coarse) {
                const int C = (int)(sp->dp8 / 16);
                const signed char* xt = (const signed char*)sp->x8h;
#define AS_TSCAN(S)                                                                                                    \
    do {                                                                                                               \
        if (pre.sc_enabled)                                                                                            \
            hipLaunchKernelGGL((scan_tile_kernel<S, true>), dim3((unsigned)nblk), dim3(256), tile_lds(S, true, C), st, xt, C, q->r0, q->r1, \
                               q->dots32, pre, rounds, tail_rows, crows);                                              \
        else                                                                                                           \
            hipLaunchKernelGGL((scan_tile_kernel<8, false>), dim3((unsigned)nblk), dim3(256), tile_lds(8, false, C), st, xt, C, q->r0, q->r1, \
                               q->dots32, pre, rounds, tail_rows, crows);                                              \
    } while (0)
                // (three chunks per wave and more: with fewer -- 200 000 rows on 2 048 waves are a chunk and a half each -- equal shares
                // end together, tickets do not: the waves that end early report against a histogram half filled, the tail kernels then
                // evaluate several times the candidates -- same scan time, 9 200 instead of 13 000 queries/s at 200k x 768)
                if (C >= 16 && tslots == 8 && tstep == 2 && pre.tile_ctrs && g_tile_dyn.load(std::memory_order_relaxed) && rows >= NW * crows * 3) {
                    // dynamic chunk schedule (the default): chunks of `crows` rows by id, the waves' first ones their own
                    const int dynv = g_tile_dyn.load(std::memory_order_relaxed);
                    if (dynv == 16 || dynv == 32) crows = std::min(crows, dynv);
                    // groups of blocks, each with a cursor of its own: block b is in group (b + b / 8) mod ng -- with ng <= 8 the first
                    // ng blocks are the groups' first members, with 16 groups every group has a member among the first 23 blocks
                    const int ng = nblk >= 32 ? SC_COPIES : (nblk >= 8 ? 8 : (nblk >= 4 ? 4 : (nblk >= 2 ? 2 : 1)));
                    if (pre.sc_enabled)
                        hipLaunchKernelGGL((scan_tile_kernel_dyn<8, true>), dim3((unsigned)nblk), dim3(256), tile_lds(8, true, C), st, xt, C, q->r0, q->r1, q->dots32, pre, crows, ng);
                    else
                        hipLaunchKernelGGL((scan_tile_kernel_dyn<8, false>), dim3((unsigned)nblk), dim3(256), tile_lds(8, false, C), st, xt, C, q->r0, q->r1, q->dots32, pre, crows, ng);
                } else if (tstep == 4 && pre.sc_enabled) {
                    if (tslots == 16) hipLaunchKernelGGL((scan_tile_kernel<16, true, 4>), dim3((unsigned)nblk), dim3(256), tile_lds(16, true, C), st, xt, C, q->r0, q->r1, q->dots32, pre, rounds, tail_rows, crows);
                    else if (tslots == 12) hipLaunchKernelGGL((scan_tile_kernel<12, true, 4>), dim3((unsigned)nblk), dim3(256), tile_lds(12, true, C), st, xt, C, q->r0, q->r1, q->dots32, pre, rounds, tail_rows, crows);
                    else hipLaunchKernelGGL((scan_tile_kernel<8, true, 4>), dim3((unsigned)nblk), dim3(256), tile_lds(8, true, C), st, xt, C, q->r0, q->r1, q->dots32, pre, rounds, tail_rows, crows);
                } else
                switch (tslots) {
                    case 16: AS_TSCAN(16); break;
                    case 12: AS_TSCAN(12); break;
                    case 8: AS_TSCAN(8); break;
                    default: AS_TSCAN(6); break;
                }
#undef AS_TSCAN
                AS_HIP(hipGetLastError());
                return AS_OK;
            }
#define AS_DSCAN8(N, S)                                                                                                \
    do {                                                                                                               \
        if (pre.sc_enabled)                                                                                            \
            hipLaunchKernelGGL((scan_dma_kernel<N, S, true, true>), dim3((unsigned)nblk), dim3(256), dma_lds(N, S, true, true), st, \
                               xrows, q->q32_src, ldrow, q->r0, q->r1, q->dots32, pre, rounds, tail_rows, crows);             \
        else                                                                                                           \
            hipLaunchKernelGGL((scan_dma_kernel<N, S, false, true>), dim3((unsigned)nblk), dim3(256), dma_lds(N, S, false, true), st, \
                               xrows, q->q32_src, ldrow, q->r0, q->r1, q->dots32, pre, rounds, tail_rows, crows);             \
    } while (0)
#define AS_DSCAN(N, S)                                                                                                 \
    do {                                                                                                               \
        if (i8)                                                                                                        \
            AS_DSCAN8(N, (N == 2 ? 4 : S));                                                                            \
        else if (pre.sc_enabled)   /* (rows of up to 512 floats: the fused form exists with the ring of 4 only) */          \
            hipLaunchKernelGGL((scan_dma_kernel<N, (N == 2 ? 4 : S), true>), dim3((unsigned)nblk), dim3(256), dma_lds(N, (N == 2 ? 4 : S), true), st, \
                               sp->x32, q->q32_src, sp->dp, q->r0, q->r1, q->dots32, pre, rounds, tail_rows, crows);          \
        else                                                                                                           \
            hipLaunchKernelGGL((scan_dma_kernel<N, S>), dim3((unsigned)nblk), dim3(256), dma_lds(N, S), st, sp->x32, q->q32_src, sp->dp, \
                               q->r0, q->r1, q->dots32, pre, rounds, tail_rows, crows);                                       \
    } while (0)
            switch (nch) {
                case 1: AS_DSCAN(1, 8); break;
                case 2:
                    if (i8 && geom % 10 == 6) AS_DSCAN8(2, 6);
                    else if (i8 && geom % 10 == 8) AS_DSCAN8(2, 8);
                    else if (i8) AS_DSCAN(2, 4);
                    else if (geom % 10 == 5) AS_DSCAN(2, 5);
                    else if (geom % 10 == 8) AS_DSCAN(2, 8);
                    else AS_DSCAN(2, 4);
                    break;
                case 3: AS_DSCAN(3, 5); break;
                case 4: AS_DSCAN(4, 4); break;
                case 5: AS_DSCAN8(5, 3); break;   // (int8 image only: the dispatch above)
                case 6: AS_DSCAN8(6, 3); break;
                case 7: AS_DSCAN8(7, 3); break;
                default: AS_DSCAN8(8, 3); break;
            }
#undef AS_DSCAN
#undef AS_DSCAN8
            AS_HIP(hipGetLastError());
            return AS_OK;
        }
        switch (nch) {
            case 1: AS_SCAN(1); break;
            case 2: AS_SCAN(2); break;
            case 3: AS_SCAN(3); break;
            case 4: AS_SCAN(4); break;
            case 5: AS_SCAN(5); break;
            case 6: AS_SCAN(6); break;
            case 7: AS_SCAN(7); break;
            case 8: AS_SCAN(8); break;
            default: {
                const unsigned grid = (unsigned)std::min<int64_t>((rows + 3) / 4, q->cus * 8);
                hipLaunchKernelGGL(scan_dots_f32_generic_kernel, dim3(grid), dim3(256), 0, st, sp->x32, q->q32, sp->dp, q->r0, q->r1, q->dots32, pre);
            }
        }
#undef AS_SCAN
    }
    AS_HIP(hipGetLastError());
    return AS_OK;
}

}  // namespace as

#ifdef AS_STAMPS
// diagnostic build only (not declared in the public header)
extern "C" int as_debug_scan_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(as::g_scan_stamps), sizeof(unsigned long long) * 2 * 4096) == hipSuccess ? 0 : 1;
}
#endif
