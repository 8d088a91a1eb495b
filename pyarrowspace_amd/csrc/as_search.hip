// Lambda-blended search on gfx950.  Replaces `prepare_query_item` + `search_lambda_aware`
// (/root/reference/src/lib.rs:154,173; scorer form TAUMODE.md:33).  SPEC = DESIGN.md
// section 2 (S10, S11).  One HBM pass over the fp32 item matrix per query (scan_dots);
// everything after it works on N-length vectors that stay in L2 / Infinity Cache.
//
// Selection design: the k-NN side is a *filter* (rows whose fp32
// key beats the eps bound are appended to a small candidate buffer by the scan kernel
// itself); the scorer side derives a provable threshold from group maxima (the M-th best
// of G-row group maxima is a lower bound of the M-th best score), filters against it and
// ranks the few survivors.  Survivors are re-evaluated in fp64 and an a-posteriori check
// proves the result equals the fp64 answer, else the search is rerun in fp64.  A
// wavefront-shuffle list path (WaveList) is kept as the overflow fallback.
#include <algorithm>
#include <atomic>
#include <mutex>

#include "as_common.hpp"

namespace as {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CAND_CAP = 4096;  // candidate buffer of the filter path
constexpr int REC_CAP = 512;    // k-NN records q_lambda accepts (ranks x k)
constexpr int MAX_TOPK = 1024;  // largest topk
constexpr int MS_MAX = MAX_TOPK + 64;  // widest scorer candidate list (topk + margin, rounded to 64)
constexpr int HIT_CAP = 8 * (MAX_TOPK + 1) + 8;  // hit records hits_final accepts (ranks x (topk + 1))
constexpr int QB = 8;           // queries per VALU batched scan launch (query fragments live in registers)
constexpr int GQ = 32;          // queries per MFMA (GEMM-shaped) batched scan pass == slots of the batched workspace

// Batched searches run QB independent query "slots" side by side: every per-query buffer is
// an array over slots and blockIdx.z selects the slot (z = 0 for single-query searches).
struct SlotStride {
    int64_t dots;   // elements between consecutive slots' dots inside one 32-row tile (32: tile-major, batched workspace)
    int64_t dots_ts; // elements between consecutive 32-row tiles (32 = plain row order, single slot; 32 * slots batched)
    int64_t q;      // dp
    int64_t qin;    // d
    int64_t knn;    // k records
    int64_t hits;   // topk + 1 records
};

struct QInfo {
    double nq;        // |q|^2
    double lambda_q;
    double tau;
    double inq;       // 1/|q|
    double thr64;     // scorer threshold key (fp64 mode)
    float nq32, inq32;
    float thr32;      // scorer threshold key (fp32 mode)
    int status;       // as_status of the lambda step
    int knn_inexact;  // a-posteriori check of the k-NN candidate list failed
    int score_inexact;
    int knn_total;    // candidates that passed the eps prefilter
    int nhit;
    int knn_cnt;      // filter path: appended k-NN candidates
    int sc_cnt;       // filter path: appended scorer candidates
    int overflow;     // bit0: the k-NN candidate buffer overflowed (-> threshold repair over the kept dots), bit1: the scorer's (-> list path)
};

// everything a search writes into QInfo after the query itself was prepared (norms stay)
__device__ __forceinline__ void reset_query_state(QInfo* info) {
    info->lambda_q = 0.0;
    info->status = AS_OK;
    info->knn_inexact = 0;
    info->score_inexact = 0;
    info->knn_total = 0;
    info->nhit = 0;
    info->knn_cnt = 0;
    info->sc_cnt = 0;
    info->overflow = 0;
    info->thr32 = 0.0f;
    info->thr64 = 0.0;
}

struct HostOut {
    volatile int64_t seq;
    int64_t len;
    double lambda_q;
    int status, knn_inexact, score_inexact, overflow;
    int64_t idx[MAX_TOPK];
    double score[MAX_TOPK];
};

static std::atomic<int> g_search_stats{0};
struct RSel;

}  // namespace as

struct as_query {
    const as_space* sp = nullptr;
    const as_graph* gr = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    int own_records = 1;
    int64_t k = 0, topk = 0;
    int Mk = 32, Ms = 32;
    int nwaves = 0;
    int reuse = 0;           // staged path: the next scan call repairs the previous scan's overflow from its dots
    int cap = 1;             // query slots (GQ for the batched workspace)
    int gemm_variant = 0;    // batched MFMA scan (ARROWSPACE_GEMM_VARIANT): 1 = ring of 3 slabs, 2 = default cache policy, 16 = no MFMA (timing only)
    int nb = 1;              // active slots of the current launch sequence
    as::SlotStride ss{};
    int cus = 256;
    int scan_grid = 0;
    int scan_variant = 0;    // bit2: register-staged scan instead of the LDS-DMA ring; with it, bit0: alternate scan direction per query, bit1: temporal row loads
    int64_t scan_count = 0;
    int64_t r0 = 0, r1 = 0;
    int exact = 0;
    int robust = 0;          // 1: wavefront-list path instead of the filter path
    int64_t seq = 0;
    double* hq = nullptr;    // pinned host staging of the query (device-readable)
    double* hq_dev = nullptr;
    double* q64 = nullptr;   // [dp] zero padded
    float* q32 = nullptr;    // [dp]
    as::QInfo* info = nullptr;
    float* dots32 = nullptr; // [np]
    double* dots64 = nullptr;
    void* pkey = nullptr;    // list path: [nwaves][64] keys (sized for double)
    int* pidx = nullptr;
    void* ckey_k = nullptr;  // filter path candidate buffers (sized for double)
    int* cidx_k = nullptr;
    void* ckey_s = nullptr;
    int* cidx_s = nullptr;
    void* gmin = nullptr;    // group minima of the scorer key
    as::RSel* rsel = nullptr; // state of the exact global selection
    as_knn_rec* knn = nullptr;
    as_hit_rec* hits = nullptr;
    as::HostOut* hout = nullptr;  // pinned
    as::HostOut* hout_dev = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    int ev_valid = 0;
    double stats[4] = {0, 0, 0, 0};
};

namespace as {

// ------------------------------------------------------------------ small helpers
__device__ __forceinline__ unsigned int ord_bits(float v) {
    const unsigned int b = (unsigned int)__float_as_int(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ unsigned long long ord_bits(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ float from_ord(unsigned int u) {
    const unsigned int b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __int_as_float((int)b);
}
__device__ __forceinline__ double from_ord(unsigned long long u) {
    const unsigned long long b = (u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u;
    return __longlong_as_double((long long)b);
}

template <typename T> struct ord_of;
template <> struct ord_of<float> { typedef unsigned int type; static constexpr int passes = 4; };
template <> struct ord_of<double> { typedef unsigned long long type; static constexpr int passes = 8; };

// ------------------------------------------------------------------ query staging
__global__ void q_prepare_kernel(const double* __restrict__ qin, int64_t d, int64_t dp, double* __restrict__ q64,
                                 float* __restrict__ q32, QInfo* info, double tau) {
    __shared__ double sh[256];
    qin += (int64_t)blockIdx.z * d;
    q64 += (int64_t)blockIdx.z * dp;
    q32 += (int64_t)blockIdx.z * dp;
    info += blockIdx.z;
    double s = 0.0;
    for (int64_t c = threadIdx.x; c < dp; c += blockDim.x) {
        const double v = c < d ? qin[c] : 0.0;
        q64[c] = v;
        q32[c] = (float)v;
        s += v * v;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double nq = sh[0];
        info->nq = nq;
        info->inq = nq > 0.0 ? 1.0 / sqrt(nq) : 0.0;
        info->nq32 = (float)nq;
        info->inq32 = nq > 0.0 ? (float)(1.0 / sqrt(nq)) : 0.0f;
        info->tau = tau;
        reset_query_state(info);
    }
}

__global__ void q_from_row_kernel(const float* __restrict__ x32, const double* __restrict__ x64, int64_t d, int64_t dp,
                                  int64_t row, double* __restrict__ qin) {
    for (int64_t c = threadIdx.x; c < d; c += blockDim.x) qin[c] = x64 ? x64[row * d + c] : (double)x32[row * dp + c];
}

// ------------------------------------------------------------------ K7a scan: dots[i] = x_i . q  (+ k-NN prefilter)
struct PreArgs {
    const float* n32;
    const float* inorm32;
    const double* n64;
    const QInfo* info;
    QInfo* infow;
    void* ckey;
    int* cidx;
    double epskey, coef;
    int64_t n, exclude;
    int metric, enabled;
};

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
                                                              int64_t sd, int64_t ts, PreArgs pre) {
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
    float* __restrict__ mydots = dots + (int64_t)myq * sd;   // tile-major: [32-row tile][slot][32]
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
            mydots[(row >> 5) * ts + (row & 31)] = k1;
            prefilter_f32(mine, row, k1, aux, nq32, inq32, full);
        }
    }
}

// Batched scan as a GEMM (rows up to 768 floats): dots[GQ x rows] = Q . X^T on fp32 MFMA
// (v_mfma_f32_32x32x2_f32, A = 32 queries, B = 32 item rows).  A block is a team of 4 waves
// that splits K: wave w keeps the Q fragments of its quarter of the columns in registers for the
// whole launch (<= 6 slabs of 32 floats -> 96 VGPRs) and streams the matching quarter of every
// 32-row block by LDS-DMA into a private ring of NBUF XOR-swizzled slabs (same image as
// knn_mfma_dma_kernel) -- the K loop has no block barrier, only the wave's own vmcnt, and all
// of LDS is staging (2 blocks per CU, ~100 KB of rows in flight per CU).  The four partial
// 32x32 tiles meet in LDS once per row block; wave w then owns queries [8w, 8w+8) of the
// epilogue (store + fused kNN prefilter).  One HBM pass serves GQ queries: 2*GQ*dp flops per
// row against dp*4 bytes -- still HBM-bound at GQ=32.
constexpr int GEMM_NSW = 6;   // slabs per wave: dp <= 4 * 6 * 32

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
    if (n == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if (n == 4 || n == 5) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// always issued (an exec-masked store the compiler may not branch around): the wave's count of outstanding
// operations must never be smaller than the ring's bookkeeping assumes
__device__ __forceinline__ void store_dword_issued(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
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
// and cost the per-slot selection kernels their cache hits.
template <int NBUF, int DIAG = 0, int AUX = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void scan_gemm_kernel(
    const float* __restrict__ x32, const float* __restrict__ q32, int64_t dp, int64_t r0, int64_t r1, float* __restrict__ dots,
    int64_t sd, int64_t ts, PreArgs pre, int nb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    float* St = (float*)smem;   // per wave: NBUF slabs x [32 rows][32 floats]; Ex[owner wave][3 senders][64 lanes][4]; Ax[wave][64]
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ex0 = lds0 + 4 * NBUF * 4096, ax0 = ex0 + 4 * 3 * 64 * 16;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nslab = (int)(dp / 32), nsw = (nslab + 3) / 4;
    const int ks0 = wu * nsw;
    const int myns = max(0, min(nsw, nslab - ks0));
    f32x4 qf[GEMM_NSW][4];
#pragma unroll
    for (int ks = 0; ks < GEMM_NSW; ++ks)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qf[ks][s] = ks < myns ? *(const f32x4*)(q32 + (int64_t)l31 * dp + (ks0 + ks) * 32 + (2 * s + h) * 4) : f32x4{0, 0, 0, 0};
    // per-query constants of the prefilter, for the 4 queries this lane finishes: b = e + 8 wu + 4 h
    float nqv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) nqv[e] = pre.metric == AS_METRIC_L2 ? pre.info[e + 8 * wu + 4 * h].nq32 : pre.info[e + 8 * wu + 4 * h].inq32;
    // the loads above complete here, once: otherwise the compiler has to assume they are still pending inside
    // the loop and puts a vmcnt(0) -- which also waits for the whole prefetch ring -- in front of their first use
#pragma unroll
    for (int ks = 0; ks < GEMM_NSW; ++ks)
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[ks][s]));
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(nqv[e]));
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
    int64_t prb = myns > 0 ? (int64_t)blockIdx.x : nrb;
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
            const char* base_ = (const char*)(x32 + (size_t)(r0 + prb * 32) * dp + (ks0 + pks) * 32);                     \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                               \
                const char* src_ = base_ + (size_t)(8 * j) * dp * 4 + ((j & 1) ? lo1 : lo0);                              \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,                     \
                                                 (__attribute__((address_space(3))) void*)(my + pbuf * 1024 + 8 * j * 32), 16, 0, AUX); \
            }                                                                                                             \
            pbuf = pbuf + 1 == NBUF ? 0 : pbuf + 1;                                                                       \
            if (++pks == myns) {                                                                                          \
                pks = 0;                                                                                                  \
                prb += gridDim.x;                                                                                         \
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
    for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        // the row's norm for the prefilter: issued before this block's slabs, so it is older than every DMA still in
        // flight at the epilogue when the wave has at least NBUF-1 slabs per block
        const int64_t row = r0 + rb * 32 + l31;
        {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(auxv + row),   // padded arrays: readable
                                             (__attribute__((address_space(3))) void*)(St + 4 * NBUF * 1024 + 4 * 3 * 64 * 4 + wu * 64), 4, 0, 0);
            x0 += 1;
            x1 += 1;
            x2 += 1;
        }
#pragma unroll
        for (int ks = 0; ks < GEMM_NSW; ++ks) {
            if (ks < myns) {
                // slab `cur` has landed once at most inflight-1 newer slabs (4 DMA ops each) are outstanding:
                // loads retire in order, so the count is conservative whatever else is in flight
                wait_vmcnt(4 * (inflight - 1) + (x0 >= 5 ? 5 : 0));
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
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[ks][0][t], x0[t], acc, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[ks][1][t], x1[t], acc, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[ks][2][t], x2[t], acc, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[ks][3][t], x3[t], acc, 0, 0, 0);
                cur = cur + 4096 == NBUF * 4096 ? 0 : cur + 4096;
            }
        }
        // C[i = query][j = row]: register r <-> query (r&3) + 8 (r>>2) + 4 h, lane <-> row l31.  Wave o owns
        // registers [4o, 4o+4) = queries 8o + {0..3} + 4h; the other three waves send it their partials.
        // Raw barriers: __syncthreads() carries a vmcnt(0) fence that would drain the prefetch ring.
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
        {
            // all 4 stores are issued whatever the lane holds: rows past r1 land in the padding behind the last
            // tile (np + ROW_TILE rows are allocated), idle slots have their places in every tile.  Tile-major
            // dots: the wave's four stores fill 1 KiB of the row block's contiguous 4 KiB
            float* const tile = dots + (row >> 5) * ts + (row & 31);
#pragma unroll
            for (int e = 0; e < 4; ++e) store_dword_issued(tile + (int64_t)(e + 8 * wu + 4 * h) * sd, mine[e]);
            x0 += 4;
            x1 += 4;
            x2 += 4;
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
template <int NCH, int NSLOT>
__global__ __launch_bounds__(256) void scan_dma_kernel(const float* __restrict__ x32, const float* __restrict__ q32, int64_t dp,
                                                       int64_t r0, int64_t r1, float* __restrict__ dots, PreArgs pre, int rounds,
                                                       int tail_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RING = NSLOT * NCH * 1024;   // bytes per wave
    constexpr int WAVE_LDS = RING + 256;       // + the chunk's 64 norms
    constexpr int K1 = NCH * (NSLOT - 2);      // DMA operations younger than the oldest row of a full ring
    const int tid = threadIdx.x, lane = tid & 63;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* myp = smem + wu * WAVE_LDS;
    const unsigned my0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wu * WAVE_LDS;
    const unsigned ax0 = my0 + RING;
    // lanes past the end of a row never receive DMA data: they must read zeros, not stale bits
    for (int i = lane; i < RING / 16; i += 64) *(f32x4*)(myp + i * 16) = f32x4{0, 0, 0, 0};
    f32x4 qv[NCH];
    bool on[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        on[u] = 4 * (lane + 64 * u) < dp;
        qv[u] = on[u] ? *(const f32x4*)(q32 + 4 * (lane + 64 * u)) : f32x4{0, 0, 0, 0};
    }
    float nq32 = pre.info->nq32, inq32 = pre.info->inq32;
#pragma unroll
    for (int u = 0; u < NCH; ++u) asm volatile("" : "+v"(qv[u]));   // loads complete here, once (see scan_gemm_kernel)
    asm volatile("" : "+v"(nq32), "+v"(inq32));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zero fill
    const float* __restrict__ auxv = pre.metric == AS_METRIC_L2 ? pre.n32 : pre.inorm32;
    const int64_t NW = (int64_t)gridDim.x * 4, gw = (int64_t)blockIdx.x * 4 + wu;
    const int64_t tail0 = r0 + (int64_t)rounds * NW * 64;
    // chunk t of this wave: 64 rows for t < rounds, then its share of the remainder
#define AS_CHUNK(t, base, cnt)                                                           \
    do {                                                                                 \
        if ((t) < rounds) {                                                              \
            base = r0 + ((int64_t)(t) * NW + gw) * 64;                                    \
            cnt = 64;                                                                    \
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
    int full = 0;
    bool first = true;
    for (int t = 0; t <= rounds; ++t) {
        int64_t base;
        int cnt;
        AS_CHUNK(t, base, cnt);
        if (cnt <= 0) continue;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(auxv + base + lane),   // padded: readable
                                         (__attribute__((address_space(3))) void*)(myp + RING), 4, 0, 0);
        marked = first ? 0 : inflight;   // the first chunk has only the norm DMA behind its rows: assume nothing
        first = false;
        float mydot = 0.0f;
        for (int r = 0; r < cnt; ++r) {
            // operations retire in issue order: the oldest row has landed once at most (rows behind it) * NCH
            // (+ 2 for a chunk boundary behind it) operations are outstanding
            if (inflight == NSLOT - 1) {
                if (marked > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K1 + 2) : "memory");
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
            if (NCH == 4)
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(xv[0]), "=&v"(xv[NCH > 1 ? 1 : 0]), "=&v"(xv[NCH > 2 ? 2 : 0]), "=&v"(xv[NCH > 3 ? 3 : 0]) : "v"(a0) : "memory");
            cur = cur + NCH * 1024 == RING ? 0 : cur + NCH * 1024;
            float sacc = 0.0f;
#pragma unroll
            for (int u = 0; u < NCH; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) sacc = fmaf(xv[u][e], qv[u][e], sacc);
            sacc = wave_sum_dpp(sacc);
            mydot = lane == r ? sacc : mydot;
        }
        // the norms are older than every row issued inside this chunk; one of those has been consumed once the
        // chunk is at least as long as the ring
        if (cnt < NSLOT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float aux = lds_read1(ax0 + lane * 4);
        const int64_t row = base + lane;
        if (lane < cnt) {   // cnt >= 1: the store is always issued (the ring's bookkeeping counts it)
            store_dword_issued(dots + row, mydot);
            prefilter_f32(pre, row, mydot, aux, nq32, inq32, full);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef AS_ISSUE_ROW
#undef AS_CHUNK
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
                    key = 1.0 - (c > 0.0 ? c : 0.0);
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

// ------------------------------------------------------------------ scorer key (= -score) of one row
template <typename T>
struct SelArgs {
    const T* dots;
    const float* n32;
    const float* inorm32;
    const double* n64;
    const float* lam32;
    const double* lam64;
    const QInfo* info;
    QInfo* info_w;
    int64_t n, r0, r1, exclude;
    int M, metric;
    double epskey, coef, tau;
    T* pkey;
    int* pidx;
    T* gmin;      // filter path: group minima, candidate buffers (CAND_CAP per slot)
    T* ckey;
    int* cidx;
    int64_t sd;   // dots: elements between slots inside a 32-row tile
    int64_t ts;   // dots: elements between 32-row tiles (32 = plain row order)
};

// The batched workspace keeps dots tile-major -- [32-row tile][slot][32 rows] -- so that the MFMA scan writes one
// contiguous 4 KiB block per row block instead of 32 segments 4 MiB apart; with ts = 32 this is plain row order.
template <typename T>
__device__ __forceinline__ T dot_at(const SelArgs<T>& a, int64_t row) {
    return a.dots[(row >> 5) * a.ts + (row & 31)];
}

template <typename T>
__device__ __forceinline__ void sel_slot(SelArgs<T>& a) {
    const int z = blockIdx.z;
    a.dots += (int64_t)z * a.sd;
    a.info += z;
    a.info_w += z;
    a.gmin += (int64_t)z * CAND_CAP;
    a.ckey += (int64_t)z * CAND_CAP;
    a.cidx += (int64_t)z * CAND_CAP;
}

struct ScoreCtx {
    double nq, tau, lq;
    float tau32, lq32, inq32;
};
__device__ __forceinline__ ScoreCtx load_ctx(const QInfo* info, double tau) {
    ScoreCtx c;
    c.nq = info->nq;
    c.tau = tau;
    c.lq = info->lambda_q;
    c.tau32 = (float)c.tau;
    c.lq32 = (float)c.lq;
    c.inq32 = info->inq32;
    return c;
}
template <typename T>
__device__ __forceinline__ T score_key(const SelArgs<T>& a, const ScoreCtx& c, int64_t row);
template <>
__device__ __forceinline__ float score_key<float>(const SelArgs<float>& a, const ScoreCtx& c, int64_t row) {
    const float cs = dot_at(a, row) * a.inorm32[row] * c.inq32;
    const float term = 1.0f / (1.0f + fabsf(c.lq32 - a.lam32[row]));
    return -(c.tau32 * cs + (1.0f - c.tau32) * term);
}
template <>
__device__ __forceinline__ double score_key<double>(const SelArgs<double>& a, const ScoreCtx& c, int64_t row) {
    const double den = sqrt(a.n64[row] * c.nq);
    const double cs = den > 0.0 ? dot_at(a, row) / den : 0.0;
    return -(c.tau * cs + (1.0 - c.tau) / (1.0 + fabs(c.lq - a.lam64[row])));
}

template <typename T>
__device__ __forceinline__ T knn_key(const SelArgs<T>& a, int64_t row, T dot, double nq);
template <>
__device__ __forceinline__ float knn_key<float>(const SelArgs<float>& a, int64_t row, float dot, double) {
    if (a.metric == AS_METRIC_L2) return fmaf(-2.0f, dot, a.n32[row] + a.info->nq32);
    return 1.0f - fmaxf(0.0f, dot * a.inorm32[row] * a.info->inq32);
}
template <>
__device__ __forceinline__ double knn_key<double>(const SelArgs<double>& a, int64_t row, double dot, double nq) {
    if (a.metric == AS_METRIC_L2) return a.n64[row] + nq - 2.0 * dot;
    const double den = sqrt(a.n64[row] * nq);
    const double c = den > 0.0 ? dot / den : 0.0;
    return 1.0 - (c > 0.0 ? c : 0.0);
}

// ------------------------------------------------------------------ filter path
// KEY = 0: the scorer key (-score) of every scanned row.  KEY = 1: the k-NN key of the rows inside the eps
// bound, +inf elsewhere -- the second chance of a query whose neighbourhood overflowed the scan's candidate
// buffer: the dots are still there, so the k nearest are found by threshold without a second scan.
template <typename T, int KEY>
__device__ __forceinline__ T sel_key(const SelArgs<T>& a, const ScoreCtx& c, int64_t row) {
    if (KEY == 0) return score_key<T>(a, c, row);
    if (row >= a.n || row == a.exclude) return key_traits<T>::inf();
    const T key = knn_key<T>(a, row, dot_at(a, row), c.nq);
    const double ni = sizeof(T) == 4 ? (double)a.n32[row] : a.n64[row];
    const double bound = a.metric == AS_METRIC_L2 ? a.epskey + a.coef * (ni + c.nq) : a.epskey + a.coef;
    return (double)key <= bound * 1.000001 ? key : key_traits<T>::inf();
}

// (1) per-group minimum of the key; one wave per group of G rows
template <typename T, int KEY>
__global__ __launch_bounds__(256) void score_gmin_kernel(SelArgs<T> a, int64_t G, int ngroups) {
    sel_slot(a);
    T* __restrict__ gmin = a.gmin;
    const int lane = lane_id();
    const int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (KEY == 1 && g == 0 && lane == 0) reset_query_state(a.info_w);   // the search restarts behind the scan
    if (g >= ngroups) return;
    const ScoreCtx c = load_ctx(a.info, a.tau);
    const int64_t lo = a.r0 + g * G;
    const int64_t hi = lo + G < a.r1 ? lo + G : a.r1;
    T m = key_traits<T>::inf();
    for (int64_t row = lo + lane; row < hi; row += 64) {
        const T k = sel_key<T, KEY>(a, c, row);
        m = k < m ? k : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T other = __shfl_xor(m, o, 64);
        m = other < m ? other : m;
    }
    if (lane == 0) gmin[g] = m;
}

// k-th smallest (0-based) of up to 4 x 1024 ordered-bit values held 4 per thread; whole block
// (1024 threads) calls it.  PASSES x 8-bit radix select with an LDS histogram.
template <typename U, int PASSES>
__device__ __forceinline__ U block_kth(const U (&v)[4], const bool (&have)[4], int kth, unsigned int* hist, U* s_prefix,
                                       int* s_rank) {
    const int tid = threadIdx.x;
    if (tid == 0) {
        *s_prefix = 0;
        *s_rank = kth;
    }
    __syncthreads();
    for (int pass = 0; pass < PASSES; ++pass) {
        const int shift = 8 * (PASSES - 1 - pass);
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const U prefix = *s_prefix;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!have[q]) continue;
            if (pass > 0 && (v[q] >> (shift + 8)) != prefix) continue;
            atomicAdd(&hist[(unsigned int)((v[q] >> shift) & (U)255)], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            // wave 0: 4 bins per lane, inclusive scan over lanes, the lane whose range holds the rank finishes
            const int rank = *s_rank;
            unsigned int h[4];
            unsigned int tot = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h[j] = hist[4 * tid + j];
                tot += h[j];
            }
            unsigned int incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int t = __shfl_up(incl, o, 64);
                if (tid >= o) incl += t;
            }
            const int excl = (int)(incl - tot);
            if (rank >= excl && rank < (int)incl) {
                int run = excl, b = 4 * tid;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (run + (int)h[j] > rank) break;
                    run += (int)h[j];
                    b += 1;
                }
                *s_rank = rank - run;
                *s_prefix = (prefix << 8) | (U)b;
            }
        }
        __syncthreads();
    }
    return *s_prefix;
}

// (2) threshold = M-th smallest group minimum (radix select over the ordered bit pattern).
// At least M rows have key <= threshold, so the M best rows all pass the filter.
template <typename T, typename U, int PASSES>
__global__ __launch_bounds__(1024) void pick_thr_kernel(const T* __restrict__ gmin, int ng, int M, QInfo* info) {
    __shared__ unsigned int hist[256];
    __shared__ U s_prefix;
    __shared__ int s_rank;
    gmin += (int64_t)blockIdx.z * CAND_CAP;
    info += blockIdx.z;
    const int tid = threadIdx.x;
    if (ng <= M) {
        if (tid == 0) {
            if (sizeof(T) == 4) info->thr32 = key_traits<float>::inf();
            else info->thr64 = key_traits<double>::inf();
        }
        return;
    }
    U v[4];
    bool have[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = tid + q * 1024;
        have[q] = i < ng;
        v[q] = have[q] ? ord_bits(gmin[i]) : (U)0;
    }
    const U kth = block_kth<U, PASSES>(v, have, M - 1, hist, &s_prefix, &s_rank);
    if (tid == 0) {
        const T thr = from_ord(kth);
        if (sizeof(T) == 4) info->thr32 = (float)thr;
        else info->thr64 = (double)thr;
    }
}

// (3) append every row whose key <= threshold
template <typename T, int KEY>
__global__ __launch_bounds__(256) void score_filter_kernel(SelArgs<T> a) {
    sel_slot(a);
    T* __restrict__ ckey = a.ckey;
    int* __restrict__ cidx = a.cidx;
    int* counter = KEY ? &a.info_w->knn_cnt : &a.info_w->sc_cnt;
    const ScoreCtx c = load_ctx(a.info, a.tau);
    const T thr = sizeof(T) == 4 ? (T)a.info->thr32 : (T)a.info->thr64;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool full = false;   // mass ties at the threshold: the counter only has to exceed CAND_CAP
    for (int64_t row = a.r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < a.r1 && !full; row += stride) {
        const T k = sel_key<T, KEY>(a, c, row);
        if (k <= thr && (KEY == 0 || k < key_traits<T>::inf())) {
            const int slot = atomicAdd(counter, 1);
            if (slot < CAND_CAP) {
                ckey[slot] = k;
                cidx[slot] = (int)row;
            } else {
                full = true;
            }
        }
    }
}

// ------------------------------------------------------------------ exact global selection (overflow fallback for wide lists)
// Radix select over the composite (ordered key bits, row index) of ALL scanned rows: KP passes
// over the key bytes, then 4 over the index bytes among the rows tied at the M-th key.  The
// filter that follows appends exactly the M smallest (key, idx) rows -- no tie can overflow it.
struct RSel {
    unsigned long long kprefix;  // decided key bits (ordered), complete after the key passes
    unsigned int iprefix;        // decided index bits
    int rank;                    // remaining 0-based rank inside the current bucket
    int take_all;                // fewer than M rows exist
    unsigned int hist[256];
};

template <typename T>
__global__ __launch_bounds__(256) void rsel_hist_kernel(SelArgs<T> a, int pass, int KP, RSel* rs) {
    typedef typename ord_of<T>::type U;
    __shared__ unsigned int sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    const ScoreCtx c = load_ctx(a.info, a.tau);
    const unsigned long long kprefix = rs->kprefix;
    const unsigned int iprefix = rs->iprefix;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t row = a.r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < a.r1; row += stride) {
        const U kb = ord_bits(score_key<T>(a, c, row));
        unsigned int digit;
        if (pass < KP) {
            const int shift = 8 * (KP - 1 - pass);
            if (pass > 0 && (unsigned long long)(kb >> (shift + 8)) != kprefix) continue;
            digit = (unsigned int)((kb >> shift) & (U)255);
        } else {
            if ((unsigned long long)kb != kprefix) continue;
            const int ip = pass - KP, shift = 8 * (3 - ip);
            const unsigned int ib = (unsigned int)row;
            if (ip > 0 && (ib >> (shift + 8)) != iprefix) continue;
            digit = (ib >> shift) & 255u;
        }
        atomicAdd(&sh[digit], 1u);
    }
    __syncthreads();
    if (sh[threadIdx.x]) atomicAdd(&rs->hist[threadIdx.x], sh[threadIdx.x]);
}

__global__ void rsel_pick_kernel(int pass, int KP, int M, RSel* rs) {
    if (threadIdx.x != 0) return;
    if (pass == 0) {
        long long tot = 0;
        for (int b = 0; b < 256; ++b) tot += rs->hist[b];
        rs->take_all = tot <= M ? 1 : 0;
        rs->rank = M - 1;
        rs->kprefix = 0;
        rs->iprefix = 0;
    }
    if (!rs->take_all) {
        int run = 0, b = 0;
        for (; b < 256; ++b) {
            if (run + (int)rs->hist[b] > rs->rank) break;
            run += (int)rs->hist[b];
        }
        rs->rank -= run;
        if (pass < KP) rs->kprefix = (rs->kprefix << 8) | (unsigned long long)b;
        else rs->iprefix = (rs->iprefix << 8) | (unsigned int)b;
    }
    for (int b = 0; b < 256; ++b) rs->hist[b] = 0;
}

template <typename T>
__global__ __launch_bounds__(256) void rsel_filter_kernel(SelArgs<T> a, const RSel* rs) {
    typedef typename ord_of<T>::type U;
    const ScoreCtx c = load_ctx(a.info, a.tau);
    const unsigned long long kprefix = rs->kprefix;
    const unsigned int iprefix = rs->iprefix;
    const bool all = rs->take_all != 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t row = a.r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < a.r1; row += stride) {
        const T k = score_key<T>(a, c, row);
        const unsigned long long kb = (unsigned long long)ord_bits(k);
        if (all || kb < kprefix || (kb == kprefix && (unsigned int)row <= iprefix)) {
            const int slot = atomicAdd(&a.info_w->sc_cnt, 1);
            if (slot < CAND_CAP) {
                a.ckey[slot] = k;
                a.cidx[slot] = (int)row;
            }
        }
    }
}

// ------------------------------------------------------------------ list path (overflow fallback): wavefront-shuffle partial lists
template <typename T>
__global__ __launch_bounds__(256) void knn_partial_kernel(SelArgs<T> a) {
    const int lane = lane_id();
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    const double nq = a.info->nq;
    WaveList<T> lst;
    lst.init();
    int npass = 0;
    for (int64_t base = a.r0 + gw * 64; base < a.r1; base += nw * 64) {
        const int64_t row = base + lane;
        bool valid = row < a.r1 && row < a.n && row != a.exclude;
        T key = key_traits<T>::inf();
        if (valid) {
            key = knn_key<T>(a, row, dot_at(a, row), nq);
            const double ni = sizeof(T) == 4 ? (double)a.n32[row] : a.n64[row];
            const double bound = a.metric == AS_METRIC_L2 ? a.epskey + a.coef * (ni + nq) : a.epskey + a.coef;
            valid = (double)key <= bound * 1.000001;
        }
        npass += valid ? 1 : 0;
        lst.offer(a.M, key, (int)row, valid);
    }
    npass = wave_sum(npass);
    if (lane == 0 && npass) atomicAdd(&a.info_w->knn_total, npass);
    a.pkey[gw * 64 + lane] = lst.key;
    a.pidx[gw * 64 + lane] = lst.idx;
}

template <typename T>
__global__ __launch_bounds__(256) void score_partial_kernel(SelArgs<T> a) {
    const int lane = lane_id();
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    const ScoreCtx c = load_ctx(a.info, a.tau);
    WaveList<T> lst;
    lst.init();
    for (int64_t base = a.r0 + gw * 64; base < a.r1; base += nw * 64) {
        const int64_t row = base + lane;
        const bool valid = row < a.r1 && row < a.n;
        const T key = valid ? score_key<T>(a, c, row) : key_traits<T>::inf();
        lst.offer(a.M, key, (int)row, valid);
    }
    a.pkey[gw * 64 + lane] = lst.key;
    a.pidx[gw * 64 + lane] = lst.idx;
}

// merge nlists partial lists (64 slots each) down to one sorted list in LDS (fk, fi)
template <typename T>
__device__ __forceinline__ void merge_partials(const T* pkey, const int* pidx, int nlists, int M, T* wk, int* wi, T* fk,
                                               int* fi, int* fcount) {
    const int lane = lane_id(), w = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    WaveList<T> lst;
    lst.init();
    for (int p = w; p < nlists; p += nwv) {
        const T k = pkey[(size_t)p * 64 + lane];
        const int i = pidx[(size_t)p * 64 + lane];
        lst.offer(M, k, i, i != 0x7fffffff);
    }
    wk[w * 64 + lane] = lst.key;
    wi[w * 64 + lane] = lst.idx;
    __syncthreads();
    if (w == 0) {
        WaveList<T> fin;
        fin.init();
        for (int p = 0; p < nwv; ++p) {
            const T k = wk[p * 64 + lane];
            const int i = wi[p * 64 + lane];
            fin.offer(M, k, i, i != 0x7fffffff);
        }
        fk[lane] = fin.key;
        fi[lane] = fin.idx;
        const int c = __popcll(__ballot(lane < M && fin.idx != 0x7fffffff));
        if (lane == 0) *fcount = c;
    }
    __syncthreads();
}

constexpr int PRUNE_CAP = 2048;  // compact list of the radix-pruned candidates

// rank-select the M smallest (key, idx) of the C buffered candidates into (fk, fi), sorted.
// Large C (dense neighbourhoods) is first pruned to the candidates at or below the M-th
// smallest key by a radix select, so the quadratic ranking only ever sees ~M entries.
template <typename T>
__device__ __forceinline__ void select_candidates(const T* ckey, const int* cidx, int C, int M, T* sk, int* si, T* pk, int* pi,
                                                  T* fk, int* fi, int* fcount) {
    typedef typename ord_of<T>::type U;
    __shared__ unsigned int hist[256];
    __shared__ U s_prefix;
    __shared__ int s_rank, s_cnt;
    for (int t = threadIdx.x; t < C; t += blockDim.x) {
        sk[t] = ckey[t];
        si[t] = cidx[t];
    }
    if (threadIdx.x == 0) {
        *fcount = C < M ? C : M;
        s_cnt = 0;
    }
    __syncthreads();
    const T* rk = sk;
    const int* ri = si;
    int R = C;
    if (C > 512) {
        U v[4];
        bool have[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = threadIdx.x + q * 1024;
            have[q] = i < C;
            v[q] = have[q] ? ord_bits(sk[i]) : (U)0;
        }
        const U kth = block_kth<U, ord_of<T>::passes>(v, have, M - 1, hist, &s_prefix, &s_rank);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (have[q] && v[q] <= kth) {
                const int slot = atomicAdd(&s_cnt, 1);
                if (slot < PRUNE_CAP) {
                    const int i = threadIdx.x + q * 1024;
                    pk[slot] = sk[i];
                    pi[slot] = si[i];
                }
            }
        }
        __syncthreads();
        if (s_cnt <= PRUNE_CAP) {  // otherwise (mass ties at the threshold) rank the full set
            rk = pk;
            ri = pi;
            R = s_cnt;
        }
    }
    for (int t = threadIdx.x; t < R; t += blockDim.x) {
        const T k = rk[t];
        const int i = ri[t];
        int rank = 0;
        for (int s = 0; s < R; ++s) rank += lex_less<T>(rk[s], ri[s], k, i) ? 1 : 0;
        if (rank < M) {
            fk[rank] = k;
            fi[rank] = i;
        }
    }
    __syncthreads();
}

// exact fp64 (squared distance, dot) of the query against up to 64 candidate rows at once:
// 16 lanes per candidate (thread t -> candidate t>>4), block of 1024 threads
__device__ __forceinline__ void exact_eval_all(const float* x32, const double* x64, const double* q64, int64_t d, int64_t dp,
                                               const int* fi, int Mp, double* o_sq, double* o_dot) {
    const int c = threadIdx.x >> 4, sub = threadIdx.x & 15;
    double s = 0.0, g = 0.0;
    if (c < Mp) {
        const int64_t j = fi[c];
        if (x64) {
            const double* pj = x64 + j * d;
            for (int64_t e = sub; e < d; e += 16) {
                const double a = q64[e], b = pj[e], t = a - b;
                s += t * t;
                g += a * b;
            }
        } else {
            const float* pj = x32 + j * dp;  // rows and the query are zero padded to dp
            for (int64_t e = 4 * sub; e < dp; e += 64) {
                const f32x4 v = *(const f32x4*)(pj + e);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double a = q64[e + u], b = (double)v[u], t = a - b;
                    s += t * t;
                    g += a * b;
                }
            }
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        g += __shfl_xor(g, o, 64);
    }
    if (c < Mp && sub == 0) {
        o_sq[c] = s;
        o_dot[c] = g;
    }
}

struct FinishArgs {
    const float* x32;
    const double* x64;
    const double* n64;
    const double* q64;
    const double* deg;  // may be null (build fallback)
    const double* ny;
    const double* lam64;
    QInfo* info;
    int64_t n, d, dp, k, topk, nrows;
    int nlists, M, metric, kernel;
    double epskey, coef, nmax, sigma, p, tau0, tau;
    as_knn_rec* recs;
    as_hit_rec* hits;
    HostOut* hout;      // non-null: also publish the final answer (single-GPU fused tail)
    int64_t seq;
    const void* ck;     // candidate keys / indices (filter buffers or wave lists)
    const int* ci;
    SlotStride ss;
    int fuse;           // knn: compute lambda_q in the same launch; score: publish to hout
    int from_list;      // candidates come from the wavefront lists instead of the filter buffer
    int thresholded;    // the buffer holds the rows under a selection threshold, not every row inside eps
    // build-fallback outputs (row-list form); null for searches
    int32_t* o_idx;
    double* o_key;
    double* o_dist;
    double* o_gy;
    int32_t* o_cnt;
};

// SPEC S10 given the selected neighbours in ascending index order in LDS; one wave, lane t
// owns neighbour t, sums are fixed-order butterflies (deterministic).
__device__ __forceinline__ void lambda_from_sorted(int cnt, const double* s_dist, const double* s_gy, const double* s_deg,
                                                   const double* s_ny, int metric, int kernel, double sigma, double p,
                                                   double tau0, QInfo* info) {
    const int lane = lane_id();
    const double nq = info->nq;
    const double nyq = metric == AS_METRIC_L2 ? nq : (nq > 0.0 ? 1.0 : 0.0);
    const bool on = lane < cnt;
    const double at = on ? edge_weight(s_dist[lane], sigma, p, kernel) : 0.0;
    const double degq = wave_sum(at);
    double lam = 0.0;
    if (cnt > 0 && nyq > 0.0 && degq > 0.0) {
        double ev = 0.0;
        if (on) {
            const double dj = s_deg[lane] + at;
            const double sdd = sqrt(degq * dj);
            const double v = at * (nyq / degq + s_ny[lane] / dj - 2.0 * s_gy[lane] / sdd);
            ev = v > 0.0 ? v : 0.0;
        }
        const double S = wave_sum(ev);
        const double Eq = 0.5 * S / nyq;
        double Gq = 0.0;
        if (S > 0.0) {
            const double r = ev / S;
            Gq = wave_sum(r * r);
            Gq = Gq < 0.0 ? 0.0 : (Gq > 1.0 ? 1.0 : Gq);
        }
        lam = tau0 * (Eq / (Eq + tau0)) + (1.0 - tau0) * Gq;
    }
    if (lane == 0) {
        info->lambda_q = lam;
        info->status = lam == 0.0 ? AS_EZEROLAMBDA : AS_OK;
    }
}

// k-NN of the query: candidates -> M smallest fp32 keys -> fp64 re-evaluation -> (key64, idx)
// order, eps, k cap, a-posteriori exactness check; optionally lambda_q in the same launch.
template <typename T>
__global__ __launch_bounds__(1024) void knn_finish_kernel(FinishArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int z = blockIdx.z;
    const T* ckey = (const T*)a.ck + (int64_t)z * CAND_CAP;
    const int* cidx = a.ci + (int64_t)z * CAND_CAP;
    a.info += z;
    a.q64 += (int64_t)z * a.ss.q;
    if (a.recs) a.recs += (int64_t)z * a.ss.knn;
    T* sk = (T*)smem;                       // CAND_CAP keys (filter) or 16x64 wave lists
    int* si = (int*)(sk + CAND_CAP);
    T* pk = (T*)(si + CAND_CAP);            // PRUNE_CAP pruned candidates
    int* pi = (int*)(pk + PRUNE_CAP);
    __shared__ T fk[64];
    __shared__ int fi[64];
    __shared__ double ek[64], ed[64], eg[64], sk2[64];
    __shared__ double l_dist[64], l_gy[64], l_deg[64], l_ny[64];
    __shared__ int fcount;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    int total;
    if (a.from_list) {
        merge_partials<T>(ckey, cidx, a.nlists, a.M, sk, si, fk, fi, &fcount);
        total = a.info->knn_total;
    } else {
        const int raw = a.info->knn_cnt;
        if (raw > CAND_CAP && threadIdx.x == 0) a.info->overflow |= 1;
        total = raw < CAND_CAP ? raw : CAND_CAP;
        select_candidates<T>(ckey, cidx, total, a.M, sk, si, pk, pi, fk, fi, &fcount);
    }
    const int Mp = fcount;
    const double nq = a.info->nq;
    exact_eval_all(a.x32, a.x64, a.q64, a.d, a.dp, fi, Mp, ek, eg);
    __syncthreads();
    if (threadIdx.x < Mp) {
        const int t = threadIdx.x;
        const double sq = ek[t], dot = eg[t];
        if (a.metric == AS_METRIC_L2) {
            ed[t] = sqrt(sq);
        } else {
            const double den = sqrt(nq * a.n64[fi[t]]);
            const double c = den > 0.0 ? dot / den : 0.0;
            const double dd = 1.0 - (c > 0.0 ? c : 0.0);
            ek[t] = dd;
            ed[t] = dd;
            eg[t] = c;
        }
    }
    __syncthreads();
    if (w != 0) return;
    // rank by (key64, idx): one candidate per lane
    const bool have = lane < Mp;
    const double myk = have ? ek[lane] : 0.0;
    const int myi = have ? fi[lane] : 0x7fffffff;
    int rank = 0;
    for (int s = 0; s < Mp; ++s) rank += lex_less<double>(ek[s], fi[s], myk, myi) ? 1 : 0;
    if (have) sk2[rank] = myk;
    const bool pass = have && myk <= a.epskey;
    const int npass = __popcll(__ballot(pass));
    const int cnt = npass < a.k ? npass : (int)a.k;
    const bool sel = pass && rank < a.k;
    if (a.recs) {
        for (int64_t t = lane; t < a.k; t += 64) {
            as_knn_rec r;
            r.idx = -1;
            r.key = key_traits<double>::inf();
            r.dist = 0; r.gy = 0; r.deg = 0; r.ny = 0;
            a.recs[t] = r;
        }
    }
    if (a.o_idx)
        for (int64_t t = lane; t < a.k; t += 64) a.o_idx[t] = -1;
    const double mydeg = sel && a.deg ? a.deg[myi] : 0.0;
    const double myny = sel && a.ny ? a.ny[myi] : 0.0;
    if (sel) {
        if (a.recs) {
            as_knn_rec r;
            r.idx = myi;
            r.key = myk;
            r.dist = ed[lane];
            r.gy = eg[lane];
            r.deg = mydeg;
            r.ny = myny;
            a.recs[rank] = r;
        }
        if (a.o_idx) {
            a.o_idx[rank] = myi;
            a.o_key[rank] = myk;
            a.o_dist[rank] = ed[lane];
            a.o_gy[rank] = eg[lane];
        }
    }
    if (a.fuse) {
        // ascending-index order of the selected neighbours
        int irank = 0;
        for (int s = 0; s < 64; ++s) {
            const int oi = bcast_lane(myi, s);
            const bool osel = (__ballot(sel) >> s) & 1ull;
            irank += (osel && oi < myi) ? 1 : 0;
        }
        if (sel) {
            l_dist[irank] = ed[lane];
            l_gy[irank] = eg[lane];
            l_deg[irank] = mydeg;
            l_ny[irank] = myny;
        }
    }
    AS_LDS_FENCE();
    if (lane == 0) {
        if (a.o_cnt) *a.o_cnt = cnt;
        int bad = 0;
        a.info->knn_total = total;
        if ((total > a.M || a.thresholded) && Mp > 0) {
            const double B = npass >= a.k ? sk2[a.k - 1] : a.epskey;
            // dropped items' norms are unknown: bound them by the largest norm in the space
            const double e = a.metric == AS_METRIC_L2 ? a.coef * (a.nmax + nq) : a.coef;
            const double Tm = (double)fk[Mp - 1];
            bad = !(Tm - e > B);
        }
        a.info->knn_inexact = bad;
    }
    if (a.fuse) lambda_from_sorted(cnt, l_dist, l_gy, l_deg, l_ny, a.metric, a.kernel, a.sigma, a.p, a.tau0, a.info);
}

// SPEC S10 from m candidate records (this shard's, or all shards' all-gathered)
__global__ __launch_bounds__(64) void q_lambda_kernel(const as_knn_rec* __restrict__ recs, int64_t m, int64_t k, int metric,
                                                      int kernel, double sigma, double p, double tau0, QInfo* info) {
    __shared__ double r_key[REC_CAP];
    __shared__ int r_idx[REC_CAP];
    __shared__ double l_dist[64], l_gy[64], l_deg[64], l_ny[64];
    __shared__ int l_pos[64];
    const int lane = lane_id();
    const int mm = (int)(m < REC_CAP ? m : REC_CAP);
    for (int t = lane; t < mm; t += 64) {
        const bool valid = recs[t].idx >= 0;
        r_key[t] = valid ? recs[t].key : key_traits<double>::inf();
        r_idx[t] = valid ? (int)recs[t].idx : 0x7fffffff;
    }
    AS_LDS_FENCE();
    // rank every record by (key, idx); the k best valid ones are the neighbours
    int cnt_l = 0;
    for (int t = lane; t < mm; t += 64) {
        if (r_idx[t] == 0x7fffffff) continue;
        int rank = 0;
        for (int s = 0; s < mm; ++s) rank += lex_less<double>(r_key[s], r_idx[s], r_key[t], r_idx[t]) ? 1 : 0;
        if (rank < k && rank < 64) {
            l_pos[rank] = t;
            cnt_l += 1;
        }
    }
    const int cnt = wave_sum(cnt_l);
    AS_LDS_FENCE();
    // ascending-index order
    if (lane < cnt) {
        const int t = l_pos[lane];
        int irank = 0;
        for (int s = 0; s < cnt; ++s) irank += r_idx[l_pos[s]] < r_idx[t] ? 1 : 0;
        l_dist[irank] = recs[t].dist;
        l_gy[irank] = recs[t].gy;
        l_deg[irank] = recs[t].deg;
        l_ny[irank] = recs[t].ny;
    }
    AS_LDS_FENCE();
    lambda_from_sorted(cnt, l_dist, l_gy, l_deg, l_ny, metric, kernel, sigma, p, tau0, info);
}

__device__ __forceinline__ void publish(HostOut* out, int64_t seq) {
    __threadfence_system();
    out->seq = seq;
}

template <typename T>
__global__ __launch_bounds__(1024) void score_finish_kernel(FinishArgs a, double coef_s) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int z = blockIdx.z;
    const T* ckey = (const T*)a.ck + (int64_t)z * CAND_CAP;
    const int* cidx = a.ci + (int64_t)z * CAND_CAP;
    a.info += z;
    a.q64 += (int64_t)z * a.ss.q;
    if (a.hits) a.hits += (int64_t)z * a.ss.hits;
    if (a.hout) a.hout += z;
    T* sk = (T*)smem;
    int* si = (int*)(sk + CAND_CAP);
    T* pk = (T*)(si + CAND_CAP);
    int* pi = (int*)(pk + PRUNE_CAP);
    double* es = (double*)(pi + PRUNE_CAP);   // MS_MAX exact scores
    double* sk2 = es + MS_MAX;                 // MS_MAX: scratch, then scores in rank order
    T* fk = (T*)(sk2 + MS_MAX);               // MS_MAX best fp32 keys, sorted
    int* fi = (int*)(fk + MS_MAX);
    __shared__ int fcount;
    int total;
    if (a.from_list) {
        merge_partials<T>(ckey, cidx, a.nlists, a.M, sk, si, fk, fi, &fcount);
        total = (int)(a.nrows < 0x7fffffff ? a.nrows : 0x7fffffff);
    } else {
        const int raw = a.info->sc_cnt;
        if (raw > CAND_CAP && threadIdx.x == 0) a.info->overflow |= 2;
        total = raw < CAND_CAP ? raw : CAND_CAP;
        select_candidates<T>(ckey, cidx, total, a.M, sk, si, pk, pi, fk, fi, &fcount);
    }
    const int Mp = fcount;
    const double nq = a.info->nq, tau = a.tau, lq = a.info->lambda_q;
    for (int base = 0; base < Mp; base += 64)   // 64 candidates per round, 16 lanes each
        exact_eval_all(a.x32, a.x64, a.q64, a.d, a.dp, fi + base, Mp - base < 64 ? Mp - base : 64, sk2 + base, es + base);
    __syncthreads();
    for (int t = threadIdx.x; t < Mp; t += blockDim.x) {
        const int j = fi[t];
        const double den = sqrt(a.n64[j] * nq);
        const double c = den > 0.0 ? es[t] / den : 0.0;
        es[t] = tau * c + (1.0 - tau) / (1.0 + fabs(lq - a.lam64[j]));
    }
    __syncthreads();
    const int64_t want = a.topk < a.nrows ? a.topk : a.nrows;
    const int nhit = (int)(Mp < want ? Mp : want);
    if (a.hits) {
        for (int64_t t = threadIdx.x; t < a.topk; t += blockDim.x) {
            as_hit_rec r;
            r.idx = -1;
            r.score = -key_traits<double>::inf();
            a.hits[t] = r;
        }
    }
    __syncthreads();
    // rank by (score desc, idx asc)
    for (int t = threadIdx.x; t < Mp; t += blockDim.x) {
        const double myk = -es[t];
        const int myi = fi[t];
        int rank = 0;
        for (int s2 = 0; s2 < Mp; ++s2) rank += lex_less<double>(-es[s2], fi[s2], myk, myi) ? 1 : 0;
        sk2[rank] = es[t];
        if (a.hits && rank < a.topk) {
            as_hit_rec r;
            r.idx = myi;
            r.score = es[t];
            a.hits[rank] = r;
        }
        if (a.fuse && a.hout && rank < nhit) {
            a.hout->idx[rank] = myi;
            a.hout->score[rank] = es[t];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        int bad = 0;
        if (a.nrows > a.M && Mp > 0) {
            // every row outside the list has score32 <= -fk[Mp-1]; its exact score <= that + coef_s
            const double kth = Mp >= want ? sk2[want - 1] : -key_traits<double>::inf();
            const double ub = -(double)fk[Mp - 1] + coef_s;
            bad = !(ub < kth);
        }
        a.info->score_inexact = bad;
        a.info->nhit = nhit;
        if (a.hits) {
            // trailing flag record: every rank sees every rank's flags after the all-gather
            as_hit_rec r;
            r.idx = -2;
            r.score = (double)((a.info->knn_inexact ? 1 : 0) | (bad ? 2 : 0) | ((a.info->overflow & 1) ? 4 : 0) | ((a.info->overflow & 2) ? 8 : 0));
            a.hits[a.topk] = r;
        }
        if (a.fuse && a.hout) {
            a.hout->len = nhit;
            a.hout->lambda_q = a.info->lambda_q;
            a.hout->status = a.info->status;
            a.hout->knn_inexact = a.info->knn_inexact;
            a.hout->score_inexact = bad;
            a.hout->overflow = a.info->overflow;
            publish(a.hout, a.seq);
        }
    }
}

// merge m hit records (own or all-gathered) -> final topk, written to pinned host memory
__global__ __launch_bounds__(1024) void hits_final_kernel(const as_hit_rec* __restrict__ hits, int64_t m, int64_t topk,
                                                          const QInfo* info, HostOut* out, int64_t seq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* r_key = (double*)smem;           // mm
    int* r_idx = (int*)(r_key + HIT_CAP);
    __shared__ int s_flags, s_cnt;
    const int mm = (int)(m < HIT_CAP ? m : HIT_CAP);
    if (threadIdx.x == 0) {
        s_flags = 0;
        s_cnt = 0;
    }
    __syncthreads();
    int flags_l = 0;
    for (int t = threadIdx.x; t < mm; t += blockDim.x) {
        const bool valid = hits[t].idx >= 0;
        if (hits[t].idx == -2) flags_l |= (int)hits[t].score;
        r_key[t] = valid ? -hits[t].score : key_traits<double>::inf();
        r_idx[t] = valid ? (int)hits[t].idx : 0x7fffffff;
    }
    if (flags_l) atomicOr(&s_flags, flags_l);
    __syncthreads();
    int cnt_l = 0;
    for (int t = threadIdx.x; t < mm; t += blockDim.x) {
        if (r_idx[t] == 0x7fffffff) continue;
        int rank = 0;
        for (int s2 = 0; s2 < mm; ++s2) rank += lex_less<double>(r_key[s2], r_idx[s2], r_key[t], r_idx[t]) ? 1 : 0;
        if (rank < topk && rank < MAX_TOPK) {
            out->idx[rank] = r_idx[t];
            out->score[rank] = -r_key[t];
            cnt_l += 1;
        }
    }
    if (cnt_l) atomicAdd(&s_cnt, cnt_l);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int fl = s_flags;
        out->len = s_cnt;
        out->lambda_q = info->lambda_q;
        out->status = info->status;
        out->knn_inexact = (info->knn_inexact || (fl & 1)) ? 1 : 0;
        out->score_inexact = (info->score_inexact || (fl & 2)) ? 1 : 0;
        out->overflow = (info->overflow & 3) | ((fl & 4) ? 1 : 0) | ((fl & 8) ? 2 : 0);
        publish(out, seq);
    }
}

// ------------------------------------------------------------------ host side
static int score_width(int64_t topk) {
    if (topk > MAX_TOPK) return -1;
    const int64_t need = topk + 8;
    if (need <= 32) return 32;
    return (int)((need + 63) / 64 * 64);   // <= MS_MAX
}

static int list_width(int64_t k) {
    const int64_t need = k + 8;
    if (need <= 32) return 32;
    if (need <= 64) return 64;
    return -1;
}

// fp32/fp64 error coefficient of one dot product: (terms in the longest rounding chain + slack) * u.
// Wave-per-row scans sum dp/64 fused terms per lane before a 6-level butterfly; the MFMA pass
// accumulates a quarter of the columns in sequence (two roundings per term, in case the matrix
// core rounds the products) and adds four partials.
static double coef_query(const as_query* q, bool exact) {
    const double u = exact ? 1.1102230246251565e-16 : 5.9604644775390625e-8;
    const int64_t dp = q->sp->dp;
    if (!exact && q->cap > 1 && dp <= 4 * GEMM_NSW * 32) return (double)(2 * (((dp / 32 + 3) / 4) * 32) + 3 + 24) * u;
    return (double)(dp / 64 + 24) * u;
}

static PreArgs make_pre(as_query* q, double eps, int64_t exclude, bool enabled) {
    const as_space* sp = q->sp;
    PreArgs p;
    p.n32 = sp->n32; p.inorm32 = sp->inorm32; p.n64 = sp->n64; p.info = q->info; p.infow = q->info;
    p.ckey = q->ckey_k; p.cidx = q->cidx_k;
    p.metric = sp->opts.metric;
    p.epskey = p.metric == AS_METRIC_L2 ? eps * eps : eps;
    p.coef = coef_query(q, q->exact != 0);
    p.n = sp->n; p.exclude = exclude; p.enabled = enabled ? 1 : 0;
    return p;
}

static as_status launch_scan(as_query* q, const PreArgs& pre) {
    const as_space* sp = q->sp;
    const int64_t rows = q->r1 - q->r0;
    if (rows <= 0) return AS_OK;
    hipStream_t st = q->stream;
    if (q->exact) {
        if (!q->dots64) AS_HIP(hipMalloc(&q->dots64, sizeof(double) * (sp->np + ROW_TILE)));
        const unsigned grid = (unsigned)std::min<int64_t>((rows + 3) / 4, 4096);
        hipLaunchKernelGGL(scan_dots_f64_kernel, dim3(grid), dim3(256), 0, st, sp->x32, sp->x64, q->q64, sp->d, sp->dp, q->r0,
                           q->r1, q->dots64, pre);
    } else {
        const int nch = (int)((sp->dp + 255) / 256);
        if (q->cap > 1 && sp->dp <= 4 * GEMM_NSW * 32) {
            // batched pass, GEMM-shaped: fp32 MFMA, K split over the 4 waves of a block, 2 blocks per CU
#define AS_GSCAN(NB_, DG, AX)                                                                                                \
    do {                                                                                                               \
        const size_t lds = sizeof(float) * ((size_t)4 * (NB_) * 1024 + 4 * 3 * 64 * 4 + 4 * 64);                       \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            AS_HIP(hipFuncSetAttribute((const void*)scan_gemm_kernel<NB_, DG, AX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        const int64_t nrb = (rows + 31) / 32;                                                                          \
        const unsigned grid = (unsigned)std::min<int64_t>(nrb, 2 * q->cus);                                            \
        hipLaunchKernelGGL((scan_gemm_kernel<NB_, DG, AX>), dim3(grid), dim3(256), lds, st, sp->x32, q->q32, sp->dp, q->r0, q->r1, \
                           q->dots32, q->ss.dots, q->ss.dots_ts, pre, q->nb);                                                         \
    } while (0)
            if (q->gemm_variant == 1) AS_GSCAN(3, 0, 2);
            else if (q->gemm_variant == 2) AS_GSCAN(4, 0, 0);
            else if (q->gemm_variant == 16) AS_GSCAN(4, 1, 2);
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
                               sp->dp, q->r0, q->r1, q->dots32 + (int64_t)j0 * q->ss.dots, q->ss.dots, q->ss.dots_ts, pj);             \
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
        if (nch <= 4 && !(q->scan_variant & 4)) {
            const int64_t want = std::max<int64_t>(1, (rows + 63) / 64);                 // blocks that still get >= 16 rows per wave
            // 2 blocks per CU, ring of 5 rows at 768 columns; rings of 4 or 6 rows and 3 blocks per CU measured the same or slower
            const int64_t nblk = std::min<int64_t>(want, 2 * (int64_t)q->cus);
            const int64_t NW = nblk * 4;
            const int rounds = (int)(rows / (NW * 64));
            const int64_t rem = rows - (int64_t)rounds * NW * 64;
            const int tail_rows = (int)((rem + NW - 1) / NW);
#define AS_DSCAN(N, S)                                                                                                 \
    do {                                                                                                               \
        const size_t lds = 4 * ((size_t)(S) * (N) * 1024 + 256);                                                       \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            AS_HIP(hipFuncSetAttribute((const void*)scan_dma_kernel<N, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL((scan_dma_kernel<N, S>), dim3((unsigned)nblk), dim3(256), lds, st, sp->x32, q->q32, sp->dp, q->r0, \
                           q->r1, q->dots32, pre, rounds, tail_rows);                                                  \
    } while (0)
            switch (nch) {
                case 1: AS_DSCAN(1, 8); break;
                case 2: AS_DSCAN(2, 8); break;
                case 3: AS_DSCAN(3, 5); break;
                default: AS_DSCAN(4, 4); break;
            }
#undef AS_DSCAN
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

template <typename T>
static SelArgs<T> make_sel(as_query* q, const T* dots, int M, int64_t exclude) {
    const as_space* sp = q->sp;
    SelArgs<T> a;
    a.dots = dots; a.n32 = sp->n32; a.inorm32 = sp->inorm32; a.n64 = sp->n64; a.lam32 = sp->lam32; a.lam64 = sp->lam64;
    a.info = q->info; a.info_w = q->info; a.n = sp->n; a.r0 = q->r0; a.r1 = q->r1; a.exclude = exclude;
    a.M = M; a.metric = sp->opts.metric;
    a.epskey = 0; a.coef = 0; a.tau = 1.0;
    a.pkey = (T*)q->pkey; a.pidx = q->pidx;
    a.gmin = (T*)q->gmin; a.ckey = (T*)q->ckey_s; a.cidx = q->cidx_s; a.sd = q->ss.dots; a.ts = q->ss.dots_ts;
    return a;
}

static int sel_grid(as_query* q, int* nwaves) {
    const int64_t rows = std::max<int64_t>(q->r1 - q->r0, 1);
    int64_t nw = (rows + 255) / 256;  // >= 256 rows per wave
    nw = std::min<int64_t>(std::max<int64_t>(nw, 4), q->nwaves);
    nw = (nw + 3) / 4 * 4;
    *nwaves = (int)nw;
    return (int)(nw / 4);
}

static FinishArgs make_finish(as_query* q) {
    const as_space* sp = q->sp;
    FinishArgs f;
    memset(&f, 0, sizeof(f));
    f.x32 = sp->x32; f.x64 = sp->x64; f.n64 = sp->n64; f.q64 = q->q64; f.lam64 = sp->lam64;
    f.deg = q->gr ? q->gr->deg : nullptr; f.ny = q->gr ? q->gr->ny : nullptr;
    f.info = q->info; f.n = sp->n; f.d = sp->d; f.dp = sp->dp; f.k = q->k; f.topk = q->topk; f.nrows = q->r1 - q->r0;
    f.metric = sp->opts.metric; f.kernel = sp->opts.kernel; f.nmax = sp->nmax;
    if (q->gr) {
        f.sigma = q->gr->gp.sigma; f.p = q->gr->gp.p; f.tau0 = q->gr->tau0;
    }
    f.from_list = q->robust;
    f.ss = q->ss;
    return f;
}

template <typename T>
static size_t score_lds() {
    return (sizeof(T) + sizeof(int)) * (size_t)(CAND_CAP + PRUNE_CAP + MS_MAX) + 2 * sizeof(double) * MS_MAX;
}

template <typename T>
static size_t finish_lds() {
    return (sizeof(T) + sizeof(int)) * (size_t)(CAND_CAP + PRUNE_CAP);
}

// k-NN candidates of the scanned rows -> records (or row lists for the build fallback)
// Second chance of a neighbourhood that overflowed the scan's candidate buffer (more than CAND_CAP rows inside
// eps): the dots are still in HBM, so the candidates are re-derived by threshold -- the Mk-th smallest group
// minimum of the k-NN key -- exactly as the scorer's filter path does, instead of scanning the items again.
template <typename T, typename U, int PASSES>
static void launch_knn_repair(as_query* q, const T* dots, double eps, int64_t exclude) {
    hipStream_t st = q->stream;
    const int64_t rows = q->r1 - q->r0;
    int64_t G = (rows + CAND_CAP - 1) / CAND_CAP;
    G = std::max<int64_t>(64, (G + 63) / 64 * 64);
    const int ng = (int)((rows + G - 1) / G);
    SelArgs<T> a = make_sel<T>(q, dots, q->Mk, exclude);
    a.epskey = q->sp->opts.metric == AS_METRIC_L2 ? eps * eps : eps;
    a.coef = coef_query(q, sizeof(T) == 8);
    a.ckey = (T*)q->ckey_k;
    a.cidx = q->cidx_k;
    const unsigned nb = (unsigned)q->nb;
    hipLaunchKernelGGL((score_gmin_kernel<T, 1>), dim3((unsigned)((ng + 3) / 4), 1, nb), dim3(256), 0, st, a, G, ng);
    hipLaunchKernelGGL((pick_thr_kernel<T, U, PASSES>), dim3(1, 1, nb), dim3(1024), 0, st, (const T*)q->gmin, ng, q->Mk, q->info);
    const unsigned fg = (unsigned)std::min<int64_t>((rows + 255) / 256, 2048);
    hipLaunchKernelGGL((score_filter_kernel<T, 1>), dim3(fg, 1, nb), dim3(256), 0, st, a);
}

static as_status knn_repair(as_query* q, double eps, int64_t exclude) {
    if (q->r1 - q->r0 <= 0) return AS_OK;
    if (q->exact) launch_knn_repair<double, unsigned long long, 8>(q, q->dots64, eps, exclude);
    else launch_knn_repair<float, unsigned int, 4>(q, q->dots32, eps, exclude);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

static as_status run_knn(as_query* q, double eps, int64_t exclude, int fuse_lambda, int32_t* o_idx, double* o_key,
                         double* o_dist, double* o_gy, int32_t* o_cnt, int thresholded = 0) {
    const as_space* sp = q->sp;
    hipStream_t st = q->stream;
    const int metric = sp->opts.metric;
    const double epskey = metric == AS_METRIC_L2 ? eps * eps : eps;
    FinishArgs f = make_finish(q);
    f.M = q->Mk; f.epskey = epskey; f.coef = coef_query(q, q->exact != 0);
    f.recs = o_idx ? nullptr : q->knn;
    f.o_idx = o_idx; f.o_key = o_key; f.o_dist = o_dist; f.o_gy = o_gy; f.o_cnt = o_cnt;
    f.fuse = fuse_lambda;
    f.thresholded = thresholded;
    if (q->robust) {
        int nw = 0;
        const int grid = sel_grid(q, &nw);
        f.nlists = nw;
        if (q->exact) {
            SelArgs<double> a = make_sel<double>(q, q->dots64, q->Mk, exclude);
            a.epskey = epskey; a.coef = f.coef;
            hipLaunchKernelGGL(knn_partial_kernel<double>, dim3(grid), dim3(256), 0, st, a);
            f.ck = q->pkey; f.ci = q->pidx;
            hipLaunchKernelGGL(knn_finish_kernel<double>, dim3(1), dim3(1024), finish_lds<double>(), st, f);
        } else {
            SelArgs<float> a = make_sel<float>(q, q->dots32, q->Mk, exclude);
            a.epskey = epskey; a.coef = f.coef;
            hipLaunchKernelGGL(knn_partial_kernel<float>, dim3(grid), dim3(256), 0, st, a);
            f.ck = q->pkey; f.ci = q->pidx;
            hipLaunchKernelGGL(knn_finish_kernel<float>, dim3(1), dim3(1024), finish_lds<float>(), st, f);
        }
    } else {
        f.ck = q->ckey_k; f.ci = q->cidx_k;
        if (q->exact)
            hipLaunchKernelGGL(knn_finish_kernel<double>, dim3(1, 1, q->nb), dim3(1024), finish_lds<double>(), st, f);
        else
            hipLaunchKernelGGL(knn_finish_kernel<float>, dim3(1, 1, q->nb), dim3(1024), finish_lds<float>(), st, f);
    }
    AS_HIP(hipGetLastError());
    return AS_OK;
}

template <typename T, typename U, int PASSES>
static void launch_score(as_query* q, const T* dots, FinishArgs f, int fuse_final) {
    hipStream_t st = q->stream;
    f.M = q->Ms; f.hits = q->hits; f.fuse = fuse_final; f.hout = q->hout_dev; f.seq = q->seq;
    const double coef_s = coef_query(q, sizeof(T) == 8);
    if (q->robust && q->Ms > MAX_LIST) {
        // wide lists: exact global selection, then the filter-path finish kernel on exactly M rows
        const int64_t rows = q->r1 - q->r0;
        const int KP = (int)sizeof(T);
        SelArgs<T> a = make_sel<T>(q, dots, q->Ms, -1);
        a.tau = f.tau;
        const unsigned fg = (unsigned)std::min<int64_t>((rows + 255) / 256, 2048);
        hipMemsetAsync(q->rsel, 0, sizeof(RSel), st);
        hipMemsetAsync(&q->info->sc_cnt, 0, sizeof(int), st);
        for (int pass = 0; pass < KP + 4; ++pass) {
            hipLaunchKernelGGL(rsel_hist_kernel<T>, dim3(fg), dim3(256), 0, st, a, pass, KP, q->rsel);
            hipLaunchKernelGGL(rsel_pick_kernel, dim3(1), dim3(64), 0, st, pass, KP, q->Ms, q->rsel);
        }
        hipLaunchKernelGGL(rsel_filter_kernel<T>, dim3(fg), dim3(256), 0, st, a, (const RSel*)q->rsel);
        f.ck = q->ckey_s; f.ci = q->cidx_s; f.from_list = 0;
        hipLaunchKernelGGL((score_finish_kernel<T>), dim3(1), dim3(1024), score_lds<T>(), st, f, coef_s);
    } else if (q->robust) {
        int nw = 0;
        const int grid = sel_grid(q, &nw);
        f.nlists = nw;
        SelArgs<T> a = make_sel<T>(q, dots, q->Ms, -1);
        a.tau = f.tau;
        hipLaunchKernelGGL(score_partial_kernel<T>, dim3(grid), dim3(256), 0, st, a);
        f.ck = q->pkey; f.ci = q->pidx;
        hipLaunchKernelGGL((score_finish_kernel<T>), dim3(1), dim3(1024), score_lds<T>(), st, f, coef_s);
    } else {
        const int64_t rows = q->r1 - q->r0;
        int64_t G = (rows + CAND_CAP - 1) / CAND_CAP;
        G = std::max<int64_t>(64, (G + 63) / 64 * 64);
        const int ng = (int)((rows + G - 1) / G);
        SelArgs<T> a = make_sel<T>(q, dots, q->Ms, -1);
        a.tau = f.tau;
        const unsigned nb = (unsigned)q->nb;
        hipLaunchKernelGGL((score_gmin_kernel<T, 0>), dim3((unsigned)((ng + 3) / 4), 1, nb), dim3(256), 0, st, a, G, ng);
        hipLaunchKernelGGL((pick_thr_kernel<T, U, PASSES>), dim3(1, 1, nb), dim3(1024), 0, st, (const T*)q->gmin, ng, q->Ms, q->info);
        const unsigned fg = (unsigned)std::min<int64_t>((rows + 255) / 256, 2048);
        hipLaunchKernelGGL((score_filter_kernel<T, 0>), dim3(fg, 1, nb), dim3(256), 0, st, a);
        f.ck = q->ckey_s; f.ci = q->cidx_s;
        hipLaunchKernelGGL((score_finish_kernel<T>), dim3(1, 1, nb), dim3(1024), score_lds<T>(), st, f, coef_s);
    }
}

static as_status run_score(as_query* q, double tau, int fuse_final) {
    hipStream_t st = q->stream;

    if (q->r1 - q->r0 <= 0) {
        AS_HIP(hipMemsetAsync(q->hits, 0xff, sizeof(as_hit_rec) * (q->topk + 1), st));
        return AS_OK;
    }
    FinishArgs f = make_finish(q);
    f.tau = tau;
    if (q->exact) launch_score<double, unsigned long long, 8>(q, q->dots64, f, fuse_final);
    else launch_score<float, unsigned int, 4>(q, q->dots32, f, fuse_final);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

static as_status query_begin(as_query* q, const double* query_host, int64_t src_row, int64_t d, int64_t r0, int64_t r1,
                             double eps, int64_t exclude) {
    const as_space* sp = q->sp;
    if (d != sp->d) {
        set_err("query length %lld must match nfeatures %lld", (long long)d, (long long)sp->d);
        return AS_EINVAL;
    }
    if (r0 < 0 || r1 > sp->n || r0 > r1) {
        set_err("as_query_scan: bad row range");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(sp->device));
    q->r0 = r0;
    q->r1 = r1;
    hipStream_t st = q->stream;
    const bool stats = g_search_stats.load(std::memory_order_relaxed) != 0;
    if (query_host) {
        memcpy(q->hq, query_host, sizeof(double) * d * q->nb);  // pinned + device-mapped: read in place by the kernel
        if (q->cap > q->nb) memset(q->hq + d * q->nb, 0, sizeof(double) * d * (q->cap - q->nb));  // idle slots: zero query
    } else {
        hipLaunchKernelGGL(q_from_row_kernel, dim3(1), dim3(256), 0, st, sp->x32, sp->x64, sp->d, sp->dp, src_row, q->hq_dev);
    }
    hipLaunchKernelGGL(q_prepare_kernel, dim3(1, 1, q->cap > 1 ? q->cap : q->nb), dim3(256), 0, st, q->hq_dev, sp->d, sp->dp, q->q64, q->q32, q->info, 1.0);
    if (stats) AS_HIP(hipEventRecord(q->ev[0], st));
    const PreArgs pre = make_pre(q, eps, exclude, !q->robust);
    AS_TRY(launch_scan(q, pre));
    if (stats) AS_HIP(hipEventRecord(q->ev[1], st));
    q->ev_valid = stats ? 1 : 0;
    return AS_OK;
}

// wait for the final kernel's publication (pinned memory), without the driver's sync path
static as_status wait_published(as_query* q, int slot = 0) {
    const int64_t want = q->seq;
    for (int spin = 0; spin < 2000000; ++spin) {
        if (q->hout[slot].seq == want) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return AS_OK;
        }
        if ((spin & 1023) == 1023 && hipStreamQuery(q->stream) == hipSuccess) break;
    }
    AS_HIP(hipStreamSynchronize(q->stream));
    if (q->hout[slot].seq != want) {
        set_err("search result was not published (seq %lld != %lld)", (long long)q->hout[slot].seq, (long long)want);
        return AS_EHIP;
    }
    return AS_OK;
}

static as_status collect(as_query* q, int64_t* out_idx, double* out_score, int64_t* out_len, double* out_lambda_q,
                         int slot = 0) {
    const HostOut* h = q->hout + slot;
    if (out_lambda_q) *out_lambda_q = h->lambda_q;
    if (h->status == AS_EZEROLAMBDA) {
        if (out_len) *out_len = 0;
        set_err("The lambdas are zero, check the magnitude of items and eps.");
        return AS_EZEROLAMBDA;
    }
    const int64_t len = h->len;
    for (int64_t t = 0; t < len; ++t) {
        out_idx[t] = h->idx[t];
        out_score[t] = h->score[t];
    }
    if (out_len) *out_len = len;
    return AS_OK;
}

}  // namespace as

using namespace as;

extern "C" {

void as_enable_search_stats(int32_t enabled) { g_search_stats.store(enabled ? 1 : 0, std::memory_order_relaxed); }

}  // extern "C"

namespace as {
as_status query_create(const as_space* sp, const as_graph* gr, int cap, as_query** out) {
    if (!sp || !out) {
        set_err("as_query_create: null argument");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(sp->device));
    as_query* q = new as_query();
    q->cap = cap;
    const size_t C = (size_t)cap;
    q->sp = sp;
    q->gr = gr;
    q->k = gr ? gr->gp.k : 1;
    q->topk = gr ? std::min<int64_t>(gr->gp.topk, sp->n) : 1;
    q->Mk = list_width(std::min<int64_t>(q->k, sp->n));
    q->Ms = score_width(q->topk);
    if (q->Mk < 0 || q->Ms < 0) {
        set_err("k=%lld exceeds the supported maximum of 56, or topk=%lld the maximum of 1024", (long long)q->k, (long long)q->topk);
        delete q;
        return AS_EUNSUPPORTED;
    }
    q->nwaves = 4096;
    if (const char* ev = getenv("ARROWSPACE_SCAN_VARIANT")) q->scan_variant = atoi(ev) & 7;
    if (const char* ev = getenv("ARROWSPACE_GEMM_VARIANT")) q->gemm_variant = atoi(ev);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, sp->device) == hipSuccess) q->cus = prop.multiProcessorCount;
    }
    AS_HIP(hipStreamCreateWithFlags(&q->own_stream, hipStreamNonBlocking));
    q->stream = q->own_stream;
    AS_HIP(hipHostMalloc(&q->hq, sizeof(double) * sp->d * C, hipHostMallocMapped | hipHostMallocCoherent));
    AS_HIP(hipHostGetDevicePointer((void**)&q->hq_dev, q->hq, 0));
    AS_HIP(hipMalloc(&q->q64, sizeof(double) * sp->dp * C));
    AS_HIP(hipMalloc(&q->q32, sizeof(float) * sp->dp * C));
    AS_HIP(hipMalloc(&q->info, sizeof(QInfo) * C));
    AS_HIP(hipMalloc(&q->dots32, sizeof(float) * (sp->np + ROW_TILE) * C));
    q->ss.dots = C > 1 ? 32 : sp->np + ROW_TILE;
    q->ss.dots_ts = C > 1 ? 32 * (int64_t)C : 32;
    q->ss.q = sp->dp;
    q->ss.qin = sp->d;
    q->ss.knn = std::max<int64_t>(q->k, 1);
    q->ss.hits = q->topk + 1;
    AS_HIP(hipMalloc(&q->pkey, sizeof(double) * (size_t)q->nwaves * 64));
    AS_HIP(hipMalloc(&q->pidx, sizeof(int) * (size_t)q->nwaves * 64));
    AS_HIP(hipMalloc(&q->ckey_k, sizeof(double) * CAND_CAP * C));
    AS_HIP(hipMalloc(&q->cidx_k, sizeof(int) * CAND_CAP * C));
    AS_HIP(hipMalloc(&q->ckey_s, sizeof(double) * CAND_CAP * C));
    AS_HIP(hipMalloc(&q->cidx_s, sizeof(int) * CAND_CAP * C));
    AS_HIP(hipMalloc(&q->gmin, sizeof(double) * CAND_CAP * C));
    AS_HIP(hipMalloc(&q->rsel, sizeof(RSel)));
    AS_HIP(hipMalloc(&q->knn, sizeof(as_knn_rec) * q->ss.knn * C));
    AS_HIP(hipMalloc(&q->hits, sizeof(as_hit_rec) * q->ss.hits * C));
    AS_HIP(hipHostMalloc(&q->hout, sizeof(HostOut) * C, hipHostMallocMapped | hipHostMallocCoherent));
    AS_HIP(hipHostGetDevicePointer((void**)&q->hout_dev, q->hout, 0));
    memset(q->hout, 0, sizeof(HostOut) * C);
    for (int i = 0; i < 3; ++i) AS_HIP(hipEventCreate(&q->ev[i]));
    AS_HIP(hipFuncSetAttribute((const void*)knn_finish_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)finish_lds<double>()));
    AS_HIP(hipFuncSetAttribute((const void*)score_finish_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)score_lds<double>()));
    AS_HIP(hipFuncSetAttribute((const void*)score_finish_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)score_lds<float>()));
    AS_HIP(hipFuncSetAttribute((const void*)hits_final_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((sizeof(double) + sizeof(int)) * HIT_CAP)));
    q->r0 = 0;
    q->r1 = sp->n;
    *out = q;
    return AS_OK;
}
}  // namespace as

extern "C" {

as_status as_query_create(const as_space* sp, const as_graph* gr, as_query** out) { return query_create(sp, gr, 1, out); }

void as_query_free(as_query* q) {
    if (!q) return;
    hipSetDevice(q->sp->device);
    hipStreamSynchronize(q->stream);
    hipHostFree(q->hq); hipFree(q->q64); hipFree(q->q32); hipFree(q->info); hipFree(q->dots32);
    if (q->dots64) hipFree(q->dots64);
    hipFree(q->pkey); hipFree(q->pidx); hipFree(q->ckey_k); hipFree(q->cidx_k); hipFree(q->ckey_s); hipFree(q->cidx_s);
    hipFree(q->gmin);
    hipFree(q->rsel);
    if (q->own_records) {
        hipFree(q->knn);
        hipFree(q->hits);
    }
    hipHostFree(q->hout);
    for (int i = 0; i < 3; ++i) hipEventDestroy(q->ev[i]);
    hipStreamDestroy(q->own_stream);
    delete q;
}

void* as_query_stream(const as_query* q) { return (void*)q->stream; }

void as_query_set_stream(as_query* q, void* stream) {
    if (!q) return;
    hipStreamSynchronize(q->stream);
    q->stream = stream ? (hipStream_t)stream : q->own_stream;
}

as_status as_query_bind_records(as_query* q, as_knn_rec* knn_dev, as_hit_rec* hits_dev) {
    if (!q || !knn_dev || !hits_dev) {
        set_err("as_query_bind_records: null argument");
        return AS_EINVAL;
    }
    if (q->own_records) {
        hipFree(q->knn);
        hipFree(q->hits);
        q->own_records = 0;
    }
    q->knn = knn_dev;
    q->hits = hits_dev;
    return AS_OK;
}
const as_knn_rec* as_query_knn_records(const as_query* q) { return q->knn; }
int64_t as_query_knn_capacity(const as_query* q) { return q->k; }
const as_hit_rec* as_query_hit_records(const as_query* q) { return q->hits; }
int64_t as_query_hit_capacity(const as_query* q) { return q->topk + 1; }

as_status as_query_scan(as_query* q, const double* query_host, int64_t d, int64_t row_begin, int64_t row_end) {
    if (!q || !query_host || !q->gr) {
        set_err("as_query_scan: null argument");
        return AS_EINVAL;
    }
    if (q->reuse && !q->robust) {
        // same query again after a candidate-buffer overflow: threshold repair over the kept dots
        if (row_begin != q->r0 || row_end != q->r1) {
            set_err("as_query_scan: the repair pass must cover the rows of the scan it repairs");
            return AS_EINVAL;
        }
        AS_TRY(knn_repair(q, q->gr->gp.eps, -1));
        return run_knn(q, q->gr->gp.eps, -1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1);
    }
    AS_TRY(query_begin(q, query_host, -1, d, row_begin, row_end, q->gr->gp.eps, -1));
    return run_knn(q, q->gr->gp.eps, -1, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
}

as_status as_query_lambda(as_query* q, const as_knn_rec* recs_dev, int64_t m) {
    if (!q || !q->gr || !recs_dev) {
        set_err("as_query_lambda: null argument");
        return AS_EINVAL;
    }
    if (m > REC_CAP) {
        set_err("as_query_lambda: %lld records exceed the supported %d", (long long)m, REC_CAP);
        return AS_EUNSUPPORTED;
    }
    const as_graph* gr = q->gr;
    hipLaunchKernelGGL(q_lambda_kernel, dim3(1), dim3(64), 0, q->stream, recs_dev, m, q->k, gr->metric, gr->kernel,
                       gr->gp.sigma, gr->gp.p, gr->tau0, q->info);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

as_status as_query_score(as_query* q, double tau) {
    if (!q || !q->gr) {
        set_err("as_query_score: null argument");
        return AS_EINVAL;
    }
    return run_score(q, tau, 0);
}

as_status as_query_finish(as_query* q, const as_hit_rec* hits_dev, int64_t m, int64_t* out_idx, double* out_score,
                          int64_t* out_len, double* out_lambda_q) {
    if (!q || !hits_dev) {
        set_err("as_query_finish: null argument");
        return AS_EINVAL;
    }
    if (m > HIT_CAP) {
        set_err("as_query_finish: %lld records exceed the supported %d", (long long)m, HIT_CAP);
        return AS_EUNSUPPORTED;
    }
    hipStream_t st = q->stream;
    const int64_t topk = std::min<int64_t>(q->gr->gp.topk, q->sp->n);
    q->seq += 1;
    hipLaunchKernelGGL(hits_final_kernel, dim3(1), dim3(1024), (sizeof(double) + sizeof(int)) * HIT_CAP, st, hits_dev, m, topk, q->info,
                       q->hout_dev, q->seq);
    AS_HIP(hipGetLastError());
    if (q->ev_valid) AS_HIP(hipEventRecord(q->ev[2], st));
    AS_TRY(wait_published(q));
    return collect(q, out_idx, out_score, out_len, out_lambda_q);
}

void as_query_set_exact(as_query* q, int32_t flags) {
    if (!q) return;
    q->exact = (flags & 1) ? 1 : 0;
    q->robust = (flags & 2) ? 1 : 0;
    q->reuse = (flags & 4) ? 1 : 0;
}

as_status as_query_flags(const as_query* q, int32_t* knn_inexact, int32_t* score_inexact) {
    if (!q) return AS_EINVAL;
    if (knn_inexact) *knn_inexact = q->hout->knn_inexact | ((q->hout->overflow & 1) ? 2 : 0);
    if (score_inexact) *score_inexact = q->hout->score_inexact | ((q->hout->overflow & 2) ? 2 : 0);
    return AS_OK;
}

as_status as_query_stats(const as_query* q, double* out, int32_t n) {
    if (!q || !out) return AS_EINVAL;
    float ms01 = 0, ms12 = 0;
    if (q->ev_valid) {
        hipEventSynchronize(q->ev[2]);
        hipEventElapsedTime(&ms01, q->ev[0], q->ev[1]);
        hipEventElapsedTime(&ms12, q->ev[1], q->ev[2]);
    }
    const double v[3] = {ms01 * 1e3, ms12 * 1e3, q->stats[2]};
    for (int i = 0; i < n && i < 3; ++i) out[i] = v[i];
    return AS_OK;
}

}  // extern "C"

namespace as {

void query_flags(const as_query* q, int* knn_inexact, int* score_inexact) {
    *knn_inexact = q->hout->knn_inexact;
    *score_inexact = q->hout->score_inexact;
    if (q->hout->overflow) *knn_inexact |= 2;
}

// one full single-GPU search on q's stream: 6 launches, one host wait
as_status search_once(as_query* q, const double* query, int64_t d, double tau, int mode, int64_t* out_idx, double* out_score,
                      int64_t* out_len, double* out_lambda_q) {
    q->exact = (mode & 1) || q->sp->opts.force_exact;
    q->robust = (mode & 2) ? 1 : 0;
    AS_TRY(query_begin(q, query, -1, d, 0, q->sp->n, q->gr->gp.eps, -1));
    AS_TRY(run_knn(q, q->gr->gp.eps, -1, 1, nullptr, nullptr, nullptr, nullptr, nullptr));
    q->seq += 1;
    AS_TRY(run_score(q, tau, 1));
    if (q->ev_valid) AS_HIP(hipEventRecord(q->ev[2], q->stream));
    AS_TRY(wait_published(q));
    if (!q->robust && (q->hout->overflow & 1)) {
        // more than CAND_CAP rows inside eps: re-derive the candidates from the kept dots, no second scan
        AS_TRY(knn_repair(q, q->gr->gp.eps, -1));
        AS_TRY(run_knn(q, q->gr->gp.eps, -1, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 1));
        q->seq += 1;
        AS_TRY(run_score(q, tau, 1));
        AS_TRY(wait_published(q));
    }
    return collect(q, out_idx, out_score, out_len, out_lambda_q);
}

// up to QB queries in one pass over the items (filter path, fp32 prefilters); out_status[b] is
// AS_OK / AS_EZEROLAMBDA, or -1 when slot b must be rerun on the single-query path (a candidate
// buffer overflowed or an a-posteriori check failed)
as_status search_batch_once(as_query* q, const double* queries, int nb, int64_t d, double tau, int64_t topk, int64_t* out_idx,
                            double* out_score, int64_t* out_len, double* out_lambda_q, int32_t* out_status) {
    q->exact = 0;
    q->robust = 0;
    q->nb = nb;
    AS_TRY(query_begin(q, queries, -1, d, 0, q->sp->n, q->gr->gp.eps, -1));
    AS_TRY(run_knn(q, q->gr->gp.eps, -1, 1, nullptr, nullptr, nullptr, nullptr, nullptr));
    q->seq += 1;
    AS_TRY(run_score(q, tau, 1));
    bool crowded = false;
    for (int b = 0; b < nb; ++b) {
        AS_TRY(wait_published(q, b));
        crowded = crowded || (q->hout[b].overflow & 1);
    }
    if (crowded) {
        // some neighbourhood overflowed its candidate buffer: threshold repair of every slot over the kept dots
        AS_TRY(knn_repair(q, q->gr->gp.eps, -1));
        AS_TRY(run_knn(q, q->gr->gp.eps, -1, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 1));
        q->seq += 1;
        AS_TRY(run_score(q, tau, 1));
    }
    for (int b = 0; b < nb; ++b) {
        AS_TRY(wait_published(q, b));
        const HostOut* h = q->hout + b;
        if (h->overflow || h->knn_inexact || h->score_inexact) {
            out_status[b] = -1;
            continue;
        }
        const as_status s = collect(q, out_idx + b * topk, out_score + b * topk, out_len + b, out_lambda_q ? out_lambda_q + b : nullptr, b);
        out_status[b] = (int32_t)s;
    }
    return AS_OK;
}

as_status exact_row_knn(as_query* ws, const as_graph_params* gp, int64_t row, int32_t* out_idx, double* out_key,
                        double* out_dist, double* out_gy, int32_t* out_cnt) {
    const as_space* sp = ws->sp;
    ws->k = gp->k;
    ws->Mk = list_width(std::min<int64_t>(gp->k, sp->n));
    if (ws->Mk < 0) {
        set_err("k=%lld exceeds the supported maximum of 56", (long long)gp->k);
        return AS_EUNSUPPORTED;
    }
    ws->exact = 1;
    ws->robust = 1;  // rows with many near-ties are exactly the ones that overflow a filter buffer
    AS_TRY(query_begin(ws, nullptr, row, sp->d, 0, sp->n, gp->eps, row));
    return run_knn(ws, gp->eps, row, 0, out_idx, out_key, out_dist, out_gy, out_cnt);
}

}  // namespace as
