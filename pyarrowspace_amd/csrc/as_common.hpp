// Shared host/device definitions for the gfx950 hot path (not part of the C ABI).
#pragma once
#include <memory>
#include <atomic>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>

#include "arrowspace_hip.h"

namespace as {

constexpr int WAVE = 64;
constexpr int ROW_TILE = 256;   // rows of the item matrix are padded to a multiple of this
constexpr int COL_PAD = 32;     // feature dimension is padded to a multiple of this (floats)
constexpr double TAU_MIN = 1e-12;
constexpr int MAX_LIST = 64;    // widest wave-resident candidate list of one slot per lane (WaveList; WaveList2 holds two)
constexpr int MAX_KLIST = 128;  // widest k-NN candidate list (k + 8 <= 128)

// ---------------------------------------------------------------- errors
std::string& err_slot();
void set_err(const char* fmt, ...);
bool debug_enabled();
void dbg(const char* fmt, ...);

#define AS_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t _e = (call);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::as::set_err("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return AS_EHIP;                                                                   \
        }                                                                                     \
    } while (0)

#define AS_TRY(call)                    \
    do {                                \
        as_status _s = (call);          \
        if (_s != AS_OK) return _s;     \
    } while (0)

}  // namespace as

// ---------------------------------------------------------------- handles
// HBM layout: x32 is the scan/MFMA operand, [np][dp] fp32,
// zero padded so that every tile load is a full 16-byte vector inside the allocation.
struct as_space {
    int device = 0;
    int64_t n = 0, d = 0, np = 0, dp = 0;
    float* x32 = nullptr;     // [np][dp]
    mutable float* xs = nullptr;  // [np][dp] bf16 head + tail image of x32 (as_k2bf.hip), made by the first k-NN pass; null until then
    // int8 two-digit image (as_k2bf.hip, quant_rows_i8): x ~ s_i (128 a1 + a2) / 16256, per row and 64-column slab 64 bytes of
    // a1 then 64 of a2 ([np + 256][dp8 * 2] bytes, dp8 = dp rounded up to 64); fa8[i] = s_i sqrt(128) / 16256; coef8 = the
    // error coefficient its products carry (from max_i s_i / |x_i| and max_i |x_i|_1 / |x_i|); k2_i8: the k-NN pass in
    // flight runs on it (set and cleared by knn_rows: err_coef follows it)
    mutable void* x8 = nullptr;
    mutable void* x8h = nullptr;   // the HIGH digits alone, planar ([np + 256][dp8] bytes): the single-query scan's coarse operand (space_i8h_image)
    mutable float* fa8 = nullptr;
    mutable double coef8 = 0.0;
    // ring build (one process per GPU): the ranks agreed to run the block passes on the int8 images (as_ring_i8_set); the
    // coefficient of a product of rows of two shards from the ring-wide maxima of U and V
    int ring_i8 = 0;
    double ring_u8 = 0.0, ring_v8 = 0.0, ring_coef8 = 0.0;
    mutable double uq_est = 0.0, vq_est = 0.0;   // batched int8 pass: 1.15 x the queries' measured residue norms of the previous passes (as_search.hip, host_batch_coef)
    mutable double u8max = 0.0, v8max = 0.0;   // max_i s_i |theta_i|_2 / (16256 |x_i|), max_i s_i |a2_i|_2 / (16256 |x_i|)
    mutable int x8_bad = 0;           // 1: non-finite items (the image exists but cannot be used), 2: no memory for the image (not retried)
    // image state for readers that do not take imu: 0 nothing decided, 1 x8 / fa8 / coef8 published (or x8_bad set): read-only from now on
    mutable std::atomic<int> x8_ready{0};
    mutable std::atomic<int> x8h_ready{0};   // ... the planar high digits: 1 x8h published, 2 not available (no memory)
    mutable std::mutex imu;           // makes the images (first use), per space
    mutable int k2_i8 = 0;
    mutable int k2_last_pipe = -1;   // matrix pipe of the last k-NN pass on this space: 0 fp32, 1 bf16 head + tail, 2 int8 two digits (as_space_knn_pipe)
    int64_t dp8 = 0;
    double* x64 = nullptr;    // [n][d] or null when the items are exactly fp32-representable
    double* n64 = nullptr;    // [n] squared norms (fp64, from the fp64 items)
    float* n32 = nullptr;     // [np] squared norms rounded to fp32 (0 on pad rows)
    float* inorm32 = nullptr; // [np] 1/|x| (0 for zero rows / pad rows)
    double* lam64 = nullptr;  // [n] lambdas (written by as_graph_from_knn)
    float* lam32 = nullptr;   // [np]
    int64_t row_offset = 0;   // global index of row 0 (a rank's shard of a row-sharded index); 0 for a whole index
    double nmax = 0.0;        // max squared norm
    int lossless = 0;         // items exactly representable in fp32
    as_opts opts{};
    hipStream_t stream = nullptr;
    // as_search is re-entrant across host threads: a pool of single-query workspaces (own stream, buffers, pinned results
    // each), lazily grown up to QPOOL; a call takes a free one under qmu and runs WITHOUT the lock, so one thread's scan
    // overlaps another's finish kernel and host turnaround.  qcache == qpool[0] (what a single thread ever uses).
    static constexpr int QPOOL = 4;
    mutable as_query* qpool[QPOOL] = {nullptr, nullptr, nullptr, nullptr};
    mutable bool qbusy[QPOOL] = {false, false, false, false};
    mutable std::condition_variable qcv;
    mutable std::mutex bmu;                   // the batched workspaces below (as_search_batch calls are serialised)
    mutable as_query* qcache = nullptr;       // lazily created by as_search
    mutable const as_graph* qcache_gr = nullptr;
    mutable as_query* qcache_b = nullptr;     // batched workspace (QUERY_BATCH slots), lazily created
    mutable as_query* qcache_b2 = nullptr;    // its twin: the two launch their passes as a PAIR that shares one scan (search_batch_launch_pair)
    mutable as_query* qcache_b3 = nullptr;    // a second pair (calls of more than 64 queries): one pair scans while the other's selection and
    mutable as_query* qcache_b4 = nullptr;    // finish kernels run and its results are read
    mutable std::atomic<long long> batch_dual_scans{0};   // scans that served two workspaces' passes (as_batch_counters)
    mutable const as_graph* qcache_b_gr = nullptr;
    mutable std::mutex qmu;
    // Gang scans (as_search.hip, gang_launch): single queries of host threads that arrive together share ONE pass over the tiles
    // of the coarse image.  active_callers: threads inside as_search on this space right now; gang_hint > 0: two or more were seen
    // within the last 64 searches (a lone thread never lingers); gang_width: the most seen at once lately; gang_open: the gang
    // that is gathering members (under gmu).
    mutable std::mutex gmu;
    mutable std::atomic<int> active_callers{0};
    mutable std::atomic<int> gang_hint{0};
    mutable std::atomic<int> gang_width{1};
    mutable std::atomic<unsigned> gang_tick{0};
    mutable std::shared_ptr<struct as_gang> gang_open;
    // one shared scan at a time, in the order the gangs were opened: the ticket of the next gang to open / to launch, and the event
    // behind the scan launched last (a workspace's gang_ev)
    mutable int64_t gang_seq_next = 0;                 // under gmu
    mutable std::atomic<int64_t> gang_seq_launched{0};
    mutable std::atomic<hipEvent_t> last_scan_ev{nullptr};
    mutable int64_t gang_scans[5] = {0, 0, 0, 0, 0};   // scans launched with 1 .. 4 members (index = members; under gmu)
    // host-prepared single-query scans that did NOT take the shared path, by first reason: [0] no concurrent callers lately, [1] the
    // search is not one for the fused tail (tau < 0.4, crowded candidates lately, ...), [2] not a coarse scan, [3] per-search
    // state still to be reset on the workspace's stream, [4] per-launch timing on, [5] a row range
    mutable std::atomic<int64_t> gang_skip[6] = {};
    // as_search_counters: [0] searches, [1] zero-lambda results, [2] reruns because the k-NN a-posteriori check failed,
    // [3] reruns because a candidate buffer overflowed, [4] reruns because the scorer's check failed, [5] searches that
    // took at least one rerun (under qmu)
    mutable int64_t scount[6] = {0, 0, 0, 0, 0, 0};
    mutable int64_t unproven_searches = 0;   // searches returned although their a-posteriori check failed on the strongest path
    mutable double kstats[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // accumulated by as_knn_rows
};

struct as_graph {
    int device = 0;
    int64_t n = 0;
    int64_t nnz = 0;          // adjacency entries (both directions, no diagonal)
    as_graph_params gp{};     // sigma resolved
    int metric = 0, kernel = 0;
    int64_t* indptr = nullptr;  // [n+1]
    int32_t* indices = nullptr; // [nnz] ascending per row
    double* dist = nullptr;     // [nnz] d_ij
    double* gy = nullptr;       // [nnz] y_i . y_j
    double* w = nullptr;        // [nnz] a_ij
    double* lap = nullptr;      // [nnz] -a_ij / sqrt(deg_i deg_j)
    double* deg = nullptr;      // [n]
    double* ny = nullptr;       // [n] |y_i|^2
    double* E = nullptr;        // [n]
    double* G = nullptr;        // [n]
    double tau0 = 0.0;
    double stats[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // feature mode (AS_LAMBDA_FEATURE): n == nfeatures, the CSR above is the feature graph (lap = -w, the
    // off-diagonal of L = D - W), nitems the number of items E / G / the lambdas cover
    int lambda_mode = 0;
    int64_t nitems = 0;
    int64_t e_rows = 0;         // feature mode: entries E / G hold -- nitems after as_feat_lambdas_global (every item's, all-gathered),
                                // the space's own rows after a single-space build or a load
    int64_t ne = 0;             // edges a < b
    int32_t* ea = nullptr;      // [ne]
    int32_t* eb = nullptr;
    double* ew = nullptr;
    double* colm = nullptr;     // [n] squared column norms (the Gram's diagonal)
    // row-sharded item graph (as_graph_shard_*): the CSR holds the rows [row0, row0 + n) of the graph over ncols
    // items, column ids global; deg / ny / E / G cover these rows only.  ncols == 0: a whole graph (ncols = n).
    int64_t ncols = 0;
    int64_t row0 = 0;
};
inline int64_t graph_items(const as_graph* gr) {   // items the index covers
    return gr->lambda_mode == AS_LAMBDA_FEATURE ? gr->nitems : (gr->ncols ? gr->ncols : gr->n);
}

// ---------------------------------------------------------------- device helpers
#if defined(__HIPCC__)
namespace as {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// LDS hand-off between lanes of ONE wave: the hardware keeps a wave's LDS operations in
// order; wait for them and stop the compiler from caching LDS values across the point.
#define AS_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// Scratch device allocation of a host function: freed on every return path, error paths included.
template <typename T>
struct dev_tmp {
    T* p = nullptr;
    dev_tmp() = default;
    dev_tmp(const dev_tmp&) = delete;
    dev_tmp& operator=(const dev_tmp&) = delete;
    ~dev_tmp() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t count) { return hipMalloc((void**)&p, sizeof(T) * (count ? count : 1)); }
    operator T*() const { return p; }
};

// HIP events of a host function: destroyed on every return path, error paths included.
template <int N>
struct dev_events {
    hipEvent_t e[N] = {};
    dev_events() = default;
    dev_events(const dev_events&) = delete;
    dev_events& operator=(const dev_events&) = delete;
    ~dev_events() {
        for (int i = 0; i < N; ++i)
            if (e[i]) (void)hipEventDestroy(e[i]);
    }
    hipError_t create() {
        for (int i = 0; i < N; ++i) {
            const hipError_t r = hipEventCreate(&e[i]);
            if (r != hipSuccess) return r;
        }
        return hipSuccess;
    }
};

// SPEC S6 edge energy w ||y_i/sqrt(d_i) - y_j/sqrt(d_j)||^2 from the stored pair quantities; a value inside the rounding
// noise of the expanded form's terms is zero: a neighbour identical to the node must not become a 1e-16 "energy"
// whose share of the sum is 1
// (DESIGN.md section 2, S7).
// The form does not cancel for near-identical neighbours (alpha = d_i^-1/2, beta = d_j^-1/2):
//   l2:     alpha beta dist^2 + (alpha - beta)(alpha n_i - beta n_j)   (dist^2 summed as differences)
//   cosine: (alpha - beta)^2 + 2 alpha beta (1 - c) for unit vectors, alpha^2 n_i + beta^2 n_j if one is zero
__host__ __device__ __forceinline__ double edge_energy(double w, int metric, double dist, double g, double di, double dj,
                                                        double nyi, double nyj) {
    const double alpha = 1.0 / sqrt(di), beta = 1.0 / sqrt(dj);
    double core;
    if (metric == AS_METRIC_L2) core = alpha * beta * (dist * dist) + (alpha - beta) * (alpha * nyi - beta * nyj);
    else if (nyi > 0.0 && nyj > 0.0) core = (alpha - beta) * (alpha - beta) + 2.0 * alpha * beta * (1.0 - g);
    else core = alpha * alpha * nyi + beta * beta * nyj;
    const double v = w * core;
    const double floor_ = w * 0x1p-46 * (alpha * alpha * nyi + beta * beta * nyj + 2.0 * alpha * beta * fabs(g));
    return v > floor_ ? v : 0.0;
}
// SPEC S2 cosine distance: rounding can push a cosine past 1 -- no negative distances (a fractional p would turn
// them into NaN weights)
__host__ __device__ __forceinline__ double cosine_distance(double c) { return 1.0 - (c > 0.0 ? (c < 1.0 ? c : 1.0) : 0.0); }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float bcast_lane(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ int bcast_lane(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ double bcast_lane(double v, int src) {
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <typename T>
__device__ __forceinline__ bool lex_less(T ka, int ia, T kb, int ib) {
    return ka < kb || (ka == kb && ia < ib);
}

template <typename T>
struct key_traits;
template <>
struct key_traits<float> {
    static __device__ __forceinline__ float inf() { return __int_as_float(0x7f800000); }
};
template <>
struct key_traits<double> {
    static __device__ __forceinline__ double inf() { return __longlong_as_double(0x7ff0000000000000LL); }
};

// Wave-resident sorted candidate list ("wavefront-shuffle top-k"): lane t holds
// the t-th smallest (key, idx) seen so far; slots >= m stay (+inf, INT_MAX).
template <typename T>
struct WaveList {
    T key;
    int idx;
    __device__ __forceinline__ void init() {
        key = key_traits<T>::inf();
        idx = 0x7fffffff;
    }
    // threshold = entry m-1
    __device__ __forceinline__ void thr(int m, T& tk, int& ti) const {
        tk = bcast_lane(key, m - 1);
        ti = bcast_lane(idx, m - 1);
    }
    // insert a wave-uniform candidate; caller guarantees (ck,ci) < entry m-1
    __device__ __forceinline__ void insert(T ck, int ci) {
        const int lane = lane_id();
        const bool less = lex_less<T>(key, idx, ck, ci);
        const int pos = __popcll(__ballot(less));
        T pk = __shfl_up(key, 1, 64);
        int pi = __shfl_up(idx, 1, 64);
        if (lane == pos) {
            key = ck;
            idx = ci;
        } else if (lane > pos) {
            key = pk;
            idx = pi;
        }
    }
    // offer one candidate per lane (inactive lanes pass valid=false)
    __device__ __forceinline__ void offer(int m, T ck, int ci, bool valid) {
        T tk;
        int ti;
        thr(m, tk, ti);
        bool pass = valid && lex_less<T>(ck, ci, tk, ti);
        unsigned long long mask = __ballot(pass);
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            const T bk = bcast_lane(ck, src);
            const int bi = bcast_lane(ci, src);
            if (lex_less<T>(bk, bi, tk, ti)) {
                insert(bk, bi);
                thr(m, tk, ti);
            }
            mask &= mask - 1;
            // drop lanes that no longer beat the tightened threshold
            pass = pass && lex_less<T>(ck, ci, tk, ti);
            mask &= __ballot(pass);
        }
    }
};

// 128-slot variant (k up to 120): slot t in (key, idx), slot 64 + t in (key2, idx2); same contract as WaveList.
template <typename T>
struct WaveList2 {
    T key, key2;
    int idx, idx2;
    __device__ __forceinline__ void init() {
        key = key2 = key_traits<T>::inf();
        idx = idx2 = 0x7fffffff;
    }
    __device__ __forceinline__ void thr(int m, T& tk, int& ti) const {
        if (m <= 64) {
            tk = bcast_lane(key, m - 1);
            ti = bcast_lane(idx, m - 1);
        } else {
            tk = bcast_lane(key2, m - 65);
            ti = bcast_lane(idx2, m - 65);
        }
    }
    __device__ __forceinline__ void insert(T ck, int ci) {
        const int lane = lane_id();
        const int pos = __popcll(__ballot(lex_less<T>(key, idx, ck, ci))) + __popcll(__ballot(lex_less<T>(key2, idx2, ck, ci)));
        const T pk = __shfl_up(key, 1, 64), pk2 = __shfl_up(key2, 1, 64);
        const int pi = __shfl_up(idx, 1, 64), pi2 = __shfl_up(idx2, 1, 64);
        const T lk = bcast_lane(key, 63);      // slot 63 moves up into slot 64
        const int li = bcast_lane(idx, 63);
        const int g2 = 64 + lane;
        if (g2 == pos) {
            key2 = ck;
            idx2 = ci;
        } else if (g2 > pos) {
            key2 = lane == 0 ? lk : pk2;
            idx2 = lane == 0 ? li : pi2;
        }
        if (lane == pos) {
            key = ck;
            idx = ci;
        } else if (lane > pos) {
            key = pk;
            idx = pi;
        }
    }
    __device__ __forceinline__ void offer(int m, T ck, int ci, bool valid) {
        T tk;
        int ti;
        thr(m, tk, ti);
        bool pass = valid && lex_less<T>(ck, ci, tk, ti);
        unsigned long long mask = __ballot(pass);
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            const T bk = bcast_lane(ck, src);
            const int bi = bcast_lane(ci, src);
            if (lex_less<T>(bk, bi, tk, ti)) {
                insert(bk, bi);
                thr(m, tk, ti);
            }
            mask &= mask - 1;
            pass = pass && lex_less<T>(ck, ci, tk, ti);
            mask &= __ballot(pass);
        }
    }
};

// SPEC S4 edge weight.  gaussian: exp(-0.5 (d/sigma)^p); rational: 1/(1+(d/sigma)^p)
__device__ __forceinline__ double edge_weight(double d, double sigma, double p, int kernel) {
    const double t = d / sigma;
    const double u = (p == 2.0) ? t * t : pow(t, p);
    return kernel == AS_KERNEL_GAUSSIAN ? exp(-0.5 * u) : 1.0 / (1.0 + u);
}

}  // namespace as
#endif  // __HIPCC__

// ---------------------------------------------------------------- internal entry points
namespace as {

struct Scratch;  // growable device scratch owned by a call

as_status resolve_params(const as_graph_params* gp, as_graph_params* out);
as_status check_limits(const as_graph_params* resolved, int64_t n, int lambda_mode);
// per-pair fp32 error coefficient: |key32 - key64| <= coef * (n_i + n_j) for L2,
// <= coef for cosine
double err_coef(const as_space* sp);
as_status ring_i8_stats(as_space* sp, double* out3);
as_status ring_i8_set(as_space* sp, double u_max, double v_max, int32_t usable);
// the int8 two-digit image of the space's items (x8, fa8, u8max, v8max, coef8), made on first use; *present: it exists and
// holds no non-finite item (whether its error is acceptable is the caller's call: build pass, scan)
as_status space_i8_image(const as_space* sp, bool* present);
as_status space_i8h_image(const as_space* sp, bool* present);

// build stages (as_build.hip)
as_status ingest(as_space* sp, const void* items_dev, int dtype, int64_t ld);
as_status ingest_host(as_space* sp, const double* items, int64_t row_stride, int64_t col_stride);
as_status knn_rows(const as_space* sp, const as_graph_params* gp, int64_t r0, int64_t r1, int32_t* out_idx,
                   double* out_key, double* out_dist, double* out_gy, int32_t* out_cnt, double* stats);
as_status graph_from_knn(as_space* sp, const as_graph_params* gp, const int32_t* idx, const double* dist,
                         const double* gy, const int32_t* cnt, as_graph* gr);
as_status csr_from_knn(hipStream_t st, int64_t n, int64_t k, const int32_t* idx, const double* dist, const double* gy,
                       const int32_t* cnt, double sigma, double p, int kernel, as_graph* gr);
as_status median_lambda(as_space* sp, as_graph* gr, const double* E, const double* G);
as_status median_lambda_n(hipStream_t st, int64_t n, const double* E, const double* G, double* lam64, float* lam32, double* tau0_out);
as_status lam_slice(hipStream_t st, int64_t n, const double* src, double* lam64, float* lam32);
as_status graph_shard_csr(as_space* sp, const as_graph_params* gp, int64_t n_global, int64_t row_offset, const int32_t* idx,
                          const double* dist, const double* gy, const int32_t* cnt, int64_t n_in, const int32_t* in_row,
                          const int32_t* in_col, const double* in_dist, const double* in_gy, as_graph* gr);
as_status graph_shard_energy(as_space* sp, as_graph* gr, const double* deg_global, const double* n64_global);
as_status graph_shard_lambdas(as_space* sp, as_graph* gr, const double* E_global, int64_t n_global);
as_status graph_from_knn_global(as_space* sp, const as_graph_params* gp, int64_t n_global, int64_t row_offset, const int32_t* idx,
                                const double* dist, const double* gy, const int32_t* cnt, const double* n64_global, as_graph* gr);
// k-NN over visiting column blocks (multi-GPU ring, as_build.hip)
as_status knn_block(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t r0, int64_t r1, int64_t row_goff,
                    int64_t col_goff, int M, double* p_key, double* p_dist, double* p_gy, int32_t* p_idx, int32_t* p_cnt, float* p_t32);
as_status knn_merge(const as_space* sp, const as_graph_params* gp, int64_t r0, int64_t r1, int nblocks, int M, const double* p_key,
                    const double* p_dist, const double* p_gy, const int32_t* p_idx, const int32_t* p_cnt, const float* p_t32,
                    const double* block_nmax_host, int32_t* out_idx, double* out_key, double* out_dist, double* out_gy,
                    int32_t* out_cnt, int32_t* flag, double* out_B, int64_t* nflagged);
as_status knn_block_band(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t r0, int64_t r1, int64_t row_goff,
                         int64_t col_goff, int M, int32_t* flag, const double* B, double* p_key, double* p_dist, double* p_gy,
                         int32_t* p_idx, int32_t* p_cnt, float* p_t32, int64_t* overflowed);
int knn_list_width(int64_t k);
as_status knn_block_exact(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t r0, int64_t r1, int64_t row_goff,
                          int64_t col_goff, int M, const int32_t* flag, double* p_key, double* p_dist, double* p_gy, int32_t* p_idx,
                          int32_t* p_cnt, float* p_t32);
as_status knn_block_pair(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t r0, int64_t r1, int64_t ct0,
                         int64_t ct1, int64_t row_goff, int64_t col_goff, const float* col_thr, const float* row_thr, int M, double* p_key,
                         double* p_dist, double* p_gy, int32_t* p_idx, int32_t* p_cnt, float* p_t32, double* q_key, double* q_dist, double* q_gy,
                         int32_t* q_idx, int32_t* q_cnt, float* q_t32);
as_status knn_thresholds(const as_space* sp, int64_t r0, int64_t r1, int M, double nmax_all, const double* r_key, const int32_t* r_cnt,
                         float* out_thr);
as_status knn_fold(const as_space* sp, int64_t r0, int64_t r1, int M, int mode, double block_nmax, const int32_t* flag, double* r_key,
                   double* r_dist, double* r_gy, int32_t* r_idx, int32_t* r_cnt, float* r_t32, const double* b_key, const double* b_dist,
                   const double* b_gy, const int32_t* b_idx, const int32_t* b_cnt, const float* b_t32);

// feature mode (as_feat.hip)
as_status feat_gram(const as_space* sp, int64_t r0, int64_t r1, double* gram);
as_status feat_graph(const as_space* sp, const as_graph_params* gp, const double* gram, as_graph* gr);
as_status feat_energy(const as_space* sp, const as_graph* gr, int64_t r0, int64_t r1, double* E, double* G);
as_status feat_build(as_space* sp, const as_graph_params* gp, as_graph* gr);
as_status feat_edges_from_csr(as_graph* gr, hipStream_t st);   // rebuilds ea/eb/ew from the CSR (index load)
struct QInfo;
as_status feat_query_lambda(const as_graph* gr, const double* q64, int64_t dp, QInfo* info, int nslots, hipStream_t st);
as_status feat_query_prepare(const as_graph* gr, const double* qin, int64_t d, int64_t dp, double* q64, float* q32, QInfo* info,
                             int nslots, hipStream_t st);

// query-as-row exact k-NN used by the build fallback (as_search.hip)
as_status exact_row_knn(as_query* ws, const as_graph_params* gp, int64_t row, int32_t* out_idx,
                        double* out_key, double* out_dist, double* out_gy, int32_t* out_cnt);
as_status search_once(as_query* q, const double* query, int64_t d, double tau, int exact, int64_t* out_idx,
                      double* out_score, int64_t* out_len, double* out_lambda_q);
void query_flags(const as_query* q, int* knn_inexact, int* score_inexact);
int query_overflow_bits(const as_query* q);
as_status query_create(const as_space* sp, const as_graph* gr, int cap, as_query** out, int pool_slot = 0);
as_status search_batch_launch(as_query* q, const double* queries, int nb, int64_t d, double tau);
as_status search_batch_launch_pair(as_query* a, as_query* b, const double* qa, int nba, const double* qb, int nbb, int64_t d, double tau);
as_status search_batch_collect(as_query* q, int nb, double tau, int64_t topk, int64_t* out_idx, double* out_score, int64_t* out_len,
                               double* out_lambda_q, int32_t* out_status);
constexpr int QUERY_BATCH = 32;  // == GQ in as_search.hip: query slots of the batched workspace

}  // namespace as
