// Lambda-blended search on gfx950.  Replaces `prepare_query_item` + `search_lambda_aware`
// (/root/reference/src/lib.rs:154,173; scorer form TAUMODE.md:33).  SPEC = DESIGN.md
// section 2 (S10, S11).  One HBM pass over the fp32 item matrix per query (scan_dots),
// everything after it works on N-length vectors that stay in L2 / Infinity Cache.
#include <algorithm>
#include <mutex>

#include "as_common.hpp"

namespace as {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct QInfo {
    double nq;        // |q|^2
    double lambda_q;
    double tau;
    float nq32, inq32;
    double inq;       // 1/|q|
    int status;       // as_status of the lambda step
    int knn_inexact;  // a-posteriori check of the k-NN candidate list failed
    int score_inexact;
    int knn_total;    // candidates that passed the eps prefilter (all waves)
    int nhit;
    int pad;
};

struct HostOut {
    int64_t len;
    double lambda_q;
    int status, knn_inexact, score_inexact, pad;
    int64_t idx[MAX_LIST];
    double score[MAX_LIST];
};

}  // namespace as

struct as_query {
    const as_space* sp = nullptr;
    const as_graph* gr = nullptr;
    hipStream_t stream = nullptr;
    int64_t k = 0, topk = 0;
    int Mk = 32, Ms = 32;
    int nwaves = 0;
    int64_t r0 = 0, r1 = 0;
    int exact = 0;
    double* qin = nullptr;   // [d] raw query
    double* q64 = nullptr;   // [dp] zero padded
    float* q32 = nullptr;    // [dp]
    as::QInfo* info = nullptr;
    float* dots32 = nullptr; // [np]
    double* dots64 = nullptr;
    void* pkey = nullptr;    // [nwaves][64] keys (sized for double)
    int* pidx = nullptr;     // [nwaves][64]
    as_knn_rec* knn = nullptr;
    as_hit_rec* hits = nullptr;
    as::HostOut* hout = nullptr;  // pinned
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    double stats[4] = {0, 0, 0, 0};
    std::mutex mu;
};

namespace as {

// ------------------------------------------------------------------ query staging
__global__ void q_prepare_kernel(const double* __restrict__ qin, int64_t d, int64_t dp, double* __restrict__ q64,
                                 float* __restrict__ q32, QInfo* info, double tau) {
    __shared__ double sh[256];
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
        info->lambda_q = 0.0;
        info->tau = tau;
        info->status = AS_OK;
        info->knn_inexact = 0;
        info->score_inexact = 0;
        info->knn_total = 0;
        info->nhit = 0;
    }
}

__global__ void q_from_row_kernel(const float* __restrict__ x32, const double* __restrict__ x64, int64_t d, int64_t dp,
                                  int64_t row, double* __restrict__ qin) {
    for (int64_t c = threadIdx.x; c < d; c += blockDim.x) qin[c] = x64 ? x64[row * d + c] : (double)x32[row * dp + c];
}

// ------------------------------------------------------------------ K7a scan: dots[i] = x_i . q
// HBM-bound: one wave per row, 16 B per lane per load, query fragment in registers,
// two rows in flight per wave.  NCH = ceil(dp / 256) chunks of 256 floats.
template <int NCH>
__global__ __launch_bounds__(256) void scan_dots_f32_kernel(const float* __restrict__ x32, const float* __restrict__ q32,
                                                            int64_t dp, int64_t r0, int64_t r1, float* __restrict__ dots) {
    const int lane = lane_id();
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    f32x4 qv[NCH];
    bool on[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        const int64_t c = 4 * (lane + 64 * u);
        on[u] = c < dp;
        qv[u] = on[u] ? *(const f32x4*)(q32 + c) : f32x4{0, 0, 0, 0};
    }
    int64_t row = r0 + gw;
    for (; row + nw < r1; row += 2 * nw) {
        const float* pa = x32 + row * dp + 4 * lane;
        const float* pb = pa + nw * dp;
        f32x4 va[NCH], vb[NCH];
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            va[u] = on[u] ? __builtin_nontemporal_load((const f32x4*)(pa + 256 * u)) : f32x4{0, 0, 0, 0};
            vb[u] = on[u] ? __builtin_nontemporal_load((const f32x4*)(pb + 256 * u)) : f32x4{0, 0, 0, 0};
        }
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
        if (lane == 0) {
            dots[row] = sa;
            dots[row + nw] = sb;
        }
    }
    if (row < r1) {
        const float* pa = x32 + row * dp + 4 * lane;
        float sa = 0.0f;
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            if (on[u]) {
                const f32x4 v = __builtin_nontemporal_load((const f32x4*)(pa + 256 * u));
#pragma unroll
                for (int e = 0; e < 4; ++e) sa = fmaf(v[e], qv[u][e], sa);
            }
        }
        sa = wave_sum(sa);
        if (lane == 0) dots[row] = sa;
    }
}

// generic width (dp > 2048): query re-read from L1 per chunk
__global__ __launch_bounds__(256) void scan_dots_f32_generic_kernel(const float* __restrict__ x32, const float* __restrict__ q32,
                                                                    int64_t dp, int64_t r0, int64_t r1, float* __restrict__ dots) {
    const int lane = lane_id();
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
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
        if (lane == 0) dots[row] = s;
    }
}

// exact mode: fp64 accumulation over the fp64 items (or the widened fp32 items when lossless)
__global__ __launch_bounds__(256) void scan_dots_f64_kernel(const float* __restrict__ x32, const double* __restrict__ x64,
                                                            const double* __restrict__ q64, int64_t d, int64_t dp, int64_t r0,
                                                            int64_t r1, double* __restrict__ dots) {
    const int lane = lane_id();
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
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
        if (lane == 0) dots[row] = s;
    }
}

// ------------------------------------------------------------------ wavefront-shuffle partial selections
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
    double epskey, coef;
    T* pkey;
    int* pidx;
};

template <typename T>
__device__ __forceinline__ T knn_key(const SelArgs<T>& a, int64_t row, T dot, double nq, double inq);
template <>
__device__ __forceinline__ float knn_key<float>(const SelArgs<float>& a, int64_t row, float dot, double, double) {
    if (a.metric == AS_METRIC_L2) return fmaf(-2.0f, dot, a.n32[row] + a.info->nq32);
    return 1.0f - fmaxf(0.0f, dot * a.inorm32[row] * a.info->inq32);
}
template <>
__device__ __forceinline__ double knn_key<double>(const SelArgs<double>& a, int64_t row, double dot, double nq, double) {
    if (a.metric == AS_METRIC_L2) return a.n64[row] + nq - 2.0 * dot;
    const double den = sqrt(a.n64[row] * nq);
    const double c = den > 0.0 ? dot / den : 0.0;
    return 1.0 - (c > 0.0 ? c : 0.0);
}

template <typename T>
__global__ __launch_bounds__(256) void knn_partial_kernel(SelArgs<T> a) {
    const int lane = lane_id();
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    const double nq = a.info->nq, inq = a.info->inq;
    WaveList<T> lst;
    lst.init();
    int npass = 0;
    for (int64_t base = a.r0 + gw * 64; base < a.r1; base += nw * 64) {
        const int64_t row = base + lane;
        bool valid = row < a.r1 && row < a.n && row != a.exclude;
        T key = key_traits<T>::inf();
        if (valid) {
            key = knn_key<T>(a, row, a.dots[row], nq, inq);
            const double ni = sizeof(T) == 4 ? (double)a.n32[row] : a.n64[row];
            const double bound = a.metric == AS_METRIC_L2 ? a.epskey + a.coef * (ni + nq) : a.epskey + a.coef;
            valid = (double)key <= bound;
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
    const double nq = a.info->nq;
    const double tau = a.info->tau, lq = a.info->lambda_q;
    const float tau32 = (float)tau, lq32 = (float)lq, inq32 = a.info->inq32;
    WaveList<T> lst;
    lst.init();
    for (int64_t base = a.r0 + gw * 64; base < a.r1; base += nw * 64) {
        const int64_t row = base + lane;
        const bool valid = row < a.r1 && row < a.n;
        T key = key_traits<T>::inf();
        if (valid) {
            if (sizeof(T) == 4) {
                const float c = (float)a.dots[row] * a.inorm32[row] * inq32;
                const float term = 1.0f / (1.0f + fabsf(lq32 - a.lam32[row]));
                key = (T)(-(tau32 * c + (1.0f - tau32) * term));
            } else {
                const double den = sqrt(a.n64[row] * nq);
                const double c = den > 0.0 ? (double)a.dots[row] / den : 0.0;
                key = (T)(-(tau * c + (1.0 - tau) / (1.0 + fabs(lq - a.lam64[row]))));
            }
        }
        lst.offer(a.M, key, (int)row, valid);
    }
    a.pkey[gw * 64 + lane] = lst.key;
    a.pidx[gw * 64 + lane] = lst.idx;
}

// merge nlists partial lists (64 slots each) down to one sorted list in LDS (fk, fi);
// block of 1024 threads (16 waves).  Returns valid count in *fcount.
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

__device__ __forceinline__ void exact_pair_q(const float* x32, const double* x64, const double* q64, int64_t d, int64_t dp,
                                             int64_t j, double& sq, double& dot) {
    const int lane = lane_id();
    double s = 0.0, g = 0.0;
    if (x64) {
        const double* pj = x64 + j * d;
        for (int64_t c = lane; c < d; c += 64) {
            const double a = q64[c], b = pj[c], t = a - b;
            s += t * t;
            g += a * b;
        }
    } else {
        const float* pj = x32 + j * dp;
        for (int64_t c = lane; c < d; c += 64) {
            const double a = q64[c], b = (double)pj[c], t = a - b;
            s += t * t;
            g += a * b;
        }
    }
    sq = wave_sum(s);
    dot = wave_sum(g);
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
    int nlists, M, metric;
    double epskey, coef, nmax;
    as_knn_rec* recs;
    as_hit_rec* hits;
    // build-fallback outputs (row-list form); null for searches
    int32_t* o_idx;
    double* o_key;
    double* o_dist;
    double* o_gy;
    int32_t* o_cnt;
};

template <typename T>
__global__ __launch_bounds__(1024) void knn_finish_kernel(FinishArgs a, const T* pkey, const int* pidx) {
    __shared__ T wk[16 * 64];
    __shared__ int wi[16 * 64];
    __shared__ T fk[64];
    __shared__ int fi[64];
    __shared__ double ek[64], ed[64], eg[64], sk[64];
    __shared__ int fcount;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    merge_partials<T>(pkey, pidx, a.nlists, a.M, wk, wi, fk, fi, &fcount);
    const int Mp = fcount;
    const double nq = a.info->nq;
    for (int t = w; t < Mp; t += 16) {
        const int j = fi[t];
        double sq, dot;
        exact_pair_q(a.x32, a.x64, a.q64, a.d, a.dp, j, sq, dot);
        if (lane == 0) {
            if (a.metric == AS_METRIC_L2) {
                ek[t] = sq;
                ed[t] = sqrt(sq);
                eg[t] = dot;
            } else {
                const double den = sqrt(nq * a.n64[j]);
                const double c = den > 0.0 ? dot / den : 0.0;
                const double dd = 1.0 - (c > 0.0 ? c : 0.0);
                ek[t] = dd;
                ed[t] = dd;
                eg[t] = c;
            }
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
    if (have) sk[rank] = myk;
    const bool pass = have && myk <= a.epskey;
    const int npass = __popcll(__ballot(pass));
    const int cnt = npass < a.k ? npass : (int)a.k;
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
    if (pass && rank < a.k) {
        if (a.recs) {
            as_knn_rec r;
            r.idx = myi;
            r.key = myk;
            r.dist = ed[lane];
            r.gy = eg[lane];
            r.deg = a.deg ? a.deg[myi] : 0.0;
            r.ny = a.ny ? a.ny[myi] : 0.0;
            a.recs[rank] = r;
        }
        if (a.o_idx) {
            a.o_idx[rank] = myi;
            a.o_key[rank] = myk;
            a.o_dist[rank] = ed[lane];
            a.o_gy[rank] = eg[lane];
        }
    }
    if (lane == 0) {
        if (a.o_cnt) *a.o_cnt = cnt;
        int bad = 0;
        if (a.info->knn_total > a.M && Mp > 0) {
            const double B = npass >= a.k ? sk[a.k - 1] : a.epskey;
            // dropped items' norms are unknown: bound them by the largest norm in the space
            const double e = a.metric == AS_METRIC_L2 ? a.coef * (a.nmax + nq) : a.coef;
            const double Tm = (double)fk[Mp - 1];
            bad = !(Tm - e > B);
        }
        a.info->knn_inexact = bad;
    }
}

// SPEC S10: lambda_q from m candidate records (this shard's, or all shards' gathered)
__global__ __launch_bounds__(64) void q_lambda_kernel(const as_knn_rec* __restrict__ recs, int64_t m, int64_t k, int M,
                                                      int metric, int kernel, double sigma, double p, double tau0, QInfo* info) {
    __shared__ double s_dist[64], s_gy[64], s_deg[64], s_ny[64];
    __shared__ int s_id[64];
    const int lane = lane_id();
    WaveList<double> lst;
    lst.init();
    for (int64_t base = 0; base < m; base += 64) {
        const int64_t t = base + lane;
        const bool valid = t < m && recs[t].idx >= 0;
        const double key = valid ? recs[t].key : key_traits<double>::inf();
        const int id = valid ? (int)recs[t].idx : 0x7fffffff;
        lst.offer(M, key, id, valid);
    }
    const bool sel = lane < k && lst.idx != 0x7fffffff;
    const int cnt = __popcll(__ballot(sel));
    // ascending-index order
    int rank = 0;
    for (int s = 0; s < 64; ++s) {
        const int oi = bcast_lane(lst.idx, s);
        const bool osel = s < k && oi != 0x7fffffff;
        rank += (osel && oi < lst.idx) ? 1 : 0;
    }
    if (sel) {
        // locate the record carrying this item (first match; duplicates are identical)
        int64_t pos = -1;
        for (int64_t t = 0; t < m; ++t)
            if (pos < 0 && recs[t].idx == (int64_t)lst.idx) pos = t;
        s_id[rank] = lst.idx;
        s_dist[rank] = recs[pos].dist;
        s_gy[rank] = recs[pos].gy;
        s_deg[rank] = recs[pos].deg;
        s_ny[rank] = recs[pos].ny;
    }
    __syncthreads();
    if (lane != 0) return;
    double lam = 0.0;
    const double nq = info->nq;
    const double nyq = metric == AS_METRIC_L2 ? nq : (nq > 0.0 ? 1.0 : 0.0);
    if (cnt > 0 && nyq > 0.0) {
        double degq = 0.0;
        for (int t = 0; t < cnt; ++t) degq += edge_weight(s_dist[t], sigma, p, kernel);
        if (degq > 0.0) {
            double S = 0.0;
            for (int t = 0; t < cnt; ++t) {
                const double at = edge_weight(s_dist[t], sigma, p, kernel);
                const double dj = s_deg[t] + at;
                const double sdd = sqrt(degq * dj);
                const double v = at * (nyq / degq + s_ny[t] / dj - 2.0 * s_gy[t] / sdd);
                const double ev = v > 0.0 ? v : 0.0;
                s_dist[t] = ev;  // reuse as the edge energy
                S += ev;
            }
            const double Eq = 0.5 * S / nyq;
            double Gq = 0.0;
            if (S > 0.0) {
                for (int t = 0; t < cnt; ++t) {
                    const double r = s_dist[t] / S;
                    Gq += r * r;
                }
                Gq = Gq < 0.0 ? 0.0 : (Gq > 1.0 ? 1.0 : Gq);
            }
            lam = tau0 * (Eq / (Eq + tau0)) + (1.0 - tau0) * Gq;
        }
    }
    info->lambda_q = lam;
    info->status = lam == 0.0 ? AS_EZEROLAMBDA : AS_OK;
}

template <typename T>
__global__ __launch_bounds__(1024) void score_finish_kernel(FinishArgs a, const T* pkey, const int* pidx, double coef_s) {
    __shared__ T wk[16 * 64];
    __shared__ int wi[16 * 64];
    __shared__ T fk[64];
    __shared__ int fi[64];
    __shared__ double es[64], sk[64];
    __shared__ int fcount;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    merge_partials<T>(pkey, pidx, a.nlists, a.M, wk, wi, fk, fi, &fcount);
    const int Mp = fcount;
    const double nq = a.info->nq, tau = a.info->tau, lq = a.info->lambda_q;
    for (int t = w; t < Mp; t += 16) {
        const int j = fi[t];
        double sq, dot;
        exact_pair_q(a.x32, a.x64, a.q64, a.d, a.dp, j, sq, dot);
        if (lane == 0) {
            const double den = sqrt(a.n64[j] * nq);
            const double c = den > 0.0 ? dot / den : 0.0;
            es[t] = tau * c + (1.0 - tau) / (1.0 + fabs(lq - a.lam64[j]));
        }
    }
    __syncthreads();
    if (w != 0) return;
    const bool have = lane < Mp;
    const double myk = have ? -es[lane] : 0.0;
    const int myi = have ? fi[lane] : 0x7fffffff;
    int rank = 0;
    for (int s = 0; s < Mp; ++s) rank += lex_less<double>(-es[s], fi[s], myk, myi) ? 1 : 0;
    if (have) sk[rank] = es[lane];
    for (int64_t t = lane; t < a.topk; t += 64) {
        as_hit_rec r;
        r.idx = -1;
        r.score = -key_traits<double>::inf();
        a.hits[t] = r;
    }
    if (have && rank < a.topk) {
        as_hit_rec r;
        r.idx = myi;
        r.score = es[lane];
        a.hits[rank] = r;
    }
    if (lane == 0) {
        int bad = 0;
        const int64_t want = a.topk < a.nrows ? a.topk : a.nrows;
        if (a.nrows > a.M && Mp > 0) {
            // every dropped item has score32 <= -fk[Mp-1]; exact score <= that + coef_s
            const double kth = Mp >= want ? sk[want - 1] : -key_traits<double>::inf();
            const double ub = -(double)fk[Mp - 1] + coef_s;
            bad = !(ub < kth);
        }
        a.info->score_inexact = bad;
        a.info->nhit = (int)(Mp < want ? Mp : want);
    }
}

// merge m hit records -> final topk, written to pinned host memory
__global__ __launch_bounds__(64) void hits_final_kernel(const as_hit_rec* __restrict__ hits, int64_t m, int64_t topk, int M,
                                                        const QInfo* info, HostOut* out) {
    const int lane = lane_id();
    WaveList<double> lst;
    lst.init();
    for (int64_t base = 0; base < m; base += 64) {
        const int64_t t = base + lane;
        const bool valid = t < m && hits[t].idx >= 0;
        lst.offer(M, valid ? -hits[t].score : key_traits<double>::inf(), valid ? (int)hits[t].idx : 0x7fffffff, valid);
    }
    const bool sel = lane < topk && lst.idx != 0x7fffffff;
    const int cnt = __popcll(__ballot(sel));
    if (sel) {
        out->idx[lane] = lst.idx;
        out->score[lane] = -lst.key;
    }
    if (lane == 0) {
        out->len = cnt;
        out->lambda_q = info->lambda_q;
        out->status = info->status;
        out->knn_inexact = info->knn_inexact;
        out->score_inexact = info->score_inexact;
    }
}

// ------------------------------------------------------------------ host side
static int list_width(int64_t k) {
    const int64_t need = k + 8;
    if (need <= 32) return 32;
    if (need <= 64) return 64;
    return -1;
}

static double coef_query(int64_t dp, bool exact) {
    const double u = exact ? 1.1102230246251565e-16 : 5.9604644775390625e-8;
    return (double)(dp / 64 + 24) * u;
}

static as_status launch_scan(as_query* q) {
    const as_space* sp = q->sp;
    const int64_t rows = q->r1 - q->r0;
    if (rows <= 0) return AS_OK;
    hipStream_t st = q->stream;
    if (q->exact) {
        if (!q->dots64) AS_HIP(hipMalloc(&q->dots64, sizeof(double) * (sp->np + ROW_TILE)));
        const unsigned grid = (unsigned)std::min<int64_t>((rows + 3) / 4, 4096);
        hipLaunchKernelGGL(scan_dots_f64_kernel, dim3(grid), dim3(256), 0, st, sp->x32, sp->x64, q->q64, sp->d, sp->dp, q->r0,
                           q->r1, q->dots64);
    } else {
        const unsigned grid = (unsigned)std::min<int64_t>((rows + 3) / 4, 2048);
        const int nch = (int)((sp->dp + 255) / 256);
#define AS_SCAN(N) hipLaunchKernelGGL(scan_dots_f32_kernel<N>, dim3(grid), dim3(256), 0, st, sp->x32, q->q32, sp->dp, q->r0, q->r1, q->dots32)
        switch (nch) {
            case 1: AS_SCAN(1); break;
            case 2: AS_SCAN(2); break;
            case 3: AS_SCAN(3); break;
            case 4: AS_SCAN(4); break;
            case 5: AS_SCAN(5); break;
            case 6: AS_SCAN(6); break;
            case 7: AS_SCAN(7); break;
            case 8: AS_SCAN(8); break;
            default:
                hipLaunchKernelGGL(scan_dots_f32_generic_kernel, dim3(grid), dim3(256), 0, st, sp->x32, q->q32, sp->dp, q->r0, q->r1, q->dots32);
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
    a.epskey = 0; a.coef = 0;
    a.pkey = (T*)q->pkey; a.pidx = q->pidx;
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

static as_status run_knn(as_query* q, double eps, int64_t exclude, int32_t* o_idx, double* o_key, double* o_dist, double* o_gy,
                         int32_t* o_cnt) {
    const as_space* sp = q->sp;
    hipStream_t st = q->stream;
    const int metric = sp->opts.metric;
    const double epskey = metric == AS_METRIC_L2 ? eps * eps : eps;
    int nw = 0;
    const int grid = sel_grid(q, &nw);
    FinishArgs f;
    memset(&f, 0, sizeof(f));
    f.x32 = sp->x32; f.x64 = sp->x64; f.n64 = sp->n64; f.q64 = q->q64;
    f.deg = q->gr ? q->gr->deg : nullptr; f.ny = q->gr ? q->gr->ny : nullptr; f.lam64 = sp->lam64;
    f.info = q->info; f.n = sp->n; f.d = sp->d; f.dp = sp->dp; f.k = q->k; f.topk = q->topk; f.nrows = q->r1 - q->r0;
    f.nlists = nw; f.M = q->Mk; f.metric = metric; f.epskey = epskey; f.nmax = sp->nmax;
    f.recs = o_idx ? nullptr : q->knn; f.hits = nullptr;
    f.o_idx = o_idx; f.o_key = o_key; f.o_dist = o_dist; f.o_gy = o_gy; f.o_cnt = o_cnt;
    if (q->exact) {
        SelArgs<double> a = make_sel<double>(q, q->dots64, q->Mk, exclude);
        a.epskey = epskey;
        a.coef = coef_query(sp->dp, true);
        hipLaunchKernelGGL(knn_partial_kernel<double>, dim3(grid), dim3(256), 0, st, a);
        f.coef = a.coef;
        hipLaunchKernelGGL(knn_finish_kernel<double>, dim3(1), dim3(1024), 0, st, f, (const double*)q->pkey, (const int*)q->pidx);
    } else {
        SelArgs<float> a = make_sel<float>(q, q->dots32, q->Mk, exclude);
        a.epskey = epskey;
        a.coef = coef_query(sp->dp, false);
        hipLaunchKernelGGL(knn_partial_kernel<float>, dim3(grid), dim3(256), 0, st, a);
        f.coef = a.coef;
        hipLaunchKernelGGL(knn_finish_kernel<float>, dim3(1), dim3(1024), 0, st, f, (const float*)q->pkey, (const int*)q->pidx);
    }
    AS_HIP(hipGetLastError());
    return AS_OK;
}

}  // namespace as

using namespace as;

extern "C" {

as_status as_query_create(const as_space* sp, const as_graph* gr, as_query** out) {
    if (!sp || !out) {
        set_err("as_query_create: null argument");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(sp->device));
    as_query* q = new as_query();
    q->sp = sp;
    q->gr = gr;
    q->k = gr ? gr->gp.k : 1;
    q->topk = gr ? std::min<int64_t>(gr->gp.topk, sp->n) : 1;
    q->Mk = list_width(std::min<int64_t>(q->k, sp->n));
    q->Ms = list_width(q->topk);
    if (q->Mk < 0 || q->Ms < 0) {
        set_err("k=%lld / topk=%lld exceed the supported maximum of 56", (long long)q->k, (long long)q->topk);
        delete q;
        return AS_EUNSUPPORTED;
    }
    q->nwaves = 4096;
    AS_HIP(hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking));
    AS_HIP(hipMalloc(&q->qin, sizeof(double) * sp->d));
    AS_HIP(hipMalloc(&q->q64, sizeof(double) * sp->dp));
    AS_HIP(hipMalloc(&q->q32, sizeof(float) * sp->dp));
    AS_HIP(hipMalloc(&q->info, sizeof(QInfo)));
    AS_HIP(hipMalloc(&q->dots32, sizeof(float) * (sp->np + ROW_TILE)));
    AS_HIP(hipMalloc(&q->pkey, sizeof(double) * (size_t)q->nwaves * 64));
    AS_HIP(hipMalloc(&q->pidx, sizeof(int) * (size_t)q->nwaves * 64));
    AS_HIP(hipMalloc(&q->knn, sizeof(as_knn_rec) * std::max<int64_t>(q->k, 1)));
    AS_HIP(hipMalloc(&q->hits, sizeof(as_hit_rec) * std::max<int64_t>(q->topk, 1)));
    AS_HIP(hipHostMalloc(&q->hout, sizeof(HostOut), hipHostMallocDefault));
    for (int i = 0; i < 3; ++i) AS_HIP(hipEventCreate(&q->ev[i]));
    q->r0 = 0;
    q->r1 = sp->n;
    *out = q;
    return AS_OK;
}

void as_query_free(as_query* q) {
    if (!q) return;
    hipSetDevice(q->sp->device);
    hipStreamSynchronize(q->stream);
    hipFree(q->qin); hipFree(q->q64); hipFree(q->q32); hipFree(q->info); hipFree(q->dots32);
    if (q->dots64) hipFree(q->dots64);
    hipFree(q->pkey); hipFree(q->pidx); hipFree(q->knn); hipFree(q->hits);
    hipHostFree(q->hout);
    for (int i = 0; i < 3; ++i) hipEventDestroy(q->ev[i]);
    hipStreamDestroy(q->stream);
    delete q;
}

void* as_query_stream(const as_query* q) { return (void*)q->stream; }
const as_knn_rec* as_query_knn_records(const as_query* q) { return q->knn; }
int64_t as_query_knn_capacity(const as_query* q) { return q->k; }
const as_hit_rec* as_query_hit_records(const as_query* q) { return q->hits; }
int64_t as_query_hit_capacity(const as_query* q) { return q->topk; }

static as_status query_scan_impl(as_query* q, const double* query_host, const double* query_dev_row_of, int64_t src_row,
                                 int64_t d, int64_t r0, int64_t r1, double tau_unused) {
    (void)tau_unused;
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
    if (query_host) {
        AS_HIP(hipMemcpyAsync(q->qin, query_host, sizeof(double) * d, hipMemcpyHostToDevice, st));
    } else {
        (void)query_dev_row_of;
        hipLaunchKernelGGL(q_from_row_kernel, dim3(1), dim3(256), 0, st, sp->x32, sp->x64, sp->d, sp->dp, src_row, q->qin);
    }
    hipLaunchKernelGGL(q_prepare_kernel, dim3(1), dim3(256), 0, st, q->qin, sp->d, sp->dp, q->q64, q->q32, q->info, 1.0);
    AS_HIP(hipEventRecord(q->ev[0], st));
    AS_TRY(launch_scan(q));
    AS_HIP(hipEventRecord(q->ev[1], st));
    return AS_OK;
}

as_status as_query_scan(as_query* q, const double* query_host, int64_t d, int64_t row_begin, int64_t row_end) {
    if (!q || !query_host) {
        set_err("as_query_scan: null argument");
        return AS_EINVAL;
    }
    AS_TRY(query_scan_impl(q, query_host, nullptr, -1, d, row_begin, row_end, 0.0));
    return run_knn(q, q->gr->gp.eps, -1, nullptr, nullptr, nullptr, nullptr, nullptr);
}

as_status as_query_lambda(as_query* q, const as_knn_rec* recs_dev, int64_t m) {
    if (!q || !q->gr || !recs_dev) {
        set_err("as_query_lambda: null argument");
        return AS_EINVAL;
    }
    const as_graph* gr = q->gr;
    hipLaunchKernelGGL(q_lambda_kernel, dim3(1), dim3(64), 0, q->stream, recs_dev, m, q->k, q->Mk, gr->metric, gr->kernel,
                       gr->gp.sigma, gr->gp.p, gr->tau0, q->info);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

__global__ void set_tau_kernel(QInfo* info, double tau) { info->tau = tau; }

as_status as_query_score(as_query* q, double tau) {
    if (!q || !q->gr) {
        set_err("as_query_score: null argument");
        return AS_EINVAL;
    }
    const as_space* sp = q->sp;
    hipStream_t st = q->stream;
    hipLaunchKernelGGL(set_tau_kernel, dim3(1), dim3(1), 0, st, q->info, tau);
    int nw = 0;
    const int grid = sel_grid(q, &nw);
    FinishArgs f;
    memset(&f, 0, sizeof(f));
    f.x32 = sp->x32; f.x64 = sp->x64; f.n64 = sp->n64; f.q64 = q->q64; f.lam64 = sp->lam64;
    f.info = q->info; f.n = sp->n; f.d = sp->d; f.dp = sp->dp; f.k = q->k; f.topk = q->topk; f.nrows = q->r1 - q->r0;
    f.nlists = nw; f.M = q->Ms; f.metric = sp->opts.metric; f.hits = q->hits;
    if (q->r1 - q->r0 <= 0) {
        // empty shard: publish empty hit records
        AS_HIP(hipMemsetAsync(q->hits, 0xff, sizeof(as_hit_rec) * q->topk, st));
        return AS_OK;
    }
    if (q->exact) {
        SelArgs<double> a = make_sel<double>(q, q->dots64, q->Ms, -1);
        hipLaunchKernelGGL(score_partial_kernel<double>, dim3(grid), dim3(256), 0, st, a);
        hipLaunchKernelGGL(score_finish_kernel<double>, dim3(1), dim3(1024), 0, st, f, (const double*)q->pkey, (const int*)q->pidx,
                           coef_query(sp->dp, true));
    } else {
        SelArgs<float> a = make_sel<float>(q, q->dots32, q->Ms, -1);
        hipLaunchKernelGGL(score_partial_kernel<float>, dim3(grid), dim3(256), 0, st, a);
        hipLaunchKernelGGL(score_finish_kernel<float>, dim3(1), dim3(1024), 0, st, f, (const float*)q->pkey, (const int*)q->pidx,
                           coef_query(sp->dp, false));
    }
    AS_HIP(hipGetLastError());
    return AS_OK;
}

as_status as_query_finish(as_query* q, const as_hit_rec* hits_dev, int64_t m, int64_t* out_idx, double* out_score,
                          int64_t* out_len, double* out_lambda_q) {
    if (!q || !hits_dev) {
        set_err("as_query_finish: null argument");
        return AS_EINVAL;
    }
    hipStream_t st = q->stream;
    const int64_t topk = std::min<int64_t>(q->gr->gp.topk, q->sp->n);
    hipLaunchKernelGGL(hits_final_kernel, dim3(1), dim3(64), 0, st, hits_dev, m, topk, q->Ms, q->info, q->hout);
    AS_HIP(hipGetLastError());
    AS_HIP(hipEventRecord(q->ev[2], st));
    AS_HIP(hipStreamSynchronize(st));
    const HostOut* h = q->hout;
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

void as_query_set_exact(as_query* q, int32_t exact) {
    if (q) q->exact = exact ? 1 : 0;
}

as_status as_query_flags(const as_query* q, int32_t* knn_inexact, int32_t* score_inexact) {
    if (!q) return AS_EINVAL;
    if (knn_inexact) *knn_inexact = q->hout->knn_inexact;
    if (score_inexact) *score_inexact = q->hout->score_inexact;
    return AS_OK;
}

as_status as_query_stats(const as_query* q, double* out, int32_t n) {
    if (!q || !out) return AS_EINVAL;
    float ms01 = 0, ms12 = 0;
    hipEventElapsedTime(&ms01, q->ev[0], q->ev[1]);
    hipEventElapsedTime(&ms12, q->ev[1], q->ev[2]);
    const double v[3] = {ms01 * 1e3, ms12 * 1e3, q->stats[2]};
    for (int i = 0; i < n && i < 3; ++i) out[i] = v[i];
    return AS_OK;
}

}  // extern "C"

namespace as {

void query_flags(const as_query* q, int* knn_inexact, int* score_inexact) {
    *knn_inexact = q->hout->knn_inexact;
    *score_inexact = q->hout->score_inexact;
}

// one full search on q's stream; exact=1 reruns everything in fp64
as_status search_once(as_query* q, const double* query, int64_t d, double tau, int exact, int64_t* out_idx, double* out_score,
                      int64_t* out_len, double* out_lambda_q) {
    q->exact = exact || q->sp->opts.force_exact;
    AS_TRY(as_query_scan(q, query, d, 0, q->sp->n));
    AS_TRY(as_query_lambda(q, q->knn, q->k));
    AS_TRY(as_query_score(q, tau));
    return as_query_finish(q, q->hits, q->topk, out_idx, out_score, out_len, out_lambda_q);
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
    AS_TRY(query_scan_impl(ws, nullptr, nullptr, row, sp->d, 0, sp->n, 0.0));
    AS_TRY(run_knn(ws, gp->eps, row, out_idx, out_key, out_dist, out_gy, out_cnt));
    return AS_OK;
}


}  // namespace as
