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
#include <chrono>
#include "as_query.hpp"

namespace as {

static std::atomic<int> g_search_stats{0};

// In-kernel phase stamps of the two single-block finish kernels (diagnostic build only: make STAMPS=1; no stamp
// executes in the product library).  100 MHz wall clock; slots 0..15 knn_finish, 16..31 score_finish.
#ifdef AS_STAMPS
__device__ unsigned long long g_stamps[32];
#define AS_STAMP(i)                                                           \
    do {                                                                      \
        if (threadIdx.x == 0 && blockIdx.z == 0) g_stamps[i] = wall_clock64(); \
    } while (0)
#define AS_STAMP_VAL(i, v) do { g_stamps[i] = (unsigned long long)(v); } while (0)
#else
#define AS_STAMP(i) do {} while (0)
#define AS_STAMP_VAL(i, v) do {} while (0)
#endif

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
#pragma clang fp contract(off)   // product and sum rounded separately: host_query_norm reproduces this sum bit for bit
        const double v = c < d ? qin[c] : 0.0;
        q64[c] = v;
        q32[c] = (float)v;
        const double sq = v * v;
        s = s + sq;
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

// The batched pass on the int8 images: the slots' staged fp32 queries quantised like the items (as_k2bf.hip, quant_i8_kernel) into
// the items' image layout -- a block per slot; idle slots (zero queries) get zero digits and scale 0.
__global__ __launch_bounds__(256) void q_quant_batch_kernel(const float* __restrict__ q32, int64_t dp, int64_t dp8, signed char* __restrict__ img,
                                                            float* __restrict__ faqv, const QInfo* __restrict__ info, float* __restrict__ stat_host) {
    __shared__ float sh[4], sh2[2][4];
    const float* x = q32 + (int64_t)blockIdx.x * dp;
    float m = 0.0f;
    for (int64_t c = threadIdx.x; c < dp; c += 256) m = fmaxf(m, fabsf(x[c]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    const float inv = m > 0.0f ? 16256.0f / m : 0.0f;
    signed char* dst = img + (int64_t)blockIdx.x * dp8 * 2;
    float st2 = 0.0f, sa2 = 0.0f;
    for (int64_t c = threadIdx.x; c < dp8; c += 256) {
        const float v = c < dp ? x[c] : 0.0f;
        const float sc = v * inv;
        int q = (int)rintf(sc);
        q = q > 16256 ? 16256 : (q < -16256 ? -16256 : q);
        const int a2 = ((q + 64 + (1 << 20)) & 127) - 64;
        const int a1 = (q - a2) >> 7;
        dst[(c >> 6) * 128 + (c & 63)] = (signed char)a1;
        dst[(c >> 6) * 128 + 64 + (c & 63)] = (signed char)a2;
        const float th = fabsf(sc - (float)q) + 0.004f;   // (the residue, with the product's rounding on top: quant_i8_kernel)
        st2 += th * th;
        sa2 += (float)(a2 * a2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        st2 += __shfl_xor(st2, o, 64);
        sa2 += __shfl_xor(sa2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sh2[0][threadIdx.x >> 6] = st2;
        sh2[1][threadIdx.x >> 6] = sa2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        faqv[blockIdx.x] = m * (11.313708498984761f / 16256.0f);   // s_q sqrt(128) / 16256
        // the slot's measured u_q = s_q |theta|_2 / (16256 |q|) and v_q = s_q |q2|_2 / (16256 |q|), for the host to hold against
        // the values it ASSUMED when it priced the pass's error (host_batch_coef; pinned memory, a slot per block: plain stores)
        const double nq = info[blockIdx.x].nq;
        float u = 0.0f, v = 0.0f;
        if (m > 0.0f) {
            const float den = 16256.0f * sqrtf((float)nq);
            u = m * sqrtf((sh2[0][0] + sh2[0][1]) + (sh2[0][2] + sh2[0][3])) / den * 1.002f;   // (fp32 sums of up to dp8 terms, rounded up generously)
            v = m * sqrtf((sh2[1][0] + sh2[1][1]) + (sh2[1][2] + sh2[1][3])) / den * 1.002f;
            if (!(nq > 0.0)) u = v = __int_as_float(0x7fc00000);
        }
        stat_host[2 * blockIdx.x] = u;
        stat_host[2 * blockIdx.x + 1] = v;
    }
}

// |q|^2 exactly as q_prepare_kernel forms it (256 strided partial sums, rounded products and sums, then the halving
// tree): the host-prepared fast path and the staged / batched paths must give a query the same norm, bit for bit.
// (No fma on either side: std::fma without -mfma is a library call, 6 us for 768 elements.)
static double host_query_norm(const double* q, int64_t d) {
    double p[256];
    for (int t = 0; t < 256; ++t) p[t] = 0.0;
    for (int64_t c = 0; c < d; ++c) {     // element c belongs to partial c % 256, visited in increasing c: the kernel's order
        volatile double sq = q[c] * q[c];   // rounded product, kept apart from the sum (no contraction whatever the flags)
        p[c & 255] = p[c & 255] + sq;
    }
    for (int o = 128; o > 0; o >>= 1)
        for (int t = 0; t < o; ++t) p[t] += p[t + o];
    return p[0];
}

// the per-search state of QInfo, cleared behind a finished search so that the next one starts without a launch for it
__global__ void reset_info_kernel(QInfo* info, unsigned int* sc_hist) {
    if (threadIdx.x == 0 && blockIdx.x == 0) reset_query_state(info);
    if (blockIdx.x == 0) reset_query_hist(sc_hist, threadIdx.x, blockDim.x);
}

__global__ void q_from_row_kernel(const float* __restrict__ x32, const double* __restrict__ x64, int64_t d, int64_t dp,
                                  int64_t row, double* __restrict__ qin) {
    for (int64_t c = threadIdx.x; c < d; c += blockDim.x) qin[c] = x64 ? x64[row * d + c] : (double)x32[row * dp + c];
}

// ------------------------------------------------------------------ scorer key (= -score) of one row
template <typename T>
struct SelArgs {
    const T* dots;
    const float* dots32;   // fp64 keys over the fp32 scan (SelArgs<double> only): the dot carries the fp32 error, nothing else does
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
    double margin;   // filter path: rows up to this far above the picked threshold are kept as well (the coarse chain: twice the keys' error)
    T* pkey;
    int* pidx;
    T* gmin;      // filter path: group minima, candidate buffers (CAND_CAP per slot)
    T* ckey;
    int* cidx;
    int64_t sd;   // dots: elements between slots inside a 32-row tile
    int64_t ts;   // dots: elements between 32-row tiles (32 = plain row order)
    int rs;       // dots: elements between rows of one slot inside a tile (SlotStride::dots_rs)
};

// The batched workspace keeps dots tile-major -- [32-row tile][slot][32 rows] -- so that the MFMA scan writes one
// contiguous 4 KiB block per row block instead of 32 segments 4 MiB apart; with ts = 32 this is plain row order.
template <typename T>
__device__ __forceinline__ T dot_at(const SelArgs<T>& a, int64_t row) {
    return a.dots[(row >> 5) * a.ts + (row & 31) * a.rs];
}

template <typename T>
__device__ __forceinline__ void sel_slot(SelArgs<T>& a) {
    const int z = blockIdx.z;
    if (a.dots) a.dots += dots_slot_off(z, a.sd, a.rs);
    if (a.dots32) a.dots32 += dots_slot_off(z, a.sd, a.rs);
    a.info += z;
    a.info_w += z;
    a.gmin += (int64_t)z * CAND_CAP;
    a.ckey += (int64_t)z * CAND_CAP;
    a.cidx += (int64_t)z * CAND_CAP;
}

struct ScoreCtx {
    double nq, tau, lq, rq;   // rq = 1/|q|
    float tau32, lq32, inq32;
};
__device__ __forceinline__ ScoreCtx load_ctx(const QInfo* info, double tau) {
    ScoreCtx c;
    c.nq = info->nq;
    c.rq = info->nq > 0.0 ? rsqrt(info->nq) : 0.0;
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
    if (a.dots32) {
        // mixed form: the dot of the fp32 scan, everything else in fp64 -- |key - exact| <= tau * coef32 (the proof's
        // bound), so rankings the lambda term decides (tau small) never depend on fp32.  No square root or division
        // for the cosine (reciprocal square roots: a few ulp, inside the bound's slack).
        const double nrow = a.n64[row];
        const double dv = (double)a.dots32[(row >> 5) * a.ts + (row & 31) * a.rs];
        const double cs = nrow > 0.0 ? dv * rsqrt(nrow) * c.rq : 0.0;
        return -(c.tau * cs + (1.0 - c.tau) / (1.0 + fabs(c.lq - a.lam64[row])));
    }
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
    return cosine_distance(c);
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
    // lambda_q == 0 is the reference's panic (src/lib.rs:156-159): no scores will be returned, none are formed (at tau = 0
    // such a query ties with every isolated item: thousands of candidates at the threshold for an answer nobody reads)
    if (KEY == 0 && a.info->status == AS_EZEROLAMBDA) return;
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

// (2) threshold: a value T with at least M group minima <= T (then at least M rows pass the filter, so the M best rows
// all do).  One pass: 1024 linear bins between the smallest and the largest finite minimum, the bin b in which the
// cumulative count reaches M, and T = the LARGEST minimum that fell into bins <= b -- valid by construction whatever
// the rounding of the bin arithmetic, and within one bin width of the exact M-th minimum (the 4-pass radix select it
// replaces took 7.7 us of every query).  Non-finite minima (NaN / inf keys of poisoned rows) are left out of the
// range and counted last.
// Threshold of a 1024-thread block over ng <= 4096 group minima: an upper bound of the M-th smallest, itself one of
// the minima (so at least M rows have keys <= it).  One pass: range, 1024-bin linear histogram, the bin holding the
// M-th, the largest minimum at or below that bin.  Every thread gets the result; +inf when fewer than M minima are
// finite (no finite threshold is provable: everything passes).  Deterministic: any block computes the same value.
template <typename T>
__device__ __forceinline__ double pick_thr_block(const T* __restrict__ gmin, int ng, int M) {
    __shared__ unsigned int lhist[1024];
    __shared__ double s_lo[16], s_hi[16];
    __shared__ int s_bin, s_nfin[16];
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const double big = 1.0e300;
    T v[4];
    bool fin[4];
    double lo = big, hi = -big;
    int nfin = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = tid + q * 1024;
        v[q] = i < ng ? gmin[i] : key_traits<T>::inf();
        fin[q] = i < ng && (double)v[q] > -big && (double)v[q] < big;
        if (fin[q]) {
            lo = (double)v[q] < lo ? (double)v[q] : lo;
            hi = (double)v[q] > hi ? (double)v[q] : hi;
            nfin += 1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
        nfin += __shfl_xor(nfin, o, 64);
    }
    if (lane == 0) {
        s_lo[w] = lo;
        s_hi[w] = hi;
        s_nfin[w] = nfin;
    }
    lhist[tid] = 0;
    __syncthreads();
    lo = s_lo[0];
    hi = s_hi[0];
    nfin = s_nfin[0];
    for (int w2 = 1; w2 < 16; ++w2) {
        lo = s_lo[w2] < lo ? s_lo[w2] : lo;
        hi = s_hi[w2] > hi ? s_hi[w2] : hi;
        nfin += s_nfin[w2];
    }
    if (nfin < M) return (double)key_traits<double>::inf();   // block-uniform
    const double scale = hi > lo ? 1024.0 / (hi - lo) : 0.0;
    int bin[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bin[q] = 1 << 20;
        if (fin[q]) {
            const int b = (int)(((double)v[q] - lo) * scale);
            bin[q] = b < 0 ? 0 : (b > 1023 ? 1023 : b);
            atomicAdd(&lhist[bin[q]], 1u);
        }
    }
    __syncthreads();
    if (tid < 64) {
        unsigned int h[16], tot = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            h[j] = lhist[16 * tid + j];
            tot += h[j];
        }
        unsigned int incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int t2 = __shfl_up(incl, o, 64);
            if (tid >= o) incl += t2;
        }
        const unsigned int excl = incl - tot;
        if (excl < (unsigned)M && incl >= (unsigned)M) {
            unsigned int run = excl;
            int b = 16 * tid;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                run += h[j];
                if (run >= (unsigned)M) break;
                b += 1;
            }
            s_bin = b;
        }
    }
    __syncthreads();
    const int bsel = s_bin;
    double mx = -big;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (fin[q] && bin[q] <= bsel) mx = (double)v[q] > mx ? (double)v[q] : mx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double m2 = __shfl_xor(mx, o, 64);
        mx = m2 > mx ? m2 : mx;
    }
    __syncthreads();   // s_hi has been read by everybody
    if (lane == 0) s_hi[w] = mx;
    __syncthreads();
    for (int w2 = 0; w2 < 16; ++w2) mx = s_hi[w2] > mx ? s_hi[w2] : mx;
    return mx;   // one of the minima: exact in T
}

template <typename T>
__global__ __launch_bounds__(1024) void pick_thr_kernel(const T* __restrict__ gmin, int ng, int M, QInfo* info) {
    gmin += (int64_t)blockIdx.z * CAND_CAP;
    info += blockIdx.z;
    const double thr = pick_thr_block<T>(gmin, ng, M);
    if (threadIdx.x == 0) {
        if (sizeof(T) == 4) info->thr32 = (float)thr;
        else info->thr64 = thr;
    }
}

// (2) + (3) threshold and filter in one launch (single-query chain: one dependent launch and one single-block kernel less).
// Every block of 1024 threads derives the threshold from the group minima itself -- 32 KiB out of L2, the same value
// in every block -- and filters its share of the rows.
template <typename T, int KEY>
__global__ __launch_bounds__(1024) void score_pickfilter_kernel(SelArgs<T> a, int ng, int M) {
    sel_slot(a);
    if (KEY == 0 && a.info->status == AS_EZEROLAMBDA) return;   // (as score_gmin_kernel: an empty candidate list for the finish kernel)
    const double thr_d = pick_thr_block<T>(a.gmin, ng, M) + a.margin;
    const T thr = (T)thr_d;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (sizeof(T) == 4) a.info_w->thr32 = (float)thr_d;
        else a.info_w->thr64 = thr_d;
    }
    T* __restrict__ ckey = a.ckey;
    int* __restrict__ cidx = a.cidx;
    int* counter = KEY ? &a.info_w->knn_cnt : &a.info_w->sc_cnt;
    const ScoreCtx c = load_ctx(a.info, a.tau);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // The rows a block keeps are gathered in LDS and reach the candidate buffer through ONE returning atomic per block: the
    // counter is a single word every XCD has to reach -- a thousand returning atomics on it, one per kept row, were 45 of this
    // kernel's 50 us (a coarse chain keeps a whole cluster inside its margin).  Mass ties at the threshold (tau = 0 with
    // thousands of isolated items at lambda = 0): once the counter has passed the capacity nobody needs a slot any more.
    constexpr int LCAP = 1024;
    __shared__ T l_key[LCAP];
    __shared__ int l_idx[LCAP];
    __shared__ int s_ln, s_gbase;
    if (threadIdx.x == 0) s_ln = 0;
    __syncthreads();
    const bool full0 = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > CAND_CAP;   // (block-uniform enough: a late reader only does needless work)
    for (int64_t row = a.r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < a.r1 && !full0; row += stride) {
        const T k = sel_key<T, KEY>(a, c, row);
        if (k <= thr && (KEY == 0 || k < key_traits<T>::inf())) {
            const int ls = atomicAdd(&s_ln, 1);
            if (ls < LCAP) {
                l_key[ls] = k;
                l_idx[ls] = (int)row;
            } else {
                // (more than the block's list holds: straight to the buffer, as every row did before)
                const int slot = atomicAdd(counter, 1);
                if (slot < CAND_CAP) {
                    ckey[slot] = k;
                    cidx[slot] = (int)row;
                }
            }
        }
    }
    __syncthreads();
    const int ln = s_ln < LCAP ? s_ln : LCAP;
    if (threadIdx.x == 0 && ln > 0) s_gbase = atomicAdd(counter, ln);
    __syncthreads();
    if (ln > 0) {
        const int gb = s_gbase;
        for (int t = threadIdx.x; t < ln; t += blockDim.x)
            if (gb + t < CAND_CAP) {
                ckey[gb + t] = l_key[t];
                cidx[gb + t] = l_idx[t];
            }
    }
}

// ------------------------------------------------------------------ batched selection: all query slots per read of a row
// The per-slot kernels above read a row's norm and lambda once per slot: 32 slots x 20 B per row.  Here a thread owns
// a row, reads its norm and lambda once and walks the NS slots' dots (tile-major: [32-row tile][slot][32 rows], 128
// contiguous bytes per slot and half-wave) -- 4 NS + 16 bytes per row.  Mixed keys (fp32 dots, fp64 everything else).
struct BatchSel {
    const float* dots32;      // slot 0
    const double* n64;
    const double* lam64;
    const float* lam32;       // (half: the keys are coarse anyway -- fp32 arithmetic, 4 bytes of lambda per row)
    const QInfo* info;        // [NS]
    QInfo* info_w;
    int64_t r0, r1, sd, ts;
    int rs;                   // SlotStride::dots_rs
    int half;                 // the scan left fp16 cosines in the dots' places (as_query::dots_half), not fp32 dots: 1 = the keys in
                              // fp32 too, 2 = the lambda term in fp64 (small tau: the scores are all lambda term, an fp32 one costs the proof)
    double tau;
    double* gmin;             // [NS][CAND_CAP]
    double* ckey;             // [NS][CAND_CAP]
    int* cidx;
    int ns;
};

// wave-wide minimum on the DPP crossbar (as wave_sum_dpp of as_scan.hip: no LDS round trips); the same value in every lane
__device__ __forceinline__ float wave_min_dpp(float v) {
#define AS_MIN_DPP(ctrl, rmask) v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rmask, 0xF, false)))
    AS_MIN_DPP(0xB1, 0xF);    // quad_perm [1,0,3,2]
    AS_MIN_DPP(0x4E, 0xF);    // quad_perm [2,3,0,1]
    AS_MIN_DPP(0x141, 0xF);   // row_half_mirror
    AS_MIN_DPP(0x140, 0xF);   // row_mirror
    AS_MIN_DPP(0x142, 0xA);   // row_bcast15 -> rows 1, 3
    AS_MIN_DPP(0x143, 0xC);   // row_bcast31 -> rows 2, 3
#undef AS_MIN_DPP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// the coarse key of the fp16-cosine workspace, in fp32 (see batch_key)
__device__ __forceinline__ float batch_key32(float t32, float cs, float lrow, float lq) {
    return -(t32 * cs + (1.0f - t32) * __builtin_amdgcn_rcpf(1.0f + fabsf(lq - lrow)));
}
// rn = 1/|x_row|, rq = 1/|q_s|: no square root or division for the cosine; one reciprocal for the lambda term
__device__ __forceinline__ double batch_key(const BatchSel& a, float dot, double rn, double lrow, double rq, double lq) {
    if (a.half == 1) {
        // the stored value is the cosine itself, good to 2^-11: the rest in fp32 (a dozen roundings of values below 1 and a
        // reciprocal good to an ulp: under 1e-6 in all, launch_score's e_key32)
        return (double)batch_key32((float)a.tau, dot, (float)lrow, (float)lq);
    }
    const double cs = a.half ? (double)dot : (double)dot * rn * rq;
    return -(a.tau * cs + (1.0 - a.tau) / (1.0 + fabs(lq - lrow)));
}
// the dots of slots [s0, s0 + NSW) of one row: two dwordx4 in the batched workspace's [slot quad][32 rows][4] tiles
template <int NSW>
__device__ __forceinline__ void batch_dots(const BatchSel& a, int s0, int64_t row, float (&dv)[NSW]) {
    const float* __restrict__ t = a.dots32 + (row >> 5) * a.ts;
    if (a.half) {   // [slot quad][32 rows][4 slots] of fp16: a row's four slots are one 8-byte load
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const _Float16* __restrict__ th = (const _Float16*)a.dots32 + (row >> 5) * a.ts;
#pragma unroll
        for (int g = 0; g < NSW / 4; ++g) {
            const h4 v = *(const h4*)(th + ((s0 >> 2) + g) * 128 + (row & 31) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) dv[4 * g + e] = (float)v[e];
        }
    } else if (a.rs == 4) {
#pragma unroll
        for (int g = 0; g < NSW / 4; ++g) {
            const f32x4 v = *(const f32x4*)(t + ((s0 >> 2) + g) * 128 + (row & 31) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) dv[4 * g + e] = v[e];
        }
    } else {
#pragma unroll
        for (int s = 0; s < NSW; ++s) dv[s] = t[(int64_t)(s0 + s) * a.sd + (row & 31)];
    }
}

// group minima: one wave per (group of G rows, NSW slots); blockIdx.y = slot octet
template <int NSW>
__global__ __launch_bounds__(256) void score_gmin_batch_kernel(BatchSel a, int64_t G, int ngroups) {
    __shared__ double s_rq[NSW], s_lq[NSW];
    const int s0 = blockIdx.y * NSW;
    if (threadIdx.x < NSW) {
        const double nq = a.info[s0 + threadIdx.x].nq;
        s_rq[threadIdx.x] = nq > 0.0 ? rsqrt(nq) : 0.0;
        s_lq[threadIdx.x] = a.info[s0 + threadIdx.x].lambda_q;
    }
    __syncthreads();
    const int lane = lane_id();
    const int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= ngroups) return;
    const int64_t lo = a.r0 + g * G;
    const int64_t hi = lo + G < a.r1 ? lo + G : a.r1;
    if (a.half == 1) {
        // fp16 cosines: fp32 keys, fp32 minima, the wave's minimum over the DPP crossbar (the fp64 butterfly below is 96
        // ds_bpermute per wave -- as long as the wave's four trips over its rows)
        float m32[NSW];
        const float t32 = (float)a.tau;
        float lq32[NSW];
#pragma unroll
        for (int s = 0; s < NSW; ++s) {
            m32[s] = __int_as_float(0x7f800000);
            lq32[s] = (float)s_lq[s];
        }
        for (int64_t row = lo + lane; row < hi; row += 64) {
            const float lrow = a.lam32[row];
            float dv[NSW];
            batch_dots<NSW>(a, s0, row, dv);
#pragma unroll
            for (int s = 0; s < NSW; ++s) m32[s] = fminf(m32[s], batch_key32(t32, dv[s], lrow, lq32[s]));
        }
#pragma unroll
        for (int s = 0; s < NSW; ++s) {
            const float v = wave_min_dpp(m32[s]);
            if (lane == s && s0 + s < a.ns) a.gmin[(int64_t)(s0 + s) * CAND_CAP + g] = (double)v;
        }
        return;
    }
    double m[NSW];
#pragma unroll
    for (int s = 0; s < NSW; ++s) m[s] = key_traits<double>::inf();
    for (int64_t row = lo + lane; row < hi; row += 64) {
        const double lrow = a.half == 1 ? (double)a.lam32[row] : a.lam64[row];
        double rn = 0.0;
        if (!a.half) {
            const double nrow = a.n64[row];
            rn = nrow > 0.0 ? rsqrt(nrow) : 0.0;
        }
        float dv[NSW];
        batch_dots<NSW>(a, s0, row, dv);
#pragma unroll
        for (int s = 0; s < NSW; ++s) {
            const double k = batch_key(a, dv[s], rn, lrow, s_rq[s], s_lq[s]);
            m[s] = k < m[s] ? k : m[s];
        }
    }
#pragma unroll
    for (int s = 0; s < NSW; ++s) {
        double v = m[s];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double other = __shfl_xor(v, o, 64);
            v = other < v ? other : v;
        }
        if (lane == s && s0 + s < a.ns) a.gmin[(int64_t)(s0 + s) * CAND_CAP + g] = v;
    }
}

template <int NSW>
__global__ __launch_bounds__(256) void score_filter_batch_kernel(BatchSel a, int64_t G) {
    __shared__ double s_rq[NSW], s_lq[NSW], s_thr[NSW];
    const int s0 = blockIdx.y * NSW;
    if (threadIdx.x < NSW) {
        const double nq = a.info[s0 + threadIdx.x].nq;
        s_rq[threadIdx.x] = nq > 0.0 ? rsqrt(nq) : 0.0;
        s_lq[threadIdx.x] = a.info[s0 + threadIdx.x].lambda_q;
        // idle slots never pass
        s_thr[threadIdx.x] = s0 + (int)threadIdx.x < a.ns ? a.info[s0 + threadIdx.x].thr64 : -key_traits<double>::inf();
    }
    __syncthreads();
    unsigned int full = 0;   // bit s: the slot has overflowed its candidate buffer (mass ties): stop adding to its counter
    // A wave takes 64 consecutive rows -- one group: G is a multiple of 64 -- and asks ONCE which of its NSW slots can still
    // take rows of that group (lanes 0 .. NSW-1 read the slots' group minima, a ballot spreads the answer): a group whose
    // minimum key is above a slot's threshold holds nothing for that slot, and for nearly every (group, slot octet) nothing at
    // all -- the wave moves on without touching the rows.  (Every thread used to read the NSW minima for its own row.)
    const int lane = lane_id();
    const int64_t nchunk = (a.r1 - a.r0 + 63) / 64;
    const int64_t nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t ch = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); ch < nchunk; ch += nwave) {
        const int64_t g = (ch * 64) / G;
        const bool mine = lane < NSW && a.gmin[(int64_t)(s0 + lane) * CAND_CAP + g] <= s_thr[lane < NSW ? lane : 0];
        const unsigned int actm = (unsigned int)__ballot(mine);
        if (!actm) continue;
        const int64_t row = a.r0 + ch * 64 + lane;
        if (row >= a.r1) continue;
        const double lrow = a.half == 1 ? (double)a.lam32[row] : a.lam64[row];
        double rn = 0.0;
        if (!a.half) {
            const double nrow = a.n64[row];
            rn = nrow > 0.0 ? rsqrt(nrow) : 0.0;
        }
        float dv[NSW];
        batch_dots<NSW>(a, s0, row, dv);
#pragma unroll
        for (int s = 0; s < NSW; ++s) {
            if (!((actm >> s) & 1u)) continue;
            const double k = batch_key(a, dv[s], rn, lrow, s_rq[s], s_lq[s]);
            if (k <= s_thr[s] && !((full >> s) & 1u)) {
                if (__hip_atomic_load(&a.info_w[s0 + s].sc_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > CAND_CAP) {   // (as score_pickfilter_kernel)
                    full |= 1u << s;
                    continue;
                }
                const int slot = atomicAdd(&a.info_w[s0 + s].sc_cnt, 1);
                if (slot < CAND_CAP) {
                    a.ckey[(int64_t)(s0 + s) * CAND_CAP + slot] = k;
                    a.cidx[(int64_t)(s0 + s) * CAND_CAP + slot] = (int)row;
                } else {
                    full |= 1u << s;
                }
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
// W = 1: lists of 64 slots (WaveList), W = 2: of 128 (WaveList2; k above 56)
template <typename T, int W>
struct wave_list_of { typedef WaveList<T> type; };
template <typename T>
struct wave_list_of<T, 2> { typedef WaveList2<T> type; };
template <typename T>
__device__ __forceinline__ void list_store(const WaveList<T>& l, T* k, int* i, size_t at, int lane) {
    k[at * 64 + lane] = l.key;
    i[at * 64 + lane] = l.idx;
}
template <typename T>
__device__ __forceinline__ void list_store(const WaveList2<T>& l, T* k, int* i, size_t at, int lane) {
    k[at * 128 + lane] = l.key;
    i[at * 128 + lane] = l.idx;
    k[at * 128 + 64 + lane] = l.key2;
    i[at * 128 + 64 + lane] = l.idx2;
}

template <typename T, int W>
__global__ __launch_bounds__(256) void knn_partial_kernel(SelArgs<T> a) {
    const int lane = lane_id();
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    const double nq = a.info->nq;
    typename wave_list_of<T, W>::type lst;
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
    list_store(lst, a.pkey, a.pidx, (size_t)gw, lane);
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

// merge nlists partial lists (64 W slots each) down to one sorted list in LDS (fk, fi)
template <typename T, int W = 1>
__device__ __forceinline__ void merge_partials(const T* pkey, const int* pidx, int nlists, int M, T* wk, int* wi, T* fk,
                                               int* fi, int* fcount) {
    const int lane = lane_id(), w = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    typename wave_list_of<T, W>::type lst;
    lst.init();
    for (int p = w; p < nlists; p += nwv) {
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const T k = pkey[(size_t)p * 64 * W + 64 * u + lane];
            const int i = pidx[(size_t)p * 64 * W + 64 * u + lane];
            lst.offer(M, k, i, i != 0x7fffffff);
        }
    }
    list_store(lst, wk, wi, (size_t)w, lane);
    __syncthreads();
    if (w == 0) {
        typename wave_list_of<T, W>::type fin;
        fin.init();
        for (int p = 0; p < nwv; ++p) {
#pragma unroll
            for (int u = 0; u < W; ++u) {
                const T k = wk[p * 64 * W + 64 * u + lane];
                const int i = wi[p * 64 * W + 64 * u + lane];
                fin.offer(M, k, i, i != 0x7fffffff);
            }
        }
        list_store(fin, fk, fi, 0, lane);
        int c = 0;
#pragma unroll
        for (int u = 0; u < W; ++u) c += __popcll(__ballot(64 * u + lane < M && fi[64 * u + lane] != 0x7fffffff));
        if (lane == 0) *fcount = c;
    }
    __syncthreads();
}

constexpr int PRUNE_CAP = 2048;  // compact list of the radix-pruned candidates
constexpr int Q_LDS_MAX = 4096;  // widest query (padded floats) the finish kernels keep in LDS (beyond: read from where it is staged)

// rank-select the M smallest (key, idx) of the C buffered candidates into (fk, fi), sorted.
// Large C (dense neighbourhoods) is first pruned to the candidates at or below the M-th
// smallest key by a radix select, so the quadratic ranking only ever sees ~M entries.
// ckey == nullptr: the caller has filled (sk, si) itself.
template <typename T>
__device__ __forceinline__ void select_candidates(const T* ckey, const int* cidx, int C, int M, T* sk, int* si, T* pk, int* pi,
                                                  T* fk, int* fi, int* fcount) {
    typedef typename ord_of<T>::type U;
    __shared__ unsigned int hist[256];
    __shared__ U s_prefix;
    __shared__ int s_rank, s_cnt;
    if (ckey)
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
    // a list as wide as the candidate set (topk above the number of rows that passed) keeps everything: a k-th
    // smallest beyond the set does not exist, and a select would return an arbitrary threshold
    if (C > 128 && M < C) {
        // One-pass prune: 1024 linear bins between the smallest and the largest key; the bin in which the cumulative
        // count reaches M bounds the M smallest (monotone binning), and the quadratic ranking below only sees that
        // prefix -- about M entries unless the keys pile up in one bin, which the radix select then handles.
        // (The ranking over all C was 2.5 us at C = 500, the 4-pass radix select 8 us: most of knn_finish.)
        __shared__ unsigned int lhist[1024];
        __shared__ double s_mm[2][16];
        __shared__ int s_bin;
        T v[4];
        bool have[4];
        double lo = 1.0e300, hi = -1.0e300;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = threadIdx.x + q * 1024;
            have[q] = i < C;
            v[q] = have[q] ? sk[i] : (T)0;
            if (have[q]) {
                lo = (double)v[q] < lo ? (double)v[q] : lo;
                hi = (double)v[q] > hi ? (double)v[q] : hi;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        if (lane_id() == 0) {
            s_mm[0][threadIdx.x >> 6] = lo;
            s_mm[1][threadIdx.x >> 6] = hi;
        }
        lhist[threadIdx.x] = 0;
        __syncthreads();
        {
            const int nwv = blockDim.x >> 6;
            lo = s_mm[0][0];
            hi = s_mm[1][0];
            for (int w2 = 1; w2 < nwv; ++w2) {
                lo = s_mm[0][w2] < lo ? s_mm[0][w2] : lo;
                hi = s_mm[1][w2] > hi ? s_mm[1][w2] : hi;
            }
        }
        const double scale = hi > lo ? 1024.0 / (hi - lo) : 0.0;
        int bin[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bin[q] = 0;
            if (have[q]) {
                const int b = (int)(((double)v[q] - lo) * scale);
                bin[q] = b < 0 ? 0 : (b > 1023 ? 1023 : b);
                atomicAdd(&lhist[bin[q]], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            // wave 0: 16 bins per lane, inclusive scan over lanes, the lane whose range reaches M finishes
            unsigned int h[16], tot = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                h[j] = lhist[16 * threadIdx.x + j];
                tot += h[j];
            }
            unsigned int incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int t2 = __shfl_up(incl, o, 64);
                if ((int)threadIdx.x >= o) incl += t2;
            }
            const unsigned int excl = incl - tot;
            if (excl < (unsigned)M && incl >= (unsigned)M) {
                unsigned int run = excl;
                int b = 16 * threadIdx.x;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    run += h[j];
                    if (run >= (unsigned)M) break;
                    b += 1;
                }
                s_bin = b;
            }
        }
        __syncthreads();
        const int bsel = s_bin;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (have[q] && bin[q] <= bsel) {
                const int slot = atomicAdd(&s_cnt, 1);
                if (slot < PRUNE_CAP) {
                    const int i = threadIdx.x + q * 1024;
                    pk[slot] = v[q];
                    pi[slot] = si[i];
                }
            }
        }
        __syncthreads();
        if (s_cnt <= PRUNE_CAP) {
            rk = pk;
            ri = pi;
            R = s_cnt;
        }
        __syncthreads();   // everybody has read s_cnt before it is reused
    }
    if (R > PRUNE_CAP && M < C) {
        // the linear bins did not separate the keys (mass ties, outliers): exact radix select
        if (threadIdx.x == 0) s_cnt = 0;
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
            // twelve 64-float chunks (a whole 768-float row) per trip, every load issued before the first use: one
            // chunk per trip was a chain of dp/64 dependent HBM round trips and most of these kernels' time
            for (int64_t e0 = 4 * sub; e0 < dp; e0 += 768) {
                f32x4 v[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int64_t e = e0 + 64 * u;
                    v[u] = e < dp ? *(const f32x4*)(pj + e) : f32x4{0, 0, 0, 0};
                }
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int64_t e = e0 + 64 * u;
                    if (e < dp) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const double a = q64[e + t], b = (double)v[u][t], df = a - b;
                            s += df * df;
                            g += a * b;
                        }
                    }
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
    int64_t goff;       // global id of local row 0 (a shard of a row-sharded index); graph arrays and records use global ids
    int exhaustive;     // evaluate EVERY buffered candidate in fp64 (near-ties at the k-th distance that fp32 cannot order)
    // build-fallback outputs (row-list form); null for searches
    int32_t* o_idx;
    double* o_key;
    double* o_dist;
    double* o_gy;
    int32_t* o_cnt;
    int* unproven;      // build fallback: counts the rows whose list failed the a-posteriori check even in fp64
    int auto_reset;     // score_finish (fused publish, single query): clear QInfo's per-search state behind a clean search
    int sc_nw;          // fused tail: waves of the scan, each with a report of SC_WCAP words in ci (count, then rows)
    const unsigned int* sc_hist = nullptr;   // ... and, where the tail validates lossy reports (PreArgs::sc_late): the scan's histogram as it
    int sc_m = 0;                            // ended, the rows its bound needs and the window below it
    float sc_w = 0.0f;
};

// SPEC S10 given the selected neighbours in LDS in (key, index) rank order; one wave, lane t
// owns neighbour t, sums are fixed-order butterflies (deterministic).
__device__ __forceinline__ void lambda_from_sorted(int cnt, const double* s_dist, const double* s_gy, const double* s_deg,
                                                   const double* s_ny, int metric, int kernel, double sigma, double p,
                                                   double tau0, QInfo* info) {
    // lane t owns neighbours t and 64 + t (k up to 120); every sum adds a lane's two terms, then runs the fixed butterfly
    const int lane = lane_id();
    const double nq = info->nq;
    const double nyq = metric == AS_METRIC_L2 ? nq : (nq > 0.0 ? 1.0 : 0.0);
    const bool on0 = lane < cnt, on1 = 64 + lane < cnt;
    const double at0 = on0 ? edge_weight(s_dist[lane], sigma, p, kernel) : 0.0;
    const double at1 = on1 ? edge_weight(s_dist[64 + lane], sigma, p, kernel) : 0.0;
    const double degq = wave_sum(at0 + at1);
    double lam = 0.0;
    if (cnt > 0 && nyq > 0.0 && degq > 0.0) {
        double ev0 = 0.0, ev1 = 0.0;
        if (on0) ev0 = edge_energy(at0, metric, s_dist[lane], s_gy[lane], degq, s_deg[lane] + at0, nyq, s_ny[lane]);
        if (on1) ev1 = edge_energy(at1, metric, s_dist[64 + lane], s_gy[64 + lane], degq, s_deg[64 + lane] + at1, nyq, s_ny[64 + lane]);
        const double S = wave_sum(ev0 + ev1);
        const double Eq = 0.5 * S / nyq;
        double Gq = 0.0;
        if (S > 0.0) {
            const double r0 = ev0 / S, r1 = ev1 / S;
            Gq = wave_sum(r0 * r0 + r1 * r1);
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
// qx_pre: the query already staged in LDS by the caller (the fused tail), else staged here behind the work area.
// Every thread returns from the body (the fused kernel goes on behind it): wave 0 does the ranking and lambda_q alone.
template <typename T>
__device__ __forceinline__ void knn_finish_body(FinishArgs a, char* smem, const double* qx_pre) {
    AS_STAMP(0);
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
    // the query, read dozens of times by the exact evaluation: once from HBM into LDS (visible behind the barriers
    // of the candidate selection)
    const double* qx = qx_pre ? qx_pre : a.q64;
    if (!qx_pre && a.dp <= Q_LDS_MAX) {
        double* qs = (double*)(pi + PRUNE_CAP) + CAND_CAP;
        for (int64_t c = threadIdx.x; c < a.dp; c += blockDim.x) qs[c] = a.q64[c];
        qx = qs;
    }
    constexpr int KL = MAX_KLIST;   // widest list: 128 candidates (k up to 120), two per lane where a wave owns the list
    __shared__ T fk[KL];
    __shared__ int fi[KL];
    __shared__ double ek[KL], ed[KL], eg[KL], sk2[KL];
    __shared__ double l_dist[KL], l_gy[KL], l_deg[KL], l_ny[KL];
    __shared__ int fcount;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    int total;
    if (a.from_list) {
        if (a.M > 64) merge_partials<T, 2>(ckey, cidx, a.nlists, a.M, sk, si, fk, fi, &fcount);
        else merge_partials<T, 1>(ckey, cidx, a.nlists, a.M, sk, si, fk, fi, &fcount);
        total = a.info->knn_total;
    } else {
        const int raw = a.info->knn_cnt;
        if (raw > CAND_CAP && threadIdx.x == 0) a.info->overflow |= 1;
        total = raw < CAND_CAP ? raw : CAND_CAP;
        // (the exhaustive evaluation below ranks every candidate by its exact key: no fp32 selection in front of it)
        const bool exh = a.exhaustive && !a.thresholded && raw <= CAND_CAP && total > a.M;
        if (!exh) select_candidates<T>(ckey, cidx, total, a.M, sk, si, pk, pi, fk, fi, &fcount);
    }
    AS_STAMP(1);
    const double nq = a.info->nq;
    bool complete = false;   // the list below was chosen among ALL candidates by their exact keys: nothing to prove
    if (a.exhaustive && !a.from_list && !a.thresholded && a.info->knn_cnt <= CAND_CAP && total > a.M) {
        // Every row inside the eps bound is in the buffer.  Evaluate them all exactly, 64 per round, and keep the
        // (up to) KL smallest by (key64, index): the answer cannot depend on how fp32 ordered near-ties.
        double* xk = (double*)(pi + PRUNE_CAP);   // CAND_CAP exact keys
        __shared__ double t_sq[64], t_dot[64];
        __shared__ int s_npass;
        if (threadIdx.x == 0) s_npass = 0;
        for (int t = threadIdx.x; t < total; t += blockDim.x) si[t] = cidx[t];
        __syncthreads();
        for (int base = 0; base < total; base += 64) {
            const int m = total - base < 64 ? total - base : 64;
            exact_eval_all(a.x32, a.x64, qx, a.d, a.dp, si + base, m, t_sq, t_dot);
            __syncthreads();
            if (threadIdx.x < m) {
                double kk = t_sq[threadIdx.x];
                if (a.metric != AS_METRIC_L2) {
                    const double den = sqrt(nq * a.n64[si[base + threadIdx.x]]);
                    kk = cosine_distance(den > 0.0 ? t_dot[threadIdx.x] / den : 0.0);
                }
                xk[base + threadIdx.x] = kk;
            }
            __syncthreads();
        }
        int np_l = 0;
        for (int t = threadIdx.x; t < total; t += blockDim.x) {
            const double kk = xk[t];
            if (!(kk <= a.epskey)) continue;
            np_l += 1;
            int rank = 0;
            for (int s2 = 0; s2 < total; ++s2) rank += lex_less<double>(xk[s2], si[s2], kk, si[t]) ? 1 : 0;
            if (rank < KL) fi[rank] = si[t];
        }
        if (np_l) atomicAdd(&s_npass, np_l);
        __syncthreads();
        if (threadIdx.x == 0) fcount = s_npass < KL ? s_npass : KL;
        __syncthreads();
        complete = true;
    }
    const int Mp = fcount;
    // the selected candidates' degrees / norms: in flight under the exact evaluation instead of behind it
    double pre_deg[2] = {0.0, 0.0}, pre_ny[2] = {0.0, 0.0};
    if (w == 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (64 * u + lane < Mp) {
                pre_deg[u] = a.deg ? a.deg[fi[64 * u + lane] + a.goff] : 0.0;
                pre_ny[u] = a.ny ? a.ny[fi[64 * u + lane] + a.goff] : 0.0;
            }
    }
    exact_eval_all(a.x32, a.x64, qx, a.d, a.dp, fi, Mp < 64 ? Mp : 64, ek, eg);
    if (Mp > 64) exact_eval_all(a.x32, a.x64, qx, a.d, a.dp, fi + 64, Mp - 64, ek + 64, eg + 64);   // (block-uniform)
    __syncthreads();
    AS_STAMP(2);
    if ((int)threadIdx.x < Mp) {
        const int t = threadIdx.x;
        const double sq = ek[t], dot = eg[t];
        if (a.metric == AS_METRIC_L2) {
            ed[t] = sqrt(sq);
        } else {
            const double den = sqrt(nq * a.n64[fi[t]]);
            const double c = den > 0.0 ? dot / den : 0.0;
            const double dd = cosine_distance(c);
            ek[t] = dd;
            ed[t] = dd;
            eg[t] = c;
        }
    }
    __syncthreads();
    if (w != 0) return;   // (of the body: the other waves have nothing to do with the ranking)
    AS_STAMP(3);
    // rank by (key64, idx): candidates lane and 64 + lane per lane
    bool have[2], sel[2];
    double myk[2], mydeg[2], myny[2];
    int myi[2], rank[2];
    int npass = 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = 64 * u + lane;
        have[u] = t < Mp;
        myk[u] = have[u] ? ek[t] : 0.0;
        myi[u] = have[u] ? fi[t] : 0x7fffffff;
        rank[u] = 0;
        if (64 * u < Mp)   // (wave-uniform: lists of up to 64 candidates skip the second half altogether)
            for (int s = 0; s < Mp; ++s) rank[u] += lex_less<double>(ek[s], fi[s], myk[u], myi[u]) ? 1 : 0;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (have[u]) sk2[rank[u]] = myk[u];
        npass += __popcll(__ballot(have[u] && myk[u] <= a.epskey));
    }
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
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = 64 * u + lane;
        sel[u] = have[u] && myk[u] <= a.epskey && rank[u] < a.k;
        mydeg[u] = sel[u] ? pre_deg[u] : 0.0;
        myny[u] = sel[u] ? pre_ny[u] : 0.0;
        if (sel[u]) {
            if (a.recs) {
                as_knn_rec r;
                r.idx = myi[u] + a.goff;
                r.key = myk[u];
                r.dist = ed[t];
                r.gy = eg[t];
                r.deg = mydeg[u];
                r.ny = myny[u];
                a.recs[rank[u]] = r;
            }
            if (a.o_idx) {
                a.o_idx[rank[u]] = myi[u];
                a.o_key[rank[u]] = myk[u];
                a.o_dist[rank[u]] = ed[t];
                a.o_gy[rank[u]] = eg[t];
            }
        }
    }
    AS_STAMP(4);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (a.fuse && sel[u]) {
            // order of the sums below = (key64, index) rank: fixed by the data alone, and the same in q_lambda_kernel
            const int t = 64 * u + lane;
            l_dist[rank[u]] = ed[t];
            l_gy[rank[u]] = eg[t];
            l_deg[rank[u]] = mydeg[u];
            l_ny[rank[u]] = myny[u];
        }
    }
    AS_LDS_FENCE();
    if (lane == 0) {
        if (a.o_cnt) *a.o_cnt = cnt;
        int bad = 0;
        a.info->knn_total = total;
        if ((total > a.M || a.thresholded) && Mp > 0 && !complete) {
            const double B = npass >= a.k ? sk2[a.k - 1] : a.epskey;
            // dropped items' norms are unknown: bound them by the largest norm in the space
            const double e = a.metric == AS_METRIC_L2 ? a.coef * (a.nmax + nq) : a.coef;
            const double Tm = (double)fk[Mp - 1];
            bad = !(Tm - e > B);
        }
        a.info->knn_inexact = bad;
        if (bad && a.unproven) atomicAdd(a.unproven, 1);
    }
    AS_STAMP(5);
    if (a.fuse) lambda_from_sorted(cnt, l_dist, l_gy, l_deg, l_ny, a.metric, a.kernel, a.sigma, a.p, a.tau0, a.info);
    AS_STAMP(6);
}

template <typename T>
__global__ __launch_bounds__(1024) void knn_finish_kernel(FinishArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    knn_finish_body<T>(a, smem, nullptr);
}

// SPEC S10 from m candidate records (this shard's, or all shards' all-gathered).  blockIdx.x = query slot: the
// all-gathered buffer is [rank][slot][per] records, a slot's m records are `per` from every rank, `rstride` apart.
// (one wave; `slot` = the query slot, blockIdx.x of q_lambda_kernel)
__device__ __forceinline__ void q_lambda_body(const as_knn_rec* __restrict__ recs_all, int64_t m, int64_t per, int64_t rstride, int64_t k, int metric,
                                              int kernel, double sigma, double p, double tau0, QInfo* info, int slot) {
    info += slot;
    const as_knn_rec* __restrict__ recs0 = recs_all + (int64_t)slot * per;
#define recs_at(t) recs0[((t) / per) * rstride + ((t) % per)]
    __shared__ double r_key[REC_CAP];
    __shared__ int r_idx[REC_CAP];
    __shared__ double l_dist[MAX_KLIST], l_gy[MAX_KLIST], l_deg[MAX_KLIST], l_ny[MAX_KLIST];
    __shared__ int l_pos[MAX_KLIST];
    const int lane = lane_id();
    const int mm = (int)(m < REC_CAP ? m : REC_CAP);
    for (int t = lane; t < mm; t += 64) {
        const as_knn_rec r = recs_at(t);
        const bool valid = r.idx >= 0;
        r_key[t] = valid ? r.key : key_traits<double>::inf();
        r_idx[t] = valid ? (int)r.idx : 0x7fffffff;
    }
    AS_LDS_FENCE();
    // rank every record by (key, idx); the k best valid ones are the neighbours
    int cnt_l = 0;
    for (int t = lane; t < mm; t += 64) {
        if (r_idx[t] == 0x7fffffff) continue;
        int rank = 0;
        for (int s = 0; s < mm; ++s) rank += lex_less<double>(r_key[s], r_idx[s], r_key[t], r_idx[t]) ? 1 : 0;
        if (rank < k && rank < MAX_KLIST) {
            l_pos[rank] = t;
            cnt_l += 1;
        }
    }
    const int cnt = wave_sum(cnt_l);
    AS_LDS_FENCE();
    // lane order = (key, index) rank, as in knn_finish_kernel
    for (int t = lane; t < cnt; t += 64) {
        const as_knn_rec r = recs_at(l_pos[t]);
        l_dist[t] = r.dist;
        l_gy[t] = r.gy;
        l_deg[t] = r.deg;
        l_ny[t] = r.ny;
    }
#undef recs_at
    AS_LDS_FENCE();
    lambda_from_sorted(cnt, l_dist, l_gy, l_deg, l_ny, metric, kernel, sigma, p, tau0, info);
}

__global__ __launch_bounds__(64) void q_lambda_kernel(const as_knn_rec* __restrict__ recs_all, int64_t m, int64_t per, int64_t rstride,
                                                      int64_t k, int metric, int kernel, double sigma, double p, double tau0, QInfo* info, int zero_sc) {
    q_lambda_body(recs_all, m, per, rstride, k, metric, kernel, sigma, p, tau0, info, blockIdx.x);
    // (the coarse chain: the scorer's candidate count starts at zero for the selection kernels behind this one -- a 4-byte memset
    // between two kernels is 5 us of an idle GPU)
    if (zero_sc && threadIdx.x == 0) info[blockIdx.x].sc_cnt = 0;
}

// the blend of src/lib.rs:166-173 as SPEC S11 has it, from an exact cosine: ONE definition, so that every path that ranks
// exact scores (single GPU, staged, one-exchange) rounds alike
__device__ __forceinline__ double blend_score(double tau, double c, double lq, double lj) {
    return tau * c + (1.0 - tau) / (1.0 + fabs(lq - lj));
}

__device__ __forceinline__ void publish(HostOut* out, int64_t seq) {
    __threadfence_system();
    out->seq = seq;
}

// scan_dots (fused tail, T = double): the candidate buffer holds the ROWS the scan kept by its cosine bound (QInfo::sc_cnt of
// them in a.ci); their keys -- the mixed form of score_key<double>: fp32 dot, everything else fp64 -- are formed here, now
// that lambda_q exists.  The M smallest of them are the M smallest keys of all scanned rows (scan_dma_kernel, SC), so
// what follows is what follows the threshold filter.  qx_pre: the query already staged in LDS by the caller.
template <typename T>
__device__ __forceinline__ void score_finish_body(FinishArgs a, double coef_s, char* smem, const double* qx_pre, const float* scan_dots, int sc_total) {
    AS_STAMP(16);
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
    const double* qx = qx_pre ? qx_pre : a.q64;
    if (!qx_pre && a.dp <= Q_LDS_MAX) {
        double* qs = (double*)(fi + MS_MAX);
        for (int64_t c = threadIdx.x; c < a.dp; c += blockDim.x) qs[c] = a.q64[c];
        qx = qs;
    }
    __shared__ int fcount;
    int total;
    if (scan_dots) {
        // the caller (fused_finish_kernel) has gathered the waves' reports: si[t] = row, sk[t] = its cosine over the fp32 dot
        // (NaN: not formed yet), sc_total of them (-1: they did not fit)
        const int raw = sc_total < 0 ? CAND_CAP + 1 : sc_total;
        if (raw > CAND_CAP && threadIdx.x == 0) a.info->overflow |= 4;   // the host reruns the scorer on the threshold chain
        total = raw <= CAND_CAP ? raw : 0;
        const double nq_ = a.info->nq, lq_ = a.info->lambda_q;
        const double rq_ = nq_ > 0.0 ? rsqrt(nq_) : 0.0;
        // Keys and selection in one go.  A key is minus a score: tau cos + (1 - tau) L lies in [-1, 1], so the histogram that
        // prunes the list to about M entries needs no range pass -- 1024 fixed bins over [-1, 1] are filled while the keys are
        // formed (the generic select_candidates finds the range first: two more barriers, 2.5 us of this kernel).
        __shared__ unsigned int khist[1024];
        __shared__ int k_bin, k_cnt;
        khist[threadIdx.x] = 0u;
        if (threadIdx.x == 0) k_cnt = 0;
        __syncthreads();
        int mybin[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = (int)threadIdx.x + u * (int)blockDim.x;
            mybin[u] = -1;
            if (t < total) {
                const int j = si[t];
                double cs = sk[t];
                if (cs != cs) {
                    const double nrow = a.n64[j];
                    cs = nrow > 0.0 ? (double)scan_dots[j] * rsqrt(nrow) * rq_ : 0.0;
                }
                double key = -(a.tau * cs + (1.0 - a.tau) / (1.0 + fabs(lq_ - a.lam64[j])));
                key = key == key ? key : key_traits<T>::inf();   // (a NaN key would rank as 0-th: it compares less than nothing)
                sk[t] = key;
                const double fb = (key + 1.0) * 512.0;
                mybin[u] = fb < 0.0 ? 0 : (fb >= 1023.0 ? 1023 : (int)fb);
                atomicAdd(&khist[mybin[u]], 1u);
            }
        }
        __syncthreads();
        AS_STAMP(25);
        if (threadIdx.x == 0) AS_STAMP_VAL(26, total);
        if (threadIdx.x < 64) {   // wave 0: 16 bins per lane, inclusive scan over the lanes, the lane whose bins reach M finishes
            unsigned int hh[16], tot = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                hh[j] = khist[16 * threadIdx.x + j];
                tot += hh[j];
            }
            unsigned int incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int t2 = __shfl_up(incl, o, 64);
                if ((int)threadIdx.x >= o) incl += t2;
            }
            const unsigned int excl = incl - tot;
            const unsigned int want_ = (unsigned)(a.M < total ? a.M : total);
            if (threadIdx.x == 0) k_bin = 1023;
            if (want_ > 0 && excl < want_ && incl >= want_) {
                unsigned int run = excl;
                int b = 16 * (int)threadIdx.x;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    run += hh[j];
                    if (run >= want_) break;
                    b += 1;
                }
                k_bin = b;
            }
        }
        __syncthreads();
        const int bsel = k_bin;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = (int)threadIdx.x + u * (int)blockDim.x;
            if (mybin[u] >= 0 && mybin[u] <= bsel) {
                const int slot = atomicAdd(&k_cnt, 1);
                if (slot < PRUNE_CAP) {
                    pk[slot] = sk[t];
                    pi[slot] = si[t];
                }
            }
        }
        __syncthreads();
        const int R = k_cnt;
        if (R <= PRUNE_CAP) {
            if (threadIdx.x == 0) fcount = total < a.M ? total : a.M;
            for (int t = threadIdx.x; t < R; t += blockDim.x) {
                const T k = pk[t];
                const int i = pi[t];
                int rank = 0;
                for (int s2 = 0; s2 < R; ++s2) rank += lex_less<T>(pk[s2], pi[s2], k, i) ? 1 : 0;
                if (rank < a.M) {
                    fk[rank] = k;
                    fi[rank] = i;
                }
            }
            __syncthreads();
        } else {   // the keys pile up in one bin (mass ties): the generic selection sorts it out
            select_candidates<T>(nullptr, nullptr, total, a.M, sk, si, pk, pi, fk, fi, &fcount);
        }
    } else if (a.from_list) {
        merge_partials<T>(ckey, cidx, a.nlists, a.M, sk, si, fk, fi, &fcount);
        total = (int)(a.nrows < 0x7fffffff ? a.nrows : 0x7fffffff);
    } else {
        const int raw = a.info->sc_cnt;
        if (raw > CAND_CAP && threadIdx.x == 0) a.info->overflow |= 2;
        total = raw < CAND_CAP ? raw : CAND_CAP;
        select_candidates<T>(ckey, cidx, total, a.M, sk, si, pk, pi, fk, fi, &fcount);
    }
    const int Mp = fcount;
    AS_STAMP(17);
    const double nq = a.info->nq, tau = a.tau, lq = a.info->lambda_q;
    // norms and lambdas of the candidates this thread scores afterwards: in flight under the exact evaluation
    double pre_n = 0.0, pre_l = 0.0;
    if ((int)threadIdx.x < Mp) {
        pre_n = a.n64[fi[threadIdx.x]];
        pre_l = a.lam64[fi[threadIdx.x]];
    }
    for (int base = 0; base < Mp; base += 64)   // 64 candidates per round, 16 lanes each
        exact_eval_all(a.x32, a.x64, qx, a.d, a.dp, fi + base, Mp - base < 64 ? Mp - base : 64, sk2 + base, es + base);
    __syncthreads();
    for (int t = threadIdx.x; t < Mp; t += blockDim.x) {
        const int j = fi[t];
        const double nj = t < (int)blockDim.x ? pre_n : a.n64[j], lj = t < (int)blockDim.x ? pre_l : a.lam64[j];
        const double den = sqrt(nj * nq);
        const double c = den > 0.0 ? es[t] / den : 0.0;
        es[t] = blend_score(tau, c, lq, lj);
    }
    __syncthreads();
    AS_STAMP(18);
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
            r.idx = myi + a.goff;
            r.score = es[t];
            a.hits[rank] = r;
        }
        if (a.fuse && a.hout && rank < nhit) {
            a.hout->idx[rank] = myi + a.goff;
            a.hout->score[rank] = es[t];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    AS_STAMP(19);
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
            r.score = (double)((a.info->knn_inexact ? 1 : 0) | (bad ? 2 : 0) | ((a.info->overflow & 1) ? 4 : 0) | ((a.info->overflow & 2) ? 8 : 0) |
                               ((a.info->overflow & 4) ? 16 : 0));
            a.hits[a.topk] = r;
        }
        if (a.fuse && a.hout) {
            a.hout->len = nhit;
            a.hout->lambda_q = a.info->lambda_q;
            a.hout->status = a.info->status;
            a.hout->knn_inexact = a.info->knn_inexact;
            a.hout->score_inexact = bad;
            a.hout->overflow = a.info->overflow;
            // a clean search leaves nothing for a rerun to read: clear the per-search state here, so that the next query's
            // scan (whose prefilter counts into it from its first wave on) needs no kernel in front of it
            const int clean = a.auto_reset && !a.info->knn_inexact && !bad && !a.info->overflow;
            a.hout->state_reset = clean;
            if (clean) reset_query_state(a.info);
            publish(a.hout, a.seq);
        }
    }
    AS_STAMP(20);
}

template <typename T>
__global__ __launch_bounds__(1024) void score_finish_kernel(FinishArgs a, double coef_s) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    score_finish_body<T>(a, coef_s, smem, nullptr, nullptr, 0);
}

// Waves 1..15 of a finish kernel: gather the reports of the scan's waves (SC_WCAP words each: the count, then the rows;
// mostly empty) into one list in the work area -- slots by an LDS atomic, so the order varies, the ranking behind it
// goes by (key, row) alone -- and form the cosines.  Two dependent round trips for the whole block, not two per report:
// every thread first loads the heads of all its reports (count + the first three rows in 16 bytes), then everything
// their candidates need.
__device__ __forceinline__ void gather_reports(const FinishArgs& as_, char* work, const float* scan_dots, int* s_tot_p, int* s_ovf_p) {
    int& s_tot = *s_tot_p;
    int& s_ovf = *s_ovf_p;
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        double* sk = (double*)work;
        int* si = (int*)(sk + CAND_CAP);
        const double nq_ = as_.info->nq;
        const double rq_ = nq_ > 0.0 ? rsqrt(nq_) : 0.0;
        constexpr int NREP = 5;                    // reports per thread and pass: 16 * 256 + 64 waves over 960 threads in one pass
        const int nthr = (int)blockDim.x - 64;
        // (as many passes as the scan had waves: a device with more than 296 CUs, or another scan geometry, takes a second one)
        for (int t0 = (int)threadIdx.x - 64; t0 < as_.sc_nw; t0 += NREP * nthr) {
        i32x4 head[NREP];
#pragma unroll
        for (int i = 0; i < NREP; ++i) {
            const int w2 = t0 + i * nthr;
            head[i] = w2 < as_.sc_nw ? *(const i32x4*)(as_.ci + (int64_t)w2 * SC_WCAP) : i32x4{0, 0, 0, 0};
        }
        int base[NREP];
        double nrow[NREP][3];
        float dt[NREP][3];
#pragma unroll
        for (int i = 0; i < NREP; ++i) {
            const int c2 = head[i][0];
            base[i] = 0;
            if (c2 < 0) s_ovf = 1;
            else if (c2 > 0) base[i] = atomicAdd(&s_tot, c2);
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                nrow[i][e] = 0.0;
                dt[i][e] = 0.0f;
                if (e < c2 && base[i] + c2 <= CAND_CAP) {
                    const int j = head[i][1 + e];
                    nrow[i][e] = as_.n64[j];
                    dt[i][e] = scan_dots[j];
                    if (as_.lam64[j] == -1.0) dt[i][e] = 0.0f;   // (never: the load warms the line the key reads behind the barrier)
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NREP; ++i) {
            const int c2 = head[i][0];
            if (c2 <= 0 || base[i] + c2 > CAND_CAP) continue;
#pragma unroll
            for (int e = 0; e < 3; ++e)
                if (e < c2) {
                    si[base[i] + e] = head[i][1 + e];
                    sk[base[i] + e] = nrow[i][e] > 0.0 ? (double)dt[i][e] * rsqrt(nrow[i][e]) * rq_ : 0.0;
                }
            const int w2 = t0 + i * nthr;
            for (int e = 3; e < c2; ++e) {   // a long report: its cosines are left to the whole block behind the barrier
                si[base[i] + e] = as_.ci[(int64_t)w2 * SC_WCAP + 1 + e];
                sk[base[i] + e] = __longlong_as_double(0x7ff8000000000000ll);
            }
        }
        }
}

// Fused tail of a single query: ONE launch behind the scan.  Phase 1 = knn_finish (the k nearest of the scan's k-NN
// candidates, exact, lambda_q); phase 2 = score_finish over the scan's own scorer candidates (scan_dma_kernel, SC).
// The query is staged in LDS once; the two phases share the rest of the dynamic LDS.  Clears the scan's cosine
// histogram behind itself (the next scan counts into it from its first wave on).
__global__ __launch_bounds__(1024) void fused_finish_kernel(FinishArgs ak, FinishArgs as_, double coef_s, const float* scan_dots, unsigned int* sc_hist) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    AS_STAMP(22);
    double* qs = (double*)smem;
    const double* qx = nullptr;
    if (ak.dp <= Q_LDS_MAX) {
        for (int64_t c = threadIdx.x; c < ak.dp; c += blockDim.x) qs[c] = ak.q64[c];
        qx = qs;
    }
    char* work = smem + sizeof(double) * Q_LDS_MAX;
    __shared__ int s_tot, s_ovf;
    if (threadIdx.x == 0) {
        s_tot = 0;
        s_ovf = 0;
    }
    knn_finish_body<float>(ak, work, qx ? qx : nullptr);   // (its barriers publish s_tot / s_ovf)
    if (threadIdx.x >= 64) gather_reports(as_, work, scan_dots, &s_tot, &s_ovf);   // while wave 0 ranks the neighbours and forms lambda_q
    __syncthreads();   // lambda_q, status and the flags wave 0 has filed in QInfo, and the gathered list: visible to the block
    AS_STAMP(24);
    reset_query_hist(sc_hist, threadIdx.x, blockDim.x);
    score_finish_body<double>(as_, coef_s, work, qx ? qx : nullptr, scan_dots, s_ovf || s_tot > CAND_CAP ? -1 : s_tot);
}

// Staged (row-sharded) search, the step between the two exchanges, in ONE launch: lambda_q from the all-gathered k-NN
// records (wave 0, as q_lambda_kernel) while waves 1..15 gather the scan's scorer candidates (the waves' reports of
// scan_dma_kernel, SC), then the scorer finish over them (score_finish_body: hit records + flags, no publication) --
// instead of q_lambda, score_gmin, score_pickfilter and score_finish.
__global__ __launch_bounds__(1024) void staged_score_kernel(FinishArgs as_, double coef_s, const float* scan_dots, unsigned int* sc_hist,
                                                            const as_knn_rec* recs_all, int64_t m, int64_t k, int metric, int kernel, double sigma,
                                                            double p, double tau0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* qs = (double*)smem;
    const double* qx = nullptr;
    if (as_.dp <= Q_LDS_MAX) {
        for (int64_t c = threadIdx.x; c < as_.dp; c += blockDim.x) qs[c] = as_.q64[c];
        qx = qs;
    }
    char* work = smem + sizeof(double) * Q_LDS_MAX;
    __shared__ int s_tot, s_ovf;
    if (threadIdx.x == 0) {
        s_tot = 0;
        s_ovf = 0;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        q_lambda_body(recs_all, m, m, m, k, metric, kernel, sigma, p, tau0, as_.info, 0);
    } else {
        gather_reports(as_, work, scan_dots, &s_tot, &s_ovf);
    }
    __syncthreads();
    reset_query_hist(sc_hist, threadIdx.x, blockDim.x);
    score_finish_body<double>(as_, coef_s, work, qx, scan_dots, s_ovf || s_tot > CAND_CAP ? -1 : s_tot);
}

// ------------------------------------------------------------------ one exchange per sharded query
// Every rank finishes what only it can finish -- its k exact nearest rows (records) and the EXACT cosine and lambda of every
// row its scan kept as a scorer candidate -- into one block of bytes; the blocks are all-gathered ONCE; every rank then forms
// lambda_q from all records, scores all candidates (exact scores: there is nothing left to prove about the list -- that the
// rows outside it cannot reach the top k is the scan's cosine-bound argument, as on one GPU) and ranks them identically.
// Block of a rank: [k as_knn_rec][XHead][xcap XCand], padded to a multiple of sizeof(as_knn_rec).
struct XHead {
    int count;   // candidates written (may exceed xcap: overflow)
    int flags;   // bit0 k-NN list not proven, bit2 k-NN buffer overflow, bit3 scorer buffer overflow, bit4 candidates did not fit /
                 // this rank had none to offer (no fused scan), bit5 this rank failed -- the trailing hit record's bits (score_finish_body)
    int pad[2];
};
struct XCand {
    int64_t idx;   // global item id
    double cosv;   // exact fp64 cosine
    double lam;    // lambda of the item
};
constexpr int X1_BLOCKS = 16;        // candidate blocks beside the k-NN block
constexpr int X1_BLOCKS_COARSE = 128; // blocks of the coarse scan's tail: scorer candidates AND k-NN candidates by the thousand, one round of 64 rows per block
constexpr int X1_LOCAL_CAP = 4096;   // candidates one block gathers from its share of the scan's reports
static std::atomic<int> g_x1_blocks{X1_BLOCKS_COARSE};   // measurement: as_set_tuning("x1_blocks", v), 16 .. 256
void set_x1_blocks(int v) { g_x1_blocks.store(v < 16 ? 16 : (v > 256 ? 256 : v), std::memory_order_relaxed); }

// grid 1 + X1_BLOCKS: block 0 = knn_finish (records into the exchange block), blocks 1.. = the scan waves' reports -> exact
// cosines.  The two halves do not depend on each other: the k-NN phase (17 us, one block) hides the candidates' evaluation.
// xk != null (the coarse scan: k-NN candidates by the hundred, every one of them to be evaluated exactly): no k-NN block --
// every block takes its share of the k-NN candidate buffer as well and appends (id, exact key, distance, gy) of the candidates
// inside eps to xk; the last block to finish ranks them and writes the records (below).
struct XKnn {
    int idx;   // local row
    int pad;
    double key, dist, gy;
};
// xmode (xk form only): 0 both shares, 1 the k-NN share alone (the coarse chain: the scorer's candidates come later), 2 the scorer's
// candidates alone, from a FLAT list (as_.ci, as_.info->sc_cnt entries: the threshold filter over the kept dots) -- no ranking
__global__ __launch_bounds__(1024) void staged_x1_kernel(FinishArgs ak, FinishArgs as_, XHead* head, XCand* cands, int xcap, int preset_flags, XKnn* xk, int xmode) {
    int* xk_count = (int*)(xk + CAND_CAP);   // (behind the entries: zero at the start of a pass -- the finish kernel leaves it so)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* qs = (double*)smem;
    const double* qx = nullptr;
    if (ak.dp <= Q_LDS_MAX) {
        for (int64_t c = threadIdx.x; c < ak.dp; c += blockDim.x) qs[c] = ak.q64[c];
        qx = qs;
    }
    // (the coarse tail's blocks take only what the query needs in front of their 64 KB of work area -- 70 KB at 768 columns: a CU
    // keeps two blocks of a scan beside one of them; with the 137 KB of the other form a tail kernel held back the scan blocks of
    // the next caller's search on half the chip)
    char* work = smem + (xk ? (ak.dp <= Q_LDS_MAX ? sizeof(double) * (size_t)ak.dp : 0) : sizeof(double) * Q_LDS_MAX);
    if (blockIdx.x == 0 && !xk) {
        knn_finish_body<float>(ak, work, qx);
        if (threadIdx.x == 0) {   // (wave 0 ran the whole body: its own stores)
            const int fl = preset_flags | (ak.info->knn_inexact ? 1 : 0) | ((ak.info->overflow & 1) ? 4 : 0) | ((ak.info->overflow & 2) ? 8 : 0);
            if (fl) atomicOr(&head->flags, fl);
        }
        return;
    }
    __shared__ int s_tot, s_ovf, s_base;
    __shared__ float s_thrf;
    if (threadIdx.x < 64) {
        // the bound of the FINAL histogram: what a lossy wave report is held against (report_rows)
        float thrf = 3.0e38f;   // (no histogram handed over: a lossy report is an overflow)
        if (as_.sc_hist) {
            int jb;
            const unsigned h = __hip_atomic_load(&as_.sc_hist[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            thrf = sc_bound(h, as_.sc_m, (int)threadIdx.x, jb) - as_.sc_w;
        }
        if (threadIdx.x == 0) s_thrf = thrf;
    }
    int* si = (int*)work;                          // X1_LOCAL_CAP rows
    double* o_sq = (double*)(si + X1_LOCAL_CAP);   // 64 + 64 results of a round
    double* o_dot = o_sq + 64;
    if (threadIdx.x == 0) {
        s_tot = 0;
        s_ovf = 0;
    }
    __syncthreads();
    // (reports dealt round robin: the waves that end first -- under a bound still loose -- keep the most rows, and they are neighbours)
    const int nb = xk ? (int)gridDim.x : (int)gridDim.x - 1, b = xk ? (int)blockIdx.x : (int)blockIdx.x - 1;
    if (xk) {
        // The coarse scan's tail: ONE list per block -- its share of the k-NN candidate buffer (entries b, b + nb, ...) in front,
        // its share of the scan waves' scorer reports behind -- evaluated exactly in rounds of 64 rows (one round at 1M x 768 on
        // 128 blocks: 10 + 9 rows), each row then filed by its kind.  (Before: the two shares in rounds of their own on 32 blocks,
        // three serial rounds per block.)
        const int raw = xmode == 2 ? 0 : ak.info->knn_cnt;
        int mine = 0;
        if (raw > CAND_CAP) {
            if (b == 0 && threadIdx.x == 0) {
                ak.info->overflow |= 1;
                atomicOr(&head->flags, 4);
            }
        } else {
            mine = raw > b ? (raw - b + nb - 1) / nb : 0;   // <= CAND_CAP / nb + 1
            for (int t = threadIdx.x; t < mine; t += blockDim.x) si[t] = ak.ci[b + nb * t];
        }
        if (b == 0 && threadIdx.x == 0 && (preset_flags || (ak.info->overflow & 2))) atomicOr(&head->flags, preset_flags | ((ak.info->overflow & 2) ? 8 : 0));
        const int room = X1_LOCAL_CAP - mine;
        if (xmode == 2) {
            const int fc = as_.info->sc_cnt;
            if (fc > CAND_CAP) {
                s_ovf = 1;
            } else {
                const int ms = fc > b ? (fc - b + nb - 1) / nb : 0;   // this block's entries b, b + nb, ...
                for (int t = threadIdx.x; t < ms; t += blockDim.x) si[t] = as_.ci[b + nb * t];
                if (threadIdx.x == 0) s_tot = ms;
            }
        }
        for (int w = b + nb * (int)threadIdx.x; xmode == 0 && w < as_.sc_nw; w += nb * (int)blockDim.x) {
            const int* rep = as_.ci + (int64_t)w * SC_WCAP;
            const int c2 = report_rows(rep, s_thrf);
            if (c2 < 0) s_ovf = 1;
            else if (c2 > 0) {
                const int base = atomicAdd(&s_tot, c2);
                if (base + c2 <= room)
                    for (int e = 0; e < c2; ++e) si[mine + base + e] = rep[1 + e];
            }
        }
        __syncthreads();
        int tot = s_tot;
        if (s_ovf || tot > room) {
            if (threadIdx.x == 0) {
                atomicOr(&head->flags, 16);
                atomicMax(&head->pad[0], tot >> 4);       // (what did not fit, in sixteens: read by the host's debug line only)
                if (s_ovf) atomicAdd(&head->pad[1], 1);
            }
            tot = 0;
        }
        const double nqk = ak.info->nq;
        const int total = mine + tot;
        for (int base = 0; base < total; base += 64) {   // (block-uniform)
            const int m = total - base < 64 ? total - base : 64;
            const int sc_lo = base > mine ? base : mine;          // first scorer entry of this round
            const int m_sc = base + m - sc_lo;
            // (in flight under the evaluation instead of behind it: the round's ticket and, per row, its norm and lambda)
            int ticket = 0;
            if (threadIdx.x == 0 && m_sc > 0) ticket = atomicAdd(&head->count, m_sc);   // one ticket per block and round
            double pre_n = 0.0, pre_l = 0.0;
            if ((int)threadIdx.x < m) {
                const int jp = si[base + (int)threadIdx.x];
                pre_n = ak.n64[jp];
                pre_l = base + (int)threadIdx.x >= mine ? as_.lam64[jp] : 0.0;
            }
            exact_eval_all(ak.x32, ak.x64, qx ? qx : ak.q64, ak.d, ak.dp, si + base, m, o_sq, o_dot);
            if (threadIdx.x == 0 && m_sc > 0) s_base = ticket;
            __syncthreads();
            if ((int)threadIdx.x < m) {
                const int e_ = base + (int)threadIdx.x;
                const int j = si[e_];
                const double sq = o_sq[threadIdx.x], dot = o_dot[threadIdx.x];
                if (e_ < mine) {
                    XKnn e;
                    e.idx = j;
                    e.pad = 0;
                    if (ak.metric == AS_METRIC_L2) {   // (knn_finish_body's expressions, one for one)
                        e.key = sq;
                        e.dist = sqrt(sq);
                        e.gy = dot;
                    } else {
                        const double den = sqrt(nqk * pre_n);
                        const double c = den > 0.0 ? dot / den : 0.0;
                        const double dd = cosine_distance(c);
                        e.key = dd;
                        e.dist = dd;
                        e.gy = c;
                    }
                    if (e.key <= ak.epskey) xk[atomicAdd(xk_count, 1)] = e;   // (the few dozen inside eps: the last block ranks them)
                } else {
                    const int slot = s_base + (e_ - sc_lo);
                    if (slot < xcap) {
                        const double den = sqrt(pre_n * nqk);
                        XCand c;
                        c.idx = (int64_t)j + as_.goff;
                        c.cosv = den > 0.0 ? dot / den : 0.0;
                        c.lam = pre_l;
                        cands[slot] = c;
                    } else {
                        atomicOr(&head->flags, 16);
                    }
                }
            }
            __syncthreads();
        }
    }
    if (!xk) {
        for (int w = b + nb * (int)threadIdx.x; w < as_.sc_nw; w += nb * (int)blockDim.x) {
            const int* rep = as_.ci + (int64_t)w * SC_WCAP;
            const int c2 = report_rows(rep, s_thrf);
            if (c2 < 0) s_ovf = 1;
            else if (c2 > 0) {
                const int base = atomicAdd(&s_tot, c2);
                if (base + c2 <= X1_LOCAL_CAP)
                    for (int e = 0; e < c2; ++e) si[base + e] = rep[1 + e];
            }
        }
        __syncthreads();
        const int tot = s_tot;
        const bool sc_fits = !(s_ovf || tot > X1_LOCAL_CAP);
        if (!sc_fits && threadIdx.x == 0) {
            atomicOr(&head->flags, 16);
            atomicMax(&head->pad[0], tot >> 4);       // (what did not fit, in sixteens: read by the host's debug line only)
            if (s_ovf) atomicAdd(&head->pad[1], 1);
        }
        const double nq = as_.info->nq;
        for (int base = 0; sc_fits && base < tot; base += 64) {   // (block-uniform)
            const int m = tot - base < 64 ? tot - base : 64;
            exact_eval_all(as_.x32, as_.x64, qx ? qx : as_.q64, as_.d, as_.dp, si + base, m, o_sq, o_dot);
            if (threadIdx.x == 0) s_base = atomicAdd(&head->count, m);   // one ticket per block and round
            __syncthreads();
            if ((int)threadIdx.x < m) {
                const int slot = s_base + (int)threadIdx.x;
                if (slot < xcap) {
                    const int j = si[base + threadIdx.x];
                    const double den = sqrt(as_.n64[j] * nq);
                    XCand c;
                    c.idx = (int64_t)j + as_.goff;
                    c.cosv = den > 0.0 ? o_dot[threadIdx.x] / den : 0.0;
                    c.lam = as_.lam64[j];
                    cands[slot] = c;
                } else {
                    atomicOr(&head->flags, 16);
                }
            }
            __syncthreads();
        }
        return;
    }
    if (xmode == 2) return;   // (the records stand from the launch that evaluated the k-NN share)
    // The coarse scan's k-NN records: the LAST block to get here ranks the candidates inside eps that all blocks have appended
    // (a few dozen) by (key, id) and writes the k nearest as records -- knn_finish_body's selection with nothing left to prove.
    // (Release / acquire at agent scope around one ticket per block: the blocks sit on different XCDs, each with its own L2.)
    // One fence per block, by the thread that draws the ticket, behind a barrier (the block's appends happen before it): sixteen
    // waves fencing in each of 32 blocks -- an L2 write-back each -- made this kernel 38 us instead of 15.
    __shared__ int s_ticket;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_ticket = atomicAdd(xk_count + 1, 1);
        if (s_ticket == nb - 1) __threadfence();
    }
    __syncthreads();
    if (s_ticket != nb - 1) return;
    const int raw = ak.info->knn_cnt;
    const int P = raw <= CAND_CAP ? __hip_atomic_load(xk_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    double* rk = (double*)work;            // P keys (P <= CAND_CAP: 32 KB + 16 KB + 8 KB of the work area)
    int* ri = (int*)(rk + CAND_CAP);
    unsigned short* sel = (unsigned short*)(ri + CAND_CAP);   // entries that may be among the k nearest (P <= CAND_CAP = 4096: 16 bits)
    __shared__ unsigned int khist[1024];
    __shared__ unsigned long long s_kmin, s_kmax;
    __shared__ int s_kbin, s_nsel;
    khist[threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        s_kmin = ~0ull;
        s_kmax = 0ull;
        s_kbin = 1023;
        s_nsel = 0;
    }
    __syncthreads();
    {
        // (keys are >= 0 -- squared distances, rectified-cosine distances --: their bit patterns order like the values)
        unsigned long long lmin = ~0ull, lmax = 0ull;
        for (int u = threadIdx.x; u < P; u += blockDim.x) {
            const double kk = xk[u].key;
            rk[u] = kk;
            ri[u] = xk[u].idx;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(kk);
            lmin = bits < lmin ? bits : lmin;
            lmax = bits > lmax ? bits : lmax;
        }
        if (lmax >= lmin) {
            atomicMin(&s_kmin, lmin);
            atomicMax(&s_kmax, lmax);
        }
    }
    if (ak.recs)
        for (int64_t t = threadIdx.x; t < ak.k; t += blockDim.x) {
            as_knn_rec r;
            r.idx = -1;
            r.key = key_traits<double>::inf();
            r.dist = 0; r.gy = 0; r.deg = 0; r.ny = 0;
            ak.recs[t] = r;
        }
    __syncthreads();
    // The k nearest of P entries: a dense neighbourhood puts hundreds of rows inside eps (a query at the heart of a cluster: a
    // thousand), and ranking every entry against every other in ONE block was this kernel's long tail (P = 500: +20 us, P = 1000:
    // +80 us; a tenth of the queries at 1M x 768).  A 1024-bin histogram over [min key, max key] finds the bin that holds the k-th
    // key; only the entries up to that bin are ranked (entries of later bins have larger keys), one WAVE per entry, the P
    // comparisons of an entry spread over its lanes.
    const double kmin = __longlong_as_double((long long)s_kmin), kmax = __longlong_as_double((long long)s_kmax);
    const double kscale = P > 0 && kmax > kmin ? 1023.0 / (kmax - kmin) : 0.0;
    auto kbin_of = [&](double kk) {
        const double fb = (kk - kmin) * kscale;
        return fb >= 1023.0 ? 1023 : (fb > 0.0 ? (int)fb : 0);   // (monotone in the key: a larger bin means a larger key)
    };
    for (int u = threadIdx.x; u < P; u += blockDim.x) atomicAdd(&khist[kbin_of(rk[u])], 1u);
    __syncthreads();
    const int want_k = (int)(ak.k < (int64_t)P ? ak.k : (int64_t)P);
    if (threadIdx.x < 64) {   // 16 bins per lane, inclusive scan over the lanes, the lane whose bins reach k finishes
        unsigned int hh[16], tot = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            hh[j] = khist[16 * threadIdx.x + j];
            tot += hh[j];
        }
        unsigned int incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int t2 = __shfl_up(incl, o, 64);
            if ((int)threadIdx.x >= o) incl += t2;
        }
        const unsigned int excl = incl - tot, want_ = (unsigned)want_k;
        if (want_ > 0 && excl < want_ && incl >= want_) {
            unsigned int run = excl;
            int bsel = 16 * (int)threadIdx.x;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                run += hh[j];
                if (run >= want_) break;
                bsel += 1;
            }
            s_kbin = bsel;
        }
    }
    __syncthreads();
    const int kbin = s_kbin;
    for (int u = threadIdx.x; u < P; u += blockDim.x)
        if (kbin_of(rk[u]) <= kbin) sel[atomicAdd(&s_nsel, 1)] = (unsigned short)u;
    __syncthreads();
    const int nsel = s_nsel;
    {
        const int wv = (int)(threadIdx.x >> 6), nwv = (int)(blockDim.x >> 6), lane = (int)(threadIdx.x & 63);
        for (int c = wv; c < nsel; c += nwv) {   // (wave-uniform)
            const int u = sel[c];
            const double myk = rk[u];
            const int myi = ri[u];
            int cnt = 0;
            for (int s2 = lane; s2 < P; s2 += 64) cnt += lex_less<double>(rk[s2], ri[s2], myk, myi) ? 1 : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            if (lane == 0 && cnt < ak.k && ak.recs) {
                as_knn_rec r;
                r.idx = (int64_t)myi + ak.goff;
                r.key = myk;
                r.dist = xk[u].dist;
                r.gy = xk[u].gy;
                r.deg = ak.deg ? ak.deg[myi + ak.goff] : 0.0;
                r.ny = ak.ny ? ak.ny[myi + ak.goff] : 0.0;
                ak.recs[cnt] = r;
            }
        }
    }
    if (threadIdx.x == 0) {
        ak.info->knn_total = raw <= CAND_CAP ? raw : CAND_CAP;
        ak.info->knn_inexact = 0;
        xk_count[0] = 0;   // (the next pass's appends and tickets start at zero)
        xk_count[1] = 0;
    }
}

// After the exchange, on every rank alike: lambda_q (wave 0) while the other waves pull the ranks' candidates into LDS; exact
// scores; the topk by (score desc, id asc) through a fixed-range histogram prune (a score lies in [-1, 1]) and a rank count;
// publication.  Clears the per-search state and the scan's histograms behind a clean search.
__global__ __launch_bounds__(1024) void staged_x1_final_kernel(const char* __restrict__ all, int world, int64_t xbytes, int64_t krec, int xcap, int64_t k,
                                                               int metric, int kernel, double sigma, double p, double tau0, double tau, int64_t topk,
                                                               int64_t ntotal, QInfo* info, HostOut* out, int64_t seq, unsigned int* sc_hist, XHead* own_head) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sc = (double*)smem;                 // CAND_CAP cosines, then scores
    double* sl = sc + CAND_CAP;                 // lambdas
    int* sid = (int*)(sl + CAND_CAP);           // ids
    double* pk = (double*)(sid + CAND_CAP);     // PRUNE_CAP pruned scores
    int* pi = (int*)(pk + PRUNE_CAP);
    __shared__ unsigned int khist[1024];
    __shared__ int s_tot, s_flags, k_bin, k_cnt;
    khist[threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        s_tot = 0;
        s_flags = 0;
        k_cnt = 0;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        q_lambda_body((const as_knn_rec*)all, krec * world, krec, xbytes / (int64_t)sizeof(as_knn_rec), k, metric, kernel, sigma, p, tau0, info, 0);
    } else {
        const int wv = (int)(threadIdx.x >> 6) - 1, nwv = 15, lane = (int)(threadIdx.x & 63);
        // (a rank per wave; a rank's base in the list by one LDS ticket)
        for (int r = wv; r < world; r += nwv) {
            const XHead* h = (const XHead*)(all + (int64_t)r * xbytes + krec * (int64_t)sizeof(as_knn_rec));
            const XCand* cs = (const XCand*)(h + 1);
            int cnt = h->count, fl = h->flags;
            if (cnt < 0 || cnt > xcap) {   // (a block of 0xff bytes is a failed rank's: as_query_search_staged)
                cnt = 0;
                fl |= 16;
            }
            int base = 0;
            if (lane == 0) {
                if (fl) atomicOr(&s_flags, fl);
                base = atomicAdd(&s_tot, cnt);
            }
            base = __shfl(base, 0, 64);
            if (base + cnt <= CAND_CAP)
                for (int t = lane; t < cnt; t += 64) {
                    const XCand c = cs[t];
                    sc[base + t] = c.cosv;
                    sl[base + t] = c.lam;
                    sid[base + t] = (int)c.idx;
                }
        }
    }
    __syncthreads();
    reset_query_hist(sc_hist, threadIdx.x, blockDim.x);
    int flags = s_flags;
    int total = s_tot;
    if (total > CAND_CAP) {
        flags |= 16;
        total = 0;
    }
    const double lq = info->lambda_q;
    int mybin[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = (int)threadIdx.x + u * (int)blockDim.x;
        mybin[u] = -1;
        if (t < total) {
            double s = blend_score(tau, sc[t], lq, sl[t]);
            s = s == s ? s : -key_traits<double>::inf();
            sc[t] = s;
            const double fb = (1.0 - s) * 512.0;   // descending scores = ascending bins
            mybin[u] = fb < 0.0 ? 0 : (fb >= 1023.0 ? 1023 : (int)fb);
            atomicAdd(&khist[mybin[u]], 1u);
        }
    }
    __syncthreads();
    const int64_t want64 = topk < ntotal ? topk : ntotal;
    const int nhit = (int)(total < want64 ? total : want64);
    if (threadIdx.x < 64) {   // 16 bins per lane, inclusive scan over the lanes, the lane whose bins reach nhit finishes
        unsigned int hh[16], tot = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            hh[j] = khist[16 * threadIdx.x + j];
            tot += hh[j];
        }
        unsigned int incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int t2 = __shfl_up(incl, o, 64);
            if ((int)threadIdx.x >= o) incl += t2;
        }
        const unsigned int excl = incl - tot, want_ = (unsigned)nhit;
        if (threadIdx.x == 0) k_bin = 1023;
        if (want_ > 0 && excl < want_ && incl >= want_) {
            unsigned int run = excl;
            int bsel = 16 * (int)threadIdx.x;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                run += hh[j];
                if (run >= want_) break;
                bsel += 1;
            }
            k_bin = bsel;
        }
    }
    __syncthreads();
    const int bsel = k_bin;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = (int)threadIdx.x + u * (int)blockDim.x;
        if (mybin[u] >= 0 && mybin[u] <= bsel) {
            const int slot = atomicAdd(&k_cnt, 1);
            if (slot < PRUNE_CAP) {
                pk[slot] = sc[t];
                pi[slot] = sid[t];
            }
        }
    }
    __syncthreads();
    const int R = k_cnt;
    if (R > PRUNE_CAP) flags |= 16;   // mass ties in one bin: the two-exchange chain sorts them out
    else
        for (int t = threadIdx.x; t < R; t += blockDim.x) {
            const double myk = -pk[t];
            const int myi = pi[t];
            int rank = 0;
            for (int s2 = 0; s2 < R; ++s2) rank += lex_less<double>(-pk[s2], pi[s2], myk, myi) ? 1 : 0;
            if (rank < nhit) {
                out->idx[rank] = myi;
                out->score[rank] = pk[t];
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        out->len = nhit;
        out->lambda_q = info->lambda_q;
        out->status = info->status;
        out->knn_inexact = (flags & 1) ? 1 : 0;
        out->score_inexact = 0;
        out->overflow = ((flags & 4) ? 1 : 0) | ((flags & 8) ? 2 : 0) | ((flags & 16) ? 4 : 0) | ((flags & 32) ? 8 : 0);
        const int clean = !flags;
        out->state_reset = clean;
        if (clean) reset_query_state(info);
        if (own_head) {   // this rank's block starts the next pass empty (its count and flags are accumulated by atomics)
            // (what did not fit, for the host's debug line: candidates written, the largest share of a block that did not fit, blocks
            // that met an overflowed wave report)
            out->pad_ = (flags & 16) ? ((own_head->count & 0xffff) | ((own_head->pad[0] & 0xff) << 16) | ((own_head->pad[1] & 0xff) << 24)) : 0;
            own_head->count = 0;
            own_head->flags = 0;
            own_head->pad[0] = 0;
            own_head->pad[1] = 0;
        }
        publish(out, seq);
    }
}

// merge m hit records (own or all-gathered) -> final topk, written to pinned host memory
// blockIdx.x = query slot; [rank][slot][per] layout as in q_lambda_kernel
__global__ __launch_bounds__(1024) void hits_final_kernel(const as_hit_rec* __restrict__ hits_all, int64_t m, int64_t per, int64_t rstride,
                                                          int64_t topk, const QInfo* info, HostOut* out, int64_t seq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    info += blockIdx.x;
    out += blockIdx.x;
    const as_hit_rec* __restrict__ hits0 = hits_all + (int64_t)blockIdx.x * per;
#define hits_at(t) hits0[((t) / per) * rstride + ((t) % per)]
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
        const as_hit_rec r = hits_at(t);
        const bool valid = r.idx >= 0;
        if (r.idx == -2) flags_l |= (int)r.score;
        r_key[t] = valid ? -r.score : key_traits<double>::inf();
        r_idx[t] = valid ? (int)r.idx : 0x7fffffff;
    }
#undef hits_at
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
        // (bit 3: some rank could not finish its part of a staged search -- as_query_search_staged files the flag so that every
        // rank leaves the pass with the same error instead of waiting for a collective its peer never enters)
        out->overflow = (info->overflow & 7) | ((fl & 4) ? 1 : 0) | ((fl & 8) ? 2 : 0) | ((fl & 16) ? 4 : 0) | ((fl & 32) ? 8 : 0);
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
    if (need <= MAX_KLIST) return MAX_KLIST;   // two candidates per lane where one wave owns a list (k up to 120)
    return -1;
}

template <typename T>
static SelArgs<T> make_sel(as_query* q, const T* dots, int M, int64_t exclude) {
    const as_space* sp = q->sp;
    SelArgs<T> a;
    a.dots = dots; a.dots32 = nullptr; a.n32 = sp->n32; a.inorm32 = sp->inorm32; a.n64 = sp->n64; a.lam32 = sp->lam32; a.lam64 = sp->lam64;
    a.info = q->info; a.info_w = q->info; a.n = sp->n; a.r0 = q->r0; a.r1 = q->r1; a.exclude = exclude;
    a.M = M; a.metric = sp->opts.metric;
    a.epskey = 0; a.coef = 0; a.tau = 1.0; a.margin = 0.0;
    a.pkey = (T*)q->pkey; a.pidx = q->pidx;
    a.gmin = (T*)q->gmin; a.ckey = (T*)q->ckey_s; a.cidx = q->cidx_s; a.sd = q->ss.dots; a.ts = q->ss.dots_ts; a.rs = q->ss.dots_rs;
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
    f.x32 = sp->x32; f.x64 = sp->x64; f.n64 = sp->n64; f.q64 = q->q64_src ? q->q64_src : q->q64; f.lam64 = sp->lam64;
    // deg / ny are read at item id (local row + goff): a sharded graph holds its own rows only, based at row0
    f.deg = q->gr ? q->gr->deg - q->gr->row0 : nullptr; f.ny = q->gr ? q->gr->ny - q->gr->row0 : nullptr;
    f.info = q->info; f.n = sp->n; f.d = sp->d; f.dp = sp->dp; f.k = q->k; f.topk = q->topk; f.nrows = q->r1 - q->r0;
    f.metric = sp->opts.metric; f.kernel = sp->opts.kernel; f.nmax = sp->nmax; f.goff = sp->row_offset;
    if (q->gr) {
        f.sigma = q->gr->gp.sigma; f.p = q->gr->gp.p; f.tau0 = q->gr->tau0;
    }
    f.from_list = q->robust;
    f.ss = q->ss;
    return f;
}

template <typename T>
static size_t score_lds() {
    return (sizeof(T) + sizeof(int)) * (size_t)(CAND_CAP + PRUNE_CAP + MS_MAX) + 2 * sizeof(double) * MS_MAX + sizeof(double) * Q_LDS_MAX;
}

template <typename T>
static size_t finish_lds() {
    // candidates + pruned candidates + exact keys of the exhaustive pass + the query
    return (sizeof(T) + sizeof(int)) * (size_t)(CAND_CAP + PRUNE_CAP) + sizeof(double) * CAND_CAP + sizeof(double) * Q_LDS_MAX;
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
    // (coarse dots: the picked threshold bounds the k-th COARSE key; the k nearest rows' coarse keys lie up to twice the keys'
    // error above it -- everything the filter keeps is evaluated exactly by the tail, nothing is proven from these keys)
    if (q->coarse) a.margin = (q->sp->opts.metric == AS_METRIC_L2 ? 2.0 * a.coef * (q->sp->nmax + q->h_nq) : 2.0 * a.coef) * 1.0001 + 1.0e-6 * a.epskey;
    a.ckey = (T*)q->ckey_k;
    a.cidx = q->cidx_k;
    const unsigned nb = (unsigned)q->nb;
    hipLaunchKernelGGL((score_gmin_kernel<T, 1>), dim3((unsigned)((ng + 3) / 4), 1, nb), dim3(256), 0, st, a, G, ng);
    // threshold + filter in one launch: every block derives the threshold itself (pick_thr_block)
    const unsigned pg = (unsigned)std::min<int64_t>((rows + 1023) / 1024, std::max(q->cus, 1));
    hipLaunchKernelGGL((score_pickfilter_kernel<T, 1>), dim3(pg, 1, nb), dim3(1024), 0, st, a, ng, q->Mk);
}

static as_status knn_repair(as_query* q, double eps, int64_t exclude) {
    if (q->r1 - q->r0 <= 0) return AS_OK;
    if (q->exact) launch_knn_repair<double, unsigned long long, 8>(q, q->dots64, eps, exclude);
    else launch_knn_repair<float, unsigned int, 4>(q, q->dots32, eps, exclude);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

static as_status run_knn(as_query* q, double eps, int64_t exclude, int fuse_lambda, int32_t* o_idx, double* o_key,
                         double* o_dist, double* o_gy, int32_t* o_cnt, int thresholded = 0, int exhaustive = 0) {
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
    f.exhaustive = exhaustive;
    f.unproven = o_idx ? q->unproven_dev : nullptr;
    if (q->robust) {
        int nw = 0;
        const int grid = sel_grid(q, &nw);
        f.nlists = nw;
        if (q->exact) {
            SelArgs<double> a = make_sel<double>(q, q->dots64, q->Mk, exclude);
            a.epskey = epskey; a.coef = f.coef;
            if (q->Mk > 64) hipLaunchKernelGGL((knn_partial_kernel<double, 2>), dim3(grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((knn_partial_kernel<double, 1>), dim3(grid), dim3(256), 0, st, a);
            f.ck = q->pkey; f.ci = q->pidx;
            hipLaunchKernelGGL(knn_finish_kernel<double>, dim3(1), dim3(1024), finish_lds<double>(), st, f);
        } else {
            SelArgs<float> a = make_sel<float>(q, q->dots32, q->Mk, exclude);
            a.epskey = epskey; a.coef = f.coef;
            if (q->Mk > 64) hipLaunchKernelGGL((knn_partial_kernel<float, 2>), dim3(grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((knn_partial_kernel<float, 1>), dim3(grid), dim3(256), 0, st, a);
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

// the fused tail's dynamic LDS: the query once, then the larger of the two phases' work areas (each sized with a query of its own)
static size_t fused_lds() { return sizeof(double) * Q_LDS_MAX + std::max(finish_lds<float>(), score_lds<double>()) - sizeof(double) * Q_LDS_MAX; }

// Fused tail (single query, fp32 scan with scan-side scorer candidates): ONE launch behind the scan -- knn_finish and
// score_finish as the two phases of fused_finish_kernel.
static as_status run_fused(as_query* q, double eps, double tau) {
    const as_space* sp = q->sp;
    hipStream_t st = q->stream;
    FinishArgs fk = make_finish(q);
    fk.M = q->Mk; fk.epskey = sp->opts.metric == AS_METRIC_L2 ? eps * eps : eps; fk.coef = coef_query(q, false);
    fk.recs = nullptr; fk.fuse = 1; fk.ck = q->ckey_k; fk.ci = q->cidx_k; fk.from_list = 0;   // (records are the staged path's: nobody reads them here)
    fk.exhaustive = q->coarse;
    FinishArgs fs = make_finish(q);
    fs.tau = tau; fs.M = q->Ms; fs.hits = nullptr; fs.fuse = 1; fs.hout = q->hout_dev; fs.seq = q->seq; fs.auto_reset = 1;
    fs.ck = q->ckey_s; fs.ci = q->sc_widx; fs.from_list = 0; fs.sc_nw = q->sc_nw;
    const double coef_s = tau * (coef_query(q, false) + 1.0e-14) + 4.0 * 2.220446049250313e-16;   // as launch_score's mixed keys
    hipLaunchKernelGGL(fused_finish_kernel, dim3(1), dim3(1024), fused_lds(), st, fk, fs, coef_s, (const float*)q->dots32, q->sc_hist);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

template <typename T, typename U, int PASSES>
static void launch_score(as_query* q, const T* dots, FinishArgs f, int fuse_final, const float* dots32 = nullptr) {
    hipStream_t st = q->stream;
    f.M = q->Ms; f.hits = q->hits; f.fuse = fuse_final; f.hout = q->hout_dev; f.seq = q->seq;
    f.auto_reset = fuse_final && q->cap == 1 ? 1 : 0;
    // mixed (fp32 dots, fp64 keys): the only error of a key is the dot's, scaled by tau
    // fp16 cosines (batched MFMA pass): the rounding to nearest of a value in [-1, 1] (2^-11 relative), the fp32 reciprocal
    // norms and their products in front of it (a few ulp of fp32), on top of the dot's own error
    const double e_half = q->dots_half && dots32 ? 4.8828125e-4 + 1.0e-6 : 0.0;
    // the coarse keys' fp32 evaluation (batch_key) is not scaled by tau: below tau = 0.05 -- where the scores are nearly all
    // lambda term and differ in its last digits -- the lambda term stays in fp64 (tau = 0: 42 000 queries/s against 6 600,
    // every slot failing its proof by 1.5e-6 and going to the single-query path)
    const int half_mode = q->dots_half && dots32 ? (f.tau < 0.05 ? 2 : 1) : 0;
    const double e_key32 = half_mode == 1 ? 1.5e-6 : 0.0;
    const double coef_s = dots32 ? f.tau * (coef_query(q, false) + 1.0e-14 + e_half) + e_key32 + 4.0 * 2.220446049250313e-16 : coef_query(q, sizeof(T) == 8);
    if (q->robust && q->Ms > MAX_LIST) {
        // wide lists: exact global selection, then the filter-path finish kernel on exactly M rows
        const int64_t rows = q->r1 - q->r0;
        const int KP = (int)sizeof(T);
        SelArgs<T> a = make_sel<T>(q, dots, q->Ms, -1);
        a.dots32 = dots32;
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
        a.dots32 = dots32;
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
        a.dots32 = dots32;
        a.tau = f.tau;
        const unsigned nb = (unsigned)q->nb;
        const unsigned fg = (unsigned)std::min<int64_t>((rows + 255) / 256, 2048);
        if (dots32 && q->cap == GQ && sizeof(T) == 8) {
            // batched workspace: every slot's keys from one read of a row's norm and lambda (all GQ slots: idle ones hold
            // zero queries and cost nothing but their share of the dots)
            BatchSel b;
            b.dots32 = dots32; b.n64 = q->sp->n64; b.lam64 = q->sp->lam64; b.lam32 = q->sp->lam32; b.info = q->info; b.info_w = q->info;
            b.r0 = q->r0; b.r1 = q->r1; b.sd = q->ss.dots; b.ts = q->ss.dots_ts; b.rs = q->ss.dots_rs; b.tau = f.tau;
            b.gmin = (double*)q->gmin; b.ckey = (double*)q->ckey_s; b.cidx = q->cidx_s; b.ns = q->nb; b.half = half_mode;
            constexpr int NSW = 8;   // slots per wave: a row's norm and lambda are read GQ / NSW times instead of GQ
            const unsigned ny = (unsigned)((q->nb + NSW - 1) / NSW);
            hipLaunchKernelGGL((score_gmin_batch_kernel<NSW>), dim3((unsigned)((ng + 3) / 4), ny), dim3(256), 0, st, b, G, ng);
            hipLaunchKernelGGL((pick_thr_kernel<T>), dim3(1, 1, nb), dim3(1024), 0, st, (const T*)q->gmin, ng, q->Ms, q->info);
            hipLaunchKernelGGL((score_filter_batch_kernel<NSW>), dim3(fg, ny), dim3(256), 0, st, b, G);
        } else {
            hipLaunchKernelGGL((score_gmin_kernel<T, 0>), dim3((unsigned)((ng + 3) / 4), 1, nb), dim3(256), 0, st, a, G, ng);
            const unsigned pg = (unsigned)std::min<int64_t>((rows + 1023) / 1024, std::max(q->cus, 1));
            hipLaunchKernelGGL((score_pickfilter_kernel<T, 0>), dim3(pg, 1, nb), dim3(1024), 0, st, a, ng, q->Ms);
        }
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
    // fp32 scan: the scorer's keys are evaluated in fp64 all the same (lambda term, norms), over the fp32 dots
    if (q->exact) launch_score<double, unsigned long long, 8>(q, q->dots64, f, fuse_final);
    else launch_score<double, unsigned long long, 8>(q, (const double*)nullptr, f, fuse_final, q->dots32);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

// The scan of the int8 two-digit image (as_scan.hip, scan_dma_kernel<..., I8>; the image: as_k2bf.hip, quant_i8_kernel) reads half
// the bytes of the fp32 items.  The query is quantised the same way -- q ~ s_q (128 q1 + q2) / 16256 -- on the host (D elements),
// its digits laid out in the lanes' register order: for the 16-byte chunk ci of an image row (slab ci / 8; chunks 0-3 of a slab
// are a1 of 16 columns each, chunks 4-7 a2) 16 bytes `qa` that multiply into the 16384-weighted sum (q1 of those columns for an
// a1 chunk, zeros for an a2 chunk) and 16 bytes `qb` for the 128-weighted one (q2 for an a1 chunk, q1 for an a2 chunk).  The
// dropped q2.a2 and the two quantisation residues are bounded through the measured norms (err_coef_i8 in as_build.hip):
//   |dot - x.q| <= |x||q| (u_q + 1.001 U + v_q V) + roundings,   u_q = s_q |theta_q|_2 / (16256 |q|),  v_q = s_q |q2|_2 / (16256 |q|).
// False: no usable image (none yet and it cannot be made, non-finite items, rows too wide for the DMA scan), a query the
// fp32 scan serves better (zero, non-finite, coefficient beyond 2e-3), or ARROWSPACE_SCAN_FP32=1.
static bool host_query_digits(as_query* q, int64_t d) {
    const as_space* sp = q->sp;
    if (getenv("ARROWSPACE_SCAN_FP32") || !q->hq8 || sp->opts.force_exact) return false;
    bool present = false;
    if (space_i8_image(sp, &present) != AS_OK || !present || sp->dp8 / 2 > 2048) return false;
    // (loops shaped for the host compiler's vectoriser: 768 calls of nearbyint were 4 us in front of every scan)
    float mm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t c = 0;
    for (; c + 8 <= d; c += 8)
        for (int j = 0; j < 8; ++j) {
            const float av = std::fabs(q->hq32[c + j]);
            mm[j] = av > mm[j] ? av : mm[j];
        }
    for (; c < d; ++c) mm[0] = std::max(mm[0], std::fabs(q->hq32[c]));
    float m = 0.0f;
    for (int j = 0; j < 8; ++j) m = std::max(m, mm[j]);
    if (!(m > 0.0f) || !(m < 3.0e38f) || !(q->h_nq > 0.0)) return false;   // (a NaN element: h_nq is NaN)
    const float inv = 16256.0f / m;
    const int64_t dp8 = sp->dp8;
    signed char* out = (signed char*)q->hq8;
    double st2 = 0.0, sa2 = 0.0;
    for (int64_t c0 = 0; c0 < dp8; c0 += 16) {   // 16 columns = one a1 chunk and one a2 chunk of the image row
        signed char q1v[16], q2v[16];
        float tq[16];
        int sa = 0;
        for (int j = 0; j < 16; ++j) {
            const float v = c0 + j < d ? q->hq32[c0 + j] : 0.0f;   // (hq32 is zero beyond d up to dp; dp8 may reach beyond dp)
            const float sc = v * inv;
            const float r = (sc + 12582912.0f) - 12582912.0f;   // round to nearest even: |sc| <= 16256 (1 + 2^-23)
            int qq = (int)r;
            qq = qq > 16256 ? 16256 : (qq < -16256 ? -16256 : qq);
            const int q2 = ((qq + 64 + (1 << 20)) & 127) - 64;
            const int q1 = (qq - q2) >> 7;
            const float th = std::fabs(sc - (float)qq) + 0.004f;
            tq[j] = th * th;
            sa += q2 * q2;
            q1v[j] = (signed char)q1;
            q2v[j] = (signed char)q2;
        }
        float st = 0.0f;
        for (int j = 0; j < 16; ++j) st += tq[j];
        st2 += (double)st * 1.00001;   // (16 fp32 additions)
        sa2 += (double)sa;
        const int64_t slab = c0 >> 6, c4 = (c0 & 63) >> 4;
        signed char* a1chunk = out + ((slab * 8 + c4) * 32);       // the a1 chunk of these columns: qa = q1, qb = q2
        signed char* a2chunk = out + ((slab * 8 + 4 + c4) * 32);   // the a2 chunk: qa = 0 (zeroed once, query_create), qb = q1
        memcpy(a1chunk, q1v, 16);
        memcpy(a1chunk + 16, q2v, 16);
        memcpy(a2chunk + 16, q1v, 16);
        if (q->hq8h) {   // the planar rows of the coarse scan: chunk c0 / 16 = these 16 columns' high digits -- qa = q1, qb = q2
            signed char* hc = (signed char*)q->hq8h + (c0 >> 4) * 32;
            memcpy(hc, q1v, 16);
            memcpy(hc + 16, q2v, 16);
        }
    }
    const double nq = std::sqrt(q->h_nq);
    const double uq = (double)m * std::sqrt(st2) / (16256.0 * nq) * 1.001, vq = (double)m * std::sqrt(sa2) / (16256.0 * nq) * 1.001;
    // (fp32 roundings behind the exact integer sums: ten for the scaling -- and, for rows wider than 1 024 columns, whose wave
    // totals pass 2^24, six more for each of the two wave sums' additions)
    const double rnd = (sp->dp8 > 1024 ? 22.0 : 10.0) * 5.9604644775390625e-8;
    const double coef = uq + 1.001 * sp->u8max + vq * sp->v8max + rnd;
    if (!(coef <= 2.0e-3)) return false;
    q->coef_i8 = coef;
    // the coarse scan drops a2 . (128 q1 + q2) as well: at most V |x||q| (1 + u_q)
    q->coef_i8h = uq + 1.001 * sp->u8max + 1.002 * sp->v8max + rnd;
    q->h_faq = m * (11.313708498984761f / 16256.0f);   // s_q sqrt(128) / 16256
    return true;
}

// The batched pass on the int8 images (as_scan.hip, scan_gemm_kernel<..., I8>): the slots' queries are quantised like the items
// by q_quant_batch_kernel, behind the staging kernel -- which also MEASURES every slot's u_q = s_q |theta|_2 / (16256 |q|) and
// v_q = s_q |q2|_2 / (16256 |q|) into pinned memory.  The host prices the pass's error BEFORE the launch with the values it
// ASSUMES -- 1.05 times what the previous passes over this space measured (queries of one workload quantise alike); before
// the first pass an a-priori bound from the slots' s_q / |q|: |theta_c| <= 1/2 + 0.004, |q2_c| <= 64 over the d columns -- and
// holds the measured values against them when it collects the pass: a pass beyond its assumption is run once more, priced
// with what was measured (a non-finite query: its slots go to the single-query path like any slot that fails a proof).  (Reading the queries on the host costs 50 us per MB: measuring there put 140 us in
// front of every pass of 32 x 4 096 elements; the a-priori bound alone is 1.7 times looser -- at topk = 100 most slots then
// failed the scorer's proof.)
// False (the bf16 / fp32 pass serves the batch): no usable image, an assumed coefficient beyond 2e-3.
static bool host_batch_coef(as_query* q, const double* query_host, int64_t d) {
    const as_space* sp = q->sp;
    if (getenv("ARROWSPACE_SCAN_FP32") || !q->q8img_dev || !q->hx8stat || !q->half_enabled || q->ss.dots_rs != 4 || q->cap != 32 || sp->opts.force_exact)
        return false;
    bool present = false;
    if (space_i8_image(sp, &present) != AS_OK || !present) return false;
    // (a row-sharded index's batched pass -- as_query_scan_batch -- prices a priori always: what a rank measures stays on that
    // rank, and the ranks must take the same reruns)
    double au = q->batch_assume ? sp->uq_est : 0.0, av = q->batch_assume ? sp->vq_est : 0.0;
    if (q->batch_assume == 2) {   // the same queries again, priced with what the device measured on them (search_batch_collect)
        au = q->x8_au;
        av = q->x8_av;
    }
    q->x8_verify = au > 0.0 ? 1 : 0;
    if (!(au > 0.0)) {
        double rmax = 0.0;   // the slots' largest s_q / |q|
        for (int b = 0; b < q->nb; ++b) {
            const double* src = query_host + (int64_t)b * d;
            double m = 0.0, nq = 0.0;
            for (int64_t c = 0; c < d; ++c) {
                const double v = src[c], a = std::fabs(v);
                m = a > m ? a : m;
                nq += v * v;
            }
            if (!(m > 1.0e-30) || !(m < 3.0e38) || !(nq > 0.0) || !(nq < 1.0e300)) return false;
            rmax = std::max(rmax, m / std::sqrt(nq));
        }
        const double sd = std::sqrt((double)d) / 16256.0 * 1.001;
        au = rmax * 0.504 * sd;
        av = rmax * 64.0 * sd;
    }
    // the rounding to fp32 of 128 * acc1 + accx, three additions of the waves' quarters, P - 1 of the passes' partials, two
    // multiplications by the scales, the scales' own roundings (two each)
    int64_t chunk = 0;
    const int P = gemm_chunks(sp->dp8 / 2, &chunk, true);
    const double coef = au + 1.001 * sp->u8max + av * sp->v8max + (double)(12 + P) * 5.9604644775390625e-8;
    if (!(coef <= 2.0e-3)) return false;
    q->coef_i8 = coef;
    q->x8_au = au;
    q->x8_av = av;
    return true;
}

// the pass has run: the slots' measured u_q, v_q against the assumed ones; the space's estimate for the next passes
static bool batch_coef_holds(as_query* q) {
    const as_space* sp = q->sp;
    float mu = 0.0f, mv = 0.0f;
    bool ok = true;
    for (int b = 0; b < q->cap; ++b) {
        const float u = q->hx8stat[2 * b], v = q->hx8stat[2 * b + 1];
        if (!((double)u <= q->x8_au) || !((double)v <= q->x8_av)) ok = false;   // (NaN: a non-finite query)
        if (u == u && v == v) {
            mu = std::max(mu, u);
            mv = std::max(mv, v);
        }
    }
    if (mu > 0.0f) {
        // (the largest residue of 32 queries moves by several per cent from pass to pass -- s_q / |q| is the queries' largest component
        // over their norm --, and a pass beyond its assumption costs a whole second pass: 1.05 x with a decay of 3 % per pass failed
        // two passes in eight on 256 distinct queries of the bench's recipe; the margin costs a few per cent of ONE term of the error)
        sp->uq_est = std::max(1.15 * (double)mu, 0.99 * sp->uq_est);
        sp->vq_est = std::max(1.15 * (double)mv, 0.99 * sp->vq_est);
    }
    if (!ok) {   // what a second pass over the same queries is priced with
        q->x8_au = (double)mu * 1.0005;
        q->x8_av = (double)mv * 1.0005;
        q->x8_nan = 0;
        for (int b = 0; b < q->cap; ++b)
            if (q->hx8stat[2 * b] != q->hx8stat[2 * b] || q->hx8stat[2 * b + 1] != q->hx8stat[2 * b + 1]) q->x8_nan = 1;
    }
    return ok;
}

}  // namespace as
as_gang::~as_gang() { delete[] pre; }
namespace as {

// Gang scans.  as_search is re-entrant, but N host threads that each launch a scan only share the HBM bandwidth one scan already
// takes: 4 threads measured 0.84 x the single-thread rate in round 4.  Callers that arrive together share ONE pass instead
// (scan_tile_gang_kernel: up to four queries per read of the tiles).  The first caller to arrive opens a gang and -- only when
// concurrent callers were seen lately (a lone thread never waits) -- lingers a few microseconds for the callers it expects; those
// that arrive meanwhile file their prepared workspace and wait for the leader's launch; the leader launches one kernel for all
// on its own stream and records an event, every follower's stream waits for the event and each member runs its own tail and
// publishes on its own stream as always.  Per-call results are bit-identical to serial runs: the scan's integer sums are exact
// and everything a search returns is re-evaluated behind it.
static double gang_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static as_status gang_launch(as_query* q, const PreArgs& pre) {
    const as_space* sp = q->sp;
    static const bool gang_off = getenv("ARROWSPACE_GANG") && atoi(getenv("ARROWSPACE_GANG")) == 0;
    static const double linger_us = getenv("ARROWSPACE_GANG_LINGER_US") ? atof(getenv("ARROWSPACE_GANG_LINGER_US")) : 60.0;
    if (gang_off || !q->gang_ok || sp->gang_hint.load(std::memory_order_relaxed) <= 0) return launch_scan(q, pre);
    std::shared_ptr<as_gang> g;
    bool leader = false;
    {
        std::lock_guard<std::mutex> lk(sp->gmu);
        g = sp->gang_open;
        if (g && g->n.load(std::memory_order_relaxed) < 4) {
            const int my = g->n.load(std::memory_order_relaxed);
            g->m[my] = q;
            g->pre[my] = pre;
            g->n.store(my + 1, std::memory_order_release);
            if (my + 1 == 4) sp->gang_open.reset();
        } else {
            g = std::make_shared<as_gang>();
            g->pre = new PreArgs[4];
            g->m[0] = q;
            g->pre[0] = pre;
            g->n.store(1, std::memory_order_release);
            g->seq = sp->gang_seq_next++;
            sp->gang_open = g;
            leader = true;
        }
    }
    if (!leader) {
        // (the leader launches once the scan in front of it has finished; a leader that failed says so)
        while (g->state.load(std::memory_order_acquire) == 0) __builtin_ia32_pause();
        if (g->state.load(std::memory_order_acquire) != 1) return launch_scan(q, pre);
        AS_HIP(hipStreamWaitEvent(q->stream, g->ev, 0));
        return AS_OK;
    }
    // The leader waits for (a) its turn -- the gangs opened before this one have launched, and the scan launched last has
    // FINISHED: one shared scan at a time per space.  Two scans at once only split the bandwidth, and a scan that shares the
    // chip learns its cosine bound late (scan_chunk_end).  Callers that arrive meanwhile join: under load a gang gathers
    // everybody who arrived during the previous scan.  (b) From its turn on, the callers it still expects (the most seen at once
    // over the last 64 searches: they are busy with the tail of their previous scan or between two calls, and back within tens
    // of microseconds), for at most linger_us.  Four closed-loop threads otherwise settle into scans of one and three callers
    // in turn: the one that returns first never meets the others.  (Counting the callers inside as_search at that moment
    // instead let the leader go the instant its partner was between two calls.)  A wait beyond 5 ms gives up waiting (never
    // observed; a stuck event must not hang a search).
    const int expect = std::min(4, std::max(sp->gang_width.load(std::memory_order_relaxed), sp->active_callers.load(std::memory_order_relaxed)));
    const double t_open = gang_now_us();
    double t_turn = -1.0;
    for (;;) {
        const double now = gang_now_us();
        bool turn = sp->gang_seq_launched.load(std::memory_order_acquire) == g->seq;
        if (turn) {
            const hipEvent_t pe = sp->last_scan_ev.load(std::memory_order_acquire);
            if (pe && hipEventQuery(pe) == hipErrorNotReady) turn = false;
            else (void)hipGetLastError();
        }
        if (turn) {
            if (t_turn < 0.0) t_turn = now;
            const int n_now = g->n.load(std::memory_order_acquire);
            if (n_now >= expect || now - t_turn >= linger_us) break;
        }
        if (now - t_open > 5000.0) break;
        __builtin_ia32_pause();
    }
    int n;
    {
        std::lock_guard<std::mutex> lk(sp->gmu);
        if (sp->gang_open == g) sp->gang_open.reset();   // closed: whoever comes now opens the next gang
        n = g->n.load(std::memory_order_acquire);
        sp->gang_scans[n] += 1;
    }
    as_status s = n == 1 ? launch_scan(q, pre) : launch_scan_gang(g->m, g->pre, n, q->stream);
    if (s == AS_OK) {
        if (!q->gang_ev && hipEventCreateWithFlags(&q->gang_ev, hipEventDisableTiming) != hipSuccess) s = AS_EHIP;
        if (s == AS_OK && hipEventRecord(q->gang_ev, q->stream) != hipSuccess) s = AS_EHIP;
        if (s != AS_OK) set_err("gang scan: %s", hipGetErrorString(hipGetLastError()));
    }
    g->ev = q->gang_ev;
    if (s == AS_OK) sp->last_scan_ev.store(q->gang_ev, std::memory_order_release);
    // (the next gang's turn: only ever advanced by the gang whose turn it is -- or by one that gave up waiting, to its own successor)
    int64_t cur = sp->gang_seq_launched.load(std::memory_order_relaxed);
    while (cur <= g->seq && !sp->gang_seq_launched.compare_exchange_weak(cur, g->seq + 1, std::memory_order_release)) {}
    g->state.store(s == AS_OK ? 1 : 2, std::memory_order_release);
    return s;
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
    const bool feature = q->gr && q->gr->lambda_mode == AS_LAMBDA_FEATURE;
    q->host_q = 0;
    q->i8_scan = 0;
    q->coarse = 0;
    q->q64_src = q->q64;
    q->q32_src = q->q32;
    // Host-prepared query: one query, fp32 LDS-DMA scan (rows up to 1024 floats), item mode.  The host converts the query
    // and forms its norm (768 elements: a fraction of a microsecond), the scan reads the fp32 query from pinned host
    // memory and files the norms, the kernels behind it read the fp64 query from the pinned buffer: the staging kernel
    // and the idle gap behind it (6 + 5 us in front of every scan) are gone.
    static const bool no_hostq = getenv("ARROWSPACE_NO_HOSTQ") != nullptr;
    // Rows of 1025 .. 4096 floats: their int8 image is a row of at most 2048 image floats -- the same scan, when the image serves
    // this query; otherwise the generic path below (and no fused tail: search_once reads q->fused_tail back).
    const bool narrow = sp->dp <= 1024, wide8 = !narrow && (sp->dp + 63) / 64 * 64 <= 4096;
    bool host_path = query_host && q->cap == 1 && !q->exact && !feature && (narrow || wide8) && !(q->scan_variant & 4) && q->hq32 && !no_hostq;
    if (host_path) {
        for (int64_t c = 0; c < d; ++c) {
            const double v = query_host[c];
            q->hq[c] = v;
            q->hq32[c] = (float)v;
        }
        q->h_nq = host_query_norm(query_host, d);
        q->h_inq = q->h_nq > 0.0 ? 1.0 / sqrt(q->h_nq) : 0.0;
        q->i8_scan = host_query_digits(q, d) ? 1 : 0;
        if (!narrow && !q->i8_scan) host_path = false;
        // Coarse scan: only where everything it collects is re-evaluated exactly (the fused tail: tau >= 0.4, scan-side scorer
        // candidates) and its error leaves the eps ball recognisable (coefficient up to 4e-2: 1.5e-2 on clustered unit rows)
        q->coarse = 0;
        const char* coarse_env = getenv("ARROWSPACE_SCAN_COARSE");   // (per call: an A/B switch)
        const bool coarse_env_off = coarse_env && atoi(coarse_env) == 0;
        const bool coarse_always = coarse_env && atoi(coarse_env) == 2;   // (tests: probe with every search, whatever the last one did)
        if (host_path && q->allow_coarse && !q->coarse_never && q->i8_scan && (q->fused_tail || q->chainc) && q->hq8h && !coarse_env_off && !q->robust && (!q->crowded_direct || q->chainc) && q->coef_i8h <= 4.0e-2 &&
            (coarse_always || !(q->coarse_off > 0 && (q->coarse_off++ & 63) != 0))) {
            bool have = false;
            if (space_i8h_image(sp, &have) == AS_OK && have) {
                q->coarse = 1;
                q->coef_i8 = q->coef_i8h;   // (coef_query returns it: prefilter, window and proofs all price the coarse dots)
            }
        }
    }
    if (!host_path && !narrow) q->fused_tail = 0;
    if (host_path) {
        q->host_q = 1;
        // (device memory: the scan copies the fp64 query there on its way, PreArgs::q64_dev -- an empty row range launches no scan)
        q->q64_src = r1 > r0 ? q->q64 : q->hq_dev;
        q->q32_src = q->hq32_dev;
        const bool was_clean = q->info_clean != 0;
        if (!q->info_clean) hipLaunchKernelGGL(reset_info_kernel, dim3(1), dim3(64), 0, st, q->info, q->sc_hist);
        q->info_clean = 0;
        if (stats) AS_HIP(hipEventRecord(q->ev[0], st));
        const PreArgs pre = make_pre(q, eps, exclude, !q->robust && !q->crowded_direct);
        // (a scan that may be shared with other callers': the coarse scan of a whole single space that collects scorer candidates,
        // nothing queued on this workspace's stream that the scan must follow, no per-launch timing asked for)
        const int why = sp->gang_hint.load(std::memory_order_relaxed) <= 0 ? 0 : (!q->gang_ok || !pre.sc_enabled) ? 1 : !q->coarse ? 2 : !was_clean ? 3 : stats ? 4 : (r0 != 0 || r1 != sp->n) ? 5 : -1;
        if (why < 0) {
            AS_TRY(gang_launch(q, pre));
        } else {
            if (q->cap == 1) sp->gang_skip[why].fetch_add(1, std::memory_order_relaxed);
            AS_TRY(launch_scan(q, pre));
        }
        if (stats) AS_HIP(hipEventRecord(q->ev[1], st));
        q->ev_valid = stats ? 1 : 0;
        return AS_OK;
    }
    q->info_clean = 0;
    if (query_host == q->hq) {
        // (the slots' queries are already staged here: a second pass over them, search_batch_collect)
    } else if (query_host) {
        memcpy(q->hq, query_host, sizeof(double) * d * q->nb);  // pinned + device-mapped: read in place by the kernel
        if (q->cap > q->nb) memset(q->hq + d * q->nb, 0, sizeof(double) * d * (q->cap - q->nb));  // idle slots: zero query
    } else {
        hipLaunchKernelGGL(q_from_row_kernel, dim3(1), dim3(256), 0, st, sp->x32, sp->x64, sp->d, sp->dp, src_row, q->hq_dev);
    }
    if (query_host && q->cap > 1 && !q->exact && !feature) q->i8_scan = host_batch_coef(q, query_host, d) ? 1 : 0;
    const int nslots = q->cap > 1 ? q->cap : q->nb;
    // feature mode: lambda_q is a functional of the query alone (SPEC F6/F7) -- no neighbour search, no prefilter, and
    // it is computed by the staging kernel itself
    if (feature) AS_TRY(feat_query_prepare(q->gr, q->hq_dev, sp->d, sp->dp, q->q64, q->q32, q->info, nslots, st));
    else hipLaunchKernelGGL(q_prepare_kernel, dim3(1, 1, nslots), dim3(256), 0, st, q->hq_dev, sp->d, sp->dp, q->q64, q->q32, q->info, 1.0);
    if (q->i8_scan && q->cap > 1)
        hipLaunchKernelGGL(q_quant_batch_kernel, dim3((unsigned)q->cap), dim3(256), 0, st, (const float*)q->q32, sp->dp, sp->dp8, q->q8img_dev, q->faqv_dev,
                           (const QInfo*)q->info, q->hx8stat_dev);
    if (stats) AS_HIP(hipEventRecord(q->ev[0], st));
    const PreArgs pre = make_pre(q, eps, exclude, !q->robust && !feature && !q->crowded_direct);
    if (q->defer_pre) {   // (a batched pass that may share its scan with the other workspace of its pair: search_batch_launch_pair)
        *q->defer_pre = pre;
        q->ev_valid = 0;
        return AS_OK;
    }
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

int32_t as_set_tuning(const char* key, int32_t value) {
    if (key && !strcmp(key, "tile_geom")) {
        set_tile_geom(value);
        return 0;
    }
    if (key && !strcmp(key, "tile_dyn")) {
        set_tile_dyn(value);
        return 0;
    }
    if (key && !strcmp(key, "x1_blocks")) {
        set_x1_blocks(value);
        return 0;
    }
    return 1;
}

#ifdef AS_STAMPS
// diagnostic build only (not declared in the public header): the raw stamps of the last finish kernels
int as_debug_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(as::g_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : 1;
}
#endif

}  // extern "C"

namespace as {
static as_status query_alloc(as_query* q);

as_status query_create(const as_space* sp, const as_graph* gr, int cap, as_query** out, int pool_slot) {
    if (!sp || !out) {
        set_err("as_query_create: null argument");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(sp->device));
    as_query* q = new as_query();
    q->cap = cap;
    q->pool_slot = pool_slot;
    q->sp = sp;
    q->gr = gr;
    const bool feature = gr && gr->lambda_mode == AS_LAMBDA_FEATURE;
    q->k = gr && !feature ? gr->gp.k : 1;   // feature mode: the query has no k-NN step
    // the record layout ([topk hits][flags]) must be the same on every rank of a row-sharded index: topk is capped by the
    // items the whole index covers, not by this shard's rows (the kernels take min(topk, rows scanned) themselves)
    q->topk = gr ? std::min<int64_t>(gr->gp.topk, std::max<int64_t>(sp->n, graph_items(gr))) : 1;
    q->Mk = list_width(std::min<int64_t>(q->k, sp->n));
    q->Ms = score_width(q->topk);
    // (batched workspace, deep lists: the pass's coarse keys -- fp16 cosines, the int8 image's error term -- leave the scorer's
    // proof little room between the topk-th score and the Ms-th key in a dense cluster: 64 more candidates per slot.  topk = 100
    // at 1M x 768: 3 % of the slots failed the proof and went to the single-query path, 36 000 queries/s; A/B: ARROWSPACE_BATCH_MS_EXTRA)
    static const int ms_extra = getenv("ARROWSPACE_BATCH_MS_EXTRA") ? atoi(getenv("ARROWSPACE_BATCH_MS_EXTRA")) : 64;
    if (cap > 1 && q->topk > 56 && ms_extra > 0) {
        const int w = score_width(std::min<int64_t>(q->topk + ms_extra, MAX_TOPK));
        if (w > 0) q->Ms = std::max(q->Ms, w);
    }
    if (q->Mk < 0 || q->Ms < 0) {
        set_err("k=%lld exceeds the supported maximum of 120, or topk=%lld the maximum of 1024", (long long)q->k, (long long)q->topk);
        delete q;
        return AS_EUNSUPPORTED;
    }
    const as_status st_alloc = query_alloc(q);
    if (st_alloc != AS_OK) {
        as_query_free(q);   // tolerates a partially built workspace
        return st_alloc;
    }
    *out = q;
    return AS_OK;
}

static as_status query_alloc(as_query* q) {
    const as_space* sp = q->sp;
    const size_t C = (size_t)q->cap;
    q->nwaves = 4096;
    if (const char* ev = getenv("ARROWSPACE_STAGED_X1")) q->x1_off = atoi(ev) == 0 ? 1 : 0;
    if (const char* ev = getenv("ARROWSPACE_SCAN_VARIANT")) q->scan_variant = atoi(ev) & 7;
    if (const char* ev = getenv("ARROWSPACE_GEMM_VARIANT")) q->gemm_variant = atoi(ev);
    q->half_enabled = getenv("ARROWSPACE_BATCH_F32_DOTS") ? 0 : 1;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, sp->device) == hipSuccess) q->cus = prop.multiProcessorCount;
    }
    if (C > 1) {
        // Batched workspaces come in pairs whose passes are meant to overlap (as_search_batch) -- which they only do on
        // DIFFERENT hardware queues, and the runtime multiplexes a process's streams over 4 of them (GPU_MAX_HW_QUEUES): the
        // pair shared one and ran strictly one kernel after the other (tools/hwq_ab.sh: 0.655 ms per pass, 0.61-0.62 with 8
        // queues).  Streams of different priorities never share a queue: every other batched workspace asks for the high one.
        static std::atomic<int> nbatch{0};
        int lo_prio = 0, hi_prio = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
        const int prio = (nbatch.fetch_add(1) & 1) ? hi_prio : (lo_prio + hi_prio) / 2;
        if (hipStreamCreateWithPriority(&q->own_stream, hipStreamNonBlocking, prio) != hipSuccess) {
            (void)hipGetLastError();
            AS_HIP(hipStreamCreateWithFlags(&q->own_stream, hipStreamNonBlocking));
        }
    } else if (q->pool_slot > 0) {
        // the workspaces of as_search's pool (concurrent host threads): a stream priority of its own per slot, for the same
        // reason -- one thread's scan is meant to run under another's finish kernel and host turnaround
        int lo_prio = 0, hi_prio = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
        const int prio = q->pool_slot % 3 == 1 ? hi_prio : (q->pool_slot % 3 == 2 ? lo_prio : (lo_prio + hi_prio) / 2);
        if (hipStreamCreateWithPriority(&q->own_stream, hipStreamNonBlocking, prio) != hipSuccess) {
            (void)hipGetLastError();
            AS_HIP(hipStreamCreateWithFlags(&q->own_stream, hipStreamNonBlocking));
        }
    } else {
        AS_HIP(hipStreamCreateWithFlags(&q->own_stream, hipStreamNonBlocking));
    }
    q->stream = q->own_stream;
    {   // (dp doubles at least: the host-prepared fast path reads the zero padding behind the d elements)
        const size_t hq_n = std::max<size_t>((size_t)sp->d * C, (size_t)sp->dp);
        AS_HIP(hipHostMalloc(&q->hq, sizeof(double) * hq_n, hipHostMallocMapped | hipHostMallocCoherent));
        memset(q->hq, 0, sizeof(double) * hq_n);
    }
    AS_HIP(hipHostGetDevicePointer((void**)&q->hq_dev, q->hq, 0));
    if (C == 1) {
        AS_HIP(hipHostMalloc(&q->hq32, sizeof(float) * sp->dp, hipHostMallocMapped | hipHostMallocCoherent));
        memset(q->hq32, 0, sizeof(float) * sp->dp);
        AS_HIP(hipHostGetDevicePointer((void**)&q->hq32_dev, q->hq32, 0));
        // the query's digit registers for the scan of the int8 image: 32 bytes per 16-byte chunk of an image row
        const size_t dp8 = (size_t)(sp->dp + 63) / 64 * 64;
        AS_HIP(hipHostMalloc(&q->hq8, dp8 * 4, hipHostMallocMapped | hipHostMallocCoherent));
        memset(q->hq8, 0, dp8 * 4);
        AS_HIP(hipHostGetDevicePointer((void**)&q->hq8_dev, q->hq8, 0));
        AS_HIP(hipHostMalloc(&q->hq8h, dp8 * 2, hipHostMallocMapped | hipHostMallocCoherent));
        memset(q->hq8h, 0, dp8 * 2);
        AS_HIP(hipHostGetDevicePointer((void**)&q->hq8h_dev, q->hq8h, 0));
    }
    AS_HIP(hipMalloc(&q->q64, sizeof(double) * sp->dp * C));
    AS_HIP(hipMalloc(&q->q32, sizeof(float) * sp->dp * C));
    AS_HIP(hipMalloc(&q->info, sizeof(QInfo) * C));
    // (on the query's own stream: it is non-blocking, a memset on the null stream may run LATER than the first search)
    AS_HIP(hipMemsetAsync(q->info, 0, sizeof(QInfo) * C, q->stream));
    AS_HIP(hipMalloc(&q->dots32, sizeof(float) * (sp->np + ROW_TILE) * C));
    if (C == 32) {   // batched workspace: the slots' int8 image and scales (q_quant_batch_kernel)
        const size_t dp8 = (size_t)(sp->dp + 63) / 64 * 64;
        AS_HIP(hipMalloc(&q->q8img_dev, dp8 * 2 * C));
        AS_HIP(hipMalloc(&q->faqv_dev, sizeof(float) * C));
        AS_HIP(hipHostMalloc(&q->hx8stat, sizeof(float) * 2 * C, hipHostMallocMapped | hipHostMallocCoherent));
        memset(q->hx8stat, 0, sizeof(float) * 2 * C);
        AS_HIP(hipHostGetDevicePointer((void**)&q->hx8stat_dev, q->hx8stat, 0));
    }
    if (C > 1 && C % 4 == 0) {   // batched workspace: the K-chunk passes' partial dots of rows wider than 768 floats
        int64_t chunk = 0;
        const int npass = gemm_chunks(sp->dp, &chunk, false);   // (sized for the fp32 form: the most passes either form takes)
        if (npass > 1) AS_HIP(hipMalloc(&q->part32, sizeof(float) * (sp->np + ROW_TILE) * C * npass));
    }
    q->ss.dots = C > 1 ? 32 : sp->np + ROW_TILE;
    q->ss.dots_ts = C > 1 ? 32 * (int64_t)C : 32;
    q->ss.dots_rs = C > 1 && C % 4 == 0 ? 4 : 1;
    q->ss.q = sp->dp;
    q->ss.qin = sp->d;
    q->ss.knn = std::max<int64_t>(q->k, 1);
    q->ss.hits = q->topk + 1;
    AS_HIP(hipMalloc(&q->pkey, sizeof(double) * (size_t)q->nwaves * MAX_KLIST));
    AS_HIP(hipMalloc(&q->pidx, sizeof(int) * (size_t)q->nwaves * MAX_KLIST));
    AS_HIP(hipMalloc(&q->ckey_k, sizeof(double) * CAND_CAP * C));
    AS_HIP(hipMalloc(&q->cidx_k, sizeof(int) * CAND_CAP * C));
    AS_HIP(hipMalloc(&q->ckey_s, sizeof(double) * CAND_CAP * C));
    AS_HIP(hipMalloc(&q->cidx_s, sizeof(int) * CAND_CAP * C));
    AS_HIP(hipMalloc(&q->gmin, sizeof(double) * CAND_CAP * C));
    if (C == 1) {   // fused tail: a report of SC_WCAP words per wave of the scan (at most 4 blocks of 4 waves per CU)
        const size_t nwmax = (size_t)16 * std::max(q->cus, 1) + 64;
        AS_HIP(hipMalloc(&q->sc_widx, sizeof(int) * nwmax * SC_WCAP));
        AS_HIP(hipMemsetAsync(q->sc_widx, 0, sizeof(int) * nwmax * SC_WCAP, q->stream));
        AS_HIP(hipMalloc(&q->sc_hist, sizeof(unsigned int) * SC_COPIES * SC_HSTRIDE));
        AS_HIP(hipMemsetAsync(q->sc_hist, 0, sizeof(unsigned int) * SC_COPIES * SC_HSTRIDE, q->stream));
    }
    AS_HIP(hipMalloc(&q->rsel, sizeof(RSel)));
    AS_HIP(hipMalloc(&q->knn, sizeof(as_knn_rec) * q->ss.knn * C));
    AS_HIP(hipMalloc(&q->hits, sizeof(as_hit_rec) * q->ss.hits * C));
    AS_HIP(hipMemsetAsync(q->knn, 0xff, sizeof(as_knn_rec) * q->ss.knn * C, q->stream));   // idx = -1: empty records
    AS_HIP(hipHostMalloc(&q->hout, sizeof(HostOut) * C, hipHostMallocMapped | hipHostMallocCoherent));
    AS_HIP(hipHostGetDevicePointer((void**)&q->hout_dev, q->hout, 0));
    memset(q->hout, 0, sizeof(HostOut) * C);
    for (int i = 0; i < 3; ++i) AS_HIP(hipEventCreate(&q->ev[i]));
    AS_HIP(hipFuncSetAttribute((const void*)knn_finish_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)finish_lds<double>()));
    AS_HIP(hipFuncSetAttribute((const void*)knn_finish_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)finish_lds<float>()));
    AS_HIP(hipFuncSetAttribute((const void*)score_finish_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)score_lds<double>()));
    AS_HIP(hipFuncSetAttribute((const void*)score_finish_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)score_lds<float>()));
    AS_HIP(hipFuncSetAttribute((const void*)hits_final_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((sizeof(double) + sizeof(int)) * HIT_CAP)));
    AS_HIP(hipFuncSetAttribute((const void*)fused_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds()));
    AS_HIP(hipFuncSetAttribute((const void*)staged_score_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds()));
    AS_TRY(set_scan_attrs());
    q->no_fused = getenv("ARROWSPACE_NO_FUSED_TAIL") != nullptr;
    AS_HIP(hipStreamSynchronize(q->stream));   // the fills above: done before a caller can move the query to another stream
    q->r0 = 0;
    q->r1 = sp->n;
    return AS_OK;
}
}  // namespace as

extern "C" {

as_status as_query_create(const as_space* sp, const as_graph* gr, as_query** out) { return query_create(sp, gr, 1, out); }

void as_query_free(as_query* q) {
    if (!q) return;
    hipSetDevice(q->sp->device);
    if (q->stream) hipStreamSynchronize(q->stream);
    if (q->hq) hipHostFree(q->hq);
    if (q->hq32) hipHostFree(q->hq32);
    if (q->hq8) hipHostFree(q->hq8);
    if (q->hq8h) hipHostFree(q->hq8h);
    if (q->gang_ev) {
        hipEvent_t mine = q->gang_ev;
        if (q->sp) q->sp->last_scan_ev.compare_exchange_strong(mine, nullptr);   // (nobody may poll an event that is gone)
        hipEventDestroy(q->gang_ev);
    }
    if (q->hx8stat) hipHostFree(q->hx8stat);
    if (q->q8img_dev) hipFree(q->q8img_dev);
    if (q->faqv_dev) hipFree(q->faqv_dev);
    hipFree(q->q64); hipFree(q->q32); hipFree(q->info); hipFree(q->dots32); hipFree(q->part32);
    if (q->dots64) hipFree(q->dots64);
    hipFree(q->pkey); hipFree(q->pidx); hipFree(q->ckey_k); hipFree(q->cidx_k); hipFree(q->ckey_s); hipFree(q->cidx_s);
    hipFree(q->gmin);
    if (q->sc_widx) hipFree(q->sc_widx);
    if (q->sc_hist) hipFree(q->sc_hist);
    if (q->knn_all) hipFree(q->knn_all);
    if (q->hits_all) hipFree(q->hits_all);
    if (q->xsend) hipFree(q->xsend);
    if (q->xall) hipFree(q->xall);
    if (q->x1_own) hipFree(q->x1_own);
    if (q->xknn) hipFree(q->xknn);
    hipFree(q->rsel);
    if (q->own_records) {
        hipFree(q->knn);
        hipFree(q->hits);
    }
    if (q->hout) hipHostFree(q->hout);
    for (int i = 0; i < 3; ++i)
        if (q->ev[i]) hipEventDestroy(q->ev[i]);
    if (q->own_stream) hipStreamDestroy(q->own_stream);
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
int32_t as_record_capacity(int32_t which) { return which == 0 ? REC_CAP : HIT_CAP; }

as_status as_query_scan(as_query* q, const double* query_host, int64_t d, int64_t row_begin, int64_t row_end) {
    if (!q || !query_host || !q->gr) {
        set_err("as_query_scan: null argument");
        return AS_EINVAL;
    }
    if (q->reuse && !q->robust && q->gr->lambda_mode != AS_LAMBDA_FEATURE) {
        // same query again after a candidate-buffer overflow: threshold repair over the kept dots
        if (row_begin != q->r0 || row_end != q->r1) {
            set_err("as_query_scan: the repair pass must cover the rows of the scan it repairs");
            return AS_EINVAL;
        }
        AS_TRY(knn_repair(q, q->gr->gp.eps, -1));
        return run_knn(q, q->gr->gp.eps, -1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1);
    }
    // Staged search driven by the library (as_query_search_staged announces tau before the scan): the scan collects the
    // scorer's candidates by its cosine bound, as on one GPU (search_once), and ONE kernel does the step between the two
    // exchanges (staged_score_kernel).
    const bool sc = q->staged_tau >= 0.4 && q->staged_tau <= 1.0 && q->gr->lambda_mode != AS_LAMBDA_FEATURE && !q->robust && !q->exact &&
                    !q->no_fused && q->cap == 1 && q->sc_widx && q->sp->dp <= 4096 && !(q->scan_variant & 4) && !q->crowded_direct;
    q->fused_tail = sc ? 1 : 0;
    q->tau_cur = q->staged_tau;
    const as_status qb = query_begin(q, query_host, -1, d, row_begin, row_end, q->gr->gp.eps, -1);
    const bool sc_ran = q->fused_tail != 0;   // (rows of 1025 .. 4096 floats: only when the int8 image served the scan)
    q->fused_tail = 0;
    q->staged_sc = sc_ran && qb == AS_OK && row_end > row_begin ? 1 : 0;
    AS_TRY(qb);
    if (q->gr->lambda_mode == AS_LAMBDA_FEATURE) return AS_OK;   // lambda_q is already there; the k-NN records stay empty
    return run_knn(q, q->gr->gp.eps, -1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, q->coarse);
}

// ---- one exchange per sharded query (staged_x1_kernel / staged_x1_final_kernel above)
static int x1_cap(int world) { return std::max(512, std::min(CAND_CAP, 8192 / std::max(world, 1))); }
static size_t x1_lds_a() { return std::max(fused_lds(), sizeof(double) * Q_LDS_MAX + sizeof(int) * X1_LOCAL_CAP + sizeof(double) * 128); }
static size_t x1_lds_b() { return (2 * sizeof(double) + sizeof(int)) * (size_t)CAND_CAP + (sizeof(double) + sizeof(int)) * (size_t)PRUNE_CAP; }

int64_t as_query_x1_bytes(const as_query* q, int32_t world) {
    if (!q || world < 1) return 0;
    const int64_t raw = (int64_t)sizeof(as_knn_rec) * std::max<int64_t>(q->k, 1) + (int64_t)sizeof(XHead) + (int64_t)sizeof(XCand) * x1_cap(world);
    return (raw + (int64_t)sizeof(as_knn_rec) - 1) / (int64_t)sizeof(as_knn_rec) * (int64_t)sizeof(as_knn_rec);
}

// 1 when a search with this tau may take the one-exchange pass on this workspace (the same answer on every rank: it depends on
// the graph's mode, tau and the workspace's kind only); what a RANK cannot offer at run time travels as a flag in its block.
int32_t as_query_x1_usable(const as_query* q, double tau) {
    // (the A/B switch ARROWSPACE_STAGED_X1 is read ONCE, when the workspace is made -- x1_off -- and a row-sharded index agrees on
    // it over its ranks, as_query_set_x1: ranks that decided per call from their own environment issued different collectives)
    const bool off = q && q->x1_off;
    return q && q->gr && !off && tau >= 0.4 && tau <= 1.0 && q->gr->lambda_mode != AS_LAMBDA_FEATURE && q->cap == 1 && !q->sp->opts.force_exact &&
                   (q->sp->opts.search_mode & 3) == 0 ? 1 : 0;
}

// the block kernels behind a scan that has run (query_begin): this workspace's k-NN records and finished candidates into `send_dev`
static FinishArgs x1_knn_args(as_query* q, void* send_dev) {
    const as_space* sp = q->sp;
    const double eps = q->gr->gp.eps;
    FinishArgs fk = make_finish(q);
    fk.M = q->Mk; fk.epskey = sp->opts.metric == AS_METRIC_L2 ? eps * eps : eps; fk.coef = coef_query(q, false);
    fk.recs = (as_knn_rec*)send_dev; fk.fuse = 0; fk.ck = q->ckey_k; fk.ci = q->cidx_k; fk.from_list = 0;
    return fk;
}

// exact_knn: the single space's coarse scan -- no k-NN block, every block evaluates its share of the k-NN candidates into q->xknn
static as_status x1_launch_block(as_query* q, void* send_dev, int world, bool sc_ran, bool exact_knn = false, int xmode = 0) {
    const as_space* sp = q->sp;
    const int64_t krec = std::max<int64_t>(q->k, 1);
    XHead* head = (XHead*)((char*)send_dev + sizeof(as_knn_rec) * krec);
    XCand* cands = (XCand*)(head + 1);
    const int64_t rows = q->r1 - q->r0;
    FinishArgs fk = x1_knn_args(q, send_dev);
    fk.exhaustive = q->coarse;   // (the coarse scan's k-NN candidates: every one of them evaluated exactly)
    if (exact_knn && !q->xknn) {
        AS_HIP(hipMalloc(&q->xknn, sizeof(XKnn) * CAND_CAP + 16));
        AS_HIP(hipMemsetAsync((char*)q->xknn + sizeof(XKnn) * CAND_CAP, 0, 16, q->stream));
    } else if (exact_knn && q->xknn_dirty && xmode != 2) {
        // (a pass that never reached its finish kernel.  Not in front of the coarse chain's SECOND launch, xmode 2: the first one's last
        // block has left the counters at zero, and the memset is 5 us between two kernels)
        AS_HIP(hipMemsetAsync((char*)q->xknn + sizeof(XKnn) * CAND_CAP, 0, 16, q->stream));
    }
    if (exact_knn) q->xknn_dirty = 1;
    FinishArgs fs = make_finish(q);
    fs.ci = q->sc_widx; fs.sc_nw = sc_ran && rows > 0 ? q->sc_nw : 0;
    fs.sc_hist = q->sc_hist; fs.sc_m = q->last_sc_m; fs.sc_w = q->last_sc_w;
    if (xmode == 2) fs.ci = q->cidx_s;   // (the flat list of the threshold filter: coarse_score_stage)
    static bool attr_set[64] = {};
    if (sp->device >= 0 && sp->device < 64 && !attr_set[sp->device]) {
        AS_HIP(hipFuncSetAttribute((const void*)staged_x1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)x1_lds_a()));
        AS_HIP(hipFuncSetAttribute((const void*)staged_x1_final_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)x1_lds_b()));
        attr_set[sp->device] = true;
    }
    // (a rank that could not collect candidates -- no fused scan for this query here -- says so: every rank reads the flag and
    // the pass is rerun on the two-exchange chain)
    const size_t lds_xk = (sp->dp <= Q_LDS_MAX ? sizeof(double) * (size_t)sp->dp : 0) + (sizeof(double) + sizeof(int) + sizeof(short)) * (size_t)CAND_CAP + 64;
    hipLaunchKernelGGL(staged_x1_kernel, dim3(exact_knn ? g_x1_blocks.load(std::memory_order_relaxed) : 1 + X1_BLOCKS), dim3(1024), exact_knn ? lds_xk : x1_lds_a(), q->stream, fk, fs, head, cands, x1_cap(world),
                       !sc_ran && rows > 0 && xmode == 0 ? 16 : 0, exact_knn ? (XKnn*)q->xknn : (XKnn*)nullptr, xmode);
    AS_HIP(hipGetLastError());
    q->x1_head = head;
    return AS_OK;
}

static as_status x1_launch_final(as_query* q, const void* all_dev, int world, double tau) {
    const as_graph* gr = q->gr;
    const int64_t krec = std::max<int64_t>(q->k, 1);
    hipLaunchKernelGGL(staged_x1_final_kernel, dim3(1), dim3(1024), x1_lds_b(), q->stream, (const char*)all_dev, (int)world, as_query_x1_bytes(q, world),
                       krec, x1_cap(world), q->k, gr->metric, gr->kernel, gr->gp.sigma, gr->gp.p, gr->tau0, tau, q->topk,
                       (int64_t)0x7fffffffffffffffll, q->info, q->hout_dev, q->seq, q->sc_hist, (XHead*)q->x1_head);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

// The coarse chain's scorer (single space; the k-NN records of this query stand in `block`): lambda_q from the records, the
// threshold chain's two selection kernels over the KEPT dots of the coarse scan -- group minima of the scorer key, the Ms-th
// smallest as threshold -- keeping every row up to twice the keys' error above the threshold (SelArgs::margin), and the exact
// evaluation of everything kept by the tail's blocks (staged_x1_kernel, xmode 2).  Nothing is proven from the coarse keys: a
// row of the true top-k has a coarse key within twice the error of the k-th smallest coarse key, so it is among the rows
// kept, and the final kernel ranks exact scores.  Serves the queries the scan cannot collect candidates for by a cosine window
// -- tau below 0.4, data whose background cosine fills the window (isotropic rows, embeddings with a large common direction) --
// in ONE pass over the one-byte image instead of a pass over the two-byte image and the proof-carrying chain.
static as_status coarse_score_stage(as_query* q, void* block, double tau) {
    const as_graph* gr = q->gr;
    const int64_t krec = std::max<int64_t>(q->k, 1);
    hipStream_t st = q->stream;
    hipLaunchKernelGGL(q_lambda_kernel, dim3(1), dim3(64), 0, st, (const as_knn_rec*)block, krec, krec, krec, q->k, gr->metric, gr->kernel, gr->gp.sigma,
                       gr->gp.p, gr->tau0, q->info, 1);
    const int64_t rows = q->r1 - q->r0;
    int64_t G = (rows + CAND_CAP - 1) / CAND_CAP;
    G = std::max<int64_t>(64, (G + 63) / 64 * 64);
    const int ng = (int)((rows + G - 1) / G);
    SelArgs<double> a = make_sel<double>(q, (const double*)nullptr, q->Ms, -1);
    a.dots32 = q->dots32;
    a.tau = tau;
    a.margin = 2.0 * (tau * coef_query(q, false) * 1.0001 + 1.0e-9);
    hipLaunchKernelGGL((score_gmin_kernel<double, 0>), dim3((unsigned)((ng + 3) / 4), 1, 1), dim3(256), 0, st, a, G, ng);
    const unsigned pg = (unsigned)std::min<int64_t>((rows + 1023) / 1024, std::max(q->cus, 1));
    hipLaunchKernelGGL((score_pickfilter_kernel<double, 0>), dim3(pg, 1, 1), dim3(1024), 0, st, a, ng, q->Ms);
    AS_HIP(hipGetLastError());
    return x1_launch_block(q, block, 1, false, true, 2);
}

as_status as_query_x1_begin(as_query* q, const double* query_host, int64_t d, int64_t row_begin, int64_t row_end, double tau, void* send_dev,
                            int32_t world) {
    if (!q || !query_host || !q->gr || !send_dev || world < 1 || !as_query_x1_usable(q, tau)) {
        set_err("as_query_x1_begin: null argument, or a search the one-exchange pass does not serve (as_query_x1_usable)");
        return AS_EINVAL;
    }
    if ((int64_t)world * std::max<int64_t>(q->k, 1) > REC_CAP) {
        set_err("as_query_x1_begin: %d ranks x k = %lld exceed the merge capacity %d", world, (long long)q->k, REC_CAP);
        return AS_EUNSUPPORTED;
    }
    const as_space* sp = q->sp;
    AS_HIP(hipSetDevice(sp->device));
    q->exact = 0;
    q->robust = 0;
    q->reuse = 0;
    // the block's header starts a pass at zero: the finish kernel of the previous pass over the same block left it so
    // (staged_x1_final_kernel); a first pass, another block or a pass that never reached its finish clears it here
    void* head = (char*)send_dev + sizeof(as_knn_rec) * std::max<int64_t>(q->k, 1);
    if (q->x1_dirty || q->x1_head != head) AS_HIP(hipMemsetAsync(head, 0, sizeof(XHead), q->stream));
    q->x1_dirty = 1;
    q->x1_head = head;
    const bool sc = !q->no_fused && q->sc_widx && sp->dp <= 4096 && !(q->scan_variant & 4) && !q->crowded_direct;
    q->fused_tail = sc ? 1 : 0;
    q->tau_cur = tau;
    q->allow_coarse = sc ? 1 : 0;   // (the block kernels evaluate every k-NN candidate of a coarse scan: x1_launch_block, exact_knn)
    q->sc_late = sc ? 1 : 0;
    const as_status qb = query_begin(q, query_host, -1, d, row_begin, row_end, q->gr->gp.eps, -1);
    q->allow_coarse = 0;
    q->sc_late = 0;
    const bool sc_ran = q->fused_tail != 0;   // (rows of 1025 .. 4096 floats: only when the int8 image served the scan)
    q->fused_tail = 0;
    q->staged_sc = 0;
    AS_TRY(qb);
    return x1_launch_block(q, send_dev, world, sc_ran, q->coarse != 0);
}

as_status as_query_x1_finish(as_query* q, const void* all_dev, int32_t world, double tau, int64_t* out_idx, double* out_score, int64_t* out_len,
                             double* out_lambda_q) {
    if (!q || !q->gr || !all_dev || world < 1 || !out_idx || !out_score || !out_len) {
        set_err("as_query_x1_finish: null argument");
        return AS_EINVAL;
    }
    q->seq += 1;
    AS_TRY(x1_launch_final(q, all_dev, world, tau));
    if (q->ev_valid) AS_HIP(hipEventRecord(q->ev[2], q->stream));
    AS_TRY(wait_published(q));
    q->x1_dirty = 0;
    q->xknn_dirty = 0;
    if (q->coarse) q->coarse_off = (q->hout->overflow & 5) ? 1 : 0;   // (this rank's next 63 scans: the two-digit image; the pass itself is rerun by every rank alike)
    q->info_clean = q->hout->state_reset ? 1 : 0;
    q->x1_passes += 1;
    return collect(q, out_idx, out_score, out_len, out_lambda_q);
}

int32_t as_query_x1_redo(const as_query* q) { return q && (q->hout->overflow & 4) ? 1 : 0; }
// allowed = 0: the following one-exchange passes of this workspace scan both digits of the image (the retry of a pass whose
// coarse candidates did not fit, on every rank alike); 1: the default again
int32_t as_query_x1_enabled(const as_query* q) { return q && !q->x1_off ? 1 : 0; }
void as_query_set_x1(as_query* q, int32_t enabled) {
    if (q) q->x1_off = enabled ? 0 : 1;
}
void as_query_set_coarse(as_query* q, int32_t allowed) {
    if (q) q->coarse_never = allowed ? 0 : 1;
}
int64_t as_query_x1_passes(const as_query* q) { return q ? q->x1_passes : 0; }

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
    if (gr->lambda_mode == AS_LAMBDA_FEATURE) return AS_OK;   // computed by as_query_scan, identically on every rank
    q->staged_recs = recs_dev;
    q->staged_m = m;
    if (q->staged_sc) return AS_OK;   // lambda_q is formed by staged_score_kernel (as_query_score), in front of the scorer
    hipLaunchKernelGGL(q_lambda_kernel, dim3(1), dim3(64), 0, q->stream, recs_dev, m, m, m, q->k, gr->metric, gr->kernel,
                       gr->gp.sigma, gr->gp.p, gr->tau0, q->info, 0);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

as_status as_query_score(as_query* q, double tau) {
    if (!q || !q->gr) {
        set_err("as_query_score: null argument");
        return AS_EINVAL;
    }
    if (q->staged_sc && q->staged_recs && q->r1 > q->r0) {
        const as_graph* gr = q->gr;
        FinishArgs fs = make_finish(q);
        fs.tau = tau; fs.M = q->Ms; fs.hits = q->hits; fs.fuse = 0; fs.hout = nullptr; fs.seq = q->seq; fs.auto_reset = 0;
        fs.ck = q->ckey_s; fs.ci = q->sc_widx; fs.from_list = 0; fs.sc_nw = q->sc_nw;
        const double coef_s = tau * (coef_query(q, false) + 1.0e-14) + 4.0 * 2.220446049250313e-16;
        hipLaunchKernelGGL(staged_score_kernel, dim3(1), dim3(1024), fused_lds(), q->stream, fs, coef_s, (const float*)q->dots32, q->sc_hist,
                           q->staged_recs, q->staged_m, q->k, gr->metric, gr->kernel, gr->gp.sigma, gr->gp.p, gr->tau0);
        AS_HIP(hipGetLastError());
        return AS_OK;
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
    const int64_t topk = q->topk;   // capped by the items of the whole index, not by this shard's rows
    q->seq += 1;
    hipLaunchKernelGGL(hits_final_kernel, dim3(1), dim3(1024), (sizeof(double) + sizeof(int)) * HIT_CAP, st, hits_dev, m, m, m, topk,
                       q->info, q->hout_dev, q->seq);
    AS_HIP(hipGetLastError());
    if (q->ev_valid) AS_HIP(hipEventRecord(q->ev[2], st));
    AS_TRY(wait_published(q));
    return collect(q, out_idx, out_score, out_len, out_lambda_q);
}

// ---- batched staged search: up to as_query_slots(q) queries per pass over this rank's rows; the record buffers hold
// [slot][k] / [slot][topk + 1] records, the all-gathered ones [rank][slot][...] (one collective per step and pass)
as_status as_query_create_batch(const as_space* sp, const as_graph* gr, as_query** out) {
    return query_create(sp, gr, QUERY_BATCH, out);   // (any row width: rows beyond 768 floats are scanned in K-chunk passes)
}
int32_t as_query_slots(const as_query* q) { return q ? q->cap : 0; }

as_status as_query_scan_batch(as_query* q, const double* queries_host, int32_t nb, int64_t d, int64_t row_begin, int64_t row_end) {
    if (!q || !queries_host || !q->gr || nb < 1 || nb > q->cap) {
        set_err("as_query_scan_batch: null argument or more queries than slots");
        return AS_EINVAL;
    }
    q->exact = q->sp->opts.force_exact ? 1 : 0;
    q->robust = 0;
    q->reuse = 0;
    if (q->exact) {
        set_err("as_query_scan_batch: the batched pass is the fp32 fast path only");
        return AS_EUNSUPPORTED;
    }
    q->nb = nb;
    q->batch_assume = 0;
    AS_TRY(query_begin(q, queries_host, -1, d, row_begin, row_end, q->gr->gp.eps, -1));
    if (q->gr->lambda_mode == AS_LAMBDA_FEATURE) return AS_OK;
    q->nb = q->cap;   // idle slots get empty records too: the gathered buffers are read slot by slot
    const as_status s = run_knn(q, q->gr->gp.eps, -1, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
    q->nb = nb;
    return s;
}

as_status as_query_lambda_batch(as_query* q, const as_knn_rec* recs_dev, int32_t nranks) {
    if (!q || !q->gr || !recs_dev || nranks < 1) {
        set_err("as_query_lambda_batch: null argument");
        return AS_EINVAL;
    }
    const as_graph* gr = q->gr;
    if (gr->lambda_mode == AS_LAMBDA_FEATURE) return AS_OK;
    const int64_t per = q->ss.knn, m = per * nranks;
    if (m > REC_CAP) {
        set_err("as_query_lambda_batch: %lld records per query exceed the supported %d", (long long)m, REC_CAP);
        return AS_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(q_lambda_kernel, dim3((unsigned)q->cap), dim3(64), 0, q->stream, recs_dev, m, per, per * q->cap, q->k, gr->metric,
                       gr->kernel, gr->gp.sigma, gr->gp.p, gr->tau0, q->info, 0);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

as_status as_query_score_batch(as_query* q, double tau) {
    if (!q || !q->gr) {
        set_err("as_query_score_batch: null argument");
        return AS_EINVAL;
    }
    const int nb = q->nb;
    q->nb = q->cap;
    const as_status s = run_score(q, tau, 0);
    q->nb = nb;
    return s;
}

// out_idx / out_score: [nb][topk]; out_status[b]: AS_OK, AS_EZEROLAMBDA, or -1 = not provably exact on the batched fast
// path (a candidate buffer overflowed, an a-posteriori check failed): rerun that query on the single-query path
as_status as_query_finish_batch(as_query* q, const as_hit_rec* hits_dev, int32_t nranks, int64_t* out_idx, double* out_score,
                                int64_t* out_len, double* out_lambda_q, int32_t* out_status) {
    if (!q || !hits_dev || !out_idx || !out_score || !out_len || !out_status || nranks < 1) {
        set_err("as_query_finish_batch: null argument");
        return AS_EINVAL;
    }
    const int64_t per = q->ss.hits, m = per * nranks;
    if (m > HIT_CAP) {
        set_err("as_query_finish_batch: %lld records per query exceed the supported %d", (long long)m, HIT_CAP);
        return AS_EUNSUPPORTED;
    }
    hipStream_t st = q->stream;
    const int64_t topk = q->topk;
    q->seq += 1;
    hipLaunchKernelGGL(hits_final_kernel, dim3((unsigned)q->nb), dim3(1024), (sizeof(double) + sizeof(int)) * HIT_CAP, st, hits_dev, m, per,
                       per * q->cap, topk, q->info, q->hout_dev, q->seq);
    AS_HIP(hipGetLastError());
    for (int b = 0; b < q->nb; ++b) {
        AS_TRY(wait_published(q, b));
        const HostOut* h = q->hout + b;
        if (h->overflow || h->knn_inexact || h->score_inexact) {
            out_status[b] = -1;
            out_len[b] = 0;
            continue;
        }
        const as_status s1 = collect(q, out_idx + (int64_t)b * topk, out_score + (int64_t)b * topk, out_len + b,
                                     out_lambda_q ? out_lambda_q + b : nullptr, b);
        out_status[b] = (int32_t)s1;
    }
    return AS_OK;
}

void as_query_set_exact(as_query* q, int32_t flags) {
    if (!q) return;
    q->exact = ((flags & 1) || q->sp->opts.force_exact) ? 1 : 0;   // force_exact: build option, or items outside the fp32-safe range
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

int query_overflow_bits(const as_query* q) { return q->hout->overflow; }
extern "C" int32_t as_query_scan_int8(const as_query* q) { return q ? (q->i8_scan ? 1 + q->coarse : 0) : 0; }   // 0 fp32 items, 1 the int8 image, 2 its high digits alone (coarse scan)   // bit0 k-NN candidates, bit1 scorer's, bit2 the scan's scorer candidates

// one full single-GPU search on q's stream: 6 launches, one host wait
as_status search_once(as_query* q, const double* query, int64_t d, double tau, int mode, int64_t* out_idx, double* out_score,
                      int64_t* out_len, double* out_lambda_q) {
    q->exact = (mode & 1) || q->sp->opts.force_exact;
    q->robust = (mode & 2) ? 1 : 0;
    const bool feature = q->gr->lambda_mode == AS_LAMBDA_FEATURE;
    // Crowded neighbourhoods (an eps that admits thousands of rows, e.g. `eps: 10` of tests/test_3_beir.py under the
    // cosine distance): once a query has overflowed the scan's candidate buffer, the following ones skip the prefilter
    // and derive their neighbours from the dots by threshold straight away -- not a whole selection chain and a host
    // wait later.  Every 64th query probes the plain path again.
    const bool direct = !feature && !q->robust && q->crowded > 0 && (q->crowded++ & 63) != 0;
    q->crowded_direct = direct ? 1 : 0;
    // Fused tail: the scan collects the scorer's candidates by a cosine bound and ONE kernel finishes the query.  The
    // bound's window is (1 - tau) / (2 tau) wide in cosine: pointless below tau = 0.4 (every row would qualify).  A query
    // whose candidates overflow the buffer falls back to the threshold chain over the kept dots, and the following 63
    // queries take that chain directly.
    const bool sc_skip = q->sc_crowded > 0 && (q->sc_crowded++ & 63) != 0;
    const bool want_fused = !feature && !q->robust && !q->exact && !direct && !q->no_fused && !sc_skip && q->cap == 1 && tau >= 0.4 && tau <= 1.0 &&
                            q->sp->dp <= 4096 && !(q->scan_variant & 4);
    q->fused_tail = want_fused ? 1 : 0;
    q->tau_cur = tau;
    static const bool fused_x1_on = !(getenv("ARROWSPACE_FUSED_X1") && atoi(getenv("ARROWSPACE_FUSED_X1")) == 0);
    // Coarse chain: what the fused tail cannot serve -- tau below 0.4, scorer candidates that overflowed lately (sc_skip), crowded
    // neighbourhoods (direct) -- still scans the one-byte image: no scan-side scorer candidates, k-NN candidates from the scan's
    // prefilter or (direct) by threshold over the kept dots, the scorer's by threshold over the kept dots once lambda_q is known,
    // everything evaluated exactly by the tail's blocks (coarse_score_stage).  A query it does not serve cleanly is redone on the
    // two-digit image through the proof-carrying chain, and the next 63 skip it.
    static const bool chainc_on = !(getenv("ARROWSPACE_COARSE_CHAIN") && atoi(getenv("ARROWSPACE_COARSE_CHAIN")) == 0);
    const bool chainc_skip = q->chainc_off > 0 && (q->chainc_off++ & 63) != 0;
    const bool want_chainc = chainc_on && fused_x1_on && !want_fused && !feature && !q->robust && !q->exact && !q->no_fused && q->cap == 1 && tau >= 0.0 && tau <= 1.0 &&
                             q->sp->dp <= 4096 && !(q->scan_variant & 4) && !chainc_skip && !q->coarse_never && (int64_t)std::max<int64_t>(q->k, 1) <= REC_CAP;
    q->chainc = want_chainc ? 1 : 0;
    q->allow_coarse = (want_fused || want_chainc) && fused_x1_on ? 1 : 0;   // (the coarse scan needs the two-launch tail: its k-NN candidates are evaluated by all blocks)
    q->gang_ok = want_fused && fused_x1_on ? 1 : 0;
    q->sc_late = want_fused && fused_x1_on ? 1 : 0;   // (the two-launch tail validates lossy wave reports against the final histogram)
    // (ARROWSPACE_HOST_TIMING=1: host microseconds of the fused path's parts -- preparation + scan launch, the two tail launches,
    // the wait for the publication -- averaged over 200 searches, on stderr)
    static const bool host_timing = getenv("ARROWSPACE_HOST_TIMING") != nullptr;
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double ht0 = host_timing ? now_us() : 0.0;
    const as_status qb = query_begin(q, query, -1, d, 0, q->sp->n, q->gr->gp.eps, -1);
    const double ht1 = host_timing ? now_us() : 0.0;
    q->allow_coarse = 0;
    q->gang_ok = 0;
    q->sc_late = 0;
    const bool fused = q->fused_tail != 0;   // (rows of 1025 .. 4096 floats: only when the int8 image served the scan)
    const bool chainc = q->chainc && q->coarse;   // (the coarse scan did serve this query's scan)
    q->chainc = 0;
    q->crowded_direct = 0;
    q->fused_tail = 0;
    AS_TRY(qb);
    if (chainc) {
        if (!q->x1_own) {
            AS_HIP(hipMalloc(&q->x1_own, (size_t)as_query_x1_bytes(q, 1)));
            AS_HIP(hipMemsetAsync(q->x1_own, 0, (size_t)as_query_x1_bytes(q, 1), q->stream));
        } else if (q->x1_dirty) {
            AS_HIP(hipMemsetAsync(q->x1_own + sizeof(as_knn_rec) * std::max<int64_t>(q->k, 1), 0, 16, q->stream));
        }
        q->x1_dirty = 1;
        q->seq += 1;
        if (direct) AS_TRY(knn_repair(q, q->gr->gp.eps, -1));   // (crowded neighbourhood: the k-NN candidates by threshold over the kept dots)
        AS_TRY(x1_launch_block(q, q->x1_own, 1, false, true, 1));
        AS_TRY(coarse_score_stage(q, q->x1_own, tau));
        AS_TRY(x1_launch_final(q, q->x1_own, 1, tau));
        if (q->ev_valid) AS_HIP(hipEventRecord(q->ev[2], q->stream));
        AS_TRY(wait_published(q));
        q->x1_dirty = 0;
        q->xknn_dirty = 0;
        if (!direct) q->crowded = (q->hout->overflow & 1) ? 1 : 0;
        if (q->hout->overflow || q->hout->knn_inexact || q->hout->score_inexact) {
            // not served cleanly (a candidate list that did not fit): this query again without the coarse chain -- the next search
            // of a crowded neighbourhood takes the threshold repair (crowded, above), anything else sends the next 63 past it
            dbg("coarse chain: overflow bits %d (coefficient %.3e) -> this search again on the two-digit image", q->hout->overflow, q->coef_i8h);
            if (!(!direct && (q->hout->overflow & 1) && !(q->hout->overflow & 6))) q->chainc_off = 1;
            q->coarse_never = 1;
            q->info_clean = 0;
            const as_status redo = search_once(q, query, d, tau, mode, out_idx, out_score, out_len, out_lambda_q);
            q->coarse_never = 0;
            return redo;
        }
        q->chainc_off = 0;
        q->info_clean = q->hout->state_reset ? 1 : 0;
        return collect(q, out_idx, out_score, out_len, out_lambda_q);
    }
    // The fused tail as TWO launches (the one-exchange pass's kernels without an exchange: staged_x1_kernel = the k-NN block beside
    // 16 blocks that finish the scan's scorer candidates to exact cosines, staged_x1_final_kernel = lambda_q, exact scores, ranking)
    // instead of the one 1024-thread block that does the phases one after the other (fused_finish_kernel, ARROWSPACE_FUSED_X1=0).
    // Same box, interleaved (tools/fused_x1_ab.sh): 1M x 768 3 240-3 268 -> 3 340 queries/s, 400k x 384 k = 4 topk = 2 9 560-9 590 ->
    // 10 950-11 120, 200k x 768 8 620-8 720 -> 8 620-8 660.
    static const bool fused_x1 = !(getenv("ARROWSPACE_FUSED_X1") && atoi(getenv("ARROWSPACE_FUSED_X1")) == 0);
    if (fused && fused_x1 && (int64_t)std::max<int64_t>(q->k, 1) <= REC_CAP) {
        if (!q->x1_own) {
            AS_HIP(hipMalloc(&q->x1_own, (size_t)as_query_x1_bytes(q, 1)));
            AS_HIP(hipMemsetAsync(q->x1_own, 0, (size_t)as_query_x1_bytes(q, 1), q->stream));
        } else if (q->x1_dirty) {
            AS_HIP(hipMemsetAsync(q->x1_own + sizeof(as_knn_rec) * std::max<int64_t>(q->k, 1), 0, 16, q->stream));   // (behind the scan, in front of the block kernels)
        }
        q->x1_dirty = 1;
        q->seq += 1;
        AS_TRY(x1_launch_block(q, q->x1_own, 1, true, q->coarse != 0));
        AS_TRY(x1_launch_final(q, q->x1_own, 1, tau));
        const double ht2 = host_timing ? now_us() : 0.0;
        if (q->ev_valid) AS_HIP(hipEventRecord(q->ev[2], q->stream));
        AS_TRY(wait_published(q));
        if (host_timing) {
            static double acc[4] = {0, 0, 0, 0}, last_end = 0.0;
            static int cnt = 0;
            const double ht3 = now_us();
            acc[0] += ht1 - ht0; acc[1] += ht2 - ht1; acc[2] += ht3 - ht2;
            if (last_end > 0.0) acc[3] += ht0 - last_end;
            last_end = ht3;
            if (++cnt == 200) {
                fprintf(stderr, "[pyarrowspace] host us per search: prepare + scan launch %.1f, tail launches %.1f, wait %.1f, between calls (return, caller, entry) %.1f\n",
                        acc[0] / 200, acc[1] / 200, acc[2] / 200, acc[3] / 199);
                acc[0] = acc[1] = acc[2] = acc[3] = 0.0;
                cnt = 0;
                last_end = 0.0;
            }
        }
        q->x1_dirty = 0;
        q->xknn_dirty = 0;
        if (q->coarse && (q->hout->overflow & 5) && chainc_on && !chainc_skip && !q->coarse_never) {
            // the scan's candidate lists did not fit -- a neighbourhood of more than 4 096 rows (bit 0), a cosine window that takes in
            // too many rows (bit 2): this query again as a coarse chain, which derives those lists from the kept dots by threshold;
            // the next 63 searches go there directly
            dbg("coarse scan: candidates did not fit (overflow bits %d) -> the coarse chain for this and the next 63 searches", q->hout->overflow);
            if (q->hout->overflow & 1) q->crowded = 1;
            if (q->hout->overflow & 4) q->sc_crowded = 1;
            q->info_clean = 0;
            return search_once(q, query, d, tau, mode, out_idx, out_score, out_len, out_lambda_q);
        }
        if (q->coarse && (q->hout->overflow & 5)) {
            // the coarse scan's candidates did not fit (its wider windows took in too many rows): the same query on the two-digit
            // image straight away -- the chains behind an overflow would price the coarse dots and fail their proofs
            dbg("coarse scan: candidates did not fit (overflow bits %d, coefficient %.3e; candidates written %d, a block's share that did not fit / 16: %d, blocks "
                "with an overflowed wave report %d, k-NN candidates %d) -> the two-digit scan for this and the next 63 searches",
                q->hout->overflow, q->coef_i8h, q->hout->pad_ & 0xffff, (q->hout->pad_ >> 16) & 0xff, (q->hout->pad_ >> 24) & 0xff, 0);
            q->coarse_off = 1;
            q->coarse_never = 1;   // (for the call below, whatever the switches say: it must not come back here)
            q->info_clean = 0;
            const as_status redo = search_once(q, query, d, tau, mode, out_idx, out_score, out_len, out_lambda_q);
            q->coarse_never = 0;
            return redo;
        }
        q->crowded = (q->hout->overflow & 1) ? 1 : 0;
        q->sc_crowded = (q->hout->overflow & 4) ? 1 : 0;
        if ((q->hout->overflow & 4) && !(q->hout->overflow & 1) && !q->hout->knn_inexact) {
            AS_HIP(hipMemsetAsync(&q->info->sc_cnt, 0, sizeof(int), q->stream));
            AS_HIP(hipMemsetAsync(&q->info->overflow, 0, sizeof(int), q->stream));
            q->seq += 1;
            AS_TRY(run_score(q, tau, 1));
            AS_TRY(wait_published(q));
        }
    } else if (fused) {
        q->seq += 1;
        AS_TRY(run_fused(q, q->gr->gp.eps, tau));
        if (q->ev_valid) AS_HIP(hipEventRecord(q->ev[2], q->stream));
        AS_TRY(wait_published(q));
        q->crowded = (q->hout->overflow & 1) ? 1 : 0;
        q->sc_crowded = (q->hout->overflow & 4) ? 1 : 0;
        if ((q->hout->overflow & 4) && !(q->hout->overflow & 1) && !q->hout->knn_inexact) {
            // the scan's scorer candidates did not fit: lambda_q stands, the scorer runs on the threshold chain
            AS_HIP(hipMemsetAsync(&q->info->sc_cnt, 0, sizeof(int), q->stream));
            AS_HIP(hipMemsetAsync(&q->info->overflow, 0, sizeof(int), q->stream));
            q->seq += 1;
            AS_TRY(run_score(q, tau, 1));
            AS_TRY(wait_published(q));
        }
    } else if (direct) {
        AS_TRY(knn_repair(q, q->gr->gp.eps, -1));
        AS_TRY(run_knn(q, q->gr->gp.eps, -1, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 1));
    } else if (!feature) {
        AS_TRY(run_knn(q, q->gr->gp.eps, -1, 1, nullptr, nullptr, nullptr, nullptr, nullptr));
    }
    if (!fused) {
        q->seq += 1;
        AS_TRY(run_score(q, tau, 1));
        if (q->ev_valid) AS_HIP(hipEventRecord(q->ev[2], q->stream));
        AS_TRY(wait_published(q));
    }
    if (!fused && !direct && !feature && !q->robust) q->crowded = (q->hout->overflow & 1) ? 1 : 0;
    if (!direct && !feature && !q->robust && q->hout->knn_inexact && !(q->hout->overflow & 1)) {
        // near-ties at the k-th distance the fp32 keys cannot order (duplicates, near-duplicates): every row inside
        // the eps bound is still in the candidate buffer -- evaluate them all exactly, no second scan
        AS_TRY(run_knn(q, q->gr->gp.eps, -1, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1));
        AS_HIP(hipMemsetAsync(&q->info->sc_cnt, 0, sizeof(int), q->stream));
        q->seq += 1;
        AS_TRY(run_score(q, tau, 1));
        AS_TRY(wait_published(q));
    }
    if (!direct && !q->robust && (q->hout->overflow & 1)) {
        // more than CAND_CAP rows inside eps: re-derive the candidates from the kept dots, no second scan
        AS_TRY(knn_repair(q, q->gr->gp.eps, -1));
        AS_TRY(run_knn(q, q->gr->gp.eps, -1, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 1));
        q->seq += 1;
        AS_TRY(run_score(q, tau, 1));
        AS_TRY(wait_published(q));
    }
    q->info_clean = q->hout->state_reset ? 1 : 0;   // the publishing kernel cleared the per-search state behind itself
    if (q->coarse) {   // a query the coarse scan did not serve cleanly: the next 63 take the two-digit one
        q->coarse_off = (q->hout->overflow || q->hout->knn_inexact || q->hout->score_inexact) ? 1 : 0;
        if (q->coarse_off) dbg("coarse scan: overflow bits %d, k-NN %d, scorer %d (coefficient %.3e) -> the two-digit scan for the next 63 searches",
                               q->hout->overflow, q->hout->knn_inexact, q->hout->score_inexact, q->coef_i8h);
    }
    return collect(q, out_idx, out_score, out_len, out_lambda_q);
}

// up to QB queries in one pass over the items (filter path, fp32 prefilters); out_status[b] is
// AS_OK / AS_EZEROLAMBDA, or -1 when slot b must be rerun on the single-query path (a candidate
// buffer overflowed or an a-posteriori check failed)
// The pass in two halves, so that two workspaces can alternate: every kernel of a pass is queued by _launch (nothing
// waits), _collect waits for its slots and reads them.  While one workspace's selection and finish kernels run, the
// other's scan -- on its own stream -- already streams the items.
as_status search_batch_launch(as_query* q, const double* queries, int nb, int64_t d, double tau) {
    q->exact = 0;
    q->robust = 0;
    q->nb = nb;
    q->batch_assume = 1;
    AS_TRY(query_begin(q, queries, -1, d, 0, q->sp->n, q->gr->gp.eps, -1));
    if (q->gr->lambda_mode != AS_LAMBDA_FEATURE) AS_TRY(run_knn(q, q->gr->gp.eps, -1, 1, nullptr, nullptr, nullptr, nullptr, nullptr));
    q->seq += 1;
    return run_score(q, tau, 1);
}

// Two passes launched together: both workspaces stage and quantise their queries on their own streams, ONE scan serves the 64
// queries (scan_gemm_dual_kernel, on a's stream; b's stream joins it through events), the selection and finish kernels of the
// two passes run side by side.  Where the pair cannot share (no int8 image for these queries, rows wider than 768 columns) each
// workspace scans for itself as before.
as_status search_batch_launch_pair(as_query* a, as_query* b, const double* qa, int nba, const double* qb, int nbb, int64_t d, double tau) {
    as_query* m[2] = {a, b};
    const double* qs[2] = {qa, qb};
    const int nbs[2] = {nba, nbb};
    PreArgs pre[2];
    for (int s = 0; s < 2; ++s) {
        as_query* q = m[s];
        q->exact = 0;
        q->robust = 0;
        q->nb = nbs[s];
        q->batch_assume = 1;
        q->defer_pre = &pre[s];
        const as_status st = query_begin(q, qs[s], -1, d, 0, q->sp->n, q->gr->gp.eps, -1);
        q->defer_pre = nullptr;
        AS_TRY(st);
    }
    // (Measured and dropped, profiles/r05_batch_dual.txt: this pair's scan ordered behind the other pair's selection and finish
    // kernels -- they starve beside a scan that holds every CU's LDS and registers --: 117 700 against 129 700 queries/s; the scan
    // on a stream of its own at the lowest priority, with blocks that retire during the scan: 77 900 .. 107 000; on a stream whose
    // CU mask leaves 1 .. 3 of every 8 CUs to those kernels: 80 300 .. 93 600 against 131 700.)
    if (scan_dual_ok(a, b)) {
        for (int s = 0; s < 2; ++s)
            if (!m[s]->gang_ev) AS_HIP(hipEventCreateWithFlags(&m[s]->gang_ev, hipEventDisableTiming));
        AS_HIP(hipEventRecord(b->gang_ev, b->stream));          // b's queries are staged ...
        AS_HIP(hipStreamWaitEvent(a->stream, b->gang_ev, 0));
        AS_TRY(launch_scan_dual(a, b, pre[0], pre[1], a->stream));
        AS_HIP(hipEventRecord(a->gang_ev, a->stream));          // ... and its tail follows the shared scan
        AS_HIP(hipStreamWaitEvent(b->stream, a->gang_ev, 0));
        a->sp->batch_dual_scans.fetch_add(1, std::memory_order_relaxed);
    } else {
        for (int s = 0; s < 2; ++s) AS_TRY(launch_scan(m[s], pre[s]));
    }
    for (int s = 0; s < 2; ++s) {
        as_query* q = m[s];
        if (q->gr->lambda_mode != AS_LAMBDA_FEATURE) AS_TRY(run_knn(q, q->gr->gp.eps, -1, 1, nullptr, nullptr, nullptr, nullptr, nullptr));
        q->seq += 1;
        AS_TRY(run_score(q, tau, 1));
    }
    return AS_OK;
}

as_status search_batch_collect(as_query* q, int nb, double tau, int64_t topk, int64_t* out_idx, double* out_score, int64_t* out_len,
                               double* out_lambda_q, int32_t* out_status) {
    bool crowded = false;
    for (int b = 0; b < nb; ++b) {
        AS_TRY(wait_published(q, b));
        crowded = crowded || (q->hout[b].overflow & 1);
    }
    if (crowded && !q->dots_half) {   // (fp16 cosines cannot repair a neighbourhood: those slots go to the single-query path below)
        // some neighbourhood overflowed its candidate buffer: threshold repair of every slot over the kept dots
        AS_TRY(knn_repair(q, q->gr->gp.eps, -1));
        AS_TRY(run_knn(q, q->gr->gp.eps, -1, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 1));
        q->seq += 1;
        AS_TRY(run_score(q, tau, 1));
    }
    // (int8 pass: its error was priced with ASSUMED residue norms of the queries -- hold the measured ones against them)
    const bool held = q->i8_scan && q->cap > 1 ? batch_coef_holds(q) : true;   // (always: the space's estimate follows the measurements)
    bool priced = held || !q->x8_verify;
    if (!priced && !q->x8_nan && q->batch_assume == 1) {
        // the assumption did not hold: ONE more pass over the same slots (their queries are still staged in this workspace's
        // pinned buffer), priced with the measured values -- not 32 single-query searches
        q->batch_assume = 2;
        const as_status s2 = query_begin(q, q->hq, -1, q->sp->d, 0, q->sp->n, q->gr->gp.eps, -1);
        q->batch_assume = 1;
        AS_TRY(s2);
        if (q->gr->lambda_mode != AS_LAMBDA_FEATURE) AS_TRY(run_knn(q, q->gr->gp.eps, -1, 1, nullptr, nullptr, nullptr, nullptr, nullptr));
        q->seq += 1;
        AS_TRY(run_score(q, tau, 1));
        for (int b = 0; b < nb; ++b) AS_TRY(wait_published(q, b));
        priced = !(q->i8_scan && q->cap > 1) || batch_coef_holds(q) || !q->x8_verify;
    }
    for (int b = 0; b < nb; ++b) {
        AS_TRY(wait_published(q, b));
        const HostOut* h = q->hout + b;
        if (h->overflow || h->knn_inexact || h->score_inexact || !priced) {
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
        set_err("k=%lld exceeds the supported maximum of 120", (long long)gp->k);
        return AS_EUNSUPPORTED;
    }
    ws->exact = 1;
    ws->robust = 1;  // rows with many near-ties are exactly the ones that overflow a filter buffer
    AS_TRY(query_begin(ws, nullptr, row, sp->d, 0, sp->n, gp->eps, row));
    return run_knn(ws, gp->eps, row, 0, out_idx, out_key, out_dist, out_gy, out_cnt);
}

}  // namespace as
