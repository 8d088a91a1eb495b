// Feature mode of the index build and of the query's lambda on gfx950 (AS_LAMBDA_FEATURE): the synthetic index
// the reference's notes document -- lambda = tau E/(E+tau) + (1-tau) G with E = x^T L x / x^T x and G the clipped
// dispersion of the edgewise Dirichlet shares on an F x F feature-space Laplacian (/root/reference/TAUMODE.md:8,
// 12-27), whose nodes are the D columns of the item matrix (GRAPH_VARIABLES.md:17) joined by the same
// distance / eps / k / kernel rules as GRAPH_VARIABLES.md:7-10.  SPEC F1-F7 = DESIGN.md section 2.
//   FK1  gram_f64_kernel     X^T X over a row range, fp64 MFMA (v_mfma_f64_16x16x4_f64), split over rows  -- MFMA-bound
//   FK1b gram_reduce_kernel  fixed-order sum of the row-split partials, mirrored                            -- HBM-bound
//   FK2  feat_knn_kernel     per column: keys to all other columns, eps, (key, index) rank, first k
//   (K3)  csr_from_knn       union symmetrisation, weights, degrees (shared with the item graph)
//   FK3  feat_energy_kernel  per item: T = sum_{a<b} w_ab (x_a - x_b)^2, sum of squares, E and G            -- LDS-gather-bound
//   FK4  feat_qlambda_kernel the same functional for a query (one wave per query slot)
#include <chrono>
#include <vector>

#include "as_query.hpp"

namespace as {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ FK1 Gram of the columns
// Block = 4 waves, output tile 128 x 128 of the Gram (column tiles ta <= tb), wave (wi, wj) owns 64 x 64 of it as
// 4 x 4 accumulators of 16 x 16.  The K loop runs over the items: 32 rows of the two 128-column strips are
// staged in LDS per step (as stored: fp32, or fp64 when the index keeps an fp64 copy) and widened to fp64 at
// the fragment read.  A operand = X^T (lane l: column l & 15 of row l >> 4), B operand = X, so both fragments
// are the same kind of read.  Row stride 272 elements: the two rows a 32-lane LDS group touches land in disjoint
// bank halves.  Every Gram entry is one k-ordered fp64 fma chain over the rows of a split, whatever tile or wave
// computes it -- duplicate columns give bit-identical entries.
constexpr int GT = 128;        // Gram tile edge
constexpr int GR = 32;         // rows per LDS stage
constexpr int GLD = 2 * GT + 16;   // LDS row stride in elements

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gram_f64_kernel(
    const T* __restrict__ x, int64_t ld, int64_t ncols, int64_t r0, int64_t r1, int64_t rows_per_split, int ntile,
    double* __restrict__ part, int64_t d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* S = (T*)smem;   // [GR][GLD]: columns [0,128) = strip ta, [128,256) = strip tb
    // tile pair from the linear index: (ta, tb) with ta <= tb, row-major over the upper triangle
    int ta = 0, rem = (int)blockIdx.x;
    while (rem >= ntile - ta) {
        rem -= ntile - ta;
        ++ta;
    }
    const int tb = ta + rem;
    const int64_t s0 = r0 + (int64_t)blockIdx.y * rows_per_split;
    const int64_t s1 = s0 + rows_per_split < r1 ? s0 + rows_per_split : r1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wi = w >> 1, wj = w & 1;
    const int lk = lane >> 4, lc = lane & 15;
    f64x4 acc[4][4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = f64x4{0.0, 0.0, 0.0, 0.0};
    // staging map: thread t loads 4 consecutive columns (t & 63) * 4 of rows (t >> 6) + 4 i
    const int sc = (tid & 63) * 4, sr = tid >> 6;
    const int64_t gcol = (sc < GT ? (int64_t)ta * GT + sc : (int64_t)tb * GT + (sc - GT));
    for (int64_t base = s0; base < s1; base += GR) {
        __syncthreads();   // the previous stage has been consumed
#pragma unroll
        for (int i = 0; i < GR / 4; ++i) {
            const int r = sr + 4 * i;
            const int64_t row = base + r;
            T v[4] = {0, 0, 0, 0};
            if (row < s1) {
                const T* src = x + row * ld + gcol;
                if (sizeof(T) == 4 && gcol + 3 < ncols) {   // padded fp32 rows: 16-byte aligned
                    const f32x4 q = *(const f32x4*)src;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (T)q[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (gcol + e < ncols) v[e] = src[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) S[r * GLD + sc + e] = v[e];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < GR / 4; ++ks) {
            const T* rowp = S + (4 * ks + lk) * GLD + lc;
            double a[4], b[4];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) a[ib] = (double)rowp[64 * wi + 16 * ib];
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) b[jb] = (double)rowp[GT + 64 * wj + 16 * jb];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ib], b[jb], acc[ib][jb], 0, 0, 0);
        }
    }
    // C/D of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 reg
    double* out = part + (size_t)blockIdx.y * d * d;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t ga = (int64_t)ta * GT + 64 * wi + 16 * ib + lk + 4 * r;
                const int64_t gb = (int64_t)tb * GT + 64 * wj + 16 * jb + lc;
                if (ga < d && gb < d) out[ga * d + gb] = acc[ib][jb][r];
            }
}

// fixed-order sum over the splits; entry (a, b) is taken from the tile pair that computed it (tile(a) <= tile(b))
__global__ void gram_reduce_kernel(const double* __restrict__ part, int nsplit, int64_t d, double* __restrict__ gram) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d * d) return;
    const int64_t a = e / d, b = e % d;
    const bool direct = a / GT <= b / GT;
    const int64_t src = direct ? a * d + b : b * d + a;
    double s = 0.0;
    for (int t = 0; t < nsplit; ++t) s += part[(size_t)t * d * d + src];
    gram[e] = s;
}

as_status feat_gram(const as_space* sp, int64_t r0, int64_t r1, double* gram) {
    const int64_t d = sp->d;
    if (r0 < 0 || r1 > sp->n || r0 > r1) {
        set_err("as_feat_gram: bad row range [%lld,%lld) for n=%lld", (long long)r0, (long long)r1, (long long)sp->n);
        return AS_EINVAL;
    }
    hipStream_t st = sp->stream;
    const int ntile = (int)((d + GT - 1) / GT);
    const int npair = ntile * (ntile + 1) / 2;
    const int64_t rows = r1 - r0;
    // the number of row splits depends on the shape only (never on the device): the summation order, and with it
    // every bit of the Gram, is a function of (n, d, row range)
    int64_t nsplit = (rows + 16383) / 16384;
    nsplit = std::max<int64_t>(1, std::min<int64_t>(nsplit, 64));
    while (nsplit > 1 && (double)nsplit * d * d * 8.0 > 2.0e9) nsplit /= 2;   // partials within 2 GB
    const int64_t rps = ((rows + nsplit - 1) / nsplit + GR - 1) / GR * GR;
    dev_tmp<double> part;
    AS_HIP(part.alloc((size_t)nsplit * d * d));
    if (rows == 0) AS_HIP(hipMemsetAsync(part, 0, sizeof(double) * nsplit * d * d, st));
    if (rows > 0) {
        const dim3 grid((unsigned)npair, (unsigned)nsplit);
        if (sp->x64) {
            const size_t lds = sizeof(double) * GR * GLD;
            AS_HIP(hipFuncSetAttribute((const void*)gram_f64_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(gram_f64_kernel<double>, grid, dim3(256), lds, st, (const double*)sp->x64, d, d, r0, r1, rps, ntile, part, d);
        } else {
            const size_t lds = sizeof(float) * GR * GLD;
            AS_HIP(hipFuncSetAttribute((const void*)gram_f64_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(gram_f64_kernel<float>, grid, dim3(256), lds, st, (const float*)sp->x32, sp->dp, sp->dp, r0, r1, rps, ntile, part, d);
        }
        AS_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)((d * d + 255) / 256)), dim3(256), 0, st, (const double*)part, (int)nsplit, d, gram);
    AS_HIP(hipGetLastError());
    AS_HIP(hipStreamSynchronize(st));
    return AS_OK;
}

// ------------------------------------------------------------------ FK2 k-NN lists of the columns (SPEC F2, F3)
// One block per column a: key to every other column from the Gram, eps test, rank by (key, index), first k.
__global__ __launch_bounds__(256) void feat_knn_kernel(const double* __restrict__ gram, int64_t d, int64_t k, int metric,
                                                       double epskey, int32_t* __restrict__ out_idx, double* __restrict__ out_dist,
                                                       int32_t* __restrict__ out_cnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* key = (double*)smem;         // [d]
    double* dst = key + d;               // [d]
    __shared__ int s_pass;
    const int64_t a = blockIdx.x;
    const double ma = gram[a * d + a];
    if (threadIdx.x == 0) s_pass = 0;
    for (int64_t b = threadIdx.x; b < d; b += blockDim.x) {
        const double g = gram[a * d + b], mb = gram[b * d + b];
        double kk, dd;
        if (metric == AS_METRIC_L2) {
            kk = ma + mb - 2.0 * g;
            kk = kk > 0.0 ? kk : 0.0;
            dd = sqrt(kk);
        } else {
            const double den = sqrt(ma * mb);
            const double c = den > 0.0 ? g / den : 0.0;
            kk = dd = cosine_distance(c);
        }
        key[b] = (b != a && kk <= epskey) ? kk : key_traits<double>::inf();
        dst[b] = dd;
    }
    __syncthreads();
    int npass = 0;
    for (int64_t b = threadIdx.x; b < d; b += blockDim.x) {
        const double kb = key[b];
        if (!(kb < key_traits<double>::inf())) continue;
        npass += 1;
        int64_t rank = 0;
        for (int64_t c = 0; c < d; ++c) rank += lex_less<double>(key[c], (int)c, kb, (int)b) ? 1 : 0;
        if (rank < k) {
            out_idx[a * k + rank] = (int32_t)b;
            out_dist[a * k + rank] = dst[b];
        }
    }
    if (npass) atomicAdd(&s_pass, npass);
    __syncthreads();
    const int cnt = s_pass < k ? s_pass : (int)k;
    for (int64_t t = cnt + threadIdx.x; t < k; t += blockDim.x) out_idx[a * k + t] = -1;
    if (threadIdx.x == 0) out_cnt[a] = cnt;
}

__global__ void feat_diag_lap_kernel(int64_t d, int64_t nnz, const double* __restrict__ gram, const double* __restrict__ w,
                                     double* __restrict__ colm, double* __restrict__ lap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d) colm[i] = gram[i * d + i];
    if (i < nnz) lap[i] = -w[i];   // off-diagonal of L = D - W
}

// edges a < b in ascending (a, b) order: count per row, scan on the host (d rows), fill
__global__ void feat_edge_count_kernel(int64_t d, const int64_t* __restrict__ indptr, const int32_t* __restrict__ col,
                                       int32_t* __restrict__ up) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= d) return;
    int c = 0;
    for (int64_t e = indptr[a]; e < indptr[a + 1]; ++e) c += col[e] > a ? 1 : 0;
    up[a] = c;
}
__global__ void feat_edge_fill_kernel(int64_t d, const int64_t* __restrict__ indptr, const int32_t* __restrict__ col,
                                      const double* __restrict__ w, const int64_t* __restrict__ off, int32_t* __restrict__ ea,
                                      int32_t* __restrict__ eb, double* __restrict__ ew) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= d) return;
    int64_t o = off[a];
    for (int64_t e = indptr[a]; e < indptr[a + 1]; ++e)
        if (col[e] > a) {
            ea[o] = (int32_t)a;
            eb[o] = col[e];
            ew[o] = w[e];
            ++o;
        }
}

static as_status feat_query_attr(int64_t d);   // per-device dynamic-LDS opt-in of the query kernel (defined below)

as_status feat_edges_from_csr(as_graph* gr, hipStream_t st) {
    const int64_t d = gr->n;
    AS_TRY(feat_query_attr(d));
    dev_tmp<int32_t> up;
    dev_tmp<int64_t> off;
    AS_HIP(up.alloc(d));
    AS_HIP(off.alloc(d + 1));
    const unsigned g = (unsigned)((d + 255) / 256);
    hipLaunchKernelGGL(feat_edge_count_kernel, dim3(g), dim3(256), 0, st, d, gr->indptr, gr->indices, up);
    AS_HIP(hipGetLastError());
    std::vector<int32_t> hup(d);
    AS_HIP(hipMemcpyAsync(hup.data(), up, sizeof(int32_t) * d, hipMemcpyDeviceToHost, st));
    AS_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> hoff(d + 1);
    hoff[0] = 0;
    for (int64_t a = 0; a < d; ++a) hoff[a + 1] = hoff[a] + hup[a];
    gr->ne = hoff[d];
    const size_t na = (size_t)std::max<int64_t>(gr->ne, 1);
    AS_HIP(hipMalloc(&gr->ea, sizeof(int32_t) * na));
    AS_HIP(hipMalloc(&gr->eb, sizeof(int32_t) * na));
    AS_HIP(hipMalloc(&gr->ew, sizeof(double) * na));
    AS_HIP(hipMemcpyAsync(off, hoff.data(), sizeof(int64_t) * (d + 1), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(feat_edge_fill_kernel, dim3(g), dim3(256), 0, st, d, gr->indptr, gr->indices, gr->w, (const int64_t*)off, gr->ea, gr->eb, gr->ew);
    AS_HIP(hipGetLastError());
    AS_HIP(hipStreamSynchronize(st));
    return AS_OK;
}

as_status feat_graph(const as_space* sp, const as_graph_params* gp, const double* gram, as_graph* gr) {
    const int64_t d = sp->d;
    hipStream_t st = sp->stream;
    const int64_t k = std::min<int64_t>(gp->k, std::max<int64_t>(d - 1, 1));
    if (2 * sizeof(double) * d > 150 * 1024) {
        set_err("feature mode supports up to %d features (got %lld)", (int)(150 * 1024 / 16), (long long)d);
        return AS_EUNSUPPORTED;
    }
    gr->n = d;
    gr->nitems = sp->n;
    gr->lambda_mode = AS_LAMBDA_FEATURE;
    gr->device = sp->device;
    gr->gp = *gp;
    gr->metric = sp->opts.metric;
    gr->kernel = sp->opts.kernel;
    dev_tmp<int32_t> idx, cnt;
    dev_tmp<double> dist;
    AS_HIP(idx.alloc((size_t)d * k));
    AS_HIP(dist.alloc((size_t)d * k));
    AS_HIP(cnt.alloc(d));
    AS_HIP(hipMemsetAsync(dist, 0, sizeof(double) * d * k, st));
    const double epskey = gr->metric == AS_METRIC_L2 ? gp->eps * gp->eps : gp->eps;
    const size_t lds = 2 * sizeof(double) * d;
    AS_HIP(hipFuncSetAttribute((const void*)feat_knn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(feat_knn_kernel, dim3((unsigned)d), dim3(256), lds, st, gram, d, k, gr->metric, epskey, (int32_t*)idx, (double*)dist, (int32_t*)cnt);
    AS_HIP(hipGetLastError());
    // the pair payload `gy` of the item graph has no use here: the distance array stands in for it
    AS_TRY(csr_from_knn(st, d, k, idx, dist, dist, cnt, gp->sigma, gp->p, gr->kernel, gr));
    AS_HIP(hipMalloc(&gr->colm, sizeof(double) * d));
    const int64_t m = std::max<int64_t>(d, gr->nnz);
    hipLaunchKernelGGL(feat_diag_lap_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d, gr->nnz, gram, gr->w, gr->colm, gr->lap);
    AS_HIP(hipGetLastError());
    AS_TRY(feat_edges_from_csr(gr, st));
    dbg("feature graph: nodes=%lld nnz=%lld edges=%lld", (long long)d, (long long)gr->nnz, (long long)gr->ne);
    return AS_OK;
}

// ------------------------------------------------------------------ FK3 / FK4 Rayleigh energy and dispersion (SPEC F6)
// One wave per vector: the vector sits in LDS as fp64, lanes stride the edge list (a, b, w read once per wave
// step for IPW vectors), gather x_a, x_b from LDS.  Sums: per-lane in edge order, then a fixed butterfly.
constexpr int FE_IPW = 2;      // items per wave (share every edge fetch)
constexpr int FE_WAVES = 4;

struct FeatRes {
    double T, S2, nx;
};

// (E, G) from the accumulated sums; G = sum (e/T)^2 = S2 / T^2 -- evaluated by a second pass over the edges when
// squares of the energies would leave the fp64 range
__device__ __forceinline__ bool feat_sums_safe(double T) { return T > 1e-140 && T < 1e140; }

template <int NV>
__device__ __forceinline__ void feat_accumulate(const double* __restrict__ xs, int64_t stride, int64_t ne,
                                                const int32_t* __restrict__ ea, const int32_t* __restrict__ eb,
                                                const double* __restrict__ ew, double (&T)[NV], double (&S2)[NV]) {
    const int lane = lane_id();
#pragma unroll
    for (int v = 0; v < NV; ++v) T[v] = S2[v] = 0.0;
    for (int64_t e = lane; e < ne; e += 64) {
        const int a = ea[e], b = eb[e];
        const double w = ew[e];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const double t = xs[v * stride + a] - xs[v * stride + b];
            const double en = w * (t * t);
            T[v] += en;
            S2[v] += en * en;
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        T[v] = wave_sum(T[v]);
        S2[v] = wave_sum(S2[v]);
    }
}

__device__ __forceinline__ double feat_ratio_pass(const double* __restrict__ xs, int64_t ne, const int32_t* __restrict__ ea,
                                                  const int32_t* __restrict__ eb, const double* __restrict__ ew, double T) {
    double g = 0.0;
    for (int64_t e = lane_id(); e < ne; e += 64) {
        const double t = xs[ea[e]] - xs[eb[e]];
        const double r = ew[e] * (t * t) / T;
        g += r * r;
    }
    return wave_sum(g);
}

__device__ __forceinline__ void feat_finish(double T, double S2, double nx, const double* xs, int64_t ne, const int32_t* ea,
                                            const int32_t* eb, const double* ew, double& E, double& G) {
    E = nx > 0.0 ? T / nx : 0.0;
    G = 0.0;
    if (T > 0.0) {
        const double g = feat_sums_safe(T) ? S2 / (T * T) : feat_ratio_pass(xs, ne, ea, eb, ew, T);
        G = g < 0.0 ? 0.0 : (g > 1.0 ? 1.0 : g);
    }
}

__global__ __launch_bounds__(64 * FE_WAVES) void feat_energy_kernel(const float* __restrict__ x32, const double* __restrict__ x64,
                                                                   int64_t d, int64_t dp, int64_t r0, int64_t r1, int64_t ne,
                                                                   const int32_t* __restrict__ ea, const int32_t* __restrict__ eb,
                                                                   const double* __restrict__ ew, double* __restrict__ E,
                                                                   double* __restrict__ G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), w = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    double* xs = (double*)smem + (size_t)w * FE_IPW * d;
    const int64_t nwave = (int64_t)gridDim.x * nwv;
    for (int64_t base = r0 + ((int64_t)blockIdx.x * nwv + w) * FE_IPW; base < r1; base += nwave * FE_IPW) {
        double nx[FE_IPW];
#pragma unroll
        for (int v = 0; v < FE_IPW; ++v) {
            const int64_t row = base + v;
            double s = 0.0;
            for (int64_t c = lane; c < d; c += 64) {
                double xv = 0.0;
                if (row < r1) xv = x64 ? x64[row * d + c] : (double)x32[row * dp + c];
                xs[v * d + c] = xv;
                s += xv * xv;
            }
            nx[v] = wave_sum(s);
        }
        AS_LDS_FENCE();
        double T[FE_IPW], S2[FE_IPW];
        feat_accumulate<FE_IPW>(xs, d, ne, ea, eb, ew, T, S2);
#pragma unroll
        for (int v = 0; v < FE_IPW; ++v) {
            const int64_t row = base + v;
            if (row >= r1) continue;   // wave-uniform
            double e, g;
            feat_finish(T[v], S2[v], nx[v], xs + v * d, ne, ea, eb, ew, e, g);
            if (lane == 0) {
                E[row] = e;
                G[row] = g;
            }
        }
        AS_LDS_FENCE();   // the rows are overwritten by the next trip
    }
}

as_status feat_energy(const as_space* sp, const as_graph* gr, int64_t r0, int64_t r1, double* E, double* G) {
    if (r0 < 0 || r1 > sp->n || r0 > r1) {
        set_err("as_feat_energy: bad row range");
        return AS_EINVAL;
    }
    if (r1 == r0) return AS_OK;
    hipStream_t st = sp->stream;
    const size_t per_wave = sizeof(double) * FE_IPW * sp->d;
    const int waves = (int)std::max<size_t>(1, std::min<size_t>(FE_WAVES, (150 * 1024) / per_wave));   // wide rows: fewer waves per block
    const size_t lds = per_wave * waves;
    AS_HIP(hipFuncSetAttribute((const void*)feat_energy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, sp->device) == hipSuccess) cus = prop.multiProcessorCount;
    const int64_t want = (r1 - r0 + waves * FE_IPW - 1) / (waves * FE_IPW);
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (150 * 1024) / std::max<size_t>(lds, 1)));
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)cus * per_cu));
    hipLaunchKernelGGL(feat_energy_kernel, dim3(grid), dim3(64 * waves), lds, st, sp->x32, sp->x64, sp->d, sp->dp, r0, r1, gr->ne,
                       gr->ea, gr->eb, gr->ew, E, G);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

// FK4: lambda_q of the query slots (blockIdx.x = slot), one 1024-thread block each (a single wave walking ~19 000
// edges took 55 us on the critical path of every search); q64 is the zero-padded query the prepare kernel wrote.
// lambda_q == 0 is the reference's zero-lambda assert (src/lib.rs:156-159).
__global__ __launch_bounds__(1024) void feat_qlambda_kernel(const double* __restrict__ q64, int64_t d, int64_t dp, int64_t ne,
                                                            const int32_t* __restrict__ ea, const int32_t* __restrict__ eb,
                                                            const double* __restrict__ ew, double tau0, QInfo* info) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* xs = (double*)smem;
    __shared__ double s_red[3][16];
    const int lane = lane_id(), w = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    q64 += (int64_t)blockIdx.x * dp;
    info += blockIdx.x;
    double nxl = 0.0;
    for (int64_t c = threadIdx.x; c < d; c += blockDim.x) {
        const double v = q64[c];
        xs[c] = v;
        nxl += v * v;
    }
    __syncthreads();
    double T = 0.0, S2 = 0.0;
    for (int64_t e = threadIdx.x; e < ne; e += blockDim.x) {
        const double t = xs[ea[e]] - xs[eb[e]];
        const double en = ew[e] * (t * t);
        T += en;
        S2 += en * en;
    }
    T = wave_sum(T);
    S2 = wave_sum(S2);
    nxl = wave_sum(nxl);
    if (lane == 0) {
        s_red[0][w] = T;
        s_red[1][w] = S2;
        s_red[2][w] = nxl;
    }
    __syncthreads();
    if (w != 0) return;
    T = S2 = nxl = 0.0;
    for (int w2 = 0; w2 < nwv; ++w2) {   // fixed order: deterministic
        T += s_red[0][w2];
        S2 += s_red[1][w2];
        nxl += s_red[2][w2];
    }
    double e, g;
    feat_finish(T, S2, nxl, xs, ne, ea, eb, ew, e, g);
    if (lane == 0) {
        const double lam = tau0 * (e / (e + tau0)) + (1.0 - tau0) * g;
        info->lambda_q = lam;
        info->status = lam == 0.0 ? AS_EZEROLAMBDA : AS_OK;
    }
}

// q_prepare (query staging: fp64 + fp32 copies, norms, state reset) and FK4 in one launch: in feature mode lambda_q
// needs nothing but the query, so the whole pre-scan work of a search is this one kernel.
__global__ __launch_bounds__(1024) void q_prepare_feat_kernel(const double* __restrict__ qin, int64_t d, int64_t dp, double* __restrict__ q64,
                                                              float* __restrict__ q32, QInfo* info, double tau, int64_t ne,
                                                              const int32_t* __restrict__ ea, const int32_t* __restrict__ eb,
                                                              const double* __restrict__ ew, double tau0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* xs = (double*)smem;
    __shared__ double s_red[3][16];
    const int lane = lane_id(), w = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    qin += (int64_t)blockIdx.z * d;
    q64 += (int64_t)blockIdx.z * dp;
    q32 += (int64_t)blockIdx.z * dp;
    info += blockIdx.z;
    double nxl = 0.0;
    for (int64_t c = threadIdx.x; c < dp; c += blockDim.x) {
        const double v = c < d ? qin[c] : 0.0;
        q64[c] = v;
        q32[c] = (float)v;
        if (c < d) xs[c] = v;
        nxl += v * v;
    }
    __syncthreads();
    double T = 0.0, S2 = 0.0;
    for (int64_t e = threadIdx.x; e < ne; e += blockDim.x) {
        const double t = xs[ea[e]] - xs[eb[e]];
        const double en = ew[e] * (t * t);
        T += en;
        S2 += en * en;
    }
    T = wave_sum(T);
    S2 = wave_sum(S2);
    nxl = wave_sum(nxl);
    if (lane == 0) {
        s_red[0][w] = T;
        s_red[1][w] = S2;
        s_red[2][w] = nxl;
    }
    __syncthreads();
    if (w != 0) return;
    T = S2 = nxl = 0.0;
    for (int w2 = 0; w2 < nwv; ++w2) {
        T += s_red[0][w2];
        S2 += s_red[1][w2];
        nxl += s_red[2][w2];
    }
    double e, g;
    feat_finish(T, S2, nxl, xs, ne, ea, eb, ew, e, g);
    if (lane == 0) {
        const double nq = nxl;
        info->nq = nq;
        info->inq = nq > 0.0 ? 1.0 / sqrt(nq) : 0.0;
        info->nq32 = (float)nq;
        info->inq32 = nq > 0.0 ? (float)(1.0 / sqrt(nq)) : 0.0f;
        info->tau = tau;
        reset_query_state(info);
        const double lam = tau0 * (e / (e + tau0)) + (1.0 - tau0) * g;
        info->lambda_q = lam;
        info->status = lam == 0.0 ? AS_EZEROLAMBDA : AS_OK;
    }
}

as_status feat_query_prepare(const as_graph* gr, const double* qin, int64_t d, int64_t dp, double* q64, float* q32, QInfo* info,
                             int nslots, hipStream_t st) {
    hipLaunchKernelGGL(q_prepare_feat_kernel, dim3(1, 1, (unsigned)nslots), dim3(1024), sizeof(double) * gr->n, st, qin, d, dp, q64, q32,
                       info, 1.0, gr->ne, gr->ea, gr->eb, gr->ew, gr->tau0);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

static as_status feat_query_attr(int64_t d) {
    AS_HIP(hipFuncSetAttribute((const void*)feat_qlambda_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * d)));
    AS_HIP(hipFuncSetAttribute((const void*)q_prepare_feat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * d)));
    return AS_OK;
}

as_status feat_query_lambda(const as_graph* gr, const double* q64, int64_t dp, QInfo* info, int nslots, hipStream_t st) {
    const size_t lds = sizeof(double) * gr->n;
    hipLaunchKernelGGL(feat_qlambda_kernel, dim3((unsigned)nslots), dim3(1024), lds, st, q64, gr->n, dp, gr->ne, gr->ea, gr->eb, gr->ew,
                       gr->tau0, info);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

// ------------------------------------------------------------------ the whole feature-mode build on one device
as_status feat_build(as_space* sp, const as_graph_params* gp, as_graph* gr) {
    const int64_t n = sp->n, d = sp->d;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    dev_tmp<double> gram;
    AS_HIP(gram.alloc((size_t)d * d));
    AS_TRY(feat_gram(sp, 0, n, gram));
    const double t1 = now();
    AS_TRY(feat_graph(sp, gp, gram, gr));
    const double t2 = now();
    AS_HIP(hipMalloc(&gr->E, sizeof(double) * n));
    AS_HIP(hipMalloc(&gr->G, sizeof(double) * n));
    AS_TRY(feat_energy(sp, gr, 0, n, gr->E, gr->G));
    AS_HIP(hipStreamSynchronize(sp->stream));
    const double t3 = now();
    gr->e_rows = n;
    AS_TRY(median_lambda(sp, gr, gr->E, gr->G));
    const double t4 = now();
    // stats slots shared with the item build: [1] = the MFMA block (here: the Gram), [2] = refine (feature
    // graph), [4] = graph stage (energies + lambdas), [7] = MFMA flops issued (fp64 here)
    const int64_t ntile = (d + GT - 1) / GT;
    gr->stats[1] = t1 - t0;
    gr->stats[2] = t2 - t1;
    gr->stats[4] = t4 - t2;
    gr->stats[7] = 2.0 * (double)(ntile * (ntile + 1) / 2) * GT * GT * (double)((n + GR - 1) / GR * GR);
    dbg("feature build: gram=%.4fs graph=%.4fs energy=%.4fs lambda=%.4fs tau0=%.6g", t1 - t0, t2 - t1, t3 - t2, t4 - t3, gr->tau0);
    return AS_OK;
}

}  // namespace as

using namespace as;

extern "C" {

as_status as_feat_gram(const as_space* sp, int64_t row_begin, int64_t row_end, double* out_gram_dev) {
    if (!sp || !out_gram_dev) {
        set_err("as_feat_gram: null argument");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(sp->device));
    return feat_gram(sp, row_begin, row_end, out_gram_dev);
}

as_status as_feat_graph(const as_space* sp, const as_graph_params* gp, const double* gram_dev, as_graph** out_graph) {
    if (!sp || !gram_dev || !out_graph) {
        set_err("as_feat_graph: null argument");
        return AS_EINVAL;
    }
    *out_graph = nullptr;
    as_graph_params r;
    AS_TRY(resolve_params(gp, &r));
    AS_HIP(hipSetDevice(sp->device));
    as_graph* gr = new as_graph();
    const as_status s = feat_graph(sp, &r, gram_dev, gr);
    if (s != AS_OK) {
        as_free_graph(gr);
        return s;
    }
    *out_graph = gr;
    return AS_OK;
}

as_status as_feat_energy(const as_space* sp, const as_graph* gr, int64_t row_begin, int64_t row_end, double* E_dev, double* G_dev) {
    if (!sp || !gr || !E_dev || !G_dev || gr->lambda_mode != AS_LAMBDA_FEATURE) {
        set_err("as_feat_energy: null argument or not a feature-mode graph");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(sp->device));
    AS_TRY(feat_energy(sp, gr, row_begin, row_end, E_dev, G_dev));
    AS_HIP(hipStreamSynchronize(sp->stream));
    return AS_OK;
}

// row-sharded form: E / G of ALL n_global items (all-gathered), tau0 over them, this shard's lambdas into the space
as_status as_feat_lambdas_global(as_space* sp, as_graph* gr, const double* E_dev, const double* G_dev, int64_t n_global, int64_t row_offset) {
    if (!sp || !gr || !E_dev || !G_dev || gr->lambda_mode != AS_LAMBDA_FEATURE || row_offset < 0 || row_offset + sp->n > n_global) {
        set_err("as_feat_lambdas_global: null argument, not a feature-mode graph, or rows outside the items");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(sp->device));
    hipStream_t st = sp->stream;
    hipFree(gr->E);
    hipFree(gr->G);
    gr->E = gr->G = nullptr;
    AS_HIP(hipMalloc(&gr->E, sizeof(double) * n_global));
    AS_HIP(hipMalloc(&gr->G, sizeof(double) * n_global));
    AS_HIP(hipMemcpyAsync(gr->E, E_dev, sizeof(double) * n_global, hipMemcpyDeviceToDevice, st));
    AS_HIP(hipMemcpyAsync(gr->G, G_dev, sizeof(double) * n_global, hipMemcpyDeviceToDevice, st));
    dev_tmp<double> lam;
    AS_HIP(lam.alloc(n_global));
    AS_TRY(median_lambda_n(st, n_global, gr->E, gr->G, lam, nullptr, &gr->tau0));
    AS_TRY(lam_slice(st, sp->n, (const double*)lam + row_offset, sp->lam64, sp->lam32));
    AS_HIP(hipStreamSynchronize(st));
    sp->row_offset = row_offset;
    gr->nitems = n_global;
    gr->e_rows = n_global;
    return AS_OK;
}

as_status as_feat_lambdas(as_space* sp, as_graph* gr, const double* E_dev, const double* G_dev) {
    if (!sp || !gr || !E_dev || !G_dev || gr->lambda_mode != AS_LAMBDA_FEATURE) {
        set_err("as_feat_lambdas: null argument or not a feature-mode graph");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(sp->device));
    const int64_t n = sp->n;
    if (!gr->E) AS_HIP(hipMalloc(&gr->E, sizeof(double) * n));
    if (!gr->G) AS_HIP(hipMalloc(&gr->G, sizeof(double) * n));
    if (gr->E != E_dev) AS_HIP(hipMemcpyAsync(gr->E, E_dev, sizeof(double) * n, hipMemcpyDeviceToDevice, sp->stream));
    if (gr->G != G_dev) AS_HIP(hipMemcpyAsync(gr->G, G_dev, sizeof(double) * n, hipMemcpyDeviceToDevice, sp->stream));
    gr->e_rows = n;
    return median_lambda(sp, gr, gr->E, gr->G);
}

}  // extern "C"
