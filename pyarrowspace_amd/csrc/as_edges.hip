// Sharded graph stage, the step in front of the variable-count all-to-all (pyarrowspace_amd/dist.py, _exchange_edges): the
// directed edges i -> j of this rank's k-NN lists, bucketed by the rank that owns row j.  Count / scan / scatter over the
// [rows][k] lists in one pass each -- the host used torch.bucketize + a stable argsort + bincount + four gathers (O(E log E)
// and six temporaries of E entries; E = 25 M at an 8M-row shard).  The order inside a bucket is the lists' own (row, slot)
// order: what a stable sort by owner gives, so the receiver sees the same bytes as before.
// Reference path this serves: ArrowSpaceBuilder::build, /root/reference/src/lib.rs:281-331 (the graph stage of a row-sharded build).
#include "as_common.hpp"

namespace as {

constexpr int EB_MAXW = 64;       // ranks
constexpr int EB_CHUNK = 4096;    // list entries per block

struct EbBounds {
    int64_t b[EB_MAXW + 1];   // rank r owns items [b[r], b[r + 1])
};

__device__ __forceinline__ int eb_owner(const EbBounds& bd, int world, int64_t tgt) {
    int o = 0;
    for (int r = 1; r < world; ++r) o += tgt >= bd.b[r] ? 1 : 0;   // (bounds ascend: the count of lower bounds at or below tgt)
    return o;
}

__global__ __launch_bounds__(256) void eb_count_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ cnt, int64_t total, int64_t k,
                                                       EbBounds bd, int world, int* __restrict__ blockcnt) {
    __shared__ int c[EB_MAXW];
    if (threadIdx.x < EB_MAXW) c[threadIdx.x] = 0;
    __syncthreads();
    const int64_t e0 = (int64_t)blockIdx.x * EB_CHUNK;
    for (int64_t e = e0 + threadIdx.x; e < e0 + EB_CHUNK && e < total; e += 256) {
        const int64_t row = e / k;
        if ((int)(e - row * k) < cnt[row]) atomicAdd(&c[eb_owner(bd, world, idx[e])], 1);
    }
    __syncthreads();
    if ((int)threadIdx.x < world) blockcnt[(int64_t)blockIdx.x * world + threadIdx.x] = c[threadIdx.x];
}

// one block: per owner the exclusive prefix of the blocks' counts, on top of the owners' bases; totals[o] for the host
__global__ __launch_bounds__(1024) void eb_scan_kernel(int* __restrict__ blockcnt, int64_t nblocks, int world, int64_t* __restrict__ blockoff,
                                                       int64_t* __restrict__ totals) {
    __shared__ int64_t part[1024];
    __shared__ int64_t base_o[EB_MAXW + 1];
    const int per = (int)((nblocks + 1023) / 1024);
    if (threadIdx.x == 0) base_o[0] = 0;
    for (int o = 0; o < world; ++o) {
        const int64_t b0 = (int64_t)threadIdx.x * per;
        int64_t s = 0;
        for (int64_t b = b0; b < b0 + per && b < nblocks; ++b) s += blockcnt[b * world + o];
        part[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x == 0) {   // (1 024 partial sums: a serial pass is a microsecond)
            int64_t run = 0;
            for (int t = 0; t < 1024; ++t) {
                const int64_t v = part[t];
                part[t] = run;
                run += v;
            }
            totals[o] = run;
            base_o[o + 1] = base_o[o] + run;
        }
        __syncthreads();
        int64_t run = base_o[o] + part[threadIdx.x];
        for (int64_t b = b0; b < b0 + per && b < nblocks; ++b) {
            blockoff[b * world + o] = run;
            run += blockcnt[b * world + o];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void eb_scatter_kernel(const int32_t* __restrict__ idx, const double* __restrict__ dist, const double* __restrict__ gy,
                                                         const int32_t* __restrict__ cnt, int64_t total, int64_t k, int64_t row0, EbBounds bd, int world,
                                                         const int64_t* __restrict__ blockoff, int32_t* __restrict__ out_ints, double* __restrict__ out_reals) {
    __shared__ int wc[4][EB_MAXW];
    __shared__ int64_t run[EB_MAXW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if ((int)threadIdx.x < world) run[threadIdx.x] = blockoff[(int64_t)blockIdx.x * world + threadIdx.x];
    __syncthreads();
    const int64_t e0 = (int64_t)blockIdx.x * EB_CHUNK;
    for (int64_t eb = e0; eb < e0 + EB_CHUNK && eb < total; eb += 256) {   // (block-uniform trip count)
        const int64_t e = eb + threadIdx.x;
        int owner = -1;
        int64_t row = 0;
        int32_t tgt = 0;
        if (e < total) {
            row = e / k;
            if ((int)(e - row * k) < cnt[row]) {
                tgt = idx[e];
                owner = eb_owner(bd, world, tgt);
            }
        }
        int before = 0;   // entries of my owner in lower lanes of my wave
        for (int o = 0; o < world; ++o) {
            const unsigned long long m = __ballot(owner == o);
            if (owner == o) before = __popcll(m & ((1ull << lane) - 1));
            if (lane == 0) wc[w][o] = __popcll(m);
        }
        __syncthreads();
        if (owner >= 0) {
            int64_t pos = run[owner] + before;
            for (int w2 = 0; w2 < w; ++w2) pos += wc[w2][owner];
            out_ints[2 * pos] = (int32_t)(tgt - bd.b[owner]);
            out_ints[2 * pos + 1] = (int32_t)(row0 + row);
            out_reals[2 * pos] = dist[e];
            out_reals[2 * pos + 1] = gy[e];
        }
        __syncthreads();
        if ((int)threadIdx.x < world) run[threadIdx.x] += wc[0][threadIdx.x] + wc[1][threadIdx.x] + wc[2][threadIdx.x] + wc[3][threadIdx.x];
        __syncthreads();
    }
}

}  // namespace as

using namespace as;

extern "C" as_status as_edges_bucket(const int32_t* idx_dev, const double* dist_dev, const double* gy_dev, const int32_t* cnt_dev, int64_t rows,
                                     int64_t k, int64_t row0, const int64_t* bounds_host, int32_t world, int32_t* out_ints_dev, double* out_reals_dev,
                                     int64_t* out_counts_host, void* hip_stream) {
    if (!idx_dev || !dist_dev || !gy_dev || !cnt_dev || !bounds_host || !out_ints_dev || !out_reals_dev || !out_counts_host || rows < 0 || k < 1 ||
        world < 1 || world > EB_MAXW) {
        set_err("as_edges_bucket: null argument, or more than %d ranks", EB_MAXW);
        return AS_EINVAL;
    }
    for (int r = 0; r < world; ++r) out_counts_host[r] = 0;
    const int64_t total = rows * k;
    if (total == 0) return AS_OK;
    EbBounds bd;
    for (int r = 0; r <= world; ++r) bd.b[r] = bounds_host[r];
    for (int r = world + 1; r <= EB_MAXW; ++r) bd.b[r] = bounds_host[world];
    hipStream_t st = (hipStream_t)hip_stream;
    const int64_t nblocks = (total + EB_CHUNK - 1) / EB_CHUNK;
    dev_tmp<int> blockcnt;
    dev_tmp<int64_t> blockoff, totals;
    AS_HIP(blockcnt.alloc((size_t)nblocks * world));
    AS_HIP(blockoff.alloc((size_t)nblocks * world));
    AS_HIP(totals.alloc(EB_MAXW));
    hipLaunchKernelGGL(eb_count_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, idx_dev, cnt_dev, total, k, bd, (int)world, (int*)blockcnt);
    hipLaunchKernelGGL(eb_scan_kernel, dim3(1), dim3(1024), 0, st, (int*)blockcnt, nblocks, (int)world, (int64_t*)blockoff, (int64_t*)totals);
    hipLaunchKernelGGL(eb_scatter_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, idx_dev, dist_dev, gy_dev, cnt_dev, total, k, row0, bd, (int)world,
                       (const int64_t*)blockoff, out_ints_dev, out_reals_dev);
    AS_HIP(hipGetLastError());
    AS_HIP(hipMemcpyAsync(out_counts_host, (int64_t*)totals, sizeof(int64_t) * world, hipMemcpyDeviceToHost, st));
    AS_HIP(hipStreamSynchronize(st));
    return AS_OK;
}
