// C ABI glue (include/arrowspace_hip.h): argument validation with the reference shim's
// error behaviour (/root/reference/src/helpers.rs:24-76, src/lib.rs:100-120,140-159),
// handle lifetime, and the composition of the staged build / search entry points.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <new>
#include <vector>

#include "as_common.hpp"

namespace as {

static std::atomic<int> g_debug{0};

std::string& err_slot() {
    static thread_local std::string s;
    return s;
}
void set_err(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err_slot() = buf;
}
bool debug_enabled() { return g_debug.load(std::memory_order_relaxed) != 0; }
void dbg(const char* fmt, ...) {
    if (!debug_enabled()) return;
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    fprintf(stderr, "[pyarrowspace] %s\n", buf);  // src/helpers.rs:18-20
}

static inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

as_status resolve_params(const as_graph_params* gp, as_graph_params* out) {
    if (!gp) {
        set_err("graph_params is required");
        return AS_EINVAL;
    }
    *out = *gp;
    if (!gp->has_sigma) out->sigma = gp->eps * 0.5;  // src/helpers.rs:68-72
    out->has_sigma = 1;
    if (!(out->eps > 0.0) || !(out->sigma > 0.0) || !(out->p > 0.0)) {
        set_err("graph_params: eps, sigma and p must be positive (eps=%g sigma=%g p=%g)", out->eps, out->sigma, out->p);
        return AS_EINVAL;
    }
    if (out->k < 1 || out->topk < 1) {
        set_err("graph_params: k and topk must be >= 1 (k=%lld topk=%lld)", (long long)out->k, (long long)out->topk);
        return AS_EINVAL;
    }
    return AS_OK;
}

// Hard limits of the selection kernels, checked before any upload or GPU work (the reference takes any usize,
// src/helpers.rs:56-63; INTEGRATION.md lists the caps): k-NN candidate lists are one slot per lane of a wave
// or two (k + 8 <= 128), scorer lists live in LDS (topk <= 1024).  Both are capped by the number of items first.
as_status check_limits(const as_graph_params* r, int64_t n, int lambda_mode) {
    const int64_t k = std::min<int64_t>(r->k, std::max<int64_t>(n - 1, 1));
    const int64_t topk = std::min<int64_t>(r->topk, n);
    if (k > 120 && lambda_mode != AS_LAMBDA_FEATURE) {   // the feature graph ranks whole columns: any k up to D - 1
        set_err("graph_params['k']=%lld exceeds the supported maximum of 120 for n=%lld", (long long)r->k, (long long)n);
        return AS_EUNSUPPORTED;
    }
    if (topk > 1024) {
        set_err("graph_params['topk']=%lld exceeds the supported maximum of 1024", (long long)r->topk);
        return AS_EUNSUPPORTED;
    }
    return AS_OK;
}

static as_status pick_device(const as_opts* opts, int* dev) {
    int d = opts ? opts->device : -1;
    if (d < 0) AS_HIP(hipGetDevice(&d));
    AS_HIP(hipSetDevice(d));
    *dev = d;
    return AS_OK;
}

}  // namespace as

using namespace as;

extern "C" {

void as_set_debug(int32_t enabled) { g_debug.store(enabled ? 1 : 0, std::memory_order_relaxed); }
const char* as_last_error(void) { return err_slot().c_str(); }
const char* as_version(void) { return "arrowspace-hip 0.1.0 (gfx950)"; }
int32_t as_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void as_free_space(as_space* sp) {
    if (!sp) return;
    hipSetDevice(sp->device);
    for (int i = 0; i < as_space::QPOOL; ++i)
        if (sp->qpool[i]) as_query_free(sp->qpool[i]);
    if (sp->qcache_b) as_query_free(sp->qcache_b);
    if (sp->qcache_b2) as_query_free(sp->qcache_b2);
    if (sp->qcache_b3) as_query_free(sp->qcache_b3);
    if (sp->qcache_b4) as_query_free(sp->qcache_b4);
    if (sp->stream) hipStreamSynchronize(sp->stream);
    hipFree(sp->x32); hipFree(sp->xs); hipFree(sp->x8); hipFree(sp->x8h); hipFree(sp->fa8); hipFree(sp->x64); hipFree(sp->n64); hipFree(sp->n32); hipFree(sp->inorm32);
    hipFree(sp->lam64); hipFree(sp->lam32);
    if (sp->stream) hipStreamDestroy(sp->stream);
    delete sp;
}

void as_free_graph(as_graph* gr) {
    if (!gr) return;
    hipSetDevice(gr->device);
    hipFree(gr->indptr); hipFree(gr->indices); hipFree(gr->dist); hipFree(gr->gy); hipFree(gr->w); hipFree(gr->lap);
    hipFree(gr->deg); hipFree(gr->ny); hipFree(gr->E); hipFree(gr->G);
    hipFree(gr->ea); hipFree(gr->eb); hipFree(gr->ew); hipFree(gr->colm);
    delete gr;
}

// a space without items yet: validated options, device, private stream
static as_status space_new(int64_t n, int64_t d, const as_opts* opts, as_space** out) {
    if (n >= (int64_t)1 << 31) {
        set_err("as_space_create_dev: n=%lld does not fit 32-bit item indices on one device", (long long)n);
        return AS_EUNSUPPORTED;
    }
    int dev = 0;
    AS_TRY(pick_device(opts, &dev));
    as_space* sp = new as_space();
    sp->device = dev;
    sp->n = n;
    sp->d = d;
    if (opts) sp->opts = *opts;
    sp->opts.device = dev;
    if (sp->opts.metric != AS_METRIC_L2 && sp->opts.metric != AS_METRIC_COSINE) {
        set_err("unknown metric %d", sp->opts.metric);
        delete sp;
        return AS_EINVAL;
    }
    if (sp->opts.kernel != AS_KERNEL_GAUSSIAN && sp->opts.kernel != AS_KERNEL_RATIONAL) {
        set_err("unknown kernel %d", sp->opts.kernel);
        delete sp;
        return AS_EINVAL;
    }
    if (sp->opts.lambda_mode != AS_LAMBDA_ITEM && sp->opts.lambda_mode != AS_LAMBDA_FEATURE) {
        set_err("unknown lambda_mode %d", sp->opts.lambda_mode);
        delete sp;
        return AS_EINVAL;
    }
    hipError_t e = hipStreamCreateWithFlags(&sp->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_err("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete sp;
        return AS_EHIP;
    }
    *out = sp;
    return AS_OK;
}

as_status as_space_create_dev(const void* items_dev, int32_t dtype, int64_t n, int64_t d, int64_t ld, const as_opts* opts,
                              as_space** out_space) {
    if (!out_space) {
        set_err("as_space_create_dev: null output");
        return AS_EINVAL;
    }
    *out_space = nullptr;
    if (!items_dev || n <= 0 || d <= 0) {
        set_err("items must be non-empty 2D array");  // src/helpers.rs:27-29
        return AS_EINVAL;
    }
    if (ld < d || (dtype != AS_DTYPE_F32 && dtype != AS_DTYPE_F64)) {
        set_err("as_space_create_dev: bad leading dimension or dtype");
        return AS_EINVAL;
    }
    as_space* sp = nullptr;
    AS_TRY(space_new(n, d, opts, &sp));
    // the items come from the caller's own stream(s), which a raw pointer does not name: everything queued on the
    // device has to be finished before the ingest (on the space's private non-blocking stream) reads them
    if (hipDeviceSynchronize() != hipSuccess) {
        set_err("as_space_create_dev: %s", hipGetErrorString(hipGetLastError()));
        as_free_space(sp);
        return AS_EHIP;
    }
    as_status s = ingest(sp, items_dev, dtype, ld);
    if (s != AS_OK) {
        as_free_space(sp);
        return s;
    }
    *out_space = sp;
    return AS_OK;
}

as_status as_knn_rows(const as_space* sp, const as_graph_params* gp, int64_t row_begin, int64_t row_end, int32_t* out_idx_dev,
                      double* out_key_dev, double* out_dist_dev, double* out_gy_dev, int32_t* out_cnt_dev) {
    if (!sp || !out_idx_dev || !out_key_dev || !out_dist_dev || !out_gy_dev || !out_cnt_dev) {
        set_err("as_knn_rows: null argument");
        return AS_EINVAL;
    }
    as_graph_params r;
    AS_TRY(resolve_params(gp, &r));
    AS_HIP(hipSetDevice(sp->device));
    return knn_rows(sp, &r, row_begin, row_end, out_idx_dev, out_key_dev, out_dist_dev, out_gy_dev, out_cnt_dev, sp->kstats);
}

as_status as_graph_from_knn(as_space* sp, const as_graph_params* gp, const int32_t* idx_dev, const double* dist_dev,
                            const double* gy_dev, const int32_t* cnt_dev, as_graph** out_graph) {
    if (!sp || !idx_dev || !dist_dev || !gy_dev || !cnt_dev || !out_graph) {
        set_err("as_graph_from_knn: null argument");
        return AS_EINVAL;
    }
    as_graph_params r;
    AS_TRY(resolve_params(gp, &r));
    AS_HIP(hipSetDevice(sp->device));
    as_graph* gr = new as_graph();
    const double t0 = now_s();
    as_status s = graph_from_knn(sp, &r, idx_dev, dist_dev, gy_dev, cnt_dev, gr);
    if (s != AS_OK) {
        as_free_graph(gr);
        return s;
    }
    for (int i = 0; i < 10; ++i) gr->stats[i] = sp->kstats[i];  // k-NN stage of this rank's rows
    gr->stats[4] = now_s() - t0;
    *out_graph = gr;
    return AS_OK;
}

as_status as_graph_from_knn_global(as_space* sp, const as_graph_params* gp, int64_t n_global, int64_t row_offset, const int32_t* idx_dev,
                                   const double* dist_dev, const double* gy_dev, const int32_t* cnt_dev, const double* n64_global_dev,
                                   as_graph** out_graph) {
    if (!sp || !idx_dev || !dist_dev || !gy_dev || !cnt_dev || !n64_global_dev || !out_graph) {
        set_err("as_graph_from_knn_global: null argument");
        return AS_EINVAL;
    }
    as_graph_params r;
    AS_TRY(resolve_params(gp, &r));
    AS_HIP(hipSetDevice(sp->device));
    as_graph* gr = new as_graph();
    const double t0 = now_s();
    as_status s = graph_from_knn_global(sp, &r, n_global, row_offset, idx_dev, dist_dev, gy_dev, cnt_dev, n64_global_dev, gr);
    if (s != AS_OK) {
        as_free_graph(gr);
        return s;
    }
    for (int i = 0; i < 10; ++i) gr->stats[i] = sp->kstats[i];
    gr->stats[4] = now_s() - t0;
    *out_graph = gr;
    return AS_OK;
}

as_status as_graph_shard_csr(as_space* sp, const as_graph_params* gp, int64_t n_global, int64_t row_offset, const int32_t* idx_dev,
                             const double* dist_dev, const double* gy_dev, const int32_t* cnt_dev, int64_t n_in, const int32_t* in_row_dev,
                             const int32_t* in_col_dev, const double* in_dist_dev, const double* in_gy_dev, as_graph** out_graph) {
    if (!sp || !idx_dev || !dist_dev || !gy_dev || !cnt_dev || !out_graph || (n_in > 0 && (!in_row_dev || !in_col_dev || !in_dist_dev || !in_gy_dev))) {
        set_err("as_graph_shard_csr: null argument");
        return AS_EINVAL;
    }
    if (sp->opts.lambda_mode == AS_LAMBDA_FEATURE) {
        set_err("as_graph_shard_csr: the space was created for the feature-mode lambda");
        return AS_EINVAL;
    }
    as_graph_params r;
    AS_TRY(resolve_params(gp, &r));
    AS_HIP(hipSetDevice(sp->device));
    as_graph* gr = new as_graph();
    const double t0 = now_s();
    as_status s = graph_shard_csr(sp, &r, n_global, row_offset, idx_dev, dist_dev, gy_dev, cnt_dev, n_in, in_row_dev, in_col_dev, in_dist_dev,
                                  in_gy_dev, gr);
    if (s != AS_OK) {
        as_free_graph(gr);
        return s;
    }
    for (int i = 0; i < 10; ++i) gr->stats[i] = sp->kstats[i];
    gr->stats[4] = now_s() - t0;
    *out_graph = gr;
    return AS_OK;
}
static as_status graph_rows_copy(const as_graph* gr, const double* src, double* out_dev, const char* who) {
    if (!gr || !src || !out_dev) {
        set_err("%s: null argument or the stage that fills it has not run", who);
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(gr->device));
    // a device-to-device hipMemcpy returns before the copy has run (legacy default stream), and the caller's streams
    // are non-blocking: without the synchronisation its next kernel could read the destination too early
    AS_HIP(hipMemcpy(out_dev, src, sizeof(double) * gr->n, hipMemcpyDeviceToDevice));
    AS_HIP(hipDeviceSynchronize());
    return AS_OK;
}
as_status as_graph_deg_copy(const as_graph* gr, double* out_dev) { return graph_rows_copy(gr, gr ? gr->deg : nullptr, out_dev, "as_graph_deg_copy"); }
as_status as_graph_energy_copy(const as_graph* gr, double* out_dev) { return graph_rows_copy(gr, gr ? gr->E : nullptr, out_dev, "as_graph_energy_copy"); }
int64_t as_graph_row_offset(const as_graph* gr) { return gr ? gr->row0 : 0; }
int64_t as_graph_ncols(const as_graph* gr) { return gr ? (gr->ncols ? gr->ncols : gr->n) : 0; }
int64_t as_graph_nitems(const as_graph* gr) { return gr ? graph_items(gr) : 0; }
as_status as_graph_shard_energy(as_space* sp, as_graph* gr, const double* deg_global_dev, const double* n64_global_dev) {
    if (!sp || !gr || !deg_global_dev || !n64_global_dev || !gr->ncols || gr->n != sp->n) {
        set_err("as_graph_shard_energy: null argument or not the sharded graph of this space");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(sp->device));
    const double t0 = now_s();
    AS_TRY(graph_shard_energy(sp, gr, deg_global_dev, n64_global_dev));
    gr->stats[4] += now_s() - t0;
    return AS_OK;
}
as_status as_graph_shard_lambdas(as_space* sp, as_graph* gr, const double* E_global_dev, int64_t n_global) {
    if (!sp || !gr || !E_global_dev || !gr->ncols || gr->n != sp->n || !gr->E) {
        set_err("as_graph_shard_lambdas: null argument, not the sharded graph of this space, or as_graph_shard_energy has not run");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(sp->device));
    const double t0 = now_s();
    AS_TRY(graph_shard_lambdas(sp, gr, E_global_dev, n_global));
    gr->stats[4] += now_s() - t0;
    return AS_OK;
}

int32_t as_knn_list_width(int64_t k) { return knn_list_width(k); }
double as_space_nmax(const as_space* sp) { return sp ? sp->nmax : 0.0; }
int64_t as_unproven_searches(const as_space* sp) { return sp ? sp->unproven_searches : 0; }
as_status as_space_norms(const as_space* sp, double* out_dev) {
    if (!sp || !out_dev) {
        set_err("as_space_norms: null argument");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(sp->device));
    AS_HIP(hipMemcpy(out_dev, sp->n64, sizeof(double) * sp->n, hipMemcpyDeviceToDevice));
    AS_HIP(hipDeviceSynchronize());   // device-to-device: asynchronous to the host, and the caller's streams are non-blocking
    return AS_OK;
}
int64_t as_space_row_offset(const as_space* sp) { return sp ? sp->row_offset : 0; }

as_status as_ring_i8_stats(as_space* sp, double* out3) {
    if (!sp || !out3) {
        set_err("as_ring_i8_stats: null argument");
        return AS_EINVAL;
    }
    return ring_i8_stats(sp, out3);
}
as_status as_ring_i8_set(as_space* sp, double u_max, double v_max, int32_t usable) {
    if (!sp) {
        set_err("as_ring_i8_set: null argument");
        return AS_EINVAL;
    }
    return ring_i8_set(sp, u_max, v_max, usable);
}
int32_t as_ring_i8(const as_space* sp) { return sp ? sp->ring_i8 : 0; }

as_status as_knn_block(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t row_begin, int64_t row_end,
                       int64_t row_goff, int64_t col_goff, double* p_key_dev, double* p_dist_dev, double* p_gy_dev, int32_t* p_idx_dev,
                       int32_t* p_cnt_dev, float* p_t32_dev) {
    if (!sp || !cols || !p_key_dev || !p_dist_dev || !p_gy_dev || !p_idx_dev || !p_cnt_dev || !p_t32_dev) {
        set_err("as_knn_block: null argument");
        return AS_EINVAL;
    }
    as_graph_params r;
    AS_TRY(resolve_params(gp, &r));
    const int M = knn_list_width(r.k);
    if (M < 0) {
        set_err("graph_params['k']=%lld exceeds the supported maximum of 120", (long long)r.k);
        return AS_EUNSUPPORTED;
    }
    AS_HIP(hipSetDevice(sp->device));
    const double t0 = now_s();
    const as_status s = knn_block(sp, cols, &r, row_begin, row_end, row_goff, col_goff, M, p_key_dev, p_dist_dev, p_gy_dev, p_idx_dev,
                                  p_cnt_dev, p_t32_dev);
    sp->kstats[1] += now_s() - t0;
    return s;
}

as_status as_knn_block_pair(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t row_begin, int64_t row_end,
                            int64_t col_tile_begin, int64_t col_tile_end, int64_t row_goff, int64_t col_goff, const float* col_thr_dev,
                            const float* row_thr_dev, double* p_key_dev, double* p_dist_dev, double* p_gy_dev, int32_t* p_idx_dev, int32_t* p_cnt_dev, float* p_t32_dev,
                            double* q_key_dev, double* q_dist_dev, double* q_gy_dev, int32_t* q_idx_dev, int32_t* q_cnt_dev, float* q_t32_dev) {
    if (!sp || !cols || !p_key_dev || !p_dist_dev || !p_gy_dev || !p_idx_dev || !p_cnt_dev || !p_t32_dev || !q_key_dev || !q_dist_dev ||
        !q_gy_dev || !q_idx_dev || !q_cnt_dev || !q_t32_dev) {
        set_err("as_knn_block_pair: null argument");
        return AS_EINVAL;
    }
    if (sp == cols) {
        set_err("as_knn_block_pair: a block is not paired with itself (as_knn_block)");
        return AS_EINVAL;
    }
    as_graph_params r;
    AS_TRY(resolve_params(gp, &r));
    const int M = knn_list_width(r.k);
    if (M < 0) {
        set_err("graph_params['k']=%lld exceeds the supported maximum of 120", (long long)r.k);
        return AS_EUNSUPPORTED;
    }
    AS_HIP(hipSetDevice(sp->device));
    const double t0 = now_s();
    const as_status s = knn_block_pair(sp, cols, &r, row_begin, row_end, col_tile_begin, col_tile_end, row_goff, col_goff, col_thr_dev, row_thr_dev, M,
                                       p_key_dev, p_dist_dev, p_gy_dev, p_idx_dev, p_cnt_dev, p_t32_dev, q_key_dev, q_dist_dev, q_gy_dev,
                                       q_idx_dev, q_cnt_dev, q_t32_dev);
    sp->kstats[1] += now_s() - t0;
    return s;
}

as_status as_knn_thresholds(const as_space* sp, const as_graph_params* gp, int64_t row_begin, int64_t row_end, double nmax_all,
                            const double* r_key_dev, const int32_t* r_cnt_dev, float* out_thr_dev) {
    if (!sp || !r_key_dev || !r_cnt_dev || !out_thr_dev || row_begin < 0 || row_end > sp->n || row_begin > row_end) {
        set_err("as_knn_thresholds: null argument or bad row range");
        return AS_EINVAL;
    }
    as_graph_params r;
    AS_TRY(resolve_params(gp, &r));
    const int M = knn_list_width(r.k);
    if (M < 0) {
        set_err("graph_params['k']=%lld exceeds the supported maximum of 120", (long long)r.k);
        return AS_EUNSUPPORTED;
    }
    AS_HIP(hipSetDevice(sp->device));
    return knn_thresholds(sp, row_begin, row_end, M, nmax_all, r_key_dev, r_cnt_dev, out_thr_dev);
}

as_status as_knn_merge(const as_space* sp, const as_graph_params* gp, int64_t row_begin, int64_t row_end, int32_t nblocks,
                       const double* p_key_dev, const double* p_dist_dev, const double* p_gy_dev, const int32_t* p_idx_dev,
                       const int32_t* p_cnt_dev, const float* p_t32_dev, const double* block_nmax_host, int32_t* out_idx_dev,
                       double* out_key_dev, double* out_dist_dev, double* out_gy_dev, int32_t* out_cnt_dev, int32_t* out_flag_dev,
                       double* out_band_dev, int64_t* out_nflagged) {
    if (!sp || !p_key_dev || !p_dist_dev || !p_gy_dev || !p_idx_dev || !p_cnt_dev || !p_t32_dev || !out_idx_dev ||
        !out_key_dev || !out_dist_dev || !out_gy_dev || !out_cnt_dev || !out_flag_dev || !out_band_dev || !out_nflagged || nblocks < 0 ||
        (nblocks > 0 && !block_nmax_host)) {
        set_err("as_knn_merge: null argument");
        return AS_EINVAL;
    }
    as_graph_params r;
    AS_TRY(resolve_params(gp, &r));
    AS_HIP(hipSetDevice(sp->device));
    const double t0 = now_s();
    const as_status s = knn_merge(sp, &r, row_begin, row_end, nblocks, knn_list_width(r.k), p_key_dev, p_dist_dev, p_gy_dev, p_idx_dev,
                                  p_cnt_dev, p_t32_dev, block_nmax_host, out_idx_dev, out_key_dev, out_dist_dev, out_gy_dev, out_cnt_dev,
                                  out_flag_dev, out_band_dev, out_nflagged);
    sp->kstats[2] += now_s() - t0;
    return s;
}

as_status as_knn_fold(const as_space* sp, const as_graph_params* gp, int64_t row_begin, int64_t row_end, int32_t mode, double block_nmax,
                      const int32_t* flag_dev, double* r_key_dev, double* r_dist_dev, double* r_gy_dev, int32_t* r_idx_dev,
                      int32_t* r_cnt_dev, float* r_t32_dev, const double* b_key_dev, const double* b_dist_dev, const double* b_gy_dev,
                      const int32_t* b_idx_dev, const int32_t* b_cnt_dev, const float* b_t32_dev) {
    if (!sp || !r_key_dev || !r_dist_dev || !r_gy_dev || !r_idx_dev || !r_cnt_dev || !r_t32_dev || !b_key_dev || !b_dist_dev ||
        !b_gy_dev || !b_idx_dev || !b_cnt_dev || !b_t32_dev || (mode != 0 && !flag_dev) || mode < 0 || mode > 2) {
        set_err("as_knn_fold: null argument or bad mode");
        return AS_EINVAL;
    }
    as_graph_params r;
    AS_TRY(resolve_params(gp, &r));
    AS_HIP(hipSetDevice(sp->device));
    return knn_fold(sp, row_begin, row_end, knn_list_width(r.k), mode, block_nmax, flag_dev, r_key_dev, r_dist_dev, r_gy_dev, r_idx_dev,
                    r_cnt_dev, r_t32_dev, b_key_dev, b_dist_dev, b_gy_dev, b_idx_dev, b_cnt_dev, b_t32_dev);
}

as_status as_knn_block_band(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t row_begin, int64_t row_end,
                            int64_t row_goff, int64_t col_goff, int32_t* flag_dev, const double* band_dev, double* p_key_dev,
                            double* p_dist_dev, double* p_gy_dev, int32_t* p_idx_dev, int32_t* p_cnt_dev, float* p_t32_dev,
                            int64_t* out_overflowed) {
    if (!sp || !cols || !flag_dev || !band_dev || !p_key_dev || !p_dist_dev || !p_gy_dev || !p_idx_dev || !p_cnt_dev || !p_t32_dev ||
        !out_overflowed) {
        set_err("as_knn_block_band: null argument");
        return AS_EINVAL;
    }
    as_graph_params r;
    AS_TRY(resolve_params(gp, &r));
    AS_HIP(hipSetDevice(sp->device));
    const double t0 = now_s();
    const as_status s = knn_block_band(sp, cols, &r, row_begin, row_end, row_goff, col_goff, knn_list_width(r.k), flag_dev, band_dev,
                                       p_key_dev, p_dist_dev, p_gy_dev, p_idx_dev, p_cnt_dev, p_t32_dev, out_overflowed);
    sp->kstats[2] += now_s() - t0;
    if (s == AS_OK) sp->kstats[8] += (double)*out_overflowed;
    return s;
}

as_status as_knn_block_exact(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t row_begin, int64_t row_end,
                             int64_t row_goff, int64_t col_goff, const int32_t* flag_dev, double* p_key_dev, double* p_dist_dev,
                             double* p_gy_dev, int32_t* p_idx_dev, int32_t* p_cnt_dev, float* p_t32_dev) {
    if (!sp || !cols || !flag_dev || !p_key_dev || !p_dist_dev || !p_gy_dev || !p_idx_dev || !p_cnt_dev || !p_t32_dev) {
        set_err("as_knn_block_exact: null argument");
        return AS_EINVAL;
    }
    as_graph_params r;
    AS_TRY(resolve_params(gp, &r));
    AS_HIP(hipSetDevice(sp->device));
    const double t0 = now_s();
    const as_status s = knn_block_exact(sp, cols, &r, row_begin, row_end, row_goff, col_goff, knn_list_width(r.k), flag_dev, p_key_dev,
                                        p_dist_dev, p_gy_dev, p_idx_dev, p_cnt_dev, p_t32_dev);
    sp->kstats[3] += now_s() - t0;
    return s;
}

// graph + Laplacian + lambdas over a space whose items are in place (t0: the call's start, t1: the end of the ingest)
static as_status build_from_space(as_space* sp, const as_graph_params& r, double t0, double t1, as_space** out_space, as_graph** out_graph) {
    const int64_t n = sp->n, d = sp->d;
    int32_t *idx = nullptr, *cnt = nullptr;
    double *key = nullptr, *dist = nullptr, *gy = nullptr;
    as_graph* gr = new as_graph();
    as_status s = AS_OK;
    if (sp->opts.lambda_mode == AS_LAMBDA_FEATURE) {
        // lambda on the F x F feature-space Laplacian: no N x N item graph at all
        s = feat_build(sp, &r, gr);
        if (s != AS_OK) {
            as_free_graph(gr);
            as_free_space(sp);
            return s;
        }
        gr->stats[0] = t1 - t0;
        gr->stats[5] = now_s() - t0;
        dbg("built ArrowSpace: nitems=%lld, nfeatures=%lld, lambdas_len=%lld", (long long)n, (long long)d, (long long)n);
        *out_space = sp;
        *out_graph = gr;
        return AS_OK;
    }
    do {
        hipError_t e;
        if ((e = hipMalloc(&idx, sizeof(int32_t) * n * r.k)) != hipSuccess || (e = hipMalloc(&cnt, sizeof(int32_t) * n)) != hipSuccess ||
            (e = hipMalloc(&key, sizeof(double) * n * r.k)) != hipSuccess || (e = hipMalloc(&dist, sizeof(double) * n * r.k)) != hipSuccess ||
            (e = hipMalloc(&gy, sizeof(double) * n * r.k)) != hipSuccess) {
            set_err("hipMalloc of the k-NN lists failed: %s", hipGetErrorString(e));
            s = AS_ENOMEM;
            break;
        }
        s = knn_rows(sp, &r, 0, n, idx, key, dist, gy, cnt, gr->stats);
        if (s != AS_OK) break;
        const double t2 = now_s();
        s = graph_from_knn(sp, &r, idx, dist, gy, cnt, gr);
        if (s != AS_OK) break;
        const double t3 = now_s();
        gr->stats[0] = t1 - t0;
        gr->stats[4] = t3 - t2;
        gr->stats[5] = t3 - t0;
    } while (0);
    hipFree(idx); hipFree(cnt); hipFree(key); hipFree(dist); hipFree(gy);
    if (s != AS_OK) {
        as_free_graph(gr);
        as_free_space(sp);
        return s;
    }
    dbg("built ArrowSpace: nitems=%lld, nfeatures=%lld, lambdas_len=%lld", (long long)n, (long long)d, (long long)n);
    *out_space = sp;
    *out_graph = gr;
    return AS_OK;
}

as_status as_build_dev(const void* items_dev, int32_t dtype, int64_t n, int64_t d, int64_t ld, const as_graph_params* gp,
                       const as_opts* opts, as_space** out_space, as_graph** out_graph) {
    if (!out_space || !out_graph) {
        set_err("as_build: null output");
        return AS_EINVAL;
    }
    *out_space = nullptr;
    *out_graph = nullptr;
    as_graph_params r;
    AS_TRY(resolve_params(gp, &r));
    if (n > 0) AS_TRY(check_limits(&r, n, opts ? opts->lambda_mode : 0));
    const double t0 = now_s();
    as_space* sp = nullptr;
    AS_TRY(as_space_create_dev(items_dev, dtype, n, d, ld, opts, &sp));
    return build_from_space(sp, r, t0, now_s(), out_space, out_graph);
}

// The reference's own call: float64 items in HOST memory, any numpy strides (/root/reference/src/lib.rs:271-277,
// src/helpers.rs:24-46 -- `as_array()` accepts them).  The rows stream to the device through two pinned chunks, the ingest
// kernel converting one chunk under the copy of the next (ingest_host): no device copy of the whole fp64 matrix.
as_status as_build(const double* items, int64_t n, int64_t d, int64_t row_stride, int64_t col_stride, const as_graph_params* gp,
                   const as_opts* opts, as_space** out_space, as_graph** out_graph) {
    if (!out_space || !out_graph) {
        set_err("as_build: null output");
        return AS_EINVAL;
    }
    *out_space = nullptr;
    *out_graph = nullptr;
    if (!items || n <= 0 || d <= 0) {
        set_err("items must be non-empty 2D array");  // src/helpers.rs:27-29
        return AS_EINVAL;
    }
    as_graph_params r;
    AS_TRY(resolve_params(gp, &r));
    AS_TRY(check_limits(&r, n, opts ? opts->lambda_mode : 0));   // before the upload
    dbg("items shape: (%lld, %lld)", (long long)n, (long long)d);  // src/helpers.rs:31
    const double t0 = now_s();
    as_space* sp = nullptr;
    AS_TRY(space_new(n, d, opts, &sp));
    const as_status s = ingest_host(sp, items, row_stride, col_stride);
    if (s != AS_OK) {
        as_free_space(sp);
        return s;
    }
    return build_from_space(sp, r, t0, now_s(), out_space, out_graph);
}

// ---------------------------------------------------------------- search
// ---- the pool of single-query workspaces (as_space::qpool).  pool_acquire hands out the lowest free slot -- a single
// thread always gets slot 0 --, grows the pool by one workspace when every existing one is busy (ARROWSPACE_SEARCH_POOL
// caps it, 1 .. QPOOL, default QPOOL) and otherwise waits for a release.  A search against ANOTHER graph handle waits for
// the pool to fall idle and rebuilds it (the workspaces carry the graph's k / topk layout).
static int pool_limit() {
    static const int lim = [] {
        const char* e = getenv("ARROWSPACE_SEARCH_POOL");
        const int v = e ? atoi(e) : as_space::QPOOL;
        return std::max(1, std::min(v, (int)as_space::QPOOL));
    }();
    return lim;
}

static as_status pool_acquire(const as_space* sp, const as_graph* gr, int* slot, as_query** out) {
    std::unique_lock<std::mutex> lk(sp->qmu);
    for (;;) {
        bool any = false, busy = false;
        for (int i = 0; i < as_space::QPOOL; ++i) {
            // (a slot reserved while its workspace is being made -- outside the lock -- belongs to qcache_gr like a finished one: a
            // caller with another graph handle must wait for it, not reserve a slot of its own beside it)
            any = any || sp->qpool[i] != nullptr || sp->qbusy[i];
            busy = busy || sp->qbusy[i];
        }
        if (any && sp->qcache_gr != gr) {
            if (busy) {
                sp->qcv.wait(lk);
                continue;
            }
            for (int i = 0; i < as_space::QPOOL; ++i) {
                if (sp->qpool[i]) as_query_free(sp->qpool[i]);
                sp->qpool[i] = nullptr;
            }
            sp->qcache = nullptr;
        }
        for (int i = 0; i < as_space::QPOOL; ++i)
            if (sp->qpool[i] && !sp->qbusy[i]) {
                sp->qbusy[i] = true;
                *slot = i;
                *out = sp->qpool[i];
                return AS_OK;
            }
        for (int i = 0; i < pool_limit(); ++i)
            if (!sp->qpool[i] && !sp->qbusy[i]) {
                sp->qbusy[i] = true;   // reserved while the workspace is being made (allocations: outside the lock)
                sp->qcache_gr = gr;
                lk.unlock();
                as_query* q = nullptr;
                const as_status s = query_create(sp, gr, 1, &q, i);
                lk.lock();
                if (s != AS_OK) {
                    sp->qbusy[i] = false;
                    sp->qcv.notify_all();
                    return s;
                }
                sp->qpool[i] = q;
                if (i == 0) sp->qcache = q;
                *slot = i;
                *out = q;
                return AS_OK;
            }
        sp->qcv.wait(lk);
    }
}

static void pool_release(const as_space* sp, int slot) {
    {
        std::lock_guard<std::mutex> lk(sp->qmu);
        sp->qbusy[slot] = false;
    }
    sp->qcv.notify_all();
}

static as_status search_single(const as_space* sp, const as_graph* gr, as_query* q, const double* query, int64_t d, double tau,
                               int64_t* out_idx, double* out_score, int64_t* out_len, double* out_lambda_q);

// one single-query search on a workspace of the pool
static as_status search_pooled(const as_space* sp, const as_graph* gr, const double* query, int64_t d, double tau, int64_t* out_idx,
                               double* out_score, int64_t* out_len, double* out_lambda_q) {
    AS_HIP(hipSetDevice(sp->device));
    // (what gang_launch goes by: callers inside as_search on this space right now, and whether two or more were seen lately)
    const int act = sp->active_callers.fetch_add(1, std::memory_order_relaxed) + 1;
    // (gang_width: the most callers seen at once over the last 64 searches or so -- restarted from the current count every 64th)
    if ((sp->gang_tick.fetch_add(1, std::memory_order_relaxed) & 63) == 0 || act > sp->gang_width.load(std::memory_order_relaxed))
        sp->gang_width.store(act, std::memory_order_relaxed);
    if (act >= 2) sp->gang_hint.store(64, std::memory_order_relaxed);
    else if (sp->gang_hint.load(std::memory_order_relaxed) > 0) sp->gang_hint.fetch_sub(1, std::memory_order_relaxed);
    int slot = -1;
    as_query* q = nullptr;
    as_status s = pool_acquire(sp, gr, &slot, &q);
    if (s == AS_OK) {
        s = search_single(sp, gr, q, query, d, tau, out_idx, out_score, out_len, out_lambda_q);
        pool_release(sp, slot);
    }
    sp->active_callers.fetch_sub(1, std::memory_order_relaxed);
    return s;
}

// the graph handle belongs to this space: item graphs have one node per item, feature graphs one per column
static as_status graph_matches(const as_space* sp, const as_graph* gr, const char* who) {
    // a shard of a row-sharded index searches against the graph of all items
    const bool ok = gr->lambda_mode == AS_LAMBDA_FEATURE ? (gr->n == sp->d && gr->nitems >= sp->row_offset + sp->n)
                    : gr->ncols                          ? (gr->row0 == sp->row_offset && gr->n == sp->n && gr->ncols >= gr->row0 + gr->n)
                                                         : gr->n >= sp->row_offset + sp->n;
    if (!ok) {
        set_err("%s: the graph (%lld nodes) was not built for this space (%lld items x %lld features)", who, (long long)gr->n,
                (long long)sp->n, (long long)sp->d);
        return AS_EINVAL;
    }
    return AS_OK;
}

int32_t as_space_knn_pipe(const as_space* sp) { return sp ? sp->k2_last_pipe : -1; }
int32_t as_last_scan_int8(const as_space* sp) { return sp && sp->qcache ? as_query_scan_int8(sp->qcache) : 0; }
int64_t as_batch_dual_scans(const as_space* sp) { return sp ? (int64_t)sp->batch_dual_scans.load(std::memory_order_relaxed) : 0; }
int32_t as_last_batch_int8(const as_space* sp) { return sp && sp->qcache_b ? as_query_scan_int8(sp->qcache_b) : 0; }

as_status as_gang_counters(const as_space* sp, int64_t* out, int32_t n) {
    if (!sp || !out) {
        set_err("as_gang_counters: null argument");
        return AS_EINVAL;
    }
    std::lock_guard<std::mutex> lk(sp->gmu);
    for (int i = 0; i < n && i < 4; ++i) out[i] = sp->gang_scans[i + 1];
    for (int i = 4; i < n && i < 10; ++i) out[i] = sp->gang_skip[i - 4].load(std::memory_order_relaxed);
    return AS_OK;
}

int32_t as_search_pool_size(const as_space* sp) {
    if (!sp) return 0;
    std::lock_guard<std::mutex> lk(sp->qmu);
    int n = 0;
    for (int i = 0; i < as_space::QPOOL; ++i) n += sp->qpool[i] ? 1 : 0;
    return n;
}

as_status as_search(const as_space* sp, const as_graph* gr, const double* query, int64_t d, double tau, int64_t* out_idx,
                    double* out_score, int64_t* out_len, double* out_lambda_q) {
    if (!sp || !gr || !query || !out_idx || !out_score) {
        set_err("as_search: null argument");
        return AS_EINVAL;
    }
    if (d != sp->d) {  // src/lib.rs:140-146
        set_err("query length %lld must match nfeatures %lld", (long long)d, (long long)sp->d);
        return AS_EINVAL;
    }
    AS_TRY(graph_matches(sp, gr, "as_search"));
    return search_pooled(sp, gr, query, d, tau, out_idx, out_score, out_len, out_lambda_q);
}

static as_status search_single(const as_space* sp, const as_graph* gr, as_query* q, const double* query, int64_t d, double tau,
                               int64_t* out_idx, double* out_score, int64_t* out_len, double* out_lambda_q) {
    (void)gr;
    // mode bit0: fp64 end to end, bit1: wavefront-list selection (candidate buffer overflowed)
    int mode = sp->opts.search_mode & 3;  // tests start directly on a fallback path
    as_status s = AS_OK;
    int64_t cnt[6] = {1, 0, 0, 0, 0, 0};
    for (int attempt = 0; attempt < 3; ++attempt) {
        s = search_once(q, query, d, tau, mode, out_idx, out_score, out_len, out_lambda_q);
        if (s != AS_OK && s != AS_EZEROLAMBDA) break;
        // lambda_q == 0 (no item within eps, proven: the k-NN check below still escalates an unproven empty list) is the
        // reference's panic (src/lib.rs:156-159): no result will be returned, so the scorer's flags of such a query --
        // every item ties at tau = 0 -- must not buy it two more passes over the items
        if (s == AS_EZEROLAMBDA) {
            int ki0 = 0, si0 = 0;
            query_flags(q, &ki0, &si0);
            if (!(ki0 & 1) && !(query_overflow_bits(q) & 1)) break;
        }
        int ki = 0, si = 0;
        query_flags(q, &ki, &si);
        int next = mode;
        if (ki & 2) next |= 2;
        if (((ki & 1) || si) && !sp->opts.force_exact) next |= 1;
        if (next == mode) break;
        dbg("search: fast path not provably exact (knn=%d score=%d), rerunning with mode %d", ki, si, next);
        cnt[2] += (ki & 1) ? 1 : 0;
        cnt[3] += (ki & 2) ? 1 : 0;
        cnt[4] += si ? 1 : 0;
        cnt[5] = 1;
        mode = next;
    }
    cnt[1] = s == AS_EZEROLAMBDA ? 1 : 0;
    {
        std::lock_guard<std::mutex> lk(sp->qmu);
        for (int i = 0; i < 6; ++i) sp->scount[i] += cnt[i];
    }
    if (s == AS_OK && out_lambda_q) dbg("search: qlen=%lld, lambda_q=%.6f", (long long)d, *out_lambda_q);  // src/lib.rs:161-165
    {   // the strongest path has run and the answer still fails its a-posteriori check (more near-ties than fp64
        // can order): it is returned, but never silently -- counted, and reported on stderr once per space
        int ki = 0, si = 0;
        query_flags(q, &ki, &si);
        if (s == AS_OK && ((ki & 1) || (si & 1))) {
            std::lock_guard<std::mutex> lk(sp->qmu);
            if (sp->unproven_searches++ == 0)
                fprintf(stderr, "[pyarrowspace] warning: a search result could not be proven exact (ties at the k-th distance or score "
                                "inside fp64 rounding); see as_unproven_searches()\n");
        }
    }
    return s;
}

as_status as_search_batch(const as_space* sp, const as_graph* gr, const double* queries, int64_t b, int64_t d, double tau,
                          int64_t* out_idx, double* out_score, int64_t* out_len, double* out_lambda_q, int32_t* out_status) {
    if (!sp || !gr || !queries || !out_idx || !out_score || !out_len) {
        set_err("as_search_batch: null argument");
        return AS_EINVAL;
    }
    if (d != sp->d) {
        set_err("query length %lld must match nfeatures %lld", (long long)d, (long long)sp->d);
        return AS_EINVAL;
    }
    AS_TRY(graph_matches(sp, gr, "as_search_batch"));
    const int64_t topk = std::min<int64_t>(gr->gp.topk, sp->n);
    std::lock_guard<std::mutex> lock(sp->bmu);   // (the batched workspaces; single-query fallbacks below go through the pool)
    AS_HIP(hipSetDevice(sp->device));
    // the batched pass serves QUERY_BATCH queries per read of the items (MFMA pass; rows wider than 768 floats in
    // K-chunk passes of the same kernel): fp32 fast path only
    const bool batched = !sp->opts.force_exact && (sp->opts.search_mode & 3) == 0 && b > 1;
    int unit = 1;   // passes launched together
    if (batched) {
        if (sp->qcache_b && sp->qcache_b_gr != gr) {
            for (as_query** w : {&sp->qcache_b, &sp->qcache_b2, &sp->qcache_b3, &sp->qcache_b4}) {
                if (*w) as_query_free(*w);
                *w = nullptr;
            }
        }
        if (!sp->qcache_b) {
            AS_TRY(query_create(sp, gr, QUERY_BATCH, &sp->qcache_b));
            sp->qcache_b_gr = gr;
        }
        // More than one pass: a second workspace.  Rows of up to 768 columns: the two launch their passes as a PAIR, one scan for
        // both (64 queries per read of the items: search_batch_launch_pair); more than one pair: a second pair of workspaces, pair
        // p + 1 is queued before pair p is waited for.  Wider rows (a scan serves one workspace): the passes alternate between
        // the two workspaces -- pass p + 1 is queued before pass p is waited for.  (No memory: fewer workspaces, the same results.)
        static const bool no_pipe = getenv("ARROWSPACE_NO_BATCH_PIPELINE") != nullptr;
        static const bool no_dual = getenv("ARROWSPACE_NO_BATCH_DUAL") != nullptr;
        const bool pairs = !no_dual && (sp->dp + 63) / 64 * 64 / 2 <= 4 * 3 * 32 && gr->lambda_mode != AS_LAMBDA_FEATURE;
        if (b > QUERY_BATCH && !sp->qcache_b2 && !no_pipe) {
            if (query_create(sp, gr, QUERY_BATCH, &sp->qcache_b2) != AS_OK) sp->qcache_b2 = nullptr;
        }
        if (pairs && b > 2 * QUERY_BATCH && sp->qcache_b2 && !sp->qcache_b4 && !no_pipe) {
            if (!sp->qcache_b3 && query_create(sp, gr, QUERY_BATCH, &sp->qcache_b3) != AS_OK) sp->qcache_b3 = nullptr;
            if (sp->qcache_b3 && query_create(sp, gr, QUERY_BATCH, &sp->qcache_b4) != AS_OK) sp->qcache_b4 = nullptr;
        }
        unit = pairs && sp->qcache_b2 && b > QUERY_BATCH ? 2 : 1;
    }
    // a UNIT = the passes launched together: a pair (two workspaces), or one pass; two sets of workspaces alternate when there is
    // more than one unit and the workspaces exist (`piped`)
    as_query* ws[4] = {sp->qcache_b, sp->qcache_b2, sp->qcache_b3, sp->qcache_b4};
    const bool piped = batched && b > unit * QUERY_BATCH && ws[2 * unit - 1] != nullptr;
    const int nws = batched ? unit * (piped ? 2 : 1) : 0;
    // an error leaves no pass in flight behind it (the other workspaces' kernels would otherwise still be running when the
    // caller comes back)
    auto drain = [&](as_status s) {
        if (nws > 1)
            for (int w = 0; w < nws; ++w) (void)hipStreamSynchronize((hipStream_t)as_query_stream(ws[w]));
        return s;
    };
    int32_t st_chunk[QUERY_BATCH];
    const int64_t UNIT = (int64_t)unit * QUERY_BATCH;
    auto launch_unit = [&](int64_t j) -> as_status {
        as_query* const* w = ws + (piped ? unit * (j & 1) : 0);
        const int64_t i0 = j * UNIT, i1 = i0 + QUERY_BATCH;
        const int nb0 = (int)std::min<int64_t>(QUERY_BATCH, b - i0);
        if (unit == 2 && i1 < b) return search_batch_launch_pair(w[0], w[1], queries + i0 * d, nb0, queries + i1 * d, (int)std::min<int64_t>(QUERY_BATCH, b - i1), d, tau);
        return search_batch_launch(w[0], queries + i0 * d, nb0, d, tau);
    };
    if (piped) {
        const as_status s0 = launch_unit(0);
        if (s0 != AS_OK) return drain(s0);
    }
    int64_t pass = 0;
    static const bool timing = getenv("ARROWSPACE_DEBUG") != nullptr;   // host time of the two halves of a pass, per call
    double t_launch = 0.0, t_collect = 0.0;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int64_t i0 = 0; i0 < b; i0 += QUERY_BATCH, ++pass) {
        const int nb = (int)std::min<int64_t>(QUERY_BATCH, b - i0);
        if (batched) {
            const int64_t j = pass / unit;
            as_query* cur = ws[(piped ? unit * (j & 1) : 0) + (int)(pass % unit)];
            as_status s = AS_OK;
            const double t0 = timing ? now() : 0.0;
            if (pass % unit == 0) {   // a unit's first pass: queue the next unit (piped), or this one
                if (!piped) s = launch_unit(j);
                else if ((j + 1) * UNIT < b) s = launch_unit(j + 1);
            }
            const double t1 = timing ? now() : 0.0;
            if (s == AS_OK)
                s = search_batch_collect(cur, nb, tau, topk, out_idx + i0 * topk, out_score + i0 * topk, out_len + i0,
                                         out_lambda_q ? out_lambda_q + i0 : nullptr, st_chunk);
            if (timing) {
                t_launch += t1 - t0;
                t_collect += now() - t1;
            }
            if (s != AS_OK) return drain(s);
        } else {
            for (int t = 0; t < nb; ++t) st_chunk[t] = -1;
        }
        for (int t = 0; t < nb; ++t) {
            const int64_t i = i0 + t;
            if (st_chunk[t] == -1) {  // not provably exact on the batched fast path (or batching unavailable)
                double lq = 0.0;
                const as_status s1 = search_pooled(sp, gr, queries + i * d, d, tau, out_idx + i * topk, out_score + i * topk, out_len + i, &lq);
                if (out_lambda_q) out_lambda_q[i] = lq;
                st_chunk[t] = (int32_t)s1;
                if (s1 != AS_OK && s1 != AS_EZEROLAMBDA) return drain(s1);
            }
            if (st_chunk[t] == AS_EZEROLAMBDA) out_len[i] = 0;
            if (out_status) out_status[i] = st_chunk[t];
        }
    }
    if (timing && batched) dbg("as_search_batch: %lld passes, host time per pass: launch half %.0f us, collect half (with its wait) %.0f us",
                               (long long)pass, t_launch / std::max<int64_t>(pass, 1), t_collect / std::max<int64_t>(pass, 1));
    return AS_OK;
}

// ---------------------------------------------------------------- accessors
int64_t as_nitems(const as_space* sp) { return sp ? sp->n : 0; }
int64_t as_nfeatures(const as_space* sp) { return sp ? sp->d : 0; }
int64_t as_nnodes(const as_graph* gr) { return gr ? gr->n : 0; }
int64_t as_graph_nnz(const as_graph* gr) { return gr ? gr->nnz + gr->n : 0; }
double as_graph_tau0(const as_graph* gr) { return gr ? gr->tau0 : 0.0; }
int32_t as_graph_lambda_mode(const as_graph* gr) { return gr ? gr->lambda_mode : 0; }
const double* as_lambdas_dev(const as_space* sp) { return sp ? sp->lam64 : nullptr; }

as_status as_get_item(const as_space* sp, int64_t idx, double* out_vec, double* out_lambda) {
    if (!sp || !out_vec) {
        set_err("as_get_item: null argument");
        return AS_EINVAL;
    }
    if (idx < 0 || idx >= sp->n) {  // src/lib.rs:101-107
        set_err("index %lld out of range [0, %lld)", (long long)idx, (long long)sp->n);
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(sp->device));
    if (sp->x64) {
        AS_HIP(hipMemcpy(out_vec, sp->x64 + idx * sp->d, sizeof(double) * sp->d, hipMemcpyDeviceToHost));
    } else {
        std::vector<float> tmp(sp->d);
        AS_HIP(hipMemcpy(tmp.data(), sp->x32 + idx * sp->dp, sizeof(float) * sp->d, hipMemcpyDeviceToHost));
        for (int64_t c = 0; c < sp->d; ++c) out_vec[c] = (double)tmp[c];
    }
    if (out_lambda) AS_HIP(hipMemcpy(out_lambda, sp->lam64 + idx, sizeof(double), hipMemcpyDeviceToHost));
    return AS_OK;
}

as_status as_lambdas(const as_space* sp, double* out) {
    if (!sp || !out) {
        set_err("as_lambdas: null argument");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(sp->device));
    AS_HIP(hipMemcpy(out, sp->lam64, sizeof(double) * sp->n, hipMemcpyDeviceToHost));
    return AS_OK;
}

as_status as_get_graph_params(const as_graph* gr, as_graph_params* out) {
    if (!gr || !out) {
        set_err("as_get_graph_params: null argument");
        return AS_EINVAL;
    }
    *out = gr->gp;
    return AS_OK;
}

as_status as_graph_degrees(const as_graph* gr, double* out) {
    if (!gr || !out) {
        set_err("as_graph_degrees: null argument");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(gr->device));
    AS_HIP(hipMemcpy(out, gr->deg, sizeof(double) * gr->n, hipMemcpyDeviceToHost));
    return AS_OK;
}

as_status as_graph_csr(const as_graph* gr, int64_t* indptr, int64_t* indices, double* values) {
    if (!gr || !indptr || !indices || !values) {
        set_err("as_graph_csr: null argument");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(gr->device));
    const int64_t n = gr->n, nnz = gr->nnz;
    std::vector<int64_t> ip(n + 1);
    std::vector<int32_t> col(std::max<int64_t>(nnz, 1));
    std::vector<double> lap(std::max<int64_t>(nnz, 1)), deg(n);
    AS_HIP(hipMemcpy(ip.data(), gr->indptr, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost));
    if (nnz) {
        AS_HIP(hipMemcpy(col.data(), gr->indices, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost));
        AS_HIP(hipMemcpy(lap.data(), gr->lap, sizeof(double) * nnz, hipMemcpyDeviceToHost));
    }
    AS_HIP(hipMemcpy(deg.data(), gr->deg, sizeof(double) * n, hipMemcpyDeviceToHost));
    // insert the diagonal keeping columns ascending: normalised Laplacian 1 for connected nodes, 0 for isolated
    // ones; feature graph (L = D - W) the degree
    const bool comb = gr->lambda_mode == AS_LAMBDA_FEATURE;
    int64_t w = 0;
    for (int64_t r = 0; r < n; ++r) {
        const int64_t i = r + gr->row0;   // a sharded graph holds the rows [row0, row0 + n): the diagonal sits at column row0 + r
        indptr[r] = w;
        bool placed = false;
        for (int64_t e = ip[r]; e < ip[r + 1]; ++e) {
            if (!placed && col[e] > i) {
                indices[w] = i;
                values[w++] = comb ? deg[r] : (deg[r] > 0.0 ? 1.0 : 0.0);
                placed = true;
            }
            indices[w] = col[e];
            values[w++] = lap[e];
        }
        if (!placed) {
            indices[w] = i;
            values[w++] = comb ? deg[r] : (deg[r] > 0.0 ? 1.0 : 0.0);
        }
    }
    indptr[n] = w;
    return AS_OK;
}

as_status as_last_search_stats(const as_space* sp, double* out, int32_t n) {
    if (!sp || !out || !sp->qcache) {
        set_err("as_last_search_stats: no search has run on this space");
        return AS_EINVAL;
    }
    return as_query_stats(sp->qcache, out, n);
}

as_status as_search_counters(const as_space* sp, int64_t* out, int32_t n) {
    if (!sp || !out) {
        set_err("as_search_counters: null argument");
        return AS_EINVAL;
    }
    std::lock_guard<std::mutex> lk(sp->qmu);
    for (int i = 0; i < n && i < 6; ++i) out[i] = sp->scount[i];
    return AS_OK;
}

as_status as_build_stats(const as_graph* gr, double* out, int32_t n) {
    if (!gr || !out) {
        set_err("as_build_stats: null argument");
        return AS_EINVAL;
    }
    for (int i = 0; i < n && i < 10; ++i) out[i] = gr->stats[i];
    return AS_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- index persistence
// One flat little-endian file: header, the items (fp64 when an fp64 copy is kept, else the exact
// fp32 values), lambdas, and the graph arrays.  Loading re-ingests the items (norms and the padded
// fp32 layout are recomputed deterministically) and uploads the rest; no k-NN work is redone.
// The header carries its own size and a format version; the loader checks the file length against the
// header-derived sizes before allocating anything and validates the CSR it is about to trust.
namespace {
struct IndexHeader {
    char magic[8];           // "ASIDX04\0"
    int32_t header_bytes;    // sizeof(IndexHeader) of the writer
    int32_t version;         // 4
    int64_t n, d, nnz;       // items held by this file, features, adjacency entries
    int64_t nnodes;          // graph rows in this file: the items of the whole index (item mode, replicated graph), this
                             // shard's n rows (item mode, sharded graph: graph_ncols > 0) or d (feature mode)
    int64_t graph_ncols;     // sharded item graph: the items its columns range over; 0 for a whole graph.  Feature mode: the
                             // items the lambdas were ranked over (all ranks' rows)
    int64_t graph_row0;      // sharded item graph: its first row (== row_offset)
    int64_t row_offset;      // global index of this file's first item (0 for a whole index)
    int32_t has_f64, metric, kernel, lambda_mode;
    as_graph_params gp;
    double tau0;
};
constexpr int32_t INDEX_VERSION = 4;

template <typename T>
as_status dev_to_file(FILE* f, const T* dev, size_t count) {
    std::vector<T> h(std::max<size_t>(count, 1));
    if (count) AS_HIP(hipMemcpy(h.data(), dev, sizeof(T) * count, hipMemcpyDeviceToHost));
    if (count && fwrite(h.data(), sizeof(T), count, f) != count) {
        set_err("as_index_save: short write");
        return AS_EINVAL;
    }
    return AS_OK;
}
template <typename T>
as_status file_to_host(FILE* f, std::vector<T>& h, size_t count) {
    h.resize(std::max<size_t>(count, 1));
    if (count && fread(h.data(), sizeof(T), count, f) != count) {
        set_err("as_index_load: truncated file");
        return AS_EINVAL;
    }
    return AS_OK;
}
template <typename T>
as_status host_to_dev(const std::vector<T>& h, T** dev, size_t count) {
    AS_HIP(hipMalloc(dev, sizeof(T) * std::max<size_t>(count, 1)));
    if (count) AS_HIP(hipMemcpy(*dev, h.data(), sizeof(T) * count, hipMemcpyHostToDevice));
    return AS_OK;
}
template <typename T>
as_status file_to_dev(FILE* f, T** dev, size_t count) {
    std::vector<T> h;
    AS_TRY(file_to_host(f, h, count));
    return host_to_dev(h, dev, count);
}
// bytes the arrays behind the header occupy, or -1 when the header's sizes are not credible
int64_t payload_bytes(const IndexHeader& h) {
    const int64_t lim = (int64_t)1 << 46;
    if (h.n <= 0 || h.d <= 0 || h.nnz < 0 || h.nnodes <= 0 || h.n >= ((int64_t)1 << 31) || h.d >= ((int64_t)1 << 31) ||
        h.nnz > lim || h.n > lim / h.d)
        return -1;
    const int64_t item = h.has_f64 ? 8 : 4;
    int64_t b = h.n * h.d * item + h.n * 8;                 // items, lambdas
    b += (h.nnodes + 1) * 8 + h.nnz * 4 + h.nnz * 8 * 4;    // indptr, indices, dist gy w lap
    b += h.nnodes * 8;                                      // deg
    if (h.lambda_mode == AS_LAMBDA_FEATURE) b += h.n * 8 * 2 + h.d * 8;   // E, G, colm
    else b += h.nnodes * 8 * 3;                             // ny, E, G of every graph node
    return b;
}
}  // namespace

extern "C" {

as_status as_index_save(const as_space* sp, const as_graph* gr, const char* path) {
    if (!sp || !gr || !path) {
        set_err("as_index_save: null argument");
        return AS_EINVAL;
    }
    AS_HIP(hipSetDevice(sp->device));
    FILE* f = fopen(path, "wb");
    if (!f) {
        set_err("as_index_save: cannot open %s", path);
        return AS_EINVAL;
    }
    IndexHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "ASIDX04", 8);
    h.header_bytes = (int32_t)sizeof(IndexHeader);
    h.version = INDEX_VERSION;
    h.n = sp->n; h.d = sp->d; h.nnz = gr->nnz; h.nnodes = gr->n; h.row_offset = sp->row_offset;
    h.graph_ncols = gr->lambda_mode == AS_LAMBDA_FEATURE ? gr->nitems : gr->ncols;   // feature mode: the items the lambdas were ranked over
    h.graph_row0 = gr->row0;
    h.has_f64 = sp->x64 ? 1 : 0; h.metric = gr->metric; h.kernel = gr->kernel; h.lambda_mode = gr->lambda_mode;
    h.gp = gr->gp; h.tau0 = gr->tau0;
    as_status s = fwrite(&h, sizeof(h), 1, f) == 1 ? AS_OK : AS_EINVAL;
    const size_t n = (size_t)sp->n, d = (size_t)sp->d, nnz = (size_t)gr->nnz, nn = (size_t)gr->n;
    if (s == AS_OK) {
        if (sp->x64) {
            s = dev_to_file(f, sp->x64, n * d);
        } else {
            std::vector<float> all(n * d);
            hipError_t e = hipMemcpy2D(all.data(), sizeof(float) * d, sp->x32, sizeof(float) * sp->dp, sizeof(float) * d, n, hipMemcpyDeviceToHost);
            if (e != hipSuccess) s = AS_EHIP;
            if (s == AS_OK && fwrite(all.data(), sizeof(float), n * d, f) != n * d) s = AS_EINVAL;
        }
    }
    if (s == AS_OK) s = dev_to_file(f, sp->lam64, n);
    if (s == AS_OK) s = dev_to_file(f, gr->indptr, nn + 1);
    if (s == AS_OK) s = dev_to_file(f, gr->indices, nnz);
    if (s == AS_OK) s = dev_to_file(f, gr->dist, nnz);
    if (s == AS_OK) s = dev_to_file(f, gr->gy, nnz);
    if (s == AS_OK) s = dev_to_file(f, gr->w, nnz);
    if (s == AS_OK) s = dev_to_file(f, gr->lap, nnz);
    if (s == AS_OK) s = dev_to_file(f, gr->deg, nn);
    if (gr->lambda_mode == AS_LAMBDA_FEATURE) {
        // a shard of a row-sharded index holds the energies of every item right after its build (e_rows == nitems) and
        // its own rows' after a load (e_rows == n): this file keeps its own rows' either way
        const int64_t eoff = gr->e_rows > sp->n ? sp->row_offset : 0;
        if (s == AS_OK && gr->e_rows < eoff + sp->n) {
            set_err("as_index_save: the graph holds %lld energies, the space rows [%lld, %lld)", (long long)gr->e_rows, (long long)eoff, (long long)(eoff + sp->n));
            s = AS_EINVAL;
        }
        if (s == AS_OK) s = dev_to_file(f, gr->E + eoff, n);
        if (s == AS_OK) s = dev_to_file(f, gr->G + eoff, n);
        if (s == AS_OK) s = dev_to_file(f, gr->colm, d);
    } else {
        if (s == AS_OK) s = dev_to_file(f, gr->ny, nn);
        if (s == AS_OK) s = dev_to_file(f, gr->E, nn);
        if (s == AS_OK) s = dev_to_file(f, gr->G, nn);
    }
    if (fclose(f) != 0 && s == AS_OK) s = AS_EINVAL;
    if (s != AS_OK && err_slot().empty()) set_err("as_index_save: write to %s failed", path);
    return s;
}

__global__ void lam32_kernel(int64_t n, const double* __restrict__ lam64, float* __restrict__ lam32) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lam32[i] = (float)lam64[i];
}

static as_status index_load_impl(FILE* f, const char* path, const as_opts* opts, as_space** out_space, as_graph** out_graph) {
    IndexHeader h;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "ASIDX04", 8) != 0 || h.header_bytes != (int32_t)sizeof(IndexHeader) ||
        h.version != INDEX_VERSION) {
        set_err("as_index_load: %s is not an arrowspace index file of format %d", path, INDEX_VERSION);
        return AS_EINVAL;
    }
    const int64_t need = payload_bytes(h);
    const bool modes_ok = (h.metric == AS_METRIC_L2 || h.metric == AS_METRIC_COSINE) &&
                          (h.kernel == AS_KERNEL_GAUSSIAN || h.kernel == AS_KERNEL_RATIONAL) &&
                          (h.lambda_mode == AS_LAMBDA_ITEM || h.lambda_mode == AS_LAMBDA_FEATURE) &&
                          (h.lambda_mode == AS_LAMBDA_FEATURE ? (h.nnodes == h.d && h.row_offset >= 0)
                          : h.graph_ncols ? (h.row_offset >= 0 && h.graph_row0 == h.row_offset && h.nnodes == h.n &&
                                             h.graph_ncols >= h.row_offset + h.n && h.graph_ncols < ((int64_t)1 << 31))
                                          : (h.row_offset >= 0 && h.graph_row0 == 0 && h.nnodes >= h.row_offset + h.n));
    as_graph_params gpr;
    if (need < 0 || !modes_ok || resolve_params(&h.gp, &gpr) != AS_OK || !(h.tau0 >= 0.0) || !(h.tau0 <= 1.0)) {
        set_err("as_index_load: %s has an inconsistent header", path);
        return AS_EINVAL;
    }
    {   // the file must hold exactly the arrays the header announces
        const long pos = ftell(f);
        if (pos < 0 || fseek(f, 0, SEEK_END) != 0) {
            set_err("as_index_load: cannot seek in %s", path);
            return AS_EINVAL;
        }
        const long end = ftell(f);
        if (fseek(f, pos, SEEK_SET) != 0 || (int64_t)(end - pos) != need) {
            set_err("as_index_load: %s is truncated or has trailing bytes (%lld payload bytes, header implies %lld)", path,
                    (long long)(end - pos), (long long)need);
            return AS_EINVAL;
        }
    }
    as_opts o{};
    if (opts) o = *opts;
    o.metric = h.metric;
    o.kernel = h.kernel;
    o.lambda_mode = h.lambda_mode;
    o.keep_f64 = h.has_f64 ? AS_KEEP_F64_ALWAYS : AS_KEEP_F64_AUTO;
    int dev = 0;
    AS_TRY(pick_device(&o, &dev));
    o.device = dev;
    as_space* sp = nullptr;
    as_graph* gr = nullptr;
    void* items = nullptr;
    const size_t n = (size_t)h.n, d = (size_t)h.d, nnz = (size_t)h.nnz, nn = (size_t)h.nnodes;
    as_status s = AS_OK;
    do {
        if (h.has_f64) {
            double* p = nullptr;
            s = file_to_dev(f, &p, n * d);
            items = p;
        } else {
            float* p = nullptr;
            s = file_to_dev(f, &p, n * d);
            items = p;
        }
        if (s != AS_OK) break;
        s = as_space_create_dev(items, h.has_f64 ? AS_DTYPE_F64 : AS_DTYPE_F32, h.n, h.d, h.d, &o, &sp);
        if (s != AS_OK) break;
        hipFree(sp->lam64);
        sp->lam64 = nullptr;
        s = file_to_dev(f, &sp->lam64, n);
        if (s != AS_OK) break;
        hipLaunchKernelGGL(lam32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sp->stream, h.n, sp->lam64, sp->lam32);
        gr = new as_graph();
        gr->device = dev; gr->n = h.nnodes; gr->nitems = h.row_offset + h.n; gr->nnz = h.nnz; gr->gp = gpr; gr->metric = h.metric; gr->kernel = h.kernel;
        gr->lambda_mode = h.lambda_mode; gr->tau0 = h.tau0; gr->row0 = h.graph_row0;
        if (h.lambda_mode == AS_LAMBDA_FEATURE) gr->nitems = std::max<int64_t>(gr->nitems, h.graph_ncols);
        else gr->ncols = h.graph_ncols;
        {   // the CSR is walked on trust afterwards: monotone row pointers ending at nnz, columns inside the graph
            std::vector<int64_t> ip;
            std::vector<int32_t> col;
            if ((s = file_to_host(f, ip, nn + 1)) != AS_OK) break;
            if ((s = file_to_host(f, col, nnz)) != AS_OK) break;
            bool ok = ip[0] == 0 && ip[nn] == h.nnz;
            for (size_t i = 0; ok && i < nn; ++i) ok = ip[i] <= ip[i + 1];
            const int64_t colmax = h.lambda_mode != AS_LAMBDA_FEATURE && h.graph_ncols ? h.graph_ncols : h.nnodes;   // (feature mode: the field counts items)
            for (size_t e = 0; ok && e < nnz; ++e) ok = col[e] >= 0 && (int64_t)col[e] < colmax;
            if (!ok) {
                set_err("as_index_load: %s holds an inconsistent graph (row pointers or column indices out of range)", path);
                s = AS_EINVAL;
                break;
            }
            if ((s = host_to_dev(ip, &gr->indptr, nn + 1)) != AS_OK) break;
            if ((s = host_to_dev(col, &gr->indices, nnz)) != AS_OK) break;
        }
        if ((s = file_to_dev(f, &gr->dist, nnz)) != AS_OK) break;
        if ((s = file_to_dev(f, &gr->gy, nnz)) != AS_OK) break;
        if ((s = file_to_dev(f, &gr->w, nnz)) != AS_OK) break;
        if ((s = file_to_dev(f, &gr->lap, nnz)) != AS_OK) break;
        if ((s = file_to_dev(f, &gr->deg, nn)) != AS_OK) break;
        if (h.lambda_mode == AS_LAMBDA_FEATURE) {
            if ((s = file_to_dev(f, &gr->E, n)) != AS_OK) break;
            if ((s = file_to_dev(f, &gr->G, n)) != AS_OK) break;
            gr->e_rows = h.n;
            if ((s = file_to_dev(f, &gr->colm, d)) != AS_OK) break;
            if ((s = feat_edges_from_csr(gr, sp->stream)) != AS_OK) break;
            sp->row_offset = h.row_offset;
        } else {
            if ((s = file_to_dev(f, &gr->ny, nn)) != AS_OK) break;
            if ((s = file_to_dev(f, &gr->E, nn)) != AS_OK) break;
            if ((s = file_to_dev(f, &gr->G, nn)) != AS_OK) break;
            sp->row_offset = h.row_offset;
        }
        if (hipStreamSynchronize(sp->stream) != hipSuccess) s = AS_EHIP;
    } while (0);
    if (items) hipFree(items);
    if (s != AS_OK) {
        if (gr) as_free_graph(gr);
        if (sp) as_free_space(sp);
        return s;
    }
    *out_space = sp;
    *out_graph = gr;
    return AS_OK;
}

as_status as_index_load(const char* path, const as_opts* opts, as_space** out_space, as_graph** out_graph) {
    if (!path || !out_space || !out_graph) {
        set_err("as_index_load: null argument");
        return AS_EINVAL;
    }
    *out_space = nullptr;
    *out_graph = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) {
        set_err("as_index_load: cannot open %s", path);
        return AS_EINVAL;
    }
    as_status s;
    try {
        s = index_load_impl(f, path, opts, out_space, out_graph);
    } catch (const std::bad_alloc&) {   // nothing may unwind through the C ABI
        set_err("as_index_load: out of host memory reading %s", path);
        s = AS_ENOMEM;
    }
    fclose(f);
    return s;
}

}  // extern "C"
