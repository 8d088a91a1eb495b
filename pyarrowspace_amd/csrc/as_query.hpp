// Internal declarations shared by the search translation units (as_scan.hip: the scan kernels of K7a;
// as_search.hip: selection, finish kernels and the host side).  Not part of the C ABI.
#pragma once
#include <algorithm>
#include <atomic>
#include <mutex>

#include "as_common.hpp"

namespace as {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CAND_CAP = 4096;  // candidate buffer of the filter path
constexpr int REC_CAP = 1024;   // k-NN records q_lambda accepts (ranks x k: k = 120 on 8 ranks)
constexpr int MAX_TOPK = 1024;  // largest topk
constexpr int MS_MAX = MAX_TOPK + 64;  // widest scorer candidate list (topk + margin, rounded to 64)
constexpr int HIT_CAP = 8 * (MAX_TOPK + 1) + 8;  // hit records hits_final accepts (ranks x (topk + 1))
constexpr int QB = 8;           // queries per VALU batched scan launch (query fragments live in registers)
constexpr int SC_WCAP = 64;      // fused tail: words of a scan wave's report (the count + up to 63 candidate rows)
constexpr int GQ = 32;          // queries per MFMA (GEMM-shaped) batched scan pass == slots of the batched workspace

// Batched searches run GQ independent query "slots" side by side: every per-query buffer is
// an array over slots and blockIdx.z selects the slot (z = 0 for single-query searches).
struct SlotStride {
    int64_t dots;   // elements between consecutive slots' dots inside one 32-row tile (32: tile-major, batched workspace)
    int64_t dots_ts; // elements between consecutive 32-row tiles (32 = plain row order, single slot; 32 * slots batched)
    int dots_rs;     // elements between consecutive rows of one slot inside a tile: 1, or 4 in the batched workspace, whose
                     // tiles are [slot quad][32 rows][4 slots] -- a lane of the MFMA scan holds 4 slots of one row and
                     // stores them as one dwordx4, the selection kernels read a row's 8 slots as two dwordx4
    int64_t q;      // dp
    int64_t qin;    // d
    int64_t knn;    // k records
    int64_t hits;   // topk + 1 records
};

// element offset of slot s inside a 32-row tile of the dots (see SlotStride)
__host__ __device__ inline int64_t dots_slot_off(int64_t s, int64_t sd, int rs) { return rs == 4 ? (s >> 2) * 128 + (s & 3) : s * sd; }

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
    int overflow;     // bit0: the k-NN candidate buffer overflowed (-> threshold repair over the kept dots), bit1: the scorer's (-> list path),
                      // bit2: the scan's own scorer candidates overflowed (fused tail -> the threshold chain over the kept dots)
};

// Fused tail (scan-side scorer candidates): counts of rows by cosine bin, bin b = [-1 + b/32, -1 + (b+1)/32) -- the waves
// of the scan publish their chunks' best rows here and read it back: the lower edge of the highest bin with at least M
// rows at or above it is a lower bound of the M-th largest cosine of the scanned rows.  SC_COPIES identical copies,
// SC_HSTRIDE words apart (different memory channels): a publisher adds to all of them (one posted atomic per lane), a
// reader takes copy (wave % SC_COPIES) -- the reads go past the L2 (agent scope), and thousands of waves reading ONE
// 256-byte line at the end of a short scan queued up for 40 us.
constexpr int SC_COPIES = 16;
constexpr int SC_HSTRIDE = 1088;
constexpr int SC_CTR_WORD = 128;   // word of a copy's stride that holds a chunk cursor of the tile scan's dynamic schedule (scan_tile_kernel_dyn)
// a scan wave's report whose count carries this bit may lack rows of cosine up to the float in the report's last word: valid
// when that cosine lies below the bound of the FINAL histogram (scan_wave_report, PreArgs::sc_late; report_rows below)
constexpr int SC_REPORT_LOSSY = 0x40000000;

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
// the cosine histograms of the fused tail, by the threads of one block
__device__ __forceinline__ void reset_query_hist(unsigned int* hist, int tid, int nthreads) {
    if (hist) {
        for (int i = tid; i < SC_COPIES * 64; i += nthreads) hist[(i >> 6) * SC_HSTRIDE + (i & 63)] = 0u;
        if (tid < SC_COPIES) hist[tid * SC_HSTRIDE + SC_CTR_WORD] = 0u;   // (the tile scan's chunk cursors)
    }
}

#ifdef __HIPCC__
// lower edge of the highest cosine bin with at least m rows at or above it (-2: no such bin yet), and that bin;
// lane b holds the count of bin b
__device__ __forceinline__ float sc_bound(unsigned h, int m, int lane, int& jb) {
    // suffix sums over the lanes: S_b = sum of the bins >= b (6 shuffle steps, once per 64 rows)
    unsigned sfx = h;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_down(sfx, o, 64);
        if (lane + o < 64) sfx += t;
    }
    const unsigned long long ok = __ballot(sfx >= (unsigned)m);
    jb = ok ? 63 - __builtin_clzll(ok) : -1;
    return jb >= 0 ? (float)jb * (1.0f / 32.0f) - 1.0f : -2.0f;
}
// rows of a scan wave's report against the final bound: the count, or -1 (overflow: the report cannot be used)
__device__ __forceinline__ int report_rows(const int* rep, float thr_final) {
    const int c2 = rep[0];
    if (c2 < 0 || !(c2 & SC_REPORT_LOSSY)) return c2;
    return __int_as_float(rep[SC_WCAP - 1]) < thr_final ? (c2 & 0xffff) : -1;
}
#endif

struct HostOut {
    volatile int64_t seq;
    int64_t len;
    double lambda_q;
    int status, knn_inexact, score_inexact, overflow;
    int state_reset, pad_;   // the publishing kernel has cleared QInfo's per-search state (clean search, single-query path)
    int64_t idx[MAX_TOPK];
    double score[MAX_TOPK];
};

struct RSel;
struct PreArgs;

}  // namespace as

// a gang: the workspaces whose coarse scans run as ONE launch (as_search.hip, gang_launch)
struct as_gang {
    as_query* m[4] = {nullptr, nullptr, nullptr, nullptr};
    as::PreArgs* pre = nullptr;      // [4], owned
    std::atomic<int> n{0};           // members so far (the leader is member 0)
    std::atomic<int> state{0};       // 0 gathering, 1 launched (the leader's gang_ev is recorded), 2 the launch failed
    hipEvent_t ev = nullptr;         // the leader's event
    int64_t seq = 0;                 // the gang's place in the space's order of shared scans
    ~as_gang();
};

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
    int half_enabled = 1;    // batched MFMA scan: keep the 32 slots' cosines as fp16 instead of their dots as fp32 (ARROWSPACE_BATCH_F32_DOTS=1: off)
    int dots_half = 0;       // ... and that is what the last batched scan wrote
    int gemm_variant = 0;    // batched MFMA scan (ARROWSPACE_GEMM_VARIANT): 1 = ring of 3 slabs, 2 = default cache policy, 16 = no MFMA (timing only; -DAS_ABLATION builds)
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
    double* hq = nullptr;    // pinned host staging of the query (device-readable; dp doubles per slot region, zero padded)
    double* hq_dev = nullptr;
    float* hq32 = nullptr;   // pinned fp32 query of the host-prepared fast path (dp floats, zero padded)
    float* hq32_dev = nullptr;
    const double* q64_src = nullptr;   // where the kernels behind the scan read the fp64 query this time (q64 or hq_dev)
    const float* q32_src = nullptr;
    int host_q = 0;          // this query was prepared by the host (PreArgs::host_q)
    double h_nq = 0.0, h_inq = 0.0;
    int info_clean = 0;      // a reset of QInfo's per-search state is queued behind everything that used it
    double* q64 = nullptr;   // [dp] zero padded
    float* q32 = nullptr;    // [dp]
    as::QInfo* info = nullptr;
    // int8-image scan of the host-prepared single query (as_scan.hip): the query's digit registers (pinned, read by the
    // kernel in place), its scale, and the error coefficient of THIS query's products (coef_query returns it while i8_scan is set)
    int* hq8 = nullptr;
    int* hq8_dev = nullptr;
    // batched workspace: the int8 image of the slots' queries (the items' image layout, [slots][dp8 * 2 bytes]) and their scales,
    // quantised on the device behind the staging kernel (q_quant_batch_kernel)
    signed char* q8img_dev = nullptr;
    float* faqv_dev = nullptr;
    float* hx8stat = nullptr;            // pinned: the slots' measured (u_q, v_q) of the last int8 batched pass (q_quant_batch_kernel)
    float* hx8stat_dev = nullptr;
    double x8_au = 0.0, x8_av = 0.0;     // ... and the values the host assumed when it priced that pass
    int batch_assume = 0;                // 1: the pass may be priced with the space's estimate (as_search_batch; a sharded pass: 0, a priori), 2: with x8_au / x8_av as they stand
    int x8_nan = 0;                      // ... a slot's measurement was not a number (non-finite query)
    int x8_verify = 0;                   // the last pass was: hold the measured values against the assumed ones at collect
    float h_faq = 0.0f;
    double coef_i8 = 0.0;
    int i8_scan = 0;
    // COARSE scan (single query, fused tail): the planar high digits of the items alone -- half the image's bytes again -- against
    // the query's two digits; what it drops (a2 . q, at most V |x||q|) makes every k-NN candidate's exact evaluation necessary
    // (FinishArgs::exhaustive); a query whose candidates overflow sends the next 63 to the two-digit scan
    int* hq8h = nullptr;                 // pinned: the query's digit registers in the planar rows' chunk order
    int* hq8h_dev = nullptr;
    double coef_i8h = 0.0;               // the coarse scan's coefficient for THIS query (host_query_digits)
    int coarse = 0;                      // the last scan was the coarse one
    int coarse_off = 0;                  // > 0: counting the searches that skip it
    int coarse_never = 0;                // set around the redo of a query whose coarse candidates did not fit
    int allow_coarse = 0;                // set by search_once around query_begin: the caller's tail evaluates every k-NN candidate exactly
    int chainc = 0;                      // ... and this search is a COARSE CHAIN: coarse scan without scan-side scorer candidates, every candidate list
                                         // derived from the kept dots and evaluated exactly (search_once, coarse_score_stage)
    int chainc_off = 0;                  // > 0: a coarse chain did not serve a recent query cleanly (counts the searches that skip it)
    int xknn_dirty = 0;                  // ... its counter may be non-zero (a pass died before its finish kernel)
    void* xknn = nullptr;                // [CAND_CAP] exact (id, key, distance, gy) of the coarse scan's k-NN candidates (staged_x1_kernel, xk)
    int pool_slot = 0;       // slot in the space's pool of single-query workspaces (as_search): picks the stream's priority
    int gang_ok = 0;         // set by search_once around query_begin: this scan may be shared with other callers' (gang_launch)
    as::PreArgs* defer_pre = nullptr;   // set: query_begin stops in front of the scan's launch and leaves its arguments there (search_batch_launch_pair)
    hipEvent_t gang_ev = nullptr;   // recorded behind a gang's scan on its leader's stream: the followers' tails wait for it
    float* dots32 = nullptr; // [np]
    float* part32 = nullptr; // batched workspace of rows wider than 768 floats: [K-chunk pass][slots x np] fp32 partial dots (as_scan.hip, gemm_chunks)
    double* dots64 = nullptr;
    void* pkey = nullptr;    // list path: [nwaves][64] keys (sized for double)
    int* pidx = nullptr;
    void* ckey_k = nullptr;  // filter path candidate buffers (sized for double)
    int* cidx_k = nullptr;
    void* ckey_s = nullptr;
    int* cidx_s = nullptr;
    int* sc_widx = nullptr;  // fused tail: the scan's scorer candidates, a report of SC_WCAP words per wave of the scan
    unsigned int* sc_hist = nullptr;   // SC_COPIES x SC_HSTRIDE words, zero between searches
    int sc_nw = 0;           // waves of the last fused scan
    int sc_late = 0;         // this search's tail validates lossy wave reports (set around query_begin by the paths whose tail is staged_x1_kernel)
    int last_sc_m = 0;       // what the last fused scan collected with (make_pre): the tail recomputes the bound from the final histogram
    float last_sc_w = 0.0f;
    void* gmin = nullptr;    // group minima of the scorer key
    as::RSel* rsel = nullptr; // state of the exact global selection
    as_knn_rec* knn = nullptr;
    as_hit_rec* hits = nullptr;
    as::HostOut* hout = nullptr;  // pinned
    as::HostOut* hout_dev = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    int ev_valid = 0;
    double stats[4] = {0, 0, 0, 0};
    int crowded = 0;         // > 0: recent queries overflowed the scan's candidate buffer (counts queries since)
    int crowded_direct = 0;  // this query skips the prefilter and takes the threshold repair straight away
    int fused_tail = 0;      // this search: the scan collects the scorer's candidates, ONE kernel behind it finishes the query
    int sc_crowded = 0;      // > 0: a recent query's scan-side scorer candidates overflowed (counts queries since): plain chain
    int no_fused = 0;        // ARROWSPACE_NO_FUSED_TAIL: always the plain chain (A/B runs)
    double tau_cur = 1.0;    // the tau of the search being launched (the scan's cosine window depends on it)
    int* unproven_dev = nullptr;   // build fallback: device counter (caller-owned) of rows that stay unproven
    double staged_tau = -1.0;            // as_query_search_staged: the tau of the search whose scan comes next (-1: not announced)
    int staged_sc = 0;                   // the last as_query_scan collected the scorer's candidates (SC)
    const as_knn_rec* staged_recs = nullptr;   // the gathered k-NN records as_query_lambda was given
    int64_t staged_m = 0;
    struct as_comm* comm = nullptr;      // as_query_set_comm: the library exchanges this query's records itself (as_comm.hip)
    as_knn_rec* knn_all = nullptr;       // [world][k] gathered k-NN records
    as_hit_rec* hits_all = nullptr;      // [world][topk + 1] gathered hit records
    char* xsend = nullptr;               // one-exchange pass: this rank's block (as_query_x1_bytes) ...
    char* xall = nullptr;                // ... and the gathered blocks of all ranks
    int x1_off = 0;                      // the one-exchange pass is switched off for this workspace (ARROWSPACE_STAGED_X1=0 when it was made; as_query_set_x1)
    int64_t x1_passes = 0;               // one-exchange passes this workspace has finished
    void* x1_head = nullptr;             // header of the block the last pass wrote (left zeroed by its finish kernel)
    int x1_dirty = 1;                    // ... unless that pass never reached its finish
    char* x1_own = nullptr;              // single space: the block of the fused tail's two-kernel form (search_once)
};

namespace as {

// ---- as_scan.hip
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
    // host-prepared query (single-query fast path): the host has written the fp32 query and its norms; the scan reads the
    // query from pinned host memory (as cheap as HBM at kernel start: measured, tools/probe/hostq_probe.hip) and block 0
    // files the norms in QInfo for the kernels behind it -- no staging kernel in front of the scan
    int host_q = 0;
    float nq32 = 0.0f, inq32 = 0.0f;
    double nq = 0.0, inq = 0.0;
    // ... and the scan copies the fp64 query from the host's pinned buffer into device memory on its way (the first qdp / 256
    // blocks, one element per thread, in flight with the loads every block starts with): the kernels behind the scan read it
    // from HBM / L2 -- 6 KB per BLOCK over PCIe put a floor of 4 us under a tail of 32 blocks and ruled out a wider one
    const double* q64_host = nullptr;
    double* q64_dev = nullptr;
    int qdp = 0;
    // Fused tail: the scan also collects the scorer's candidates, before lambda_q is known.  The lambda term of the
    // score lies in [(1 - tau) / 2, (1 - tau)] for lambdas in [0, 1], so a row whose cosine is more than
    // W = (1 - tau) / (2 tau) below the M-th largest cosine cannot be among the M best scores whatever lambda_q turns
    // out to be.  sc_w = W + slack; rows at or above (bound - sc_w) go to the candidate buffer (sc_idx, QInfo::sc_cnt).
    int sc_enabled = 0;
    int sc_m = 0;
    float sc_w = 0.0f;
    int* sc_idx = nullptr;   // [waves of the scan][SC_WCAP]: a wave's report -- [0] its number of candidates (-1: more than fit), [1 ..] their rows
    unsigned int* sc_hist = nullptr;   // SC_COPIES cosine histograms
    unsigned int* tile_ctrs = nullptr; // the same buffer: SC_COPIES chunk cursors in the copies' padding (set whether or not SC is)
    int sc_late = 0;         // the tail kernels validate lossy reports against the final histogram (scan_wave_report)
    int sc_dbg = 0;          // measurement only, -DAS_ABLATION builds (ARROWSPACE_SC_DBG): 1 no publication, 2 no histogram read, 4 no candidates
    // Scan of the int8 two-digit image (scan_dma_kernel<..., I8>: half the bytes of the fp32 items): the rows' scales, the
    // query's digits in the lanes' register order (per 16-byte chunk of an image row: 16 bytes that multiply into the
    // 16384-weighted sum, 16 into the 128-weighted one) and the query's scale s_q sqrt(128) / 16256
    const float* fa8 = nullptr;
    const int* q8 = nullptr;
    float faq = 0.0f;
    const float* faqv = nullptr;   // batched pass on the int8 images: the slots' scales
};

constexpr int GEMM_NSW = 6;   // slabs per wave of the batched MFMA scan: rows up to 4 * 6 * 32 floats
constexpr int GEMM_NSW_WIDE = 8;   // ... of the instantiation for wider rows (bf16 products): 1024 columns per K-chunk pass
double coef_query(const as_query* q, bool exact);
int gemm_chunks(int64_t dp, int64_t* chunk, bool bf16_products);
PreArgs make_pre(as_query* q, double eps, int64_t exclude, bool enabled);
as_status launch_scan(as_query* q, const PreArgs& pre);
bool scan_dual_ok(const as_query* a, const as_query* b);
as_status launch_scan_dual(as_query* a, as_query* b, const PreArgs& pa, const PreArgs& pb, hipStream_t st);
as_status launch_scan_gang(as_query* const* m, const PreArgs* pre, int n, hipStream_t st);
void set_tile_geom(int v);
void set_tile_dyn(int v);
void set_x1_blocks(int v);
as_status set_scan_attrs();   // per-device dynamic-LDS opt-in of the scan kernels

}  // namespace as
