/* arrowspace_hip.h -- C ABI of the MI355X-native hot path (gfx950).
 *
 * Drop-in boundary for the two calls the reference's PyO3 shim forwards to the
 * `arrowspace` crate:
 *     ArrowSpaceBuilder.build(graph_params, items)   /root/reference/src/lib.rs:271-300
 *     ArrowSpace.search(item, gl, tau)               /root/reference/src/lib.rs:132-174
 * plus the accessors the shim exposes (src/lib.rs:40-61, 78-124) and the debug
 * switch (src/helpers.rs:12-21).  Plain pointers and sizes only; no torch / numpy
 * types.  INTEGRATION.md shows the PyO3-side and ctypes-side bindings.
 *
 * Conventions
 *   - every function that can fail returns an as_status; 0 == AS_OK;
 *     as_last_error() returns a thread-local message for the last failure;
 *   - outputs are caller-allocated; handles are opaque and immutable after build;
 *   - AS_EINVAL maps to the shim's PyValueError, AS_EZEROLAMBDA to the
 *     `assert_ne!(lambda_q, 0.0)` panic at src/lib.rs:156-159, AS_EHIP to a
 *     runtime failure of the device / HIP runtime.
 *   - "host" pointers are ordinary CPU memory, "dev" pointers are HIP device memory
 *     on the device the handle lives on.
 */
#ifndef ARROWSPACE_HIP_H
#define ARROWSPACE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    AS_OK = 0,
    AS_EINVAL = 1,       /* bad argument            -> ValueError   */
    AS_EZEROLAMBDA = 2,  /* lambda_q == 0           -> PanicException (src/lib.rs:156-159) */
    AS_EHIP = 3,         /* HIP runtime failure     -> RuntimeError */
    AS_EUNSUPPORTED = 4, /* outside supported range -> ValueError   */
    AS_ENOMEM = 5
} as_status;

/* graph_params dict of the reference (src/helpers.rs:48-76).  has_sigma == 0
 * reproduces "sigma missing or None -> eps * 0.5" (src/helpers.rs:68-72). */
typedef struct {
    double eps;
    int64_t k;
    int64_t topk;
    double p;
    double sigma;
    int32_t has_sigma;
    int32_t _pad;
} as_graph_params;

enum { AS_METRIC_L2 = 0, AS_METRIC_COSINE = 1 };
enum { AS_KERNEL_GAUSSIAN = 0, AS_KERNEL_RATIONAL = 1 };
/* Which lambda the index carries.  ITEM: node-local spectral energy on the N-node item graph (BASELINE.json
 * north_star).  FEATURE: the synthetic index of the reference's notes -- a Rayleigh quotient and an edgewise
 * dispersion on an F x F feature-space Laplacian whose nodes are the D columns of the item matrix
 * (/root/reference/TAUMODE.md:8,12-27, GRAPH_VARIABLES.md:17); the GraphLaplacian is then that F x F object
 * (nnodes == nfeatures), the one `prepare_query_item(&v, gl)` takes the query's quotient against
 * (/root/reference/src/lib.rs:154). */
enum { AS_LAMBDA_ITEM = 0, AS_LAMBDA_FEATURE = 1 };
enum { AS_KEEP_F64_AUTO = 0, AS_KEEP_F64_ALWAYS = 1 };
enum { AS_DTYPE_F32 = 0, AS_DTYPE_F64 = 1 };

/* Options that have no counterpart in the reference's dict; zero-initialised ==
 * north_star defaults (L2 distance, Gaussian weights, current device). */
typedef struct {
    int32_t metric;   /* AS_METRIC_*  (cosine = GRAPH_VARIABLES.md:7 variant) */
    int32_t kernel;   /* AS_KERNEL_*  (rational = GRAPH_VARIABLES.md:9 variant) */
    int32_t device;   /* HIP device ordinal; -1 = current device */
    int32_t keep_f64; /* AS_KEEP_F64_*: AUTO keeps an fp64 copy of the items only when
                         they are not exactly representable in fp32 */
    int32_t force_exact; /* 1: skip the fp32 fast paths (fp64 everywhere; for tests) */
    int32_t search_mode; /* test hook: start searches on a fallback path (bit0 fp64, bit1 wavefront-list selection) */
    int32_t lambda_mode; /* AS_LAMBDA_* */
    int32_t reserved;
} as_opts;

typedef struct as_space as_space; /* crate `ArrowSpace`      (src/lib.rs:64-67)  */
typedef struct as_graph as_graph; /* crate `GraphLaplacian`  (src/lib.rs:26-29)  */
typedef struct as_query as_query; /* per-search device workspace (no reference counterpart) */

/* ---- index build: replaces RustBuilder::...build(rows), src/lib.rs:278-289 ---- */

/* items: host fp64, element strides (numpy `as_array()` accepts any strides,
 * src/helpers.rs:25).  n == 0 or d == 0 -> AS_EINVAL (src/helpers.rs:27-29). */
as_status as_build(const double* items, int64_t n, int64_t d, int64_t row_stride, int64_t col_stride,
                   const as_graph_params* gp, const as_opts* opts, as_space** out_space, as_graph** out_graph);

/* Same, items already resident in HBM (row-major, leading dimension ld elements).  Synchronises the device
 * first: whatever stream produced the items, they are complete when the ingest reads them. */
as_status as_build_dev(const void* items_dev, int32_t dtype, int64_t n, int64_t d, int64_t ld,
                       const as_graph_params* gp, const as_opts* opts, as_space** out_space, as_graph** out_graph);

/* ---- staged build: the three steps as_build composes; multi-GPU hosts call them
 *      with a row range per rank and all-gather the lists in between ---- */

/* step 1: ingest items into the HBM layout (fp32 padded tile layout + norms). */
as_status as_space_create_dev(const void* items_dev, int32_t dtype, int64_t n, int64_t d, int64_t ld,
                              const as_opts* opts, as_space** out_space);
/* step 2: exact directed k-NN lists for rows [row_begin, row_end) against all n items.
 * Outputs (device, caller-allocated, (row_end-row_begin) x k, row-major):
 *   out_idx int32 (-1 padded), out_key / out_dist / out_gy fp64, out_cnt int32 per row. */
as_status as_knn_rows(const as_space* sp, const as_graph_params* gp, int64_t row_begin, int64_t row_end,
                      int32_t* out_idx_dev, double* out_key_dev, double* out_dist_dev, double* out_gy_dev,
                      int32_t* out_cnt_dev);
/* step 3: symmetrise + normalised Laplacian + per-item energy + lambdas from the
 * complete n x k lists (device).  Writes lambdas into the space. */
as_status as_graph_from_knn(as_space* sp, const as_graph_params* gp, const int32_t* idx_dev,
                            const double* dist_dev, const double* gy_dev, const int32_t* cnt_dev,
                            as_graph** out_graph);

/* ---- staged build without replication (row-sharded multi-GPU): a rank's space holds only ITS rows;
 *      the other ranks' shards visit one at a time as temporary spaces (ring send/recv in the host).  Per visiting
 *      block as_knn_block fills that block's slice of the partial lists (M = as_knn_list_width(k) exact entries per
 *      row, ids global); as_knn_merge ranks the slices of all blocks, applies eps and k and flags the rows for which
 *      something a block dropped could still belong to the answer; those rows go round once more through
 *      as_knn_block_band (complete collection inside the proven band), then as_knn_merge again.  Partial arrays are
 *      device memory, caller-allocated: key / dist / gy fp64 and idx int32 [nblocks][rows][M], cnt int32 and t32 fp32
 *      [nblocks][rows]; the pointers passed to as_knn_block / _band are those of ONE block's slice. ---- */
int32_t as_knn_list_width(int64_t k);                 /* M; < 0 when k is not supported */
int32_t as_record_capacity(int32_t which);            /* 0: k-NN records a query merges (ranks x k); 1: hit records (ranks x (topk + 1)) */
double as_space_nmax(const as_space* sp);             /* largest squared norm (error bound of a block's dropped candidates) */
as_status as_space_norms(const as_space* sp, double* out_dev); /* fp64 squared norms of the space's rows, device to device */
int64_t as_space_row_offset(const as_space* sp);      /* Ring build: may the block passes (as_knn_block / _pair / _band) run on the int8 two-digit images of the shards?  Every rank
 * reports what its own rows measure -- out3 = {U, V, 1.0 when its image cannot be used (non-finite rows, ARROWSPACE_K2_NO_I8) else
 * 0.0} (as_ring_i8_stats makes the image) --, the host all-gathers the three numbers and hands every rank the ring-wide maxima:
 * as_ring_i8_set(sp, max U, max V, usable = no rank said 1).  usable = 0 or a coefficient 2.001 U + V^2 beyond 1e-3 keeps the
 * bf16 head + tail form, on every rank alike (as_ring_i8: what was decided).  Graphs are the same bits either way.
 * Serves ArrowSpaceBuilder::build, /root/reference/src/lib.rs:281-331, on a row-sharded index. */
as_status as_ring_i8_stats(as_space* sp, double* out3);
as_status as_ring_i8_set(as_space* sp, double u_max, double v_max, int32_t usable);
int32_t as_ring_i8(const as_space* sp);
/* global index of row 0 (set by as_graph_from_knn_global) */
as_status as_knn_block(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t row_begin, int64_t row_end,
                       int64_t row_goff, int64_t col_goff, double* p_key_dev, double* p_dist_dev, double* p_gy_dev,
                       int32_t* p_idx_dev, int32_t* p_cnt_dev, float* p_t32_dev);
/* Symmetric ring: an unordered pair of blocks is computed ONCE.  as_knn_block_pair runs the own rows
 * [row_begin, row_end) against the visiting block's column tiles [col_tile_begin, col_tile_end) (128 items each; -1 /
 * -1 = all) and lets every key serve both items: p_* is the own rows' slice exactly as as_knn_block's (indexed from
 * row_begin), q_* the slice of ALL visiting items [cols nitems][M] with the own rows as their columns (ids global),
 * refined here while both shards are resident, to be sent home and folded there (as_knn_fold with the sender's
 * block_nmax).  A visiting item's candidates are admitted up to min(eps bound, col_thr_dev[item]); q_t32 carries the
 * bound of what was turned away (-inf when its buffer overflowed: the item fails its proof and takes the second
 * round).  as_knn_thresholds derives such thresholds from a rank's folded list (an upper bound of each row's M-th
 * smallest fp32 key over all columns: its M-th exact key so far + the error bound; +inf while the list is not full);
 * nmax_all = the largest squared norm of the whole index.  row_thr_dev (may be NULL): the same kind of thresholds for the
 * OWN rows, indexed by the row's number in sp ([nitems]) -- a row then starts every unit from min(eps bound, threshold)
 * instead of the eps bound (an eps that admits every pair leaves only the thresholds to prune with).  Scratch: the
 * visiting block is taken in chunks of column
 * tiles whose transposed buffers (8 KiB per item at M = 64) fit in an eighth of the free memory, at most 16 GB. */
as_status as_knn_block_pair(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t row_begin, int64_t row_end,
                            int64_t col_tile_begin, int64_t col_tile_end, int64_t row_goff, int64_t col_goff,
                            const float* col_thr_dev, const float* row_thr_dev, double* p_key_dev, double* p_dist_dev, double* p_gy_dev,
                            int32_t* p_idx_dev, int32_t* p_cnt_dev, float* p_t32_dev, double* q_key_dev, double* q_dist_dev,
                            double* q_gy_dev, int32_t* q_idx_dev, int32_t* q_cnt_dev, float* q_t32_dev);
as_status as_knn_thresholds(const as_space* sp, const as_graph_params* gp, int64_t row_begin, int64_t row_end, double nmax_all,
                            const double* r_key_dev, const int32_t* r_cnt_dev, float* out_thr_dev);
as_status as_knn_merge(const as_space* sp, const as_graph_params* gp, int64_t row_begin, int64_t row_end, int32_t nblocks,
                       const double* p_key_dev, const double* p_dist_dev, const double* p_gy_dev, const int32_t* p_idx_dev,
                       const int32_t* p_cnt_dev, const float* p_t32_dev, const double* block_nmax_host, int32_t* out_idx_dev,
                       double* out_key_dev, double* out_dist_dev, double* out_gy_dev, int32_t* out_cnt_dev,
                       int32_t* out_flag_dev, double* out_band_dev, int64_t* out_nflagged);
/* Running fold (what dist.py uses: two slices of list memory whatever the number of ranks): r_* <- the M smallest exact
 * entries of r_* and b_* per row; r_t32 keeps min over blocks of (T32_b - e_b).  mode 0: every row; 1: rows with
 * flag_dev != 0; 2: those rows, their running list discarded first (first block of the second round).  as_knn_merge
 * with nblocks == 0 then finalises the one folded slice. */
as_status as_knn_fold(const as_space* sp, const as_graph_params* gp, int64_t row_begin, int64_t row_end, int32_t mode,
                      double block_nmax, const int32_t* flag_dev, double* r_key_dev, double* r_dist_dev, double* r_gy_dev,
                      int32_t* r_idx_dev, int32_t* r_cnt_dev, float* r_t32_dev, const double* b_key_dev, const double* b_dist_dev,
                      const double* b_gy_dev, const int32_t* b_idx_dev, const int32_t* b_cnt_dev, const float* b_t32_dev);
as_status as_knn_block_band(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t row_begin,
                            int64_t row_end, int64_t row_goff, int64_t col_goff, int32_t* flag_dev /* bit 1: band overflow */,
                            const double* band_dev, double* p_key_dev, double* p_dist_dev, double* p_gy_dev,
                            int32_t* p_idx_dev, int32_t* p_cnt_dev, float* p_t32_dev, int64_t* out_overflowed);
/* Last resort of the ring (third round): rows whose band overflowed the collection buffers somewhere (flag_dev != 0 --
 * thousands of exact duplicates, or of items at one distance) take the block's k nearest inside eps by exact evaluation
 * of every pair, ordered by (key, global id); the slice carries no drop bound.  Row-serial cost, as the last resort of
 * as_knn_rows (the reference's own loop is this for every row: the crate's all-pairs build, src/lib.rs:278-289). */
as_status as_knn_block_exact(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t row_begin, int64_t row_end,
                             int64_t row_goff, int64_t col_goff, const int32_t* flag_dev, double* p_key_dev, double* p_dist_dev,
                             double* p_gy_dev, int32_t* p_idx_dev, int32_t* p_cnt_dev, float* p_t32_dev);
/* step 3 over the lists of ALL n_global items for a space that holds the rows [row_offset, row_offset + nitems):
 * global graph, Laplacian, energies, tau0; this shard's lambdas into the space.  n64_global_dev: fp64 squared norms
 * of all items (device).  Searches on the space then report global item ids. */
as_status as_graph_from_knn_global(as_space* sp, const as_graph_params* gp, int64_t n_global, int64_t row_offset,
                                   const int32_t* idx_dev, const double* dist_dev, const double* gy_dev,
                                   const int32_t* cnt_dev, const double* n64_global_dev, as_graph** out_graph);

/* Sharded graph stage, in front of the host's variable-count all-to-all: the directed edges i -> j of this rank's k-NN lists
 * (idx / dist / gy [rows][k], cnt[rows] valid entries per row; device pointers), bucketed by the rank that owns item j
 * (rank r owns [bounds_host[r], bounds_host[r + 1])): out_ints [E][2] int32 = (j - bounds[owner], row0 + i), out_reals [E][2]
 * = (dist, gy), buckets in rank order, the lists' (row, slot) order inside a bucket; out_counts_host[r] = entries for rank r.
 * The output buffers hold rows * k entries.  Count / scan / scatter kernels on `hip_stream`; synchronises it.
 * Serves ArrowSpaceBuilder::build, /root/reference/src/lib.rs:281-331, on a row-sharded index. */
as_status as_edges_bucket(const int32_t* idx_dev, const double* dist_dev, const double* gy_dev, const int32_t* cnt_dev, int64_t rows, int64_t k,
                          int64_t row0, const int64_t* bounds_host, int32_t world, int32_t* out_ints_dev, double* out_reals_dev,
                          int64_t* out_counts_host, void* hip_stream);
/* step 3 without replication (SURVEY 8e "Symmetrise + Laplacian": one exchange step).  The space holds the rows
 * [row_offset, row_offset + nitems); idx/dist/gy/cnt are ITS rows' lists (ids global).  in_*: the directed edges
 * (in_col_dev[e] -> row_offset + in_row_dev[e]) of ALL ranks whose target row lives here, this rank's own included
 * -- what the host's variable-count all-to-all delivers.  as_graph_shard_csr builds these rows of the symmetrised
 * graph (column ids global) and their degrees (as_graph_deg_copy: nitems doubles, device to device);
 * as_graph_shard_energy takes the degrees and squared norms of all items (all-gathered: 16 B per item) and leaves
 * the rows' energies behind (as_graph_energy_copy); as_graph_shard_lambdas takes the energies of all items (8 B per
 * item), selects tau0 over them and writes this shard's lambdas into the space.  O(N k) data never leaves its rank
 * except as the edges themselves. */
as_status as_graph_shard_csr(as_space* sp, const as_graph_params* gp, int64_t n_global, int64_t row_offset,
                             const int32_t* idx_dev, const double* dist_dev, const double* gy_dev, const int32_t* cnt_dev,
                             int64_t n_in, const int32_t* in_row_dev, const int32_t* in_col_dev, const double* in_dist_dev,
                             const double* in_gy_dev, as_graph** out_graph);
as_status as_graph_deg_copy(const as_graph* gr, double* out_dev);
as_status as_graph_shard_energy(as_space* sp, as_graph* gr, const double* deg_global_dev, const double* n64_global_dev);
as_status as_graph_energy_copy(const as_graph* gr, double* out_dev);
as_status as_graph_shard_lambdas(as_space* sp, as_graph* gr, const double* E_global_dev, int64_t n_global);
int64_t as_graph_row_offset(const as_graph* gr);      /* first row a sharded graph holds (0 for a whole graph) */
int64_t as_graph_ncols(const as_graph* gr);           /* items the graph's columns range over */
int64_t as_graph_nitems(const as_graph* gr);         /* items the whole index covers (all ranks' rows), item or feature mode */

/* ---- staged build, feature mode (AS_LAMBDA_FEATURE): what as_build composes when opts->lambda_mode selects
 *      the F x F feature-space Laplacian; multi-GPU hosts call the steps with a row range per rank and exchange
 *      the D x D Gram partials and the N energies in between ---- */

/* step 2f: Gram of the columns over items [row_begin, row_end): out_gram_dev[a*d + b] = sum_i x_ia x_ib
 * (fp64, d x d, symmetric, device).  Partial Grams of disjoint row ranges add up to the full one. */
as_status as_feat_gram(const as_space* sp, int64_t row_begin, int64_t row_end, double* out_gram_dev);
/* step 3f: feature graph (k-NN over the D columns, union symmetrisation, weights, degrees; L = D - W) from
 * the complete Gram. */
as_status as_feat_graph(const as_space* sp, const as_graph_params* gp, const double* gram_dev, as_graph** out_graph);
/* step 4f: Rayleigh energy E and dispersion G of items [row_begin, row_end), written at those positions of
 * the n-long device arrays. */
as_status as_feat_energy(const as_space* sp, const as_graph* gr, int64_t row_begin, int64_t row_end,
                         double* E_dev, double* G_dev);
/* step 5f: tau0 = median of the positive E, lambdas of all n items into the space (E, G complete, device). */
as_status as_feat_lambdas(as_space* sp, as_graph* gr, const double* E_dev, const double* G_dev);
/* row-sharded form: E, G of ALL n_global items; this shard (rows [row_offset, row_offset + nitems)) gets its lambdas */
as_status as_feat_lambdas_global(as_space* sp, as_graph* gr, const double* E_dev, const double* G_dev, int64_t n_global,
                                 int64_t row_offset);

/* ---- search: replaces prepare_query_item + search_lambda_aware, src/lib.rs:154,173 ---- */

/* query: host fp64, contiguous, length d (d != nfeatures -> AS_EINVAL,
 * src/lib.rs:140-146).  out_idx/out_score: capacity >= min(topk, nitems).
 * lambda_q == 0 -> AS_EZEROLAMBDA.  Results: score desc, index asc.
 * RE-ENTRANT across host threads: every call runs on a workspace of its own -- stream, device buffers, pinned
 * results -- taken from a pool the space grows on demand (up to 4, ARROWSPACE_SEARCH_POOL); no lock is held while a
 * search runs, so one thread's scan overlaps another's finish kernel and host turnaround.  The reference holds the GIL
 * through prepare_query_item + search_lambda_aware (src/lib.rs:132-174): its searches are serialised. */
as_status as_search(const as_space* sp, const as_graph* gr, const double* query, int64_t d, double tau,
                    int64_t* out_idx, double* out_score, int64_t* out_len, double* out_lambda_q);
/* matrix pipe the last k-NN pass over this space's rows ran on (as_build, as_knn_rows): 0 fp32 (ARROWSPACE_K2_FP32=1), 1 bf16
 * head + tail, 2 int8 two-digit image (the default where the items' quantisation error allows it); -1: none yet.  The graphs
 * are the same bits on every pipe: the pipe only prefilters, the refinement is exact. */
int32_t as_space_knn_pipe(const as_space* sp);
/* 1 when the last single-query scan (of this workspace / of the space's first pooled workspace) read the int8 two-digit image
 * of the items -- half the bytes of the fp32 matrix -- instead of the fp32 matrix (ARROWSPACE_SCAN_FP32=1, rows wider than 2 048
 * floats, queries or items the image represents badly).  Either way the scan only prefilters: results are exact. */
int32_t as_query_scan_int8(const as_query* q);
int32_t as_last_scan_int8(const as_space* sp);
/* 1 when the last batched pass of as_search_batch (its first workspace) ran on the int8 images of items and queries
 * (v_mfma_i32_32x32x32_i8, three products per column) rather than on the bf16 head + tail of the fp32 items. */
int32_t as_last_batch_int8(const as_space* sp);
/* Scans of as_search_batch that served TWO passes (64 queries) with one read of the items: calls of more than 32 queries launch
 * their passes in pairs, and a pair on the int8 images of rows up to 768 columns shares its scan (ARROWSPACE_NO_BATCH_DUAL=1:
 * never).  A count since the space was made. */
int64_t as_batch_dual_scans(const as_space* sp);
/* workspaces the pool of as_search holds at the moment (1 after single-threaded use) */
int32_t as_search_pool_size(const as_space* sp);
/* Concurrent as_search callers on one space share a pass over the items where they can (coarse scans of tau >= 0.4 searches that
 * arrive within a few microseconds of each other run as ONE launch for up to four queries: "gang scans"; ARROWSPACE_GANG=0:
 * never).  out[i] = scans this space launched through that path with i + 1 members (i = 0 .. 3); out[4 .. 9] = host-prepared scans
 * that did not take it, by first reason: no concurrent callers lately, not a search for the fused tail (tau < 0.4, crowded
 * candidates lately), not a coarse scan, per-search state still to be reset, per-launch timing on, a row range.  Results never
 * depend on any of it. */
as_status as_gang_counters(const as_space* sp, int64_t* out, int32_t n);

/* B queries, row-major [b][d]; outputs [b][topk]; status per query in out_status
 * (AS_OK / AS_EZEROLAMBDA).  Extension. */
as_status as_search_batch(const as_space* sp, const as_graph* gr, const double* queries, int64_t b, int64_t d,
                          double tau, int64_t* out_idx, double* out_score, int64_t* out_len,
                          double* out_lambda_q, int32_t* out_status);

/* ---- staged search (row-sharded multi-GPU; as_search composes these on one GPU) ---- */

/* fixed-size device records exchanged between ranks */
typedef struct {
    int64_t idx;  /* global item index, -1 = empty slot */
    double key;   /* eps-test / ordering key: squared L2 distance or cosine distance */
    double dist;  /* d(q, x_idx) */
    double gy;    /* y_q . y_idx */
    double deg;   /* degree of item idx in the index graph */
    double ny;    /* |y_idx|^2 */
} as_knn_rec;

typedef struct {
    int64_t idx; /* global item index, -1 = empty slot */
    double score;
} as_hit_rec;

as_status as_query_create(const as_space* sp, const as_graph* gr, as_query** out);
void as_query_free(as_query* q);
/* rows [row_begin,row_end) are this rank's shard; row_offset maps local rows of a
 * shard-only space to global indices (0 when the space holds all items). */
as_status as_query_scan(as_query* q, const double* query_host, int64_t d, int64_t row_begin, int64_t row_end);
/* k records: this shard's exact nearest items to the query (device pointer) */
const as_knn_rec* as_query_knn_records(const as_query* q);
int64_t as_query_knn_capacity(const as_query* q);
/* lambda_q from m records (own, or all-gathered from every rank) */
as_status as_query_lambda(as_query* q, const as_knn_rec* recs_dev, int64_t m);
/* local scoring of the scanned rows; topk records */
as_status as_query_score(as_query* q, double tau);
const as_hit_rec* as_query_hit_records(const as_query* q);
int64_t as_query_hit_capacity(const as_query* q);
/* merge m hit records (own or all-gathered), copy to host; synchronises. */
as_status as_query_finish(as_query* q, const as_hit_rec* hits_dev, int64_t m, int64_t* out_idx,
                          double* out_score, int64_t* out_len, double* out_lambda_q);
/* Batched staged search (extension: 32 query slots per pass over this rank's rows, GEMM-shaped scan; the two
 * record exchanges of a pass are amortised over the slots).  Record buffers: [slot][k] as_knn_rec and
 * [slot][topk + 1] as_hit_rec (as_query_bind_records); the all-gathered buffers handed back are
 * [rank][slot][...].  out_status[b]: AS_OK / AS_EZEROLAMBDA, or -1 = rerun query b through the single-query steps. */
as_status as_query_create_batch(const as_space* sp, const as_graph* gr, as_query** out);
int32_t as_query_slots(const as_query* q);
as_status as_query_scan_batch(as_query* q, const double* queries_host, int32_t nb, int64_t d, int64_t row_begin, int64_t row_end);
as_status as_query_lambda_batch(as_query* q, const as_knn_rec* recs_dev, int32_t nranks);
as_status as_query_score_batch(as_query* q, double tau);
as_status as_query_finish_batch(as_query* q, const as_hit_rec* hits_dev, int32_t nranks, int64_t* out_idx, double* out_score,
                                int64_t* out_len, double* out_lambda_q, int32_t* out_status);
/* flags bit0: run the following scans in fp64 end to end (the fallback as_search takes when
 * the fp32 candidate lists are not provably exact); bit1: wavefront-list selection instead
 * of the filter buffers (taken when a buffer overflowed); bit2: the next as_query_scan does
 * not scan -- it re-derives the k-NN candidates of the SAME query and row range from the dot
 * products the previous scan left in HBM (threshold selection; taken when only the k-NN
 * buffer overflowed: more than 4096 rows inside eps).  as_query_flags reports the last
 * finished search: bit0 = not provably exact, bit1 = this side's candidate buffer overflowed
 * (knn_inexact: the k-NN buffer, score_inexact: the scorer's). */
void as_query_set_exact(as_query* q, int32_t flags);
as_status as_query_flags(const as_query* q, int32_t* knn_inexact, int32_t* score_inexact);
/* ONE exchange per sharded query (tau in [0.4, 1], item-graph lambda; as_query_x1_usable says so, identically on every rank):
 * as_query_x1_begin scans this rank's rows and finishes what only this rank can finish -- its k exact nearest rows (records)
 * and the exact cosine and lambda of every row its scan kept as a scorer candidate -- into ONE block of as_query_x1_bytes
 * bytes at send_dev (device memory of the caller: an all-gather send buffer); the host all-gathers the blocks of the `world`
 * ranks in rank order; as_query_x1_finish forms lambda_q, scores and ranks every rank's candidates (identically on every rank)
 * and copies the answer out.  Afterwards as_query_flags as after as_query_finish; as_query_x1_redo != 0: some rank's candidates
 * did not fit its block (or it had none to offer): run the query again through as_query_scan / _lambda / _score / _finish.
 * as_query_search_staged takes this pass by itself (ARROWSPACE_STAGED_X1=0: never).
 * Serves PyArrowSpace::search, /root/reference/src/lib.rs:132-174, on a row-sharded index. */
int64_t as_query_x1_bytes(const as_query* q, int32_t world);
int32_t as_query_x1_usable(const as_query* q, double tau);
/* The pass's switch is a property of the workspace: read from ARROWSPACE_STAGED_X1 when the workspace is made
 * (as_query_x1_enabled), set for good by as_query_set_x1 -- the host of a row-sharded index agrees on it over its ranks (one
 * all-reduce, MIN) when it opens the query: every rank must issue the same collectives for every query that follows. */
int32_t as_query_x1_enabled(const as_query* q);
void as_query_set_x1(as_query* q, int32_t enabled);
as_status as_query_x1_begin(as_query* q, const double* query_host, int64_t d, int64_t row_begin, int64_t row_end, double tau, void* send_dev,
                            int32_t world);
as_status as_query_x1_finish(as_query* q, const void* all_dev, int32_t world, double tau, int64_t* out_idx, double* out_score, int64_t* out_len,
                             double* out_lambda_q);
int32_t as_query_x1_redo(const as_query* q);
/* A pass that came back with as_query_x1_redo != 0 may be the coarse scan's doing (tau >= 0.4: a rank scans the high digits of
 * its int8 image alone, with wider candidate windows): as_query_set_coarse(q, 0), run the pass once more (every rank alike),
 * as_query_set_coarse(q, 1); only if it comes back with redo again take the two-exchange steps. */
void as_query_set_coarse(as_query* q, int32_t allowed);
int64_t as_query_x1_passes(const as_query* q);   /* one-exchange passes this workspace has finished */
/* The per-query exchange steps issued by the library itself (RCCL on the query's stream; librccl is taken from the
 * process or /opt/rocm/lib by dlopen -- as_comm_available() == 0 on a box without it).  The ranks of an index share one
 * communicator: rank 0 draws an id (as_comm_unique_id: 128 bytes), the host hands it to every rank (any broadcast),
 * every rank calls as_comm_create (collective).  as_query_set_comm binds a single-query workspace to it;
 * as_query_search_staged is then ONE host call per query: the one-exchange pass above where it applies, else scan of this
 * rank's rows, all-gather of the k-NN records, lambda_q, scorer, all-gather of the hit records, merge; and the escalation to
 * the exact paths (every rank with the same query and tau; same results and errors as as_search on one space holding
 * every item).
 * Serves PyArrowSpace::search, /root/reference/src/lib.rs:132-174, on a row-sharded index. */
typedef struct as_comm as_comm;
int32_t as_comm_available(void);
as_status as_comm_unique_id(void* out_128_bytes);
as_status as_comm_create(const void* id_128_bytes, int32_t rank, int32_t world, int32_t device, as_comm** out);
void as_comm_free(as_comm* c);
as_status as_query_set_comm(as_query* q, as_comm* c);
as_status as_query_search_staged(as_query* q, const double* query_host, int64_t d, int64_t row_begin, int64_t row_end, double tau,
                                 int64_t* out_idx, double* out_score, int64_t* out_len, double* out_lambda_q);
/* HIP stream (hipStream_t) the query's kernels run on, for event timing / ordering;
 * as_query_set_stream makes them run on the caller's stream (NULL restores the private one),
 * e.g. the stream torch.distributed orders its RCCL collectives against. */
void as_query_set_stream(as_query* q, void* hip_stream);
/* Make the query write its k k-NN records and its topk+1 hit records (the last one carries
 * the exactness flags) into caller-owned device buffers, e.g. all-gather send buffers. */
as_status as_query_bind_records(as_query* q, as_knn_rec* knn_dev, as_hit_rec* hits_dev);
void* as_query_stream(const as_query* q);

/* ---- accessors: src/lib.rs:40-61 (GraphLaplacian), 78-124 (ArrowSpace) ---- */
/* searches on this space whose answer was returned although it failed its a-posteriori check even on the strongest
 * (fp64) path: more near-ties at the k-th distance / score than fp64 can order.  0 in normal operation. */
int64_t as_unproven_searches(const as_space* sp);
int64_t as_nitems(const as_space* sp);
int64_t as_nfeatures(const as_space* sp);
as_status as_get_item(const as_space* sp, int64_t idx, double* out_vec, double* out_lambda);
as_status as_lambdas(const as_space* sp, double* out);
int64_t as_nnodes(const as_graph* gr);
as_status as_get_graph_params(const as_graph* gr, as_graph_params* out); /* sigma resolved */
int64_t as_graph_nnz(const as_graph* gr); /* stored Laplacian entries incl. diagonal */
/* CSR of the normalised Laplacian: indptr int64[n+1], indices int64[nnz], values fp64[nnz] */
as_status as_graph_csr(const as_graph* gr, int64_t* indptr, int64_t* indices, double* values);
as_status as_graph_degrees(const as_graph* gr, double* out);
double as_graph_tau0(const as_graph* gr);
int32_t as_graph_lambda_mode(const as_graph* gr); /* AS_LAMBDA_* */
/* device pointer to the fp64 lambdas (n) -- for multi-GPU hosts */
const double* as_lambdas_dev(const as_space* sp);

/* per-stage seconds of the last build, and kernel-only seconds of the X.X^T block:
 * out[0]=ingest out[1]=knn_mfma out[2]=refine out[3]=fallback_exact out[4]=graph
 * out[5]=total out[6]=fallback_rows (rows recomputed by the row-serial fp64 path) out[7]=mfma_flops_issued
 * out[8]=unproven_rows (fallback rows whose list could still not be proven exact: more near-ties at the k-th
 * distance than the widest candidate list holds, inside fp64 rounding of one another) out[9]=band_rows (rows
 * settled by the second, band-collecting pass) */
as_status as_build_stats(const as_graph* gr, double* out, int32_t n);
/* per-stage device microseconds of the last search on q (HIP events):
 * out[0]=scan out[1]=rest out[2]=exact_fallback_used */
as_status as_query_stats(const as_query* q, double* out, int32_t n);

/* HIP-event timing of searches is off by default (the events cost a few microseconds per
 * query); enable it before the searches whose stats are read */
void as_enable_search_stats(int32_t enabled);
/* Measurement knob, no reference counterpart: launch parameters the library otherwise picks itself.  "tile_geom" = <blocks per
 * CU><two digits: ring KiB per wave> of the single query's tile scan (e.g. 406, 308, 216; ARROWSPACE_TILE_GEOM at load);
 * "x1_blocks" = blocks of the coarse scan's exact-evaluation kernel (16 .. 256, default 128).
 * Returns 0, or 1 for an unknown key.  Results never depend on it. */
int32_t as_set_tuning(const char* key, int32_t value);
/* same, for the workspace as_search keeps inside the space (last as_search call) */
as_status as_last_search_stats(const as_space* sp, double* out, int32_t n);
/* single-query searches on this space since it was made: out[0] searches, [1] zero-lambda results (src/lib.rs:156-159),
 * [2] reruns after a failed k-NN a-posteriori check, [3] after a candidate-buffer overflow, [4] after a failed scorer
 * check, [5] searches that took at least one rerun.  A rerun costs a second pass over the items. */
as_status as_search_counters(const as_space* sp, int64_t* out, int32_t n);

/* ---- index persistence (extension; the reference exposes none): one flat file
 * holding the items, lambdas and graph arrays.  Loading re-ingests the items and uploads the
 * rest; no k-NN work is redone.  opts->device selects the GPU, metric/kernel come from the file. */
as_status as_index_save(const as_space* sp, const as_graph* gr, const char* path);
as_status as_index_load(const char* path, const as_opts* opts, as_space** out_space, as_graph** out_graph);

void as_free_space(as_space* sp);
void as_free_graph(as_graph* gr);

/* ---- misc ---- */
void as_set_debug(int32_t enabled); /* src/helpers.rs:12-14; messages "[pyarrowspace] ..." on stderr */
const char* as_last_error(void);
int32_t as_device_count(void);
const char* as_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ARROWSPACE_HIP_H */
