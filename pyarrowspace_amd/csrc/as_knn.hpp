// Shared by the k-NN build kernels (as_build.hip: the fp32 kernel and the host side; as_k2bf.hip: the bf16 head + tail
// kernel).  Not part of the C ABI.
#pragma once
#include "as_query.hpp"

namespace as {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Block tile of the fused X.X^T kernels: 256 rows x 128 columns, K-slab of 32 columns; CAP entries per row in the
// per-block append buffers.
constexpr int BM = 256, BN = 128, BK = 32, CAP = 256;
constexpr int DROW = 32;                       // floats per row of a DMA slab (no padding)
constexpr int DSLAB = (BM + BN) * DROW;        // floats per slab buffer: A rows then B rows

struct KnnArgs {
    const float* x32;
    const float* n32;
    const float* inorm32;
    int64_t n, dp;
    int64_t r0, r1;
    int nrb, S, ntile, M, metric;
    float epskey, coef, nmax;
    float* buf_key;
    int* buf_idx;
    float* out_key;  // [(r1-r0)][S][M]
    int* out_idx;
    int* out_cnt;    // [(r1-r0)][S]: count | dropped<<30 | overflowed<<31 (collect mode)
    // Row side (A operand) and column side (B operand = x32 / n32 / inorm32 / n above) are separate: the same space
    // for a single-GPU build, this rank's shard against a visiting shard on the ring.  Item ids
    // are global: row id = row_goff + row, stored column id = col_goff + column.
    const float* xa;
    const float* a_n32;
    const float* a_inorm32;
    int64_t row_goff, col_goff;
    // collect mode (second pass over the rows the first could not prove exact): the A rows are a gathered copy
    // [r1][dp] of those rows (r0 = 0), with their own norms, global ids (self exclusion) and FIXED per-row
    // thresholds -- every column whose fp32 key is inside the threshold is kept, nothing is compacted away
    const int* a_ids;
    const float* a_thr;
    // symmetric mode (whole-index builds with an eps that admits few pairs): only the column tiles at or above a row
    // block's own rows are computed -- half the MFMA work.  A unit is (row block, tile range, segment) taken from a
    // list sorted by length through an atomic cursor; a key d(i, j) computed above the diagonal also serves row j:
    // when it is inside j's static eps bound it is appended to j's transposed buffer (t_cap entries per row, counter
    // may exceed it: the row is then flagged), which the host compacts into one more segment of j's candidate lists.
    const int4* units = nullptr;   // (row block, first tile, end tile, segment)
    int nunits = 0;
    int* unit_ctr = nullptr;
    int* t_cnt = nullptr;          // [n]
    float* t_key = nullptr;        // [n][t_cap]
    int* t_idx = nullptr;
    int t_cap = 0;
    // Per-item thresholds: thr0[i] is an upper bound of the M-th smallest fp32 key of item i over ALL columns (the
    // M-th smallest over a sample of the columns is one), or +inf.  A row starts from min(eps bound, thr0) instead of
    // the eps bound alone (what that rejects is beyond the M-th smallest, like what a compaction drops), and the
    // transposed appends of the symmetric mode use the column item's.  out_thr: the row bounds a pass ends with.
    // Column tiles visited: tile index * tstride + tphase (a strided sample of the columns for the threshold pass).
    const float* thr0 = nullptr;
    const float* thr_col = nullptr;   // the column items' thresholds (== thr0 in a self build; a visiting block's on the ring)
    int t_all = 0;                    // block pairs: every tile is "above the diagonal" (rows and columns are different items)
    float* thr_pub = nullptr;   // == thr0 when the running bounds are published back during the symmetric main pass
    float* out_thr = nullptr;
    int tstride = 1, tphase = 0;
    // Gang order (bf16 kernel, symmetric mode): units[] is eight lists, one per XCD -- xoff[x] .. xoff[x + 1] -- and a
    // block takes its units from the list of the XCD it runs on (xcur[16 x]: that list's cursor), other lists once its
    // own is empty.  Consecutive entries of a list are GR row blocks x GC column pieces that share operands: run side by
    // side on one XCD they are served by its L2 instead of the fabric (as_build.hip, gang_plan).  Speed only: any block may run any unit.
    const int* xoff = nullptr;
    int* xcur = nullptr;
    // operand image geometry: ld = floats between consecutive rows of x32 / xa (dp for the bf16 image, dp8 / 2 for the int8
    // one), nslab = 128-byte slabs per row (dp / 32 or dp8 / 64); int8 image only: the rows' and columns' scales
    int64_t ld = 0;
    int nslab = 0;
    const float* fa = nullptr;     // [np] column items: s_j sqrt(128) / 16256
    const float* a_fa = nullptr;   // row side
};

__device__ __forceinline__ float ld_l2(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_l2(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Cross-lane hand-offs through LDS inside one wave: the hardware keeps a wave's LDS
// operations in order, the compiler only needs to be told that memory changed.
#define AS_CBAR() asm volatile("" ::: "memory")


// K2 on the bf16 matrix pipe (as_k2bf.hip).  ka.x32 / ka.xa point at the SPLIT images of the column / row items
// (as_space::xs: per row and 32-column slab 32 bf16 heads then 32 bf16 tails -- the bytes of the fp32 slab row).
as_status launch_k2_bf16(const KnnArgs& ka, int metric, bool collect, bool sym, int grid, hipStream_t st, bool i8 = false);
// fp32 items -> int8 two-digit image + per-row scales; maxima[0] = max_i s_i |theta_i|_2 / (16256 |x_i|), maxima[1] = max_i s_i
// |a2_i|_2 / (16256 |x_i|) (float bits, atomicMax: zero them first), maxima[2] != 0: a row with a non-finite value (unusable)
as_status quant_rows_i8(const float* x32, const float* n32, void* x8, float* fa8, int64_t rows, int64_t dp, int64_t dp8, unsigned int* maxima,
                        hipStream_t st);
// fp32 items [rows][dp] -> split image, same shape and stride
as_status split_rows_bf16(const float* x32, float* xs, int64_t rows, int64_t dp, hipStream_t st);
// true unless ARROWSPACE_K2_FP32=1 keeps the build on the fp32 matrix pipe (A/B runs, the bit-identity tests)
bool k2_bf16_enabled();

}  // namespace as
