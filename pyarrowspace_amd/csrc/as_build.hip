// Index build on gfx950: ingest -> fused X.X^T (fp32 MFMA) + streaming k-smallest ->
// fp64 refinement -> union symmetrisation (CSR) -> normalised Laplacian -> per-item
// spectral energy -> lambdas.  Replaces the crate call at /root/reference/src/lib.rs:289
// (`builder.build(rows)`); SPEC = DESIGN.md section 2.
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

#include "as_knn.hpp"

namespace as {

static inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static double err_coef_dp(int64_t dp) {
    // |fl32(n_i + n_j - 2 G32) - (exact)| <= coef * (n_i + n_j): k-ordered fp32 fma chain of
    // length dp (gamma_dp), rounded norms, two adds and the fp32 rounding of the inputs.
    const double u = 5.9604644775390625e-8;  // 2^-24
    if (!k2_bf16_enabled()) return (double)(dp + 16) * u;
    // bf16 head + tail products (as_k2bf.hip): G32 sums xh.yh + xh.yl + xl.yh -- 3 dp exact products, each addition
    // charged 2^-23 (a matrix core that truncates instead of rounding is covered) --, and what is dropped (xl.yl and
    // the two remainders, |r| <= 2^-16 |x|) is at most 3.03 * 2^-16 |x_k y_k| per term, hence <= 3.03 * 2^-16 |x||y|
    // in the sum (Cauchy-Schwarz) and, doubled in the key, <= 3.03 * 2^-16 (n_i + n_j).
    return (double)(6 * dp + 32) * u + 3.03 * 0x1p-16;
}

// int8 two-digit image (as_k2bf.hip, quant_i8_kernel): x = s (q + theta) / 16256 with |theta| <= 1/2 and q = 128 a1 + a2; the
// kernel forms s_i s_j (16384 a1.b1 + 128 (a1.b2 + a2.b1)) / 16256^2 exactly in int32.  With x~ = s q / 16256:
//   x_i.x_j - G = x_i.(x_j - x~_j) + (x_i - x~_i).x~_j + s_i s_j a2_i.a2_j / 16256^2,
//   |.| <= |x_i| s_j |theta_j|_2 / 16256 + |x~_j| s_i |theta_i|_2 / 16256 + s_i s_j |a2_i|_2 |a2_j|_2 / 16256^2   (Cauchy-Schwarz)
//       <= |x_i||x_j| (2.001 U + V^2),   U = max_i s_i |theta_i|_2 / (16256 |x_i|),  V = max_i s_i |a2_i|_2 / (16256 |x_i|)
// (the rows' ACTUAL residue and low-digit norms, measured by the quantisation kernel).  Doubled in the key and against
// n_i + n_j >= 2 |x_i||x_j| the bracket IS the coefficient; the epilogue's fp32 scaling (four roundings) and the rounded norms
// add 8 * 2^-24.  Clustered unit rows at D = 768: 3.6e-4 (bf16 head + tail: 3.2e-4).
// (12 fp32 roundings: the two integer sums' conversions -- exact below 2^24, i.e. up to D = 1 040 columns (127^2 D), one rounding each
// beyond --, their fused combination, the product of the two scales, the scaling, the scales' own two roundings each, two of slack)
static double err_coef_i8(double U, double V) { return 2.001 * U + V * V + 12.0 * 5.9604644775390625e-8; }

double err_coef(const as_space* sp) { return sp->ring_i8 ? sp->ring_coef8 : (sp->k2_i8 ? sp->coef8 : err_coef_dp(sp->dp)); }

// The operand of the k-NN kernels for this space's items: the fp32 matrix, or (default) its bf16 head + tail image,
// made on first use and kept with the space.
static as_status k2_items(const as_space* sp, const float** out) {
    if (!k2_bf16_enabled()) {
        *out = sp->x32;
        return AS_OK;
    }
    if (!sp->xs) {
        const int64_t rows_alloc = sp->np + ROW_TILE;
        float* xs = nullptr;
        AS_HIP(hipMalloc(&xs, sizeof(float) * rows_alloc * sp->dp));
        as_status s = split_rows_bf16(sp->x32, xs, rows_alloc, sp->dp, sp->stream);
        if (s == AS_OK && hipStreamSynchronize(sp->stream) != hipSuccess) {   // the kernels may run on another space's stream
            set_err("split_rows_bf16 failed: %s", hipGetErrorString(hipGetLastError()));
            s = AS_EHIP;
        }
        if (s != AS_OK) {
            (void)hipFree(xs);
            return s;
        }
#ifdef AS_ABLATION
        if (getenv("ARROWSPACE_K2_ZERO")) (void)hipMemset(xs, 0, sizeof(float) * rows_alloc * sp->dp);   // clock experiments: all-zero operands
#endif
        sp->xs = xs;
    }
    *out = sp->xs;
    return AS_OK;
}

static as_status k2_items_i8(const as_space* sp, bool* usable);
// The int8 two-digit image of this space's items and the error coefficient its products carry; made on first use.  False
// when the image cannot be used: a non-finite item, or a coefficient beyond 1e-3 (rows dominated by one element: s / |x| near
// 1; clustered unit rows measure 3.6e-4 at any D) -- the bf16 kernel then.
as_status space_i8_image(const as_space* sp, bool* present) {
    // every single-query and batched search asks: once the image (or the decision against it) stands, no lock is taken --
    // one process-wide mutex here serialised the re-entrant pool's threads and all spaces on each other
    if (sp->x8_ready.load(std::memory_order_acquire)) {
        *present = sp->x8 && !sp->x8_bad;
        return AS_OK;
    }
    std::lock_guard<std::mutex> lk(sp->imu);   // (searches of several host threads may all be the first to ask)
    bool usable = false;
    const as_status s = k2_items_i8(sp, &usable);
    if (s == AS_ENOMEM) {
        // no memory for the image: remembered -- every later search would otherwise retry a multi-GB hipMalloc under the lock
        // before it falls back to the fp32 items
        (void)hipGetLastError();
        sp->x8_bad = 2;
        sp->x8_ready.store(1, std::memory_order_release);
        *present = false;
        return AS_OK;
    }
    AS_TRY(s);
    sp->x8_ready.store(1, std::memory_order_release);
    *present = sp->x8 && !sp->x8_bad;
    return AS_OK;
}

// The high digits of the int8 image alone, in TILES: per 64 consecutive rows and 16-column chunk c one KiB holding row r's sixteen
// digits at byte 16 r ([tile][chunk][64 rows][16 bytes]; half the image's bytes) -- the operand of the single query's COARSE
// scan (as_scan.hip, scan_tile_kernel; as_search.hip, query_begin): x ~ s 128 a1 / 16256, off by at most V |x| (v8max).  One
// LDS-DMA of a scanning wave brings chunk c of 64 rows: lane r gets row r, and a row's dot never leaves its lane.
__global__ __launch_bounds__(256) void extract_a1_kernel(const signed char* __restrict__ x8, signed char* __restrict__ x8h, int64_t chunks, int64_t cpr) {
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < chunks; t += (int64_t)gridDim.x * 256) {
        const int64_t tile = t / (cpr * 64), rem = t - tile * (cpr * 64);
        const int64_t c = rem >> 6, row = tile * 64 + (rem & 63);   // c: 16-column chunk; 4 per slab of the two-digit image
        *(i32x4v*)(x8h + t * 16) = *(const i32x4v*)(x8 + (row * cpr * 2 + (c >> 2) * 8 + (c & 3)) * 16);
    }
}

as_status space_i8h_image(const as_space* sp, bool* present) {
    *present = false;
    bool have = false;
    AS_TRY(space_i8_image(sp, &have));
    if (!have) return AS_OK;
    const int rdy = sp->x8h_ready.load(std::memory_order_acquire);
    if (rdy) {
        *present = rdy == 1;
        return AS_OK;
    }
    std::lock_guard<std::mutex> lk(sp->imu);
    if (sp->x8h_ready.load(std::memory_order_relaxed) == 2) return AS_OK;
    if (!sp->x8h) {
        const int64_t rows_alloc = sp->np + ROW_TILE, cpr = sp->dp8 / 16;
        void* p = nullptr;
        if (hipMalloc(&p, (size_t)rows_alloc * sp->dp8) != hipSuccess) {
            (void)hipGetLastError();
            sp->x8h_ready.store(2, std::memory_order_release);
            return AS_OK;   // (no memory for it: the two-digit scan stays -- and the allocation is not tried again per query)
        }
        hipLaunchKernelGGL(extract_a1_kernel, dim3(256 * 8), dim3(256), 0, sp->stream, (const signed char*)sp->x8, (signed char*)p, rows_alloc * cpr, cpr);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(sp->stream) != hipSuccess) {
            (void)hipFree(p);
            set_err("extract_a1_kernel failed");
            return AS_EHIP;
        }
        sp->x8h = p;
    }
    sp->x8h_ready.store(1, std::memory_order_release);
    *present = true;
    return AS_OK;
}

static as_status k2_items_i8(const as_space* sp, bool* usable) {
    *usable = false;
    if (!sp->x8 && !sp->x8_bad) {
        const int64_t rows_alloc = sp->np + ROW_TILE;
        const int64_t dp8 = (sp->dp + 63) / 64 * 64;
        void* x8 = nullptr;
        float* fa8 = nullptr;
        dev_tmp<unsigned int> mx;
        AS_HIP(mx.alloc(4));
        AS_HIP(hipMemsetAsync(mx, 0, sizeof(unsigned int) * 4, sp->stream));
        AS_HIP(hipMalloc(&x8, (size_t)rows_alloc * dp8 * 2));
        if (hipMalloc(&fa8, sizeof(float) * rows_alloc) != hipSuccess) {
            (void)hipFree(x8);
            set_err("k2_items_i8: out of memory");
            return AS_ENOMEM;
        }
        as_status s = quant_rows_i8(sp->x32, sp->n32, x8, fa8, rows_alloc, sp->dp, dp8, mx, sp->stream);
        unsigned int h[4] = {0, 0, 0, 0};
        if (s == AS_OK && (hipMemcpyAsync(h, mx, sizeof(h), hipMemcpyDeviceToHost, sp->stream) != hipSuccess || hipStreamSynchronize(sp->stream) != hipSuccess)) {
            set_err("quant_rows_i8 failed: %s", hipGetErrorString(hipGetLastError()));
            s = AS_EHIP;
        }
        if (s != AS_OK) {
            (void)hipFree(x8);
            (void)hipFree(fa8);
            return s;
        }
        float U, V;
        memcpy(&U, &h[0], 4);
        memcpy(&V, &h[1], 4);
#ifdef AS_ABLATION
        if (getenv("ARROWSPACE_K2_ZERO")) (void)hipMemset(x8, 0, (size_t)rows_alloc * dp8 * 2);   // clock experiments: all-zero operands
#endif
        const_cast<as_space*>(sp)->dp8 = dp8;
        sp->x8 = x8;
        sp->fa8 = fa8;
        sp->coef8 = err_coef_i8(U, V);
        sp->u8max = U;
        sp->v8max = V;
        sp->x8_bad = h[2] ? 1 : 0;
        sp->x8_ready.store(1, std::memory_order_release);   // (space_i8_image's readers: everything above is published)
        dbg("k2_items_i8: U = %.3e, V = %.3e -> coefficient %.3e (bf16: %.3e)%s", U, V, sp->coef8, err_coef_dp(sp->dp),
            sp->x8_bad ? ", non-finite items: unusable" : "");
    }
    // (the bound that decides is absolute: what matters is e = coef (n_i + n_max) against the gaps between the rows' M-th and
    // k-th keys -- rows that fail their proof go to the band pass, the result is exact either way)
    *usable = sp->x8 && !sp->x8_bad && sp->coef8 <= 1.0e-3 && sp->dp <= 131072;
    return AS_OK;
}

// The operand of a ring pass (as_knn_block / _pair / _band: this rank's rows against a visiting block): the int8 images of both
// when the ring agreed on them (as_ring_i8_set: every rank's image usable, the coefficient from the ring-wide maxima of U and
// V -- a product of a row of one shard with a row of another is off by at most |x||y| (U_a + U_b + U_a U_b + V_a V_b)), else
// the bf16 head + tail images.  A visiting block's image is made from its raw rows here: the same digits its owner holds.
struct RingOperand {
    const float* a = nullptr;
    const float* b = nullptr;
    const float* fa = nullptr;   // rows' scales (int8) or null
    const float* fb = nullptr;
    int64_t ld = 0;
    int nslab = 0;
    bool i8 = false;
};
static as_status ring_operand(const as_space* sp, const as_space* cols, RingOperand* o) {
    if (sp->ring_i8) {
        bool ua = false, ub = false;
        AS_TRY(k2_items_i8(sp, &ua));
        AS_TRY(k2_items_i8(cols, &ub));
        // (the block's own maxima are at most the ring's: its owner reported them)
        if (sp->x8 && cols->x8 && !sp->x8_bad && !cols->x8_bad && cols->u8max <= sp->ring_u8 * 1.0000001 && cols->v8max <= sp->ring_v8 * 1.0000001 &&
            sp->u8max <= sp->ring_u8 * 1.0000001 && sp->v8max <= sp->ring_v8 * 1.0000001) {
            o->a = (const float*)sp->x8; o->b = (const float*)cols->x8; o->fa = sp->fa8; o->fb = cols->fa8;
            o->ld = sp->dp8 / 2; o->nslab = (int)(sp->dp8 / 64); o->i8 = true;
            return AS_OK;
        }
        set_err("ring pass: a block's int8 image does not match what the ring agreed on (as_ring_i8_set)");
        return AS_EINVAL;
    }
    AS_TRY(k2_items(sp, &o->a));
    AS_TRY(k2_items(cols, &o->b));
    o->ld = sp->dp; o->nslab = (int)(sp->dp / 32);
    return AS_OK;
}

// ------------------------------------------------------------------ K0 ingest
template <typename T>
__global__ void ingest_kernel(const T* __restrict__ src, int64_t ld, int64_t n, int64_t d, int64_t dp,
                              float* __restrict__ x32, double* __restrict__ n64, float* __restrict__ n32,
                              float* __restrict__ inorm32, int* lossless, unsigned long long* nmax_bits, unsigned long long* nmin_bits) {
    // a wave per row, rows strided over a resident grid; the largest / smallest norm and the lossless flag are kept
    // per wave and reach their global words ONCE per wave (a million same-address atomics cost 23 ms of a 25 ms ingest)
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x / 64);
    const int lane = lane_id();
    bool ok = true;
    unsigned long long wmax = 0ull, wmin = ~0ull;
    for (int64_t row = gw; row < n; row += nw) {
        double s = 0.0;
        for (int64_t c = lane; c < dp; c += 64) {
            if (c >= d) {
                x32[row * dp + c] = 0.0f;   // column padding: written here, so only the pad ROWS need a memset
                continue;
            }
            const double v = (double)src[row * ld + c];
            const float f = (float)v;
            x32[row * dp + c] = f;
            s += v * v;
            ok = ok && ((double)f == v || v != v);
        }
        s = wave_sum(s);
        if (lane == 0) {
            n64[row] = s;
            n32[row] = (float)s;
            inorm32[row] = s > 0.0 ? (float)(1.0 / sqrt(s)) : 0.0f;
        }
        // squared norms are >= 0: their bit patterns order like the values
        const unsigned long long b = (unsigned long long)__double_as_longlong(s);
        if (s == s) wmax = b > wmax ? b : wmax;
        if (s > 0.0 && s < 1.0e308) wmin = b < wmin ? b : wmin;
    }
    if (!__all(ok) && lane == 0) atomicExch(lossless, 0);
    if (lane == 0) {
        if (wmax != 0ull) atomicMax(nmax_bits, wmax);
        if (wmin != ~0ull) atomicMin(nmin_bits, wmin);
    }
}

template <typename T>
__global__ void copy_f64_kernel(const T* __restrict__ src, int64_t ld, int64_t n, int64_t d, double* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * d) return;
    dst[i] = (double)src[(i / d) * ld + (i % d)];
}

// The ingest in three steps, so that a host-resident input can stream through it chunk by chunk (ingest_host):
// ingest_begin allocates the space's arrays and the flag words, ingest_rows runs the kernel over a range of rows,
// ingest_end reads the flags back and decides about exact mode; ingest_keep_f64 says whether the fp64 items must be kept.
struct IngestState {
    int* flags = nullptr;
    unsigned long long *nmax_bits = nullptr, *nmin_bits = nullptr;
    int cus = 256;
    bool range_unsafe = false;
};

static as_status ingest_begin(as_space* sp, IngestState* st) {
    const int64_t n = sp->n, d = sp->d;
    sp->np = (n + ROW_TILE - 1) / ROW_TILE * ROW_TILE;
    sp->dp = (d + COL_PAD - 1) / COL_PAD * COL_PAD;
    const int64_t rows_alloc = sp->np + ROW_TILE;  // one extra zero tile: row blocks may start anywhere
    AS_HIP(hipMalloc(&sp->x32, sizeof(float) * rows_alloc * sp->dp));
    AS_HIP(hipMalloc(&sp->n64, sizeof(double) * n));
    AS_HIP(hipMalloc(&sp->n32, sizeof(float) * rows_alloc));
    AS_HIP(hipMalloc(&sp->inorm32, sizeof(float) * rows_alloc));
    AS_HIP(hipMalloc(&sp->lam64, sizeof(double) * n));
    AS_HIP(hipMalloc(&sp->lam32, sizeof(float) * rows_alloc));
    // the ingest kernel writes every element of the n item rows (column padding included): zero the pad rows only
    AS_HIP(hipMemsetAsync(sp->x32 + (size_t)n * sp->dp, 0, sizeof(float) * (size_t)(rows_alloc - n) * sp->dp, sp->stream));
    AS_HIP(hipMemsetAsync(sp->n32, 0, sizeof(float) * rows_alloc, sp->stream));
    AS_HIP(hipMemsetAsync(sp->inorm32, 0, sizeof(float) * rows_alloc, sp->stream));
    AS_HIP(hipMemsetAsync(sp->lam64, 0, sizeof(double) * n, sp->stream));
    AS_HIP(hipMemsetAsync(sp->lam32, 0, sizeof(float) * rows_alloc, sp->stream));
    AS_HIP(hipMalloc(&st->flags, 24));
    int hinit[6] = {1, 0, 0, 0, -1, 0x7fefffff};   // lossless, pad, nmax bits = 0, nmin bits = DBL_MAX
    AS_HIP(hipMemcpyAsync(st->flags, hinit, 24, hipMemcpyHostToDevice, sp->stream));
    st->nmax_bits = (unsigned long long*)(st->flags + 2);
    st->nmin_bits = (unsigned long long*)(st->flags + 4);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, sp->device) == hipSuccess) st->cus = prop.multiProcessorCount;
    return AS_OK;
}

// rows [row0, row0 + rows) of the space from `src` (device memory: row 0 of src is item row0, leading dimension ld)
static as_status ingest_rows(as_space* sp, const IngestState* st, const void* src, int dtype, int64_t ld, int64_t row0, int64_t rows, hipStream_t stream) {
    const int wpb = 4;
    const unsigned grid = (unsigned)std::min<int64_t>((rows + wpb - 1) / wpb, (int64_t)st->cus * 8);   // 32 waves per CU
    float* x32 = sp->x32 + (size_t)row0 * sp->dp;
    if (dtype == AS_DTYPE_F64)
        hipLaunchKernelGGL(ingest_kernel<double>, dim3(grid), dim3(64 * wpb), 0, stream, (const double*)src, ld, rows, sp->d, sp->dp, x32,
                           sp->n64 + row0, sp->n32 + row0, sp->inorm32 + row0, st->flags, st->nmax_bits, st->nmin_bits);
    else
        hipLaunchKernelGGL(ingest_kernel<float>, dim3(grid), dim3(64 * wpb), 0, stream, (const float*)src, ld, rows, sp->d, sp->dp, x32,
                           sp->n64 + row0, sp->n32 + row0, sp->inorm32 + row0, st->flags, st->nmax_bits, st->nmin_bits);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

static as_status ingest_end(as_space* sp, IngestState* st) {
    int hout[6];
    AS_HIP(hipMemcpyAsync(hout, st->flags, 24, hipMemcpyDeviceToHost, sp->stream));
    AS_HIP(hipStreamSynchronize(sp->stream));
    sp->lossless = hout[0];
    unsigned long long nb;
    memcpy(&nb, &hout[2], 8);
    long long nbs = (long long)nb;
    memcpy(&sp->nmax, &nbs, 8);
    AS_HIP(hipFree(st->flags));
    st->flags = nullptr;
    double nmin;
    memcpy(&nmin, &hout[4], 8);
    // The fp32 prefilters need every squared norm (and sums of two) inside the normal fp32 range: items scaled
    // by 1e20 overflow them, by 1e-22 flush them to zero, and a prefilter that sees inf/0 keys silently drops
    // true neighbours.  Outside a generous safe band the whole index runs in fp64 end to end (slow, exact).
    const double hi = 0x1p+120, lo = 0x1p-100;
    st->range_unsafe = (sp->nmax < 1.0e308 && sp->nmax > hi) || (nmin < lo);
    if (st->range_unsafe && !sp->opts.force_exact) {
        sp->opts.force_exact = 1;
        dbg("ingest: squared norms span [%.3g, %.3g], outside the fp32-safe range: fp64 end to end", nmin, sp->nmax);
    }
    return AS_OK;
}

static bool ingest_keep_f64(const as_space* sp, const IngestState* st, int dtype) {
    return (dtype == AS_DTYPE_F64 && !sp->lossless) || sp->opts.keep_f64 == AS_KEEP_F64_ALWAYS || st->range_unsafe;
}

as_status ingest(as_space* sp, const void* items_dev, int dtype, int64_t ld) {
    const int64_t n = sp->n, d = sp->d;
    IngestState st;
    AS_TRY(ingest_begin(sp, &st));
    AS_TRY(ingest_rows(sp, &st, items_dev, dtype, ld, 0, n, sp->stream));
    AS_TRY(ingest_end(sp, &st));
    const bool keep = ingest_keep_f64(sp, &st, dtype);
    if (keep) {
        AS_HIP(hipMalloc(&sp->x64, sizeof(double) * n * d));
        const int64_t tot = n * d;
        const unsigned g2 = (unsigned)((tot + 255) / 256);
        if (dtype == AS_DTYPE_F64)
            hipLaunchKernelGGL(copy_f64_kernel<double>, dim3(g2), dim3(256), 0, sp->stream, (const double*)items_dev, ld, n, d, sp->x64);
        else
            hipLaunchKernelGGL(copy_f64_kernel<float>, dim3(g2), dim3(256), 0, sp->stream, (const float*)items_dev, ld, n, d, sp->x64);
        AS_HIP(hipGetLastError());
        AS_HIP(hipStreamSynchronize(sp->stream));
    }
    dbg("ingest: n=%lld d=%lld np=%lld dp=%lld lossless_f32=%d keep_f64=%d nmax=%.6g", (long long)n, (long long)d,
        (long long)sp->np, (long long)sp->dp, sp->lossless, (int)keep, sp->nmax);
    return AS_OK;
}

// The reference's own call -- build(graph_params, items: float64 ndarray in HOST memory, any strides;
// /root/reference/src/lib.rs:271-277, src/helpers.rs:24-46) -- without a device copy of the whole fp64 matrix: the rows stream
// through TWO pinned chunks of about 64 MB and two device chunks of the same size; host threads pack chunk i + 1 (a memcpy per
// row, or an element gather for a non-unit column stride) while the copy engine moves chunk i and the ingest kernel converts
// chunk i - 1 -- host packing, PCIe and the kernel overlap, peak extra device memory 128 MB.  (Before: one hipMalloc of
// N D 8 bytes -- 6.1 GB at 1M x 768, 54 GB at 8.8M -- and one synchronous pageable copy in front of the ingest.)  Items that do
// not round-trip through fp32 are kept in fp64 as before: a second streaming pass straight into x64.
as_status ingest_host(as_space* sp, const double* items, int64_t row_stride, int64_t col_stride) {
    const int64_t n = sp->n, d = sp->d;
    IngestState st;
    AS_TRY(ingest_begin(sp, &st));
    const int64_t chunk_mb = getenv("ARROWSPACE_INGEST_CHUNK_MB") ? std::max(1, atoi(getenv("ARROWSPACE_INGEST_CHUNK_MB"))) : 64;   // (per build: tests stream small inputs in many chunks)
    const int64_t crows = std::max<int64_t>(1, std::min<int64_t>(n, (chunk_mb << 20) / (int64_t)(sizeof(double) * d)));
    const size_t cbytes = sizeof(double) * (size_t)crows * d;
    double* hbuf[2] = {nullptr, nullptr};
    double* dbuf[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    as_status rs = AS_OK;
    auto cleanup = [&]() {
        for (int i = 0; i < 2; ++i) {
            if (hbuf[i]) (void)hipHostFree(hbuf[i]);
            if (dbuf[i]) (void)hipFree(dbuf[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
        }
    };
    for (int i = 0; i < 2 && rs == AS_OK; ++i) {
        if (hipHostMalloc(&hbuf[i], cbytes, hipHostMallocDefault) != hipSuccess || hipMalloc(&dbuf[i], cbytes) != hipSuccess ||
            hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) {
            set_err("ingest_host: no memory for the staging chunks (%lld MB each)", (long long)(cbytes >> 20));
            rs = AS_ENOMEM;
        }
    }
    if (rs != AS_OK) {
        (void)hipGetLastError();
        cleanup();
        return rs;
    }
    static const int nthreads = [] {
        const char* e = getenv("ARROWSPACE_INGEST_THREADS");
        const int v = e ? atoi(e) : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        return std::max(1, std::min(v, 64));
    }();
    // rows [r0, r0 + rows) of the caller's array into a dense [rows][d] chunk, split over host threads
    auto pack = [&](double* dst, int64_t r0, int64_t rows) {
        auto part = [&](int64_t a, int64_t b) {
            if (col_stride == 1) {
                for (int64_t i = a; i < b; ++i) memcpy(dst + (size_t)(i - r0) * d, items + (r0 + (i - r0)) * row_stride, sizeof(double) * d);
            } else {
                for (int64_t i = a; i < b; ++i) {
                    const double* src = items + i * row_stride;
                    double* o = dst + (size_t)(i - r0) * d;
                    for (int64_t c = 0; c < d; ++c) o[c] = src[c * col_stride];
                }
            }
        };
        const int nt = (int)std::min<int64_t>(nthreads, std::max<int64_t>(1, rows / 256));
        if (nt <= 1) {
            part(r0, r0 + rows);
            return;
        }
        std::vector<std::thread> th;
        const int64_t per = (rows + nt - 1) / nt;
        for (int t = 1; t < nt; ++t) th.emplace_back(part, r0 + std::min<int64_t>(rows, t * per), r0 + std::min<int64_t>(rows, (t + 1) * per));
        part(r0, r0 + std::min<int64_t>(rows, per));
        for (auto& t : th) t.join();
    };
    // pass 0: every chunk through the ingest kernel; pass 1 (only when the fp64 items must be kept): every chunk into x64
    for (int pass = 0; pass < 2 && rs == AS_OK; ++pass) {
        if (pass == 1) {
            if ((rs = ingest_end(sp, &st)) != AS_OK) break;
            if (!ingest_keep_f64(sp, &st, AS_DTYPE_F64)) break;
            if (hipMalloc(&sp->x64, sizeof(double) * n * d) != hipSuccess) {
                (void)hipGetLastError();
                set_err("ingest_host: no memory to keep the fp64 items");
                rs = AS_ENOMEM;
                break;
            }
        }
        int64_t ci = 0;
        for (int64_t r0 = 0; r0 < n && rs == AS_OK; r0 += crows, ++ci) {
            const int b = (int)(ci & 1);
            const int64_t rows = std::min<int64_t>(crows, n - r0);
            if (ci >= 2 && hipEventSynchronize(done[b]) != hipSuccess) {   // the pinned chunk's previous copy (and the kernel behind it) has finished
                set_err("ingest_host: %s", hipGetErrorString(hipGetLastError()));
                rs = AS_EHIP;
                break;
            }
            pack(hbuf[b], r0, rows);
            double* dst = pass == 0 ? dbuf[b] : sp->x64 + (size_t)r0 * d;
            if (hipMemcpyAsync(dst, hbuf[b], sizeof(double) * (size_t)rows * d, hipMemcpyHostToDevice, sp->stream) != hipSuccess) {
                set_err("upload of items failed: %s", hipGetErrorString(hipGetLastError()));
                rs = AS_EHIP;
                break;
            }
            if (pass == 0) rs = ingest_rows(sp, &st, dbuf[b], AS_DTYPE_F64, d, r0, rows, sp->stream);
            if (rs == AS_OK && hipEventRecord(done[b], sp->stream) != hipSuccess) {
                set_err("ingest_host: %s", hipGetErrorString(hipGetLastError()));
                rs = AS_EHIP;
            }
        }
        if (rs == AS_OK && hipStreamSynchronize(sp->stream) != hipSuccess) {
            set_err("ingest_host: %s", hipGetErrorString(hipGetLastError()));
            rs = AS_EHIP;
        }
    }
    if (st.flags) (void)hipFree(st.flags);   // (an error before ingest_end)
    cleanup();
    if (rs != AS_OK) return rs;
    dbg("ingest (host, chunks of %lld rows, %d pack threads): n=%lld d=%lld np=%lld dp=%lld lossless_f32=%d keep_f64=%d nmax=%.6g",
        (long long)crows, nthreads, (long long)n, (long long)d, (long long)sp->np, (long long)sp->dp, sp->lossless, (int)(sp->x64 != nullptr), sp->nmax);
    return AS_OK;
}

// ------------------------------------------------------------------ K2 fused X.X^T + k-smallest
// Block = 4 waves, tile 256 rows x 128 cols, K-slab 32 (fp32 v_mfma_f32_32x32x2_f32).
// Wave w owns rows [64w, 64w+64) of the tile exclusively (2x4 accumulators of 32x32), so
// the candidate bookkeeping of a row never crosses waves.  Candidates that beat the row's
// running bound are appended to a per-row buffer in HBM scratch; a full buffer is compacted
// to its M smallest (key, idx) and the bound tightened.
// keep the M smallest (key, idx) of the row's cnt buffered candidates; wave-cooperative
__device__ __forceinline__ void compact_row(int rl, int M, float* bk, int* bi, float* ck, int* ci, int* s_cur,
                                            float* s_thr, int* s_drop, int tstride = 1, float* pub = nullptr) {
    const int lane = lane_id();
    AS_CBAR();
    const int cnt = s_cur[rl];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int t = lane; t < cnt; t += 64) {
        ck[t] = ld_l2(bk + t);
        ci[t] = ld_l2(bi + t);
    }
    AS_CBAR();
    for (int t = lane; t < cnt; t += 64) {
        const float k = ck[t];
        const int i = ci[t];
        int rank = 0;
        for (int s = 0; s < cnt; ++s) rank += lex_less<float>(ck[s], ci[s], k, i) ? 1 : 0;
        if (rank < M) {
            bk[rank] = k;
            bi[rank] = i;
            if (rank == M - 1) {
                s_thr[rl * tstride] = k;
                // symmetric mode: the row's new bound -- the M-th smallest key of a subset of its columns, an upper bound
                // of its M-th smallest over all of them -- is published for the blocks that hold this item as a column
                // (non-negative floats order like their bit patterns; a stale read is only less tight)
                if (pub) atomicMin((int*)pub, __float_as_int(fmaxf(k, 0.0f)));
            }
        }
    }
    if (lane == 0) {
        s_cur[rl] = M;
        s_drop[rl] = 1;
    }
    AS_CBAR();
}


#ifdef AS_ABLATION   // the A/B ladder of DESIGN.md 5.2 (tools/knn_variants.py); the product library holds one kernel per metric
constexpr int LROW = 36;   // padded LDS row of the register-staged kernels
// V bit0: XCD-grouped unit order (blocks that share an XCD's L2 work on the same few row
// blocks across the column segments); bit1: double-buffered LDS slabs (one barrier per slab);
// bit2: non-temporal loads for the streamed column operand.
template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void knn_mfma_kernel(KnnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NBUF = (V & 2) ? 2 : 1;
    constexpr int SLAB = (BM + BN) * LROW;  // floats per staged slab (A rows then B rows)
    float* As = (float*)smem;
    float* Bs = As + BM * LROW;
    float* s_thr = As + NBUF * SLAB;
    float* s_aux = s_thr + BM;
    int* s_cur = (int*)(s_aux + BM);
    int* s_drop = s_cur + BM;
    float* c_key = (float*)(s_drop + BM);
    int* c_idx = (int*)(c_key + 4 * CAP);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
    float* __restrict__ bkey = a.buf_key + (size_t)blockIdx.x * BM * CAP;
    int* __restrict__ bidx = a.buf_idx + (size_t)blockIdx.x * BM * CAP;
    float* ck = c_key + w * CAP;
    int* ci = c_idx + w * CAP;
    const int units = a.nrb * a.S;
    const int nslab = (int)(a.dp / BK);
    const float finf = __int_as_float(0x7f800000);
    const int srow = tid >> 3, sg = tid & 7;  // staging: 8 x 16-byte chunks per 128-byte row slab

    const int xslots = gridDim.x >> 3;          // blocks per XCD label (V&1: gridDim.x is a multiple of 8)
    const int rb_per_group = (V & 1) ? (xslots / a.S > 0 ? xslots / a.S : 1) : 1;
    for (int it = 0;; ++it) {
        int rb, cs;
        if (V & 1) {
            const int x = blockIdx.x & 7, sl = blockIdx.x >> 3;
            const int g = it * 8 + x;
            if (g * rb_per_group >= a.nrb) break;
            rb = g * rb_per_group + sl / a.S;
            cs = sl % a.S;
            if (sl >= rb_per_group * a.S || rb >= a.nrb) continue;
        } else {
            const int u = blockIdx.x + it * gridDim.x;
            if (u >= units) break;
            rb = u / a.S;
            cs = u % a.S;
        }
        const int64_t rowbase = a.r0 + (int64_t)rb * BM;
        const int t0 = (int)((int64_t)a.ntile * cs / a.S), t1 = (int)((int64_t)a.ntile * (cs + 1) / a.S);
        {
            const int64_t rg = rowbase + tid;
            const bool valid = rg < a.r1 && rg < a.n;
            const float ni = valid ? a.n32[rg] : 0.0f;
            const float bound = a.metric == AS_METRIC_L2 ? a.epskey + a.coef * (ni + a.nmax) : a.epskey + a.coef;
            s_thr[tid] = valid ? bound : -finf;
            s_aux[tid] = valid ? (a.metric == AS_METRIC_L2 ? ni : a.inorm32[rg]) : 0.0f;
            s_cur[tid] = 0;
            s_drop[tid] = 0;
        }
        __syncthreads();
        const float* pa = a.x32 + (size_t)(rowbase + srow) * a.dp + sg * 4;

        for (int ct = t0; ct < t1; ++ct) {
            const int64_t colbase = (int64_t)ct * BN;
            const float* pb = a.x32 + (size_t)(colbase + srow) * a.dp + sg * 4;
            f32x16 acc[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.0f;
            f32x4 ra[8], rbv[4];
            auto gload = [&](int ks) {
#pragma unroll
                for (int q = 0; q < 8; ++q) ra[q] = *(const f32x4*)(pa + (size_t)q * 32 * a.dp + ks * BK);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4* src = (const f32x4*)(pb + (size_t)q * 32 * a.dp + ks * BK);
                    rbv[q] = (V & 4) ? __builtin_nontemporal_load(src) : *src;
                }
            };
            auto lstore = [&](int buf) {
                float* Ad = As + buf * SLAB;
                float* Bd = Bs + buf * SLAB;
#pragma unroll
                for (int q = 0; q < 8; ++q) *(f32x4*)(Ad + (srow + 32 * q) * LROW + sg * 4) = ra[q];
#pragma unroll
                for (int q = 0; q < 4; ++q) *(f32x4*)(Bd + (srow + 32 * q) * LROW + sg * 4) = rbv[q];
            };
            auto compute = [&](int buf) {
                const float* Ar = As + buf * SLAB;
                const float* Br = Bs + buf * SLAB;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    f32x4 af[2], bf[4];
#pragma unroll
                    for (int m = 0; m < 2; ++m) af[m] = *(const f32x4*)(Ar + (w * 64 + m * 32 + l31) * LROW + s * 8 + h * 4);
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn) bf[nn] = *(const f32x4*)(Br + (nn * 32 + l31) * LROW + s * 8 + h * 4);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int nn = 0; nn < 4; ++nn)
                                acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m][t], bf[nn][t], acc[m][nn], 0, 0, 0);
                }
            };
            gload(0);
            lstore(0);
            __syncthreads();
            if (V & 2) {
                // double-buffered, branch-free body: slab ks+1 goes regs->LDS (other buffer) and slab
                // ks+2 global->regs in the shadow of slab ks's MFMAs (the tail iterations re-stage the
                // last slab, which nobody reads).
                gload(nslab > 1 ? 1 : 0);
                for (int ks = 0; ks < nslab; ++ks) {
                    const int cur = ks & 1;
                    lstore(cur ^ 1);
                    gload(ks + 2 < nslab ? ks + 2 : nslab - 1);
                    compute(cur);
                    __syncthreads();
                }
            } else if (V & 16) {
                // diagnostic: MFMA + fragment reads only (no staging after the first slab)
                for (int ks = 0; ks < nslab; ++ks) compute(0);
            } else {
                for (int ks = 0; ks < nslab; ++ks) {
                    const bool more = ks + 1 < nslab;
                    if (more) gload(ks + 1);
                    compute(0);
                    __syncthreads();
                    if (more) {
                        lstore(0);
                        __syncthreads();
                    }
                }
            }
            if (V & 8) {
                // diagnostic: no epilogue; keep the accumulators alive
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn) asm volatile("" ::"v"(acc[m][nn]));
                continue;
            }
            // ---- epilogue: keys, bound test, append
            {
                AS_CBAR();
                const int rl = w * 64 + lane;
                unsigned long long need = __ballot(s_cur[rl] > CAP - BN);
                while (need) {
                    const int r = __ffsll((long long)need) - 1;
                    const unsigned rr = w * 64 + r;
                    compact_row(rr, a.M, bkey + rr * CAP, bidx + rr * CAP, ck, ci, s_cur, s_thr, s_drop);
                    need &= need - 1;
                }
                AS_CBAR();
            }
            float nj[4];
            int cj[4];
            bool cv[4];
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) {
                const int64_t cg = colbase + nn * 32 + l31;
                cj[nn] = (int)cg;
                cv[nn] = cg < a.n;
                nj[nn] = a.metric == AS_METRIC_L2 ? a.n32[cg] : a.inorm32[cg];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = w * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float thr = s_thr[rl], ai = s_aux[rl];
                    const int rg = (int)(rowbase + rl);
                    float key[4];
                    bool p[4];
                    bool any = false;
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn) {
                        const float g = acc[m][nn][r];
                        key[nn] = a.metric == AS_METRIC_L2 ? fmaf(-2.0f, g, ai + nj[nn]) : 1.0f - fmaxf(0.0f, g * ai * nj[nn]);
                        p[nn] = cv[nn] && cj[nn] != rg && key[nn] <= thr;
                        any = any || p[nn];
                    }
                    if (__ballot(any)) {
#pragma unroll
                        for (int nn = 0; nn < 4; ++nn) {
                            const unsigned long long mk = __ballot(p[nn]);
                            if (!mk) continue;
                            const unsigned hm = h ? (unsigned)(mk >> 32) : (unsigned)mk;
                            const int base = s_cur[rl];
                            if (p[nn]) {
                                const unsigned slot = (unsigned)rl * CAP + base + __popc(hm & ((1u << l31) - 1u));
                                bkey[slot] = key[nn];
                                bidx[slot] = cj[nn];
                            }
                            // every lane of the half stores the same value: no cross-lane forwarding hazard
                            s_cur[rl] = base + __popc(hm);
                        }
                    }
                }
            }
        }
        // ---- finalize this unit's rows (each wave: its 64 rows)
        AS_CBAR();
        for (int r = 0; r < 64; ++r) {
            const unsigned rl = w * 64 + r;
            const int64_t rg = rowbase + rl;
            if (rg >= a.r1 || rg >= a.n) break;
            if (s_cur[rl] > a.M) compact_row(rl, a.M, bkey + rl * CAP, bidx + rl * CAP, ck, ci, s_cur, s_thr, s_drop);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int cnt = s_cur[rl];
            const size_t ob = ((size_t)(rg - a.r0) * a.S + cs) * a.M;
            for (int t = lane; t < cnt; t += 64) {
                a.out_key[ob + t] = ld_l2(bkey + rl * CAP + t);
                a.out_idx[ob + t] = ld_l2(bidx + rl * CAP + t);
            }
            if (lane == 0) a.out_cnt[(size_t)(rg - a.r0) * a.S + cs] = (int)((unsigned)cnt | ((unsigned)s_drop[rl] << 30));
        }
        __syncthreads();
    }
}

// ---- K2 (LDS-DMA form).  Same tile, same MFMA loop, same bookkeeping as knn_mfma_kernel; the
// slabs reach LDS by `global_load_lds_dwordx4` (no VGPR staging, no ds_write), double-buffered
// across slabs AND across column tiles, one barrier per slab.  LDS-DMA writes a wave-linear
// image (lane L -> base + 16 L), so rows cannot be padded; bank conflicts are avoided by an XOR
// swizzle applied to the per-lane SOURCE address and to the fragment reads: 16-byte chunk c of
// row r lives at chunk c ^ ((r >> 1) & 7) (any 16 rows distinct mod 16 then cover all 64 banks).

template <bool INTERLEAVE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void knn_mfma_dma_kernel(KnnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Sl = (float*)smem;                  // 2 slab buffers
    float2* s_ta = (float2*)(Sl + 2 * DSLAB);  // per row: (running bound, n_i or 1/|x_i|)
    int* s_cur = (int*)(s_ta + BM);
    int* s_drop = s_cur + BM;
    float* c_key = (float*)(s_drop + BM);
    int* c_idx = (int*)(c_key + 4 * CAP);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
    float* __restrict__ bkey = a.buf_key + (size_t)blockIdx.x * BM * CAP;
    int* __restrict__ bidx = a.buf_idx + (size_t)blockIdx.x * BM * CAP;
    float* ck = c_key + w * CAP;
    int* ci = c_idx + w * CAP;
    const int units = a.nrb * a.S;
    const int nslab = (int)(a.dp / BK);
    const float finf = __int_as_float(0x7f800000);

    // DMA source decomposition (see header comment): piece j of this wave covers rows
    // [8 j, 8 j + 8) of the wave's 64 A rows (32 B rows); lane L -> row + (L >> 3), stored chunk L & 7
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: SGPR address math, M0 by SALU
    const int drow = lane >> 3;
    const int csw0 = (lane & 7) ^ ((lane >> 4) & 7);            // source chunk for even pieces
    const int csw1 = (lane & 7) ^ ((4 + (lane >> 4)) & 7);      // source chunk for odd pieces
    // per-lane byte offsets inside a piece (32-bit: saddr + voffset form of the DMA)
    const unsigned lo0 = (unsigned)((drow * a.dp + csw0 * 4) * 4), lo1 = (unsigned)((drow * a.dp + csw1 * 4) * 4);
    // fragment read offsets (floats) inside a row, per k-group s: chunk (2 s + h) ^ ((l31 >> 1) & 7)
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = (((2 * s + h) ^ ((l31 >> 1) & 7)) << 2);

    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const int rb = u / a.S, cs = u % a.S;
        const int64_t rowbase = a.r0 + (int64_t)rb * BM;
        const int t0 = (int)((int64_t)a.ntile * cs / a.S), t1 = (int)((int64_t)a.ntile * (cs + 1) / a.S);
        {
            const int64_t rg = rowbase + tid;
            const bool valid = rg < a.r1 && rg < a.n;
            const float ni = valid ? a.n32[rg] : 0.0f;
            const float bound = a.metric == AS_METRIC_L2 ? a.epskey + a.coef * (ni + a.nmax) : a.epskey + a.coef;
            s_ta[tid] = make_float2(valid ? bound : -finf, valid ? (a.metric == AS_METRIC_L2 ? ni : a.inorm32[rg]) : 0.0f);
            s_cur[tid] = 0;
            s_drop[tid] = 0;
        }
        // uniform source bases (bytes): A rows of this wave, B rows of this wave's 32-row share
        const char* pa0 = (const char*)(a.x32 + (size_t)(rowbase + wu * 64) * a.dp);
        // one 1-KiB LDS-DMA piece: j < 8 -> A rows [8j, 8j+8) of this wave's 64, j >= 8 -> B rows of its 32
        auto dma_piece = [&](const char* srcA, const char* srcB, float* dst, int j) {
            if (j < 8) {
                const char* src = srcA + (size_t)(8 * j) * a.dp * 4 + ((j & 1) ? lo1 : lo0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + (wu * 64 + 8 * j) * DROW), 16, 0, 0);
            } else {
                const int jb = j - 8;
                const char* src = srcB + (size_t)(8 * jb) * a.dp * 4 + ((jb & 1) ? lo1 : lo0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + BM * DROW + (wu * 32 + 8 * jb) * DROW), 16, 0, 0);
            }
        };
        auto colptr = [&](int ct) { return (const char*)(a.x32 + (size_t)((int64_t)ct * BN + wu * 32) * a.dp); };
        {
            const char* pb0 = colptr(t0);
#pragma unroll
            for (int j = 0; j < 12; ++j) dma_piece(pa0, pb0, Sl, j);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int g = 0;
        for (int ct = t0; ct < t1; ++ct) {
            const int64_t colbase = (int64_t)ct * BN;
            f32x16 acc[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.0f;
            const char* pbc = colptr(ct);
            const char* pbn = colptr(ct + 1 < t1 ? ct + 1 : ct);
            // column norms of this tile: issued now, consumed by the epilogue (latency hidden by the K loop)
            float nj[4];
            int cj[4];
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) {
                const int64_t cg = colbase + nn * 32 + l31;
                cj[nn] = (int)cg;
                nj[nn] = a.metric == AS_METRIC_L2 ? a.n32[cg] : a.inorm32[cg];
            }
            for (int ks = 0; ks < nslab; ++ks, ++g) {
                const int cur = g & 1;
                // next slab: (ct, ks+1), or slab 0 of the next column tile; the very last slab of the unit
                // re-stages itself into the idle buffer (branch-free body, nobody reads it)
                const bool lastk = ks + 1 == nslab;
                const char* nxa = pa0 + (lastk ? 0 : ks + 1) * BK * 4;
                const char* nxb = (lastk ? pbn : pbc) + (lastk ? 0 : ks + 1) * BK * 4;
                float* nxd = Sl + (cur ^ 1) * DSLAB;
                if (!INTERLEAVE) {
#pragma unroll
                    for (int j = 0; j < 12; ++j) dma_piece(nxa, nxb, nxd, j);
                }
                const float* Ar = Sl + cur * DSLAB + (w * 64 + l31) * DROW;
                const float* Br = Sl + cur * DSLAB + BM * DROW + l31 * DROW;
                // fragments are software-pipelined by hand: the LDS-DMA intrinsic is an LDS write the
                // compiler will not move reads across, so group s+1's reads sit between two DMA pieces
                f32x4 af[2][2], bf[2][4];
#pragma unroll
                for (int m = 0; m < 2; ++m) af[0][m] = *(const f32x4*)(Ar + m * 32 * DROW + foff[0]);
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) bf[0][nn] = *(const f32x4*)(Br + nn * 32 * DROW + foff[0]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int nn = 0; nn < 4; ++nn) {
                                acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][m][t], bf[s & 1][nn][t], acc[m][nn], 0, 0, 0);
                                const int id = t * 8 + m * 4 + nn;
                                if (INTERLEAVE && id == 4) dma_piece(nxa, nxb, nxd, 3 * s);
                                if (INTERLEAVE && id == 12) dma_piece(nxa, nxb, nxd, 3 * s + 1);
                                if (id == 18 && s < 3) {
#pragma unroll
                                    for (int m2 = 0; m2 < 2; ++m2) af[(s + 1) & 1][m2] = *(const f32x4*)(Ar + m2 * 32 * DROW + foff[s + 1]);
#pragma unroll
                                    for (int n2 = 0; n2 < 4; ++n2) bf[(s + 1) & 1][n2] = *(const f32x4*)(Br + n2 * 32 * DROW + foff[s + 1]);
                                }
                                if (INTERLEAVE && id == 24) dma_piece(nxa, nxb, nxd, 3 * s + 2);
                            }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            // ---- epilogue: keys, bound test, append
            {
                AS_CBAR();
                const int rl = w * 64 + lane;
                unsigned long long need = __ballot(s_cur[rl] > CAP - BN);
                while (need) {
                    const int r = __ffsll((long long)need) - 1;
                    const unsigned rr = w * 64 + r;
                    compact_row(rr, a.M, bkey + rr * CAP, bidx + rr * CAP, ck, ci, s_cur, (float*)s_ta, s_drop, 2);
                    need &= need - 1;
                }
                AS_CBAR();
            }
            // tiles that touch the diagonal or the padded tail need the per-element exclusions
            const bool edge = colbase + BN > a.n || (colbase < rowbase + BM && colbase + BN > rowbase);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = w * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float2 ta = s_ta[rl];
                    const float thr = ta.x, ai = ta.y;
                    const int rg = (int)(rowbase + rl);
                    float key[4];
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn) {
                        const float gg = acc[m][nn][r];
                        key[nn] = a.metric == AS_METRIC_L2 ? fmaf(-2.0f, gg, ai + nj[nn]) : 1.0f - fmaxf(0.0f, gg * ai * nj[nn]);
                        if (edge && (cj[nn] >= a.n || cj[nn] == rg)) key[nn] = finf;
                    }
                    const float kmin = fminf(fminf(key[0], key[1]), fminf(key[2], key[3]));
                    if (__ballot(kmin <= thr)) {
#pragma unroll
                        for (int nn = 0; nn < 4; ++nn) {
                            const bool p = key[nn] <= thr;
                            const unsigned long long mk = __ballot(p);
                            if (!mk) continue;
                            const unsigned hm = h ? (unsigned)(mk >> 32) : (unsigned)mk;
                            const int base = s_cur[rl];
                            if (p) {
                                const unsigned slot = (unsigned)rl * CAP + base + __popc(hm & ((1u << l31) - 1u));
                                bkey[slot] = key[nn];
                                bidx[slot] = cj[nn];
                            }
                            s_cur[rl] = base + __popc(hm);
                        }
                    }
                }
            }
        }
        // ---- finalize this unit's rows (each wave: its 64 rows)
        AS_CBAR();
        for (int r = 0; r < 64; ++r) {
            const unsigned rl = w * 64 + r;
            const int64_t rg = rowbase + rl;
            if (rg >= a.r1 || rg >= a.n) break;
            if (s_cur[rl] > a.M) compact_row(rl, a.M, bkey + rl * CAP, bidx + rl * CAP, ck, ci, s_cur, (float*)s_ta, s_drop, 2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int cnt = s_cur[rl];
            const size_t ob = ((size_t)(rg - a.r0) * a.S + cs) * a.M;
            for (int t = lane; t < cnt; t += 64) {
                a.out_key[ob + t] = ld_l2(bkey + rl * CAP + t);
                a.out_idx[ob + t] = ld_l2(bidx + rl * CAP + t);
            }
            if (lane == 0) a.out_cnt[(size_t)(rg - a.r0) * a.S + cs] = (int)((unsigned)cnt | ((unsigned)s_drop[rl] << 30));
        }
        __syncthreads();
    }
}

#endif  // AS_ABLATION

// ---- K2 (LDS-DMA, 8 waves).  Same tile and LDS image as knn_mfma_dma_kernel, but 512 threads:
// two waves share every SIMD's matrix pipe (wave w owns rows [32w, 32w+32) as 1x4 accumulators),
// so one wave's DMA issue, fragment-read latency and epilogue run in the shadow of its
// partner's MFMAs.  Waves 4-7 issue their DMA pieces mid-slab, waves 0-3 at the slab start.
template <int METRIC, bool COLLECT = false, bool SYM = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void knn_mfma_dma8_kernel(KnnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Sl = (float*)smem;                  // 2 slab buffers
    float2* s_ta = (float2*)(Sl + 2 * DSLAB);  // per row: (running bound, n_i or 1/|x_i|)
    int* s_cur = (int*)(s_ta + BM);
    int* s_drop = s_cur + BM;
    float* c_key = (float*)(s_drop + BM);
    int* c_idx = (int*)(c_key + 8 * CAP);
    int* s_id = c_idx + 8 * CAP;               // collect mode: global item id of every A row

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* __restrict__ bkey = a.buf_key + (size_t)blockIdx.x * BM * CAP;
    int* __restrict__ bidx = a.buf_idx + (size_t)blockIdx.x * BM * CAP;
    float* ck = c_key + w * CAP;
    int* ci = c_idx + w * CAP;
    const int units = a.nrb * a.S;
    const int nslab = (int)(a.dp / BK);
    const float finf = __int_as_float(0x7f800000);
    const int drow = lane >> 3;
    const int csw0 = (lane & 7) ^ ((lane >> 4) & 7);
    const int csw1 = (lane & 7) ^ ((4 + (lane >> 4)) & 7);
    const unsigned lo0 = (unsigned)((drow * a.dp + csw0 * 4) * 4), lo1 = (unsigned)((drow * a.dp + csw1 * 4) * 4);
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = (((2 * s + h) ^ ((l31 >> 1) & 7)) << 2);
    const bool late = wu >= 4;   // wave-uniform: second-dispatched half issues its DMA mid-slab
    __shared__ int s_unit;

    for (int u = blockIdx.x;; u += gridDim.x) {
        int rb, cs, t0, t1;
        if (SYM) {
            // longest units first, handed out through an atomic cursor (the triangle's units differ in length)
            if (tid == 0) s_unit = atomicAdd(a.unit_ctr, 1);
            __syncthreads();   // the previous unit ended with a barrier: nobody still reads the old value
            u = s_unit;
            if (u >= a.nunits) break;
            const int4 ud = a.units[u];
            rb = ud.x; t0 = ud.y; t1 = ud.z; cs = ud.w;
        } else {
            if (u >= units) break;
            rb = u / a.S;
            cs = u % a.S;
            t0 = (int)((int64_t)a.ntile * cs / a.S);
            t1 = (int)((int64_t)a.ntile * (cs + 1) / a.S);
        }
        const int64_t rowbase = a.r0 + (int64_t)rb * BM;
        if (tid < BM) {
            const int64_t rg = rowbase + tid;
            if (COLLECT) {
                const bool valid = rg < a.r1;
                s_ta[tid] = make_float2(valid ? a.a_thr[rg] : -finf, valid ? (METRIC == AS_METRIC_L2 ? a.a_n32[rg] : a.a_inorm32[rg]) : 0.0f);
                s_id[tid] = valid ? a.a_ids[rg] : -1;
            } else {
                const bool valid = rg < a.r1;
                const float ni = valid ? a.a_n32[rg] : 0.0f;
                float bound = METRIC == AS_METRIC_L2 ? a.epskey + a.coef * (ni + a.nmax) : a.epskey + a.coef;
                int dropped = 0;
                if (a.thr0 && valid) {
                    const float t0r = ld_l2(a.thr0 + rg);   // possibly tightened by other units of this row since the threshold pass
                    if (t0r < bound) {   // what the tighter start rejects is beyond the row's M-th smallest key: a drop
                        bound = t0r;
                        dropped = 1;
                    }
                }
                s_ta[tid] = make_float2(valid ? bound : -finf, valid ? (METRIC == AS_METRIC_L2 ? ni : a.a_inorm32[rg]) : 0.0f);
                s_drop[tid] = dropped;
            }
            s_cur[tid] = 0;
            if (COLLECT) s_drop[tid] = 0;
        }
        const char* pa0 = (const char*)(a.xa + (size_t)(rowbase + wu * 32) * a.dp);
        // 6 pieces per wave and slab: j < 4 -> A rows [8j, 8j+8) of this wave's 32, j >= 4 -> B rows of its 16
        auto dma_piece = [&](const char* srcA, const char* srcB, float* dst, int j) {
            if (j < 4) {
                const char* src = srcA + (size_t)(8 * j) * a.dp * 4 + ((j & 1) ? lo1 : lo0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + (wu * 32 + 8 * j) * DROW), 16, 0, 0);
            } else {
                const int jb = j - 4;
                const char* src = srcB + (size_t)(8 * jb) * a.dp * 4 + ((jb & 1) ? lo1 : lo0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + BM * DROW + (wu * 16 + 8 * jb) * DROW), 16, 0, 0);
            }
        };
        auto colptr = [&](int ct) { return (const char*)(a.x32 + (size_t)(((int64_t)ct * a.tstride + a.tphase) * BN + wu * 16) * a.dp); };
        {
            const char* pb0 = colptr(t0);
#pragma unroll
            for (int j = 0; j < 6; ++j) dma_piece(pa0, pb0, Sl, j);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int g = 0;
        for (int ct = t0; ct < t1; ++ct) {
            const int64_t colbase = ((int64_t)ct * a.tstride + a.tphase) * BN;
            f32x16 acc[4];
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nn][r] = 0.0f;
            const char* pbc = colptr(ct);
            const char* pbn = colptr(ct + 1 < t1 ? ct + 1 : ct);
            float nj[4];
            int cj[4];
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) {
                const int64_t cg = colbase + nn * 32 + l31;
                cj[nn] = (int)cg;
                nj[nn] = METRIC == AS_METRIC_L2 ? a.n32[cg] : a.inorm32[cg];
            }
            for (int ks = 0; ks < nslab; ++ks, ++g) {
                const int cur = g & 1;
                const bool lastk = ks + 1 == nslab;
                const char* nxa = pa0 + (lastk ? 0 : ks + 1) * BK * 4;
                const char* nxb = (lastk ? pbn : pbc) + (lastk ? 0 : ks + 1) * BK * 4;
                float* nxd = Sl + (cur ^ 1) * DSLAB;
                if (!late) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) dma_piece(nxa, nxb, nxd, j);
                }
                const float* Ar = Sl + cur * DSLAB + (w * 32 + l31) * DROW;
                const float* Br = Sl + cur * DSLAB + BM * DROW + l31 * DROW;
                f32x4 af[2], bf[2][4];
                af[0] = *(const f32x4*)(Ar + foff[0]);
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) bf[0][nn] = *(const f32x4*)(Br + nn * 32 * DROW + foff[0]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (s < 3) {
                        af[(s + 1) & 1] = *(const f32x4*)(Ar + foff[s + 1]);
#pragma unroll
                        for (int nn = 0; nn < 4; ++nn) bf[(s + 1) & 1][nn] = *(const f32x4*)(Br + nn * 32 * DROW + foff[s + 1]);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int nn = 0; nn < 4; ++nn)
                            acc[nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][t], bf[s & 1][nn][t], acc[nn], 0, 0, 0);
                    if (s == 1 && late) {
#pragma unroll
                        for (int j = 0; j < 6; ++j) dma_piece(nxa, nxb, nxd, j);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            // ---- epilogue: keys, bound test, append (32 rows per wave)
            {
                AS_CBAR();
                const int rl = w * 32 + l31;
                unsigned long long need = __ballot(lane < 32 && s_cur[rl] > CAP - BN);
                if (COLLECT) {
                    // nothing may be dropped here: a row whose band does not fit stops collecting and is reported
                    if (lane < 32 && s_cur[rl] > CAP - BN) {
                        s_ta[rl].x = -finf;
                        s_drop[rl] = 2;
                    }
                    need = 0;
                }
                while (need) {
                    const int r = __ffsll((long long)need) - 1;
                    const unsigned rr = w * 32 + r;
                    compact_row(rr, a.M, bkey + rr * CAP, bidx + rr * CAP, ck, ci, s_cur, (float*)s_ta, s_drop, 2,
                                SYM && a.thr_pub ? a.thr_pub + (rowbase + rr) : nullptr);
                    need &= need - 1;
                }
                AS_CBAR();
            }
            const int64_t colg = a.col_goff + colbase, rowg = a.row_goff + rowbase;   // global ids of the tile's corner
            const bool edge = COLLECT || colbase + BN > a.n || (colg < rowg + BM && colg + BN > rowg);
            // symmetric mode: tiles strictly above the row block also serve the column items' rows (the diagonal tiles
            // hold both (i, j) and (j, i) themselves)
            const bool transp = SYM && a.t_cnt && (a.t_all || colbase >= rowbase + BM);
            float cb[4];
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) {
                cb[nn] = !transp || cj[nn] >= (int)a.n ? -finf : (METRIC == AS_METRIC_L2 ? a.epskey + a.coef * (nj[nn] + a.nmax) : a.epskey + a.coef);
                if (transp && a.thr_col && cj[nn] < (int)a.n) cb[nn] = fminf(cb[nn], ld_l2(a.thr_col + cj[nn]));   // live: past this XCD's L2
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = w * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float2 ta = s_ta[rl];
                const float thr = ta.x, ai = ta.y;
                const int rg = COLLECT ? s_id[rl] : (int)(rowg + rl);
                float key[4];
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) {
                    const float gg = acc[nn][r];
                    // both forms are symmetric in (i, j) bit for bit: a key computed once serves both rows
                    key[nn] = METRIC == AS_METRIC_L2 ? fmaf(-2.0f, gg, ai + nj[nn]) : 1.0f - fmaxf(0.0f, gg * (ai * nj[nn]));
                }
                if (transp) {
                    const bool rowok = rowbase + rl < a.r1;
                    bool tp[4];
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn) tp[nn] = rowok && key[nn] <= cb[nn];
                    if (__ballot(tp[0] || tp[1] || tp[2] || tp[3])) {
#pragma unroll
                        for (int nn = 0; nn < 4; ++nn)
                            if (tp[nn]) {
                                const int slot = atomicAdd(a.t_cnt + cj[nn], 1);
                                if (slot < a.t_cap) {
                                    a.t_key[(size_t)cj[nn] * a.t_cap + slot] = key[nn];
                                    a.t_idx[(size_t)cj[nn] * a.t_cap + slot] = rg;
                                }
                            }
                    }
                }
                if (edge) {  // wave-uniform: only tiles on the diagonal or at the padded tail pay for the exclusions
                    // collect mode: the band of a row may be unbounded (+inf: an item with an infinite norm makes
                    // every error bound infinite) -- an excluded entry must fail `key <= thr` even then: NaN
                    const float excl = COLLECT ? __int_as_float(0x7fc00000) : finf;
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn)
                        if (cj[nn] >= (int)a.n || cj[nn] + (int)a.col_goff == rg) key[nn] = excl;
                }
                const float kmin = fminf(fminf(key[0], key[1]), fminf(key[2], key[3]));
                if (__ballot(kmin <= thr)) {
#pragma unroll
                    for (int nn = 0; nn < 4; ++nn) {
                        const bool p = key[nn] <= thr;
                        const unsigned long long mk = __ballot(p);
                        if (!mk) continue;
                        const unsigned hm = h ? (unsigned)(mk >> 32) : (unsigned)mk;
                        const int base = s_cur[rl];
                        if (p) {
                            const unsigned slot = (unsigned)rl * CAP + base + __popc(hm & ((1u << l31) - 1u));
                            bkey[slot] = key[nn];
                            bidx[slot] = cj[nn] + (int)a.col_goff;
                        }
                        s_cur[rl] = base + __popc(hm);
                    }
                }
            }
        }
        // ---- finalize this unit's rows (each wave: its 32 rows)
        AS_CBAR();
        for (int r = 0; r < 32; ++r) {
            const unsigned rl = w * 32 + r;
            const int64_t rg = rowbase + rl;
            if (rg >= a.r1) break;
            if (!COLLECT && s_cur[rl] > a.M)
                compact_row(rl, a.M, bkey + rl * CAP, bidx + rl * CAP, ck, ci, s_cur, (float*)s_ta, s_drop, 2,
                            SYM && a.thr_pub ? a.thr_pub + rg : nullptr);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int cnt = s_cur[rl];
            const size_t ob = ((size_t)(rg - a.r0) * a.S + cs) * a.M;
            for (int t = lane; t < cnt; t += 64) {
                a.out_key[ob + t] = ld_l2(bkey + rl * CAP + t);
                a.out_idx[ob + t] = ld_l2(bidx + rl * CAP + t);
            }
            if (lane == 0) {
                a.out_cnt[(size_t)(rg - a.r0) * a.S + cs] = (int)((unsigned)cnt | ((unsigned)s_drop[rl] << 30));
                if (!COLLECT && a.out_thr) a.out_thr[rg] = s_ta[rl].x;   // M-th smallest key seen (or the start bound): threshold pass, S = 1
            }
        }
        __syncthreads();
    }
}

// The fused X.X^T + k-smallest kernel, every mode: bf16 head + tail products (default; ka.x32 / ka.xa are split images,
// k2_items) or the fp32 matrix pipe (ARROWSPACE_K2_FP32=1).
static as_status launch_k2(const KnnArgs& ka, int metric, bool collect, bool sym, int grid, hipStream_t st, bool i8 = false) {
    if (k2_bf16_enabled()) return launch_k2_bf16(ka, metric, collect, sym, grid, st, i8);
    const size_t lds8 = sizeof(float) * 2 * DSLAB + sizeof(float2) * BM + sizeof(int) * 3 * BM + (sizeof(float) + sizeof(int)) * 8 * CAP;
#define AS_K2(MM, CC, SS)                                                                                               \
    do {                                                                                                                \
        AS_HIP(hipFuncSetAttribute((const void*)knn_mfma_dma8_kernel<MM, CC, SS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8)); \
        hipLaunchKernelGGL((knn_mfma_dma8_kernel<MM, CC, SS>), dim3(grid), dim3(512), lds8, st, ka);                    \
    } while (0)
    if (metric == AS_METRIC_L2) {
        if (collect) AS_K2(AS_METRIC_L2, true, false);
        else if (sym) AS_K2(AS_METRIC_L2, false, true);
        else AS_K2(AS_METRIC_L2, false, false);
    } else {
        if (collect) AS_K2(AS_METRIC_COSINE, true, false);
        else if (sym) AS_K2(AS_METRIC_COSINE, false, true);
        else AS_K2(AS_METRIC_COSINE, false, false);
    }
#undef AS_K2
    AS_HIP(hipGetLastError());
    return AS_OK;
}

// ------------------------------------------------------------------ K2b fp64 refinement
// One wave per row: merge the S segment lists to the M smallest fp32 keys, evaluate those
// M pairs exactly in fp64, order by (key64, idx), apply eps and the k cap, and prove a
// posteriori that no dropped candidate could belong to the answer (else flag the row).
struct RefineArgs {
    const float* x32;
    const double* x64;
    const double* n64;
    int64_t n, d, dp;
    int64_t r0, r1;
    int S, M, metric;
    int64_t k;
    double epskey, coef, nmax;
    const float* c_key;
    const int* c_idx;
    const int* c_cnt;
    int32_t* out_idx;
    double* out_key;
    double* out_dist;
    double* out_gy;
    int32_t* out_cnt;
    int* flag;      // [(r1-r0)]
    int* nflag;     // counter
    double* out_B;  // [(r1-r0)]: upper bound of the k-th exact key of a flagged row (the band of the second pass)
};

__device__ __forceinline__ void exact_pair(const float* x32, const double* x64, int64_t d, int64_t dp, int64_t i,
                                           int64_t j, double& sq, double& dot) {
    const int lane = lane_id();
    double s = 0.0, g = 0.0;
    if (x64) {
        const double* pi = x64 + i * d;
        const double* pj = x64 + j * d;
        for (int64_t c = lane; c < d; c += 64) {
            const double a = pi[c], b = pj[c], t = a - b;
            s += t * t;
            g += a * b;
        }
    } else {
        const float* pi = x32 + i * dp;
        const float* pj = x32 + j * dp;
        for (int64_t c = lane; c < d; c += 64) {
            const double a = (double)pi[c], b = (double)pj[c], t = a - b;
            s += t * t;
            g += a * b;
        }
    }
    sq = wave_sum(s);
    dot = wave_sum(g);
}

__global__ __launch_bounds__(256) void knn_refine_kernel(RefineArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = threadIdx.x >> 6, lane = lane_id();
    const int SM = a.S * a.M;
    // per-wave carve: ek[M] eg[M] ed[M] sk[M] (double) | ck[SM] (float) | ci[SM] li[M] (int) | lk[M] (float)
    const size_t per_wave = sizeof(double) * 4 * a.M + sizeof(float) * (SM + a.M) + sizeof(int) * (SM + a.M);
    char* base = smem + (size_t)w * ((per_wave + 15) / 16 * 16);
    double* ek = (double*)base;
    double* eg = ek + a.M;
    double* ed = eg + a.M;
    double* sk = ed + a.M;
    float* ck = (float*)(sk + a.M);
    float* lk = ck + SM;
    int* ci = (int*)(lk + a.M);
    int* li = ci + SM;

    const int64_t row = a.r0 + (int64_t)blockIdx.x * 4 + w;
    if (row >= a.r1) return;
    const int64_t lr = row - a.r0;
    int C = 0, anyfull = 0, lost = 0;
    for (int cs = 0; cs < a.S; ++cs) {
        const int cc = a.c_cnt[lr * a.S + cs];
        const int c = cc & 0xffff;
        anyfull |= (cc >> 30) & 1;
        lost |= (cc >> 31) & 1;   // a segment that lost candidates it cannot bound (transposed buffer overflow)
        const size_t ob = ((size_t)lr * a.S + cs) * a.M;
        for (int t = lane; t < c; t += 64) {
            ck[C + t] = a.c_key[ob + t];
            ci[C + t] = a.c_idx[ob + t];
        }
        C += c;
    }
    if (C > a.M) anyfull = 1;
    const int Mp = C < a.M ? C : a.M;
    for (int t = lane; t < C; t += 64) {
        const float k = ck[t];
        const int i = ci[t];
        int rank = 0;
        for (int s = 0; s < C; ++s) rank += lex_less<float>(ck[s], ci[s], k, i) ? 1 : 0;
        if (rank < a.M) {
            lk[rank] = k;
            li[rank] = i;
        }
    }
    const double ni = a.n64[row];
    for (int t = 0; t < Mp; ++t) {
        const int j = li[t];
        double sq, dot;
        exact_pair(a.x32, a.x64, a.d, a.dp, row, j, sq, dot);
        if (lane == 0) {
            if (a.metric == AS_METRIC_L2) {
                ek[t] = sq;
                ed[t] = sqrt(sq);
                eg[t] = dot;
            } else {
                const double den = sqrt(ni * a.n64[j]);
                const double c = den > 0.0 ? dot / den : 0.0;
                const double dd = cosine_distance(c);
                ek[t] = dd;
                ed[t] = dd;
                eg[t] = c;
            }
        }
    }
    // order by (key64, idx); entries passing eps form a prefix of that order
    int npass_l = 0;
    for (int t = lane; t < Mp; t += 64) {
        const double k = ek[t];
        const int i = li[t];
        int rank = 0;
        for (int s = 0; s < Mp; ++s) rank += lex_less<double>(ek[s], li[s], k, i) ? 1 : 0;
        sk[rank] = k;
        if (k <= a.epskey) {
            npass_l += 1;
            if (rank < a.k) {
                a.out_idx[lr * a.k + rank] = i;
                a.out_key[lr * a.k + rank] = k;
                a.out_dist[lr * a.k + rank] = ed[t];
                a.out_gy[lr * a.k + rank] = eg[t];
            }
        }
    }
    const int npass = wave_sum(npass_l);
    const int cnt = npass < a.k ? npass : (int)a.k;
    for (int64_t t = cnt + lane; t < a.k; t += 64) a.out_idx[lr * a.k + t] = -1;
    if (lane == 0) {
        a.out_cnt[lr] = cnt;
        int bad = 0;
        if (anyfull) {
            const double B = npass >= a.k ? sk[a.k - 1] : a.epskey;
            const double e = a.metric == AS_METRIC_L2 ? a.coef * (ni + a.nmax) : a.coef;
            const double T32 = (double)lk[Mp - 1];
            bad = !(T32 - e > B);
        }
        if (lost) bad = 1;
        a.flag[lr] = bad;
        if (bad) {
            atomicAdd(a.nflag, 1);
            a.out_B[lr] = npass >= a.k ? sk[a.k - 1] : a.epskey;
        }
    }
}

// ------------------------------------------------------------------ K2c second pass: complete band collection for rows the
// first pass could not prove exact (ties / near-ties at the k-th distance: duplicate items, dense blobs).
// A flagged row i has an upper bound B_i of its k-th exact key (from the first refinement); every true neighbour has
// key64 <= B_i, hence key32 <= B_i + e_i.  The MFMA kernel runs again over a gathered copy of just those rows in
// collect mode and keeps EVERY column inside that fixed band; knn_band_refine_kernel evaluates them all in fp64 and
// takes the k smallest by (key64, index) -- exact by construction, no proof needed.  Only rows whose band does not
// fit the collection buffers (thousands of ties) are left to the row-serial fp64 path.
__global__ void band_gather_kernel(const float* __restrict__ x32, const float* __restrict__ n32, const float* __restrict__ inorm32,
                                   const double* __restrict__ n64, int64_t dp, int64_t r0, const int* __restrict__ ids, int nf,
                                   const double* __restrict__ B, int metric, double coef, double nmax, float* __restrict__ xa,
                                   float* __restrict__ a_n32, float* __restrict__ a_inorm32, int* __restrict__ a_ids,
                                   float* __restrict__ a_thr, int64_t goff, const float* __restrict__ fa = nullptr,
                                   float* __restrict__ a_fa = nullptr) {   // (dp: floats per row of the operand IMAGE x32 points at)
    const int f = blockIdx.x;
    if (f >= nf) return;
    const int lr = ids[f];
    const int64_t row = r0 + lr;
    for (int64_t c = threadIdx.x * 4; c < dp; c += blockDim.x * 4) *(f32x4*)(xa + (size_t)f * dp + c) = *(const f32x4*)(x32 + row * dp + c);
    if (threadIdx.x == 0) {
        a_n32[f] = n32[row];
        a_inorm32[f] = inorm32[row];
        if (fa) a_fa[f] = fa[row];
        a_ids[f] = (int)(goff + row);   // global item id (self exclusion against global column ids)
        const double e = metric == AS_METRIC_L2 ? coef * (n64[row] + nmax) : coef;
        // rounded up twice over: the device-side comparison must never be tighter than the fp64 band
        a_thr[f] = __double2float_ru((B[lr] + e) * 1.000001);
    }
}

struct BandArgs {
    const float* x32;
    const double* x64;
    const double* n64;
    int64_t d, dp, r0;
    int S, CW, metric, nf;
    int64_t k;          // output stride and cap
    double epskey;
    const int* ids;     // [nf] local row of every flagged row
    const float* c_key; // [nf][S][CW]
    const int* c_idx;
    const int* c_cnt;   // [nf][S]
    int32_t* out_idx;
    double* out_key;
    double* out_dist;
    double* out_gy;
    int32_t* out_cnt;
    int* flag;          // cleared for rows settled here
};

// one block (4 waves) per flagged row; up to BAND_MAX candidates in LDS.  A row whose band holds more (or whose
// collection overflowed a segment buffer) stays flagged.
constexpr int BAND_MAX = 4096;

__global__ __launch_bounds__(256) void knn_band_refine_kernel(BandArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* ek = (double*)smem;           // exact key
    double* ed = ek + BAND_MAX;           // distance
    double* eg = ed + BAND_MAX;           // gy
    int* ci = (int*)(eg + BAND_MAX);
    __shared__ int s_C, s_over, s_pass;
    const int w = threadIdx.x >> 6, lane = lane_id();
    const int f = blockIdx.x;
    const int lr = a.ids[f];
    const int64_t row = a.r0 + lr;
    if (threadIdx.x == 0) {
        // segment offsets: a serial scan over S counts (S is at most a few hundred)
        int C = 0, over = 0;
        for (int cs = 0; cs < a.S; ++cs) {
            const int cc = a.c_cnt[(size_t)f * a.S + cs];
            over |= (cc >> 31) & 1;
            C += cc & 0xffff;
        }
        s_C = C;
        s_over = over || C > BAND_MAX;
        s_pass = 0;
    }
    __syncthreads();
    if (s_over) return;   // the band did not fit: the row stays flagged
    const int C = s_C;
    {
        int off = 0;
        for (int cs = 0; cs < a.S; ++cs) {
            const int c = a.c_cnt[(size_t)f * a.S + cs] & 0xffff;
            const size_t ob = ((size_t)f * a.S + cs) * a.CW;
            for (int t = threadIdx.x; t < c; t += blockDim.x) ci[off + t] = a.c_idx[ob + t];
            off += c;
        }
    }
    __syncthreads();
    const double ni = a.n64[row];
    for (int t = w; t < C; t += 4) {
        const int j = ci[t];
        double sq, dot;
        exact_pair(a.x32, a.x64, a.d, a.dp, row, j, sq, dot);
        if (lane == 0) {
            if (a.metric == AS_METRIC_L2) {
                ek[t] = sq;
                ed[t] = sqrt(sq);
                eg[t] = dot;
            } else {
                const double den = sqrt(ni * a.n64[j]);
                const double c = den > 0.0 ? dot / den : 0.0;
                const double dd = cosine_distance(c);
                ek[t] = dd;
                ed[t] = dd;
                eg[t] = c;
            }
        }
    }
    __syncthreads();
    int npass_l = 0;
    for (int t = threadIdx.x; t < C; t += blockDim.x) {
        const double kk = ek[t];
        if (!(kk <= a.epskey)) continue;
        npass_l += 1;
        const int i = ci[t];
        int rank = 0;
        for (int s2 = 0; s2 < C; ++s2) rank += lex_less<double>(ek[s2], ci[s2], kk, i) ? 1 : 0;
        if (rank < a.k) {
            a.out_idx[(size_t)lr * a.k + rank] = i;
            a.out_key[(size_t)lr * a.k + rank] = kk;
            a.out_dist[(size_t)lr * a.k + rank] = ed[t];
            a.out_gy[(size_t)lr * a.k + rank] = eg[t];
        }
    }
    if (npass_l) atomicAdd(&s_pass, npass_l);
    __syncthreads();
    const int npass = s_pass;
    const int cnt = npass < a.k ? npass : (int)a.k;
    for (int64_t t = cnt + threadIdx.x; t < a.k; t += blockDim.x) a.out_idx[(size_t)lr * a.k + t] = -1;
    if (threadIdx.x == 0) {
        a.out_cnt[lr] = cnt;
        a.flag[lr] = 0;
    }
}

// ---- symmetric mode helpers
// threshold pass, S = 1: rows whose candidate count in the sampled columns exceeds `limit` (or that compacted: >= M)
__global__ void sym_risk_kernel(const int* __restrict__ c_cnt, int64_t rows, int limit, int* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cc = i < rows ? c_cnt[i] : 0;
    const bool risky = (cc & 0xffff) > limit || ((cc >> 30) & 1);
    const unsigned long long m = __ballot(risky);
    if (lane_id() == 0 && m) atomicAdd(out, __popcll(m));
}

// A row's transposed buffer -> the M smallest by (key32, id) as one more segment of its candidate lists; bit 30 of the
// count: something was dropped (all of it beyond the M-th kept key); bit 31: the buffer overflowed (unbounded loss).
// gate / out_bound (block pairs, ring): gate[row] is the threshold the appends were admitted with, +inf where it was the
// eps bound alone; out_bound[row] is then a lower bound of every key the buffer did NOT keep -- the M-th kept key after
// a compaction, else the gate -- and bit 30 says that such keys may exist; an overflowed buffer gets -inf.
__global__ __launch_bounds__(256) void transposed_compact_kernel(const int* __restrict__ t_cnt, const float* __restrict__ t_key,
                                                                 const int* __restrict__ t_idx, int t_cap, int64_t rows, int S, int seg,
                                                                 int M, float* __restrict__ c_key, int* __restrict__ c_idx,
                                                                 int* __restrict__ c_cnt, const float* __restrict__ gate = nullptr,
                                                                 float* __restrict__ out_bound = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = threadIdx.x >> 6, lane = lane_id();
    float* ck = (float*)smem + (size_t)w * t_cap;
    int* ci = (int*)((float*)smem + (size_t)4 * t_cap) + (size_t)w * t_cap;
    const int64_t row = (int64_t)blockIdx.x * 4 + w;
    if (row >= rows) return;
    const int raw = t_cnt[row];
    const int cnt = raw < t_cap ? raw : t_cap;
    for (int t = lane; t < cnt; t += 64) {
        ck[t] = t_key[(size_t)row * t_cap + t];
        ci[t] = t_idx[(size_t)row * t_cap + t];
    }
    AS_LDS_FENCE();
    const size_t ob = ((size_t)row * S + seg) * M;
    for (int t = lane; t < cnt; t += 64) {
        const float k = ck[t];
        const int i = ci[t];
        int rank = 0;
        for (int s2 = 0; s2 < cnt; ++s2) rank += lex_less<float>(ck[s2], ci[s2], k, i) ? 1 : 0;
        if (rank < M) {
            c_key[ob + rank] = k;
            c_idx[ob + rank] = i;
        }
    }
    float kth = 0.0f;   // the M-th kept key: the rank loop above wrote it
    if (out_bound && cnt > M) {
        for (int t = lane; t < cnt; t += 64) {
            int rank = 0;
            for (int s2 = 0; s2 < cnt; ++s2) rank += lex_less<float>(ck[s2], ci[s2], ck[t], ci[t]) ? 1 : 0;
            if (rank == M - 1) kth = ck[t];
        }
        kth = wave_sum(kth);   // exactly one lane holds it
    }
    if (lane == 0) {
        const unsigned kept = (unsigned)(cnt < M ? cnt : M);
        const float g = gate ? gate[row] : __int_as_float(0x7f800000);
        const bool gated = g < __int_as_float(0x7f800000);
        c_cnt[(size_t)row * S + seg] = (int)(kept | ((cnt > M || (out_bound && gated)) ? 1u << 30 : 0u) | (raw > t_cap ? 1u << 31 : 0u));
        if (out_bound) out_bound[row] = raw > t_cap ? -__int_as_float(0x7f800000) : (cnt > M ? kth : g);
    }
}

static int pick_list_width(int64_t k) {
    // M = k + margin rounded up to a power of two in [32, 64]; wider lists are not supported yet
    const int64_t need = k + 8;
    if (need <= 32) return 32;
    if (need <= 64) return 64;
    if (need <= 128) return 128;
    return -1;
}
int knn_list_width(int64_t k) { return pick_list_width(k); }

// Gang order of a symmetric pass's units (bf16 kernel; KnnArgs::xoff; opt-in: ARROWSPACE_K2_GANG_GC = column pieces per
// gang).  Every block streams 48 KB per tile and slab from beyond its XCD's L2 (17 % hits in the plain order) -- so the
// units are dealt to the eight XCDs in GANGS: GR consecutive row blocks x GC column pieces (GR GC <= 32, the blocks of an
// XCD): the XCD's blocks then work on 32 / GC row blocks, whose rows (768 KB each at 768 columns) stay in its 4 MB L2 from
// tile to tile.  Pieces are aligned (tile0 + multiples of Lp) so that the units of a gang walk the same tiles.  Gangs go
// to the XCD with the least work so far, longest first; xoff[0..8] are the lists' bounds.  Measured:
// L2 hits 17 -> 48 %, fabric fetch -35 %, the memory side alone 0.111 -> 0.079 s at 262144 x 768 -- and the kernel no
// faster: on random operands it is held by the chip's power management, not by its memory side.  Kept for shapes and
// chips where that changes; off by default (GC + 1 candidate segments per row cost the refinement 0.05 -> 0.15 s).
static int gang_pieces() {   // GC: column pieces of a gang (and of a row: the gang spans all columns)
    const char* e = getenv("ARROWSPACE_K2_GANG_GC");
    const int gc = e ? atoi(e) : 16;
    return std::max(1, std::min(gc, 16));
}
static void gang_plan(std::vector<int4>& units, int Lp, int tile0, int xoff[9]) {
    int npieces = 1;
    for (const int4& u : units) npieces = std::max(npieces, (u.y - tile0) / Lp + 1);
    const int GC = std::min(gang_pieces(), npieces), GR = std::max(1, 32 / GC);
    const int ngc = (npieces + GC - 1) / GC;
    struct Gang { long long key; long long work; std::vector<int4> us; };
    std::vector<Gang> gangs;
    std::vector<std::pair<long long, int>> order(units.size());
    for (size_t i = 0; i < units.size(); ++i)
        order[i] = std::make_pair((long long)(units[i].x / GR) * ngc + ((units[i].y - tile0) / Lp) / GC, (int)i);
    std::sort(order.begin(), order.end());
    for (size_t i = 0; i < order.size(); ++i) {
        if (gangs.empty() || gangs.back().key != order[i].first) gangs.push_back(Gang{order[i].first, 0, {}});
        const int4& u = units[order[i].second];
        gangs.back().us.push_back(u);
        gangs.back().work += u.z - u.y;
    }
    std::stable_sort(gangs.begin(), gangs.end(), [](const Gang& x, const Gang& y) { return x.work > y.work; });
    std::vector<std::vector<int4>> lists(8);
    long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const Gang& g : gangs) {
        int best = 0;
        for (int x = 1; x < 8; ++x)
            if (load[x] < load[best]) best = x;
        load[best] += g.work;
        lists[best].insert(lists[best].end(), g.us.begin(), g.us.end());
    }
    units.clear();
    for (int x = 0; x < 8; ++x) {
        xoff[x] = (int)units.size();
        units.insert(units.end(), lists[x].begin(), lists[x].end());
    }
    xoff[8] = (int)units.size();
}

// First pass of the k-NN stage for the rows [r0, r1) of a space against all of its items: the candidate lists
// c.ckey / c.cidx / c.ccnt ([rows][c.S][M], ids global: column + col_goff) that a refinement turns into exact lists.
// A whole-space pass runs in symmetric mode (threshold pass, upper-triangle tiles, transposed buffers as one more
// segment); row ranges run the full pass.  Shared by as_knn_rows and the ring's own-block step.
struct KnnCand {
    dev_tmp<float> bkey, ckey;
    dev_tmp<int> bidx, cidx, ccnt;
    KnnArgs ka;
    int S = 1, grid = 0, dev_cus = 256, ntile = 0, nrb = 0, units = 0;
    bool sym = false, i8 = false;
    double flops = 0, t_mfma = 0;
};

static as_status knn_candidates(const as_space* sp, const as_graph_params* gp, int64_t r0, int64_t r1, int M, int64_t row_goff,
                                int64_t col_goff, KnnCand& c) {
    const int64_t n = sp->n, rows = r1 - r0;
    hipStream_t st = sp->stream;
    const int metric = sp->opts.metric;
    const double epskey = metric == AS_METRIC_L2 ? gp->eps * gp->eps : gp->eps;
    const double coef = err_coef(sp);
    const int nrb = (int)((rows + BM - 1) / BM);
    const int ntile = (int)(sp->np / BN);
    // default: 8-wave LDS-DMA kernel (48); 32 = 4-wave LDS-DMA; 0..31 = register-staged kernel and its A/B variants
    int variant = 48;
    if (const char* ev = getenv("ARROWSPACE_KNN_VARIANT")) variant = atoi(ev) & 63;
    int S = 1;
    {   // enough units to fill the chip several times over, but never thinner than 8 column tiles
        int dev_cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, sp->device) == hipSuccess) dev_cus = prop.multiProcessorCount;
        // an eps that admits (nearly) every pair -- `eps: 10` of tests/test_3_beir.py under the cosine distance -- makes
        // every column segment converge its rows' bounds from scratch (all tiles pass until the lists have settled):
        // fewer, longer segments then (61 -> TF/s at 200k x 768 with 8 segments)
        const bool loose = metric == AS_METRIC_COSINE ? gp->eps >= 1.0 : gp->eps * gp->eps >= 4.0 * sp->nmax;
        const int target_units = dev_cus * (loose ? 2 : 16);
        while (S < 8 && nrb * S < target_units && ntile / (S * 2) >= 8) S *= 2;
        if ((variant & 1) && !(variant & 32) && ntile >= 64) S = 8;  // XCD-grouped order: 4 row blocks x 8 column segments per XCD
    }
    int dev_cus = 256;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, sp->device) == hipSuccess) dev_cus = prop.multiProcessorCount;
    }
    // ---- symmetric mode: a whole-index build whose eps admits few pairs computes only the tiles at or above each
    // row block (half the MFMA work); the keys above the diagonal reach the column items' rows through
    // transposed buffers.  Decided from a sample of the pair distribution: the buffers must hold what eps admits.
    // (from 16 column tiles on: below that the threshold pass costs what the triangle saves)
    bool sym = r0 == 0 && r1 == n && (variant & 48) == 48 && ntile >= 16 && !getenv("ARROWSPACE_NO_SYM");
    // The threshold pass visits every tstride-th column tile; a row's threshold is then about its (tstride * M)-th
    // smallest key, and its transposed buffer receives about tstride * (its sampled columns inside the threshold)
    // entries.  Every 64th tile with 16 M entries per item first (an eps that prunes keeps most rows' counts small:
    // no row flagged at 1M x 768, mean eps-degree 120); if the sample says more than 0.5 % of the rows would not
    // fit -- an eps that admits every pair -- the pass is redone at every 16th tile with 32 M entries.  The buffers
    // must fit in a quarter of the free memory, else the full pass.
    int tstride = std::max(1, std::min(64, ntile / 8));
    const char* ev_stride = getenv("ARROWSPACE_SYM_STRIDE");
    if (ev_stride) tstride = std::max(1, std::min(atoi(ev_stride), ntile / 8));
    const char* ev_tcap = getenv("ARROWSPACE_SYM_TCAP");
    const bool publish = !getenv("ARROWSPACE_SYM_NO_PUBLISH");
    int T_CAP = ev_tcap ? atoi(ev_tcap) : 16 * M;
    // The transposed buffers (T_CAP entries of 8 bytes per item: 8 KiB at M = 64) must fit a quarter of the free memory.
    // When those of all items do not -- a shard of 8M rows, 64M x 768 over 8 GPUs: 64 GB -- the main pass runs in
    // column chunks (whole pieces of L tiles, at most 7), each with the buffers of its own items only; beyond 7 chunks
    // the full pass.
    size_t mfree = 0;
    auto free_now = [&]() {   // (asked again where it matters: the candidate lists below are 33 GB at 8M rows)
        size_t f = 0, t = 0;
        mfree = hipMemGetInfo(&f, &t) == hipSuccess ? f : 0;
        // (rehearsal of a rank that holds five shards: pretend that little is free)
        if (const char* ev = getenv("ARROWSPACE_SYM_FREE_GB")) mfree = std::min<size_t>(mfree, (size_t)(atof(ev) * 1e9));
    };
    if (sym) free_now();
    if (sym && mfree && (double)n * T_CAP * 8.0 > 7.0 * 0.25 * (double)mfree) sym = false;
    int Lpiece = 0, npiece = 1, nchunk = 1;
    // bf16 kernel: units in gang order on per-XCD lists (gang_plan); pieces aligned to multiples of L, eight per row
    const bool gang = sym && (variant & 48) == 48 && k2_bf16_enabled() && getenv("ARROWSPACE_K2_GANG_GC") != nullptr;
    std::vector<int4> hunits;
    std::vector<int> hxoff;   // gang order: [chunk][9] list bounds, relative to the chunk's first unit
    dev_tmp<int> d_xoff, d_xcur;
    dev_tmp<int4> d_units, d_units2;
    dev_tmp<int> tr_cnt, tr_idx;
    dev_tmp<float> tr_key, thr0;
    double sym_tiles = 0, thr_tiles = 0;
    if (sym) {
        // units: every row block's tiles [rb * BM / BN, ntile) in pieces of at most L tiles, longest first
        const int per = BM / BN;
        double total = 0;
        for (int rb = 0; rb < nrb; ++rb) total += std::max(0, ntile - rb * per);
        int L = (int)std::max<double>(8.0, std::ceil(total / (dev_cus * 16.0)));
        if ((ntile + L - 1) / L > 7) L = (ntile + 6) / 7;   // at most 7 own segments + the transposed one
        if (gang) L = std::max(8, (ntile + gang_pieces() - 1) / gang_pieces());   // gang order: GC aligned pieces, one gang row spans all columns
        {   // as many pieces as the main pass may need column chunks (sized for the larger buffers: the sample decides later)
            int want = mfree ? (int)std::ceil((double)n * 32 * M * 8.0 / (0.25 * (double)mfree)) : 1;
            if (const char* ev_ch = getenv("ARROWSPACE_SYM_CHUNKS")) want = std::max(want, atoi(ev_ch));
            want = std::max(1, std::min(want, 7));
            if (want > 1) L = std::max(1, std::min(L, (ntile + want - 1) / want));
        }
        S = (ntile + L - 1) / L + 1;
        Lpiece = L;
        npiece = S - 1;
        for (int rb = 0; rb < nrb; ++rb) {
            const int tlo = rb * per;
            int seg = 0;
            if (gang) {   // aligned pieces (rebuilt per column chunk below: only the count matters here)
                for (int pj = tlo / L; pj < S - 1; ++pj, ++seg)
                    if (std::max(tlo, pj * L) < std::min(ntile, (pj + 1) * L)) hunits.push_back(make_int4(rb, std::max(tlo, pj * L), std::min(ntile, (pj + 1) * L), seg));
            } else
                for (int t = tlo; t < ntile; t += L, ++seg) hunits.push_back(make_int4(rb, t, std::min(ntile, t + L), seg));
        }
        std::stable_sort(hunits.begin(), hunits.end(), [](const int4& x, const int4& y) { return x.z - x.y > y.z - y.y; });
        for (const int4& u : hunits) sym_tiles += u.z - u.y;
        AS_HIP(d_units.alloc(hunits.size() + 1));
        AS_HIP(hipMemcpyAsync(d_units, hunits.data(), sizeof(int4) * hunits.size(), hipMemcpyHostToDevice, st));
        AS_HIP(tr_cnt.alloc(n + 2));   // per-item counters, the unit cursor, the count of rows at risk
        AS_HIP(thr0.alloc(n));
        AS_HIP(hipMemsetAsync(tr_cnt, 0, sizeof(int) * (n + 2), st));
    }
    const int units = sym ? (int)hunits.size() : nrb * S;
    const int grid = std::min(units, dev_cus * 2);
    dev_tmp<float>& bkey = c.bkey; dev_tmp<float>& ckey = c.ckey;
    dev_tmp<int>& bidx = c.bidx; dev_tmp<int>& cidx = c.cidx; dev_tmp<int>& ccnt = c.ccnt;
    AS_HIP(bkey.alloc((size_t)grid * BM * CAP));
    AS_HIP(bidx.alloc((size_t)grid * BM * CAP));
    AS_HIP(ckey.alloc((size_t)rows * S * M));
    AS_HIP(cidx.alloc((size_t)rows * S * M));
    AS_HIP(ccnt.alloc((size_t)rows * S));
    KnnArgs& ka = c.ka;
    const float* items = nullptr;   // fp32 rows, their bf16 head + tail image, or (sp->k2_i8: knn_rows decided) the int8 two-digit image
    const bool i8 = (sp->k2_i8 != 0 || sp->ring_i8 != 0) && (variant & 48) == 48;
    if (i8) items = (const float*)sp->x8;
    else if ((variant & 48) == 48) AS_TRY(k2_items(sp, &items));
    else items = sp->x32;
    c.i8 = i8;
    ka.x32 = items; ka.n32 = sp->n32; ka.inorm32 = sp->inorm32;
    ka.n = n; ka.dp = sp->dp; ka.r0 = r0; ka.r1 = r1;
    ka.nrb = nrb; ka.S = S; ka.ntile = ntile; ka.M = M; ka.metric = metric;
    ka.epskey = (float)epskey; ka.coef = (float)(coef * 1.0000002); ka.nmax = (float)(sp->nmax * 1.0000002);
    // round the fp32 bound ingredients up so the device-side bound is never tighter than the fp64 one
    ka.epskey = nextafterf(ka.epskey, INFINITY);
    ka.buf_key = bkey; ka.buf_idx = bidx; ka.out_key = ckey; ka.out_idx = cidx; ka.out_cnt = ccnt;
    ka.xa = items; ka.a_n32 = sp->n32; ka.a_inorm32 = sp->inorm32; ka.row_goff = row_goff; ka.col_goff = col_goff; ka.a_ids = nullptr; ka.a_thr = nullptr;
    ka.units = nullptr; ka.nunits = 0; ka.unit_ctr = nullptr; ka.t_cnt = nullptr; ka.t_key = nullptr; ka.t_idx = nullptr; ka.t_cap = 0;
    ka.ld = i8 ? sp->dp8 / 2 : sp->dp; ka.nslab = (int)(i8 ? sp->dp8 / 64 : sp->dp / 32); ka.fa = i8 ? sp->fa8 : nullptr; ka.a_fa = ka.fa;
    if (sym) {   // (the transposed buffers are allocated once the threshold pass has settled their size)
        ka.units = d_units; ka.nunits = units; ka.unit_ctr = tr_cnt + n; ka.t_cnt = tr_cnt;
    }
    dev_events<2> ev;
    AS_HIP(ev.create());
    hipEvent_t e0 = ev.e[0], e1 = ev.e[1];
    if ((variant & 48) == 48) {
        AS_HIP(hipEventRecord(e0, st));
        if (sym) {
            // threshold pass: every row against every tstride-th column tile (S = 1; the lists are not used): the
            // bound a row ends with -- the M-th smallest key it saw there, or its eps bound -- is an upper bound of
            // its M-th smallest key over all columns
            for (int attempt = 0; attempt < 2; ++attempt) {
                KnnArgs k0 = ka;
                k0.units = nullptr; k0.nunits = 0; k0.unit_ctr = nullptr; k0.t_cnt = nullptr; k0.t_key = nullptr; k0.t_idx = nullptr; k0.t_cap = 0;
                k0.S = 1; k0.tstride = tstride; k0.tphase = 0; k0.ntile = (ntile + tstride - 1) / tstride; k0.thr0 = nullptr; k0.out_thr = thr0;
                thr_tiles += (double)nrb * k0.ntile;
                AS_TRY(launch_k2(k0, metric, false, false, std::min(nrb, dev_cus), st, i8));
                if (attempt == 1 || ev_stride || ev_tcap || tstride <= 16) break;
                // rows whose sampled count says their transposed buffer would not hold what the main pass sends
                hipLaunchKernelGGL(sym_risk_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, (const int*)ccnt, rows,
                                   T_CAP / tstride, (int*)tr_cnt + n + 1);
                AS_HIP(hipGetLastError());
                int risky = 0;
                AS_HIP(hipMemcpyAsync(&risky, tr_cnt + n + 1, sizeof(int), hipMemcpyDeviceToHost, st));
                AS_HIP(hipStreamSynchronize(st));
                dbg("knn_rows: threshold pass at every %dth tile: %d of %lld rows would overflow %d-entry buffers", tstride, risky,
                    (long long)rows, T_CAP);
                if ((double)risky <= 0.005 * (double)rows) break;
                // (when the larger buffers do not fit even in 7 column chunks, the rows that overflow go to the band pass instead)
                free_now();
                if (mfree && (double)n * 32 * M * 8.0 > 7.0 * 0.25 * (double)mfree) break;
                tstride = 16;
                T_CAP = 32 * M;
            }
            // column chunks of the main pass: whole pieces, as few as hold their items' buffers in a quarter of the free memory
            free_now();
            if (mfree) nchunk = (int)std::ceil((double)n * T_CAP * 8.0 / (0.25 * (double)mfree));
            if (const char* ev_ch = getenv("ARROWSPACE_SYM_CHUNKS")) nchunk = atoi(ev_ch);
            nchunk = std::max(1, std::min(nchunk, npiece));
            std::vector<int> cfirst(nchunk + 1, 0), cpiece(nchunk + 1, 0);
            for (int c = 0; c <= nchunk; ++c) cpiece[c] = (int)((int64_t)npiece * c / nchunk);
            cfirst[nchunk] = (int)hunits.size();
            if (nchunk > 1 || gang) {
                // a chunk's units must not reach into the next chunk's columns: pieces aligned to multiples of L (not to the
                // row block's diagonal); a row block's segments count from its first piece -- at most npiece of them, as before
                const int per = BM / BN;
                std::vector<std::vector<int4>> byc(nchunk);
                for (int rb = 0; rb < nrb; ++rb) {
                    const int tlo = rb * per;
                    if (tlo >= ntile) continue;
                    const int p0 = tlo / Lpiece;
                    for (int pj = p0; pj < npiece; ++pj) {
                        const int ta = std::max(tlo, pj * Lpiece), tb = std::min(ntile, (pj + 1) * Lpiece);
                        if (ta >= tb) continue;
                        int c = 0;
                        while (cpiece[c + 1] <= pj) ++c;
                        byc[c].push_back(make_int4(rb, ta, tb, pj - p0));
                    }
                }
                hunits.clear();
                sym_tiles = 0;
                hxoff.assign((size_t)nchunk * 9, 0);
                for (int c = 0; c < nchunk; ++c) {
                    if (gang) gang_plan(byc[c], Lpiece, 0, hxoff.data() + (size_t)c * 9);
                    else std::stable_sort(byc[c].begin(), byc[c].end(), [](const int4& x, const int4& y) { return x.z - x.y > y.z - y.y; });
                    cfirst[c] = (int)hunits.size();
                    for (const int4& u : byc[c]) {
                        hunits.push_back(u);
                        sym_tiles += u.z - u.y;
                    }
                }
                cfirst[nchunk] = (int)hunits.size();
                AS_HIP(hipStreamSynchronize(st));   // (the first upload of the units is done with its source)
                AS_HIP(d_units2.alloc(hunits.size() + 1));
                AS_HIP(hipMemcpyAsync(d_units2, hunits.data(), sizeof(int4) * hunits.size(), hipMemcpyHostToDevice, st));
                if (gang) {
                    AS_HIP(d_xoff.alloc(hxoff.size()));
                    AS_HIP(d_xcur.alloc(8 * 16));
                    AS_HIP(hipMemcpyAsync(d_xoff, hxoff.data(), sizeof(int) * hxoff.size(), hipMemcpyHostToDevice, st));
                }
                if (nchunk > 1) dbg("knn_rows: symmetric pass in %d column chunks (%.1f GB of transposed buffers for all items, %.1f GB free)", nchunk,
                    (double)n * T_CAP * 8.0 / 1e9, (double)mfree / 1e9);
            }
            int64_t cmax = 0;   // items of the largest chunk
            for (int c = 0; c < nchunk; ++c)
                cmax = std::max<int64_t>(cmax, std::min<int64_t>(n, (int64_t)cpiece[c + 1] * Lpiece * BN) - std::min<int64_t>(n, (int64_t)cpiece[c] * Lpiece * BN));
            if (nchunk == 1) cmax = n;
            AS_HIP(tr_key.alloc((size_t)cmax * T_CAP));
            AS_HIP(tr_idx.alloc((size_t)cmax * T_CAP));
            ka.t_cap = T_CAP;
            AS_HIP(hipMemsetAsync(ccnt, 0, sizeof(int) * (size_t)rows * S, st));   // the pass above left its counts there
            ka.thr0 = thr0;
            ka.thr_col = thr0;
            ka.thr_pub = publish ? (float*)thr0 : nullptr;
            const size_t ldst = (sizeof(float) + sizeof(int)) * 4 * (size_t)T_CAP;
            AS_HIP(hipFuncSetAttribute((const void*)transposed_compact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldst));
            for (int c = 0; c < nchunk; ++c) {
                // the chunk's items [j0, j1): their transposed buffers, indexed by item as if all items had one
                const int64_t j0 = nchunk == 1 ? 0 : std::min<int64_t>(n, (int64_t)cpiece[c] * Lpiece * BN);
                const int64_t j1 = nchunk == 1 ? n : std::min<int64_t>(n, (int64_t)cpiece[c + 1] * Lpiece * BN);
                const int cunits = cfirst[c + 1] - cfirst[c];
                if (cunits <= 0 || j1 <= j0) continue;
                ka.units = (nchunk > 1 || gang ? (int4*)d_units2 : (int4*)d_units) + cfirst[c];
                ka.nunits = cunits;
                if (gang) {
                    ka.xoff = (const int*)d_xoff + (size_t)c * 9;
                    ka.xcur = d_xcur;
                    AS_HIP(hipMemsetAsync(d_xcur, 0, sizeof(int) * 8 * 16, st));
                }
                ka.t_key = (float*)tr_key - (size_t)j0 * T_CAP;
                ka.t_idx = (int*)tr_idx - (size_t)j0 * T_CAP;
                if (c) AS_HIP(hipMemsetAsync(ka.unit_ctr, 0, sizeof(int), st));
                const int lgrid = std::min(std::min(cunits, dev_cus), grid);
                AS_TRY(launch_k2(ka, metric, false, true, lgrid, st, i8));
                // the transposed buffers become segment S - 1 of their rows' candidate lists
                hipLaunchKernelGGL(transposed_compact_kernel, dim3((unsigned)((j1 - j0 + 3) / 4)), dim3(256), ldst, st, (const int*)tr_cnt + j0,
                                   (const float*)tr_key, (const int*)tr_idx, T_CAP, j1 - j0, S, S - 1, M, (float*)ckey + (size_t)j0 * S * M,
                                   (int*)cidx + (size_t)j0 * S * M, (int*)ccnt + (size_t)j0 * S);
                AS_HIP(hipGetLastError());
            }
        } else
            AS_TRY(launch_k2(ka, metric, false, false, std::min(units, dev_cus), st, i8));
    } else {
#ifdef AS_ABLATION
        const size_t lds = sizeof(float) * (BM + BN) * LROW * (((variant & 2) && !(variant & 16)) ? 2 : 1) + sizeof(float) * 4 * BM +
                           (sizeof(float) + sizeof(int)) * 4 * CAP;
        int lgrid = grid;
        if ((variant & 1) && !(variant & 32)) lgrid = std::max(8, std::min(units, dev_cus) / 8 * 8);  // one resident block per CU, 8 XCD labels
#define AS_KNN_LAUNCH(VV)                                                                                              \
case VV:                                                                                                           \
    AS_HIP(hipFuncSetAttribute((const void*)knn_mfma_kernel<VV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    AS_HIP(hipEventRecord(e0, st));                                                                                \
    hipLaunchKernelGGL(knn_mfma_kernel<VV>, dim3(lgrid), dim3(256), lds, st, ka);                                  \
    break;
        if (variant & 32) {
            const size_t ldsd = sizeof(float) * 2 * DSLAB + sizeof(float2) * BM + sizeof(int) * 2 * BM + (sizeof(float) + sizeof(int)) * 4 * CAP;
            AS_HIP(hipFuncSetAttribute((const void*)knn_mfma_dma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd));
            AS_HIP(hipFuncSetAttribute((const void*)knn_mfma_dma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd));
            AS_HIP(hipEventRecord(e0, st));
            if (variant & 1) hipLaunchKernelGGL(knn_mfma_dma_kernel<true>, dim3(grid), dim3(256), ldsd, st, ka);
            else hipLaunchKernelGGL(knn_mfma_dma_kernel<false>, dim3(grid), dim3(256), ldsd, st, ka);
        } else
        switch (variant) {
            AS_KNN_LAUNCH(0) AS_KNN_LAUNCH(1) AS_KNN_LAUNCH(2) AS_KNN_LAUNCH(3)
            AS_KNN_LAUNCH(4) AS_KNN_LAUNCH(5) AS_KNN_LAUNCH(6) AS_KNN_LAUNCH(7)
            AS_KNN_LAUNCH(8) AS_KNN_LAUNCH(16) AS_KNN_LAUNCH(24) AS_KNN_LAUNCH(10)
            default:
                set_err("unknown ARROWSPACE_KNN_VARIANT %d", variant);
                return AS_EINVAL;
        }
#undef AS_KNN_LAUNCH
#else
        set_err("ARROWSPACE_KNN_VARIANT=%d selects an ablation kernel: rebuild with -DAS_ABLATION (make ABLATION=1)", variant);
        return AS_EUNSUPPORTED;
#endif
    }
    AS_HIP(hipGetLastError());
    AS_HIP(hipGetLastError());
    AS_HIP(hipEventRecord(e1, st));
    AS_HIP(hipEventSynchronize(e1));
    float ms01 = 0;
    AS_HIP(hipEventElapsedTime(&ms01, e0, e1));
    c.t_mfma = ms01 * 1e-3;
    // flops actually issued: the triangle's tiles in symmetric mode
    c.flops = sym ? 2.0 * (sym_tiles + thr_tiles) * BM * BN * (double)sp->dp : 2.0 * (double)nrb * BM * (double)ntile * BN * (double)sp->dp;
    c.S = S; c.grid = grid; c.dev_cus = dev_cus; c.ntile = ntile; c.nrb = nrb; c.units = units; c.sym = sym;
    return AS_OK;
}


as_status knn_rows(const as_space* sp, const as_graph_params* gp, int64_t r0, int64_t r1, int32_t* out_idx,
                   double* out_key, double* out_dist, double* out_gy, int32_t* out_cnt, double* stats) {
    const int64_t n = sp->n;
    if (r0 < 0 || r1 > n || r0 > r1) {
        set_err("as_knn_rows: bad row range [%lld,%lld) for n=%lld", (long long)r0, (long long)r1, (long long)n);
        return AS_EINVAL;
    }
    const int64_t rows = r1 - r0;
    if (rows == 0) return AS_OK;
    const int64_t k = std::min<int64_t>(gp->k, std::max<int64_t>(n - 1, 1));
    // caller's arrays are rows x gp->k; we write the first k slots and pad the rest
    const int M = pick_list_width(k);
    if (M < 0) {
        set_err("graph_params['k']=%lld exceeds the supported maximum of 120 for n=%lld", (long long)gp->k, (long long)n);
        return AS_EUNSUPPORTED;
    }
    hipStream_t st = sp->stream;
    const int metric = sp->opts.metric;
    const double epskey = metric == AS_METRIC_L2 ? gp->eps * gp->eps : gp->eps;

    int32_t* t_idx = out_idx;
    double *t_key = out_key, *t_dist = out_dist, *t_gy = out_gy;
    // internal lists use width kk = gp->k (caller layout); k_eff may be smaller
    const int64_t kk = gp->k;
    AS_HIP(hipMemsetAsync(t_idx, 0xff, sizeof(int32_t) * rows * kk, st));
    AS_HIP(hipMemsetAsync(out_cnt, 0, sizeof(int32_t) * rows, st));

    dev_tmp<int> flag;
    AS_HIP(flag.alloc(rows + 1));
    int* nflag = flag + rows;
    AS_HIP(hipMemsetAsync(flag, 0, sizeof(int) * (rows + 1), st));
    dev_tmp<double> bandB;
    AS_HIP(bandB.alloc(rows));
    double t_mfma = 0, t_ref = 0, t_fb = 0, flops = 0;
    int nflagged = 0, unproven = 0, band_rows = 0;

    // The int8 two-digit image (twice the bf16 matrix rate, half the operand bytes) serves single-space passes whose items
    // it represents well enough (k2_items_i8); the ring's entry points keep the bf16 image.  err_coef follows sp->k2_i8: set
    // here for this pass and what it refines, cleared on every way out.
    struct I8Scope {
        const as_space* sp;
        ~I8Scope() { sp->k2_i8 = 0; }
    } i8_scope{sp};
    if (!sp->opts.force_exact && k2_bf16_enabled() && !getenv("ARROWSPACE_K2_NO_I8")) {
        bool usable = false;
        AS_TRY(k2_items_i8(sp, &usable));
        sp->k2_i8 = usable ? 1 : 0;
    }
    sp->k2_last_pipe = sp->k2_i8 ? 2 : (k2_bf16_enabled() ? 1 : 0);
    const double coef = err_coef(sp);
    if (!sp->opts.force_exact) {
        KnnCand cand;
        AS_TRY(knn_candidates(sp, gp, r0, r1, M, 0, 0, cand));
        const int S = cand.S, grid = cand.grid, dev_cus = cand.dev_cus, ntile = cand.ntile, units = cand.units;
        const bool sym = cand.sym;
        dev_tmp<float>& ckey = cand.ckey;
        dev_tmp<int>& cidx = cand.cidx;
        dev_tmp<int>& ccnt = cand.ccnt;
        dev_tmp<float>& bkey = cand.bkey;
        dev_tmp<int>& bidx = cand.bidx;
        KnnArgs& ka = cand.ka;
        (void)bkey; (void)bidx;
        dev_events<2> ev;
        AS_HIP(ev.create());
        hipEvent_t e1 = ev.e[0], e2 = ev.e[1];
        AS_HIP(hipEventRecord(e1, st));
        RefineArgs ra;
        ra.x32 = sp->x32; ra.x64 = sp->x64; ra.n64 = sp->n64; ra.n = n; ra.d = sp->d; ra.dp = sp->dp;
        ra.r0 = r0; ra.r1 = r1; ra.S = S; ra.M = M; ra.metric = metric; ra.k = k;
        ra.epskey = epskey; ra.coef = coef; ra.nmax = sp->nmax;
        ra.c_key = ckey; ra.c_idx = cidx; ra.c_cnt = ccnt;
        ra.out_idx = t_idx; ra.out_key = t_key; ra.out_dist = t_dist; ra.out_gy = t_gy; ra.out_cnt = out_cnt;
        ra.flag = flag; ra.nflag = nflag; ra.out_B = bandB;
        // the refine kernel indexes outputs with stride a.k; use the caller stride kk
        ra.k = kk;  // stride == gp->k; rows with fewer than gp->k candidates are padded with -1
        const size_t per_wave = (sizeof(double) * 4 * M + sizeof(float) * ((size_t)S * M + M) + sizeof(int) * ((size_t)S * M + M) + 15) / 16 * 16;
        const size_t lds2 = per_wave * 4;
        AS_HIP(hipFuncSetAttribute((const void*)knn_refine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(knn_refine_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), lds2, st, ra);
        AS_HIP(hipGetLastError());
        AS_HIP(hipEventRecord(e2, st));
        AS_HIP(hipMemcpyAsync(&nflagged, nflag, sizeof(int), hipMemcpyDeviceToHost, st));
        AS_HIP(hipStreamSynchronize(st));
        float ms12 = 0;
        AS_HIP(hipEventElapsedTime(&ms12, e1, e2));
        t_mfma = cand.t_mfma; t_ref = ms12 * 1e-3;
        flops = cand.flops;
        dbg("knn_rows: rows=%lld %s S=%d M=%d units=%d mfma=%.3fs (%.1f TF/s issued) refine=%.3fs flagged=%d", (long long)rows,
            sym ? "symmetric" : "full", S, M, units, t_mfma, flops / std::max(t_mfma, 1e-9) * 1e-12, t_ref, nflagged);
        if (nflagged > 0 && !getenv("ARROWSPACE_NO_BAND_PASS")) {
            // ---- second pass: complete band collection for the flagged rows (K2c)
            const double tb0 = now_s();
            std::vector<int> hflag(rows);
            AS_HIP(hipMemcpy(hflag.data(), flag, sizeof(int) * rows, hipMemcpyDeviceToHost));
            std::vector<int> ids;
            ids.reserve(nflagged);
            for (int64_t lr = 0; lr < rows; ++lr)
                if (hflag[lr]) ids.push_back((int)lr);
            const int nf = (int)ids.size();
            const int64_t nfp = ((int64_t)nf + BM - 1) / BM * BM;
            const int nrb2 = (int)(nfp / BM);
            // column segments: as many as the collection buffers allow (2^27 entries, 1 GB) -- a segment buffer takes at
            // least 128 columns before it overflows, so with one tile per segment (few flagged rows) no band can overflow
            const int S2 = (int)std::max<int64_t>(1, std::min<int64_t>(ntile, ((int64_t)1 << 27) / (nfp * CAP)));
            dev_tmp<int> d_ids, a_ids, c2idx, c2cnt;
            dev_tmp<float> xa, a_n32, a_inorm, a_thr, c2key;
            AS_HIP(d_ids.alloc(nf));
            AS_HIP(a_ids.alloc(nfp));
            AS_HIP(xa.alloc((size_t)(nfp + BM) * sp->dp));
            AS_HIP(a_n32.alloc(nfp)); AS_HIP(a_inorm.alloc(nfp)); AS_HIP(a_thr.alloc(nfp));
            AS_HIP(c2key.alloc((size_t)nfp * S2 * CAP));
            AS_HIP(c2idx.alloc((size_t)nfp * S2 * CAP));
            AS_HIP(c2cnt.alloc((size_t)nfp * S2));
            AS_HIP(hipMemcpyAsync(d_ids, ids.data(), sizeof(int) * nf, hipMemcpyHostToDevice, st));
            AS_HIP(hipMemsetAsync(xa, 0, sizeof(float) * (size_t)(nfp + BM) * sp->dp, st));
            // (ka.x32 is the operand the first pass ran on: the rows are gathered from the same image)
            dev_tmp<float> a_fa;
            AS_HIP(a_fa.alloc(nfp));
            AS_HIP(hipMemsetAsync(a_fa, 0, sizeof(float) * nfp, st));
            hipLaunchKernelGGL(band_gather_kernel, dim3((unsigned)nf), dim3(192), 0, st, ka.x32, sp->n32, sp->inorm32, sp->n64, ka.ld, r0,
                               (const int*)d_ids, nf, (const double*)bandB, metric, coef, sp->nmax, (float*)xa, (float*)a_n32, (float*)a_inorm,
                               (int*)a_ids, (float*)a_thr, (int64_t)0, ka.fa, (float*)a_fa);
            AS_HIP(hipGetLastError());
            KnnArgs kb = ka;
            kb.units = nullptr; kb.nunits = 0; kb.unit_ctr = nullptr; kb.t_cnt = nullptr; kb.t_key = nullptr; kb.t_idx = nullptr; kb.t_cap = 0;
            kb.thr0 = nullptr; kb.thr_col = nullptr; kb.thr_pub = nullptr; kb.out_thr = nullptr;
            kb.r0 = 0; kb.r1 = nf; kb.nrb = nrb2; kb.S = S2; kb.M = CAP;
            kb.out_key = c2key; kb.out_idx = c2idx; kb.out_cnt = c2cnt;
            kb.xa = xa; kb.a_n32 = a_n32; kb.a_inorm32 = a_inorm; kb.a_ids = a_ids; kb.a_thr = a_thr; kb.a_fa = a_fa;
            const int g2 = std::min(nrb2 * S2, dev_cus);
            // per-block append buffers: the first pass sized them for ITS grid
            dev_tmp<float> bkey2;
            dev_tmp<int> bidx2;
            if (g2 > grid) {
                AS_HIP(bkey2.alloc((size_t)g2 * BM * CAP));
                AS_HIP(bidx2.alloc((size_t)g2 * BM * CAP));
                kb.buf_key = bkey2;
                kb.buf_idx = bidx2;
            }
            AS_TRY(launch_k2(kb, metric, true, false, g2, st, cand.i8));
            BandArgs ba;
            ba.x32 = sp->x32; ba.x64 = sp->x64; ba.n64 = sp->n64; ba.d = sp->d; ba.dp = sp->dp; ba.r0 = r0;
            ba.S = S2; ba.CW = CAP; ba.metric = metric; ba.nf = nf; ba.k = kk; ba.epskey = epskey;
            ba.ids = d_ids; ba.c_key = c2key; ba.c_idx = c2idx; ba.c_cnt = c2cnt;
            ba.out_idx = t_idx; ba.out_key = t_key; ba.out_dist = t_dist; ba.out_gy = t_gy; ba.out_cnt = out_cnt; ba.flag = flag;
            const size_t ldsb = (sizeof(double) * 3 + sizeof(int)) * (size_t)BAND_MAX;
            AS_HIP(hipFuncSetAttribute((const void*)knn_band_refine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
            hipLaunchKernelGGL(knn_band_refine_kernel, dim3((unsigned)nf), dim3(256), ldsb, st, ba);
            AS_HIP(hipGetLastError());
            AS_HIP(hipStreamSynchronize(st));
            AS_HIP(hipMemcpy(hflag.data(), flag, sizeof(int) * rows, hipMemcpyDeviceToHost));
            int left = 0;
            for (int64_t lr = 0; lr < rows; ++lr) left += hflag[lr] ? 1 : 0;
            dbg("knn_rows: band pass over %d flagged rows (S=%d): %d left for the row-serial path, %.3fs", nf, S2, left, now_s() - tb0);
            t_ref += now_s() - tb0;
            flops += 2.0 * (double)nrb2 * BM * (double)ntile * BN * (double)sp->dp;
            band_rows = nf - left;
            nflagged = left;
        }
    } else {
        nflagged = (int)rows;
    }
    if (nflagged > 0) {
        const double tf0 = now_s();
        std::vector<int> hflag(rows, 1);
        // the fallback workspace writes the output lists on its own (non-blocking) stream: everything queued on
        // sp->stream -- the memsets of those lists above, the refine kernel -- has to be done first
        AS_HIP(hipStreamSynchronize(st));
        if (!sp->opts.force_exact) AS_HIP(hipMemcpy(hflag.data(), flag, sizeof(int) * rows, hipMemcpyDeviceToHost));
        as_query* ws = nullptr;
        AS_TRY(as_query_create(sp, nullptr, &ws));
        AS_HIP(hipMemsetAsync(nflag, 0, sizeof(int), st));
        AS_HIP(hipStreamSynchronize(st));
        ws->unproven_dev = nflag;   // reused: counts the rows that not even the fp64 path can prove
        as_status fs = AS_OK;
        for (int64_t lr = 0; lr < rows && fs == AS_OK; ++lr) {
            if (!hflag[lr]) continue;
            fs = exact_row_knn(ws, gp, r0 + lr, t_idx + lr * kk, t_key + lr * kk, t_dist + lr * kk, t_gy + lr * kk, out_cnt + lr);
        }
        as_query_free(ws);  // synchronises the workspace stream
        if (fs != AS_OK) return fs;
        AS_HIP(hipMemcpy(&unproven, nflag, sizeof(int), hipMemcpyDeviceToHost));
        if (unproven) dbg("knn_rows: %d rows have more near-ties at the k-th distance than fp64 can order: lists unproven", unproven);
        t_fb = now_s() - tf0;
    }
    if (stats) {
        stats[1] += t_mfma; stats[2] += t_ref; stats[3] += t_fb; stats[6] += nflagged; stats[7] += flops;
        stats[8] += unproven; stats[9] += band_rows;
    }
    return AS_OK;
}

// ------------------------------------------------------------------ K2 over visiting column blocks (multi-GPU ring)
// A rank holds only ITS rows.  The other shards visit one at a time (ring send/recv in the host):
// per visiting block the same fused MFMA kernel runs with A = own rows, B = the block, and a block refinement turns
// its fp32 candidate lists into M exact (key64, global id) entries per row plus the largest kept fp32 key (what was
// dropped from that block is provably beyond it).  After the last block knn_merge_kernel ranks the nblocks * M exact
// entries of every row, applies eps and k, and checks block by block that nothing dropped could belong to the answer;
// rows that fail go round the ring once more in collect mode (K2c per block).  The item matrix is never replicated.
__device__ __forceinline__ void exact_pair2(const float* xa32, const double* xa64, const float* xb32, const double* xb64, int64_t d,
                                            int64_t dp, int64_t i, int64_t j, double& sq, double& dot) {
    const int lane = lane_id();
    double s = 0.0, g = 0.0;
    // the same lane-strided order as exact_pair: a pair gives the same bits whichever path evaluated it
    for (int64_t c = lane; c < d; c += 64) {
        const double a = xa64 ? xa64[i * d + c] : (double)xa32[i * dp + c];
        const double b = xb64 ? xb64[j * d + c] : (double)xb32[j * dp + c];
        const double t = a - b;
        s += t * t;
        g += a * b;
    }
    sq = wave_sum(s);
    dot = wave_sum(g);
}

struct BlockRefineArgs {
    const float *xa32, *xb32;
    const double *xa64, *xb64, *na64, *nb64;
    int64_t d, dp, r0, r1, col_goff;
    int S, M, metric;
    const float* c_key;
    const int* c_idx;
    const int* c_cnt;
    // partial lists of this block: [rows][M] sorted by (key64, global id), count | dropped << 30, largest kept fp32 key
    double *p_key, *p_dist, *p_gy;
    int32_t* p_idx;
    int32_t* p_cnt;
    float* p_t32;
    const float* t32_given = nullptr;   // [r1]: the lower bound of everything these candidate lists did not keep (transposed slices)
};

__global__ __launch_bounds__(256) void knn_block_refine_kernel(BlockRefineArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = threadIdx.x >> 6, lane = lane_id();
    const int SM = a.S * a.M;
    const size_t per_wave = sizeof(double) * 3 * a.M + sizeof(float) * (SM + a.M) + sizeof(int) * (SM + a.M);
    char* base = smem + (size_t)w * ((per_wave + 15) / 16 * 16);
    double* ek = (double*)base;
    double* eg = ek + a.M;
    double* ed = eg + a.M;
    float* ck = (float*)(ed + a.M);
    float* lk = ck + SM;
    int* ci = (int*)(lk + a.M);
    int* li = ci + SM;
    const int64_t row = a.r0 + (int64_t)blockIdx.x * 4 + w;
    if (row >= a.r1) return;
    const int64_t lr = row - a.r0;
    int C = 0, anyfull = 0, lost = 0;
    for (int cs = 0; cs < a.S; ++cs) {
        const int cc = a.c_cnt[lr * a.S + cs];
        const int c = cc & 0xffff;
        anyfull |= (cc >> 30) & 1;
        lost |= (cc >> 31) & 1;   // a transposed buffer overflowed: candidates lost without a bound
        const size_t ob = ((size_t)lr * a.S + cs) * a.M;
        for (int t = lane; t < c; t += 64) {
            ck[C + t] = a.c_key[ob + t];
            ci[C + t] = a.c_idx[ob + t];
        }
        C += c;
    }
    if (C > a.M) anyfull = 1;
    const int Mp = C < a.M ? C : a.M;
    AS_LDS_FENCE();
    for (int t = lane; t < C; t += 64) {
        const float k = ck[t];
        const int i = ci[t];
        int rank = 0;
        for (int s2 = 0; s2 < C; ++s2) rank += lex_less<float>(ck[s2], ci[s2], k, i) ? 1 : 0;
        if (rank < a.M) {
            lk[rank] = k;
            li[rank] = i;
        }
    }
    AS_LDS_FENCE();
    const double ni = a.na64[row];
    for (int t = 0; t < Mp; ++t) {
        const int64_t j = (int64_t)li[t] - a.col_goff;
        double sq, dot;
        exact_pair2(a.xa32, a.xa64, a.xb32, a.xb64, a.d, a.dp, row, j, sq, dot);
        if (lane == 0) {
            if (a.metric == AS_METRIC_L2) {
                ek[t] = sq;
                ed[t] = sqrt(sq);
                eg[t] = dot;
            } else {
                const double den = sqrt(ni * a.nb64[j]);
                const double c = den > 0.0 ? dot / den : 0.0;
                const double dd = cosine_distance(c);
                ek[t] = dd;
                ed[t] = dd;
                eg[t] = c;
            }
        }
    }
    AS_LDS_FENCE();
    for (int t = lane; t < Mp; t += 64) {
        const double k = ek[t];
        const int i = li[t];
        int rank = 0;
        for (int s2 = 0; s2 < Mp; ++s2) rank += lex_less<double>(ek[s2], li[s2], k, i) ? 1 : 0;
        const size_t o = (size_t)lr * a.M + rank;
        a.p_key[o] = k;
        a.p_dist[o] = ed[t];
        a.p_gy[o] = eg[t];
        a.p_idx[o] = i;
    }
    if (lane == 0) {
        a.p_cnt[lr] = Mp | ((anyfull | lost) << 30);
        a.p_t32[lr] = lost ? -__int_as_float(0x7f800000) : (a.t32_given ? a.t32_given[row] : (Mp > 0 ? lk[Mp - 1] : 0.0f));
    }
}

struct MergeArgs {
    int64_t rows, k;          // k: cap and output stride
    int nblocks, M, metric;
    double epskey, coef;
    const double* na64;       // own rows' squared norms, [r0 + lr]
    int64_t r0;
    const double *p_key, *p_dist, *p_gy;   // [nblocks][rows][M]
    const int32_t* p_idx;
    const int32_t* p_cnt;     // [nblocks][rows]
    const float* p_t32;
    const double* nmax;       // [nblocks]: largest squared norm of the block (error bound of what it dropped)
    int folded;               // the one slice is a running fold of all blocks: p_t32 already holds min_b (T32_b - e_b)
    int32_t* out_idx;
    double *out_key, *out_dist, *out_gy;
    int32_t* out_cnt;
    int* flag;
    int* nflag;
    double* out_B;
};

// one wave per row
__global__ __launch_bounds__(256) void knn_merge_kernel(MergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = threadIdx.x >> 6, lane = lane_id();
    const int cap = a.nblocks * a.M;
    char* base = smem + (size_t)w * ((sizeof(double) + sizeof(int) * 2) * cap);
    double* mk = (double*)base;
    int* mi = (int*)(mk + cap);
    int* mp = mi + cap;          // position in the partial arrays
    const int64_t lr = (int64_t)blockIdx.x * 4 + w;
    if (lr >= a.rows) return;
    int C = 0;
    for (int b = 0; b < a.nblocks; ++b) {
        const int c = a.p_cnt[(size_t)b * a.rows + lr] & 0xffff;
        const size_t ob = ((size_t)b * a.rows + lr) * a.M;
        for (int t = lane; t < c; t += 64) {
            mk[C + t] = a.p_key[ob + t];
            mi[C + t] = a.p_idx[ob + t];
            mp[C + t] = (int)(ob + t - (size_t)lr * a.M);   // relative: fits an int
        }
        C += c;
    }
    AS_LDS_FENCE();
    int npass_l = 0;
    double kth_l = -1.0;
    for (int t = lane; t < C; t += 64) {
        const double kk = mk[t];
        if (!(kk <= a.epskey)) continue;
        npass_l += 1;
        int rank = 0;
        for (int s2 = 0; s2 < C; ++s2) rank += lex_less<double>(mk[s2], mi[s2], kk, mi[t]) ? 1 : 0;
        if (rank < a.k) {
            const size_t src = (size_t)lr * a.M + (size_t)mp[t];
            a.out_idx[lr * a.k + rank] = mi[t];
            a.out_key[lr * a.k + rank] = kk;
            a.out_dist[lr * a.k + rank] = a.p_dist[src];
            a.out_gy[lr * a.k + rank] = a.p_gy[src];
            if (rank == a.k - 1) kth_l = kk;
        }
    }
    const int npass = wave_sum(npass_l);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(kth_l, o, 64);
        kth_l = other > kth_l ? other : kth_l;
    }
    const int cnt = npass < a.k ? npass : (int)a.k;
    for (int64_t t = cnt + lane; t < a.k; t += 64) a.out_idx[lr * a.k + t] = -1;
    if (lane == 0) {
        a.out_cnt[lr] = cnt;
        const double B = npass >= a.k ? kth_l : a.epskey;
        const double ni = a.na64[a.r0 + lr];
        int bad = 0;
        for (int b = 0; b < a.nblocks; ++b) {
            const int cc = a.p_cnt[(size_t)b * a.rows + lr];
            if (!((cc >> 30) & 1) || (cc & 0xffff) == 0) continue;   // nothing was dropped from this block
            const double e = a.folded ? 0.0 : (a.metric == AS_METRIC_L2 ? a.coef * (ni + a.nmax[b]) : a.coef);
            if (!((double)a.p_t32[(size_t)b * a.rows + lr] - e > B)) bad = 1;
        }
        a.flag[lr] = bad;
        if (bad) {
            atomicAdd(a.nflag, 1);
            a.out_B[lr] = B;
        }
    }
}

// band pass of one visiting block for the flagged rows: exact evaluation of everything the collect-mode kernel
// kept, the M smallest (key64, id) of THIS block into the row's partial list (nothing dropped: count without flag)
struct BlockBandArgs {
    const float *xa32, *xb32;
    const double *xa64, *xb64, *na64, *nb64;
    int64_t d, dp, r0, col_goff;
    int S, CW, M, metric, nf;
    const int* ids;
    const int* c_idx;
    const int* c_cnt;
    double *p_key, *p_dist, *p_gy;
    int32_t* p_idx;
    int32_t* p_cnt;
    float* p_t32;
    int* overflow;   // counts rows whose band did not fit
    int* flag;       // bit 1 is set for such a row: its first-round list stands (unproven), the host restores it
};

__global__ __launch_bounds__(256) void knn_block_band_kernel(BlockBandArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* ek = (double*)smem;
    double* ed = ek + BAND_MAX;
    double* eg = ed + BAND_MAX;
    int* ci = (int*)(eg + BAND_MAX);
    __shared__ int s_C, s_over;
    const int w = threadIdx.x >> 6, lane = lane_id();
    const int f = blockIdx.x;
    const int lr = a.ids[f];
    const int64_t row = a.r0 + lr;
    if (threadIdx.x == 0) {
        int C = 0, over = 0;
        for (int cs = 0; cs < a.S; ++cs) {
            const int cc = a.c_cnt[(size_t)f * a.S + cs];
            over |= (cc >> 31) & 1;
            C += cc & 0xffff;
        }
        s_C = C;
        s_over = over || C > BAND_MAX;
    }
    __syncthreads();
    if (s_over) {
        if (threadIdx.x == 0) {
            atomicAdd(a.overflow, 1);
            atomicOr(&a.flag[lr], 2);
            a.p_cnt[lr] = 0;      // nothing from this block: the row's second-round list is void, the first-round one stands
            a.p_t32[lr] = 0.0f;
        }
        return;
    }
    const int C = s_C;
    {
        int off = 0;
        for (int cs = 0; cs < a.S; ++cs) {
            const int c = a.c_cnt[(size_t)f * a.S + cs] & 0xffff;
            const size_t ob = ((size_t)f * a.S + cs) * a.CW;
            for (int t = threadIdx.x; t < c; t += blockDim.x) ci[off + t] = a.c_idx[ob + t];
            off += c;
        }
    }
    __syncthreads();
    const double ni = a.na64[row];
    for (int t = w; t < C; t += 4) {
        const int64_t j = (int64_t)ci[t] - a.col_goff;
        double sq, dot;
        exact_pair2(a.xa32, a.xa64, a.xb32, a.xb64, a.d, a.dp, row, j, sq, dot);
        if (lane == 0) {
            if (a.metric == AS_METRIC_L2) {
                ek[t] = sq;
                ed[t] = sqrt(sq);
                eg[t] = dot;
            } else {
                const double den = sqrt(ni * a.nb64[j]);
                const double c = den > 0.0 ? dot / den : 0.0;
                const double dd = cosine_distance(c);
                ek[t] = dd;
                ed[t] = dd;
                eg[t] = c;
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < C; t += blockDim.x) {
        const double kk = ek[t];
        int rank = 0;
        for (int s2 = 0; s2 < C; ++s2) rank += lex_less<double>(ek[s2], ci[s2], kk, ci[t]) ? 1 : 0;
        if (rank < a.M) {
            const size_t o = (size_t)lr * a.M + rank;
            a.p_key[o] = kk;
            a.p_dist[o] = ed[t];
            a.p_gy[o] = eg[t];
            a.p_idx[o] = ci[t];
        }
    }
    if (threadIdx.x == 0) {
        a.p_cnt[lr] = C < a.M ? C : a.M;   // complete inside the band: no "dropped" flag
        a.p_t32[lr] = 0.0f;
    }
}

// Running fold of the visiting blocks: run <- the M smallest (key64, id) of run U blk, per row; the running bound
// min over blocks of (T32_b - e_b) is kept in run_t32 (rounded down: conservative).  Dropping EXACT entries beyond
// the M-th needs no proof -- they cannot be among the k <= M - 8 nearest.  HBM for the lists: two slices, whatever the
// number of ranks (all of them at once took 115 GB per rank at 64M x 768 / 8 GPUs).
// mode 0: every row; 1: flagged rows only; 2: flagged rows only, their running list discarded first (first block of
// the second round).
struct FoldArgs {
    int64_t rows, r0;
    int M, metric, mode;
    int raw = 0;   // the block's bound is taken as it is (chunks of one block: the error term is subtracted later, once)
    double coef, nmax_b;
    const double* na64;
    const int* flag;
    double *r_key, *r_dist, *r_gy;
    int32_t* r_idx;
    int32_t* r_cnt;
    float* r_t32;
    const double *b_key, *b_dist, *b_gy;
    const int32_t* b_idx;
    const int32_t* b_cnt;
    const float* b_t32;
};

__global__ __launch_bounds__(256) void knn_fold_kernel(FoldArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = threadIdx.x >> 6, lane = lane_id();
    const int cap = 2 * a.M;
    char* base = smem + (size_t)w * ((sizeof(double) * 3 + sizeof(int)) * cap);
    double* mk = (double*)base;
    double* md = mk + cap;
    double* mg = md + cap;
    int* mi = (int*)(mg + cap);
    const int64_t lr = (int64_t)blockIdx.x * 4 + w;
    if (lr >= a.rows) return;
    if (a.mode != 0 && !a.flag[lr]) return;
    const size_t o = (size_t)lr * a.M;
    const int rc = a.mode == 2 ? 0 : a.r_cnt[lr];
    const int bc = a.b_cnt[lr];
    const int cr = rc & 0xffff, cb = bc & 0xffff;
    for (int t = lane; t < cr; t += 64) {
        mk[t] = a.r_key[o + t];
        md[t] = a.r_dist[o + t];
        mg[t] = a.r_gy[o + t];
        mi[t] = a.r_idx[o + t];
    }
    for (int t = lane; t < cb; t += 64) {
        mk[cr + t] = a.b_key[o + t];
        md[cr + t] = a.b_dist[o + t];
        mg[cr + t] = a.b_gy[o + t];
        mi[cr + t] = a.b_idx[o + t];
    }
    AS_LDS_FENCE();
    const int C = cr + cb;
    for (int t = lane; t < C; t += 64) {
        int rank = 0;
        for (int s2 = 0; s2 < C; ++s2) rank += lex_less<double>(mk[s2], mi[s2], mk[t], mi[t]) ? 1 : 0;
        if (rank < a.M) {
            a.r_key[o + rank] = mk[t];
            a.r_dist[o + rank] = md[t];
            a.r_gy[o + rank] = mg[t];
            a.r_idx[o + rank] = mi[t];
        }
    }
    if (lane == 0) {
        const int anyb = (bc >> 30) & 1, anyr = (rc >> 30) & 1;
        float tb = (a.mode == 2 || !anyr) ? __int_as_float(0x7f800000) : a.r_t32[lr];
        if (anyb) {   // (a gated transposed slice may have kept nothing and still have turned candidates away: its bound counts)
            const double e = a.metric == AS_METRIC_L2 ? a.coef * (a.na64[a.r0 + lr] + a.nmax_b) : a.coef;
            const float nb = a.raw ? a.b_t32[lr] : __double2float_rd((double)a.b_t32[lr] - e);
            tb = nb < tb ? nb : tb;
        }
        a.r_cnt[lr] = (C < a.M ? C : a.M) | ((anyb | (a.mode == 2 ? 0 : anyr)) << 30);
        a.r_t32[lr] = tb;
    }
}

static as_status fold_launch(const as_space* sp, int64_t r0, int64_t r1, int M, int mode, int raw, double block_nmax, const int32_t* flag,
                             double* r_key, double* r_dist, double* r_gy, int32_t* r_idx, int32_t* r_cnt, float* r_t32, const double* b_key,
                             const double* b_dist, const double* b_gy, const int32_t* b_idx, const int32_t* b_cnt, const float* b_t32) {
    const int64_t rows = r1 - r0;
    if (rows <= 0) return AS_OK;
    FoldArgs fa;
    fa.raw = raw;
    fa.rows = rows; fa.r0 = r0; fa.M = M; fa.metric = sp->opts.metric; fa.mode = mode; fa.coef = err_coef(sp); fa.nmax_b = block_nmax;
    fa.na64 = sp->n64; fa.flag = flag;
    fa.r_key = r_key; fa.r_dist = r_dist; fa.r_gy = r_gy; fa.r_idx = r_idx; fa.r_cnt = r_cnt; fa.r_t32 = r_t32;
    fa.b_key = b_key; fa.b_dist = b_dist; fa.b_gy = b_gy; fa.b_idx = b_idx; fa.b_cnt = b_cnt; fa.b_t32 = b_t32;
    const size_t lds = 4 * (sizeof(double) * 3 + sizeof(int)) * (size_t)(2 * M);
    AS_HIP(hipFuncSetAttribute((const void*)knn_fold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(knn_fold_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), lds, sp->stream, fa);
    AS_HIP(hipGetLastError());
    AS_HIP(hipStreamSynchronize(sp->stream));
    return AS_OK;
}

as_status knn_fold(const as_space* sp, int64_t r0, int64_t r1, int M, int mode, double block_nmax, const int32_t* flag, double* r_key,
                   double* r_dist, double* r_gy, int32_t* r_idx, int32_t* r_cnt, float* r_t32, const double* b_key, const double* b_dist,
                   const double* b_gy, const int32_t* b_idx, const int32_t* b_cnt, const float* b_t32) {
    return fold_launch(sp, r0, r1, M, mode, 0, block_nmax, flag, r_key, r_dist, r_gy, r_idx, r_cnt, r_t32, b_key, b_dist, b_gy, b_idx, b_cnt, b_t32);
}

// chunks of ONE block's slice: lists merged, the smaller raw bound kept (pointers indexed from the first row)
static as_status knn_fold_raw(const as_space* sp, int64_t rows, int M, double* r_key, double* r_dist, double* r_gy, int32_t* r_idx,
                              int32_t* r_cnt, float* r_t32, const double* b_key, const double* b_dist, const double* b_gy, const int32_t* b_idx,
                              const int32_t* b_cnt, const float* b_t32) {
    return fold_launch(sp, 0, rows, M, 0, 1, 0.0, nullptr, r_key, r_dist, r_gy, r_idx, r_cnt, r_t32, b_key, b_dist, b_gy, b_idx, b_cnt, b_t32);
}

static int device_cus(int device) {
    hipDeviceProp_t prop;
    return hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 256;
}

static as_status block_check(const as_space* sp, const as_space* cols, int64_t r0, int64_t r1, const char* who) {
    if (!sp || !cols || cols->d != sp->d || cols->dp != sp->dp || cols->device != sp->device || cols->opts.metric != sp->opts.metric) {
        set_err("%s: the column block does not match the space (features, device or metric)", who);
        return AS_EINVAL;
    }
    if (r0 < 0 || r1 > sp->n || r0 > r1) {
        set_err("%s: bad row range [%lld,%lld) for n=%lld", who, (long long)r0, (long long)r1, (long long)sp->n);
        return AS_EINVAL;
    }
    if (sp->opts.force_exact || cols->opts.force_exact) {
        set_err("%s: items outside the fp32-safe range (force_exact) are not supported on the ring path", who);
        return AS_EUNSUPPORTED;
    }
    return AS_OK;
}

// Ring build: may the block passes run on the int8 two-digit images?  Every rank reports what its own rows measure
// (ring_i8_stats: U, V, 1 when the image cannot be used -- non-finite rows --; makes the image), the host all-gathers the
// three numbers and hands every rank the ring-wide maxima (ring_i8_set; usable = 0, or a coefficient beyond 1e-3: the
// bf16 head + tail form, on every rank alike).
as_status ring_i8_stats(as_space* sp, double* out3) {
    AS_HIP(hipSetDevice(sp->device));
    out3[0] = out3[1] = 0.0;
    const bool off = !k2_bf16_enabled() || getenv("ARROWSPACE_K2_NO_I8") != nullptr || sp->opts.force_exact;
    out3[2] = off ? 1.0 : 0.0;
    if (off || sp->n == 0) return AS_OK;   // (an empty shard has no say)
    bool usable = false;
    const as_status s = k2_items_i8(sp, &usable);
    if (s == AS_ENOMEM) {   // no room for this rank's image: "unusable" is an answer the ring can act on (bf16 passes on every rank), an error is not
        (void)hipGetLastError();
        out3[2] = 1.0;
        return AS_OK;
    }
    AS_TRY(s);
    out3[0] = sp->u8max;
    out3[1] = sp->v8max;
    out3[2] = sp->x8 && !sp->x8_bad ? 0.0 : 1.0;
    return AS_OK;
}

as_status ring_i8_set(as_space* sp, double u_max, double v_max, int32_t usable) {
    sp->ring_i8 = 0;
    if (!usable || !(u_max >= 0.0) || !(v_max >= 0.0)) return AS_OK;
    const double coef = err_coef_i8(u_max, v_max);
    if (!(coef <= 1.0e-3) || sp->dp > 131072) return AS_OK;
    if (sp->n > 0 && (!sp->x8 || sp->x8_bad)) {
        set_err("as_ring_i8_set: this rank has no usable int8 image (as_ring_i8_stats first)");
        return AS_EINVAL;
    }
    sp->ring_u8 = u_max;
    sp->ring_v8 = v_max;
    sp->ring_coef8 = coef;
    sp->ring_i8 = 1;
    dbg("ring: block passes on the int8 images, U = %.3e, V = %.3e -> coefficient %.3e (bf16: %.3e)", u_max, v_max, coef, err_coef_dp(sp->dp));
    return AS_OK;
}

// first pass of one visiting block: fused MFMA kernel + block refinement -> the block's slice of the partial lists
as_status knn_block(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t r0, int64_t r1, int64_t row_goff,
                    int64_t col_goff, int M, double* p_key, double* p_dist, double* p_gy, int32_t* p_idx, int32_t* p_cnt,
                    float* p_t32) {
    AS_TRY(block_check(sp, cols, r0, r1, "as_knn_block"));
    const int64_t rows = r1 - r0;
    if (rows == 0) return AS_OK;
    if (sp == cols && r0 == 0 && r1 == sp->n) {
        // the own block: the same first pass as a single-space build (symmetric: half the tiles), ids global
        KnnCand cand;
        AS_TRY(knn_candidates(sp, gp, r0, r1, M, row_goff, col_goff, cand));
        BlockRefineArgs ra;
        ra.xa32 = sp->x32; ra.xa64 = sp->x64; ra.xb32 = sp->x32; ra.xb64 = sp->x64; ra.na64 = sp->n64; ra.nb64 = sp->n64;
        ra.d = sp->d; ra.dp = sp->dp; ra.r0 = r0; ra.r1 = r1; ra.col_goff = col_goff; ra.S = cand.S; ra.M = M; ra.metric = sp->opts.metric;
        ra.c_key = cand.ckey; ra.c_idx = cand.cidx; ra.c_cnt = cand.ccnt;
        ra.p_key = p_key; ra.p_dist = p_dist; ra.p_gy = p_gy; ra.p_idx = p_idx; ra.p_cnt = p_cnt; ra.p_t32 = p_t32;
        const size_t pw = (sizeof(double) * 3 * M + sizeof(float) * ((size_t)cand.S * M + M) + sizeof(int) * ((size_t)cand.S * M + M) + 15) / 16 * 16;
        AS_HIP(hipFuncSetAttribute((const void*)knn_block_refine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(pw * 4)));
        hipLaunchKernelGGL(knn_block_refine_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), pw * 4, sp->stream, ra);
        AS_HIP(hipGetLastError());
        AS_HIP(hipStreamSynchronize(sp->stream));
        sp->kstats[7] += cand.flops;
        return AS_OK;
    }
    hipStream_t st = sp->stream;
    const int metric = sp->opts.metric;
    const double epskey = metric == AS_METRIC_L2 ? gp->eps * gp->eps : gp->eps;
    const double coef = err_coef(sp);
    const int dev_cus = device_cus(sp->device);
    const int nrb = (int)((rows + BM - 1) / BM);
    const int ntile = (int)(cols->np / BN);
    int S = 1;
    while (S < 8 && nrb * S < dev_cus * 16 && ntile / (S * 2) >= 8) S *= 2;
    const int units = nrb * S, grid = std::min(units, dev_cus);
    dev_tmp<float> bkey, ckey;
    dev_tmp<int> bidx, cidx, ccnt;
    AS_HIP(bkey.alloc((size_t)grid * BM * CAP));
    AS_HIP(bidx.alloc((size_t)grid * BM * CAP));
    AS_HIP(ckey.alloc((size_t)rows * S * M));
    AS_HIP(cidx.alloc((size_t)rows * S * M));
    AS_HIP(ccnt.alloc((size_t)rows * S));
    const double nmax = std::max(sp->nmax, cols->nmax);
    KnnArgs ka;
    RingOperand op;
    AS_TRY(ring_operand(sp, cols, &op));
    ka.x32 = op.b; ka.n32 = cols->n32; ka.inorm32 = cols->inorm32; ka.n = cols->n; ka.dp = sp->dp; ka.r0 = r0; ka.r1 = r1;
    ka.nrb = nrb; ka.S = S; ka.ntile = ntile; ka.M = M; ka.metric = metric;
    ka.epskey = nextafterf((float)epskey, INFINITY); ka.coef = (float)(coef * 1.0000002); ka.nmax = (float)(nmax * 1.0000002);
    ka.buf_key = bkey; ka.buf_idx = bidx; ka.out_key = ckey; ka.out_idx = cidx; ka.out_cnt = ccnt;
    ka.xa = op.a; ka.a_n32 = sp->n32; ka.a_inorm32 = sp->inorm32; ka.row_goff = row_goff + 0; ka.col_goff = col_goff;
    ka.a_ids = nullptr; ka.a_thr = nullptr;
    ka.ld = op.ld; ka.nslab = op.nslab; ka.fa = op.fb; ka.a_fa = op.fa;
    AS_TRY(launch_k2(ka, metric, false, false, grid, st, op.i8));
    BlockRefineArgs ra;
    ra.xa32 = sp->x32; ra.xa64 = sp->x64; ra.xb32 = cols->x32; ra.xb64 = cols->x64; ra.na64 = sp->n64; ra.nb64 = cols->n64;
    ra.d = sp->d; ra.dp = sp->dp; ra.r0 = r0; ra.r1 = r1; ra.col_goff = col_goff; ra.S = S; ra.M = M; ra.metric = metric;
    ra.c_key = ckey; ra.c_idx = cidx; ra.c_cnt = ccnt;
    ra.p_key = p_key; ra.p_dist = p_dist; ra.p_gy = p_gy; ra.p_idx = p_idx; ra.p_cnt = p_cnt; ra.p_t32 = p_t32;
    const size_t per_wave = (sizeof(double) * 3 * M + sizeof(float) * ((size_t)S * M + M) + sizeof(int) * ((size_t)S * M + M) + 15) / 16 * 16;
    AS_HIP(hipFuncSetAttribute((const void*)knn_block_refine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(per_wave * 4)));
    hipLaunchKernelGGL(knn_block_refine_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), per_wave * 4, st, ra);
    AS_HIP(hipGetLastError());
    AS_HIP(hipStreamSynchronize(st));   // the scratch dies with this frame; the block may be freed by the caller
    sp->kstats[7] += 2.0 * (double)nrb * BM * (double)ntile * BN * (double)sp->dp;
    return AS_OK;
}

// The threshold a visiting item's transposed appends were admitted with (min(eps bound, thr)), as +inf where the eps
// bound decided: what the eps bound turns away is outside eps and bounds nothing.
__global__ void pair_gate_kernel(const float* __restrict__ col_thr, const float* __restrict__ n32, int64_t nc, int64_t j0, int64_t j1,
                                 int metric, float epskey, float coef, float nmax, float* __restrict__ gate) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nc) return;
    const float eb = metric == AS_METRIC_L2 ? epskey + coef * (n32[j] + nmax) : epskey + coef;
    const float t = col_thr && j >= j0 && j < j1 ? col_thr[j] : __int_as_float(0x7f800000);   // items outside the tile range took no part
    gate[j] = t < eb ? t : __int_as_float(0x7f800000);
}

// Per-row thresholds from a folded partial list (ring): an upper bound of the row's M-th smallest fp32 key over all
// columns -- its M-th exact key so far plus the fp32 error bound -- or +inf while the list is not full.
__global__ void knn_thresholds_kernel(const double* __restrict__ r_key, const int32_t* __restrict__ r_cnt, const double* __restrict__ n64,
                                      int64_t r0, int64_t rows, int M, int metric, double coef, double nmax, float* __restrict__ out) {
    const int64_t lr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lr >= rows) return;
    const int c = r_cnt[lr] & 0xffff;
    float u = __int_as_float(0x7f800000);
    if (c >= M) {
        const double e = metric == AS_METRIC_L2 ? coef * (n64[r0 + lr] + nmax) : coef;
        u = __double2float_ru((r_key[(size_t)lr * M + M - 1] + e) * 1.000001);
    }
    out[lr] = u;
}

as_status knn_thresholds(const as_space* sp, int64_t r0, int64_t r1, int M, double nmax_all, const double* r_key, const int32_t* r_cnt,
                         float* out_thr) {
    const int64_t rows = r1 - r0;
    if (rows <= 0) return AS_OK;
    hipLaunchKernelGGL(knn_thresholds_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, sp->stream, r_key, r_cnt, sp->n64, r0, rows,
                       M, sp->opts.metric, err_coef(sp), std::max(nmax_all, sp->nmax), out_thr);
    AS_HIP(hipGetLastError());
    AS_HIP(hipStreamSynchronize(sp->stream));
    return AS_OK;
}

// One unordered pair of blocks, computed once (symmetric ring): own rows [r0, r1) against the visiting block's column
// tiles [ct0, ct1).  Every key serves both items: the own rows' candidate lists as in knn_block, and the visiting
// items' transposed buffers, gated by their thresholds (col_thr, may be null: eps bound alone).  Both sides are refined
// exactly here, while both shards are resident: p_* is the own rows' slice (as knn_block's, indexed from r0), q_* the
// visiting items' slice [cols->n][M] w.r.t. the own rows as columns -- q_t32 the lower bound of what its buffers turned
// away or dropped (-inf for a buffer that overflowed: the row fails its proof and goes round again).
//
// pair_chunk: the column tiles [ca, cb) of the visiting block -- its scratch (transposed buffers of 16 M entries per
// visiting item, 8 KiB each at M = 64) covers the chunk's items only.
static as_status pair_chunk(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t r0, int64_t r1, int64_t ca,
                            int64_t cb, int64_t row_goff, int64_t col_goff, const float* col_thr, const float* row_thr, int M, double* p_key, double* p_dist,
                            double* p_gy, int32_t* p_idx, int32_t* p_cnt, float* p_t32, double* q_key, double* q_dist, double* q_gy,
                            int32_t* q_idx, int32_t* q_cnt, float* q_t32) {
    hipStream_t st = sp->stream;
    const int64_t rows = r1 - r0;
    const int64_t j0 = ca * BN, j1 = std::min<int64_t>(cols->n, cb * BN), ncc = j1 - j0;   // the chunk's items
    const int ntile_all = (int)(cols->np / BN);
    const int metric = sp->opts.metric;
    const double epskey = metric == AS_METRIC_L2 ? gp->eps * gp->eps : gp->eps;
    const double coef = err_coef(sp);
    const int dev_cus = device_cus(sp->device);
    const int nrb = (int)((rows + BM - 1) / BM);
    const int ntile = (int)(cb - ca);
    // units: every row block's tiles in pieces of at most L, one segment per piece
    int L = (int)std::max<double>(8.0, std::ceil((double)nrb * ntile / (dev_cus * 16.0)));
    if ((ntile + L - 1) / L > 8) L = (ntile + 7) / 8;
    const bool gang = k2_bf16_enabled() && getenv("ARROWSPACE_K2_GANG_GC") != nullptr;
    if (gang) L = std::max(8, (ntile + gang_pieces() - 1) / gang_pieces());   // one gang row spans the chunk's columns (gang_plan)
    const int S = (ntile + L - 1) / L;
    std::vector<int4> hunits;
    for (int rb = 0; rb < nrb; ++rb) {
        int seg = 0;
        for (int t = (int)ca; t < cb; t += L, ++seg) hunits.push_back(make_int4(rb, t, (int)std::min<int64_t>(cb, t + L), seg));
    }
    std::stable_sort(hunits.begin(), hunits.end(), [](const int4& x, const int4& y) { return x.z - x.y > y.z - y.y; });
    int hxoff[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (gang) gang_plan(hunits, L, (int)ca, hxoff);
    const int units = (int)hunits.size(), grid = std::min(units, dev_cus);
    const int T_CAP = 16 * M;
    dev_tmp<int> d_xo;   // gang order: 9 list bounds, then the 8 cursors 16 ints apart
    dev_tmp<int4> d_units;
    dev_tmp<float> bkey, ckey, tr_key, c2key, gate, t_bound;
    dev_tmp<int> bidx, cidx, ccnt, tr_cnt, tr_idx, c2idx, c2cnt;
    AS_HIP(d_units.alloc(units));
    AS_HIP(hipMemcpyAsync(d_units, hunits.data(), sizeof(int4) * units, hipMemcpyHostToDevice, st));
    AS_HIP(bkey.alloc((size_t)grid * BM * CAP));
    AS_HIP(bidx.alloc((size_t)grid * BM * CAP));
    AS_HIP(ckey.alloc((size_t)rows * S * M));
    AS_HIP(cidx.alloc((size_t)rows * S * M));
    AS_HIP(ccnt.alloc((size_t)rows * S));
    AS_HIP(tr_cnt.alloc(ncc + 1));
    AS_HIP(tr_key.alloc((size_t)ncc * T_CAP));
    AS_HIP(tr_idx.alloc((size_t)ncc * T_CAP));
    AS_HIP(hipMemsetAsync(tr_cnt, 0, sizeof(int) * (ncc + 1), st));
    AS_HIP(hipMemsetAsync(ccnt, 0, sizeof(int) * (size_t)rows * S, st));
    const double nmax = std::max(sp->nmax, cols->nmax);
    KnnArgs ka;
    RingOperand op;
    AS_TRY(ring_operand(sp, cols, &op));
    ka.x32 = op.b; ka.n32 = cols->n32; ka.inorm32 = cols->inorm32; ka.n = cols->n; ka.dp = sp->dp; ka.r0 = r0; ka.r1 = r1;
    ka.nrb = nrb; ka.S = S; ka.ntile = ntile_all; ka.M = M; ka.metric = metric;
    ka.epskey = nextafterf((float)epskey, INFINITY); ka.coef = (float)(coef * 1.0000002); ka.nmax = (float)(nmax * 1.0000002);
    ka.buf_key = bkey; ka.buf_idx = bidx; ka.out_key = ckey; ka.out_idx = cidx; ka.out_cnt = ccnt;
    ka.xa = op.a; ka.a_n32 = sp->n32; ka.a_inorm32 = sp->inorm32; ka.row_goff = row_goff; ka.col_goff = col_goff;
    ka.a_ids = nullptr; ka.a_thr = nullptr;
    ka.ld = op.ld; ka.nslab = op.nslab; ka.fa = op.fb; ka.a_fa = op.fa;
    // the kernel addresses the transposed buffers by the item's number inside the block: bases moved back by the chunk's
    // first item (only items of the chunk's tiles are ever addressed)
    ka.units = d_units; ka.nunits = units; ka.unit_ctr = (int*)tr_cnt + ncc; ka.t_cnt = (int*)tr_cnt - j0;
    ka.t_key = (float*)tr_key - (size_t)j0 * T_CAP; ka.t_idx = (int*)tr_idx - (size_t)j0 * T_CAP; ka.t_cap = T_CAP;
    if (gang) {
        AS_HIP(d_xo.alloc(16 + 8 * 16));
        AS_HIP(hipMemsetAsync(d_xo, 0, sizeof(int) * (16 + 8 * 16), st));
        AS_HIP(hipMemcpyAsync(d_xo, hxoff, sizeof(int) * 9, hipMemcpyHostToDevice, st));
        ka.xoff = d_xo;
        ka.xcur = (int*)d_xo + 16;
    }
    ka.t_all = 1; ka.thr_col = col_thr;
    ka.thr0 = row_thr;   // the own rows' thresholds (their own-block lists' bounds): what a tighter start rejects lies beyond the row's M-th key
    AS_TRY(launch_k2(ka, metric, false, true, grid, st, op.i8));
    // own rows: as in knn_block
    BlockRefineArgs ra;
    ra.xa32 = sp->x32; ra.xa64 = sp->x64; ra.xb32 = cols->x32; ra.xb64 = cols->x64; ra.na64 = sp->n64; ra.nb64 = cols->n64;
    ra.d = sp->d; ra.dp = sp->dp; ra.r0 = r0; ra.r1 = r1; ra.col_goff = col_goff; ra.S = S; ra.M = M; ra.metric = metric;
    ra.c_key = ckey; ra.c_idx = cidx; ra.c_cnt = ccnt;
    ra.p_key = p_key; ra.p_dist = p_dist; ra.p_gy = p_gy; ra.p_idx = p_idx; ra.p_cnt = p_cnt; ra.p_t32 = p_t32;
    const size_t per_wave = (sizeof(double) * 3 * M + sizeof(float) * ((size_t)S * M + M) + sizeof(int) * ((size_t)S * M + M) + 15) / 16 * 16;
    AS_HIP(hipFuncSetAttribute((const void*)knn_block_refine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(per_wave * 4)));
    hipLaunchKernelGGL(knn_block_refine_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), per_wave * 4, st, ra);
    AS_HIP(hipGetLastError());
    // visiting items: their transposed buffers -> one candidate list each (+ the bound of what was turned away) -> the same
    // refinement with the roles swapped (rows = the chunk's items, columns = the own rows, ids global)
    AS_HIP(c2key.alloc((size_t)ncc * M));
    AS_HIP(c2idx.alloc((size_t)ncc * M));
    AS_HIP(c2cnt.alloc(ncc));
    AS_HIP(t_bound.alloc(ncc));
    AS_HIP(gate.alloc(ncc));
    hipLaunchKernelGGL(pair_gate_kernel, dim3((unsigned)((ncc + 255) / 256)), dim3(256), 0, st, col_thr ? col_thr + j0 : nullptr, cols->n32 + j0, ncc,
                       (int64_t)0, ncc, metric, ka.epskey, ka.coef, ka.nmax, (float*)gate);
    const size_t ldst = (sizeof(float) + sizeof(int)) * 4 * (size_t)T_CAP;
    AS_HIP(hipFuncSetAttribute((const void*)transposed_compact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldst));
    hipLaunchKernelGGL(transposed_compact_kernel, dim3((unsigned)((ncc + 3) / 4)), dim3(256), ldst, st, (const int*)tr_cnt, (const float*)tr_key,
                       (const int*)tr_idx, T_CAP, ncc, 1, 0, M, (float*)c2key, (int*)c2idx, (int*)c2cnt, (const float*)gate, (float*)t_bound);
    AS_HIP(hipGetLastError());
    BlockRefineArgs rb2;
    rb2.xa32 = cols->x32; rb2.xa64 = cols->x64; rb2.xb32 = sp->x32; rb2.xb64 = sp->x64; rb2.na64 = cols->n64; rb2.nb64 = sp->n64;
    rb2.d = sp->d; rb2.dp = sp->dp; rb2.r0 = j0; rb2.r1 = j1; rb2.col_goff = row_goff; rb2.S = 1; rb2.M = M; rb2.metric = metric;
    rb2.c_key = c2key; rb2.c_idx = c2idx; rb2.c_cnt = c2cnt;
    rb2.p_key = q_key + (size_t)j0 * M; rb2.p_dist = q_dist + (size_t)j0 * M; rb2.p_gy = q_gy + (size_t)j0 * M; rb2.p_idx = q_idx + (size_t)j0 * M;
    rb2.p_cnt = q_cnt + j0; rb2.p_t32 = q_t32 + j0;
    rb2.t32_given = (const float*)t_bound - j0;   // read by the item's number inside the block
    const size_t per_wave2 = (sizeof(double) * 3 * M + sizeof(float) * ((size_t)M + M) + sizeof(int) * ((size_t)M + M) + 15) / 16 * 16;
    hipLaunchKernelGGL(knn_block_refine_kernel, dim3((unsigned)((ncc + 3) / 4)), dim3(256), per_wave2 * 4, st, rb2);
    AS_HIP(hipGetLastError());
    AS_HIP(hipStreamSynchronize(st));
    sp->kstats[7] += 2.0 * (double)nrb * BM * (double)ntile * BN * (double)sp->dp;
    return AS_OK;
}

as_status knn_block_pair(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t r0, int64_t r1, int64_t ct0,
                         int64_t ct1, int64_t row_goff, int64_t col_goff, const float* col_thr, const float* row_thr, int M, double* p_key, double* p_dist,
                         double* p_gy, int32_t* p_idx, int32_t* p_cnt, float* p_t32, double* q_key, double* q_dist, double* q_gy,
                         int32_t* q_idx, int32_t* q_cnt, float* q_t32) {
    AS_TRY(block_check(sp, cols, r0, r1, "as_knn_block_pair"));
    hipStream_t st = sp->stream;
    const int64_t rows = r1 - r0, nc = cols->n;
    const int64_t ntile_all = (nc + BN - 1) / BN;   // tiles that hold items (the padded layout may end in a tile of padding)
    if (ct0 < 0) ct0 = 0;
    if (ct1 < 0 || ct1 > ntile_all) ct1 = ntile_all;
    // the visiting items that take no part keep an empty slice
    AS_HIP(hipMemsetAsync(q_cnt, 0, sizeof(int32_t) * nc, st));
    AS_HIP(hipMemsetAsync(q_idx, 0xff, sizeof(int32_t) * nc * M, st));
    if (rows <= 0 || ct1 <= ct0) {
        if (rows > 0) AS_HIP(hipMemsetAsync(p_cnt, 0, sizeof(int32_t) * rows, st));   // an empty own slice too: folding it adds nothing
        AS_HIP(hipStreamSynchronize(st));
        return AS_OK;
    }
    // The transposed buffers take 16 M x 8 B per visiting item (65 GB for a shard of 8M items at M = 64): the visiting
    // block is taken in chunks of column tiles whose buffers fit in an eighth of the free memory (at most 16 GB).  A
    // visiting item belongs to one chunk -- its slice is complete after it; the own rows' slices of the chunks are folded
    // (the M smallest exact keys of their union, the smallest drop bound).
    int64_t chunk = ct1 - ct0;
    {
        size_t mfree = 0, mtotal = 0;
        double budget = 16e9;
        if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess) budget = std::min(budget, 0.125 * (double)mfree);
        const double per_tile = (double)BN * 16.0 * M * 8.0;
        chunk = std::max<int64_t>(8, std::min<int64_t>(chunk, (int64_t)(budget / per_tile)));
        const char* ev = getenv("ARROWSPACE_PAIR_CHUNK_TILES");   // tests: force the chunked path at small sizes
        if (ev && atoll(ev) > 0) chunk = atoll(ev);
    }
    const int64_t nchunk = (ct1 - ct0 + chunk - 1) / chunk;
    if (nchunk <= 1)
        return pair_chunk(sp, cols, gp, r0, r1, ct0, ct1, row_goff, col_goff, col_thr, row_thr, M, p_key, p_dist, p_gy, p_idx, p_cnt, p_t32, q_key, q_dist,
                          q_gy, q_idx, q_cnt, q_t32);
    chunk = (ct1 - ct0 + nchunk - 1) / nchunk;   // even pieces
    dev_tmp<double> c_key, c_dist, c_gy;
    dev_tmp<int32_t> c_idx, c_cnt;
    dev_tmp<float> c_t32;
    AS_HIP(c_key.alloc((size_t)rows * M));
    AS_HIP(c_dist.alloc((size_t)rows * M));
    AS_HIP(c_gy.alloc((size_t)rows * M));
    AS_HIP(c_idx.alloc((size_t)rows * M));
    AS_HIP(c_cnt.alloc(rows));
    AS_HIP(c_t32.alloc(rows));
    AS_HIP(hipMemsetAsync(p_cnt, 0, sizeof(int32_t) * rows, st));   // an empty running slice: no entries, nothing dropped
    for (int64_t ca = ct0; ca < ct1; ca += chunk) {
        const int64_t cb = std::min(ct1, ca + chunk);
        AS_TRY(pair_chunk(sp, cols, gp, r0, r1, ca, cb, row_goff, col_goff, col_thr, row_thr, M, c_key, c_dist, c_gy, c_idx, c_cnt, c_t32, q_key, q_dist,
                          q_gy, q_idx, q_cnt, q_t32));
        // NOT the fold's error term: the chunk's bound is a raw fp32 key, as a block slice's is -- the caller's fold
        // subtracts the error once.  Here only the lists are merged and the smaller raw bound is kept.
        AS_TRY(knn_fold_raw(sp, rows, M, p_key, p_dist, p_gy, p_idx, p_cnt, p_t32, c_key, c_dist, c_gy, c_idx, c_cnt, c_t32));
    }
    return AS_OK;
}

as_status knn_merge(const as_space* sp, const as_graph_params* gp, int64_t r0, int64_t r1, int nblocks, int M, const double* p_key,
                    const double* p_dist, const double* p_gy, const int32_t* p_idx, const int32_t* p_cnt, const float* p_t32,
                    const double* block_nmax_host, int32_t* out_idx, double* out_key, double* out_dist, double* out_gy,
                    int32_t* out_cnt, int32_t* flag, double* out_B, int64_t* nflagged) {
    const int64_t rows = r1 - r0;
    *nflagged = 0;
    if (rows == 0) return AS_OK;
    hipStream_t st = sp->stream;
    dev_tmp<double> nmax;
    dev_tmp<int> nflag;
    AS_HIP(nmax.alloc(std::max(nblocks, 1)));
    AS_HIP(nflag.alloc(1));
    if (nblocks > 0) AS_HIP(hipMemcpyAsync(nmax, block_nmax_host, sizeof(double) * nblocks, hipMemcpyHostToDevice, st));
    AS_HIP(hipMemsetAsync(nflag, 0, sizeof(int), st));
    MergeArgs ma;
    ma.folded = nblocks == 0 ? 1 : 0;
    if (nblocks == 0) nblocks = 1;
    ma.rows = rows; ma.k = gp->k; ma.nblocks = nblocks; ma.M = M; ma.metric = sp->opts.metric;
    ma.epskey = ma.metric == AS_METRIC_L2 ? gp->eps * gp->eps : gp->eps; ma.coef = err_coef(sp);
    ma.na64 = sp->n64; ma.r0 = r0;
    ma.p_key = p_key; ma.p_dist = p_dist; ma.p_gy = p_gy; ma.p_idx = p_idx; ma.p_cnt = p_cnt; ma.p_t32 = p_t32; ma.nmax = nmax;
    ma.out_idx = out_idx; ma.out_key = out_key; ma.out_dist = out_dist; ma.out_gy = out_gy; ma.out_cnt = out_cnt;
    ma.flag = flag; ma.nflag = nflag; ma.out_B = out_B;
    const size_t lds = 4 * (sizeof(double) + sizeof(int) * 2) * (size_t)nblocks * M;
    if (lds > 150 * 1024) {
        set_err("as_knn_merge: %d blocks x list width %d exceed the merge kernel's LDS", nblocks, M);
        return AS_EUNSUPPORTED;
    }
    AS_HIP(hipFuncSetAttribute((const void*)knn_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), lds, st, ma);
    AS_HIP(hipGetLastError());
    int nf = 0;
    AS_HIP(hipMemcpyAsync(&nf, nflag, sizeof(int), hipMemcpyDeviceToHost, st));
    AS_HIP(hipStreamSynchronize(st));
    *nflagged = nf;
    return AS_OK;
}

// second pass of one visiting block, for the rows the merge flagged: collect mode + exact evaluation of the band
as_status knn_block_band(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t r0, int64_t r1, int64_t row_goff,
                         int64_t col_goff, int M, int32_t* flag, const double* B, double* p_key, double* p_dist, double* p_gy,
                         int32_t* p_idx, int32_t* p_cnt, float* p_t32, int64_t* overflowed) {
    AS_TRY(block_check(sp, cols, r0, r1, "as_knn_block_band"));
    const int64_t rows = r1 - r0;
    *overflowed = 0;
    if (rows == 0) return AS_OK;
    hipStream_t st = sp->stream;
    const int metric = sp->opts.metric;
    const double coef = err_coef(sp);
    std::vector<int> hflag(rows);
    AS_HIP(hipMemcpy(hflag.data(), flag, sizeof(int) * rows, hipMemcpyDeviceToHost));
    std::vector<int> ids;
    for (int64_t lr = 0; lr < rows; ++lr)
        if (hflag[lr]) ids.push_back((int)lr);
    const int nf = (int)ids.size();
    if (nf == 0) return AS_OK;
    const int dev_cus = device_cus(sp->device);
    const int64_t nfp = ((int64_t)nf + BM - 1) / BM * BM;
    const int nrb2 = (int)(nfp / BM);
    const int ntile = (int)(cols->np / BN);
    const int S2 = (int)std::max<int64_t>(1, std::min<int64_t>(ntile, ((int64_t)1 << 27) / (nfp * CAP)));
    dev_tmp<int> d_ids, a_ids, c2idx, c2cnt, over, bidx;
    dev_tmp<float> xa, a_n32, a_inorm, a_thr, c2key, bkey;
    AS_HIP(d_ids.alloc(nf)); AS_HIP(a_ids.alloc(nfp)); AS_HIP(over.alloc(1));
    AS_HIP(xa.alloc((size_t)(nfp + BM) * sp->dp));
    AS_HIP(a_n32.alloc(nfp)); AS_HIP(a_inorm.alloc(nfp)); AS_HIP(a_thr.alloc(nfp));
    AS_HIP(c2key.alloc((size_t)nfp * S2 * CAP)); AS_HIP(c2idx.alloc((size_t)nfp * S2 * CAP)); AS_HIP(c2cnt.alloc((size_t)nfp * S2));
    const int g2 = std::min(nrb2 * S2, dev_cus);
    AS_HIP(bkey.alloc((size_t)g2 * BM * CAP)); AS_HIP(bidx.alloc((size_t)g2 * BM * CAP));
    AS_HIP(hipMemcpyAsync(d_ids, ids.data(), sizeof(int) * nf, hipMemcpyHostToDevice, st));
    AS_HIP(hipMemsetAsync(xa, 0, sizeof(float) * (size_t)(nfp + BM) * sp->dp, st));
    AS_HIP(hipMemsetAsync(over, 0, sizeof(int), st));
    const double nmax = std::max(sp->nmax, cols->nmax);
    RingOperand op;
    AS_TRY(ring_operand(sp, cols, &op));
    dev_tmp<float> a_fa;
    AS_HIP(a_fa.alloc(nfp));
    AS_HIP(hipMemsetAsync(a_fa, 0, sizeof(float) * nfp, st));
    // (the rows are gathered from the image the pass runs on: op.ld floats per row -- at most dp, what xa was sized and zeroed for)
    hipLaunchKernelGGL(band_gather_kernel, dim3((unsigned)nf), dim3(192), 0, st, op.a, sp->n32, sp->inorm32, sp->n64, op.ld, r0,
                       (const int*)d_ids, nf, B, metric, coef, nmax, (float*)xa, (float*)a_n32, (float*)a_inorm, (int*)a_ids, (float*)a_thr,
                       row_goff, op.fa, (float*)a_fa);
    AS_HIP(hipGetLastError());
    KnnArgs kb;
    kb.x32 = op.b; kb.n32 = cols->n32; kb.inorm32 = cols->inorm32; kb.n = cols->n; kb.dp = sp->dp; kb.r0 = 0; kb.r1 = nf;
    kb.nrb = nrb2; kb.S = S2; kb.ntile = ntile; kb.M = CAP; kb.metric = metric;
    kb.epskey = 0; kb.coef = 0; kb.nmax = 0;
    kb.buf_key = bkey; kb.buf_idx = bidx; kb.out_key = c2key; kb.out_idx = c2idx; kb.out_cnt = c2cnt;
    kb.xa = xa; kb.a_n32 = a_n32; kb.a_inorm32 = a_inorm; kb.a_ids = a_ids; kb.a_thr = a_thr; kb.row_goff = 0; kb.col_goff = col_goff;
    kb.ld = op.ld; kb.nslab = op.nslab; kb.fa = op.fb; kb.a_fa = op.i8 ? (const float*)a_fa : nullptr;
    AS_TRY(launch_k2(kb, metric, true, false, g2, st, op.i8));
    BlockBandArgs ba;
    ba.xa32 = sp->x32; ba.xa64 = sp->x64; ba.xb32 = cols->x32; ba.xb64 = cols->x64; ba.na64 = sp->n64; ba.nb64 = cols->n64;
    ba.d = sp->d; ba.dp = sp->dp; ba.r0 = r0; ba.col_goff = col_goff; ba.S = S2; ba.CW = CAP; ba.M = M; ba.metric = metric; ba.nf = nf;
    ba.ids = d_ids; ba.c_idx = c2idx; ba.c_cnt = c2cnt;
    ba.p_key = p_key; ba.p_dist = p_dist; ba.p_gy = p_gy; ba.p_idx = p_idx; ba.p_cnt = p_cnt; ba.p_t32 = p_t32; ba.overflow = over;
    ba.flag = flag;
    const size_t ldsb = (sizeof(double) * 3 + sizeof(int)) * (size_t)BAND_MAX;
    AS_HIP(hipFuncSetAttribute((const void*)knn_block_band_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipLaunchKernelGGL(knn_block_band_kernel, dim3((unsigned)nf), dim3(256), ldsb, st, ba);
    AS_HIP(hipGetLastError());
    int hov = 0;
    AS_HIP(hipMemcpyAsync(&hov, over, sizeof(int), hipMemcpyDeviceToHost, st));
    AS_HIP(hipStreamSynchronize(st));
    *overflowed = hov;
    return AS_OK;
}

// Third pass of one visiting block (last resort): rows whose band did not fit the collection buffers somewhere -- more
// rows inside the band than BAND_MAX, e.g. thousands of exact duplicates or of items at one and the same distance --
// take the block's k nearest inside eps by EXACT evaluation of every pair, ordered by (key64, global id): nothing is
// dropped that could matter, so the slice carries no bound.  One workgroup per flagged row, a wave per column, the
// waves' sorted k-lists merged at the end; row-serial cost (block rows x D per flagged row), as the single-space path's
// last resort is.
struct BlockExactArgs {
    const float *xa32, *xb32;
    const double *xa64, *xb64, *na64, *nb64;
    int64_t d, dp, r0, nb, row_goff, col_goff;
    int M, k, metric;
    double epskey;
    const int* ids;
    double *p_key, *p_dist, *p_gy;
    int32_t* p_idx;
    int32_t* p_cnt;
    float* p_t32;
};

constexpr int EXACT_K = 128;   // >= the widest k (120)

__global__ __launch_bounds__(256) void knn_block_exact_kernel(BlockExactArgs a) {
    __shared__ double sk[4][EXACT_K], sd[4][EXACT_K], sg[4][EXACT_K];
    __shared__ int si[4][EXACT_K];
    __shared__ int sc[4];
    const int w = threadIdx.x >> 6, lane = lane_id();
    const int lr = a.ids[blockIdx.x];
    const int64_t row = a.r0 + lr;
    const int64_t self = a.row_goff + row - a.col_goff;   // the row's own number inside this block, if it lives there
    const double ni = a.na64[row];
    const int k = a.k;
    int cnt = 0;
    for (int64_t j = w; j < a.nb; j += 4) {
        if (j == self) continue;
        double sq, dot;
        exact_pair2(a.xa32, a.xa64, a.xb32, a.xb64, a.d, a.dp, row, j, sq, dot);
        sq = bcast_lane(sq, 0);
        dot = bcast_lane(dot, 0);
        double key, dist, gy;
        if (a.metric == AS_METRIC_L2) {
            key = sq;
            dist = sqrt(sq);
            gy = dot;
        } else {
            const double den = sqrt(ni * a.nb64[j]);
            const double c = den > 0.0 ? dot / den : 0.0;
            key = cosine_distance(c);
            dist = key;
            gy = c;
        }
        if (!(key <= a.epskey)) continue;
        const int gid = (int)(a.col_goff + j);
        if (cnt == k && !lex_less<double>(key, gid, sk[w][k - 1], si[w][k - 1])) continue;
        if (lane == 0) {
            int pos = cnt < k ? cnt : k - 1;
            while (pos > 0 && lex_less<double>(key, gid, sk[w][pos - 1], si[w][pos - 1])) {
                sk[w][pos] = sk[w][pos - 1];
                sd[w][pos] = sd[w][pos - 1];
                sg[w][pos] = sg[w][pos - 1];
                si[w][pos] = si[w][pos - 1];
                --pos;
            }
            sk[w][pos] = key;
            sd[w][pos] = dist;
            sg[w][pos] = gy;
            si[w][pos] = gid;
        }
        if (cnt < k) ++cnt;
        AS_LDS_FENCE();
    }
    if (lane == 0) sc[w] = cnt;
    __syncthreads();
    const int c0 = sc[0], c1 = sc[1], c2 = sc[2], c3 = sc[3];
    const int C = c0 + c1 + c2 + c3;
    for (int t = threadIdx.x; t < C; t += blockDim.x) {
        const int ww = t < c0 ? 0 : (t < c0 + c1 ? 1 : (t < c0 + c1 + c2 ? 2 : 3));
        const int e = t - (ww == 0 ? 0 : (ww == 1 ? c0 : (ww == 2 ? c0 + c1 : c0 + c1 + c2)));
        const double kk = sk[ww][e];
        const int id = si[ww][e];
        int rank = 0;
        for (int v = 0; v < 4; ++v)
            for (int s2 = 0; s2 < sc[v]; ++s2) rank += lex_less<double>(sk[v][s2], si[v][s2], kk, id) ? 1 : 0;
        if (rank < k) {
            const size_t o = (size_t)lr * a.M + rank;
            a.p_key[o] = kk;
            a.p_dist[o] = sd[ww][e];
            a.p_gy[o] = sg[ww][e];
            a.p_idx[o] = id;
        }
    }
    if (threadIdx.x == 0) {
        a.p_cnt[lr] = C < k ? C : k;   // the block's k nearest inside eps, exact: no "dropped" flag
        a.p_t32[lr] = 0.0f;
    }
}

as_status knn_block_exact(const as_space* sp, const as_space* cols, const as_graph_params* gp, int64_t r0, int64_t r1, int64_t row_goff,
                          int64_t col_goff, int M, const int32_t* flag, double* p_key, double* p_dist, double* p_gy, int32_t* p_idx,
                          int32_t* p_cnt, float* p_t32) {
    AS_TRY(block_check(sp, cols, r0, r1, "as_knn_block_exact"));
    const int64_t rows = r1 - r0;
    if (rows == 0) return AS_OK;
    if (gp->k > EXACT_K || gp->k > M) {
        set_err("as_knn_block_exact: k=%lld exceeds the list width", (long long)gp->k);
        return AS_EUNSUPPORTED;
    }
    hipStream_t st = sp->stream;
    std::vector<int> hflag(rows);
    AS_HIP(hipMemcpy(hflag.data(), flag, sizeof(int) * rows, hipMemcpyDeviceToHost));
    std::vector<int> ids;
    for (int64_t lr = 0; lr < rows; ++lr)
        if (hflag[lr]) ids.push_back((int)lr);
    const int nf = (int)ids.size();
    if (nf == 0) return AS_OK;
    dev_tmp<int> d_ids;
    AS_HIP(d_ids.alloc(nf));
    AS_HIP(hipMemcpyAsync(d_ids, ids.data(), sizeof(int) * nf, hipMemcpyHostToDevice, st));
    BlockExactArgs ea;
    ea.xa32 = sp->x32; ea.xa64 = sp->x64; ea.xb32 = cols->x32; ea.xb64 = cols->x64; ea.na64 = sp->n64; ea.nb64 = cols->n64;
    ea.d = sp->d; ea.dp = sp->dp; ea.r0 = r0; ea.nb = cols->n; ea.row_goff = row_goff; ea.col_goff = col_goff;
    ea.M = M; ea.k = (int)gp->k; ea.metric = sp->opts.metric;
    ea.epskey = ea.metric == AS_METRIC_L2 ? gp->eps * gp->eps : gp->eps;
    ea.ids = d_ids;
    ea.p_key = p_key; ea.p_dist = p_dist; ea.p_gy = p_gy; ea.p_idx = p_idx; ea.p_cnt = p_cnt; ea.p_t32 = p_t32;
    hipLaunchKernelGGL(knn_block_exact_kernel, dim3((unsigned)nf), dim3(256), 0, st, ea);
    AS_HIP(hipGetLastError());
    AS_HIP(hipStreamSynchronize(st));
    if (sp == cols) sp->kstats[6] += nf;   // rows settled by the last resort (every rank sees its own block once per round)
    return AS_OK;
}

// ------------------------------------------------------------------ K3 symmetrise -> CSR
__global__ void sym_count_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ cnt, int64_t n, int64_t k,
                                 int* __restrict__ revcnt) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * k) return;
    const int64_t i = e / k;
    const int t = (int)(e % k);
    if (t >= cnt[i]) return;
    const int j = idx[e];
    const int cj = cnt[j];
    bool found = false;
    for (int s = 0; s < cj; ++s) found = found || idx[(int64_t)j * k + s] == (int)i;
    if (!found) atomicAdd(&revcnt[j], 1);
}

__global__ void rowlen_kernel(const int32_t* __restrict__ cnt, const int* __restrict__ revcnt, int64_t n, int64_t* __restrict__ len) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) len[i] = (int64_t)cnt[i] + revcnt[i];
}

// three-phase exclusive scan over int64 (1024 elements per block)
__global__ void scan_block_sums(const int64_t* __restrict__ in, int64_t n, int64_t* __restrict__ bsum) {
    __shared__ int64_t sh[256];
    const int64_t base = (int64_t)blockIdx.x * 1024;
    int64_t s = 0;
    for (int q = 0; q < 4; ++q) {
        const int64_t i = base + threadIdx.x * 4 + q;
        if (i < n) s += in[i];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0];
}
__global__ void scan_top(int64_t* bsum, int64_t nb, int64_t* total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int64_t run = 0;
        for (int64_t b = 0; b < nb; ++b) {
            const int64_t v = bsum[b];
            bsum[b] = run;
            run += v;
        }
        *total = run;
    }
}
__global__ void scan_apply(const int64_t* __restrict__ in, int64_t n, const int64_t* __restrict__ bsum, int64_t* __restrict__ out) {
    __shared__ int64_t sh[256];
    const int64_t base = (int64_t)blockIdx.x * 1024;
    int64_t v[4];
    int64_t s = 0;
    for (int q = 0; q < 4; ++q) {
        const int64_t i = base + threadIdx.x * 4 + q;
        v[q] = i < n ? in[i] : 0;
        s += v[q];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    // inclusive Hillis-Steele over 256 partials
    for (int o = 1; o < 256; o <<= 1) {
        int64_t t = 0;
        if ((int)threadIdx.x >= o) t = sh[threadIdx.x - o];
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int64_t run = bsum[blockIdx.x] + sh[threadIdx.x] - s;
    for (int q = 0; q < 4; ++q) {
        const int64_t i = base + threadIdx.x * 4 + q;
        if (i < n) out[i] = run;
        run += v[q];
    }
}

__global__ void sym_fill_kernel(const int32_t* __restrict__ idx, const double* __restrict__ dist, const double* __restrict__ gy,
                                const int32_t* __restrict__ cnt, int64_t n, int64_t k, const int64_t* __restrict__ indptr,
                                int* __restrict__ cursor, int32_t* __restrict__ t_col, double* __restrict__ t_dist,
                                double* __restrict__ t_gy) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * k) return;
    const int64_t i = e / k;
    const int t = (int)(e % k);
    if (t >= cnt[i]) return;
    const int j = idx[e];
    int64_t pos = indptr[i] + t;
    t_col[pos] = j;
    t_dist[pos] = dist[e];
    t_gy[pos] = gy[e];
    const int cj = cnt[j];
    bool found = false;
    for (int s = 0; s < cj; ++s) found = found || idx[(int64_t)j * k + s] == (int)i;
    if (!found) {
        pos = indptr[j] + cj + atomicAdd(&cursor[j], 1);
        t_col[pos] = (int)i;
        t_dist[pos] = dist[e];
        t_gy[pos] = gy[e];
    }
}

// out-of-place rank sort of every row by column index (one wave per row); also edge weights
__global__ __launch_bounds__(256) void sym_sort_kernel(int64_t n, const int64_t* __restrict__ indptr, const int32_t* __restrict__ t_col,
                                                       const double* __restrict__ t_dist, const double* __restrict__ t_gy,
                                                       int32_t* __restrict__ col, double* __restrict__ dist, double* __restrict__ gy,
                                                       double* __restrict__ wgt, double sigma, double p, int kernel) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = lane_id();
    const int64_t lo = indptr[row], hi = indptr[row + 1];
    const int64_t len = hi - lo;
    for (int64_t t = lane; t < len; t += 64) {
        const int c = t_col[lo + t];
        int64_t rank = 0;
        for (int64_t s = 0; s < len; ++s) rank += t_col[lo + s] < c ? 1 : 0;
        const double dd = t_dist[lo + t];
        col[lo + rank] = c;
        dist[lo + rank] = dd;
        gy[lo + rank] = t_gy[lo + t];
        wgt[lo + rank] = edge_weight(dd, sigma, p, kernel);
    }
}

// ------------------------------------------------------------------ K4 degrees, K5 energies
__global__ void degree_kernel(int64_t n, const int64_t* __restrict__ indptr, const double* __restrict__ wgt, double* __restrict__ deg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) s += wgt[e];
    deg[i] = s;
}

__global__ void energy_kernel(int64_t n, const int64_t* __restrict__ indptr, const int32_t* __restrict__ col,
                              const double* __restrict__ wgt, const double* __restrict__ dist, const double* __restrict__ gy,
                              const double* __restrict__ deg,
                              const double* __restrict__ n64, int metric, double* __restrict__ ny, double* __restrict__ lap,
                              double* __restrict__ E, double* __restrict__ G, int64_t goff) {
    // rows [goff, goff + n) of the graph: deg and n64 are indexed by item id (goff = 0: the whole graph), the outputs by row
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double nyi = metric == AS_METRIC_L2 ? n64[i + goff] : (n64[i + goff] > 0.0 ? 1.0 : 0.0);
    ny[i] = nyi;
    const int64_t lo = indptr[i], hi = indptr[i + 1];
    double Ei = 0.0, Gi = 0.0;
    if (hi > lo) {
        const double di = deg[i + goff];
        double S = 0.0;
        for (int64_t e = lo; e < hi; ++e) {
            const int j = col[e];
            const double nyj = metric == AS_METRIC_L2 ? n64[j] : (n64[j] > 0.0 ? 1.0 : 0.0);
            const double sdd = sqrt(di * deg[j]);
            lap[e] = -wgt[e] / sdd;
            S += edge_energy(wgt[e], metric, dist[e], gy[e], di, deg[j], nyi, nyj);
        }
        Ei = nyi > 0.0 ? (0.5 * S) / nyi : 0.0;
        if (S > 0.0) {
            double g = 0.0;
            for (int64_t e = lo; e < hi; ++e) {
                const int j = col[e];
                const double nyj = metric == AS_METRIC_L2 ? n64[j] : (n64[j] > 0.0 ? 1.0 : 0.0);
                const double r = edge_energy(wgt[e], metric, dist[e], gy[e], di, deg[j], nyi, nyj) / S;
                g += r * r;
            }
            Gi = g < 0.0 ? 0.0 : (g > 1.0 ? 1.0 : g);
        }
    }
    E[i] = Ei;
    G[i] = Gi;
}

// ------------------------------------------------------------------ K5b lower median of the positive energies
// 8 passes of an 8-bit radix select over the IEEE bits (positive doubles order as uint64).
struct SelState {
    unsigned long long prefix;  // bits decided so far (high part)
    long long rank;             // remaining rank inside the prefix bucket
    long long npos;
    unsigned int hist[256];
};

__global__ void sel_hist_kernel(const double* __restrict__ E, int64_t n, int pass, SelState* st) {
    __shared__ unsigned int sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const unsigned long long prefix = st->prefix;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = E[i];
        if (!(v > 0.0)) continue;
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        if (pass > 0 && (b >> (shift + 8)) != prefix) continue;
        atomicAdd(&sh[(b >> shift) & 255ull], 1u);
    }
    __syncthreads();
    if (sh[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], sh[threadIdx.x]);
}
__global__ void sel_pick_kernel(int pass, SelState* st) {
    if (threadIdx.x != 0) return;
    if (pass == 0) {
        long long tot = 0;
        for (int b = 0; b < 256; ++b) tot += st->hist[b];
        st->npos = tot;
        st->rank = tot > 0 ? (tot - 1) / 2 : 0;
        st->prefix = 0;
    }
    if (st->npos > 0) {
        long long run = 0;
        int b = 0;
        for (; b < 256; ++b) {
            if (run + (long long)st->hist[b] > st->rank) break;
            run += st->hist[b];
        }
        st->rank -= run;
        st->prefix = (st->prefix << 8) | (unsigned long long)b;
    }
    for (int b = 0; b < 256; ++b) st->hist[b] = 0;
}

__global__ void lambda_kernel(int64_t n, const double* __restrict__ E, const double* __restrict__ G, const SelState* st,
                              double* __restrict__ lam64, float* __restrict__ lam32, double* tau_out) {
    double tau0 = TAU_MIN;
    if (st->npos > 0) {
        tau0 = __longlong_as_double((long long)st->prefix);
        tau0 = tau0 < TAU_MIN ? TAU_MIN : (tau0 > 1.0 ? 1.0 : tau0);
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *tau_out = tau0;
    if (i >= n) return;
    const double l = tau0 * (E[i] / (E[i] + tau0)) + (1.0 - tau0) * G[i];
    lam64[i] = l;
    if (lam32) lam32[i] = (float)l;
}

// K3 + degrees: union symmetrisation of n x k directed lists into the CSR arrays of gr (indptr, indices, dist, gy, w,
// deg; lap is allocated, not filled).  Shared by the item graph (n items) and the feature graph (n = D columns).
as_status csr_from_knn(hipStream_t st, int64_t n, int64_t k, const int32_t* idx, const double* dist, const double* gy,
                       const int32_t* cnt, double sigma, double p, int kernel, as_graph* gr) {
    dev_tmp<int> revcnt;
    dev_tmp<int64_t> len, bsum;
    AS_HIP(revcnt.alloc(n * 2));
    int* cursor = revcnt + n;
    AS_HIP(hipMemsetAsync(revcnt, 0, sizeof(int) * n * 2, st));
    AS_HIP(len.alloc(n));
    const int64_t nb = (n + 1023) / 1024;
    AS_HIP(bsum.alloc(nb + 1));
    int64_t* total_d = bsum + nb;
    AS_HIP(hipMalloc(&gr->indptr, sizeof(int64_t) * (n + 1)));
    const unsigned ge = (unsigned)((n * k + 255) / 256), gn = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(sym_count_kernel, dim3(ge), dim3(256), 0, st, idx, cnt, n, k, revcnt);
    hipLaunchKernelGGL(rowlen_kernel, dim3(gn), dim3(256), 0, st, cnt, revcnt, n, len);
    hipLaunchKernelGGL(scan_block_sums, dim3((unsigned)nb), dim3(256), 0, st, len, n, bsum);
    hipLaunchKernelGGL(scan_top, dim3(1), dim3(64), 0, st, bsum, nb, total_d);
    hipLaunchKernelGGL(scan_apply, dim3((unsigned)nb), dim3(256), 0, st, len, n, bsum, gr->indptr);
    AS_HIP(hipGetLastError());
    int64_t nnz = 0;
    AS_HIP(hipMemcpyAsync(&nnz, total_d, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    AS_HIP(hipStreamSynchronize(st));
    AS_HIP(hipMemcpyAsync(gr->indptr + n, &nnz, sizeof(int64_t), hipMemcpyHostToDevice, st));
    gr->nnz = nnz;
    const int64_t na = std::max<int64_t>(nnz, 1);
    dev_tmp<int32_t> t_col;
    dev_tmp<double> t_dist, t_gy;
    AS_HIP(t_col.alloc(na));
    AS_HIP(t_dist.alloc(na));
    AS_HIP(t_gy.alloc(na));
    AS_HIP(hipMalloc(&gr->indices, sizeof(int32_t) * na));
    AS_HIP(hipMalloc(&gr->dist, sizeof(double) * na));
    AS_HIP(hipMalloc(&gr->gy, sizeof(double) * na));
    AS_HIP(hipMalloc(&gr->w, sizeof(double) * na));
    AS_HIP(hipMalloc(&gr->lap, sizeof(double) * na));
    AS_HIP(hipMalloc(&gr->deg, sizeof(double) * n));
    hipLaunchKernelGGL(sym_fill_kernel, dim3(ge), dim3(256), 0, st, idx, dist, gy, cnt, n, k, gr->indptr, cursor, t_col, t_dist, t_gy);
    hipLaunchKernelGGL(sym_sort_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, gr->indptr, t_col, t_dist, t_gy,
                       gr->indices, gr->dist, gr->gy, gr->w, sigma, p, kernel);
    hipLaunchKernelGGL(degree_kernel, dim3(gn), dim3(256), 0, st, n, gr->indptr, gr->w, gr->deg);
    AS_HIP(hipGetLastError());
    AS_HIP(hipStreamSynchronize(st));   // the temporaries die with this frame
    return AS_OK;
}

// ------------------------------------------------------------------ K3, row-sharded (SURVEY 8e "Symmetrise + Laplacian")
// This rank owns the graph rows [row0, row0 + n).  `in_*` are the directed edges (in_col -> row0 + in_row) whose TARGET
// lives here, from every rank (this one included): the host's variable-count all-to-all delivers them.  An incoming
// edge the row's own list already holds adds nothing (union symmetrisation); the others become reverse entries.
__global__ void shard_count_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ cnt, int64_t n, int64_t k, int64_t ncols,
                                   int64_t n_in, const int32_t* __restrict__ in_row, const int32_t* __restrict__ in_col,
                                   int* __restrict__ revcnt, int* __restrict__ bad) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_in) return;
    const int j = in_row[e], i = in_col[e];
    if (j < 0 || j >= n || i < 0 || i >= ncols) {
        *bad = 1;
        return;
    }
    const int cj = cnt[j];
    bool found = false;
    for (int s = 0; s < cj; ++s) found = found || idx[(int64_t)j * k + s] == i;
    if (!found) atomicAdd(&revcnt[j], 1);
}

__global__ void shard_fill_kernel(const int32_t* __restrict__ idx, const double* __restrict__ dist, const double* __restrict__ gy,
                                  const int32_t* __restrict__ cnt, int64_t n, int64_t k, int64_t ncols, int64_t n_in,
                                  const int32_t* __restrict__ in_row, const int32_t* __restrict__ in_col,
                                  const double* __restrict__ in_dist, const double* __restrict__ in_gy,
                                  const int64_t* __restrict__ indptr, int* __restrict__ cursor, int32_t* __restrict__ t_col,
                                  double* __restrict__ t_dist, double* __restrict__ t_gy, int* __restrict__ bad) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n * k) {   // the rows' own lists
        const int64_t i = e / k;
        const int t = (int)(e % k);
        if (t < cnt[i]) {
            const int c = idx[e];
            if (c < 0 || c >= ncols) *bad = 1;
            const int64_t pos = indptr[i] + t;
            t_col[pos] = c;
            t_dist[pos] = dist[e];
            t_gy[pos] = gy[e];
        }
    }
    if (e < n_in) {    // reverse entries
        const int j = in_row[e], i = in_col[e];
        if (j < 0 || j >= n || i < 0 || i >= ncols) return;
        const int cj = cnt[j];
        bool found = false;
        for (int s = 0; s < cj; ++s) found = found || idx[(int64_t)j * k + s] == i;
        if (!found) {
            const int64_t pos = indptr[j] + cj + atomicAdd(&cursor[j], 1);
            t_col[pos] = i;
            t_dist[pos] = in_dist[e];
            t_gy[pos] = in_gy[e];
        }
    }
}

as_status graph_shard_csr(as_space* sp, const as_graph_params* gp, int64_t n_global, int64_t row_offset, const int32_t* idx,
                          const double* dist, const double* gy, const int32_t* cnt, int64_t n_in, const int32_t* in_row,
                          const int32_t* in_col, const double* in_dist, const double* in_gy, as_graph* gr) {
    const int64_t n = sp->n, k = gp->k;
    hipStream_t st = sp->stream;
    if (row_offset < 0 || row_offset + n > n_global || n_global >= ((int64_t)1 << 31) || n_in < 0) {
        set_err("as_graph_shard_csr: rows [%lld, %lld) are outside the %lld items", (long long)row_offset, (long long)(row_offset + n),
                (long long)n_global);
        return AS_EINVAL;
    }
    gr->n = n;
    gr->ncols = n_global;
    gr->row0 = row_offset;
    gr->device = sp->device;
    gr->gp = *gp;
    gr->metric = sp->opts.metric;
    gr->kernel = sp->opts.kernel;
    dev_tmp<int> revcnt;
    dev_tmp<int64_t> len, bsum;
    AS_HIP(revcnt.alloc(n * 2 + 1));
    int* cursor = revcnt + n;
    int* bad = revcnt + 2 * n;
    AS_HIP(hipMemsetAsync(revcnt, 0, sizeof(int) * (n * 2 + 1), st));
    AS_HIP(len.alloc(n));
    const int64_t nb = (n + 1023) / 1024;
    AS_HIP(bsum.alloc(nb + 1));
    int64_t* total_d = bsum + nb;
    AS_HIP(hipMalloc(&gr->indptr, sizeof(int64_t) * (n + 1)));
    const unsigned gi = (unsigned)((std::max<int64_t>(n_in, 1) + 255) / 256), gn = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(shard_count_kernel, dim3(gi), dim3(256), 0, st, idx, cnt, n, k, n_global, n_in, in_row, in_col, revcnt, bad);
    hipLaunchKernelGGL(rowlen_kernel, dim3(gn), dim3(256), 0, st, cnt, revcnt, n, len);
    hipLaunchKernelGGL(scan_block_sums, dim3((unsigned)nb), dim3(256), 0, st, len, n, bsum);
    hipLaunchKernelGGL(scan_top, dim3(1), dim3(64), 0, st, bsum, nb, total_d);
    hipLaunchKernelGGL(scan_apply, dim3((unsigned)nb), dim3(256), 0, st, len, n, bsum, gr->indptr);
    AS_HIP(hipGetLastError());
    int64_t nnz = 0;
    AS_HIP(hipMemcpyAsync(&nnz, total_d, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    AS_HIP(hipStreamSynchronize(st));
    AS_HIP(hipMemcpyAsync(gr->indptr + n, &nnz, sizeof(int64_t), hipMemcpyHostToDevice, st));
    gr->nnz = nnz;
    const int64_t na = std::max<int64_t>(nnz, 1);
    dev_tmp<int32_t> t_col;
    dev_tmp<double> t_dist, t_gy;
    AS_HIP(t_col.alloc(na));
    AS_HIP(t_dist.alloc(na));
    AS_HIP(t_gy.alloc(na));
    AS_HIP(hipMalloc(&gr->indices, sizeof(int32_t) * na));
    AS_HIP(hipMalloc(&gr->dist, sizeof(double) * na));
    AS_HIP(hipMalloc(&gr->gy, sizeof(double) * na));
    AS_HIP(hipMalloc(&gr->w, sizeof(double) * na));
    AS_HIP(hipMalloc(&gr->lap, sizeof(double) * na));
    AS_HIP(hipMalloc(&gr->deg, sizeof(double) * n));
    const unsigned gf = (unsigned)((std::max<int64_t>(std::max<int64_t>(n * k, n_in), 1) + 255) / 256);
    hipLaunchKernelGGL(shard_fill_kernel, dim3(gf), dim3(256), 0, st, idx, dist, gy, cnt, n, k, n_global, n_in, in_row, in_col, in_dist, in_gy,
                       gr->indptr, cursor, t_col, t_dist, t_gy, bad);
    hipLaunchKernelGGL(sym_sort_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, gr->indptr, t_col, t_dist, t_gy,
                       gr->indices, gr->dist, gr->gy, gr->w, gp->sigma, gp->p, gr->kernel);
    hipLaunchKernelGGL(degree_kernel, dim3(gn), dim3(256), 0, st, n, gr->indptr, gr->w, gr->deg);
    AS_HIP(hipGetLastError());
    int hbad = 0;
    AS_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    AS_HIP(hipStreamSynchronize(st));   // the temporaries die with this frame
    if (hbad) {
        set_err("as_graph_shard_csr: an edge names a row outside this shard or an item outside the %lld items", (long long)n_global);
        return AS_EINVAL;
    }
    sp->row_offset = row_offset;
    return AS_OK;
}

// Energies of this shard's rows: the degrees and squared norms of ALL items (all-gathered by the host, 16 B per item --
// cheaper than per-edge replies, nnz > N) give the neighbours' terms.
as_status graph_shard_energy(as_space* sp, as_graph* gr, const double* deg_global, const double* n64_global) {
    const int64_t n = gr->n;
    hipStream_t st = sp->stream;
    if (!gr->ny) AS_HIP(hipMalloc(&gr->ny, sizeof(double) * n));
    if (!gr->E) AS_HIP(hipMalloc(&gr->E, sizeof(double) * n));
    if (!gr->G) AS_HIP(hipMalloc(&gr->G, sizeof(double) * n));
    hipLaunchKernelGGL(energy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, gr->indptr, gr->indices, gr->w, gr->dist, gr->gy,
                       deg_global, n64_global, gr->metric, gr->ny, gr->lap, gr->E, gr->G, gr->row0);
    AS_HIP(hipGetLastError());
    AS_HIP(hipStreamSynchronize(st));
    return AS_OK;
}

__global__ void lambda_tau_kernel(int64_t n, const double* __restrict__ E, const double* __restrict__ G, double tau0,
                                  double* __restrict__ lam64, float* __restrict__ lam32) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double l = tau0 * (E[i] / (E[i] + tau0)) + (1.0 - tau0) * G[i];   // the expression of lambda_kernel, bit for bit
    lam64[i] = l;
    lam32[i] = (float)l;
}

// tau0 over the energies of ALL items (all-gathered, 8 B per item), then this shard's lambdas from its own E / G
as_status graph_shard_lambdas(as_space* sp, as_graph* gr, const double* E_global, int64_t n_global) {
    hipStream_t st = sp->stream;
    if (n_global != gr->ncols) {
        set_err("as_graph_shard_lambdas: %lld energies for a graph over %lld items", (long long)n_global, (long long)gr->ncols);
        return AS_EINVAL;
    }
    dev_tmp<double> lam;
    AS_HIP(lam.alloc(n_global));
    // G does not enter the median: the selection runs over E alone (lam is scratch here)
    AS_TRY(median_lambda_n(st, n_global, E_global, E_global, lam, nullptr, &gr->tau0));
    hipLaunchKernelGGL(lambda_tau_kernel, dim3((unsigned)((gr->n + 255) / 256)), dim3(256), 0, st, gr->n, gr->E, gr->G, gr->tau0, sp->lam64,
                       sp->lam32);
    AS_HIP(hipGetLastError());
    AS_HIP(hipStreamSynchronize(st));
    dbg("graph (rows [%lld, %lld) of %lld): nnz=%lld tau0=%.6g", (long long)gr->row0, (long long)(gr->row0 + gr->n), (long long)gr->ncols,
        (long long)gr->nnz, gr->tau0);
    return AS_OK;
}

// S8 + S9: tau0 = lower median of the positive energies (8-pass radix select), lambdas of n nodes
as_status median_lambda_n(hipStream_t st, int64_t n, const double* E, const double* G, double* lam64, float* lam32, double* tau0_out) {
    const unsigned gn = (unsigned)((n + 255) / 256);
    dev_tmp<SelState> sel;
    AS_HIP(sel.alloc(2));   // the selection state, then the slot tau0 is written to
    double* tau_d = (double*)(sel + 1);
    AS_HIP(hipMemsetAsync(sel, 0, 2 * sizeof(SelState), st));
    const unsigned gh = (unsigned)std::min<int64_t>((n + 255) / 256, 1024);
    for (int pass = 0; pass < 8; ++pass) {
        hipLaunchKernelGGL(sel_hist_kernel, dim3(gh), dim3(256), 0, st, E, n, pass, sel);
        hipLaunchKernelGGL(sel_pick_kernel, dim3(1), dim3(64), 0, st, pass, sel);
    }
    hipLaunchKernelGGL(lambda_kernel, dim3(gn), dim3(256), 0, st, n, E, G, sel, lam64, lam32, tau_d);
    AS_HIP(hipGetLastError());
    AS_HIP(hipMemcpyAsync(tau0_out, tau_d, sizeof(double), hipMemcpyDeviceToHost, st));
    AS_HIP(hipStreamSynchronize(st));
    return AS_OK;
}
as_status median_lambda(as_space* sp, as_graph* gr, const double* E, const double* G) {
    return median_lambda_n(sp->stream, sp->n, E, G, sp->lam64, sp->lam32, &gr->tau0);
}

__global__ void lam_slice_kernel(int64_t n, const double* __restrict__ src, double* __restrict__ lam64, float* __restrict__ lam32) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    lam64[i] = src[i];
    lam32[i] = (float)src[i];
}

as_status lam_slice(hipStream_t st, int64_t n, const double* src, double* lam64, float* lam32) {
    hipLaunchKernelGGL(lam_slice_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, src, lam64, lam32);
    AS_HIP(hipGetLastError());
    return AS_OK;
}

// Graph stage over the lists of ALL n_global items (all-gathered by the host) for a space that holds only the rows
// [row_offset, row_offset + sp->n): symmetrisation, Laplacian, energies and tau0 over the global graph, this
// shard's slice of the lambdas into the space.
as_status graph_from_knn_global(as_space* sp, const as_graph_params* gp, int64_t n_global, int64_t row_offset, const int32_t* idx,
                                const double* dist, const double* gy, const int32_t* cnt, const double* n64_global, as_graph* gr) {
    const int64_t n = n_global, k = gp->k;
    hipStream_t st = sp->stream;
    if (row_offset < 0 || row_offset + sp->n > n_global) {
        set_err("as_graph_from_knn_global: rows [%lld, %lld) are outside the %lld items", (long long)row_offset,
                (long long)(row_offset + sp->n), (long long)n_global);
        return AS_EINVAL;
    }
    gr->n = n;
    gr->device = sp->device;
    gr->gp = *gp;
    gr->metric = sp->opts.metric;
    gr->kernel = sp->opts.kernel;
    AS_TRY(csr_from_knn(st, n, k, idx, dist, gy, cnt, gp->sigma, gp->p, gr->kernel, gr));
    const unsigned gn = (unsigned)((n + 255) / 256);
    AS_HIP(hipMalloc(&gr->ny, sizeof(double) * n));
    AS_HIP(hipMalloc(&gr->E, sizeof(double) * n));
    AS_HIP(hipMalloc(&gr->G, sizeof(double) * n));
    hipLaunchKernelGGL(energy_kernel, dim3(gn), dim3(256), 0, st, n, gr->indptr, gr->indices, gr->w, gr->dist, gr->gy, gr->deg, n64_global,
                       gr->metric, gr->ny, gr->lap, gr->E, gr->G, (int64_t)0);
    AS_HIP(hipGetLastError());
    dev_tmp<double> lam;
    AS_HIP(lam.alloc(n));
    AS_TRY(median_lambda_n(st, n, gr->E, gr->G, lam, nullptr, &gr->tau0));
    hipLaunchKernelGGL(lam_slice_kernel, dim3((unsigned)((sp->n + 255) / 256)), dim3(256), 0, st, sp->n, (const double*)lam + row_offset,
                       sp->lam64, sp->lam32);
    AS_HIP(hipGetLastError());
    AS_HIP(hipStreamSynchronize(st));
    sp->row_offset = row_offset;
    dbg("graph (global, %lld nodes; this shard rows [%lld, %lld)): nnz=%lld tau0=%.6g", (long long)n, (long long)row_offset,
        (long long)(row_offset + sp->n), (long long)gr->nnz, gr->tau0);
    return AS_OK;
}

as_status graph_from_knn(as_space* sp, const as_graph_params* gp, const int32_t* idx, const double* dist, const double* gy,
                         const int32_t* cnt, as_graph* gr) {
    const int64_t n = sp->n, k = gp->k;
    hipStream_t st = sp->stream;
    gr->n = n;
    gr->device = sp->device;
    gr->gp = *gp;
    gr->metric = sp->opts.metric;
    gr->kernel = sp->opts.kernel;
    AS_TRY(csr_from_knn(st, n, k, idx, dist, gy, cnt, gp->sigma, gp->p, gr->kernel, gr));
    const unsigned gn = (unsigned)((n + 255) / 256);
    AS_HIP(hipMalloc(&gr->ny, sizeof(double) * n));
    AS_HIP(hipMalloc(&gr->E, sizeof(double) * n));
    AS_HIP(hipMalloc(&gr->G, sizeof(double) * n));
    hipLaunchKernelGGL(energy_kernel, dim3(gn), dim3(256), 0, st, n, gr->indptr, gr->indices, gr->w, gr->dist, gr->gy, gr->deg, sp->n64,
                       gr->metric, gr->ny, gr->lap, gr->E, gr->G, (int64_t)0);
    AS_HIP(hipGetLastError());
    AS_TRY(median_lambda(sp, gr, gr->E, gr->G));
    dbg("graph: nnz=%lld tau0=%.6g", (long long)gr->nnz, gr->tau0);
    return AS_OK;
}

}  // namespace as
