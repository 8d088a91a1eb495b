/* fp64 C/OpenMP restatement of the hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * The checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.  Same algorithm as
 * oracle/oracle_np.py (SPEC = DESIGN.md section 2), brute force, fp64.
 *
 * PARITY STATUS: "parity unpinned" for graph topology / lambda values: the
 * reference's arithmetic lives in crates.io `arrowspace 0.18.0`
 * (/root/reference/Cargo.toml:16, Cargo.lock:94-97), absent from the tree and
 * unbuildable here (no cargo).  Pinned by the reference and checked in
 * tests/test_oracle_golden.py: README.md:37-48,56-62,69 (3x3 toy scores, tau=1),
 * tests/test_0.py:4-18,24,29-32 (tau=1 order), TAUMODE.md:33 + src/lib.rs:169-173
 * (scorer form, topk results sorted descending over a full scan).
 *
 * Build: gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC (oracle/Makefile); the all-pairs tile kernel also has an
 * AVX-512 build chosen at run time.  No -ffast-math, and no FMA contraction in the pair sums.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ASO_TAU_MIN 1e-12
enum { ASO_L2 = 0, ASO_COSINE = 1 };
enum { ASO_GAUSSIAN = 0, ASO_RATIONAL = 1 };

typedef struct {
    int64_t n, d;
    double eps, p, sigma;
    int64_t k;
    int metric, kernel;
    double *X;       /* n*d */
    double *nrm;     /* n: squared norms */
    double *ny;      /* n: squared norm of the graph-space vector y_i */
    int64_t *indptr; /* n+1 */
    int64_t *indices;
    double *dist, *gy, *w, *lap;
    double *deg, *E, *G, *lam;
    double tau0;
    int64_t *knn_idx; /* n*k directed lists (-1 padded) */
    int64_t *knn_cnt;
    /* feature mode (SPEC F1-F7): the graph arrays above then describe the D-node feature graph */
    int fmode;
    int64_t nnodes;   /* n (item mode) or d (feature mode) */
    int64_t ne;       /* edges a < b */
    int64_t *ea, *eb;
    double *ew;
    double *colm;     /* d: squared column norms */
} aso_index;

static double edge_weight(double d, double sigma, double p, int kernel) {
    /* SPEC S4; GRAPH_VARIABLES.md:9 for the rational form */
    double t = d / sigma;
    double u = (p == 2.0) ? t * t : pow(t, p);
    return kernel == ASO_GAUSSIAN ? exp(-0.5 * u) : 1.0 / (1.0 + u);
}

/* sequential-order dot products (no -ffast-math): sum_c a_c*b_c and sum_c (a_c-b_c)^2; every product is rounded
 * before it is added (no FMA contraction), so that the tile kernel of the build below gives the same bits */
#define ASO_NOFMA __attribute__((optimize("fp-contract=off")))
ASO_NOFMA static inline void pair_l2(const double *a, const double *b, int64_t d, double *sq, double *dot) {
    double s = 0.0, g = 0.0;
    for (int64_t c = 0; c < d; ++c) {
        double t = a[c] - b[c];
        s += t * t;
        g += a[c] * b[c];
    }
    *sq = s;
    *dot = g;
}
ASO_NOFMA static inline double dotp(const double *a, const double *b, int64_t d) {
    double g = 0.0;
    for (int64_t c = 0; c < d; ++c) g += a[c] * b[c];
    return g;
}

/* SPEC S7 edge energy a (A + B - 2C) of the expanded form: a value inside the rounding noise of its own terms is
 * zero (identical vectors with equal degrees must not turn into a 1e-16 "energy" whose share of the sum is 1) */
#define ASO_ENERGY_NOISE 0x1p-46
/* no FMA contraction here: the numpy restatement rounds every operation, and tau = 0 rankings turn on the last bit */
__attribute__((optimize("fp-contract=off"), noinline))
static double edge_energy(double w, int metric, double dist, double g, double di, double dj, double nyi, double nyj) {
    /* w ||y_i/sqrt(d_i) - y_j/sqrt(d_j)||^2 in the form that does not cancel for near-identical neighbours
     * (oracle_np.edge_energy, same operation order) */
    double alpha = 1.0 / sqrt(di), beta = 1.0 / sqrt(dj), core;
    if (metric == ASO_L2) core = alpha * beta * (dist * dist) + (alpha - beta) * (alpha * nyi - beta * nyj);
    else if (nyi > 0.0 && nyj > 0.0) core = (alpha - beta) * (alpha - beta) + 2.0 * alpha * beta * (1.0 - g);
    else core = alpha * alpha * nyi + beta * beta * nyj;
    double v = w * core;
    double floor_ = w * ASO_ENERGY_NOISE * (alpha * alpha * nyi + beta * beta * nyj + 2.0 * alpha * beta * fabs(g));
    return v > floor_ ? v : 0.0;
}

/* SPEC S2: key (eps test + ordering), dist, gy for a pair, from its sums sum (a_c-b_c)^2 (L2 only) and sum a_c b_c */
static inline void pair_from_sums(double sq, double g, double na, double nb, int metric, double *key, double *dist, double *gy) {
    if (metric == ASO_L2) {
        *key = sq;
        *dist = sqrt(sq);
        *gy = g;
    } else {
        double den = sqrt(na * nb);
        double c = den > 0.0 ? g / den : 0.0;
        double dd = 1.0 - (c > 0.0 ? (c < 1.0 ? c : 1.0) : 0.0);   /* rounding can push a cosine past 1: no negative distances */
        *key = dd;
        *dist = dd;
        *gy = c;
    }
}
static inline void pair_q(const double *a, const double *b, int64_t d, double na, double nb, int metric,
                          double *key, double *dist, double *gy) {
    double sq = 0.0, g;
    if (metric == ASO_L2) pair_l2(a, b, d, &sq, &g);
    else g = dotp(a, b, d);
    pair_from_sums(sq, g, na, nb, metric, key, dist, gy);
}

/* ASO_NR rows a[0..] against a tile of ASO_TJ rows stored column-major (xt[c * ASO_TJ + u] = column c of the tile's
 * row u): the pair sums of pair_l2 (L2: the squared distance only -- the dot product of the few pairs inside eps is
 * taken by dotp afterwards) / dotp (cosine), lane u <-> the tile's row u, columns in order, products rounded.  Two
 * builds of the same arithmetic: 2 rows on 4-lane vectors (AVX2, the library's baseline ISA) and 4 rows on 8-lane
 * vectors where the CPU has AVX-512 (chosen at run time; the same bits lane by lane). */
#define ASO_TJ 8
#define ASO_IB 64
#define ASO_NR 4
typedef double aso_v4 __attribute__((vector_size(32)));
typedef double aso_v8 __attribute__((vector_size(64)));
ASO_NOFMA static void tile_pairs_v4(const double *const *a, int nr, const double *xt, int64_t d, int l2, double sq[ASO_NR][ASO_TJ],
                                    double g[ASO_NR][ASO_TJ]) {
    const aso_v4 z = {0.0, 0.0, 0.0, 0.0};
    for (int r0 = 0; r0 < nr; r0 += 2) {
        const double *a0 = a[r0], *a1 = a[r0 + 1 < nr ? r0 + 1 : r0];
        aso_v4 s00 = z, s01 = z, s10 = z, s11 = z, g00 = z, g01 = z, g10 = z, g11 = z;
        if (l2) {
            for (int64_t c = 0; c < d; ++c) {
                const aso_v4 b0 = *(const aso_v4 *)(xt + c * ASO_TJ), b1 = *(const aso_v4 *)(xt + c * ASO_TJ + 4);
                const aso_v4 x0 = {a0[c], a0[c], a0[c], a0[c]}, x1 = {a1[c], a1[c], a1[c], a1[c]};
                aso_v4 t;
                t = x0 - b0; s00 += t * t;
                t = x0 - b1; s01 += t * t;
                t = x1 - b0; s10 += t * t;
                t = x1 - b1; s11 += t * t;
            }
        } else {
            for (int64_t c = 0; c < d; ++c) {
                const aso_v4 b0 = *(const aso_v4 *)(xt + c * ASO_TJ), b1 = *(const aso_v4 *)(xt + c * ASO_TJ + 4);
                const aso_v4 x0 = {a0[c], a0[c], a0[c], a0[c]}, x1 = {a1[c], a1[c], a1[c], a1[c]};
                g00 += x0 * b0; g01 += x0 * b1; g10 += x1 * b0; g11 += x1 * b1;
            }
        }
        for (int u = 0; u < 4; ++u) {
            sq[r0][u] = s00[u]; sq[r0][u + 4] = s01[u];
            g[r0][u] = g00[u]; g[r0][u + 4] = g01[u];
            if (r0 + 1 < nr) {
                sq[r0 + 1][u] = s10[u]; sq[r0 + 1][u + 4] = s11[u];
                g[r0 + 1][u] = g10[u]; g[r0 + 1][u + 4] = g11[u];
            }
        }
    }
}
__attribute__((target("avx512f"), optimize("fp-contract=off")))
static void tile_pairs_v8(const double *const *a, int nr, const double *xt, int64_t d, int l2, double sq[ASO_NR][ASO_TJ],
                          double g[ASO_NR][ASO_TJ]) {
    const aso_v8 z = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const double *a0 = a[0], *a1 = a[nr > 1 ? 1 : 0], *a2 = a[nr > 2 ? 2 : 0], *a3 = a[nr > 3 ? 3 : 0];
    aso_v8 s0 = z, s1 = z, s2 = z, s3 = z, g0 = z, g1 = z, g2 = z, g3 = z;
    if (l2) {
        for (int64_t c = 0; c < d; ++c) {
            const aso_v8 b = *(const aso_v8 *)(xt + c * ASO_TJ);
            const aso_v8 x0 = z + a0[c], x1 = z + a1[c], x2 = z + a2[c], x3 = z + a3[c];
            aso_v8 t;
            t = x0 - b; s0 += t * t;
            t = x1 - b; s1 += t * t;
            t = x2 - b; s2 += t * t;
            t = x3 - b; s3 += t * t;
        }
    } else {
        for (int64_t c = 0; c < d; ++c) {
            const aso_v8 b = *(const aso_v8 *)(xt + c * ASO_TJ);
            g0 += (z + a0[c]) * b; g1 += (z + a1[c]) * b; g2 += (z + a2[c]) * b; g3 += (z + a3[c]) * b;
        }
    }
    for (int u = 0; u < ASO_TJ; ++u) {
        sq[0][u] = s0[u]; g[0][u] = g0[u];
        if (nr > 1) { sq[1][u] = s1[u]; g[1][u] = g1[u]; }
        if (nr > 2) { sq[2][u] = s2[u]; g[2][u] = g2[u]; }
        if (nr > 3) { sq[3][u] = s3[u]; g[3][u] = g3[u]; }
    }
}

typedef struct { double key; int64_t j; double dist, gy; } cand_t;

static inline int cand_less(double ka, int64_t ja, double kb, int64_t jb) {
    return ka < kb || (ka == kb && ja < jb);
}

/* keep the k smallest (key, j) in a sorted array */
static inline void cand_insert(cand_t *lst, int64_t *cnt, int64_t k, cand_t c) {
    int64_t m = *cnt;
    if (m == k) {
        if (!cand_less(c.key, c.j, lst[m - 1].key, lst[m - 1].j)) return;
        m = k - 1;
    }
    int64_t pos = m;
    while (pos > 0 && cand_less(c.key, c.j, lst[pos - 1].key, lst[pos - 1].j)) {
        lst[pos] = lst[pos - 1];
        --pos;
    }
    lst[pos] = c;
    *cnt = m + 1;
}

static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}
static int cmp_f64(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

void aso_free(aso_index *ix) {
    if (!ix) return;
    free(ix->X); free(ix->nrm); free(ix->ny); free(ix->indptr); free(ix->indices);
    free(ix->dist); free(ix->gy); free(ix->w); free(ix->lap); free(ix->deg);
    free(ix->E); free(ix->G); free(ix->lam); free(ix->knn_idx); free(ix->knn_cnt);
    free(ix->ea); free(ix->eb); free(ix->ew); free(ix->colm);
    free(ix);
}

/* ArrowSpaceBuilder.build restated (src/lib.rs:271-300 -> crate builder.build). */
aso_index *aso_build(const double *X, int64_t n, int64_t d, double eps, int64_t k, double p, double sigma,
                     int metric, int kernel) {
    if (n <= 0 || d <= 0 || k <= 0) return NULL;
    aso_index *ix = (aso_index *)calloc(1, sizeof(aso_index));
    ix->n = n; ix->d = d; ix->eps = eps; ix->k = k; ix->p = p; ix->sigma = sigma;
    ix->metric = metric; ix->kernel = kernel; ix->nnodes = n;
    ix->X = (double *)malloc(sizeof(double) * n * d);
    memcpy(ix->X, X, sizeof(double) * n * d);
    ix->nrm = (double *)malloc(sizeof(double) * n);
    ix->ny = (double *)malloc(sizeof(double) * n);
    for (int64_t i = 0; i < n; ++i) {
        ix->nrm[i] = dotp(X + i * d, X + i * d, d);
        ix->ny[i] = metric == ASO_L2 ? ix->nrm[i] : (ix->nrm[i] > 0.0 ? 1.0 : 0.0);
    }
    const double epskey = metric == ASO_L2 ? eps * eps : eps;
    /* S3 directed kNN lists */
    cand_t *lists = (cand_t *)malloc(sizeof(cand_t) * n * k);
    int64_t *cnt = (int64_t *)calloc(n, sizeof(int64_t));
    /* All pairs, cache-blocked: the items once more in tiles of 8 rows, column-major inside a tile, so that one row i
     * meets 8 rows j per step with the sums of every pair still running over the columns in order (the bits of
     * pair_l2 / dotp); a task owns ASO_IB rows i and streams the tiles past them (a tile stays in L1/L2 for all of
     * them).  Same candidates in the same order (j ascending) as the plain double loop. */
    const int64_t ntile = (n + ASO_TJ - 1) / ASO_TJ;
    const int wide = __builtin_cpu_supports("avx512f") && !getenv("ASO_NO_AVX512");
    double *Xt = (double *)aligned_alloc(64, sizeof(double) * (size_t)ntile * d * ASO_TJ);
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < ntile; ++t)
        for (int64_t c = 0; c < d; ++c)
            for (int u = 0; u < ASO_TJ; ++u) {
                int64_t j = t * ASO_TJ + u;
                Xt[((size_t)t * d + c) * ASO_TJ + u] = j < n ? X[j * d + c] : 0.0;
            }
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t ib = 0; ib < n; ib += ASO_IB) {
        const int64_t ie = ib + ASO_IB < n ? ib + ASO_IB : n;
        for (int64_t i = ib; i < ie; ++i) cnt[i] = 0;
        for (int64_t t = 0; t < ntile; ++t) {
            const double *xt = Xt + (size_t)t * d * ASO_TJ;
            for (int64_t i = ib; i < ie; i += ASO_NR) {
                const int nr = (int)(ie - i < ASO_NR ? ie - i : ASO_NR);
                double sq[ASO_NR][ASO_TJ], g[ASO_NR][ASO_TJ];
                const double *rows[ASO_NR];
                for (int r = 0; r < nr; ++r) rows[r] = X + (i + r) * d;
                if (wide) tile_pairs_v8(rows, nr, xt, d, metric == ASO_L2, sq, g);
                else tile_pairs_v4(rows, nr, xt, d, metric == ASO_L2, sq, g);
                for (int r = 0; r < nr; ++r) {
                    const int64_t ii = i + r;
                    for (int u = 0; u < ASO_TJ; ++u) {
                        const int64_t j = t * ASO_TJ + u;
                        if (j >= n || j == ii) continue;
                        cand_t c;
                        c.j = j;
                        if (metric == ASO_L2) {
                            if (!(sq[r][u] <= epskey)) continue;
                            g[r][u] = dotp(X + ii * d, X + j * d, d);
                        }
                        pair_from_sums(sq[r][u], g[r][u], ix->nrm[ii], ix->nrm[j], metric, &c.key, &c.dist, &c.gy);
                        if (c.key <= epskey) cand_insert(lists + ii * k, &cnt[ii], k, c);
                    }
                }
            }
        }
    }
    free(Xt);
    ix->knn_idx = (int64_t *)malloc(sizeof(int64_t) * n * k);
    ix->knn_cnt = cnt;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t t = 0; t < k; ++t) ix->knn_idx[i * k + t] = t < cnt[i] ? lists[i * k + t].j : -1;
    /* S4 union symmetrisation: count reverse-only edges */
    int64_t *rowlen = (int64_t *)calloc(n + 1, sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) rowlen[i] = cnt[i];
    for (int64_t i = 0; i < n; ++i)
        for (int64_t t = 0; t < cnt[i]; ++t) {
            int64_t j = lists[i * k + t].j;
            int found = 0;
            for (int64_t s = 0; s < cnt[j]; ++s)
                if (lists[j * k + s].j == i) { found = 1; break; }
            if (!found) rowlen[j] += 1;
        }
    ix->indptr = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    ix->indptr[0] = 0;
    for (int64_t i = 0; i < n; ++i) ix->indptr[i + 1] = ix->indptr[i] + rowlen[i];
    int64_t nnz = ix->indptr[n];
    ix->indices = (int64_t *)malloc(sizeof(int64_t) * (nnz ? nnz : 1));
    ix->dist = (double *)malloc(sizeof(double) * (nnz ? nnz : 1));
    ix->gy = (double *)malloc(sizeof(double) * (nnz ? nnz : 1));
    ix->w = (double *)malloc(sizeof(double) * (nnz ? nnz : 1));
    ix->lap = (double *)malloc(sizeof(double) * (nnz ? nnz : 1));
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * n);
    for (int64_t i = 0; i < n; ++i) cur[i] = ix->indptr[i];
    for (int64_t i = 0; i < n; ++i)
        for (int64_t t = 0; t < cnt[i]; ++t) ix->indices[cur[i]++] = lists[i * k + t].j;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t t = 0; t < cnt[i]; ++t) {
            int64_t j = lists[i * k + t].j;
            int found = 0;
            for (int64_t s = 0; s < cnt[j]; ++s)
                if (lists[j * k + s].j == i) { found = 1; break; }
            if (!found) ix->indices[cur[j]++] = i;
        }
    free(cur);
    /* sort each row by column; recompute the symmetric per-edge payload */
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; ++i) {
        int64_t lo = ix->indptr[i], hi = ix->indptr[i + 1];
        qsort(ix->indices + lo, (size_t)(hi - lo), sizeof(int64_t), cmp_i64);
        for (int64_t e = lo; e < hi; ++e) {
            int64_t j = ix->indices[e];
            double key;
            /* evaluate the pair with the smaller index first so (i,j) and (j,i) are bit-identical */
            int64_t a = i < j ? i : j, b = i < j ? j : i;
            pair_q(X + a * d, X + b * d, d, ix->nrm[a], ix->nrm[b], metric, &key, &ix->dist[e], &ix->gy[e]);
            ix->w[e] = edge_weight(ix->dist[e], sigma, p, kernel);
        }
    }
    free(rowlen);
    free(lists);
    /* S5 degrees */
    ix->deg = (double *)calloc(n, sizeof(double));
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int64_t e = ix->indptr[i]; e < ix->indptr[i + 1]; ++e) s += ix->w[e];
        ix->deg[i] = s;
    }
    /* S6/S7 energies */
    ix->E = (double *)calloc(n, sizeof(double));
    ix->G = (double *)calloc(n, sizeof(double));
    ix->lam = (double *)calloc(n, sizeof(double));
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; ++i) {
        int64_t lo = ix->indptr[i], hi = ix->indptr[i + 1];
        if (hi == lo) continue;
        double S = 0.0;
        for (int64_t e = lo; e < hi; ++e) {
            int64_t j = ix->indices[e];
            double sdd = sqrt(ix->deg[i] * ix->deg[j]);
            ix->lap[e] = -ix->w[e] / sdd;
            S += edge_energy(ix->w[e], ix->metric, ix->dist[e], ix->gy[e], ix->deg[i], ix->deg[j], ix->ny[i], ix->ny[j]);
        }
        ix->E[i] = ix->ny[i] > 0.0 ? (0.5 * S) / ix->ny[i] : 0.0;
        if (S > 0.0) {
            double g = 0.0;
            for (int64_t e = lo; e < hi; ++e) {
                int64_t j = ix->indices[e];
                double r = edge_energy(ix->w[e], ix->metric, ix->dist[e], ix->gy[e], ix->deg[i], ix->deg[j], ix->ny[i], ix->ny[j]) / S;
                g += r * r;
            }
            ix->G[i] = g < 0.0 ? 0.0 : (g > 1.0 ? 1.0 : g);
        }
    }
    /* S8 tau0 = lower median of positive energies, clamped */
    double *pos = (double *)malloc(sizeof(double) * n);
    int64_t np_ = 0;
    for (int64_t i = 0; i < n; ++i)
        if (ix->E[i] > 0.0) pos[np_++] = ix->E[i];
    double tau0 = ASO_TAU_MIN;
    if (np_ > 0) {
        qsort(pos, (size_t)np_, sizeof(double), cmp_f64);
        tau0 = pos[(np_ - 1) / 2];
        if (tau0 < ASO_TAU_MIN) tau0 = ASO_TAU_MIN;
        if (tau0 > 1.0) tau0 = 1.0;
    }
    free(pos);
    ix->tau0 = tau0;
    /* S9 */
    for (int64_t i = 0; i < n; ++i)
        ix->lam[i] = tau0 * (ix->E[i] / (ix->E[i] + tau0)) + (1.0 - tau0) * ix->G[i];
    return ix;
}

/* ---------------------------------------------------------------------------------------------
 * Feature mode (SPEC F1-F7, oracle_np.feature_graph / feature_energy): lambda = the synthetic index of
 * TAUMODE.md:8,12-27 on the F x F feature-space Laplacian of GRAPH_VARIABLES.md:17, whose nodes are the
 * D columns of the item matrix (same graph parameters, distance and kernel as GRAPH_VARIABLES.md:7-10). */

/* SPEC F6: (E, G) of one vector over the edge list a < b */
static void feature_energy(const aso_index *ix, const double *x, double *Eo, double *Go) {
    double T = 0.0, nx = 0.0;
    for (int64_t c = 0; c < ix->d; ++c) nx += x[c] * x[c];
    for (int64_t e = 0; e < ix->ne; ++e) {
        double t = x[ix->ea[e]] - x[ix->eb[e]];
        T += ix->ew[e] * t * t;
    }
    double G = 0.0;
    if (T > 0.0) {
        for (int64_t e = 0; e < ix->ne; ++e) {
            double t = x[ix->ea[e]] - x[ix->eb[e]];
            double r = ix->ew[e] * t * t / T;
            G += r * r;
        }
        G = G < 0.0 ? 0.0 : (G > 1.0 ? 1.0 : G);
    }
    *Eo = nx > 0.0 ? T / nx : 0.0;
    *Go = G;
}

aso_index *aso_build_feature(const double *X, int64_t n, int64_t d, double eps, int64_t k, double p, double sigma,
                             int metric, int kernel) {
    if (n <= 0 || d <= 0 || k <= 0) return NULL;
    aso_index *ix = (aso_index *)calloc(1, sizeof(aso_index));
    ix->n = n; ix->d = d; ix->eps = eps; ix->k = k; ix->p = p; ix->sigma = sigma;
    ix->metric = metric; ix->kernel = kernel; ix->fmode = 1; ix->nnodes = d;
    ix->X = (double *)malloc(sizeof(double) * n * d);
    memcpy(ix->X, X, sizeof(double) * n * d);
    ix->nrm = (double *)malloc(sizeof(double) * n);
    for (int64_t i = 0; i < n; ++i) ix->nrm[i] = dotp(X + i * d, X + i * d, d);
    /* F1: Gram of the columns (row-major accumulation over the items, fixed order) */
    double *gram = (double *)calloc((size_t)d * d, sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t a = 0; a < d; ++a)
        for (int64_t i = 0; i < n; ++i) {
            const double xa = X[i * d + a];
            const double *row = X + i * d;
            double *g = gram + a * d;
            for (int64_t b = a; b < d; ++b) g[b] += xa * row[b];
        }
    for (int64_t a = 0; a < d; ++a)
        for (int64_t b = 0; b < a; ++b) gram[a * d + b] = gram[b * d + a];
    ix->colm = (double *)malloc(sizeof(double) * d);
    for (int64_t a = 0; a < d; ++a) ix->colm[a] = gram[a * d + a];
    /* F2: key / dist per pair (stored in place of the Gram: key in gram, dist derived) */
    const double epskey = metric == ASO_L2 ? eps * eps : eps;
    cand_t *lists = (cand_t *)malloc(sizeof(cand_t) * d * k);
    int64_t *cnt = (int64_t *)calloc(d, sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t a = 0; a < d; ++a) {
        cand_t *lst = lists + a * k;
        int64_t m = 0;
        for (int64_t b = 0; b < d; ++b) {
            if (b == a) continue;
            cand_t c;
            c.j = b;
            c.gy = 0.0;
            const double g = gram[a * d + b], ma = ix->colm[a], mb = ix->colm[b];
            if (metric == ASO_L2) {
                double key = ma + mb - 2.0 * g;
                c.key = key > 0.0 ? key : 0.0;
                c.dist = sqrt(c.key);
            } else {
                double den = sqrt(ma * mb);
                double cs = den > 0.0 ? g / den : 0.0;
                c.key = 1.0 - (cs > 0.0 ? (cs < 1.0 ? cs : 1.0) : 0.0);
                c.dist = c.key;
            }
            if (c.key <= epskey) cand_insert(lst, &m, k, c);   /* F3: (key asc, b asc), first k */
        }
        cnt[a] = m;
    }
    free(gram);
    ix->knn_idx = (int64_t *)malloc(sizeof(int64_t) * d * k);
    ix->knn_cnt = cnt;
    for (int64_t a = 0; a < d; ++a)
        for (int64_t t = 0; t < k; ++t) ix->knn_idx[a * k + t] = t < cnt[a] ? lists[a * k + t].j : -1;
    /* F4: union symmetrisation (the pair's payload is symmetric: the Gram is) */
    int64_t *rowlen = (int64_t *)calloc(d + 1, sizeof(int64_t));
    for (int64_t a = 0; a < d; ++a) rowlen[a] = cnt[a];
    for (int64_t a = 0; a < d; ++a)
        for (int64_t t = 0; t < cnt[a]; ++t) {
            int64_t b = lists[a * k + t].j;
            int found = 0;
            for (int64_t s = 0; s < cnt[b]; ++s)
                if (lists[b * k + s].j == a) { found = 1; break; }
            if (!found) rowlen[b] += 1;
        }
    ix->indptr = (int64_t *)malloc(sizeof(int64_t) * (d + 1));
    ix->indptr[0] = 0;
    for (int64_t a = 0; a < d; ++a) ix->indptr[a + 1] = ix->indptr[a] + rowlen[a];
    const int64_t nnz = ix->indptr[d];
    const size_t na = (size_t)(nnz ? nnz : 1);
    ix->indices = (int64_t *)malloc(sizeof(int64_t) * na);
    ix->dist = (double *)malloc(sizeof(double) * na);
    ix->w = (double *)malloc(sizeof(double) * na);
    ix->lap = (double *)malloc(sizeof(double) * na);
    ix->gy = (double *)calloc(na, sizeof(double));
    cand_t *ent = (cand_t *)malloc(sizeof(cand_t) * na);
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * d);
    for (int64_t a = 0; a < d; ++a) cur[a] = ix->indptr[a];
    for (int64_t a = 0; a < d; ++a)
        for (int64_t t = 0; t < cnt[a]; ++t) ent[cur[a]++] = lists[a * k + t];
    for (int64_t a = 0; a < d; ++a)
        for (int64_t t = 0; t < cnt[a]; ++t) {
            int64_t b = lists[a * k + t].j;
            int found = 0;
            for (int64_t s = 0; s < cnt[b]; ++s)
                if (lists[b * k + s].j == a) { found = 1; break; }
            if (!found) {
                cand_t c = lists[a * k + t];
                c.j = a;
                ent[cur[b]++] = c;
            }
        }
    free(cur);
    free(rowlen);
    free(lists);
    /* sort rows by column (insertion sort: rows are short) and compute weights, degrees (F5) */
    ix->deg = (double *)calloc(d, sizeof(double));
    for (int64_t a = 0; a < d; ++a) {
        int64_t lo = ix->indptr[a], hi = ix->indptr[a + 1];
        for (int64_t u = lo + 1; u < hi; ++u) {
            cand_t c = ent[u];
            int64_t v = u;
            while (v > lo && ent[v - 1].j > c.j) { ent[v] = ent[v - 1]; --v; }
            ent[v] = c;
        }
        double s = 0.0;
        for (int64_t e = lo; e < hi; ++e) {
            ix->indices[e] = ent[e].j;
            ix->dist[e] = ent[e].dist;
            ix->w[e] = edge_weight(ent[e].dist, sigma, p, kernel);
            ix->lap[e] = -ix->w[e];
            s += ix->w[e];
        }
        ix->deg[a] = s;
    }
    free(ent);
    ix->ne = 0;
    ix->ea = (int64_t *)malloc(sizeof(int64_t) * na);
    ix->eb = (int64_t *)malloc(sizeof(int64_t) * na);
    ix->ew = (double *)malloc(sizeof(double) * na);
    for (int64_t a = 0; a < d; ++a)
        for (int64_t e = ix->indptr[a]; e < ix->indptr[a + 1]; ++e)
            if (a < ix->indices[e]) {
                ix->ea[ix->ne] = a;
                ix->eb[ix->ne] = ix->indices[e];
                ix->ew[ix->ne] = ix->w[e];
                ix->ne += 1;
            }
    /* F6 per item, F7 */
    ix->E = (double *)calloc(n, sizeof(double));
    ix->G = (double *)calloc(n, sizeof(double));
    ix->lam = (double *)calloc(n, sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) feature_energy(ix, X + i * d, &ix->E[i], &ix->G[i]);
    double *pos = (double *)malloc(sizeof(double) * n);
    int64_t np_ = 0;
    for (int64_t i = 0; i < n; ++i)
        if (ix->E[i] > 0.0) pos[np_++] = ix->E[i];
    double tau0 = ASO_TAU_MIN;
    if (np_ > 0) {
        qsort(pos, (size_t)np_, sizeof(double), cmp_f64);
        tau0 = pos[(np_ - 1) / 2];
        if (tau0 < ASO_TAU_MIN) tau0 = ASO_TAU_MIN;
        if (tau0 > 1.0) tau0 = 1.0;
    }
    free(pos);
    ix->tau0 = tau0;
    for (int64_t i = 0; i < n; ++i)
        ix->lam[i] = tau0 * (ix->E[i] / (ix->E[i] + tau0)) + (1.0 - tau0) * ix->G[i];
    return ix;
}

/* SPEC S10, second half: lambda of the query appended as a node with the m neighbours in lst (any order) */
static double lambda_from_list(const aso_index *ix, cand_t *lst, int64_t m, double nq) {
    double lam = 0.0;
    if (m > 0) {
        /* ascending-j order for every sum */
        for (int64_t a = 1; a < m; ++a) {
            cand_t c = lst[a];
            int64_t b = a;
            while (b > 0 && lst[b - 1].j > c.j) { lst[b] = lst[b - 1]; --b; }
            lst[b] = c;
        }
        double degq = 0.0;
        double *a = (double *)malloc(sizeof(double) * m);
        for (int64_t t = 0; t < m; ++t) {
            a[t] = edge_weight(lst[t].dist, ix->sigma, ix->p, ix->kernel);
            degq += a[t];
        }
        const double nyq = ix->metric == ASO_L2 ? nq : (nq > 0.0 ? 1.0 : 0.0);
        if (degq > 0.0 && nyq > 0.0) {
            double *es = (double *)malloc(sizeof(double) * m);
            double S = 0.0;
            for (int64_t t = 0; t < m; ++t) {
                int64_t j = lst[t].j;
                double dj = ix->deg[j] + a[t];
                es[t] = edge_energy(a[t], ix->metric, lst[t].dist, lst[t].gy, degq, dj, nyq, ix->ny[j]);
                S += es[t];
            }
            double Eq = 0.5 * S / nyq, Gq = 0.0;
            if (S > 0.0) {
                for (int64_t t = 0; t < m; ++t) { double r = es[t] / S; Gq += r * r; }
                Gq = Gq < 0.0 ? 0.0 : (Gq > 1.0 ? 1.0 : Gq);
            }
            lam = ix->tau0 * (Eq / (Eq + ix->tau0)) + (1.0 - ix->tau0) * Gq;
            free(es);
        }
        free(a);
    }
    return lam;
}

/* SPEC S10: prepare_query_item (src/lib.rs:154) restated. */
double aso_query_lambda(const aso_index *ix, const double *q) {
    if (ix->fmode) {
        double E, G;
        feature_energy(ix, q, &E, &G);
        return ix->tau0 * (E / (E + ix->tau0)) + (1.0 - ix->tau0) * G;
    }
    const int64_t n = ix->n, d = ix->d, k = ix->k;
    const double nq = dotp(q, q, d);
    const double epskey = ix->metric == ASO_L2 ? ix->eps * ix->eps : ix->eps;
    int nth = 1;
#ifdef _OPENMP
    nth = omp_get_max_threads();
#endif
    cand_t *part = (cand_t *)malloc(sizeof(cand_t) * k * nth);
    int64_t *pcnt = (int64_t *)calloc(nth, sizeof(int64_t));
#pragma omp parallel
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        cand_t *lst = part + (int64_t)t * k;
        int64_t m = 0;
#pragma omp for schedule(static)
        for (int64_t j = 0; j < n; ++j) {
            cand_t c;
            c.j = j;
            pair_q(q, ix->X + j * d, d, nq, ix->nrm[j], ix->metric, &c.key, &c.dist, &c.gy);
            if (c.key <= epskey) cand_insert(lst, &m, k, c);
        }
        pcnt[t] = m;
    }
    cand_t *lst = (cand_t *)malloc(sizeof(cand_t) * k);
    int64_t m = 0;
    for (int t = 0; t < nth; ++t)
        for (int64_t s = 0; s < pcnt[t]; ++s) cand_insert(lst, &m, k, part[(int64_t)t * k + s]);
    free(part);
    free(pcnt);
    double lam = lambda_from_list(ix, lst, m, nq);
    free(lst);
    return lam;
}

/* SPEC S11: search_lambda_aware (src/lib.rs:173, TAUMODE.md:33): full scan,
 * order (score desc, index asc), first topk.  Returns number of hits written. */
int64_t aso_search_with_lambda(const aso_index *ix, const double *q, double tau, double lambda_q, int64_t topk,
                               int64_t *out_idx, double *out_score) {
    const int64_t n = ix->n, d = ix->d;
    const double nq = dotp(q, q, d);
    if (topk > n) topk = n;
    int nth = 1;
#ifdef _OPENMP
    nth = omp_get_max_threads();
#endif
    cand_t *part = (cand_t *)malloc(sizeof(cand_t) * topk * nth);
    int64_t *pcnt = (int64_t *)calloc(nth, sizeof(int64_t));
#pragma omp parallel
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        cand_t *lst = part + (int64_t)t * topk;
        int64_t m = 0;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            double g = dotp(q, ix->X + i * d, d);
            double den = sqrt(ix->nrm[i] * nq);
            double c = den > 0.0 ? g / den : 0.0;
            double s = tau * c + (1.0 - tau) / (1.0 + fabs(lambda_q - ix->lam[i]));
            cand_t cd;
            cd.key = -s; cd.j = i; cd.dist = s; cd.gy = 0.0;
            cand_insert(lst, &m, topk, cd);
        }
        pcnt[t] = m;
    }
    cand_t *lst = (cand_t *)malloc(sizeof(cand_t) * topk);
    int64_t m = 0;
    for (int t = 0; t < nth; ++t)
        for (int64_t s = 0; s < pcnt[t]; ++s) cand_insert(lst, &m, topk, part[(int64_t)t * topk + s]);
    for (int64_t t = 0; t < m; ++t) { out_idx[t] = lst[t].j; out_score[t] = lst[t].dist; }
    free(part); free(pcnt); free(lst);
    return m;
}

/* ArrowSpace.search restated (src/lib.rs:132-174).  Returns hits written, or -1
 * when lambda_q == 0 (the reference asserts, src/lib.rs:156-159). */
int64_t aso_search(const aso_index *ix, const double *q, double tau, int64_t topk, int64_t *out_idx,
                   double *out_score, double *out_lambda_q) {
    double lq = aso_query_lambda(ix, q);
    if (out_lambda_q) *out_lambda_q = lq;
    if (lq == 0.0) return -1;
    return aso_search_with_lambda(ix, q, tau, lq, topk, out_idx, out_score);
}

/* The same search in ONE pass over the items (what a tuned CPU implementation does; bench.py's cpu_baseline):
 * the pass that finds the query's neighbours keeps every cosine, the scorer then runs over those N numbers.
 * Bit-identical results to aso_search. */
int64_t aso_search_fused(const aso_index *ix, const double *q, double tau, int64_t topk, int64_t *out_idx,
                         double *out_score, double *out_lambda_q) {
    if (ix->fmode) return aso_search(ix, q, tau, topk, out_idx, out_score, out_lambda_q);
    const int64_t n = ix->n, d = ix->d, k = ix->k;
    const double nq = dotp(q, q, d);
    const double epskey = ix->metric == ASO_L2 ? ix->eps * ix->eps : ix->eps;
    if (topk > n) topk = n;
    int nth = 1;
#ifdef _OPENMP
    nth = omp_get_max_threads();
#endif
    double *cosv = (double *)malloc(sizeof(double) * n);
    cand_t *part = (cand_t *)malloc(sizeof(cand_t) * k * nth);
    int64_t *pcnt = (int64_t *)calloc(nth, sizeof(int64_t));
#pragma omp parallel
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        cand_t *lst = part + (int64_t)t * k;
        int64_t m = 0;
#pragma omp for schedule(static)
        for (int64_t j = 0; j < n; ++j) {
            cand_t c;
            c.j = j;
            double g;
            if (ix->metric == ASO_L2) {
                pair_q(q, ix->X + j * d, d, nq, ix->nrm[j], ix->metric, &c.key, &c.dist, &c.gy);
                g = c.gy;
                double den = sqrt(ix->nrm[j] * nq);
                cosv[j] = den > 0.0 ? g / den : 0.0;
            } else {
                g = dotp(q, ix->X + j * d, d);
                double den = sqrt(nq * ix->nrm[j]);
                double cs = den > 0.0 ? g / den : 0.0;
                double den2 = sqrt(ix->nrm[j] * nq);
                cosv[j] = den2 > 0.0 ? g / den2 : 0.0;
                c.key = c.dist = 1.0 - (cs > 0.0 ? (cs < 1.0 ? cs : 1.0) : 0.0);
                c.gy = cs;
            }
            if (c.key <= epskey) cand_insert(lst, &m, k, c);
        }
        pcnt[t] = m;
    }
    cand_t *lst = (cand_t *)malloc(sizeof(cand_t) * k);
    int64_t m = 0;
    for (int t = 0; t < nth; ++t)
        for (int64_t s = 0; s < pcnt[t]; ++s) cand_insert(lst, &m, k, part[(int64_t)t * k + s]);
    free(part);
    const double lq = lambda_from_list(ix, lst, m, nq);
    free(lst);
    if (out_lambda_q) *out_lambda_q = lq;
    if (lq == 0.0) {
        free(cosv); free(pcnt);
        return -1;
    }
    cand_t *tpart = (cand_t *)malloc(sizeof(cand_t) * topk * nth);
#pragma omp parallel
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        cand_t *tl = tpart + (int64_t)t * topk;
        int64_t mm = 0;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            double sc = tau * cosv[i] + (1.0 - tau) / (1.0 + fabs(lq - ix->lam[i]));
            cand_t cd;
            cd.key = -sc; cd.j = i; cd.dist = sc; cd.gy = 0.0;
            cand_insert(tl, &mm, topk, cd);
        }
        pcnt[t] = mm;
    }
    cand_t *fl = (cand_t *)malloc(sizeof(cand_t) * topk);
    int64_t fm = 0;
    for (int t = 0; t < nth; ++t)
        for (int64_t s2 = 0; s2 < pcnt[t]; ++s2) cand_insert(fl, &fm, topk, tpart[(int64_t)t * topk + s2]);
    for (int64_t t = 0; t < fm; ++t) { out_idx[t] = fl[t].j; out_score[t] = fl[t].dist; }
    free(tpart); free(pcnt); free(fl); free(cosv);
    return fm;
}

/* all N scores (for parity checks of the scorer itself) */
void aso_scores(const aso_index *ix, const double *q, double tau, double lambda_q, double *out) {
    const int64_t n = ix->n, d = ix->d;
    const double nq = dotp(q, q, d);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double g = dotp(q, ix->X + i * d, d);
        double den = sqrt(ix->nrm[i] * nq);
        double c = den > 0.0 ? g / den : 0.0;
        out[i] = tau * c + (1.0 - tau) / (1.0 + fabs(lambda_q - ix->lam[i]));
    }
}

/* accessors for the ctypes wrapper */
int64_t aso_n(const aso_index *ix) { return ix->n; }
int64_t aso_d(const aso_index *ix) { return ix->d; }
int64_t aso_nnz(const aso_index *ix) { return ix->indptr[ix->nnodes]; }
int64_t aso_nnodes(const aso_index *ix) { return ix->nnodes; }
double aso_tau0(const aso_index *ix) { return ix->tau0; }
const int64_t *aso_indptr(const aso_index *ix) { return ix->indptr; }
const int64_t *aso_indices(const aso_index *ix) { return ix->indices; }
const double *aso_dist(const aso_index *ix) { return ix->dist; }
const double *aso_gy(const aso_index *ix) { return ix->gy; }
const double *aso_w(const aso_index *ix) { return ix->w; }
const double *aso_lap(const aso_index *ix) { return ix->lap; }
const double *aso_deg(const aso_index *ix) { return ix->deg; }
const double *aso_E(const aso_index *ix) { return ix->E; }
const double *aso_G(const aso_index *ix) { return ix->G; }
const double *aso_lambdas(const aso_index *ix) { return ix->lam; }
const double *aso_norms(const aso_index *ix) { return ix->nrm; }
const int64_t *aso_knn_idx(const aso_index *ix) { return ix->knn_idx; }
const int64_t *aso_knn_cnt(const aso_index *ix) { return ix->knn_cnt; }
int aso_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void aso_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* Search-only index from precomputed parts (bench.py cpu_baseline: the CPU scorer is
 * timed at full N without paying for an all-pairs CPU build).  The items are copied with the same static
 * schedule the scans use, so every thread first-touches (NUMA-places) the rows it will read. */
aso_index *aso_from_parts(double *X, int64_t n, int64_t d, double eps, int64_t k, double p, double sigma, int metric,
                          int kernel, const double *deg, const double *lam, double tau0) {
    aso_index *ix = (aso_index *)calloc(1, sizeof(aso_index));
    ix->n = n; ix->d = d; ix->eps = eps; ix->k = k; ix->p = p; ix->sigma = sigma;
    ix->metric = metric; ix->kernel = kernel; ix->tau0 = tau0; ix->nnodes = n;
    ix->X = (double *)malloc(sizeof(double) * n * d);
    ix->nrm = (double *)malloc(sizeof(double) * n);
    ix->ny = (double *)malloc(sizeof(double) * n);
    ix->deg = (double *)malloc(sizeof(double) * n);
    ix->lam = (double *)malloc(sizeof(double) * n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        memcpy(ix->X + i * d, X + i * d, sizeof(double) * d);
        ix->nrm[i] = dotp(X + i * d, X + i * d, d);
        ix->ny[i] = metric == ASO_L2 ? ix->nrm[i] : (ix->nrm[i] > 0.0 ? 1.0 : 0.0);
        ix->deg[i] = deg[i];
        ix->lam[i] = lam[i];
    }
    return ix;
}
void aso_free_parts(aso_index *ix) { aso_free(ix); }
