/* Sanitizer harness of the CPU checker (test infrastructure): builds item- and feature-mode indexes over odd shapes
 * (n not a multiple of the 8-row tiles, d = 1 ... 127, duplicates) and searches them under ASan + UBSan. */
#include "arrowspace_oracle.c"
#include <stdio.h>
int main(void) {
    int shapes[][3] = {{1, 3, 1}, {2, 1, 1}, {7, 5, 3}, {8, 8, 4}, {9, 33, 5}, {63, 2, 6}, {64, 7, 6}, {65, 127, 9}, {200, 24, 12}, {513, 16, 25}, {1000, 3, 30}};
    for (unsigned s = 0; s < sizeof(shapes) / sizeof(shapes[0]); ++s) {
        int n = shapes[s][0], d = shapes[s][1], k = shapes[s][2];
        double *X = (double *)malloc(sizeof(double) * n * d);
        unsigned long long st = 88172645463325252ULL + s;
        for (int i = 0; i < n * d; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; X[i] = (double)(st % 2001) / 1000.0 - 1.0; }
        if (n > 4) for (int c = 0; c < d; ++c) X[3 * d + c] = X[1 * d + c];   /* a duplicate */
        for (int metric = 0; metric < 2; ++metric) {
            aso_index *ix = aso_build(X, n, d, metric == 0 ? 1.5 : 0.6, k, 2.0, 0.7, metric, metric);
            double q[256];
            for (int c = 0; c < d; ++c) q[c] = X[c] * 1.01;
            int64_t idx[64]; double sc[64], lq;
            if (ix) {
                aso_search(ix, q, 0.62, k < 5 ? k : 5, idx, sc, &lq);
                aso_search_fused(ix, q, 0.62, k < 5 ? k : 5, idx, sc, &lq);
                aso_free(ix);
            }
            aso_index *fx = d >= 2 ? aso_build_feature(X, n, d, metric == 0 ? 30.0 : 0.9, k < d ? k : d - 1, 2.0, 0.7, metric, metric) : NULL;
            if (fx) aso_free(fx);
        }
        free(X);
    }
    printf("asan/ubsan run ok\n");
    return 0;
}
