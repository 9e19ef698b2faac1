/* A torch-free host for the C ABI: plain C, device memory through e2emv_malloc, one Sinkhorn + mutual-matching call.
 *
 *   gcc -std=c99 -Iinclude examples/sinkhorn_from_c.c -o sinkhorn_from_c \
 *       -Le2e_multi_view_matching_amd -le2emv -Wl,-rpath,$PWD/e2e_multi_view_matching_amd -lm
 *
 * Checks what can be checked without an oracle: every row / column of exp(logZ) (dustbins aside) carries the
 * prescribed marginal, and a planted permutation is recovered by the mutual arg-max.  Exit code 0 = pass.
 * (tests/test_gpu_c_example.py builds and runs it on the GPU box.) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "e2emv.h"

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != E2EMV_OK) {                                                                       \
            fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, ctx ? e2emv_last_error(ctx) : "");   \
            return 2;                                                                                \
        }                                                                                            \
    } while (0)

int main(void) {
    const int B = 2, M = 384, N = 320, iters = 100;
    e2emv_ctx* ctx = NULL;
    CHECK(e2emv_create(&ctx, 0));

    /* scores: a planted partial permutation (row i <-> column perm[i] for i < 256) on top of small noise */
    float* scores = (float*)malloc(sizeof(float) * B * M * N);
    int* perm = (int*)malloc(sizeof(int) * B * 256);
    unsigned s = 12345u;
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < M * N; ++i) {
            s = s * 1664525u + 1013904223u;
            scores[(size_t)b * M * N + i] = ((float)(s >> 8) / 16777216.f - 0.5f) * 2.f;
        }
        for (int i = 0; i < 256; ++i) perm[b * 256 + i] = (i * 37 + 11 * b) % 256; /* 37 is coprime with 256: a permutation */
        for (int i = 0; i < 256; ++i) scores[(size_t)b * M * N + (size_t)i * N + perm[b * 256 + i]] += 12.f;
    }

    float *d_scores = NULL, *d_logZ = NULL, *d_ms0 = NULL, *d_ms1 = NULL;
    int64_t *d_m0 = NULL, *d_m1 = NULL;
    const size_t nz = (size_t)B * (M + 1) * (N + 1);
    CHECK(e2emv_malloc(ctx, (void**)&d_scores, sizeof(float) * B * M * N));
    CHECK(e2emv_malloc(ctx, (void**)&d_logZ, sizeof(float) * nz));
    CHECK(e2emv_malloc(ctx, (void**)&d_m0, sizeof(int64_t) * B * M));
    CHECK(e2emv_malloc(ctx, (void**)&d_m1, sizeof(int64_t) * B * N));
    CHECK(e2emv_malloc(ctx, (void**)&d_ms0, sizeof(float) * B * M));
    CHECK(e2emv_malloc(ctx, (void**)&d_ms1, sizeof(float) * B * N));
    CHECK(e2emv_h2d(ctx, d_scores, scores, sizeof(float) * B * M * N, NULL));
    CHECK(e2emv_sinkhorn(ctx, B, M, N, d_scores, 1.0f, iters, d_logZ, NULL));
    CHECK(e2emv_extract_matches(ctx, B, M, N, d_logZ, 0.2f, d_m0, d_m1, d_ms0, d_ms1, NULL));

    float* logZ = (float*)malloc(sizeof(float) * nz);
    int64_t* m0 = (int64_t*)malloc(sizeof(int64_t) * B * M);
    CHECK(e2emv_d2h(ctx, logZ, d_logZ, sizeof(float) * nz, NULL));
    CHECK(e2emv_d2h(ctx, m0, d_m0, sizeof(int64_t) * B * M, NULL));
    CHECK(e2emv_sync(ctx, NULL));

    /* marginals: upstream's log_optimal_transport returns Z - norm, i.e. probabilities multiplied by M + N, so that every
     * ordinary row / column sums to 1; the column update is the last half-iteration, so the column sums are exact to
     * rounding and the row sums to the convergence of 100 iterations */
    const double mu = 1.0;
    double worst_col = 0.0, worst_row = 0.0;
    for (int b = 0; b < B; ++b) {
        const float* z = logZ + (size_t)b * (M + 1) * (N + 1);
        for (int j = 0; j < N; ++j) {
            double c = 0.0;
            for (int i = 0; i <= M; ++i) c += exp((double)z[(size_t)i * (N + 1) + j]);
            if (fabs(c / mu - 1.0) > worst_col) worst_col = fabs(c / mu - 1.0);
        }
        for (int i = 0; i < M; ++i) {
            double r = 0.0;
            for (int j = 0; j <= N; ++j) r += exp((double)z[(size_t)i * (N + 1) + j]);
            if (fabs(r / mu - 1.0) > worst_row) worst_row = fabs(r / mu - 1.0);
        }
    }
    int recovered = 0;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < 256; ++i) recovered += m0[b * M + i] == perm[b * 256 + i];
    printf("column marginal error %.2e, row marginal error %.2e, planted matches recovered %d / %d\n", worst_col, worst_row,
           recovered, B * 256);

    e2emv_free(ctx, d_scores); e2emv_free(ctx, d_logZ); e2emv_free(ctx, d_m0); e2emv_free(ctx, d_m1);
    e2emv_free(ctx, d_ms0); e2emv_free(ctx, d_ms1);
    e2emv_destroy(ctx);
    free(scores); free(perm); free(logZ); free(m0);
    return (worst_col < 1e-4 && worst_row < 1e-2 && recovered == B * 256) ? 0 : 1;
}
