/* parallel_callers.c -- the reference's throughput protocol from plain C: T threads, each issuing single-query
 * search-knn calls against one index (helper/parallel_search.clj:15-49; the published 4,719 - 5,376 QPS are 20 JVM
 * threads on 31,173 x 768, BENCHMARK_RESULTS_ACTUAL.md / BENCHMARK_SUMMARY.md).  libhnswgpu combines the concurrent
 * callers into one launch, so the pattern a drop-in user already has keeps scaling with the thread count.
 * Build:  gcc -O2 -pthread -Iinclude examples/parallel_callers.c -Lhnsw-clj_amd -lhnswgpu -Wl,-rpath,$PWD/hnsw-clj_amd -lm -o /tmp/parallel_callers
 * Usage:  /tmp/parallel_callers [n] [dim] [ef]      (synthetic rows on a 32-dimensional manifold, cosine, k = 10) */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "hnswgpu.h"

static hnswgpu_index *g_idx;
static const float *g_q;
static int g_dim, g_ef, g_per;
static int32_t *g_ids;

static unsigned g_seed = 12345;
static float frand(void) {
    g_seed = g_seed * 1664525u + 1013904223u;
    return (float)((g_seed >> 8) & 0xffff) / 65536.0f - 0.5f;
}

static void *worker(void *arg) {
    const long t = (long)arg;
    float d[10];
    for (long i = t * g_per; i < (t + 1) * g_per; i++)
        if (hnswgpu_hnsw_search(g_idx, g_q + i * g_dim, 1, 10, g_ef, g_ids + i * 10, d, NULL) != 0) {
            fprintf(stderr, "search failed: %s\n", hnswgpu_last_error());
            exit(1);
        }
    return NULL;
}

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 31173;
    const int dim = argc > 2 ? atoi(argv[2]) : 768, r = 32, nq = 8192;
    g_dim = dim;
    g_ef = argc > 3 ? atoi(argv[3]) : 100;
    float *w = (float *)malloc(sizeof(float) * r * dim), *z = (float *)malloc(sizeof(float) * r);
    float *base = (float *)malloc(sizeof(float) * (n + nq) * dim);
    for (int i = 0; i < r * dim; i++) w[i] = frand();
    for (int64_t i = 0; i < n + nq; i++) {  /* x = normalise(W z + 0.05 e): rows and queries from one manifold */
        for (int a = 0; a < r; a++) z[a] = frand();
        double nn = 0;
        for (int j = 0; j < dim; j++) {
            float v = 0.05f * frand();
            for (int a = 0; a < r; a++) v += z[a] * w[a * dim + j];
            base[i * dim + j] = v;
            nn += (double)v * v;
        }
        const float inv = (float)(1.0 / sqrt(nn));
        for (int j = 0; j < dim; j++) base[i * dim + j] *= inv;
    }
    g_q = base + n * dim;
    g_ids = (int32_t *)malloc(sizeof(int32_t) * nq * 10);
    int32_t *want = (int32_t *)malloc(sizeof(int32_t) * nq * 10);
    float *wd = (float *)malloc(sizeof(float) * nq * 10);
    if (hnswgpu_create(base, n, dim, HNSWGPU_COSINE, 0, &g_idx) || hnswgpu_hnsw_build(g_idx, 16, 200, 42) ||
        hnswgpu_hnsw_search(g_idx, g_q, nq, 10, g_ef, want, wd, NULL)) {
        fprintf(stderr, "setup failed: %s\n", hnswgpu_last_error());
        return 1;
    }
    const int threads[] = {1, 5, 10, 20, 50, 100, 200};
    for (unsigned c = 0; c < sizeof(threads) / sizeof(threads[0]); c++) {
        const int T = threads[c];
        g_per = T == 1 ? 1000 : nq / T;
        pthread_t th[256];
        memset(g_ids, 0xff, sizeof(int32_t) * nq * 10);
        const double t0 = now();
        for (long t = 0; t < T; t++) pthread_create(&th[t], NULL, worker, (void *)t);
        for (long t = 0; t < T; t++) pthread_join(th[t], NULL);
        const double dt = now() - t0;
        const int same = memcmp(g_ids, want, sizeof(int32_t) * 10 * (size_t)T * g_per) == 0;
        printf("%3d threads x %4d single-query calls: %.3f s = %8.0f QPS  (%s)\n", T, g_per, dt, T * (double)g_per / dt,
               same ? "ids identical to one batch" : "IDS DIFFER");
        if (!same) return 1;
    }
    hnswgpu_destroy(g_idx);
    printf("parallel_callers ok\n");
    return 0;
}
