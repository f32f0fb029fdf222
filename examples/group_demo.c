/* group_demo.c -- ONE index over several GPUs through the C ABI, from plain C (what a JNI / Panama binding does).
 *
 *   gcc -O2 -Iinclude examples/group_demo.c -Lhnsw-clj_amd -lhnswgpu -Wl,-rpath,$PWD/hnsw-clj_amd -lm -o group_demo
 *   ./group_demo [handles]        (handles on GPU 0, 1, ... modulo the GPUs present; default 4)
 *
 * An IVF-FLAT index is built on ONE handle (hnswgpu_ivf_build), exported (hnswgpu_get_ivf) and handed to a group
 * (hnswgpu_group_set_ivf): whole lists dealt to the group's handles, every handle searches what it holds, the partial
 * top-k lists are merged on the first device.  The group's answer must equal the single handle's: ids and distance bits
 * (search-partitioned's scatter / top-k / gather / sort, partitioned_hnsw.clj:149-196, on ivf_flat.clj:261-294's scan). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hnswgpu.h"

#define CHECK(call)                                                          \
    do {                                                                     \
        int rc_ = (call);                                                    \
        if (rc_ != 0) {                                                      \
            fprintf(stderr, "%s failed: %d %s\n", #call, rc_, hnswgpu_last_error()); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

static uint64_t rng_state = 88172645463325252ull;
static double rnd(void) { /* xorshift64*, [0, 1) */
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return (double)((rng_state * 2685821657736338717ull) >> 11) / 9007199254740992.0;
}
static float gauss(void) { return (float)(sqrt(-2.0 * log(rnd() + 1e-300)) * cos(6.283185307179586 * rnd())); }

int main(int argc, char **argv) {
    const int handles = argc > 1 ? atoi(argv[1]) : 4;
    const int64_t n = 20000;
    const int32_t dim = 64, nlist = 48, nprobe = 6, k = 10, nq = 50, ncl = 25;
    int32_t ngpu = 0;
    CHECK(hnswgpu_device_count(&ngpu));
    if (ngpu < 1 || handles < 1 || handles > 64) {
        fprintf(stderr, "need a GPU and 1..64 handles\n");
        return 1;
    }
    float *centres = malloc(sizeof(float) * ncl * dim), *base = malloc(sizeof(float) * n * dim), *Q = malloc(sizeof(float) * nq * dim);
    for (int i = 0; i < ncl * dim; i++) centres[i] = gauss();
    for (int64_t i = 0; i < n; i++) {
        const int c = (int)(rnd() * ncl);
        for (int j = 0; j < dim; j++) base[i * dim + j] = centres[c * dim + j] + 0.4f * gauss();
    }
    for (int i = 0; i < nq; i++) {
        const int c = (int)(rnd() * ncl);
        for (int j = 0; j < dim; j++) Q[i * dim + j] = centres[c * dim + j] + 0.4f * gauss();
    }
    /* the index on one handle */
    hnswgpu_index *one = NULL;
    CHECK(hnswgpu_create(base, n, dim, 0 /* cosine */, 0, &one));
    CHECK(hnswgpu_ivf_build(one, nlist, 5, 42));
    float *cen = malloc(sizeof(float) * nlist * dim);
    int64_t *off = malloc(sizeof(int64_t) * (nlist + 1));
    int32_t *lids = malloc(sizeof(int32_t) * n);
    CHECK(hnswgpu_get_ivf(one, cen, off, lids));
    int32_t *ids1 = malloc(sizeof(int32_t) * nq * k), *idsg = malloc(sizeof(int32_t) * nq * k);
    float *d1 = malloc(sizeof(float) * nq * k), *dg = malloc(sizeof(float) * nq * k);
    CHECK(hnswgpu_ivf_search(one, Q, nq, k, nprobe, ids1, d1, NULL));
    /* the same index served by a group */
    int32_t devs[64];
    for (int i = 0; i < handles; i++) devs[i] = i % ngpu;
    hnswgpu_group *g = NULL;
    CHECK(hnswgpu_group_create(devs, handles, dim, 0, &g));
    CHECK(hnswgpu_group_set_ivf(g, base, n, cen, nlist, off, lids));
    int64_t rows[64];
    int32_t nd = 0, kind = 0;
    int64_t gn = 0;
    CHECK(hnswgpu_group_info(g, &nd, &gn, &kind, rows));
    printf("group: %d handles on %d GPU(s), %lld rows, rows per handle:", nd, ngpu, (long long)gn);
    for (int i = 0; i < nd; i++) printf(" %lld", (long long)rows[i]);
    printf("\n");
    CHECK(hnswgpu_group_ivf_search(g, Q, nq, k, nprobe, idsg, dg));
    if (memcmp(ids1, idsg, sizeof(int32_t) * nq * k) != 0 || memcmp(d1, dg, sizeof(float) * nq * k) != 0) {
        fprintf(stderr, "the group's answer differs from the single handle's\n");
        return 1;
    }
    /* HNSW: one sub-graph per handle */
    CHECK(hnswgpu_group_hnsw_build(g, base, n, 16, 100, 42));
    CHECK(hnswgpu_group_hnsw_search(g, Q, nq, k, 64, idsg, dg));
    for (int i = 0; i < nq; i++)
        for (int j = 0; j + 1 < k; j++)
            if (idsg[i * k + j] < 0 || idsg[i * k + j] >= n || !(dg[i * k + j] <= dg[i * k + j + 1])) {
                fprintf(stderr, "group hnsw: bad result row %d\n", i);
                return 1;
            }
    if (hnswgpu_group_ivf_search(g, Q, nq, k, nprobe, idsg, dg) == 0) { /* the group now holds sub-graphs, not lists */
        fprintf(stderr, "an IVF search on an HNSW group should fail\n");
        return 1;
    }
    CHECK(hnswgpu_group_destroy(g));
    CHECK(hnswgpu_destroy(one));
    printf("group_demo ok\n");
    return 0;
}
