/* abi_demo.c -- libhnswgpu.so used from plain C, exactly as a JNI / Panama binding would: no Python, no torch.
 * Build:  gcc -O2 -Iinclude examples/abi_demo.c -Lhnsw-clj_amd -lhnswgpu -Wl,-rpath,$PWD/hnsw-clj_amd -lm -o /tmp/abi_demo
 * Mirrors the reference's smoke scenario (test/hnsw/core_test.clj:33-47): build an index over n random vectors,
 * search with the first vector, expect itself at distance ~0; then the same through IVF-FLAT and a save / load
 * round trip (test/hnsw/integration_test.clj:68-78 intent).  Exit code 0 = all checks passed. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hnswgpu.h"

#define CHECK(call)                                                            \
    do {                                                                       \
        int rc_ = (call);                                                      \
        if (rc_ != 0) {                                                        \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, hnswgpu_last_error()); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(int argc, char **argv) {
    const int64_t n = 2000;
    const int dim = 96, k = 5;
    const char *path = argc > 1 ? argv[1] : "/tmp/abi_demo_index.bin";
    float *base = (float *)malloc(sizeof(float) * n * dim);
    unsigned s = 12345;
    for (int64_t i = 0; i < n * dim; i++) {
        s = s * 1664525u + 1013904223u;
        base[i] = (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f;
    }
    int32_t ndev = 0;
    CHECK(hnswgpu_device_count(&ndev));
    if (ndev < 1) {
        fprintf(stderr, "no GPU\n");
        return 2;
    }
    hnswgpu_index *idx = NULL;
    CHECK(hnswgpu_create(base, n, dim, HNSWGPU_COSINE, 0, &idx));
    CHECK(hnswgpu_hnsw_build(idx, 16, 200, 42));
    int32_t ids[8 * 5];
    float dist[8 * 5];
    CHECK(hnswgpu_hnsw_search(idx, base, 8, k, 0, ids, dist, NULL)); /* the first 8 rows as queries, ef = max(k, 50) */
    for (int q = 0; q < 8; q++) {
        if (ids[q * k] != q || fabsf(dist[q * k]) > 1e-3f) {
            fprintf(stderr, "HNSW: query %d did not find itself (id %d, d %g)\n", q, ids[q * k], dist[q * k]);
            return 1;
        }
        for (int i = 1; i < k; i++)
            if (dist[q * k + i] < dist[q * k + i - 1]) {
                fprintf(stderr, "HNSW: results not ascending\n");
                return 1;
            }
    }
    CHECK(hnswgpu_ivf_build(idx, 24, 10, 42)); /* the reference's defaults: 24 partitions, 10 Lloyd iterations */
    int32_t iids[8 * 5], iids2[8 * 5];
    float idist[8 * 5], idist2[8 * 5];
    CHECK(hnswgpu_ivf_search(idx, base, 8, k, 12, iids, idist, NULL)); /* :precise = 12 probes */
    for (int q = 0; q < 8; q++)
        if (iids[q * k] != q) {
            fprintf(stderr, "IVF: query %d did not find itself\n", q);
            return 1;
        }
    CHECK(hnswgpu_save(idx, path));
    CHECK(hnswgpu_destroy(idx));
    hnswgpu_index *idx2 = NULL;
    CHECK(hnswgpu_load(path, 0, &idx2));
    int32_t ids2[8 * 5];
    float dist2[8 * 5];
    CHECK(hnswgpu_hnsw_search(idx2, base, 8, k, 0, ids2, dist2, NULL));
    CHECK(hnswgpu_ivf_search(idx2, base, 8, k, 12, iids2, idist2, NULL));
    if (memcmp(ids, ids2, sizeof(ids)) || memcmp(dist, dist2, sizeof(dist)) || memcmp(iids, iids2, sizeof(iids)) ||
        memcmp(idist, idist2, sizeof(idist))) {
        fprintf(stderr, "save/load round trip changed the results\n");
        return 1;
    }
    /* error behaviour: codes + message, no crash */
    if (hnswgpu_hnsw_search(idx2, base, 1, 0, 0, ids2, dist2, NULL) != HNSWGPU_EINVAL) return 1;
    CHECK(hnswgpu_destroy(idx2));
    free(base);
    printf("abi_demo ok: HNSW + IVF-FLAT + save/load through the C ABI (n=%lld, dim=%d)\n", (long long)n, dim);
    return 0;
}
