/* abi_demo.c -- libhnswgpu.so used from plain C, exactly as a JNI / Panama binding would: no Python, no torch.
 * Build:  gcc -O2 -Iinclude examples/abi_demo.c -Lhnsw-clj_amd -lhnswgpu -Wl,-rpath,$PWD/hnsw-clj_amd -lm -o /tmp/abi_demo
 * Mirrors the reference's smoke scenario (test/hnsw/core_test.clj:33-47): build an index over n random vectors,
 * search with the first vector, expect itself at distance ~0; then the same through IVF-FLAT and a save / load
 * round trip (test/hnsw/integration_test.clj:68-78 intent); then the INTEGRATION.md section 5 migration: a graph held
 * the way the reference holds it -- per node, per level, an unordered SET of neighbour ids (UltraNode.neighbors is an
 * Object[] of HashSet<String>, src/hnsw/ultra_fast.clj:99-102) -- is flattened into the levels / l0_adj / up_off / up_adj
 * arrays and served through hnswgpu_set_graph on a fresh handle.  Usage: abi_demo [index-file [migrated-index-file
 * results-file]]: with the two extra paths the migrated index and its answers are written out, so that
 * tests/test_gpu_parity.py::test_c_abi_from_plain_c can run the CPU oracle on that very graph.
 * Exit code 0 = all checks passed. */
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
    /* ---- migrate a graph built elsewhere -------------------------------------------------
     * "Elsewhere" is played by the graph just loaded: it is first turned into the reference's in-memory shape -- for
     * every node a `neighbors` array with one id SET per level (here: a malloc'ed id array in REVERSED order, a set
     * has no order to rely on) -- and only that shape is read from here on, as a JVM exporter would. */
    int32_t M = 0, M0 = 0, entry = -1, max_level = 0;
    int64_t up_blocks = 0;
    CHECK(hnswgpu_graph_sizes(idx2, &M, &M0, &up_blocks, &entry, &max_level));
    int32_t *levels = (int32_t *)malloc(sizeof(int32_t) * n), *l0 = (int32_t *)malloc(sizeof(int32_t) * n * M0);
    int64_t *upoff = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    int32_t *up = (int32_t *)malloc(sizeof(int32_t) * (up_blocks > 0 ? up_blocks : 1) * M);
    CHECK(hnswgpu_get_graph(idx2, levels, l0, upoff, up));
    typedef struct {
        int level;
        int32_t **set;  /* [level + 1] neighbour ids of each level */
        int *count;
    } Node;
    Node *nodes = (Node *)malloc(sizeof(Node) * n);
    for (int64_t i = 0; i < n; i++) {
        nodes[i].level = levels[i];
        nodes[i].set = (int32_t **)malloc(sizeof(int32_t *) * (levels[i] + 1));
        nodes[i].count = (int *)malloc(sizeof(int) * (levels[i] + 1));
        for (int lv = 0; lv <= levels[i]; lv++) {
            const int width = lv == 0 ? M0 : M;
            const int32_t *src = lv == 0 ? l0 + i * M0 : up + (upoff[i] + lv - 1) * M;
            int c = 0;
            while (c < width && src[c] >= 0) c++;
            nodes[i].set[lv] = (int32_t *)malloc(sizeof(int32_t) * (c > 0 ? c : 1));
            for (int j = 0; j < c; j++) nodes[i].set[lv][j] = src[c - 1 - j];
            nodes[i].count[lv] = c;
        }
    }
    /* the flattening a maintainer writes (layout: include/hnswgpu.h): ids -> rows in insertion order, every set padded
     * with -1 to M0 (level 0) / M (upper levels), up_off = prefix sum of the node levels */
    int32_t *f_levels = (int32_t *)malloc(sizeof(int32_t) * n), *f_l0 = (int32_t *)malloc(sizeof(int32_t) * n * M0);
    int64_t *f_upoff = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    f_upoff[0] = 0;
    for (int64_t i = 0; i < n; i++) {
        f_levels[i] = nodes[i].level;
        f_upoff[i + 1] = f_upoff[i] + nodes[i].level;
    }
    int32_t *f_up = (int32_t *)malloc(sizeof(int32_t) * (f_upoff[n] > 0 ? f_upoff[n] : 1) * M);
    for (int64_t i = 0; i < n; i++)
        for (int lv = 0; lv <= nodes[i].level; lv++) {
            const int width = lv == 0 ? M0 : M;
            int32_t *dst = lv == 0 ? f_l0 + i * M0 : f_up + (f_upoff[i] + lv - 1) * M;
            for (int j = 0; j < width; j++) dst[j] = j < nodes[i].count[lv] ? nodes[i].set[lv][j] : -1;
        }
    hnswgpu_index *mig = NULL;
    CHECK(hnswgpu_create(base, n, dim, HNSWGPU_COSINE, 0, &mig));
    CHECK(hnswgpu_set_graph(mig, f_levels, f_l0, M0, f_upoff, f_up, M, entry, max_level));
    int32_t mids[8 * 5];
    float mdist[8 * 5];
    int64_t mstats[8 * 2];
    CHECK(hnswgpu_hnsw_search(mig, base + 100 * dim, 8, k, 64, mids, mdist, mstats)); /* rows 100..107 as queries */
    for (int q = 0; q < 8; q++)
        if (mids[q * k] != 100 + q || mstats[2 * q] < 64) {
            fprintf(stderr, "migrated graph: query %d did not find itself\n", q);
            return 1;
        }
    /* a set has no order: the neighbours were handed over reversed, so the traversal visits them in another order than
     * on the source handle -- same graph, same nearest row; result lists may differ where the search is approximate */
    f_l0[5 * M0] = (int32_t)n + 7; /* and a damaged export is refused, not traversed */
    if (hnswgpu_set_graph(mig, f_levels, f_l0, M0, f_upoff, f_up, M, entry, max_level) != HNSWGPU_EINVAL) return 1;
    if (argc > 3) {
        CHECK(hnswgpu_save(mig, argv[2]));
        FILE *rf = fopen(argv[3], "wb");
        if (!rf || fwrite(mids, sizeof(mids), 1, rf) != 1 || fwrite(mdist, sizeof(mdist), 1, rf) != 1 ||
            fwrite(mstats, sizeof(mstats), 1, rf) != 1 || fclose(rf) != 0) {
            fprintf(stderr, "cannot write %s\n", argv[3]);
            return 1;
        }
    }
    CHECK(hnswgpu_destroy(mig));
    /* error behaviour: codes + message, no crash */
    if (hnswgpu_hnsw_search(idx2, base, 1, 0, 0, ids2, dist2, NULL) != HNSWGPU_EINVAL) return 1;
    CHECK(hnswgpu_destroy(idx2));
    free(base);
    printf("abi_demo ok: HNSW + IVF-FLAT + save/load + set_graph migration through the C ABI (n=%lld, dim=%d)\n",
           (long long)n, dim);
    return 0;
}
