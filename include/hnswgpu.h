/*
 * hnswgpu.h -- C ABI of libhnswgpu.so: the MI355X (gfx950) distance engine behind hnsw-clj's
 * index/search API.  This header is the drop-in boundary: every entry point names the reference
 * interface (file:line under damesek/hnsw-clj) it replaces.  INTEGRATION.md shows the Clojure
 * (Panama FFM / JNI) binding a maintainer adds on the reference side.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only.  Every function returns 0 on success or a negative
 *    HNSWGPU_E* code; hnswgpu_last_error() returns a thread-local message.  No exception crosses
 *    the boundary.
 *  - Host-pointer entry points: the caller owns every host buffer; the library copies or consumes
 *    it before returning.  `_dev` entry points take DEVICE pointers (e.g. a torch tensor's
 *    data_ptr) plus a hipStream_t passed as void* (NULL = HIP's default stream); they only
 *    enqueue work and do not synchronise.  Calls on different streams are ordered against each
 *    other with events (they share the index's scratch buffers).
 *  - Row ids are dense int32 in [0, n); the String-id <-> row table stays on the Clojure side
 *    (UltraNode.id is a String, src/hnsw/ultra_fast.clj:99).
 *  - Results are ascending by distance; fewer than k results are padded with id -1 / +inf
 *    (the reference returns shorter seqs: test/hnsw/core_test.clj:90-96, ultra_fast.clj:349-351).
 *  - Metrics: COSINE = 1 - dot/(|a||b|), 1.0 when a norm is 0 (ultra_fast.clj:53-95);
 *    L2 = sqrt(sum (a-b)^2), rooted (ultra_fast.clj:43-51); DOT = -dot (ordering key for
 *    simd-optimized/dot-product, simd_optimized.clj:283-293).
 *  - Arithmetic is float32 on the device (the reference is float64): ids identical to the f64
 *    reference order, distances within 1e-4 relative (BASELINE.json north_star).
 *  - An index handle is safe for concurrent *_search calls from several host threads (calls are
 *    serialised on the handle's stream); set_* / build calls must not overlap searches.
 */
#ifndef HNSWGPU_H
#define HNSWGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HNSWGPU_VERSION 104

#define HNSWGPU_COSINE 0
#define HNSWGPU_L2 1
#define HNSWGPU_DOT 2

#define HNSWGPU_OK 0
#define HNSWGPU_EINVAL (-1)   /* bad argument (null pointer, k < 1, dim unsupported ...)       */
#define HNSWGPU_EHIP (-2)     /* HIP runtime error; message has the hipError string            */
#define HNSWGPU_ESTATE (-3)   /* index lacks what the call needs (no graph / no IVF lists)      */
#define HNSWGPU_ENOMEM (-4)   /* host or device allocation failed                               */
#define HNSWGPU_ELIMIT (-5)   /* a documented limit exceeded (dim > 3072, ef > 4096, k > 1024)  */

typedef struct hnswgpu_index hnswgpu_index;

int hnswgpu_version(void);
const char *hnswgpu_last_error(void);
int hnswgpu_device_count(int32_t *count);

/* ---- index lifetime ----------------------------------------------------------------------------
 * Replaces the data-map of the reference indexes: ConcurrentHashMap<String,UltraNode{double[]}>
 * (ultra_fast.clj:99-111) / IVFFlatIndex.data-map (ivf_flat.clj:22-27).  `base` is n x dim float32
 * row-major host memory; it is copied to one contiguous HBM matrix (rows padded to 16 B) and the
 * per-row norms are precomputed on the device (ivf_flat.clj:171-177, simd_optimized.clj:206-216).
 * n == 0 is legal (empty index: every search returns no results, ultra_fast.clj:349-351). */
int hnswgpu_create(const float *base, int64_t n, int32_t dim, int32_t metric, int32_t device,
                   hnswgpu_index **out);
/* Same, from a device-resident n x dim matrix with row stride ld floats (copied). */
int hnswgpu_create_dev(const float *d_base, int64_t n, int32_t dim, int64_t ld, int32_t metric, int32_t device,
                       void *stream, hnswgpu_index **out);
int hnswgpu_destroy(hnswgpu_index *idx);
int hnswgpu_info(const hnswgpu_index *idx, int64_t *n, int32_t *dim, int32_t *metric, int32_t *has_graph,
                 int32_t *nlist);
int hnswgpu_sync(hnswgpu_index *idx);

/* ---- distance seams ------------------------------------------------------------------------------
 * hnswgpu_pair_distance: the :distance-fn signature (fn ^double [^doubles a ^doubles b])
 *   (ultra_fast.clj:43-95, simd_optimized.clj:145-160,283-293), one pair, computed on the device.
 * hnswgpu_batch_distances: simd-optimized/batch-cosine-distances, batch-euclidean-distances
 *   [query vectors] -> distances (simd_optimized.clj:164-184): q vs base rows ids[0..m) (ids NULL =
 *   rows 0..m-1).  This is the gather-dot used per hop inside HNSW (ultra_fast.clj:185-204).
 * hnswgpu_norms: simd-optimized/precompute-norms (simd_optimized.clj:206-216).
 * hnswgpu_exact_knn: bench/compute-exact-knn (src/hnsw/bench.clj:72-84) and
 *   simd-optimized/top-k-distances (simd_optimized.clj:271-280): brute force over the full base. */
int hnswgpu_pair_distance(int32_t metric, const float *a, const float *b, int32_t dim, int32_t device, float *out);
int hnswgpu_batch_distances(hnswgpu_index *idx, const float *q, const int32_t *ids, int32_t m, float *out);
int hnswgpu_norms(hnswgpu_index *idx, float *out_norms);
int hnswgpu_exact_knn(hnswgpu_index *idx, const float *Q, int32_t nq, int32_t k, int32_t *out_ids,
                      float *out_dist);
int hnswgpu_exact_knn_dev(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t k, int32_t *d_out_ids,
                          float *d_out_dist, void *stream);

/* ---- HNSW ---------------------------------------------------------------------------------------
 * Graph layout (what UltraGraph's nodes / neighbor HashSets flatten to, ultra_fast.clj:99-111):
 *   levels[n]                node level
 *   l0_adj[n * M0]           layer-0 neighbours, -1 padded, adjacency-array order
 *   up_off[n + 1]            prefix sum of levels: node i's layers 1..levels[i] are the blocks
 *                            up_off[i] .. up_off[i+1]-1 of up_adj
 *   up_adj[up_off[n] * M]    one block of M neighbours (-1 padded) per (node, layer >= 1)
 *   entry, max_level         entry point and its level
 * hnswgpu_set_graph uploads a graph built elsewhere (e.g. by the reference's own insert-single);
 * hnswgpu_hnsw_build builds one on the device (batched insertion; replaces build-index /
 * insert-batch ultra_fast.clj:303-344); hnswgpu_get_graph exports it in the same layout.
 * hnswgpu_hnsw_search replaces search-knn (ultra_fast.clj:346-374) for a batch of queries -- the
 * seam of BatchSearchIndex/search-batch* (api/protocol.clj:58-67) and parallel-search-futures
 * (helper/parallel_search.clj:15-49).  ef is explicit; the reference's value is max(k, 50)
 * (ultra_fast.clj:355): pass ef <= 0 to get it.  stats (optional, nq x 2 int64): distance
 * evaluations and expansions per query.
 * Ties: the reference still expands a candidate whose distance EQUALS the current ef-th distance
 * (ultra_fast.clj:175-178, `<=`).  The traversal kernel keeps up to 32 such evicted-but-tied candidates per
 * query; a query that had more (hundreds of duplicated rows) is repeated on the device, in the same call, with the
 * largest candidate list the LDS holds, so results and counters are the reference's on such data too. */
int hnswgpu_set_graph(hnswgpu_index *idx, const int32_t *levels, const int32_t *l0_adj, int32_t M0,
                      const int64_t *up_off, const int32_t *up_adj, int32_t M, int32_t entry, int32_t max_level);
int hnswgpu_hnsw_build(hnswgpu_index *idx, int32_t M, int32_t ef_construction, int64_t seed);
/* The same with build options.  flags = 0 is hnswgpu_hnsw_build: batched insertion, every node linked to its m closest
 * candidates (insert-single / prune-connections-ultra, ultra_fast.clj:216-299).
 *   HNSWGPU_BUILD_SEQUENTIAL  insert-single itself: one row at a time, the walk starting at min(level, entry-level) with the
 *                             entry point (ultra_fast.clj:247-248), an over-full neighbour pruned at once (:264-266).  The
 *                             graph equals the CPU restatement's (oracle/oracle.c: orc_hnsw_build_ex) edge for edge; one
 *                             launch round trip per row -- the parity mode, not the fast one.
 *   HNSWGPU_BUILD_HEURISTIC   links chosen by get-neighbors-heuristic (src/hnsw/graph.clj:162-198, the builder behind
 *                             README's hnsw.hnsw-search): candidates ascending by (distance, id), one is taken unless it
 *                             is closer to an already taken one than to the node itself -- for the new node's own links
 *                             (from all its ef_construction candidates) and for an over-full neighbour list
 *                             (prune-connections, graph.clj:208-232).  Clusters stay connected to each other: on the
 *                             survey's clustered-normalised 31k x 768 set recall@10 goes from 0.02 (closest-m, at any ef)
 *                             to 0.98.  The pair distances run on the device (heuristic_select_kernel).
 *   HNSWGPU_BUILD_SYMMETRIC   with HEURISTIC: an edge a pruning drops is removed from the other node's list as well
 *                             (graph.clj:226-231)
 *   HNSWGPU_BUILD_EXTEND      with HEURISTIC: extend-candidates? for the new node's own selection (graph.clj:191-195)
 * hnswgpu_hnsw_add inserts with the options the handle's graph was built with (batched). */
#define HNSWGPU_BUILD_SEQUENTIAL 1
#define HNSWGPU_BUILD_HEURISTIC 2
#define HNSWGPU_BUILD_SYMMETRIC 4
#define HNSWGPU_BUILD_EXTEND 8
int hnswgpu_hnsw_build_ex(hnswgpu_index *idx, int32_t M, int32_t ef_construction, int64_t seed, int32_t flags);
/* insert-single on a LIVE index (src/hnsw/ultra_fast.clj:216-275, reached by add-vector! src/hnsw/api.clj:30-33 and add!
 * src/hnsw/api/simple.clj:31-42): `m` more rows (m x dim floats) join the base matrix and the installed graph.  Their row
 * ids are n, n + 1, ...; their levels continue the seeded java.util.Random sequence (row i takes its i-th draw: the same
 * `seed` as the build gives the levels a from-scratch build of all rows would draw); they are inserted in batches, by
 * the search kernel and the linker of hnswgpu_hnsw_build, against the CURRENT graph (any graph with M0 = 2 M:
 * hnswgpu_hnsw_build's, or one installed by hnswgpu_set_graph / hnswgpu_load -- its edge distances, which the reference's
 * pruning sorts by (:279-299), are recomputed on the device).  Not concurrent with searches on the same handle (they
 * wait); IVF lists must be built / installed AFTER the rows they cover have been added. */
int hnswgpu_hnsw_add(hnswgpu_index *idx, const float *rows, int64_t m, int32_t ef_construction, int64_t seed);
int hnswgpu_graph_sizes(const hnswgpu_index *idx, int32_t *M, int32_t *M0, int64_t *up_blocks, int32_t *entry,
                        int32_t *max_level);
int hnswgpu_get_graph(const hnswgpu_index *idx, int32_t *levels, int32_t *l0_adj, int64_t *up_off,
                      int32_t *up_adj);
int hnswgpu_hnsw_search(hnswgpu_index *idx, const float *Q, int32_t nq, int32_t k, int32_t ef, int32_t *out_ids,
                        float *out_dist, int64_t *stats);
int hnswgpu_hnsw_search_dev(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t k, int32_t ef,
                            int32_t *d_out_ids, float *d_out_dist, int64_t *d_stats, void *stream);

/* ---- IVF-FLAT -------------------------------------------------------------------------------------
 * hnswgpu_ivf_build: build-ivf-flat-index :partition-method :kmeans (ivf_flat.clj:137-211):
 *   k-means++ seeding with java.util.Random(seed) (:32-60), max_iter Lloyd iterations (:92-131),
 *   final assignment, inverted lists in index order.  seed 42 is the reference's.
 * hnswgpu_set_ivf / hnswgpu_get_ivf: import / export centroids (nlist x dim f32), list_off[nlist+1],
 *   list_ids[n] (row ids in list order).
 * hnswgpu_kmeans_assign: assign-to-nearest-centroid for every base row (ivf_flat.clj:79-90):
 *   strict <, lowest index wins ties.
 * hnswgpu_ivf_search: search-ivf-flat (ivf_flat.clj:236-294) with explicit nprobe for a batch of
 *   queries: centroid routing (:261-269), brute-force scan of the probed lists with precomputed
 *   norms (:217-234), merge, take k.  out_probes optional (nq x nprobe list ids).
 *   Two summation orders exist for this call (cosine / dot).  The GEMV order (wave-strided f32 chain + butterfly) serves
 *   EVERY batch size on a handle with the int8 and half-precision copies of its lists (hnswgpu_set_rejection_test 1 / 2
 *   with dim >= 128, the default; k <= 256 -- divided by (mean list length / 1024) on lists longer than 2048 rows on
 *   average: k <= 36 at 7800 rows per list): only the candidates whose bounds can still reach the k nearest are evaluated
 *   at all, and nq queries in one call return bit for bit what nq single calls return.  Without the copies (mode 0,
 *   dim < 128, HNSWGPU_IVF_HALF=0 leaves a boundary of 48) or for k > 256, the GEMV order serves up to 12 (query, list)
 *   pairs per list (nq * nprobe <= 12 * nlist: one GEMV per pair; from 1.5 pairs per list the pairs of a list share one
 *   pass over its rows -- same chain, same bits) and larger batches are grouped by list and scanned by the f32-MFMA tile
 *   kernel (k-ordered f32 chain): both orders are within 1e-6 of the f64 reference, but there a query's distance BITS (and
 *   the order of candidates that tie within that) depend on which side of the boundary its batch falls.  Calls combined from
 *   concurrent host threads never change the kernel a call would get alone.  Euclidean: one arithmetic throughout. */
int hnswgpu_ivf_build(hnswgpu_index *idx, int32_t nlist, int32_t max_iter, int64_t seed);
int hnswgpu_set_ivf(hnswgpu_index *idx, const float *centroids, int32_t nlist, const int64_t *list_off,
                    const int32_t *list_ids);
int hnswgpu_get_ivf(const hnswgpu_index *idx, float *centroids, int64_t *list_off, int32_t *list_ids);
int hnswgpu_kmeans_assign(hnswgpu_index *idx, const float *centroids, int32_t nlist, int32_t *out_assign,
                          float *out_dist);
int hnswgpu_kmeanspp(hnswgpu_index *idx, int32_t nlist, int64_t seed, int32_t *out_rows);
/* compute-centroid (ivf_flat.clj:66-77, lightning.clj:24-35) for caller-given lists: f64 mean in list order,
 * stored f32; an empty list yields the zero vector (lightning.clj:118-120). */
int hnswgpu_list_means(hnswgpu_index *idx, int32_t nlist, const int64_t *list_off, const int32_t *list_ids,
                       float *out_centroids);
int hnswgpu_ivf_search(hnswgpu_index *idx, const float *Q, int32_t nq, int32_t k, int32_t nprobe,
                       int32_t *out_ids, float *out_dist, int32_t *out_probes);
int hnswgpu_ivf_search_dev(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t k, int32_t nprobe,
                           int32_t *d_out_ids, float *d_out_dist, void *stream);
/* Same scan with caller-chosen lists instead of centroid routing: probes[nq][nprobe] list ids
 * (-1 = skip).  This is search-ivf-flat's :use-centroids false path (the :turbo mode's random
 * partitions, ivf_flat.clj:243-251,271-272) and lightning's partition scan (lightning.clj:144-187). */
int hnswgpu_ivf_search_lists(hnswgpu_index *idx, const float *Q, int32_t nq, int32_t k, int32_t nprobe,
                             const int32_t *probes, int32_t *out_ids, float *out_dist);

/* ---- one IVF index over several GPUs --------------------------------------------------------------------
 * The reference shards an index with threads: split, search every part, concatenate, sort, take k
 * (ann/partition/partitioned_hnsw.clj:149-196).  For IVF-FLAT the split that keeps ONE index is by inverted list:
 * the centroid table is replicated (nlist x dim), every WHOLE list lives on exactly one GPU, every GPU routes a
 * query to the same nprobe lists (ivf_flat.clj:261-269) and scans the probed lists it holds (:281-288); the
 * per-GPU top-k lists are all-gathered and merged (:291-294).
 * hnswgpu_set_ivf_shard: like hnswgpu_set_ivf on a handle whose base holds only this shard's rows; lists the
 *   shard does not hold have list_off[l] == list_off[l+1].  global_list_len[l] = rows of list l in the WHOLE index:
 *   a search then numbers its candidates by their position in the candidate stream of the whole index.
 * hnswgpu_ivf_search_shard_dev: hnswgpu_ivf_search_dev plus d_out_order[nq][k] (uint32, 0xffffffff padded): that
 *   position.  On an ordinary index (hnswgpu_set_ivf / hnswgpu_ivf_build) it is the position in the index's own stream.
 * hnswgpu_merge_keyed_dev: [nshard][nq][k] (global id, distance, order) -> [nq][k] ascending by (distance, order):
 *   bit for bit the result (ids, distances, tie order) of searching the unsharded index with the same batch.
 * hnswgpu_list_sums: the f64 column sums of compute-centroid (ivf_flat.clj:66-77) for caller-given lists, without
 *   the division -- one shard's contribution to a Lloyd update over a row-sharded base (sums of all shards are
 *   added, e.g. by an RCCL all-reduce, then divided by the global member count).  out_sums: nlist x dim doubles. */
int hnswgpu_set_ivf_shard(hnswgpu_index *idx, const float *centroids, int32_t nlist, const int64_t *list_off,
                          const int32_t *list_ids, const int64_t *global_list_len);
/* Mode-1 handles measure, at their first IVF search, what the int8 bounds separate on their rows, and keep the survivor
 * stream only where it helps (the verdict chooses the kernel path, hence -- for cosine / dot batches past the tile
 * boundary -- the summation order).  The shards of ONE index must share ONE verdict: hnswgpu_ivf_stream_state measures
 * now (if it has not yet) and reports it (1 = stream off), hnswgpu_ivf_set_stream_state installs a verdict without
 * measuring.  hnswgpu_group_set_ivf and sharded.ShardedIVF take the OR over the shards and push it to every member.
 * A shard also decides everything that depends on the mean list length (largest k the stream serves, sample sizes) from
 * the WHOLE index's list lengths (global_list_len), not from the rows it holds. */
int hnswgpu_ivf_stream_state(hnswgpu_index *idx, int32_t *off);
int hnswgpu_ivf_set_stream_state(hnswgpu_index *idx, int32_t off);
int hnswgpu_ivf_search_shard_dev(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t k, int32_t nprobe,
                                 int32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_order, void *stream);
int hnswgpu_merge_keyed_dev(int32_t device, const int32_t *d_ids, const float *d_dist, const uint32_t *d_order,
                            int32_t nshard, int32_t nq, int32_t k, int32_t *d_out_ids, float *d_out_dist, void *stream);
int hnswgpu_list_sums(hnswgpu_index *idx, int32_t nlist, const int64_t *list_off, const int32_t *list_ids,
                      double *out_sums);

/* ---- multi-GPU merge --------------------------------------------------------------------------------
 * Merge `nshard` per-shard top-k lists (what RCCL all-gather delivers) into the global top-k:
 * the partitioned_hnsw.clj:171-196 gather/sort/take-k step.  d_ids/d_dist: [nshard][nq][k] device
 * arrays of GLOBAL ids (-1 padded) and distances; ties keep the lower shard first. */
int hnswgpu_merge_topk_dev(int32_t device, const int32_t *d_ids, const float *d_dist, int32_t nshard, int32_t nq,
                           int32_t k, int32_t *d_out_ids, float *d_out_dist, void *stream);

/* Same merge with different input and output widths: [nshard][nq][k_in] lists -> [nq][k_out].  This is
 * partitioned-hnsw's "k-per-partition results from every partition, sort, take k"
 * (partitioned_hnsw.clj:149-196, :201-231) and ivf-hnsw's "2k from every probed partition graph, sort,
 * take k" (hybrid/ivf_hnsw.clj:312-325). */
int hnswgpu_merge_lists_dev(int32_t device, const int32_t *d_ids, const float *d_dist, int32_t nshard, int32_t nq,
                            int32_t k_in, int32_t k_out, int32_t *d_out_ids, float *d_out_dist, void *stream);

/* ---- re-rank and dense distances --------------------------------------------------------------------
 * hnswgpu_rerank: per query, exact distances to its own candidate rows cand[q][0..m) (-1 or out-of-range
 * = skipped), STABLE ascending sort (ties keep the candidate order, like Collections/sort), first k.
 * Replaces P-HNSW's refine phase (ann/dimreduct/pcaf.clj:239-253), search-knn's final re-rank of the
 * layer-0 result (ultra_fast.clj:362-370) and top-k-distances (simd_optimized.clj:271-280).
 * Arithmetic: the gather order (wave-strided f32 fma chain + butterfly), as hnswgpu_batch_distances. */
int hnswgpu_rerank(hnswgpu_index *idx, const float *Q, int32_t nq, const int32_t *cand, int32_t m, int32_t k,
                   int32_t *out_ids, float *out_dist);
int hnswgpu_rerank_dev(hnswgpu_index *idx, const float *d_Q, int32_t nq, const int32_t *d_cand, int32_t m,
                       int32_t k, int32_t *d_out_ids, float *d_out_dist, void *stream);
/* out[q * n + row] = distance(query q, base row) for every row: batch-cosine-distances
 * (simd_optimized.clj:176-184) over a whole query batch, and -- on an index whose rows are a projection
 * matrix with metric DOT -- P-HNSW's project-vector-simd (pcaf.clj:47-80; out = -projection).
 * Cosine / dot with nq >= 16 runs on the MFMA tile kernel (tile order), otherwise the GEMV kernel. */
int hnswgpu_dense_distances(hnswgpu_index *idx, const float *Q, int32_t nq, float *out);
int hnswgpu_dense_distances_dev(hnswgpu_index *idx, const float *d_Q, int32_t nq, float *d_out, void *stream);

/* ---- persistence ------------------------------------------------------------------------------------------
 * One flat binary file (header + plain arrays; layout in hnsw-clj_amd/csrc/persist.hip) with the base
 * vectors, the graph and the IVF lists -- replaces helper/index-io's save-index / load-index, an EDN
 * pr-str of every node (src/hnsw/helper/index_io.clj:10-80) and api/save, api/load-index which throw
 * (src/hnsw/api.clj:40-50).  String ids are the caller's to store.  load re-validates the graph. */
int hnswgpu_save(hnswgpu_index *idx, const char *path);
int hnswgpu_load(const char *path, int32_t device, hnswgpu_index **out);

/* ---- measurement --------------------------------------------------------------------------------------
 * With profiling on, the dominant kernel of each search call is bracketed by hipEvents on the
 * launch stream.  which: 0 = IVF list scan, 1 = HNSW traversal, 2 = k-means assignment scan.
 * Returns the accumulated kernel ms and launch count since the last reset (forces a stream sync). */
int hnswgpu_set_profiling(hnswgpu_index *idx, int32_t on);
/* Process-wide counters of the kernel variants launches have taken since the library was loaded -- so that a test (or a
 * benchmark line) can say WHICH path produced the results it checked, not only that the rule would have chosen it. */
#define HNSWGPU_COUNT_BOUNDS_TWO_BLOCKS 0 /* IVF bounds pass with two 32-query column blocks per staged row (stream_bounds_kernel<.., QB = 2>) */
#define HNSWGPU_COUNT_HNSW_SOLO 1         /* small HNSW launches that spread a query over several CUs (solo_kernels.hpp) */
#define HNSWGPU_COUNT_HNSW_HELPERS 2      /* small HNSW launches of the round-2 helper kernel */
#define HNSWGPU_COUNT_HNSW_REJECTION 3    /* HNSW traversal launches with the int8 rejection test on */
#define HNSWGPU_COUNT_HNSW_PLAIN 4        /* ... and with every neighbour evaluated in f32 */
#define HNSWGPU_COUNT_HNSW_WAVE 5         /* large HNSW launches on the one-wave-per-query kernel with the admission buffer (wave_kernels.hpp) */
#define HNSWGPU_COUNT_ROUTE_TAIL_WAVES 6   /* IVF routing tails launched with one wave per query (ivf_route_tail_wave_kernel) */
#define HNSWGPU_COUNT_N 8
int hnswgpu_launch_count(int32_t which, int64_t *out);
/* The HNSW traversal decides most neighbours (those that cannot enter a full result list, ultra_fast.clj:195-198) from
 * an int8 copy of the rows: a lower bound of the distance that is already >= the list's worst needs no f32 row
 * (hnsw-clj_amd/csrc/kernels.hpp: quantize_rows_kernel; results and counters are unchanged by construction).  This
 * entry returns those bounds for query q[dim] against rows ids[0..m) -- out[i] <= the distance hnswgpu_batch_distances
 * reports for the same pair, NaN where the test abstains -- so the property can be checked from outside.
 * hnswgpu_set_rejection_test: mode 0 = off (no int8 copy is made: saves n * dim bytes; the bounds entry then fails),
 * 1 = launches of at least two queries per CU, where the traversal is bandwidth-bound, and only for dim >= 128 (an int8
 * row of a shorter vector saves no cache line) (default), 2 = every launch, every dim.  The same setting decides whether
 * the IVF lists get their int8 copy (and with it the half-precision copy, below) for the bounds pass of the list scan;
 * in mode 1 the first IVF search of a handle measures once what those bounds separate on its rows (32 list rows as
 * queries) and keeps the bounds pass only if it leaves less than a quarter of the candidates -- on rows without cluster
 * structure (i.i.d. gaussian) it leaves everything, and the handle takes the plain f32 scans instead; mode 2 forces it
 * (every batch size; k <= 256, fewer on very long lists: see hnswgpu_ivf_search).
 * HNSWGPU_PREFILTER=<mode> in the environment sets the default of new handles.  Results never depend on the mode. */
int hnswgpu_rejection_bounds(hnswgpu_index *idx, const float *q, const int32_t *ids, int32_t m, float *out);
/* The same with the UPPER bounds beside them (out_ub, may be NULL): out_lb[i] <= distance <= out_ub[i].  The IVF search's
 * bounds pass (hnsw-clj_amd/csrc/stream_kernels.hpp) derives a query's threshold from upper bounds -- k candidates whose
 * upper bound is at most tau put the k-th nearest distance at or below tau -- and drops candidates whose lower bound is
 * above it: both sides have to hold for its result to be the full f32 scan's. */
int hnswgpu_distance_bounds(hnswgpu_index *idx, const float *q, const int32_t *ids, int32_t m, float *out_lb,
                            float *out_ub);
/* Batches with 1.5 M candidates and more (48 queries x 32 lists of ~1000 rows) put a second filter between the int8 bounds and the f32 rows: a HALF-precision copy of
 * the list rows (fp16 with a power-of-two scale per row, 2 * dim bytes per row, made with the int8 copy; HNSWGPU_IVF_HALF=0
 * in the environment leaves it out) whose bounds are ~70 times narrower -- the survivors of the int8 pass meet it first,
 * and f32 rows are fetched for little more than k candidates per query instead of ~3 % of the probed lists.  This entry
 * reports those bounds for `m` rows of the LIST order (positions as hnswgpu_get_ivf's list_ids numbers them) against one
 * query, by the kernel the searches run: out_lb[i] <= distance <= out_ub[i], NaN where a row or the query has no bound. */
int hnswgpu_ivf_half_bounds(hnswgpu_index *idx, const float *q, const int32_t *list_rows, int32_t m, float *out_lb,
                            float *out_ub);
/* Large batches (from 512 queries and half a query per list; rows of whole 128-element steps) put the
 * half-precision rows of a query's NEAREST list -- where nearly all of its int8 survivors sit -- through the matrix
 * cores, once per list for all the queries it is nearest to (v_mfma_f32_16x16x32_f16, the query split into two fp16
 * planes), instead of fetching a half row per (query, survivor).  This entry reports those bounds for the list rows
 * [row_begin, row_end) (positions in list order) against `nq` queries, by the kernel the searches run:
 * out_lb[q * len + r] <= distance(q, row_begin + r) <= out_ub[q * len + r], NaN where a row or the query has no bound. */
int hnswgpu_ivf_home_bounds(hnswgpu_index *idx, const float *Q, int32_t nq, int64_t row_begin, int64_t row_end,
                            float *out_lb, float *out_ub);
int hnswgpu_set_rejection_test(hnswgpu_index *idx, int32_t mode);
/* While profiling is on the traversal counts the neighbours it evaluated and the f32 rows it had to fetch for them
 * (everything with the test off): the bytes a search really moved = neighbours * (int8 row + 16 B) + f32_rows * 4 * dim. */
int hnswgpu_get_rejection_stats(hnswgpu_index *idx, int64_t *f32_rows, int64_t *neighbours, int32_t reset);
/* Mode 1 measures what the traversal's int8 test decides on THIS graph (its first large launch counts f32 rows fetched /
 * neighbours evaluated; nothing blocks: a later launch reads the counters) and evaluates every neighbour in f32 from then
 * on where the test left more than 65 % of the rows to fetch (HNSWGPU_TUNE_HNSW_CALIBRATE_PCT) -- on such rows the int8 stage
 * costs bytes instead of saving them.  state: 0 = not measured yet, 1 = measured, not yet read, 2 = decided; off: the
 * verdict; frac: f32 rows / neighbours of the measured launch.  Results never depend on it. */
int hnswgpu_hnsw_rejection_state(hnswgpu_index *idx, int32_t *state, int32_t *off, double *frac);
int hnswgpu_get_profile(hnswgpu_index *idx, int32_t which, double *total_ms, int64_t *launches, int32_t reset);

/* ---- tuning and test switches --------------------------------------------------------------------------------------
 * One process-wide table of 64-bit values.  NONE of them changes what a search returns within the contract of the path
 * it selects (ids, distance bits, tie order -- the GEMV and the MFMA summation orders differ in the last bits, and
 * TILE_PAIRS / TILE / IVF_CODES / PREFILTER choose between them exactly as the batch size otherwise does): they pick
 * between equivalent schedules, or shrink scratch buffers so that tests reach the fallback paths at test size
 * (tests/test_gpu_parity.py).  The library calls getenv for SIX names only, once, when it is loaded (the keys marked
 * `env`); no search path reads the environment.  HNSWGPU_TUNE_DEFAULT as the value restores a key's default.
 * Diagnostic switches that make results WRONG on purpose (kernels with their epilogue cut out, for ablation timing)
 * exist only in -DHG_DIAG builds (tools/build_stamps.sh), not in this library. */
#define HNSWGPU_TUNE_DEFAULT INT64_MIN
#define HNSWGPU_TUNE_TILE_PAIRS 0 /* (query, list) pairs per list beyond which a cosine / dot IVF batch takes the f32 MFMA tile scan (k-ordered sums) instead of the GEMV-order paths; 0 = the handle's own rule (never on handles with int8 + half-precision list rows, 12 without).  env HNSWGPU_TILE_PAIRS */
#define HNSWGPU_TUNE_PREFILTER 1 /* default hnswgpu_set_rejection_test mode of NEW handles (0 / 1 / 2).  env HNSWGPU_PREFILTER */
#define HNSWGPU_TUNE_IVF_HALF 2 /* 0 = no half-precision copy of the IVF list rows (+50 % of the base).  env HNSWGPU_IVF_HALF */
#define HNSWGPU_TUNE_IVF_CALIBRATE 3 /* 0 = mode-1 handles skip the first-search measurement of what the int8 bounds separate.  env HNSWGPU_IVF_CALIBRATE */
#define HNSWGPU_TUNE_BUILD_THREADS 4 /* host threads of the HNSW linker (0 = min(16, cores)); the graph does not depend on it.  env HNSWGPU_BUILD_THREADS */
#define HNSWGPU_TUNE_PREFETCH 5 /* helper workgroups per query of small HNSW launches (0 = none; default 8 for launches that spread a query over several CUs, 4 for the round-2 helpers).  env HNSWGPU_PREFETCH */
#define HNSWGPU_TUNE_SEED_BOUNDS 6 /* 0 = every k-means++ round a full f32 pass (A/B) */
#define HNSWGPU_TUNE_TILE_WGS 7 /* target workgroup count of the tile scan's work list */
#define HNSWGPU_TUNE_TILE_PERSIST 8 /* tiles per work item of the persistent tile scan (0 = one workgroup per item) */
#define HNSWGPU_TUNE_STREAM_BUCKET 9 /* capacity of a list's bucket of (query, list) pairs; tests: tiny buckets force the fallback */
#define HNSWGPU_TUNE_STREAM_WGS 10 /* target workgroup count of the bounds pass */
#define HNSWGPU_TUNE_FINISH_ORDER 11 /* batch size from which queries are served in the order of their nearest list (0 = never) */
#define HNSWGPU_TUNE_STREAM_CAP 12 /* survivors per query that fit; tests: a tiny list forces the f32-scan fallback */
#define HNSWGPU_TUNE_STREAM_MID 13 /* batch size from which the half-precision pass runs (-1 = by candidate count, 0 = never, 1 = always) */
#define HNSWGPU_TUNE_STREAM_NARROW 14 /* bounds-pass epilogue: -1 auto, 0 lane = query, 1 lane = row */
#define HNSWGPU_TUNE_FINISH_ADAPT 15 /* 0 = short survivor lists are not spread over all waves (A/B) */
#define HNSWGPU_TUNE_FINISH_BISECT 16 /* smallest k whose final merge bisects the key space (A/B) */
#define HNSWGPU_TUNE_FINISH_SLICES 17 /* workgroups per query of the finish kernel (0 = auto) */
#define HNSWGPU_TUNE_FINISH_SPAN 18 /* entries a finish wave looks at per step (0 = auto) */
#define HNSWGPU_TUNE_STREAM_HEAVY 19 /* 0 = no separate service of heavy queries */
#define HNSWGPU_TUNE_STREAM_HEAVY_MEAN 20 /* a heavy query has this many times the batch's mean survivors (default 4) */
#define HNSWGPU_TUNE_STREAM_HEAVY_MIN 21 /* ... and at least this many (default 4096) */
#define HNSWGPU_TUNE_MID_SLICES 22 /* workgroups per query of the half-precision pass (0 = auto) */
#define HNSWGPU_TUNE_MID_COMPACT 23 /* 0 = the half-precision pass does not compact the list (A/B) */
#define HNSWGPU_TUNE_IVF_CODES 24 /* 0 = never the survivor stream, N = from N queries per batch (default 1) */
#define HNSWGPU_TUNE_SCAN_ORDER 25 /* 0 = GEMV list scans never run in list order (A/B) */
#define HNSWGPU_TUNE_IVF_FUSED 26 /* 0 = never the fused two-launch search, 1 = small batches (default), 2 = every GEMV-path batch */
#define HNSWGPU_TUNE_IVF_GROUP 27 /* 0 = never the register-row group kernel, 2 = from half a pair per list */
#define HNSWGPU_TUNE_STREAM_ROUTE 28 /* largest batch routed by the one-launch routing kernel (default 12) */
#define HNSWGPU_TUNE_STREAM_GROUP 29 /* queries from which the bounds pass groups the pairs by list (default 5) */
#define HNSWGPU_TUNE_ROUTE_GROUP 30 /* queries from which a GEMV-order batch routes through the group kernel (default 1024) */
#define HNSWGPU_TUNE_MID_WIDE 31 /* 0 = never eight waves per query in the half-precision pass (A/B) */
#define HNSWGPU_TUNE_MERGE_W 32 /* waves per query of merge_topk_kernel (0 = auto) */
#define HNSWGPU_TUNE_SCAN_BLOCKS 33 /* target workgroup count of the GEMV scan (0 = auto) */
#define HNSWGPU_TUNE_ROUTE_WGS 34 /* target workgroup count of the routing distance pass */
#define HNSWGPU_TUNE_TILE 35 /* -1 auto, 0 never the MFMA tile kernel, 1 whenever possible */
#define HNSWGPU_TUNE_SELECT_W 36 /* waves per query of select_topk_kernel (0 = auto) */
#define HNSWGPU_TUNE_HNSW_NW 37 /* waves per query of the traversal kernel: 1 / 2 / 4 (0 = by batch size) */
#define HNSWGPU_TUNE_VIS_GLOBAL 38 /* 1 = the traversal's visited set in HBM stamps whatever the index size (tests) */
#define HNSWGPU_TUNE_PF_HINTS 39 /* unexpanded list entries kept evaluated ahead of the traversal by its helpers (solo launches: the fetchers' window, default 8 below ef 256 and 16 from there; round-2 helpers: entries posted per expansion, default 4) */
#define HNSWGPU_TUNE_PF_EVAL 40 /* 0 = the helpers only warm the L2 (A/B) */
#define HNSWGPU_TUNE_ZEROCOPY 41 /* 0 = small synchronous HNSW calls stage through copies instead of mapped pinned memory (A/B) */
#define HNSWGPU_TUNE_BUILD_TIMING 42 /* 1 = hnswgpu_hnsw_build prints where its time went to stderr */
#define HNSWGPU_TUNE_BUILD_BATCH 43 /* largest insertion batch of hnswgpu_hnsw_build (default 16384; a batch never exceeds 1/8 of the graph it is searched against) */
#define HNSWGPU_TUNE_STREAM_HOME 44 /* the home-list pass of large IVF batches (every list once through the matrix cores in half precision for all the queries it is nearest to): -1 from 512 queries and half a query per list, 0 never, 1 whenever the batch is served in the order of its nearest lists */
#define HNSWGPU_TUNE_HOME_CHUNK 45 /* rows per work item of the home-list pass (a multiple of 64; 0 = auto) */
#define HNSWGPU_TUNE_HOME_DEPTH 46 /* operand loads in flight per wave of the home-list pass: 4 / 8 / 12 / 16 / 24, the largest that divides the 32-element steps of a row and does not exceed this (default 12) */
#define HNSWGPU_TUNE_HOME_STRAYS 47 /* home-list batches: a query the bounds pass appended no more than this many candidates to skips the per-survivor half-precision pass (the finish kernel fetches their f32 rows; default 32) */
#define HNSWGPU_TUNE_ROUTE_MFMA 48 /* centroid distances of IVF batches on the f32 matrix cores in the GEMV order (the same bits): -1 from 256 queries (cosine / dot, rows of 256 / 512 / 768 elements), 0 never, 1 whenever possible, > 1 that many slices of the table per group of 16 queries */
#define HNSWGPU_TUNE_STREAM_WIDE2 49 /* bounds pass of large batches with several 32-query column blocks per group (a staged list row meets 64 / 128 queries): -1 = two blocks from 256 (query, list) pairs per list, 0 never, 1 = two blocks and 4 = four blocks (measured slower: A/B, tests) whenever the wide deferring epilogue runs */
#define HNSWGPU_TUNE_SOLO 50 /* small HNSW launches, one query over several CUs (an owner workgroup keeps the reference's order, helper workgroups evaluate and chase ahead of it): 1 = from ef 96 (default), 2 = always, 0 = never (the round-2 helpers) */
#define HNSWGPU_TUNE_SOLO_CHASE 51 /* 0 = the helpers only evaluate what the owner asks for (A/B) */
#define HNSWGPU_TUNE_SOLO_SLOTS 52 /* log2 of the slots per query of the helpers' node-keyed tables (0 = auto) */
#define HNSWGPU_TUNE_HNSW_CALIBRATE 53 /* 0 = rejection mode 1 never measures what the traversal's int8 test decides (it then stays on for every large launch) */
#define HNSWGPU_TUNE_HNSW_CALIBRATE_PCT 54 /* the int8 test of the traversal is switched off for a graph when it leaves more than this many percent of the neighbours' f32 rows to fetch (default 65) */
#define HNSWGPU_TUNE_HNSW_WAVE 55 /* large HNSW launches on the one-wave-per-query kernel with the admission buffer: 1 = launches that fill the chip with one wave per query, and every launch from ef 640 (default), 0 = never (A/B), 2 = every launch it can serve (tests) */
#define HNSWGPU_TUNE_FINISH_DIRECT 56 /* small IVF batches (16 and more finish workgroups per query): survivor lists of up to this many entries hand a key per survivor straight to the query's last workgroup instead of per-wave / per-workgroup top-k lists (default 1024 = the most; 0 = never: A/B) */
#define HNSWGPU_TUNE_WORKLIST_FOLD 57 /* 0 = the bounds pass's work list of a small IVF batch stays a launch of its own instead of extra workgroups of the routing tail's launch (A/B) */
#define HNSWGPU_TUNE_SEED_HALF 58 /* 0 = a query's first threshold from f32 rows of its nearest list even where the half-precision copy exists (A/B; default 1: the k-th smallest upper bound of the sampled rows' half-precision copies, half the bytes) */
#define HNSWGPU_TUNE_BUILD_KEEP_ROWS 59 /* 0 = the builder's heuristic selection fetches the already selected rows again for every candidate instead of keeping them in registers (A/B; the graph does not depend on it) */
#define HNSWGPU_TUNE_QUERY_WAVES 60 /* the per-query kernels of large IVF batches (home-list selection) with one WAVE per query, four queries per workgroup, instead of a workgroup per query: -1 from 2048 queries, 0 never, 1 wherever a wave can serve a query (k <= 64) */
#define HNSWGPU_TUNE_COUNT 61
int hnswgpu_set_tuning(int32_t key, int64_t value);
int hnswgpu_get_tuning(int32_t key, int64_t *value, int32_t *is_set);

/* ---- ONE index over several GPUs (hnsw-clj_amd/csrc/group.hip) ---------------------------------------------------
 * The reference shards inside one process: search-partitioned scatters a query to its partitions, takes a top-k per
 * partition, concatenates, sorts and takes k (src/hnsw/ann/partition/partitioned_hnsw.clj:149-196).  A group gives the
 * JVM host the same from one call: it owns one engine handle per device and every device works on its own stream.
 *   hnswgpu_group_create: `devices` = the GPUs of the group (one GPU may be named several times: several handles on it).
 *   hnswgpu_group_set_ivf: ONE IVF-FLAT index (base rows, centroids, lists as for hnswgpu_set_ivf) -- centroids
 *     replicated, whole inverted lists dealt to the devices balanced by row count (longest list first onto the least
 *     loaded device), every device routes a batch identically and scans the probed lists it holds.
 *   hnswgpu_group_ivf_search: the per-device top-k lists are moved to devices[0] by peer copies (nq * k * 12 bytes per
 *     device, xGMI between GPUs) and merged by (distance, position in the candidate stream of the WHOLE index): ids,
 *     distances and tie order are bit for bit those of hnswgpu_ivf_search on the unsharded index (ivf_flat.clj:291-294).
 *   hnswgpu_group_hnsw_build / _hnsw_search: contiguous row ranges, one independent sub-graph per device
 *     (= PartitionedHNSWIndex, partitioned_hnsw.clj:23-27), each searched with the full k; merge by distance, ties to the
 *     lower device (the reference's stable sort of the concatenation).
 *   hnswgpu_group_member: the sub-index device i holds (info / export; do not destroy it).
 * Row ids in and out are rows of the caller's base matrix.  Calls on one group are serialised. */
typedef struct hnswgpu_group hnswgpu_group;
int hnswgpu_group_create(const int32_t *devices, int32_t ndev, int32_t dim, int32_t metric, hnswgpu_group **out);
int hnswgpu_group_destroy(hnswgpu_group *g);
int hnswgpu_group_info(const hnswgpu_group *g, int32_t *ndev, int64_t *n, int32_t *kind, int64_t *rows_per_device);
int hnswgpu_group_set_ivf(hnswgpu_group *g, const float *base, int64_t n, const float *centroids, int32_t nlist,
                          const int64_t *list_off, const int32_t *list_ids);
int hnswgpu_group_ivf_search(hnswgpu_group *g, const float *Q, int32_t nq, int32_t k, int32_t nprobe, int32_t *out_ids,
                             float *out_dist);
int hnswgpu_group_hnsw_build(hnswgpu_group *g, const float *base, int64_t n, int32_t M, int32_t ef_construction,
                             int64_t seed);
int hnswgpu_group_hnsw_search(hnswgpu_group *g, const float *Q, int32_t nq, int32_t k, int32_t ef, int32_t *out_ids,
                              float *out_dist);
hnswgpu_index *hnswgpu_group_member(hnswgpu_group *g, int32_t i);

#ifdef __cplusplus
}
#endif
#endif /* HNSWGPU_H */
