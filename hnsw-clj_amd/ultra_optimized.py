"""Mirror of ``hnsw.ultra-optimized`` (src/hnsw/wip/ultra_optimized.clj; README namespace of the headline
benchmark).  In the reference it IS hnsw.ultra-fast with ``simd-optimized/cosine-distance`` as the
``:distance-fn`` (:124-138 delegate to base/build-index, :282-286 to base/search-knn; the pools / off-heap /
mmap scaffolding :140-278 is unused), so here it is the same GPU engine with the same default metric."""
from . import simd_optimized, ultra_fast


def parallel_build_index(data, M=16, ef_construction=200, distance_fn=simd_optimized.cosine_distance, num_threads=None,
                         show_progress=True, **kw):
    """ultra_optimized.clj:124-138 (``num_threads`` is accepted and ignored: the build is batched on the GPU)."""
    return ultra_fast.build_index(data, M=M, ef_construction=ef_construction, distance_fn=distance_fn,
                                  show_progress=show_progress, **kw)


def search_optimized(graph, query_vec, k):
    """ultra_optimized.clj:282-286"""
    return ultra_fast.search_knn(graph, query_vec, k)


build_index = parallel_build_index  # :346
search = search_optimized           # :347
search_batch = ultra_fast.search_batch
