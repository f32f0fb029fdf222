"""Mirror of ``hnsw.simd-optimized`` (src/hnsw/simd_optimized.clj): the metric functions every
reference benchmark passes as ``:distance-fn``, plus its batch seams -- all evaluated by the HIP
kernels of libhnswgpu.so (there is no CPU path here)."""
import numpy as np

from . import engine
from .engine import COSINE, DOT, L2


def _metric_fn(metric, name, doc):
    def fn(v1, v2, device=0):
        return engine.pair_distance(metric, v1, v2, device)

    fn.metric = metric
    fn.__name__ = name
    fn.__doc__ = doc
    return fn


cosine_distance = _metric_fn(COSINE, "cosine_distance", "simd_optimized.clj:145-153 (-> simd.clj:129-147)")
euclidean_distance = _metric_fn(L2, "euclidean_distance", "simd_optimized.clj:155-160 (-> simd.clj:149-160), rooted")


def dot_product(a, b, device=0):
    """simd_optimized.clj:283-293 -- the raw dot product (similarity, not a distance)."""
    return -engine.pair_distance(DOT, a, b, device)


dot_product.metric = DOT


def _batch(metric, query, vectors, device):
    vectors = np.ascontiguousarray(vectors, np.float32)
    if vectors.ndim != 2:
        raise ValueError("vectors must be (m, dim)")
    if len(vectors) == 0:
        return np.empty(0, np.float32)
    with engine.Index(vectors, metric, device) as idx:
        return idx.batch_distances(query)


def batch_cosine_distances(query, vectors, device=0):
    """simd_optimized.clj:176-179: distances from ``query`` to every vector, one gather-dot launch."""
    return _batch(COSINE, query, vectors, device)


def batch_euclidean_distances(query, vectors, device=0):
    """simd_optimized.clj:181-184"""
    return _batch(L2, query, vectors, device)


def precompute_norms(vectors, device=0):
    """simd_optimized.clj:206-216"""
    vectors = np.ascontiguousarray(vectors, np.float32)
    with engine.Index(vectors, COSINE, device) as idx:
        return idx.norms()


def top_k_distances(distance_fn, query, vectors, k, device=0):
    """simd_optimized.clj:271-280 -> [[idx, dist], ...] ascending, ties by index (stable sort)."""
    metric = getattr(distance_fn, "metric", None)
    if metric is None:
        raise ValueError("distance_fn must be one of this module's metric functions (it selects the device metric)")
    vectors = np.ascontiguousarray(vectors, np.float32)
    if len(vectors) == 0:
        return []
    with engine.Index(vectors, metric, device) as idx:
        ids, d = idx.exact_knn(query, min(k, len(vectors)))
    sign = -1.0 if metric == DOT and distance_fn is dot_product else 1.0
    return [[int(i), float(sign * x)] for i, x in zip(ids[0], d[0]) if i >= 0]
