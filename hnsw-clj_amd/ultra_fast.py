"""Mirror of ``hnsw.ultra-fast`` (src/hnsw/ultra_fast.clj) -- also the engine behind
``hnsw.ultra-optimized`` (src/hnsw/wip/ultra_optimized.clj:124-138,282-286 delegate here).

Same names and argument meaning: ``build_index(data, M=16, ef_construction=200, distance_fn=...)``
takes a seq of ``[id, vector]`` pairs; ``search_knn(graph, query_vec, k)`` returns a list of
``{"id", "distance"}`` ascending, ``[]`` on an empty index, fewer than k when the index is smaller.
The vectors live in HBM as one float32 matrix; traversal runs in the HIP kernel.
"""
import numpy as np

from . import engine
from .engine import COSINE, L2


def cosine_distance_ultra(v1, v2, device=0):
    """ultra_fast.clj:53-95"""
    return engine.pair_distance(COSINE, v1, v2, device)


cosine_distance_ultra.metric = COSINE


def euclidean_distance_ultra(v1, v2, device=0):
    """ultra_fast.clj:43-51 (rooted)"""
    return engine.pair_distance(L2, v1, v2, device)


euclidean_distance_ultra.metric = L2


class UltraGraph:
    """What ultra_fast.clj:104-111's UltraGraph record becomes: the String-id table stays on the host,
    vectors + adjacency live on the device behind ``index``."""

    def __init__(self, index, ids, M, ef_construction, distance_fn):
        self.index = index
        self.ids = ids
        self.M = M
        self.max_M = 2 * M
        self.ef_construction = ef_construction
        self.distance_fn = distance_fn

    def close(self):
        if self.index is not None:
            self.index.close()


def _metric_of(distance_fn):
    m = getattr(distance_fn, "metric", None)
    if m is None:
        raise ValueError(
            ":distance-fn must be one of the engine's metric functions (cosine_distance_ultra, "
            "euclidean_distance_ultra, simd_optimized.cosine_distance/euclidean_distance/dot_product): an arbitrary "
            "host function cannot be evaluated inside the GPU traversal")
    return m


def _split(data):
    ids, vecs = [], []
    for item in data:
        i, v = item  # the reference destructures [id vector] (ultra_fast.clj:318)
        ids.append(i)
        vecs.append(np.asarray(v, dtype=np.float32))
    if vecs:
        dim = len(vecs[0])
        for v in vecs:
            if len(v) != dim:
                raise ValueError("vectors differ in length")
        base = np.stack(vecs).astype(np.float32, copy=False)
    else:
        base = np.zeros((0, 1), np.float32)
    return ids, base


def build_index(data, M=16, ef_construction=200, distance_fn=cosine_distance_ultra, show_progress=True, seed=42,
                device=0, graph=None, sequential=False, heuristic=False, symmetric=False, extend=False):
    """ultra_fast.clj:334-344.  ``graph`` (an engine.Graph) uploads an adjacency built elsewhere instead
    of building one on the device.  ``sequential`` = insert-single's own order and start level (:216-275; slow, the
    parity mode); ``heuristic`` / ``symmetric`` / ``extend`` = the neighbour selection of src/hnsw/graph.clj:162-232
    (what ``pure_hnsw.build_index`` asks for)."""
    metric = _metric_of(distance_fn)
    ids, base = _split(data)
    if show_progress:
        print("Inserting %d elements..." % len(ids))
    idx = engine.Index(base, metric, device)
    if graph is not None:
        idx.set_graph(graph)
    else:
        idx.hnsw_build(M, ef_construction, seed, sequential=sequential, heuristic=heuristic, symmetric=symmetric, extend=extend)
    return UltraGraph(idx, ids, M, ef_construction, distance_fn)


def _format(graph, ids_row, d_row):
    return [{"id": graph.ids[i], "distance": float(d)} for i, d in zip(ids_row, d_row) if i >= 0]


def search_knn(graph, query_vec, k, ef=None):
    """ultra_fast.clj:346-374; ef defaults to (max k 50) (:355)."""
    if graph.index.n == 0:
        return []
    ids, d = graph.index.hnsw_search(query_vec, int(k), ef or 0)
    return _format(graph, ids[0], d[0])


def routed_to_exact_scan(graph, queries, k, ef=None):
    """The crossover rule of ``search_batch(route=True)``: the traversal evaluates E(ef) rows per query, gathered at random;
    the exact scan (hnswgpu_exact_knn: every row once per batch, through the matrix cores) answers at recall 1.0 -- it is the
    cheaper way to at least the same recall once E(ef) >= n / 3 (31k x 768 on one MI355X: 1.5M QPS exact against 0.78M
    through the graph at ef 640, 90k at ef 3200; bench.py: by_distribution.*.routed_qps).  E(ef) is measured once per
    (graph, k, ef) on up to 32 of the queries."""
    key = (int(k), int(ef or 0))
    cache = graph.__dict__.setdefault("_route", {})
    if key not in cache:
        _, _, st = graph.index.hnsw_search(np.asarray(queries, np.float32)[:32], int(k), ef or 0, want_stats=True)
        cache[key] = float(st[:, 0].mean()) >= graph.index.n / 3.0
    return cache[key]


def search_batch(graph, queries, k, ef=None, route=False):
    """All queries in ONE launch: the seam of BatchSearchIndex/search-batch* (api/protocol.clj:58-67).  ``route=True`` (not
    in the reference) answers by the exact scan where that is cheaper than the traversal at this ef (routed_to_exact_scan):
    the neighbours are then the exact ones -- at least as good as the graph's, not necessarily the same."""
    queries = np.asarray(queries, np.float32)
    if len(queries) == 0:
        return []
    if graph.index.n == 0:
        return [[] for _ in queries]
    if route and routed_to_exact_scan(graph, queries, k, ef):
        ids, d = graph.index.exact_knn(queries, int(k))
    else:
        ids, d = graph.index.hnsw_search(queries, int(k), ef or 0)
    return [_format(graph, ids[i], d[i]) for i in range(len(queries))]


def graph_info(graph):
    """ultra_fast.clj:378-384"""
    g = graph.index.get_graph() if graph.index.n else None
    return {"num-elements": graph.index.n, "entry-point": graph.ids[g.entry] if g is not None else None,
            "M": graph.M, "ef-construction": graph.ef_construction}
