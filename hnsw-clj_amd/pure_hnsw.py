"""Mirror of ``hnsw.ann.graph.pure-hnsw`` (src/hnsw/ann/graph/pure_hnsw.clj; README ``hnsw.hnsw-search``): the
API-compatible wrapper (``build-index`` / ``search-knn index q k [mode]`` / ``index-info``) over the HNSW engine.

Modes (:136-140): turbo 50 / fast 100 / balanced 200 / accurate 300 / precise 500.  In the reference these ef
values are written into ``[:params :ef]`` but ``graph/search-knn`` never reads them (it uses (max k 50),
graph.clj:304 -- SURVEY fact 9), so every mode searches with the same breadth.  The default reproduces what the
reference DOES (``honour_modes=False``: ef = (max k 50) whatever the mode -- a drop-in must return the reference's
results); ``honour_modes=True`` opts in to what the presets' doc-string promises.

The build is ``graph/insert`` (src/hnsw/graph.clj:239-295) as far as an adjacency row of at most 2M / M ids allows: links
chosen by ``get-neighbors-heuristic`` (:162-198) on the device, an over-full neighbour list re-selected by it and the
dropped edges removed from both lists (``prune-connections``, :208-232) -- ``hnswgpu_hnsw_build_ex`` with
HNSWGPU_BUILD_HEURISTIC | HNSWGPU_BUILD_SYMMETRIC.  (The reference links a new node to EVERY candidate of its second
search, :280-287, and only prunes the neighbours; include/hnswgpu.h says why this build selects the node's own links by
the same heuristic.)  ``closest_m=True`` gives ``hnsw.ultra-fast``'s plain closest-m lists instead."""
from . import ultra_fast
from .ultra_fast import cosine_distance_ultra

MODE_EF = {"turbo": 50, "fast": 100, "balanced": 200, "accurate": 300, "precise": 500}


class PureHNSWIndex:
    def __init__(self, graph, params):
        self.graph = graph
        self.params = params

    def close(self):
        self.graph.close()


def build_pure_hnsw_index(data, M=16, ef_construction=200, ef=200, distance_fn=cosine_distance_ultra,
                          show_progress=True, closest_m=False, **kw):
    """pure_hnsw.clj:38-123 over graph/insert (graph.clj:239-295)"""
    if not closest_m and kw.get("graph") is None:
        kw.setdefault("heuristic", True)
        kw.setdefault("symmetric", True)
    g = ultra_fast.build_index(data, M=M, ef_construction=ef_construction, distance_fn=distance_fn,
                               show_progress=show_progress, **kw)
    return PureHNSWIndex(g, {"M": M, "ef-construction": ef_construction, "ef": ef})


def search_pure_hnsw(index, query_vec, k, mode="balanced", ef=None, honour_modes=False):
    """pure_hnsw.clj:129-152"""
    want = MODE_EF.get(mode, ef or 200)
    return ultra_fast.search_knn(index.graph, query_vec, k, ef=max(want, k) if honour_modes else None)


def build_index(data, **opts):
    return build_pure_hnsw_index(data, **opts)


def search_knn(index, query_vec, k, mode="balanced", honour_modes=False):
    """pure_hnsw.clj:163-175"""
    return search_pure_hnsw(index, query_vec, k, mode=mode, honour_modes=honour_modes)


def index_info(index):
    """pure_hnsw.clj:177-190"""
    g = index.graph.index.get_graph() if index.graph.index.n else None
    n = index.graph.index.n
    edges = int((g.l0_adj >= 0).sum() + (g.up_adj >= 0).sum()) if g is not None else 0
    return {"type": "Pure HNSW Index", "vectors": n, "nodes": n,
            "entry-point": index.graph.ids[g.entry] if g is not None else None,
            "avg-edges-per-node": edges / n if n else 0.0, "params": dict(index.params)}
