"""Mirror of ``hnsw.api.protocol`` (src/hnsw/api/protocol.clj): ANNIndex, BatchSearchIndex, PersistableIndex and the
default helpers (filtered search by post-filtering, :96-101; sequential batch search, :92-95) over the two GPU-served
index types."""
from . import index_io, ivf_flat, ultra_fast


class ANNIndex:
    """protocol.clj:9-28"""

    def search_knn_star(self, query, k, mode):
        raise NotImplementedError

    def index_info_star(self):
        raise NotImplementedError

    def index_type_star(self):
        raise NotImplementedError


class BatchSearchIndex:
    """protocol.clj:58-67"""

    def search_batch_star(self, queries, k, mode):
        raise NotImplementedError


class PersistableIndex:
    """protocol.clj:43-56"""

    def save_index_star(self, filepath):
        raise NotImplementedError


def default_batch_search(index, queries, k, mode):
    """protocol.clj:92-95: one search-knn* per query (what an index without search-batch* gets)."""
    return [index.search_knn_star(q, k, mode) for q in queries]


def default_filtered_search(index, query, k, filter_fn, mode):
    """protocol.clj:96-101: search for 3k candidates, keep those whose id passes the predicate, take k."""
    return [r for r in index.search_knn_star(query, 3 * k, mode) if filter_fn(r["id"])][:k]


class GpuHnswIndex(ANNIndex, BatchSearchIndex, PersistableIndex):
    def __init__(self, graph):
        self.graph = graph

    def save_index_star(self, filepath):
        index_io.save_index(self.graph, filepath)
        return True

    def search_knn_star(self, query, k, mode=None):
        return ultra_fast.search_knn(self.graph, query, k)  # modes are ignored by the reference too

    def search_batch_star(self, queries, k, mode=None):
        return ultra_fast.search_batch(self.graph, queries, k)

    def index_info_star(self):
        return ultra_fast.graph_info(self.graph)

    def index_type_star(self):
        return "ultra-fast"


class GpuIvfFlatIndex(ANNIndex, BatchSearchIndex, PersistableIndex):
    def __init__(self, index):
        self.index = index

    def save_index_star(self, filepath):
        index_io.save_index(self.index, filepath)
        return True

    def search_knn_star(self, query, k, mode="balanced"):
        return ivf_flat.search_knn(self.index, query, k, mode)

    def search_batch_star(self, queries, k, mode="balanced"):
        return ivf_flat.search_batch(self.index, queries, k, mode)

    def index_info_star(self):
        return ivf_flat.index_info(self.index)

    def index_type_star(self):
        return "ivf-flat"


def supports_batch_search(index):
    """protocol.clj:83-86"""
    return isinstance(index, BatchSearchIndex)


def supports_persistence(index):
    """protocol.clj:78-81"""
    return isinstance(index, PersistableIndex)
