"""Mirror of ``hnsw.api.protocol`` (src/hnsw/api/protocol.clj): ANNIndex + BatchSearchIndex over the
two GPU-served index types."""
from . import ivf_flat, ultra_fast


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


class GpuHnswIndex(ANNIndex, BatchSearchIndex):
    def __init__(self, graph):
        self.graph = graph

    def search_knn_star(self, query, k, mode=None):
        return ultra_fast.search_knn(self.graph, query, k)  # modes are ignored by the reference too

    def search_batch_star(self, queries, k, mode=None):
        return ultra_fast.search_batch(self.graph, queries, k)

    def index_info_star(self):
        return ultra_fast.graph_info(self.graph)

    def index_type_star(self):
        return "ultra-fast"


class GpuIvfFlatIndex(ANNIndex, BatchSearchIndex):
    def __init__(self, index):
        self.index = index

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
