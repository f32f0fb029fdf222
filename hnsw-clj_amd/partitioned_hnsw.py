"""Mirror of ``hnsw.ann.partition.partitioned-hnsw`` (src/hnsw/ann/partition/partitioned_hnsw.clj): the rows are
split into ``num_partitions`` contiguous slices of the (optionally shuffled) data, every slice gets its own HNSW
graph, a search asks every partition for ``k-per-partition`` results and keeps the k best of the concatenation.

On the device a partition is one engine handle (its rows + its graph in HBM); a query batch is searched in every
partition by the traversal kernel and the P result lists are merged by ``hnswgpu_merge_lists_dev`` (a stable
sort of the concatenation, like Collections/sort at :187-196).  The same object row-shards across GPUs with
``sharded.ShardedSearcher`` (one sub-graph per GPU).

Deviation: the reference shuffles with the unseeded ``clojure.core/shuffle`` (:82); here the permutation comes
from ``numpy.random.default_rng(seed)`` so that a build is reproducible.
"""
import math

import numpy as np

from . import engine, ultra_fast
from .ultra_fast import cosine_distance_ultra


class PartitionedHNSWIndex:
    """partitioned_hnsw.clj:23-27"""

    def __init__(self, partitions, rows, ids, num_partitions, shuffle_enabled, search_mode, metadata):
        self.partitions = partitions          # engine.Index per partition
        self.rows = rows                      # per partition: data position of each local row (np.int32)
        self.ids = ids                        # data position -> caller's id
        self.num_partitions = num_partitions
        self.shuffle_enabled = shuffle_enabled
        self.search_mode = search_mode
        self.metadata = metadata
        self._dev_rows = None

    def close(self):
        for p in self.partitions:
            p.close()


def build_partitioned_hnsw(data, num_partitions=8, shuffle=True, show_progress=False, search_mode="lightning",
                           distance_fn=cosine_distance_ultra, max_connections=16, ef_construction=50, seed=42,
                           device=0):
    """partitioned_hnsw.clj:46-143"""
    import time

    t0 = time.time()
    metric = ultra_fast._metric_of(distance_fn)
    ids, base = ultra_fast._split(data)
    n = len(ids)
    order = np.random.default_rng(seed).permutation(n).astype(np.int32) if shuffle else np.arange(n, dtype=np.int32)
    size = max(1, math.ceil(n / num_partitions))           # :87 partition-size, :88 partition-all
    rows = [order[i:i + size] for i in range(0, n, size)]
    parts = []
    for r in rows:
        p = engine.Index(base[r], metric, device)
        p.hnsw_build(max_connections, ef_construction, seed)
        parts.append(p)
    if show_progress:
        print("All partitions built in %.2f seconds" % (time.time() - t0))
    return PartitionedHNSWIndex(parts, rows, ids, num_partitions, shuffle, search_mode,
                                {"build-time": (time.time() - t0) * 1e3, "total-vectors": n,
                                 "distance-fn": distance_fn})


def k_per_partition(mode, num_partitions, k):
    """:158-162 (lightning), :204 (ultra), :236 (turbo)"""
    if mode == "ultra":
        return 2 if num_partitions <= 8 else 1
    if mode == "turbo":
        return min(int(k), 5)
    return 3 if num_partitions <= 8 else (2 if num_partitions <= 32 else 1)


def search_batch_dev(index, Q, k, mode=None):
    """Q: [nq, dim] float32 CUDA tensor -> (data positions int32 [nq, k] (-1 padded), distances [nq, k])."""
    import torch

    mode = mode or index.search_mode
    kpp = k_per_partition(mode, index.num_partitions, k)
    nq, P = Q.shape[0], len(index.partitions)
    if index._dev_rows is None:
        index._dev_rows = [torch.from_numpy(r.astype(np.int64)).to(Q.device) for r in index.rows]
    ids = torch.empty((P, nq, kpp), dtype=torch.int32, device=Q.device)
    d = torch.empty((P, nq, kpp), dtype=torch.float32, device=Q.device)
    for p, part in enumerate(index.partitions):
        part.hnsw_search_dev(Q, kpp, 0, out=(ids[p], d[p]))          # ef = (max k' 50), ultra_fast.clj:355
        loc = ids[p].to(torch.int64)
        ids[p] = torch.where(loc >= 0, index._dev_rows[p][loc.clamp(min=0)], loc).to(torch.int32)
    return engine.merge_lists_dev(ids, d, int(k))


def _to_maps(index, ids, d):
    return [[{"id": index.ids[i], "distance": float(x)} for i, x in zip(ri, rd) if i >= 0]
            for ri, rd in zip(ids.tolist(), d.tolist())]


def search_batch(index, queries, k, mode=None):
    import torch

    queries = np.ascontiguousarray(queries, np.float32)
    if len(queries) == 0:
        return []
    if not index.partitions:
        return [[] for _ in queries]
    dev = torch.device("cuda", index.partitions[0].device)
    ids, d = search_batch_dev(index, torch.from_numpy(queries).to(dev), k, mode)
    return _to_maps(index, ids.cpu(), d.cpu())


def search_partitioned_lightning(index, query_vec, k):
    """:149-196"""
    return search_knn(index, query_vec, k, "lightning")


def search_partitioned_ultra(index, query_vec, k):
    """:198-231"""
    return search_knn(index, query_vec, k, "ultra")


def search_partitioned_turbo(index, query_vec, k):
    """:233-256"""
    return search_knn(index, query_vec, k, "turbo")


def build_index(data, **opts):
    """:262-270"""
    return build_partitioned_hnsw(data, **opts)


def search_knn(index, query_vec, k, mode=None):
    """:272-288; an unknown mode falls back to :lightning"""
    q = np.asarray(query_vec, np.float32).reshape(1, -1)
    res = search_batch(index, q, k, mode or index.search_mode)
    return res[0] if res else []


def index_info(index):
    """:290-301"""
    n = index.metadata["total-vectors"]
    return {"type": "Partitioned HNSW Index", "partitions": index.num_partitions, "vectors": n,
            "build-time": index.metadata["build-time"], "shuffle": index.shuffle_enabled,
            "search-mode": index.search_mode, "avg-partition-size": n / index.num_partitions}
