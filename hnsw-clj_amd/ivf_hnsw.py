"""Mirror of ``hnsw.ann.hybrid.ivf-hnsw`` (src/hnsw/ann/hybrid/ivf_hnsw.clj): k-means partitions (the same
k-means++/Lloyd as IVF-FLAT, :31-150) with one HNSW graph per partition; a search ranks the centroids, searches
the ``num-probes`` nearest partitions' graphs for 2k each, and keeps the k best (:286-325).

Device composition: the k-means runs in ``hnswgpu_ivf_build`` on the full matrix; every partition then becomes its
own engine handle (rows + graph).  A query batch is routed by an exact scan of the centroid table, grouped by
partition, searched with one traversal launch per partition, and merged with ``hnswgpu_merge_lists_dev`` in probe
order (a stable sort of the concatenation, like Collections/sort at :320-324).

``honour_modes``: the reference writes ``:ef-search`` into ``[:params :ef]`` (:316) but ``graph/search-knn`` never
reads it and searches with (max 2k 50) (graph.clj:304).  False (default) reproduces that; True makes
the presets effective.
"""
import numpy as np

from . import engine, ultra_fast
from .ultra_fast import cosine_distance_ultra

MODE_CONFIGS = {  # :291-295
    "turbo": {"num-probes": 1, "ef-search": 50},
    "fast": {"num-probes": 2, "ef-search": 100},
    "balanced": {"num-probes": 3, "ef-search": 150},
    "accurate": {"num-probes": 4, "ef-search": 200},
    "precise": {"num-probes": 5, "ef-search": 300},
}


class IVFHNSWIndex:
    """ivf_hnsw.clj:23-28"""

    def __init__(self, partitions, rows, centroids, cent_index, ids, distance_fn, hnsw_params):
        self.partitions = partitions      # engine.Index per partition, None for an empty one
        self.rows = rows                  # per partition: data position of each local row
        self.centroids = centroids        # [num_partitions, dim] float32
        self.cent_index = cent_index      # engine.Index over the centroid table (routing)
        self.ids = ids
        self.distance_fn = distance_fn
        self.hnsw_params = hnsw_params
        self._dev_rows = None

    def close(self):
        for p in self.partitions:
            if p is not None:
                p.close()
        self.cent_index.close()


def build_ivf_hnsw_index(data, num_partitions=24, distance_fn=cosine_distance_ultra, show_progress=False, M=16,
                         ef_construction=200, max_iterations=10, parallel_build=True, seed=42, device=0):
    """ivf_hnsw.clj:172-280 (``parallel_build`` is accepted for signature parity; the device build is batched)."""
    metric = ultra_fast._metric_of(distance_fn)
    ids, base = ultra_fast._split(data)
    with engine.Index(base, metric, device) as full:
        full.ivf_build(num_partitions, max_iterations, 42)           # Random(42), :37
        cent, off, lids = full.get_ivf()
    parts, rows = [], []
    for p in range(num_partitions):
        r = np.ascontiguousarray(lids[off[p]:off[p + 1]], np.int32)
        rows.append(r)
        if len(r) == 0:
            parts.append(None)                                       # :162-163 empty graph
            continue
        idx = engine.Index(base[r], metric, device)
        idx.hnsw_build(M, ef_construction, seed)
        parts.append(idx)
    if show_progress:
        print("IVF-HNSW: %d vectors in %d partitions" % (len(ids), num_partitions))
    return IVFHNSWIndex(parts, rows, cent, engine.Index(cent, metric, device), ids, distance_fn,
                        {"M": M, "ef-construction": ef_construction, "total-partitions": num_partitions})


def search_batch_dev(index, Q, k, mode="balanced", num_probes=None, ef_search=None, honour_modes=False):
    """Q: [nq, dim] float32 CUDA tensor -> (data positions int32 [nq, k] (-1 padded), distances [nq, k])."""
    import torch

    cfg = MODE_CONFIGS.get(mode) or {"num-probes": num_probes or 3, "ef-search": ef_search or 150}  # :297-299
    P = len(index.partitions)
    nprobe = max(1, min(int(cfg["num-probes"]), P))
    k2 = 2 * int(k)
    ef = max(int(cfg["ef-search"]), k2) if honour_modes else 0
    nq = Q.shape[0]
    if index._dev_rows is None:
        index._dev_rows = [torch.from_numpy(r.astype(np.int64)).to(Q.device) for r in index.rows]
    probes, _ = index.cent_index.exact_knn_dev(Q, nprobe)            # :302-309 centroid ranking, stable
    ids = torch.full((nprobe, nq, k2), -1, dtype=torch.int32, device=Q.device)
    d = torch.full((nprobe, nq, k2), float("inf"), dtype=torch.float32, device=Q.device)
    flat_ids, flat_d = ids.view(nprobe * nq, k2), d.view(nprobe * nq, k2)
    pt = probes.t().contiguous().view(-1)                            # slot r * nq + q holds the r-th probe of query q
    order = torch.argsort(pt, stable=True)
    counts = torch.bincount(pt[order].clamp(min=0), minlength=P).tolist()
    start = int((pt < 0).sum())
    for p in range(P):
        c = counts[p]
        if c == 0:
            continue
        slots = order[start:start + c]
        start += c
        part = index.partitions[p]
        if part is None:
            continue
        li, ld = part.hnsw_search_dev(Q[slots % nq], k2, ef)
        loc = li.to(torch.int64)
        flat_ids[slots] = torch.where(loc >= 0, index._dev_rows[p][loc.clamp(min=0)], loc).to(torch.int32)
        flat_d[slots] = ld
    return engine.merge_lists_dev(ids, d, int(k))


def search_batch(index, queries, k, mode="balanced", **kw):
    import torch

    queries = np.ascontiguousarray(queries, np.float32)
    if len(queries) == 0:
        return []
    dev = torch.device("cuda", index.cent_index.device)
    ids, d = search_batch_dev(index, torch.from_numpy(queries).to(dev), k, mode, **kw)
    return [[{"id": index.ids[i], "distance": float(x)} for i, x in zip(ri, rd) if i >= 0]
            for ri, rd in zip(ids.cpu().tolist(), d.cpu().tolist())]


def search_ivf_hnsw(index, query_vec, k, mode="balanced", num_probes=None, ef_search=None, honour_modes=False):
    """ivf_hnsw.clj:286-325"""
    q = np.asarray(query_vec, np.float32).reshape(1, -1)
    return search_batch(index, q, k, mode, num_probes=num_probes, ef_search=ef_search, honour_modes=honour_modes)[0]


def build_index(data, **opts):
    """:331-334"""
    return build_ivf_hnsw_index(data, **opts)


def search_knn(index, query_vec, k, mode="balanced"):
    """:336-353; a number is the legacy search-percent: num-probes = int(24 * percent)"""
    if isinstance(mode, str):
        return search_ivf_hnsw(index, query_vec, k, mode=mode)
    return search_ivf_hnsw(index, query_vec, k, mode=None, num_probes=int(24 * mode))


def index_info(index):
    """:355-364"""
    total = sum(len(r) for r in index.rows)
    return {"type": "IVF-HNSW Index", "vectors": len(index.ids), "partitions": len(index.partitions),
            "avg-partition-size": total / max(1, len(index.partitions)),
            "hnsw-params": {"M": index.hnsw_params["M"], "ef-construction": index.hnsw_params["ef-construction"]}}
