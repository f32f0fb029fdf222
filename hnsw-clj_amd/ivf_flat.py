"""Mirror of ``hnsw.ann.partition.ivf-flat`` (src/hnsw/ann/partition/ivf_flat.clj).

``build_index(data, num_partitions=24, distance_fn=..., max_iterations=10)`` runs k-means++ /
Lloyd on the device; ``search_knn(index, q, k, mode)`` / ``search_ivf_flat(index, q, k, mode=...,
num_probes=...)`` scan the probed lists with the HIP scan kernel.
"""
import random

import numpy as np

from . import engine
from .ultra_fast import _metric_of, _split, cosine_distance_ultra

# ivf_flat.clj:243-247
MODE_CONFIGS = {
    "turbo": {"num_probes": 1, "use_centroids": False},
    "fast": {"num_probes": 2, "use_centroids": True},
    "balanced": {"num_probes": 4, "use_centroids": True},
    "accurate": {"num_probes": 8, "use_centroids": True},
    "precise": {"num_probes": 12, "use_centroids": True},
}


class IVFFlatIndex:
    """ivf_flat.clj:22-27: partitions / centroids / norms live on the device behind ``index``."""

    def __init__(self, index, ids, distance_fn, num_partitions):
        self.index = index
        self.ids = ids
        self.distance_fn = distance_fn
        self.num_partitions = num_partitions

    def close(self):
        self.index.close()


def build_ivf_flat_index(data, num_partitions=24, distance_fn=cosine_distance_ultra, show_progress=True,
                         partition_method="kmeans", max_iterations=10, seed=42, device=0):
    """ivf_flat.clj:137-211"""
    metric = _metric_of(distance_fn)
    ids, base = _split(data)
    if len(ids) == 0:
        raise ValueError("cannot partition an empty dataset")
    idx = engine.Index(base, metric, device)
    if partition_method == "kmeans":
        idx.ivf_build(num_partitions, max_iterations, seed)
    else:
        raise ValueError("unsupported :partition-method %r (only :kmeans is served by the device build)" % (partition_method,))
    return IVFFlatIndex(idx, ids, distance_fn, num_partitions)


def build_index(data, **opts):
    """ivf_flat.clj:300-303"""
    return build_ivf_flat_index(data, **opts)


def _format(index, ids_row, d_row):
    return [{"id": index.ids[i], "distance": float(d)} for i, d in zip(ids_row, d_row) if i >= 0]


def search_ivf_flat(index, query_vec, k, mode="balanced", num_probes=None, use_centroids=None):
    """ivf_flat.clj:236-294.  A preset mode wins over num_probes exactly as in the reference (:249-251);
    pass mode=None (or any non-preset) to use num_probes."""
    cfg = MODE_CONFIGS.get(mode) or {"num_probes": num_probes or 4,
                                     "use_centroids": True if use_centroids is None else use_centroids}
    q = np.asarray(query_vec, np.float32)
    single = q.ndim == 1
    Q = q[None, :] if single else q
    npb = int(cfg["num_probes"])
    if cfg["use_centroids"]:
        ids, d = index.index.ivf_search(Q, int(k), npb)
    else:  # (take num-probes (shuffle (range num-partitions))) :271-272
        probes = np.stack([np.array(random.sample(range(index.num_partitions), min(npb, index.num_partitions)),
                                    np.int32) for _ in range(len(Q))])
        ids, d = index.index.ivf_search_lists(Q, int(k), probes)
    out = [_format(index, ids[i], d[i]) for i in range(len(Q))]
    return out[0] if single else out


def search_knn(index, query_vec, k, mode="balanced"):
    """ivf_flat.clj:305-317"""
    return search_ivf_flat(index, query_vec, k, mode=mode)


def search_batch(index, queries, k, mode="balanced", num_probes=None):
    return search_ivf_flat(index, np.asarray(queries, np.float32), k, mode=mode, num_probes=num_probes)


def index_info(index):
    """ivf_flat.clj:319-327"""
    return {"type": "IVF-FLAT Index", "vectors": index.index.n, "partitions": index.num_partitions,
            "avg-partition-size": index.index.n / index.num_partitions, "method": "k-means++"}
