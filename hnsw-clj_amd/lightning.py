"""Mirror of ``hnsw.ann.partition.lightning`` (src/hnsw/ann/partition/lightning.clj): equal random partitions
(or k-means++-seeded ones), brute-force scan of a PERCENTAGE of the partitions chosen by centroid distance
or at random.  The scan is the IVF list-scan kernel (lightning.clj:144-162 == ivf_flat.clj:217-234); only the
partitioning and the list selection differ, and those stay on the host side of the ABI."""
import random

import numpy as np

from . import engine
from .ultra_fast import _metric_of, _split, cosine_distance_ultra


class LightningIndex:
    def __init__(self, index, ids, distance_fn, num_partitions):
        self.index, self.ids, self.distance_fn, self.num_partitions = index, ids, distance_fn, num_partitions

    def close(self):
        self.index.close()


def build_lightning_index(data, num_partitions=24, distance_fn=cosine_distance_ultra, show_progress=True,
                          smart_partition=False, device=0, seed=None):
    """lightning.clj:37-142.  smart_partition: k-means++ seeding + one assignment, centroids = partition means
    (this build seeds with the IVF D^2 sampler; the reference's lightning variant samples by D, :104-110)."""
    metric = _metric_of(distance_fn)
    ids, base = _split(data)
    if not ids:
        raise ValueError("cannot partition an empty dataset")
    idx = engine.Index(base, metric, device)
    n = len(ids)
    if smart_partition:
        idx.ivf_build(num_partitions, 0, 42)                      # k-means++ seeds + one assignment
        _, off, lids = idx.get_ivf()
    else:                                                         # (partition-all size (shuffle data)) :123-127
        size = -(-n // num_partitions)
        order = list(range(n))
        random.Random(seed).shuffle(order)
        parts = [order[i:i + size] for i in range(0, n, size)]
        off = np.zeros(len(parts) + 1, np.int64)
        off[1:] = np.cumsum([len(p) for p in parts])
        lids = np.concatenate(parts).astype(np.int32)
    cents = idx.list_means(off, lids)                             # compute-centroid :24-35, on the device
    idx.set_ivf(cents, off, lids)
    return LightningIndex(idx, ids, distance_fn, len(off) - 1)


def _mode_configs(p):
    """lightning.clj:198-229"""
    if p >= 64:
        pct = (0.03, 0.05, 0.10, 0.16, 0.25)
    elif p >= 32:
        pct = (0.05, 0.08, 0.15, 0.25, 0.40)
    elif p == 24:
        pct = (0.08, 0.12, 0.20, 0.33, 0.50)
    else:
        pct = (0.10, 0.15, 0.30, 0.45, 0.60)
    names = ("turbo", "fast", "balanced", "accurate", "precise")
    return {m: {"percent": f, "use_centroids": i >= 2} for i, (m, f) in enumerate(zip(names, pct))}


def search_lightning(index, query_vec, k, search_percent=None, mode=None, use_centroids=None):
    """lightning.clj:189-298"""
    p = index.num_partitions
    if mode:
        cfg = _mode_configs(p)[mode]
        percent, use_c = cfg["percent"], cfg["use_centroids"]
    else:
        percent = search_percent
        use_c = use_centroids or (search_percent >= 0.15 if search_percent else True)   # :233-237
    if percent is None:                                                                  # :246-252
        percent = 0.30 if p <= 16 else 0.20 if p == 24 else 0.15 if p <= 32 else 0.10 if p <= 64 else 0.08
    nsearch = max(1, int(p * percent))
    Q = np.asarray(query_vec, np.float32)[None, :]
    if use_c:
        ids, d = index.index.ivf_search(Q, int(k), nsearch)
    else:
        probes = np.array([random.sample(range(p), nsearch)], np.int32)
        ids, d = index.index.ivf_search_lists(Q, int(k), probes)
    return [[index.ids[i], float(x)] for i, x in zip(ids[0], d[0]) if i >= 0]             # [[id dist] ...] :295-298


def build_index(data, **opts):
    return build_lightning_index(data, **opts)


def search_knn(index, query_vec, k, mode="balanced"):
    return search_lightning(index, query_vec, k, mode=mode)
