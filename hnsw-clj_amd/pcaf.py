"""Mirror of ``hnsw.ann.dimreduct.pcaf`` (src/hnsw/ann/dimreduct/pcaf.clj, "P-HNSW"): a Gaussian random projection
to ``n_components`` dimensions, a brute-force cosine search in the projected space for min(k-filter, 3k)
candidates, and an exact cosine re-rank of those candidates in the original space (:196-253).

Device composition over the C ABI:
  * the projection matrix (Random(42) gaussians / sqrt(target-dim), :36-45) is an engine handle with metric DOT;
    ``hnswgpu_dense_distances`` against it IS project-vector-simd (:47-80) for a whole batch (out = -projection);
  * phase 1 is ``hnswgpu_exact_knn`` on the handle that holds the projected rows (:213-233);
  * phase 2 is ``hnswgpu_rerank`` on the handle that holds the original rows (:239-253).
The reference runs both phases in float32 through the Vector API, whose lane-reduction order is hardware
dependent; the kernels' fixed f32 orders are within the same 1e-7-relative envelope.
"""
import numpy as np

from . import engine
from .datagen import JavaRandom
from .ultra_fast import _split

MODE_K_FILTER = {"turbo": 16, "fast": 24, "balanced": 32, "accurate": 48, "precise": 64}  # :283-290


def create_random_projection(original_dim, target_dim):
    """pcaf.clj:33-45 -> float32 [target_dim, original_dim]"""
    g = JavaRandom(42).next_gaussians(target_dim * original_dim)
    scale = np.float32(1.0 / np.sqrt(target_dim))
    return (scale * g.astype(np.float32)).reshape(target_dim, original_dim)


class PCAFIndex:
    """pcaf.clj:108-114"""

    def __init__(self, projection, proj_index, low_index, high_index, ids, k_filter):
        self.projection = projection      # float32 [target_dim, original_dim]
        self.proj_index = proj_index      # rows = projection matrix, metric DOT
        self.low_index = low_index        # projected rows, cosine
        self.high_index = high_index      # original rows (float32), cosine
        self.ids = ids
        self.k_filter = k_filter
        self.dimension_reduction = projection.shape[1] / projection.shape[0]

    def close(self):
        for h in (self.proj_index, self.low_index, self.high_index):
            h.close()


def project_dev(index_or_proj, X):
    """project-vector-simd (:47-80) for every row of the CUDA tensor X."""
    proj = index_or_proj.proj_index if isinstance(index_or_proj, PCAFIndex) else index_or_proj
    return -proj.dense_distances_dev(X)


def build_pcaf_index(data, n_components=100, k_filter=32, show_progress=False, num_threads=4, device=0):
    """pcaf.clj:116-194 (``num_threads`` kept for signature parity)"""
    import torch

    ids, base = _split(data)
    dim = base.shape[1]
    P = create_random_projection(dim, n_components)
    proj = engine.Index(P, engine.DOT, device)
    dev = torch.device("cuda", device)
    low = torch.empty((len(ids), n_components), dtype=torch.float32, device=dev)
    for i in range(0, len(ids), 65536):
        x = torch.from_numpy(base[i:i + 65536]).to(dev)
        low[i:i + 65536] = project_dev(proj, x)
    low_index = engine.Index(low, engine.COSINE, device)      # stays in HBM: device-to-device
    high_index = engine.Index(base, engine.COSINE, device)
    if show_progress:
        print("P-HNSW: %d vectors, %d -> %d dims" % (len(ids), dim, n_components))
    return PCAFIndex(P, proj, low_index, high_index, ids, k_filter)


def search_batch_dev(index, Q, k, k_filter=None):
    """Q: [nq, dim] float32 CUDA tensor -> (data positions [nq, k] int32, exact cosine distances [nq, k])."""
    kf = min(int(k_filter or index.k_filter), 3 * int(k))            # :236
    kf = max(1, min(kf, index.low_index.n))
    cand, _ = index.low_index.exact_knn_dev(project_dev(index, Q), kf)
    return index.high_index.rerank_dev(Q, cand, int(k))


def search_batch(index, queries, k, mode=None):
    import torch

    queries = np.ascontiguousarray(queries, np.float32)
    if len(queries) == 0:
        return []
    if index.high_index.n == 0:
        return [[] for _ in queries]
    dev = torch.device("cuda", index.high_index.device)
    ids, d = search_batch_dev(index, torch.from_numpy(queries).to(dev), k, MODE_K_FILTER.get(mode))
    return [[{"id": index.ids[i], "distance": float(x)} for i, x in zip(ri, rd) if i >= 0]
            for ri, rd in zip(ids.cpu().tolist(), d.cpu().tolist())]


def search_pcaf_parallel(index, query_vec, k):
    """pcaf.clj:196-253"""
    return search_knn(index, query_vec, k)


def build_index(data, **opts):
    """:259-268"""
    return build_pcaf_index(data, **opts)


def search_knn(index, query_vec, k, mode=None):
    """:270-291"""
    q = np.asarray(query_vec, np.float32).reshape(1, -1)
    res = search_batch(index, q, k, mode)
    return res[0] if res else []


def index_info(index):
    """:293-302"""
    return {"type": "P-HNSW (SIMD Optimized)", "original-dim": index.projection.shape[1],
            "reduced-dim": index.projection.shape[0], "reduction-ratio": index.dimension_reduction,
            "k-filter": index.k_filter, "vectors": index.high_index.n, "optimization": "HIP"}


def cleanup(index):
    """:304-308"""
    index.close()
