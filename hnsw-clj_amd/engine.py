"""Index handle over libhnswgpu.so -- thin, typed wrapper of the C ABI (include/hnswgpu.h).

numpy arrays go through the host-pointer entry points; torch CUDA tensors go through the ``_dev``
entry points on torch's current stream (zero copy).  Nothing here computes distances on the CPU.
"""
import ctypes as C

import numpy as np

from . import _native
from ._native import COSINE, DOT, L2, check, debug_counter, get_tuning, lib, set_tuning  # noqa: F401

METRICS = {"cosine": COSINE, "l2": L2, "euclidean": L2, "dot": DOT, COSINE: COSINE, L2: L2, DOT: DOT}
PROF_IVF_SCAN, PROF_HNSW, PROF_ASSIGN = 0, 1, 2
BUILD_SEQUENTIAL, BUILD_HEURISTIC, BUILD_SYMMETRIC, BUILD_EXTEND = 1, 2, 4, 8     # include/hnswgpu.h: HNSWGPU_BUILD_*


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _queries(Q, dim):
    Q = _f32(Q)
    if Q.ndim == 1:
        Q = Q[None, :]
    if Q.ndim != 2 or Q.shape[1] != dim:
        raise ValueError("queries must be (nq, %d), got %s" % (dim, Q.shape))
    return Q


class Graph:
    """Flat HNSW graph (layout documented in include/hnswgpu.h)."""

    def __init__(self, levels, l0_adj, up_off, up_adj, M, entry, max_level):
        self.levels = np.ascontiguousarray(levels, np.int32)
        self.l0_adj = np.ascontiguousarray(l0_adj, np.int32)
        self.up_off = np.ascontiguousarray(up_off, np.int64)
        self.up_adj = np.ascontiguousarray(up_adj, np.int32)
        self.M = int(M)
        self.n = len(self.levels)
        self.M0 = self.l0_adj.shape[1] if self.l0_adj.ndim == 2 else (self.l0_adj.size // max(self.n, 1))
        self.entry = int(entry)
        self.max_level = int(max_level)


class Index:
    """Base vectors resident in HBM + optional HNSW graph and IVF lists."""

    def __init__(self, base, metric="cosine", device=0):
        self.metric = METRICS[metric]
        self.device = int(device)
        self._h = C.c_void_p(None)
        if hasattr(base, "is_cuda") and base.is_cuda:  # torch tensor already in HBM
            import torch

            assert base.dtype == torch.float32 and base.dim() == 2
            base = base.contiguous()
            self.n, self.dim = int(base.shape[0]), int(base.shape[1])
            st = torch.cuda.current_stream(base.device).cuda_stream
            check(lib().hnswgpu_create_dev(base.data_ptr(), self.n, self.dim, self.dim, self.metric,
                                           base.device.index or 0, st, C.byref(self._h)))
            self.device = base.device.index or 0
        else:
            base = _f32(base)
            if base.ndim != 2:
                raise ValueError("base must be (n, dim)")
            self.n, self.dim = base.shape
            check(lib().hnswgpu_create(_p(base), self.n, self.dim, self.metric, self.device, C.byref(self._h)))

    @classmethod
    def load(cls, path, device=0):
        """hnswgpu_load: base + graph + IVF lists from one flat binary file."""
        self = cls.__new__(cls)
        self._h = C.c_void_p(None)
        self.device = int(device)
        check(lib().hnswgpu_load(str(path).encode(), self.device, C.byref(self._h)))
        n, dim, metric = C.c_int64(), C.c_int32(), C.c_int32()
        check(lib().hnswgpu_info(self._h, C.byref(n), C.byref(dim), C.byref(metric), None, None))
        self.n, self.dim, self.metric = n.value, dim.value, metric.value
        return self

    def save(self, path):
        check(lib().hnswgpu_save(self._h, str(path).encode()))

    @property
    def has_graph(self):
        g = C.c_int32()
        check(lib().hnswgpu_info(self._h, None, None, None, C.byref(g), None))
        return bool(g.value)

    # -- lifetime
    def close(self):
        if self._h:
            lib().hnswgpu_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        check(lib().hnswgpu_sync(self._h))

    # -- distance seams
    def batch_distances(self, q, ids=None, m=None):
        q = _f32(q).reshape(-1)
        if len(q) != self.dim:
            raise ValueError("query has %d elements, index dim is %d" % (len(q), self.dim))
        if ids is not None:
            ids = np.ascontiguousarray(ids, np.int32)
            m = len(ids)
        elif m is None:
            m = self.n
        out = np.empty(m, np.float32)
        check(lib().hnswgpu_batch_distances(self._h, _p(q), _p(ids), m, _p(out)))
        return out

    def set_rejection_test(self, mode):
        """0 = off, 1 = large batches (default), 2 = every launch: int8 rejection test of the HNSW traversal."""
        check(lib().hnswgpu_set_rejection_test(self._h, int(mode)))

    def hnsw_rejection_state(self):
        """hnswgpu_hnsw_rejection_state: (state, off, frac) -- what rejection mode 1 has measured on this graph."""
        st, off, fr = C.c_int32(0), C.c_int32(0), C.c_double(0.0)
        check(lib().hnswgpu_hnsw_rejection_state(self._h, C.byref(st), C.byref(off), C.byref(fr)))
        return st.value, bool(off.value), fr.value

    def rejection_stats(self, reset=True):
        """(f32 rows fetched, neighbours evaluated) by the HNSW traversals since the last reset, while profiling was on."""
        a, b = C.c_int64(0), C.c_int64(0)
        check(lib().hnswgpu_get_rejection_stats(self._h, C.byref(a), C.byref(b), 1 if reset else 0))
        return a.value, b.value

    def rejection_bounds(self, q, ids):
        """Lower bounds of d(q, row) from the int8 rows of the HNSW traversal's rejection test (NaN = no bound)."""
        q = _f32(q).reshape(-1)
        if len(q) != self.dim:
            raise ValueError("query has %d elements, index dim is %d" % (len(q), self.dim))
        ids = np.ascontiguousarray(ids, np.int32)
        out = np.empty(len(ids), np.float32)
        check(lib().hnswgpu_rejection_bounds(self._h, _p(q), _p(ids), len(ids), _p(out)))
        return out

    def distance_bounds(self, q, ids):
        """(lower, upper) bounds of d(q, row) from the int8 rows: lower <= distance <= upper, NaN = no bound."""
        q = _f32(q).reshape(-1)
        if len(q) != self.dim:
            raise ValueError("query has %d elements, index dim is %d" % (len(q), self.dim))
        ids = np.ascontiguousarray(ids, np.int32)
        lb, ub = np.empty(len(ids), np.float32), np.empty(len(ids), np.float32)
        check(lib().hnswgpu_distance_bounds(self._h, _p(q), _p(ids), len(ids), _p(lb), _p(ub)))
        return lb, ub

    def ivf_half_bounds(self, q, list_rows):
        """(lower, upper) bounds of d(q, list row) from the half-precision list rows (batches of 1.5 M candidates and more
        filter the int8 survivors with them); rows are positions in list order.  NaN = no bound."""
        q = _f32(q).reshape(-1)
        if len(q) != self.dim:
            raise ValueError("query has %d elements, index dim is %d" % (len(q), self.dim))
        rows = np.ascontiguousarray(list_rows, np.int32)
        lb, ub = np.empty(len(rows), np.float32), np.empty(len(rows), np.float32)
        check(lib().hnswgpu_ivf_half_bounds(self._h, _p(q), _p(rows), len(rows), _p(lb), _p(ub)))
        return lb, ub

    def ivf_home_bounds(self, Q, row_begin, row_end):
        """(lower, upper) bounds [nq][row_end - row_begin] of d(query, list row) from the matrix-core half-precision pass of
        large batches (ivf_home_kernel); rows are positions in list order.  NaN = no bound."""
        Q = _f32(Q).reshape(-1, self.dim)
        m = int(row_end) - int(row_begin)
        lb, ub = np.empty((len(Q), m), np.float32), np.empty((len(Q), m), np.float32)
        check(lib().hnswgpu_ivf_home_bounds(self._h, _p(Q), len(Q), int(row_begin), int(row_end), _p(lb), _p(ub)))
        return lb, ub

    def norms(self):
        out = np.empty(self.n, np.float32)
        check(lib().hnswgpu_norms(self._h, _p(out)))
        return out

    def exact_knn(self, Q, k):
        Q = _queries(Q, self.dim)
        ids = np.empty((len(Q), k), np.int32)
        d = np.empty((len(Q), k), np.float32)
        check(lib().hnswgpu_exact_knn(self._h, _p(Q), len(Q), k, _p(ids), _p(d)))
        return ids, d

    def rerank(self, Q, cand, k):
        """Per query: exact distances to its candidate rows cand[q] (-1 = skip), stable sort, first k."""
        Q = _queries(Q, self.dim)
        cand = np.ascontiguousarray(cand, dtype=np.int32)
        if cand.ndim != 2 or cand.shape[0] != len(Q):
            raise ValueError("cand must be [nq, m]")
        ids = np.empty((len(Q), k), np.int32)
        d = np.empty((len(Q), k), np.float32)
        check(lib().hnswgpu_rerank(self._h, _p(Q), len(Q), _p(cand), cand.shape[1], k, _p(ids), _p(d)))
        return ids, d

    def dense_distances(self, Q):
        """[nq, n] distances from every query to every row."""
        Q = _queries(Q, self.dim)
        out = np.empty((len(Q), self.n), np.float32)
        check(lib().hnswgpu_dense_distances(self._h, _p(Q), len(Q), _p(out)))
        return out

    # -- HNSW
    def set_graph(self, g):
        check(lib().hnswgpu_set_graph(self._h, _p(g.levels), _p(g.l0_adj), g.M0, _p(g.up_off), _p(g.up_adj), g.M,
                                      g.entry, g.max_level))

    def hnsw_build(self, M=16, ef_construction=200, seed=42, sequential=False, heuristic=False, symmetric=False,
                   extend=False):
        """hnswgpu_hnsw_build_ex: batched closest-m insertion by default (ultra_fast.clj:216-299); sequential = the
        reference's insert-single order and start level (the CPU restatement's graph, oracle.c, edge for edge); heuristic / symmetric / extend =
        neighbour selection of src/hnsw/graph.clj:162-232."""
        flags = (BUILD_SEQUENTIAL if sequential else 0) | (BUILD_HEURISTIC if heuristic else 0) | \
                (BUILD_SYMMETRIC if symmetric else 0) | (BUILD_EXTEND if extend else 0)
        check(lib().hnswgpu_hnsw_build_ex(self._h, M, ef_construction, seed, flags))

    def hnsw_add(self, rows, ef_construction=200, seed=42):
        """insert-single on the live index (ultra_fast.clj:216-275): `rows` join the base and the installed graph; returns
        the row ids they got."""
        rows = _queries(rows, self.dim)
        first = self.n
        check(lib().hnswgpu_hnsw_add(self._h, _p(rows), len(rows), ef_construction, seed))
        self.n += len(rows)
        return np.arange(first, self.n, dtype=np.int32)

    def get_graph(self):
        M, M0, ent, mx = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        blocks = C.c_int64()
        check(lib().hnswgpu_graph_sizes(self._h, C.byref(M), C.byref(M0), C.byref(blocks), C.byref(ent),
                                        C.byref(mx)))
        levels = np.zeros(self.n, np.int32)
        l0 = np.full((self.n, M0.value), -1, np.int32)
        up_off = np.zeros(self.n + 1, np.int64)
        up = np.full(blocks.value * M.value, -1, np.int32)
        check(lib().hnswgpu_get_graph(self._h, _p(levels), _p(l0), _p(up_off), _p(up)))
        return Graph(levels, l0, up_off, up, M.value, ent.value, mx.value)

    def hnsw_search(self, Q, k, ef=0, want_stats=False):
        Q = _queries(Q, self.dim)
        ids = np.empty((len(Q), k), np.int32)
        d = np.empty((len(Q), k), np.float32)
        stats = np.zeros((len(Q), 2), np.int64) if want_stats else None
        check(lib().hnswgpu_hnsw_search(self._h, _p(Q), len(Q), k, int(ef or 0), _p(ids), _p(d), _p(stats)))
        return (ids, d, stats) if want_stats else (ids, d)

    # -- IVF
    def ivf_build(self, nlist=24, max_iterations=10, seed=42):
        check(lib().hnswgpu_ivf_build(self._h, nlist, max_iterations, seed))

    def set_ivf(self, centroids, list_off, list_ids):
        cen = _f32(centroids)
        off = np.ascontiguousarray(list_off, np.int64)
        ids = np.ascontiguousarray(list_ids, np.int32)
        check(lib().hnswgpu_set_ivf(self._h, _p(cen), cen.shape[0], _p(off), _p(ids)))

    def ivf_stream_state(self):
        """1 if this handle's first-search measurement switched the survivor stream off (measures now if it has not yet)."""
        off = C.c_int32(0)
        check(lib().hnswgpu_ivf_stream_state(self._h, C.byref(off)))
        return off.value

    def ivf_set_stream_state(self, off):
        """Install a verdict without measuring: the shards of one index share one (see include/hnswgpu.h)."""
        check(lib().hnswgpu_ivf_set_stream_state(self._h, 1 if off else 0))

    def set_ivf_shard(self, centroids, list_off, list_ids, global_list_len):
        """This handle holds some of the inverted lists of a larger index (see include/hnswgpu.h)."""
        cen = _f32(centroids)
        off = np.ascontiguousarray(list_off, np.int64)
        ids = np.ascontiguousarray(list_ids, np.int32)
        gl = np.ascontiguousarray(global_list_len, np.int64)
        if len(gl) != cen.shape[0] or len(off) != cen.shape[0] + 1:
            raise ValueError("need nlist global lengths and nlist + 1 offsets")
        check(lib().hnswgpu_set_ivf_shard(self._h, _p(cen), cen.shape[0], _p(off), _p(ids), _p(gl)))

    @property
    def nlist(self):
        nl = C.c_int32()
        check(lib().hnswgpu_info(self._h, None, None, None, None, C.byref(nl)))
        return nl.value

    def get_ivf(self):
        nl = self.nlist
        cen = np.empty((nl, self.dim), np.float32)
        off = np.empty(nl + 1, np.int64)
        ids = np.empty(self.n, np.int32)
        check(lib().hnswgpu_get_ivf(self._h, _p(cen), _p(off), _p(ids)))
        return cen, off, ids

    def kmeans_assign(self, centroids):
        cen = _f32(centroids)
        a = np.empty(self.n, np.int32)
        d = np.empty(self.n, np.float32)
        check(lib().hnswgpu_kmeans_assign(self._h, _p(cen), cen.shape[0], _p(a), _p(d)))
        return a, d

    def list_means(self, list_off, list_ids):
        off = np.ascontiguousarray(list_off, np.int64)
        ids = np.ascontiguousarray(list_ids, np.int32)
        out = np.empty((len(off) - 1, self.dim), np.float32)
        check(lib().hnswgpu_list_means(self._h, len(off) - 1, _p(off), _p(ids), _p(out)))
        return out

    def list_sums(self, list_off, list_ids):
        """f64 column sums of every list (compute-centroid without the division)."""
        off = np.ascontiguousarray(list_off, np.int64)
        ids = np.ascontiguousarray(list_ids, np.int32)
        out = np.empty((len(off) - 1, self.dim), np.float64)
        check(lib().hnswgpu_list_sums(self._h, len(off) - 1, _p(off), _p(ids), _p(out)))
        return out

    def kmeanspp(self, nlist, seed=42):
        out = np.empty(nlist, np.int32)
        check(lib().hnswgpu_kmeanspp(self._h, nlist, seed, _p(out)))
        return out

    def ivf_search(self, Q, k, nprobe, want_probes=False):
        Q = _queries(Q, self.dim)
        ids = np.empty((len(Q), k), np.int32)
        d = np.empty((len(Q), k), np.float32)
        pr = np.empty((len(Q), nprobe), np.int32) if want_probes else None
        check(lib().hnswgpu_ivf_search(self._h, _p(Q), len(Q), k, nprobe, _p(ids), _p(d), _p(pr)))
        return (ids, d, pr) if want_probes else (ids, d)

    def ivf_search_lists(self, Q, k, probes):
        Q = _queries(Q, self.dim)
        probes = np.ascontiguousarray(probes, np.int32).reshape(len(Q), -1)
        ids = np.empty((len(Q), k), np.int32)
        d = np.empty((len(Q), k), np.float32)
        check(lib().hnswgpu_ivf_search_lists(self._h, _p(Q), len(Q), k, probes.shape[1], _p(probes), _p(ids), _p(d)))
        return ids, d

    # -- zero-copy device entry points (torch tensors, torch's current stream)
    def _dev_args(self, Q, k, out=None):
        import torch

        assert Q.is_cuda and Q.dtype == torch.float32 and Q.dim() == 2 and Q.shape[1] == self.dim
        Q = Q.contiguous()
        if out is not None:      # caller-owned result tensors: nothing is allocated on the call path
            ids, d = out
        else:
            ids = torch.empty((Q.shape[0], k), dtype=torch.int32, device=Q.device)
            d = torch.empty((Q.shape[0], k), dtype=torch.float32, device=Q.device)
        st = torch.cuda.current_stream(Q.device).cuda_stream
        return Q, ids, d, st

    def hnsw_search_dev(self, Q, k, ef=0, out=None, stats=None):
        Q, ids, d, st = self._dev_args(Q, k, out)
        check(lib().hnswgpu_hnsw_search_dev(self._h, Q.data_ptr(), Q.shape[0], k, int(ef or 0), ids.data_ptr(),
                                            d.data_ptr(), stats.data_ptr() if stats is not None else None, st))
        return ids, d

    def ivf_search_dev(self, Q, k, nprobe, out=None):
        Q, ids, d, st = self._dev_args(Q, k, out)
        check(lib().hnswgpu_ivf_search_dev(self._h, Q.data_ptr(), Q.shape[0], k, nprobe, ids.data_ptr(),
                                           d.data_ptr(), st))
        return ids, d

    def ivf_search_shard_dev(self, Q, k, nprobe):
        """ivf_search_dev + every result's position in the candidate stream of the whole (sharded) index."""
        import torch

        Q, ids, d, st = self._dev_args(Q, k)
        order = torch.empty((Q.shape[0], k), dtype=torch.int32, device=Q.device)   # uint32 bits
        check(lib().hnswgpu_ivf_search_shard_dev(self._h, Q.data_ptr(), Q.shape[0], k, nprobe, ids.data_ptr(),
                                                 d.data_ptr(), order.data_ptr(), st))
        return ids, d, order

    def exact_knn_dev(self, Q, k, out=None):
        Q, ids, d, st = self._dev_args(Q, k, out)
        check(lib().hnswgpu_exact_knn_dev(self._h, Q.data_ptr(), Q.shape[0], k, ids.data_ptr(), d.data_ptr(), st))
        return ids, d

    def rerank_dev(self, Q, cand, k, out=None):
        import torch

        Q, ids, d, st = self._dev_args(Q, k)
        assert cand.is_cuda and cand.dtype == torch.int32 and cand.dim() == 2 and cand.shape[0] == Q.shape[0]
        cand = cand.contiguous()
        if out is not None:
            ids, d = out
        check(lib().hnswgpu_rerank_dev(self._h, Q.data_ptr(), Q.shape[0], cand.data_ptr(), cand.shape[1], k,
                                       ids.data_ptr(), d.data_ptr(), st))
        return ids, d

    def dense_distances_dev(self, Q, out=None):
        import torch

        Q, _, _, st = self._dev_args(Q, 1)
        if out is None:
            out = torch.empty((Q.shape[0], self.n), dtype=torch.float32, device=Q.device)
        check(lib().hnswgpu_dense_distances_dev(self._h, Q.data_ptr(), Q.shape[0], out.data_ptr(), st))
        return out

    # -- measurement
    def set_profiling(self, on=True):
        check(lib().hnswgpu_set_profiling(self._h, 1 if on else 0))

    def get_profile(self, which, reset=True):
        ms, cnt = C.c_double(), C.c_int64()
        check(lib().hnswgpu_get_profile(self._h, which, C.byref(ms), C.byref(cnt), 1 if reset else 0))
        return ms.value, cnt.value


def pair_distance(metric, a, b, device=0):
    """(distance-fn a b) on the device: one pair through the same kernel as everything else."""
    a, b = _f32(a).reshape(-1), _f32(b).reshape(-1)
    if len(a) != len(b):
        raise ValueError("vectors differ in length (%d vs %d)" % (len(a), len(b)))
    out = C.c_float()
    check(lib().hnswgpu_pair_distance(METRICS[metric], _p(a), _p(b), len(a), device, C.byref(out)))
    return float(out.value)


def merge_topk_dev(ids, dist, out=None):
    """[nshard][nq][k] torch CUDA tensors of global ids / distances -> merged [nq][k]."""
    import torch

    assert ids.is_cuda and ids.dtype == torch.int32 and dist.dtype == torch.float32 and ids.dim() == 3
    ids, dist = ids.contiguous(), dist.contiguous()
    ns, nq, k = ids.shape
    oi = torch.empty((nq, k), dtype=torch.int32, device=ids.device) if out is None else out[0]
    od = torch.empty((nq, k), dtype=torch.float32, device=ids.device) if out is None else out[1]
    st = torch.cuda.current_stream(ids.device).cuda_stream
    check(lib().hnswgpu_merge_topk_dev(ids.device.index or 0, ids.data_ptr(), dist.data_ptr(), ns, nq, k,
                                       oi.data_ptr(), od.data_ptr(), st))
    return oi, od


def merge_keyed_dev(ids, dist, order, out=None):
    """[nshard][nq][k] (global id, distance, order) of the shards of ONE IVF index -> [nq][k] by (distance, order)."""
    import torch

    assert ids.is_cuda and ids.dtype == torch.int32 and dist.dtype == torch.float32 and order.dtype == torch.int32
    assert ids.dim() == 3 and ids.shape == dist.shape == order.shape
    ids, dist, order = ids.contiguous(), dist.contiguous(), order.contiguous()
    ns, nq, k = ids.shape
    oi = torch.empty((nq, k), dtype=torch.int32, device=ids.device) if out is None else out[0]
    od = torch.empty((nq, k), dtype=torch.float32, device=ids.device) if out is None else out[1]
    st = torch.cuda.current_stream(ids.device).cuda_stream
    check(lib().hnswgpu_merge_keyed_dev(ids.device.index or 0, ids.data_ptr(), dist.data_ptr(), order.data_ptr(), ns, nq,
                                        k, oi.data_ptr(), od.data_ptr(), st))
    return oi, od


def merge_lists_dev(ids, dist, k, out=None):
    """[nlists][nq][k_in] torch CUDA tensors (ids -1 padded) -> the k best of every query, [nq][k];
    ties keep the lower list, then the lower rank, first (a stable sort of the concatenation)."""
    import torch

    assert ids.is_cuda and ids.dtype == torch.int32 and dist.dtype == torch.float32 and ids.dim() == 3
    ids, dist = ids.contiguous(), dist.contiguous()
    ns, nq, kin = ids.shape
    oi = torch.empty((nq, k), dtype=torch.int32, device=ids.device) if out is None else out[0]
    od = torch.empty((nq, k), dtype=torch.float32, device=ids.device) if out is None else out[1]
    st = torch.cuda.current_stream(ids.device).cuda_stream
    check(lib().hnswgpu_merge_lists_dev(ids.device.index or 0, ids.data_ptr(), dist.data_ptr(), ns, nq, kin, k,
                                        oi.data_ptr(), od.data_ptr(), st))
    return oi, od


device_count = _native.device_count


class Group:
    """ONE index over several GPUs behind the C ABI (include/hnswgpu.h: hnswgpu_group_*): whole inverted lists of an
    IVF-FLAT index dealt to the devices, or one HNSW sub-graph per device -- the reference's search-partitioned
    (partitioned_hnsw.clj:149-196) as one call.  `devices` may name a GPU several times."""

    def __init__(self, devices, dim, metric="cosine"):
        self.devices = np.ascontiguousarray(devices, np.int32)
        self.dim = int(dim)
        self.metric = METRICS[metric]
        self._h = C.c_void_p(None)
        check(lib().hnswgpu_group_create(_p(self.devices), len(self.devices), self.dim, self.metric, C.byref(self._h)))

    def close(self):
        if self._h:
            check(lib().hnswgpu_group_destroy(self._h))
            self._h = C.c_void_p(None)

    __enter__ = lambda self: self  # noqa: E731

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def info(self):
        nd, n, kind = C.c_int32(0), C.c_int64(0), C.c_int32(0)
        rows = np.zeros(len(self.devices), np.int64)
        check(lib().hnswgpu_group_info(self._h, C.byref(nd), C.byref(n), C.byref(kind), _p(rows)))
        return {"devices": nd.value, "n": n.value, "kind": {0: None, 1: "ivf", 2: "hnsw"}[kind.value], "rows_per_device": rows}

    def set_ivf(self, base, centroids, list_off, list_ids):
        base = _queries(base, self.dim)
        centroids = _queries(centroids, self.dim)
        list_off = np.ascontiguousarray(list_off, np.int64)
        list_ids = np.ascontiguousarray(list_ids, np.int32)
        if len(list_off) != len(centroids) + 1 or len(list_ids) != len(base):
            raise ValueError("list_off must have nlist + 1 entries and list_ids one per base row")
        check(lib().hnswgpu_group_set_ivf(self._h, _p(base), len(base), _p(centroids), len(centroids), _p(list_off), _p(list_ids)))

    def hnsw_build(self, base, M=16, ef_construction=200, seed=42):
        base = _queries(base, self.dim)
        check(lib().hnswgpu_group_hnsw_build(self._h, _p(base), len(base), int(M), int(ef_construction), int(seed)))

    def _search(self, fn, Q, k, param):
        Q = _queries(Q, self.dim)
        ids = np.empty((len(Q), k), np.int32)
        d = np.empty((len(Q), k), np.float32)
        check(fn(self._h, _p(Q), len(Q), int(k), int(param), _p(ids), _p(d)))
        return ids, d

    def ivf_search(self, Q, k, nprobe):
        return self._search(lib().hnswgpu_group_ivf_search, Q, k, nprobe)

    def hnsw_search(self, Q, k, ef):
        return self._search(lib().hnswgpu_group_hnsw_search, Q, k, ef)

    def member_graph(self, i):
        """The sub-graph device i holds (export: get_graph of that sub-index)."""
        h = lib().hnswgpu_group_member(self._h, int(i))
        if not h:
            raise IndexError(i)
        ix = Index.__new__(Index)
        ix._h = C.c_void_p(h)
        ix._borrowed = True
        try:
            nn, dim, metric, hg, nl = C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
            check(lib().hnswgpu_info(ix._h, C.byref(nn), C.byref(dim), C.byref(metric), C.byref(hg), C.byref(nl)))
            ix.n, ix.dim, ix.metric, ix.device = nn.value, dim.value, metric.value, int(self.devices[i])
            return ix.get_graph()
        finally:
            ix._h = C.c_void_p(None)   # borrowed: never destroyed from here, whatever get_graph did
