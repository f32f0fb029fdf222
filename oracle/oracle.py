"""ctypes/numpy front-end of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

The product (hnsw-clj_amd/) never imports this module.  Each wrapper names the reference
file:line its C function follows; the arithmetic lives in oracle/oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

COSINE, L2, DOT = 0, 1, 2
METRICS = {"cosine": COSINE, "l2": L2, "euclidean": L2, "dot": DOT}
MODE_F64, MODE_DEV, MODE_FAST, MODE_MFMA = 0, 1, 2, 3
GAUSSIAN, UNIFORM, UNIT, CLUSTERED = 0, 1, 2, 3


def build(force=False):
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        d, i32, i64, p = C.c_double, C.c_int, C.c_int64, C.c_void_p
        _lib.orc_cosine_ultra.restype = d
        _lib.orc_cosine_ultra.argtypes = [p, p, i32]
        _lib.orc_cosine_direct.restype = d
        _lib.orc_cosine_direct.argtypes = [p, p, i32]
        _lib.orc_euclid.restype = d
        _lib.orc_euclid.argtypes = [p, p, i32]
        _lib.orc_dot.restype = d
        _lib.orc_dot.argtypes = [p, p, i32]
        _lib.orc_dist_f64.restype = d
        _lib.orc_dist_f64.argtypes = [i32, p, p, i32]
        _lib.orc_dist_f32in.restype = d
        _lib.orc_dist_f32in.argtypes = [i32, p, p, i32]
        _lib.orc_dist_dev.restype = C.c_float
        _lib.orc_dist_dev.argtypes = [i32, p, p, i32, C.c_float, C.c_float]
        _lib.orc_norm_dev.restype = C.c_float
        _lib.orc_norm_dev.argtypes = [p, i32]
        _lib.orc_f32_vector.restype = d
        _lib.orc_f32_vector.argtypes = [i32, p, p, i32, i32, i32]
        _lib.orc_f32_lane_accumulate.restype = d
        _lib.orc_f32_lane_accumulate.argtypes = [i32, p, p, i32, i32, i32]
        _lib.orc_fdlibm_log.restype = d
        _lib.orc_fdlibm_log.argtypes = [d]
        _lib.orc_norms.restype = None
        _lib.orc_norms.argtypes = [p, i64, i32, i32, p]
        _lib.orc_generate_dataset.restype = None
        _lib.orc_generate_dataset.argtypes = [i64, i32, i32, i32, d, i64, p]
        _lib.orc_hnsw_search_batch.restype = d
        _lib.orc_hnsw_search_batch.argtypes = [p, i64, i32, i32, i32, p, p, p, i32, p, p, i32, i32, i32, p, i32, i32,
                                               i32, i32, p, p, p, p]
        _lib.orc_hnsw_build.restype = i32
        _lib.orc_hnsw_build.argtypes = [p, i64, i32, i32, i32, i32, i32, i64, i32, p, p, p, p, i64, p, p]
        _lib.orc_hnsw_build_ex.restype = i32
        _lib.orc_hnsw_build_ex.argtypes = [p, i64, i32, i32, i32, i32, i32, i64, i32, p, p, p, p, i64, p, p, p]
        _lib.orc_exact_knn.restype = d
        _lib.orc_exact_knn.argtypes = [p, i64, i32, i32, i32, p, p, i32, i32, i32, p, p]
        _lib.orc_recall.restype = d
        _lib.orc_recall.argtypes = [p, p, i32, i32]
        _lib.orc_kmeans_assign.restype = None
        _lib.orc_kmeans_assign.argtypes = [p, i64, i32, i32, p, i32, p, p]
        _lib.orc_kmeanspp.restype = None
        _lib.orc_kmeanspp.argtypes = [p, i64, i32, i32, i32, i64, p]
        _lib.orc_ivf_build.restype = None
        _lib.orc_ivf_build.argtypes = [p, i64, i32, i32, i32, i32, i64, p, p]
        _lib.orc_ivf_search.restype = None
        _lib.orc_ivf_search.argtypes = [p, i64, i32, i32, i32, i32, p, p, p, i32, p, p, p, i32, i32, i32, p, p, p]
        _lib.orc_ivf_build_dev.restype = None
        _lib.orc_ivf_build_dev.argtypes = [p, i64, i32, i32, i32, i32, i32, i32, i64, p, p, p]
        _lib.orc_kmeans_assign_f32.restype = None
        _lib.orc_kmeans_assign_f32.argtypes = [p, i64, i32, i32, i32, p, i32, p, p]
        # java.util.Random
        _lib.jr_init.argtypes = [p, i64]
        _lib.jr_next_int.restype = C.c_int32
        _lib.jr_next_int.argtypes = [p]
        _lib.jr_next_int_bound.restype = C.c_int32
        _lib.jr_next_int_bound.argtypes = [p, C.c_int32]
        _lib.jr_next_double.restype = d
        _lib.jr_next_double.argtypes = [p]
        _lib.jr_next_gaussian.restype = d
        _lib.jr_next_gaussian.argtypes = [p]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- distances -------------------------------------------------------------------------------
def cosine_distance_ultra(a, b):
    """src/hnsw/ultra_fast.clj:53-95"""
    a, b = _f64(a), _f64(b)
    return lib().orc_cosine_ultra(_p(a), _p(b), len(a))


def cosine_distance(a, b):
    """simd-optimized/cosine-distance: src/hnsw/simd_optimized.clj:145-153 -> src/hnsw/simd.clj:129-147"""
    a, b = _f64(a), _f64(b)
    return lib().orc_cosine_direct(_p(a), _p(b), len(a))


def euclidean_distance(a, b):
    """src/hnsw/ultra_fast.clj:43-51 == src/hnsw/simd.clj:149-160 (rooted)"""
    a, b = _f64(a), _f64(b)
    return lib().orc_euclid(_p(a), _p(b), len(a))


def dot_product(a, b):
    """src/hnsw/simd_optimized.clj:283-293"""
    a, b = _f64(a), _f64(b)
    return lib().orc_dot(_p(a), _p(b), len(a))


def distance(metric, a, b):
    """f64 distance on float32-valued inputs (what the engine stores)."""
    a, b = _f32(a), _f32(b)
    return lib().orc_dist_f32in(int(metric), _p(a), _p(b), len(a))


def f32_vector(metric, a, b, lanes=8, assoc=0):
    """src/hnsw/simd.clj:26-43,52-71,81-115: the float32 Vector-API form (chunks of `lanes`, f32 lane reduce, f64
    accumulate); assoc: 0 = lanes left to right, 1 = pairwise (reduceLanes leaves it unspecified).  dot returns +dot."""
    a, b = _f32(a), _f32(b)
    return lib().orc_f32_vector(int(metric), _p(a), _p(b), len(a), int(lanes), int(assoc))


def f32_lane_accumulate(metric, a, b, lanes=8, assoc=0):
    """src/hnsw/wip/vector.clj:21-86: f32 lane accumulators, one reduceLanes at the end."""
    a, b = _f32(a), _f32(b)
    return lib().orc_f32_lane_accumulate(int(metric), _p(a), _p(b), len(a), int(lanes), int(assoc))


def norm_dev(v):
    v = _f32(v)
    return float(lib().orc_norm_dev(_p(v), len(v)))


def distance_dev(metric, q, v):
    """Bit-mimic of the HIP kernels' f32 arithmetic (oracle.c section 2)."""
    q, v = _f32(q), _f32(v)
    return float(lib().orc_dist_dev(int(metric), _p(q), _p(v), len(q), norm_dev(q), norm_dev(v)))


def norms(base, mode=MODE_F64):
    base = _f32(base)
    out = np.empty(base.shape[0], np.float32)
    lib().orc_norms(_p(base), base.shape[0], base.shape[1], mode, _p(out))
    return out


# ---- java.util.Random / test/data_generator.clj ------------------------------------------------
class JavaRandom:
    """java.util.Random (public JDK specification)."""

    class _S(C.Structure):
        _fields_ = [("seed", C.c_uint64), ("have", C.c_int), ("nxt", C.c_double)]

    def __init__(self, seed):
        self._s = JavaRandom._S()
        lib().jr_init(C.byref(self._s), int(seed))

    def next_int(self, bound=None):
        if bound is None:
            return lib().jr_next_int(C.byref(self._s))
        return lib().jr_next_int_bound(C.byref(self._s), int(bound))

    def next_double(self):
        return lib().jr_next_double(C.byref(self._s))

    def next_gaussian(self):
        return lib().jr_next_gaussian(C.byref(self._s))


def generate_dataset(size, dim, distribution="gaussian", num_clusters=10, noise_level=0.1, seed=42):
    """test/data_generator.clj:50-87 generate-dataset -> float64 array (size, dim)."""
    code = {"gaussian": GAUSSIAN, "uniform": UNIFORM, "unit": UNIT, "clustered": CLUSTERED}[distribution]
    out = np.empty((size, dim), np.float64)
    lib().orc_generate_dataset(size, dim, code, num_clusters, float(noise_level), int(seed), _p(out))
    return out


# ---- HNSW ------------------------------------------------------------------------------------
class Graph:
    """Array-indexed HNSW graph: the layout include/hnswgpu.h's hnswgpu_set_graph takes."""

    def __init__(self, levels, l0_adj, up_off, up_adj, M, entry, max_level):
        self.levels = np.ascontiguousarray(levels, np.int32)
        self.l0_adj = np.ascontiguousarray(l0_adj, np.int32)
        self.up_off = np.ascontiguousarray(up_off, np.int64)
        self.up_adj = np.ascontiguousarray(up_adj, np.int32)
        self.M = int(M)
        self.M0 = 2 * int(M) if self.l0_adj.ndim == 1 else self.l0_adj.shape[1]
        self.entry = int(entry)
        self.max_level = int(max_level)
        self.n = len(self.levels)


BUILD_FARTHEST, BUILD_HEURISTIC, BUILD_EXTEND, BUILD_UPPER_EFC, BUILD_SYMMETRIC, BUILD_DESCEND = 1, 2, 4, 8, 16, 32
# src/hnsw/graph.clj's builder on the array graph (oracle.c: orc_hnsw_build_ex): heuristic selection, symmetric pruning,
# ef-construction on every layer
BUILD_GRAPH_CLJ = BUILD_HEURISTIC | BUILD_SYMMETRIC | BUILD_UPPER_EFC


def hnsw_build_ex(base, metric=COSINE, M=16, ef_construction=200, seed=42, flags=0, mode=MODE_F64, want_counters=False):
    """Sequential insertion on the skeleton of src/hnsw/ultra_fast.clj:216-344 with the pieces of
    src/hnsw/graph.clj:162-295 that `flags` select (oracle.c: orc_hnsw_build_ex)."""
    base = _f32(base)
    n, dim = base.shape
    levels = np.zeros(n, np.int32)
    l0 = np.full((n, 2 * M), -1, np.int32)
    up_off = np.zeros(n + 1, np.int64)
    cap = (n * 3 + 64) * M
    up = np.full(cap, -1, np.int32)
    entry = C.c_int32(-1)
    maxl = C.c_int32(0)
    cnt = np.zeros(4, np.int64)
    rc = lib().orc_hnsw_build_ex(_p(base), n, dim, int(metric), int(mode), M, ef_construction, int(seed), int(flags),
                                 _p(levels), _p(l0), _p(up_off), _p(up), cap, C.byref(entry), C.byref(maxl), _p(cnt))
    if rc != 0:
        raise RuntimeError("oracle hnsw_build: upper-level capacity exceeded")
    tot = int(up_off[n])
    g = Graph(levels, l0, up_off, up[: tot * M].copy(), M, entry.value, maxl.value)
    return (g, cnt) if want_counters else g


def hnsw_build(base, metric=COSINE, M=16, ef_construction=200, seed=42, farthest_quirk=False, mode=MODE_F64):
    """src/hnsw/ultra_fast.clj:216-344 (sequential insertion; see oracle.c for the stated deviations)."""
    base = _f32(base)
    n, dim = base.shape
    levels = np.zeros(n, np.int32)
    l0 = np.full((n, 2 * M), -1, np.int32)
    up_off = np.zeros(n + 1, np.int64)
    cap = (n * 3 + 64) * M
    up = np.full(cap, -1, np.int32)
    entry = C.c_int32(-1)
    maxl = C.c_int32(0)
    rc = lib().orc_hnsw_build(_p(base), n, dim, int(metric), int(mode), M, ef_construction, int(seed), int(farthest_quirk),
                              _p(levels), _p(l0), _p(up_off), _p(up), cap, C.byref(entry), C.byref(maxl))
    if rc != 0:
        raise RuntimeError("oracle hnsw_build: upper-level capacity exceeded")
    tot = int(up_off[n])
    return Graph(levels, l0, up_off, up[: tot * M].copy(), M, entry.value, maxl.value)


def hnsw_search(base, graph, Q, k, ef=None, metric=COSINE, mode=MODE_F64, nthreads=1, want_lat=False):
    """src/hnsw/ultra_fast.clj:346-374 over a batch (src/hnsw/helper/parallel_search.clj:15-49).

    Returns ids (nq,k) int32 (-1 padded), dists (nq,k) f64 (+inf padded), stats (nq,2) [evals, hops],
    wall_ms[, per-query latency ms]."""
    base, Q = _f32(base), _f32(Q)
    if Q.ndim == 1:
        Q = Q[None, :]
    n, dim = base.shape
    nq = Q.shape[0]
    ef = max(k, 50) if ef is None else ef  # ultra_fast.clj:355
    nr = norms(base, mode) if (mode != MODE_F64 and metric == COSINE) else None
    ids = np.full((nq, k), -1, np.int32)
    ds = np.full((nq, k), np.inf, np.float64)
    stats = np.zeros((nq, 2), np.int64)
    lat = np.zeros(nq, np.float64)
    g = graph
    ms = lib().orc_hnsw_search_batch(_p(base), n, dim, int(metric), mode, _p(nr), _p(g.levels), _p(g.l0_adj), g.M0,
                                     _p(g.up_off), _p(g.up_adj), g.M, g.entry, g.max_level, _p(Q), nq, k, ef,
                                     nthreads, _p(ids), _p(ds), _p(stats), _p(lat))
    if want_lat:
        return ids, ds, stats, ms, lat
    return ids, ds, stats, ms


# ---- ground truth ------------------------------------------------------------------------------
class LiteralGraph:
    """src/hnsw/graph.clj:239-295 `insert` restated LITERALLY (unbounded neighbour sets: the new node is linked to every node
    its second search returns and pruned only when a later insert prunes it as somebody's neighbour; ef-construction on
    every layer; the walk starts at min(level, entry-level)) with graph.clj:115-161,297-320 as its search -- a CPU-only study
    of its recall / ef curve beside the bounded variant the engine builds (oracle.c section 4b)."""

    def __init__(self, base, metric=COSINE, M=16, ef_construction=200, seed=42, mode=MODE_FAST):
        self.base = _f32(base)
        n, dim = self.base.shape
        cnt = np.zeros(8, np.int64)
        L = lib()
        L.orc_lit_build.restype = C.c_void_p
        L.orc_lit_build.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p]
        L.orc_lit_search.restype = None
        L.orc_lit_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_lit_free.restype = None
        L.orc_lit_free.argtypes = [C.c_void_p]
        self._h = C.c_void_p(L.orc_lit_build(_p(self.base), n, dim, int(metric), int(mode), M, ef_construction, int(seed), _p(cnt)))
        self.counters = dict(zip(("search_evals", "selection_evals", "prunes", "max_degree_seen", "layer0_edges", "layer0_lists_never_pruned"),
                                 cnt[:6].tolist()))

    def search(self, Q, k, ef):
        Q = _f32(Q)
        ids = np.empty((len(Q), k), np.int32)
        d = np.empty((len(Q), k), np.float64)
        ev = C.c_int64(0)
        lib().orc_lit_search(self._h, _p(Q), len(Q), k, ef, _p(ids), _p(d), C.byref(ev))
        return ids, d, ev.value / max(len(Q), 1)

    def close(self):
        if self._h:
            lib().orc_lit_free(self._h)
            self._h = None


def exact_knn(base, Q, k, metric=COSINE, mode=MODE_F64, nthreads=1):
    """src/hnsw/bench.clj:72-84 compute-exact-knn over the full base."""
    base, Q = _f32(base), _f32(Q)
    if Q.ndim == 1:
        Q = Q[None, :]
    n, dim = base.shape
    nq = Q.shape[0]
    nr = norms(base, mode) if (mode != MODE_F64 and metric == COSINE) else None
    ids = np.full((nq, k), -1, np.int32)
    ds = np.full((nq, k), np.inf, np.float64)
    ms = lib().orc_exact_knn(_p(base), n, dim, int(metric), mode, _p(nr), _p(Q), nq, k, nthreads, _p(ids), _p(ds))
    return ids, ds, ms


def recall(approx_ids, exact_ids):
    """src/hnsw/bench.clj:86-92 calc-recall, averaged over queries."""
    a = np.ascontiguousarray(approx_ids, np.int32)
    e = np.ascontiguousarray(exact_ids, np.int32)
    assert a.shape == e.shape
    return lib().orc_recall(_p(a), _p(e), a.shape[0], a.shape[1])


# ---- IVF-FLAT ------------------------------------------------------------------------------------
def kmeans_assign(base, centroids, metric=COSINE):
    """src/hnsw/ann/partition/ivf_flat.clj:79-90 for every row; centroids f64."""
    base = _f32(base)
    cen = _f64(centroids)
    n, dim = base.shape
    a = np.zeros(n, np.int32)
    d = np.zeros(n, np.float64)
    lib().orc_kmeans_assign(_p(base), n, dim, int(metric), _p(cen), cen.shape[0], _p(a), _p(d))
    return a, d


def kmeans_assign_f32(base, centroids, metric=COSINE, mode=MODE_DEV):
    """ivf_flat.clj:79-90 on float32 centroids in a device arithmetic mode -> (assign, dist f32)."""
    base, cen = _f32(base), _f32(centroids)
    n, dim = base.shape
    a = np.zeros(n, np.int32)
    d = np.zeros(n, np.float32)
    lib().orc_kmeans_assign_f32(_p(base), n, dim, int(metric), mode, _p(cen), cen.shape[0], _p(a), _p(d))
    return a, d


def kmeanspp(base, nlist, metric=COSINE, seed=42):
    """src/hnsw/ann/partition/ivf_flat.clj:32-60 -> chosen row indices."""
    base = _f32(base)
    ch = np.zeros(nlist, np.int32)
    lib().orc_kmeanspp(_p(base), base.shape[0], base.shape[1], int(metric), nlist, int(seed), _p(ch))
    return ch


def ivf_build(base, nlist=24, max_iterations=10, metric=COSINE, seed=42):
    """src/hnsw/ann/partition/ivf_flat.clj:92-131 -> (centroids f64 (nlist,dim), assign int32 (n,))."""
    base = _f32(base)
    n, dim = base.shape
    cen = np.zeros((nlist, dim), np.float64)
    a = np.zeros(n, np.int32)
    lib().orc_ivf_build(_p(base), n, dim, int(metric), nlist, max_iterations, int(seed), _p(cen), _p(a))
    return cen, a


def ivf_build_dev(base, nlist=24, max_iterations=10, metric=COSINE, seed=42):
    """The IVF build in the engine's own arithmetic (see orc_ivf_build_dev) -> (chosen rows, centroids f32, assign)."""
    base = _f32(base)
    n, dim = base.shape
    chosen = np.zeros(nlist, np.int32)
    cen = np.zeros((nlist, dim), np.float32)
    a = np.zeros(n, np.int32)
    assign_mode = MODE_DEV if metric == L2 else MODE_MFMA
    lib().orc_ivf_build_dev(_p(base), n, dim, int(metric), MODE_DEV, assign_mode, nlist, max_iterations, int(seed),
                            _p(chosen), _p(cen), _p(a))
    return chosen, cen, a


def lists_from_assign(assign, nlist):
    """Inverted lists in index order (ivf_flat.clj:126-131): list_off (nlist+1) int64, list_ids (n) int32."""
    assign = np.asarray(assign)
    order = np.argsort(assign, kind="stable").astype(np.int32)
    counts = np.bincount(assign, minlength=nlist)
    off = np.zeros(nlist + 1, np.int64)
    off[1:] = np.cumsum(counts)
    return off, order


def ivf_search(base, centroids, list_off, list_ids, Q, k, nprobe, metric=COSINE, mode=MODE_F64, scan_mode=None,
               base_norms=None):
    """src/hnsw/ann/partition/ivf_flat.clj:236-294 with explicit nprobe; centroids float32 (engine storage).
    base_norms: norms(base, mode) of an earlier call (a million-row base takes seconds per call otherwise)."""
    base, Q = _f32(base), _f32(Q)
    if Q.ndim == 1:
        Q = Q[None, :]
    cen = _f32(centroids)
    n, dim = base.shape
    nq = Q.shape[0]
    nlist = cen.shape[0]
    nprobe = min(nprobe, nlist)
    scan_mode = mode if scan_mode is None else scan_mode
    nr = cn = None
    if mode != MODE_F64 and metric == COSINE:
        nr, cn = (norms(base, mode) if base_norms is None else _f32(base_norms)), norms(cen, mode)
    off = np.ascontiguousarray(list_off, np.int64)
    lids = np.ascontiguousarray(list_ids, np.int32)
    ids = np.full((nq, k), -1, np.int32)
    ds = np.full((nq, k), np.inf, np.float64)
    probes = np.zeros((nq, nprobe), np.int32)
    lib().orc_ivf_search(_p(base), n, dim, int(metric), mode, scan_mode, _p(nr), _p(cen), _p(cn), nlist, _p(off), _p(lids),
                         _p(Q), nq, k, nprobe, _p(ids), _p(ds), _p(probes))
    return ids, ds, probes
