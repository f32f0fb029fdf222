"""ctypes binding of libhnswgpu.so (C ABI: include/hnswgpu.h).

The library is built in-tree by ``build()`` (hipcc, gfx950 only).  There is NO CPU fallback: if
the shared object is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
SO = os.environ.get("HNSWGPU_LIBRARY") or os.path.join(PKG, "libhnswgpu.so")  # override: diagnostic builds

COSINE, L2, DOT = 0, 1, 2


class HnswGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libhnswgpu error %d: %s" % (code, msg))
        self.code = code


def build(force=False, jobs=4):
    """Compile every HIP source for gfx950 into hnsw-clj_amd/libhnswgpu.so (hipcc cross-compiles
    without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(os.path.dirname(PKG), "include", "hnswgpu.h"))
    stale = (not os.path.exists(SO)) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs)
    if force or stale:
        cmd = ["make", "-C", CSRC, "-j%d" % jobs] + (["-B"] if force else [])
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return SO


_lib = None

_SIGS = {
    # name: (argtypes)
    "hnswgpu_version": [],
    "hnswgpu_device_count": ["p"],
    "hnswgpu_create": ["p", "i64", "i32", "i32", "i32", "p"],
    "hnswgpu_create_dev": ["p", "i64", "i32", "i64", "i32", "i32", "p", "p"],
    "hnswgpu_destroy": ["p"],
    "hnswgpu_info": ["p", "p", "p", "p", "p", "p"],
    "hnswgpu_sync": ["p"],
    "hnswgpu_pair_distance": ["i32", "p", "p", "i32", "i32", "p"],
    "hnswgpu_batch_distances": ["p", "p", "p", "i32", "p"],
    "hnswgpu_norms": ["p", "p"],
    "hnswgpu_exact_knn": ["p", "p", "i32", "i32", "p", "p"],
    "hnswgpu_exact_knn_dev": ["p", "p", "i32", "i32", "p", "p", "p"],
    "hnswgpu_set_graph": ["p", "p", "p", "i32", "p", "p", "i32", "i32", "i32"],
    "hnswgpu_hnsw_build": ["p", "i32", "i32", "i64"],
    "hnswgpu_hnsw_build_ex": ["p", "i32", "i32", "i64", "i32"],
    "hnswgpu_hnsw_add": ["p", "p", "i64", "i32", "i64"],
    "hnswgpu_graph_sizes": ["p", "p", "p", "p", "p", "p"],
    "hnswgpu_get_graph": ["p", "p", "p", "p", "p"],
    "hnswgpu_hnsw_search": ["p", "p", "i32", "i32", "i32", "p", "p", "p"],
    "hnswgpu_hnsw_search_dev": ["p", "p", "i32", "i32", "i32", "p", "p", "p", "p"],
    "hnswgpu_ivf_build": ["p", "i32", "i32", "i64"],
    "hnswgpu_set_ivf": ["p", "p", "i32", "p", "p"],
    "hnswgpu_get_ivf": ["p", "p", "p", "p"],
    "hnswgpu_kmeans_assign": ["p", "p", "i32", "p", "p"],
    "hnswgpu_kmeanspp": ["p", "i32", "i64", "p"],
    "hnswgpu_list_means": ["p", "i32", "p", "p", "p"],
    "hnswgpu_ivf_search": ["p", "p", "i32", "i32", "i32", "p", "p", "p"],
    "hnswgpu_ivf_search_dev": ["p", "p", "i32", "i32", "i32", "p", "p", "p"],
    "hnswgpu_ivf_search_lists": ["p", "p", "i32", "i32", "i32", "p", "p", "p"],
    "hnswgpu_set_ivf_shard": ["p", "p", "i32", "p", "p", "p"],
    "hnswgpu_ivf_stream_state": ["p", "p"],
    "hnswgpu_ivf_set_stream_state": ["p", "i32"],
    "hnswgpu_ivf_search_shard_dev": ["p", "p", "i32", "i32", "i32", "p", "p", "p", "p"],
    "hnswgpu_merge_keyed_dev": ["i32", "p", "p", "p", "i32", "i32", "i32", "p", "p", "p"],
    "hnswgpu_list_sums": ["p", "i32", "p", "p", "p"],
    "hnswgpu_merge_topk_dev": ["i32", "p", "p", "i32", "i32", "i32", "p", "p", "p"],
    "hnswgpu_merge_lists_dev": ["i32", "p", "p", "i32", "i32", "i32", "i32", "p", "p", "p"],
    "hnswgpu_rerank": ["p", "p", "i32", "p", "i32", "i32", "p", "p"],
    "hnswgpu_rerank_dev": ["p", "p", "i32", "p", "i32", "i32", "p", "p", "p"],
    "hnswgpu_dense_distances": ["p", "p", "i32", "p"],
    "hnswgpu_dense_distances_dev": ["p", "p", "i32", "p", "p"],
    "hnswgpu_save": ["p", "p"],
    "hnswgpu_load": ["p", "i32", "p"],
    "hnswgpu_set_profiling": ["p", "i32"],
    "hnswgpu_get_profile": ["p", "i32", "p", "p", "i32"],
    "hnswgpu_rejection_bounds": ["p", "p", "p", "i32", "p"],
    "hnswgpu_distance_bounds": ["p", "p", "p", "i32", "p", "p"],
    "hnswgpu_ivf_half_bounds": ["p", "p", "p", "i32", "p", "p"],
    "hnswgpu_ivf_home_bounds": ["p", "p", "i32", "i64", "i64", "p", "p"],
    "hnswgpu_set_rejection_test": ["p", "i32"],
    "hnswgpu_get_rejection_stats": ["p", "p", "p", "i32"],
    "hnswgpu_launch_count": ["i32", "p"],
    "hnswgpu_hnsw_rejection_state": ["p", "p", "p", "p"],
    "hnswgpu_set_tuning": ["i32", "i64"],
    "hnswgpu_get_tuning": ["i32", "p", "p"],
    "hnswgpu_group_create": ["p", "i32", "i32", "i32", "p"],
    "hnswgpu_group_destroy": ["p"],
    "hnswgpu_group_info": ["p", "p", "p", "p", "p"],
    "hnswgpu_group_set_ivf": ["p", "p", "i64", "p", "i32", "p", "p"],
    "hnswgpu_group_ivf_search": ["p", "p", "i32", "i32", "i32", "p", "p"],
    "hnswgpu_group_hnsw_build": ["p", "p", "i64", "i32", "i32", "i64"],
    "hnswgpu_group_hnsw_search": ["p", "p", "i32", "i32", "i32", "p", "p"],
}
# keys of hnswgpu_set_tuning (include/hnswgpu.h: HNSWGPU_TUNE_*), in the header's order
TUNE_KEYS = ['TILE_PAIRS', 'PREFILTER', 'IVF_HALF', 'IVF_CALIBRATE', 'BUILD_THREADS', 'PREFETCH', 'SEED_BOUNDS', 'TILE_WGS', 'TILE_PERSIST', 'STREAM_BUCKET', 'STREAM_WGS', 'FINISH_ORDER', 'STREAM_CAP', 'STREAM_MID', 'STREAM_NARROW', 'FINISH_ADAPT', 'FINISH_BISECT', 'FINISH_SLICES', 'FINISH_SPAN', 'STREAM_HEAVY', 'STREAM_HEAVY_MEAN', 'STREAM_HEAVY_MIN', 'MID_SLICES', 'MID_COMPACT', 'IVF_CODES', 'SCAN_ORDER', 'IVF_FUSED', 'IVF_GROUP', 'STREAM_ROUTE', 'STREAM_GROUP', 'ROUTE_GROUP', 'MID_WIDE', 'MERGE_W', 'SCAN_BLOCKS', 'ROUTE_WGS', 'TILE', 'SELECT_W', 'HNSW_NW', 'VIS_GLOBAL', 'PF_HINTS', 'PF_EVAL', 'ZEROCOPY', 'BUILD_TIMING', 'BUILD_BATCH', 'STREAM_HOME', 'HOME_CHUNK', 'HOME_DEPTH', 'HOME_STRAYS', 'ROUTE_MFMA', 'STREAM_WIDE2', 'SOLO', 'SOLO_CHASE', 'SOLO_SLOTS', 'HNSW_CALIBRATE', 'HNSW_CALIBRATE_PCT', 'HNSW_WAVE', 'FINISH_DIRECT', 'WORKLIST_FOLD', 'SEED_HALF', 'BUILD_KEEP_ROWS', 'QUERY_WAVES']
TUNE_DEFAULT = -(1 << 63)


def set_tuning(name, value=None):
    """hnswgpu_set_tuning by name (lower or upper case); value None restores the default.  Process-wide."""
    key = TUNE_KEYS.index(name.upper())
    check(lib().hnswgpu_set_tuning(key, TUNE_DEFAULT if value is None else int(value)))


LAUNCH_COUNTERS = ["bounds_two_column_blocks", "hnsw_solo", "hnsw_helpers", "hnsw_rejection", "hnsw_plain", "hnsw_wave", "route_tail_waves"]


def debug_counter(name):
    """hnswgpu_launch_count by name: launches that took the named kernel variant since the library was loaded."""
    v = C.c_int64(0)
    check(lib().hnswgpu_launch_count(LAUNCH_COUNTERS.index(name), C.byref(v)))
    return v.value


def get_tuning(name):
    """The value set for a key, or None while it is at its default."""
    v, is_set = C.c_int64(0), C.c_int32(0)
    check(lib().hnswgpu_get_tuning(TUNE_KEYS.index(name.upper()), C.byref(v), C.byref(is_set)))
    return v.value if is_set.value else None


_T = {"p": C.c_void_p, "i32": C.c_int32, "i64": C.c_int64}

EXPORTS = sorted(list(_SIGS) + ["hnswgpu_last_error", "hnswgpu_group_member"])


def lib():
    """Load libhnswgpu.so.  torch (when installed) is imported FIRST so that this library binds to
    the same HIP runtime (libamdhip64.so.7) as torch in every process."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise ImportError(
            "%s is missing: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'`). There is no CPU fallback." % SO)
    try:
        import torch  # noqa: F401  (pins the HIP runtime; torch is plumbing, not the product)
    except ImportError:
        pass
    L = C.CDLL(SO)
    for name, args in _SIGS.items():
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = [_T[a] for a in args]
    L.hnswgpu_last_error.restype = C.c_char_p
    L.hnswgpu_last_error.argtypes = []
    L.hnswgpu_group_member.restype = C.c_void_p
    L.hnswgpu_group_member.argtypes = [C.c_void_p, C.c_int32]
    _lib = L
    # developer convenience of this Python wrapper (the library itself reads six documented names, once):
    # HNSWGPU_TUNE="KEY=value,KEY=value" applies hnswgpu_set_tuning at load (tools/*.py A/B runs)
    for kv in os.environ.get("HNSWGPU_TUNE", "").split(","):
        if "=" in kv:
            k, v = kv.split("=", 1)
            set_tuning(k.strip(), int(v))
    return L


def check(rc):
    if rc != 0:
        raise HnswGpuError(rc, lib().hnswgpu_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int32(0)
    check(lib().hnswgpu_device_count(C.byref(n)))
    return n.value
