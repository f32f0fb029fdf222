"""Mirror of ``hnsw.helper.parallel-search`` (src/hnsw/helper/parallel_search.clj): the query-batch
driver.  The reference submits one Callable per query to a fresh thread pool (:15-49); here the
whole batch is ONE kernel launch, ``num_threads`` is accepted for signature compatibility."""
import time

from . import ivf_flat, ultra_fast


def _batch_fn(index):
    if isinstance(index, ultra_fast.UltraGraph):
        return ultra_fast.search_batch
    if isinstance(index, ivf_flat.IVFFlatIndex):
        return ivf_flat.search_batch
    raise TypeError("unsupported index type %r" % type(index))


def parallel_search_futures(index, queries, k, search_fn=None, num_threads=None):
    """parallel_search.clj:15-49 -> results in query order."""
    return _batch_fn(index)(index, list(queries), k)


def benchmark_parallel_search(index, queries, k, search_fn=None, num_threads=None):
    """parallel_search.clj:51-95: same metric names and QPS accounting (1000 * n / total-ms)."""
    queries = list(queries)
    fn = _batch_fn(index)
    for _ in range(10):  # warm-up (:62-63)
        fn(index, queries[:1], k)
    t0 = time.perf_counter()
    res = fn(index, queries, k)
    ms = (time.perf_counter() - t0) * 1e3
    return {"threads": num_threads, "queries": len(queries), "total-time-ms": ms, "avg-latency-ms": ms / len(queries),
            "qps": 1000.0 * len(queries) / ms, "completed": len(res)}
