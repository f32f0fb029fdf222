"""Mirror of ``hnsw.helper.parallel-search`` (src/hnsw/helper/parallel_search.clj): the query-batch driver.

Two ways to serve a batch, both equal to ``[search_fn(index, q, k) for q in queries]``:

* ``search_fn`` is None or one of the engine's own ``search_knn`` functions: the whole batch is ONE kernel launch
  (``search-batch*``, api/protocol.clj:58-67) -- ``num_threads`` is irrelevant.
* any other ``search_fn`` (a user's wrapper, a mode-binding lambda ...): the reference's protocol as written --
  a fresh pool of ``num_threads`` threads, one task per query, results in query order (:15-49).  The engine's
  synchronous entry points combine the concurrent calls into shared launches (ctypes releases the GIL), so the
  pattern a drop-in user already has keeps scaling with the thread count."""
import time
from concurrent.futures import ThreadPoolExecutor

from . import ivf_flat, ultra_fast


def _batch_fn(index, search_fn=None):
    """The one-launch batch entry for (index, search_fn), or None if search_fn has to be called per query."""
    if isinstance(index, ultra_fast.UltraGraph) and search_fn in (None, ultra_fast.search_knn):
        return ultra_fast.search_batch
    if isinstance(index, ivf_flat.IVFFlatIndex) and search_fn in (None, ivf_flat.search_knn):
        return ivf_flat.search_batch
    if search_fn is None:
        raise TypeError("unsupported index type %r" % type(index))
    return None


def parallel_search_futures(index, queries, k, search_fn=None, num_threads=None):
    """parallel_search.clj:15-49 -> results in query order."""
    queries = list(queries)
    fn = _batch_fn(index, search_fn)
    if fn is not None:
        return fn(index, queries, k)
    with ThreadPoolExecutor(max_workers=max(1, int(num_threads or 1))) as pool:     # a fresh pool per call (:31)
        futures = [pool.submit(search_fn, index, q, k) for q in queries]            # one task per query (:35-40)
        return [f.result() for f in futures]                                        # collected in order (:43)


def benchmark_parallel_search(index, queries, k, search_fn=None, num_threads=None):
    """parallel_search.clj:51-95: same metric names and QPS accounting (1000 * n / total-ms)."""
    queries = list(queries)
    for _ in range(10):  # warm-up (:62-63)
        parallel_search_futures(index, queries[:1], k, search_fn, num_threads)
    t0 = time.perf_counter()
    res = parallel_search_futures(index, queries, k, search_fn, num_threads)
    ms = (time.perf_counter() - t0) * 1e3
    return {"threads": num_threads, "queries": len(queries), "total-time-ms": ms, "avg-latency-ms": ms / len(queries),
            "qps": 1000.0 * len(queries) / ms, "completed": len(res)}
