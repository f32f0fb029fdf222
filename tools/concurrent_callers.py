"""Developer tool: the reference's multi-thread protocol (helper/parallel_search.clj:15-49 -- T threads, each issuing
single-query search-knn calls) against the synchronous C entry point, which combines concurrent callers into one
launch.  usage: python tools/concurrent_callers.py [ef]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from hnsw_clj_amd import engine

ef = int(sys.argv[1]) if len(sys.argv) > 1 else 100
base = bench.make_31k("manifold", 42, 31173)
queries = bench.make_31k("manifold", 43, 4096)
idx = engine.Index(base, "cosine", 0)
idx.hnsw_build(16, 200, 42)
want, _ = idx.hnsw_search(queries, 10, ef)
for T in (1, 5, 20, 64, 128):
    per = 4096 // T if T > 1 else 512
    got = np.full((T * per, 10), -1, np.int32)

    def work(t):
        for i in range(t * per, (t + 1) * per):
            got[i] = idx.hnsw_search(queries[i:i + 1], 10, ef)[0][0]

    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    ok = np.array_equal(got, want[:T * per])
    print("%3d threads x %4d single-query calls: %.3f s = %.0f QPS, results %s" % (T, per, dt, T * per / dt,
                                                                                   "identical to one batch" if ok else "DIFFER"), flush=True)
