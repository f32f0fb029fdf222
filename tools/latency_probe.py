"""Developer tool: single-query latency of the synchronous host entry points against the kernel time.
usage: [HNSWGPU_TUNE=ZEROCOPY=0] python tools/latency_probe.py [ef]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from hnsw_clj_amd import engine

ef = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cache = "/tmp/tune_31k.npz"
if os.path.exists(cache):
    z = np.load(cache)
    base, queries = z["base"], z["q"]
else:
    base = bench.make_31k("manifold", 42, 31173)
    queries = bench.make_31k("manifold", 43, 10000)
    np.savez(cache, base=base, q=queries)
dev = torch.device("cuda", 0)
idx = engine.Index(base, "cosine", 0)
idx.hnsw_build(16, 200, 42)
Q = torch.from_numpy(queries[:256]).to(dev)


def pct(a):
    a = sorted(a)
    return "p50 %.1f  min %.1f  p95 %.1f us" % (a[len(a) // 2], a[0], a[int(len(a) * 0.95)])


for nq in [int(x) for x in os.environ.get("PROBE_NQ", "1,20,200").split(",")]:
    lat = []
    for i in range(80):
        t = time.perf_counter()
        idx.hnsw_search(queries[i:i + nq], 10, ef)
        lat.append((time.perf_counter() - t) * 1e6)
    print("host entry, %3d queries per call: %s" % (nq, pct(lat[10:])))
    idx.set_profiling(True)
    idx.get_profile(engine.PROF_HNSW, reset=True)
    for i in range(40):
        idx.hnsw_search(queries[i:i + nq], 10, ef)
    ms, cnt = idx.get_profile(engine.PROF_HNSW, reset=True)
    idx.set_profiling(False)
    print("   traversal kernel (hipEvents): %.1f us avg over %d launches" % (ms / max(cnt, 1) * 1e3, cnt))
    lat = []
    o = (torch.empty((nq, 10), dtype=torch.int32, device=dev), torch.empty((nq, 10), dtype=torch.float32, device=dev))
    for i in range(80):
        t = time.perf_counter()
        idx.hnsw_search_dev(Q[i:i + nq], 10, ef, out=o)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t) * 1e6)
    print("   _dev entry + torch sync: %s" % pct(lat[10:]))
