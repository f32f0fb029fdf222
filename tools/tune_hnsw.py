"""Developer tool: time the HNSW traversal kernel on the bench workload (31,173 x 768 manifold data).
usage: HNSWGPU_TUNE=HNSW_NW=1 python tools/tune_hnsw.py [nq] [ef]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from hnsw_clj_amd import engine

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 128
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
Q = torch.from_numpy(queries[:nq]).to(dev)
truth, _ = idx.exact_knn_dev(Q[:1000], 10)
ids, _ = idx.hnsw_search_dev(Q[:1000], 10, ef)
rec = bench.recall_at_k(ids, truth)
out = (torch.empty((nq, 10), dtype=torch.int32, device=dev), torch.empty((nq, 10), dtype=torch.float32, device=dev))
for _ in range(3):
    idx.hnsw_search_dev(Q, 10, ef, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 10
for _ in range(steps):
    idx.hnsw_search_dev(Q, 10, ef, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
lat = []
for i in range(30):
    t1 = time.perf_counter()
    idx.hnsw_search_dev(Q[i:i + 1], 10, ef, out=(out[0][:1], out[1][:1]))
    torch.cuda.synchronize()
    lat.append((time.perf_counter() - t1) * 1e3)
print("NW=%s nq=%d ef=%d recall=%.4f  batch %.3f ms  QPS %.0f   single-query p50 %.3f ms" % (
    os.environ.get("HNSWGPU_HNSW_NW", "auto"), nq, ef, rec, dt * 1e3, nq / dt, sorted(lat)[len(lat) // 2]))
