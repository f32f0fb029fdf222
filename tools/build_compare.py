"""Developer tool: the batched device build against the oracle's sequential reference-structure build
(orc_hnsw_build, ultra_fast.clj:216-299) on clustered data: recall@10 of both graphs at equal ef, searched by the same
device kernel.  usage: python tools/build_compare.py [n] [dim] [centres]      (defaults: 31173 128 256)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hnsw_clj_amd import engine
from oracle import oracle as O

O.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 31173
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ncl = int(sys.argv[3]) if len(sys.argv) > 3 else 256
base = O.generate_dataset(n, dim, "clustered", num_clusters=ncl, noise_level=0.3).astype(np.float64)
base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
Q = O.generate_dataset(1000, dim, "clustered", num_clusters=ncl, noise_level=0.3, seed=43).astype(np.float64)
Q = (Q / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float32)
t0 = time.time()
gref = O.hnsw_build(base, O.COSINE, 16, 200, seed=42)
t_ref = time.time() - t0
with engine.Index(base, "cosine") as idx:
    ti, _ = idx.exact_knn(Q, 10)
    t0 = time.time()
    idx.hnsw_build(16, 200, 42)
    t_dev = time.time() - t0
    print("%d x %d, %d centres: device build %.2f s, oracle (sequential, reference structure) build %.1f s" % (n, dim, ncl, t_dev, t_ref), flush=True)
    rows = {}
    for ef in (50, 100, 200, 400, 800):
        ids, _ = idx.hnsw_search(Q, 10, ef)
        rows[ef] = [O.recall(ids, ti)]
    idx.set_graph(gref)
    for ef in rows:
        ids, _ = idx.hnsw_search(Q, 10, ef)
        rows[ef].append(O.recall(ids, ti))
    for ef, (a, b) in rows.items():
        print("ef %4d: recall@10 device-built %.4f   reference-structure %.4f   difference %+.4f" % (ef, a, b, a - b))
