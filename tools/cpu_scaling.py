"""Developer tool: thread scaling of the CPU oracle's HNSW batch search on the GPU box's host."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from hnsw_clj_amd import engine
from oracle import oracle as O

base = bench.make_31k("manifold", 42, 31173)
queries = bench.make_31k("manifold", 43, 4000)
idx = engine.Index(base, "cosine", 0)
idx.hnsw_build(16, 200, 42)
g = idx.get_graph()
og = O.Graph(g.levels, g.l0_adj, g.up_off, g.up_adj, g.M, g.entry, g.max_level)
print("cpu count", os.cpu_count())
for mode, name in ((O.MODE_F64, "f64"), (O.MODE_FAST, "f32-fast")):
    for T in (1, 8, 16, 32, 64, 128, 256):
        nq = min(len(queries), 60 * T)
        _, _, _, ms = O.hnsw_search(base, og, queries[:nq], 10, ef=128, mode=mode, nthreads=T)
        qs = np.resize(queries, (max(nq, int(nq * 3000 / max(ms, 1))), queries.shape[1]))[:40000]
        _, _, _, ms = O.hnsw_search(base, og, qs, 10, ef=128, mode=mode, nthreads=T)
        print("%s threads %3d: %6d queries in %8.1f ms = %8.0f QPS" % (name, T, len(qs), ms, len(qs) / ms * 1e3), flush=True)
