"""Developer tool: how many of the neighbours a small HNSW launch evaluates the traversal still has to gather itself when
the helper workgroups evaluate the hinted nodes' neighbours and publish the distances (kernels.hpp: pf_res).
usage: [HNSWGPU_TUNE=PF_HINTS=n] [HNSWGPU_PREFETCH=groups] python tools/helper_eval_stats.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hnsw_clj_amd import engine
base = bench.make_31k("manifold", 42, 31173)
queries = bench.make_31k("manifold", 43, 64)
idx = engine.Index(base, "cosine", 0)
idx.hnsw_build(16, 200, 42)
idx.set_rejection_test(0)   # no int8 test: every neighbour not gathered locally was published by a helper
idx.set_profiling(True)
for nq in (1, 8, 16):
    tot = [0, 0]
    for i in range(20):
        idx.rejection_stats(reset=True)
        idx.hnsw_search(queries[i:i + nq], 10, 100)
        a, b = idx.rejection_stats(reset=True)
        tot[0] += a; tot[1] += b
    print("nq %d: gathered by the traversal itself: %.0f of %.0f neighbours per query (%.1f %%)" % (nq, tot[0] / 20 / nq, tot[1] / 20 / nq, 100.0 * tot[0] / tot[1]))
