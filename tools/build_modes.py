#!/usr/bin/env python3
"""Developer tool: the device HNSW build options (closest-m / heuristic / heuristic+symmetric) on the 31,173 x 768 sets of
BASELINE.md section 3: build time, mean layer-0 degree, recall@10 and QPS over an ef sweep (4096 held-out queries).
usage: python tools/build_modes.py [dist ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hnsw_clj_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)
dists = sys.argv[1:] or ["clustered", "clustered_in", "gaussian", "uniform01", "manifold"]
nq = 4096
for dist in dists:
    if dist == "clustered_in":
        allx = bench.make_31k("clustered", 42, bench.N31K + nq)
        base, qh = allx[:bench.N31K], allx[bench.N31K:]
    else:
        base, qh = bench.make_31k(dist, 42, bench.N31K), bench.make_31k(dist, 43, nq)
    Q = torch.from_numpy(qh).to(dev)
    with engine.Index(base, "cosine", 0) as idx:
        truth, _ = idx.exact_knn_dev(Q, 10)
        for name, kw in (("closest", {}), ("heuristic", dict(heuristic=True)), ("heur+sym", dict(heuristic=True, symmetric=True)),
                         ("heur+ext", dict(heuristic=True, extend=True))):
            t0 = time.time()
            idx.hnsw_build(16, 200, 42, **kw)
            tb = time.time() - t0
            g = idx.get_graph()
            deg = (g.l0_adj.reshape(len(base), -1) >= 0).sum(1).mean()
            pts = []
            for ef in (50, 100, 200, 400, 800, 1600, 3200):
                ids, _ = idx.hnsw_search_dev(Q, 10, ef)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    idx.hnsw_search_dev(Q, 10, ef)
                torch.cuda.synchronize()
                qps = 3 * nq / (time.perf_counter() - t1)
                r = bench.recall_at_k(ids, truth)
                pts.append((ef, round(r, 4), int(qps)))
                if r >= 0.98:
                    break
            print("%-12s %-9s build %.2fs deg0 %.1f  (ef, recall, qps): %s" % (dist, name, tb, deg, pts), flush=True)
