#!/usr/bin/env python3
"""Developer tool: waves per query of the traversal kernel at large ef (the LDS candidate list bounds the queries a CU
holds): QPS of 4096 / 10000 queries on the clustered 31k x 768 set, heuristic graph, for HNSW_NW = auto / 1 / 2 / 4."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hnsw_clj_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)
dist = sys.argv[1] if len(sys.argv) > 1 else "clustered"
base, qh = bench.make_31k(dist, 42, bench.N31K), bench.make_31k(dist, 43, 10000)
Q = torch.from_numpy(qh).to(dev)
with engine.Index(base, "cosine", 0) as idx:
    idx.hnsw_build(16, 200, 42, heuristic=True, symmetric=True)
    truth, _ = idx.exact_knn_dev(Q, 10)
    for ef in (100, 400, 600, 800, 1600, 3200):
        row = []
        for nw in (0, 1, 2, 4):
            engine.set_tuning("HNSW_NW", nw if nw else None)
            ids, _ = idx.hnsw_search_dev(Q, 10, ef)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                idx.hnsw_search_dev(Q, 10, ef)
            torch.cuda.synchronize()
            row.append("nw %s: %7.0f" % (nw or "auto", 3 * len(qh) / (time.perf_counter() - t1)))
        print("ef %4d recall %.4f  QPS  %s" % (ef, bench.recall_at_k(ids, truth), "   ".join(row)), flush=True)
