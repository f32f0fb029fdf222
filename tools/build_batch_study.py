#!/usr/bin/env python3
"""Developer tool: does the insertion batch size of the device HNSW build cost recall on many well-separated clusters?
200k x 256, 1024 clusters (the CPU study's set: tools/heuristic_study2.py reaches recall 1.0 at ef 128 sequentially)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hnsw_clj_amd import engine  # noqa: E402

n, dim, ncl, nq = 200000, 256, 1024, 1000
rs = np.random.RandomState(1)
cen = rs.randn(ncl, dim).astype(np.float32)
x = cen[rs.randint(0, ncl, n + nq)] + 0.3 * rs.randn(n + nq, dim).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
base, qh = np.ascontiguousarray(x[:n]), np.ascontiguousarray(x[n:])
dev = torch.device("cuda", 0)
Q = torch.from_numpy(qh).to(dev)
with engine.Index(base, "cosine", 0) as idx:
    truth, _ = idx.exact_knn_dev(Q, 10)
    for mb in (16384, 2048, 256, 32):
        engine.set_tuning("BUILD_BATCH", mb)
        t0 = time.time()
        idx.hnsw_build(16, 200, 42, heuristic=True, symmetric=True)
        tb = time.time() - t0
        pts = []
        for ef in (64, 128, 256, 512, 1024):
            ids, _ = idx.hnsw_search_dev(Q, 10, ef)
            pts.append((ef, round(bench.recall_at_k(ids, truth), 4)))
        print("max batch %5d: build %.2fs  %s" % (mb, tb, pts), flush=True)
