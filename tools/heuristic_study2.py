#!/usr/bin/env python3
"""CPU study (oracle only): does ef-construction on every layer (graph.clj:275-278) lift the recall plateau of the
heuristic graph on MANY clusters?  usage: heuristic_study2.py <n> <dim> <clusters> <flags>"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

n, dim, ncl, flags = (int(a) for a in sys.argv[1:5])
nq = 1000
rs = np.random.RandomState(1)
cen = rs.randn(ncl, dim).astype(np.float32)
x = cen[rs.randint(0, ncl, n + nq)] + 0.3 * rs.randn(n + nq, dim).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
base, Q = np.ascontiguousarray(x[:n]), np.ascontiguousarray(x[n:])
t0 = time.time()
g = O.hnsw_build_ex(base, O.COSINE, 16, 200, 42, flags, mode=O.MODE_FAST)
tb = time.time() - t0
ti = O.exact_knn(base, Q, 10, mode=O.MODE_FAST, nthreads=8)[0]
out = []
for ef in (64, 128, 256, 512, 1024):
    ids, _, st, _ = O.hnsw_search(base, g, Q, 10, ef=ef, mode=O.MODE_FAST, nthreads=8)
    rec = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(ids, ti)])
    out.append((ef, round(float(rec), 4), int(st[:, 0].mean())))
print("n=%d dim=%d clusters=%d flags=%d build %.0fs: %s" % (n, dim, ncl, flags, tb, out), flush=True)
