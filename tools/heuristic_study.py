#!/usr/bin/env python3
"""CPU study (oracle only): recall@10 of the graph the oracle's builder makes under the graph.clj build options
(oracle.c: orc_hnsw_build_ex flags), on the S1 data sets of BASELINE.md section 3.
usage: heuristic_study.py <dist> <n> <flags> [dim]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402

dist, n, flags = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
nq = 500
if dist == "clustered_in":       # held-out rows of the SAME mixture: rows [n, n + nq) of the seed-42 stream
    allx = bench.make_31k("clustered", 42, n + nq)
    base, Q = allx[:n], allx[n:]
else:
    base = bench.make_31k(dist, 42, n)
    Q = bench.make_31k(dist, 43, nq)
t0 = time.time()
g, cnt = O.hnsw_build_ex(base, O.COSINE, 16, 200, 42, flags, mode=O.MODE_FAST, want_counters=True)
tb = time.time() - t0
deg0 = (g.l0_adj >= 0).sum(1)
ti, _, _ = O.exact_knn(base, Q, 10, mode=O.MODE_FAST, nthreads=8)[:3]
out = []
for ef in (50, 100, 200, 400, 800, 1600):
    ids, _, st, _ = O.hnsw_search(base, g, Q, 10, ef=ef, mode=O.MODE_FAST, nthreads=8)
    rec = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(ids, ti)])
    out.append((ef, round(float(rec), 4), int(st[:, 0].mean())))
    if rec >= 0.98:
        break
print("%s n=%d flags=%d build %.0fs counters %s mean deg0 %.1f  (ef, recall, evals): %s" % (dist, n, flags, tb, cnt.tolist(), deg0.mean(), out), flush=True)
