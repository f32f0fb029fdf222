"""CPU-only study (VERDICT r04 "missing 3"): hnsw.graph/insert restated literally (oracle.c section 4b: unbounded own lists,
ef-construction on every layer, no descent from the top) against the bounded variant the engine and orc_hnsw_build_ex build
(graph.clj's heuristic selection into rows of 2M / M ids, a dropped edge removed from both lists): recall@10 over ef on the
31,173 x 768 sets, 256 held-out queries, ground truth by the oracle's exact scan.  Distances in the oracle's vectorised f32
mode (both builders alike).
usage: python tools/literal_insert_study.py <clustered|uniform01|gaussian> [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from oracle import oracle as O

dist = sys.argv[1] if len(sys.argv) > 1 else "clustered"
n = int(sys.argv[2]) if len(sys.argv) > 2 else bench.N31K
base = bench.make_31k(dist, 42, n)
Q = bench.make_31k(dist, 43, 256)
truth = O.exact_knn(base, Q, 10, metric=O.COSINE, mode=O.MODE_FAST, nthreads=4)[0]
efs = (50, 100, 200, 400, 640, 800, 1600, 3200)


def recall(ids):
    return float(np.mean([len(set(ids[i].tolist()) & set(truth[i].tolist())) / 10.0 for i in range(len(Q))]))


t = time.time()
lit = O.LiteralGraph(base, O.COSINE, 16, 200, 42, O.MODE_FAST)
print("%s %d x 768: graph/insert literally: built in %.0f s; %s" % (dist, n, time.time() - t, lit.counters), flush=True)
for ef in efs:
    ids, _, ev = lit.search(Q, 10, ef)
    print("  literal   ef %4d  recall@10 %.4f  evaluations per query %.0f" % (ef, recall(ids), ev), flush=True)
lit.close()
t = time.time()
g = O.hnsw_build_ex(base, O.COSINE, M=16, ef_construction=200, seed=42, flags=2 | 16, mode=O.MODE_FAST)   # ORC_BUILD_HEURISTIC | _SYMMETRIC
print("%s %d x 768: bounded variant (heuristic + symmetric, what hnsw.gpu builds with :select :graph-clj): built in %.0f s" % (dist, n, time.time() - t), flush=True)
for ef in efs:
    ids, _, st, _ = O.hnsw_search(base, g, Q, 10, ef=ef, metric=O.COSINE, mode=O.MODE_FAST, nthreads=4)
    print("  bounded   ef %4d  recall@10 %.4f  evaluations per query %.0f" % (ef, recall(ids), st[:, 0].mean()), flush=True)
