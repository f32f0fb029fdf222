"""Developer study (CPU, oracle): how many of an HNSW search's distance evaluations could a compressed copy of the rows
decide?  A neighbour evaluated while the result list is full only matters if its distance is below the list's worst
(ultra_fast.clj:195-198, strict <).  With a rigorous lower bound  lb = d(q, v_hat) - err(v)  from a compressed row
v_hat (err(v) = |v - v_hat| / |v| by Cauchy-Schwarz for cosine on the query side), every evaluation with lb >= worst
needs no f32 row at all.  Prints, for the bench's 31k x 768 leg at ef 100: evaluations per query, the share that is
admitted-or-list-not-full (what an exact test lets through) and the share an int8 / bf16 lower bound lets through.

usage: python tools/prefilter_study.py [dist] [n] [nq]"""
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import oracle as O

dist = sys.argv[1] if len(sys.argv) > 1 else "manifold"
n = int(sys.argv[2]) if len(sys.argv) > 2 else bench.N31K
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ef = int(os.environ.get("EF", "100"))

x = bench.make_31k(dist, 42, n + nq)
base, Q = x[:n].copy(), x[n:].copy()
g = O.hnsw_build(base, O.COSINE, M=bench.M, ef_construction=bench.EFC, mode=O.MODE_FAST)
print("graph built: n=%d entry=%d max_level=%d" % (n, g.entry, g.max_level), flush=True)

b64 = base.astype(np.float64)
norms = np.linalg.norm(b64, axis=1)


def quant_int8(v):
    s = np.abs(v).max(axis=1, keepdims=True) / 127.0
    return np.rint(v / s) * s


def quant_bf16(v):
    u = v.astype(np.float32).view(np.uint32)
    u = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16   # round to nearest even
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def quant_int4(v):
    s = np.abs(v).max(axis=1, keepdims=True) / 7.0
    return np.rint(v / s) * s


forms = {"int8": quant_int8(b64), "bf16": quant_bf16(b64), "int4": quant_int4(b64)}
err = {k: np.linalg.norm(b64 - h, axis=1) / norms for k, h in forms.items()}
for k in forms:
    print("%s: Cauchy-Schwarz bound on the cosine-distance error: mean %.2e max %.2e" % (k, err[k].mean(), err[k].max()))


def cosd(q, qn, rows, ids):
    return 1.0 - (rows[ids] @ q) / (qn * norms[ids])


tot = dict(evals=0, full=0, exact=0, **{k: 0 for k in forms}, **{k + "_act": 0 for k in forms})
for qi in range(nq):
    q = Q[qi].astype(np.float64)
    qn = np.linalg.norm(q)
    # upper levels: greedy (ef 1)
    cur = g.entry
    curd = cosd(q, qn, b64, np.array([cur]))[0]
    for level in range(g.max_level, 0, -1):
        improved = True
        while improved:
            improved = False
            if g.levels[cur] < level:
                break
            a = g.up_adj.reshape(-1, g.M)[g.up_off[cur] + level - 1]
            a = a[a >= 0]
            if len(a) == 0:
                break
            d = cosd(q, qn, b64, a)
            j = int(np.argmin(d))
            if d[j] < curd:
                cur, curd, improved = int(a[j]), d[j], True
    vis = {cur}
    cand = [(curd, cur)]
    near = [(-curd, cur)]
    while cand:
        d0, c = heapq.heappop(cand)
        if len(near) >= ef and d0 > -near[0][0]:
            continue
        a = g.l0_adj[c]
        a = np.array([v for v in a if v >= 0 and v not in vis], np.int64)
        vis.update(a.tolist())
        if len(a) == 0:
            continue
        d = cosd(q, qn, b64, a)
        tot["evals"] += len(a)
        full = len(near) >= ef
        worst0 = -near[0][0]
        if full:
            tot["full"] += len(a)
            tot["exact"] += int((d < worst0).sum())
            for k, h in forms.items():
                da = 1.0 - (h[a] @ q) / (qn * norms[a])
                tot[k] += int((da - err[k][a] - 1e-4 < worst0).sum())
                tot[k + "_act"] += int((da - 0.1 * err[k][a] < worst0).sum())   # what a 10x tighter bound would pass
        for dj, v in zip(d, a):
            if len(near) < ef or dj < -near[0][0]:
                heapq.heappush(cand, (dj, int(v)))
                heapq.heappush(near, (-dj, int(v)))
                if len(near) > ef:
                    heapq.heappop(near)
e = tot["evals"]
print("ef %d, %s, %d queries: %.0f evaluations per query, %.1f %% of them with the list full" % (ef, dist, nq, e / nq, 100.0 * tot["full"] / e))
notfull = e - tot["full"]
print("need the f32 row (list not full, or below the worst): exact test %.1f %%" % (100.0 * (notfull + tot["exact"]) / e))
for k in forms:
    print("  %s lower bound lets through %.1f %% (a 10x tighter bound: %.1f %%)" % (
        k, 100.0 * (notfull + tot[k]) / e, 100.0 * (notfull + tot[k + "_act"]) / e))
