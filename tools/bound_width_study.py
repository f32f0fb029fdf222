"""Developer study (CPU, oracle): what would a finer QUERY code buy the traversal's int8 rejection test?
Today both sides are int8 (kernels.hpp: code_bounds): lb = d(q', v') - r_v/|v| - (r_q/|q|)(1 + r_v/|v|) - 1e-4.  With the
query in 16-bit codes (two int8 planes, two v_dot4c per dword) r_q all but vanishes.  Prints, for the bench's clustered
31k x 768 leg at the headline's ef, the share of the evaluations that still need their f32 row under each form.

usage: EF=640 python tools/bound_width_study.py [dist] [nq]"""
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import oracle as O

dist = sys.argv[1] if len(sys.argv) > 1 else "clustered"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ef = int(os.environ.get("EF", "640"))
n = bench.N31K

base = bench.make_31k(dist, 42, n)
Q = bench.make_31k(dist, 43, nq)
g = O.hnsw_build_ex(base, O.COSINE, M=bench.M, ef_construction=bench.EFC, flags=O.BUILD_HEURISTIC, mode=O.MODE_FAST)
print("graph built: n=%d entry=%d max_level=%d" % (n, g.entry, g.max_level), flush=True)

b64 = base.astype(np.float64)
norms = np.linalg.norm(b64, axis=1)


def quant(v, levels):
    s = np.abs(v).max(axis=-1, keepdims=True) / levels
    return np.rint(v / s) * s


v8 = quant(b64, 127.0)
rv = np.linalg.norm(b64 - v8, axis=1) / norms
print("rows: r_v/|v| mean %.4f max %.4f" % (rv.mean(), rv.max()))

tot = dict(evals=0, full=0, exact=0, cur=0, q16=0, q16v=0)
for qi in range(nq):
    q = Q[qi].astype(np.float64)
    qn = np.linalg.norm(q)
    q8 = quant(q, 127.0)
    q16 = quant(q, 32767.0)
    rq8 = np.linalg.norm(q - q8) / qn
    rq16 = np.linalg.norm(q - q16) / qn

    def cosd(qq, ids, rows=b64):
        return 1.0 - (rows[ids] @ qq) / (qn * norms[ids])

    cur = g.entry
    curd = cosd(q, np.array([cur]))[0]
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
            d = cosd(q, a)
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
        d = cosd(q, a)
        tot["evals"] += len(a)
        full = len(near) >= ef
        worst0 = -near[0][0]
        if full:
            tot["full"] += len(a)
            tot["exact"] += int((d < worst0).sum())
            d88 = cosd(q8, a, v8)
            tot["cur"] += int((d88 - rv[a] - rq8 * (1 + rv[a]) - 1e-4 < worst0).sum())
            d168 = cosd(q16, a, v8)
            tot["q16"] += int((d168 - rv[a] - rq16 * (1 + rv[a]) - 1e-4 < worst0).sum())
        for dj, v in zip(d, a):
            if len(near) < ef or dj < -near[0][0]:
                heapq.heappush(cand, (dj, int(v)))
                heapq.heappush(near, (-dj, int(v)))
                if len(near) > ef:
                    heapq.heappop(near)
e = tot["evals"]
notfull = e - tot["full"]
print("ef %d, %s, %d queries: %.0f evaluations per query, %.1f %% with the list full" % (ef, dist, nq, e / nq, 100.0 * tot["full"] / e))
for k in ("exact", "cur", "q16"):
    print("  %-6s needs the f32 row for %.1f %%" % (k, 100.0 * (notfull + tot[k]) / e))
