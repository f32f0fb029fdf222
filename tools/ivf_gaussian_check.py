"""Developer tool: the survivor stream on rows WITHOUT cluster structure (i.i.d. gaussian, 1M x 768), where distances
concentrate and int8 bounds separate little -- against the plain f32 scans.  usage: python tools/ivf_gaussian_check.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hnsw_clj_amd import engine

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(3)
x = torch.randn(1_000_000, 768, generator=g, device=dev)
Q = torch.randn(1024, 768, generator=g, device=dev)
for metric in ("l2", "cosine"):
    idx = engine.Index(x, metric, 0)
    idx.ivf_build(1024, 3, 42)
    idx.set_profiling(True)
    out = {}
    for mode in (1, 0):
        idx.set_rejection_test(mode)
        res = {}
        for nq in (1, 32, 256, 1024):
            q = Q[:nq].contiguous()
            for _ in range(2):
                idx.ivf_search_dev(q, 10, 32)
            torch.cuda.synchronize()
            idx.rejection_stats(reset=True)
            t0 = time.perf_counter()
            for _ in range(3):
                r = idx.ivf_search_dev(q, 10, 32)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 3 * 1e3
            surv, cand = idx.rejection_stats(reset=True)
            res[nq] = (wall, r[0].cpu().numpy(), r[1].cpu().numpy(), surv / 3 / nq, cand / 3 / nq)
        out[mode] = res
    for nq in (1, 32, 256, 1024):
        a, b = out[1][nq], out[0][nq]
        same = np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
        print("%s nq %d: stream %.3f ms (f32 rows/query %.0f of %.0f candidates) vs plain f32 %.3f ms%s" % (
            metric, nq, a[0], a[3], a[4], b[0], "" if (same or metric != "l2") else " MISMATCH"), flush=True)
    idx.close()
