"""Developer tool: single-query (and small-batch) IVF search latency on the bench index (1M x 768, nlist 1024, nprobe 32).
usage: [HNSWGPU_TUNE=IVF_FUSED=0] python tools/ivf_latency.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from hnsw_clj_amd import engine

dev = torch.device("cuda", 0)
x, Qa = bench.ivf_dataset(dev, 1_000_000, 1024, 1024)
idx = engine.Index(x, "cosine", 0)
del x
idx.ivf_build(1024, 10, 42)
qh = Qa.cpu().numpy()


def pct(a):
    a = sorted(a)
    return "p50 %.1f  min %.1f  p95 %.1f us" % (a[len(a) // 2], a[0], a[int(len(a) * 0.95)])


for nq in (1, 4, 32):
    o = (torch.empty((nq, 10), dtype=torch.int32, device=dev), torch.empty((nq, 10), dtype=torch.float32, device=dev))
    lat = []
    for i in range(100):
        t = time.perf_counter()
        idx.ivf_search_dev(Qa[i:i + nq], 10, 32, out=o)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t) * 1e6)
    print("nq=%2d  _dev entry + sync: %s" % (nq, pct(lat[20:])))
    lat = []
    for i in range(100):
        t = time.perf_counter()
        idx.ivf_search(qh[i:i + nq], 10, 32)
        lat.append((time.perf_counter() - t) * 1e6)
    print("        host entry      : %s" % pct(lat[20:]))
    # back-to-back launches: device time per search without the host round trip
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(200):
        idx.ivf_search_dev(Qa[i % 64:i % 64 + nq], 10, 32, out=o)
    torch.cuda.synchronize()
    print("        200 searches back to back: %.1f us each" % ((time.perf_counter() - t) / 200 * 1e6))
