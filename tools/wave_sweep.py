#!/usr/bin/env python3
"""Developer tool: the one-wave-per-query kernel with the admission buffer (wave_kernels.hpp) against the single-workgroup
kernel (HNSW_WAVE 0 / 1 = the default rule / 2 = whenever it can) over ef and batch size; 31k x 768, heuristic graph.
usage: python tools/wave_sweep.py [distribution]"""
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
Qa = torch.from_numpy(qh).to(dev)
with engine.Index(base, "cosine", 0) as idx:
    idx.hnsw_build(16, 200, 42, heuristic=True)
    for nq in (256, 512, 1024, 2048, 4096, 10000):
        Q = Qa[:nq].contiguous()
        for ef in (50, 100, 200, 640, 1024, 1600, 3200):
            row = []
            for w in (0, 1, 2):
                engine.set_tuning("HNSW_WAVE", w)
                idx.hnsw_search_dev(Q, 10, ef)
                torch.cuda.synchronize()
                reps = 3 if nq * ef > 2e6 else 10
                t1 = time.perf_counter()
                for _ in range(reps):
                    idx.hnsw_search_dev(Q, 10, ef)
                torch.cuda.synchronize()
                row.append("%s %8.0f" % (("old", "rule", "wave")[w], reps * nq / (time.perf_counter() - t1)))
            print("nq %5d ef %4d  QPS  %s" % (nq, ef, "   ".join(row)), flush=True)
