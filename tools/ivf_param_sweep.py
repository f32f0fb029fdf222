"""Developer tool: the survivor stream against the plain f32 scans over k / nprobe / nlist on the bench's rows (1M x 768,
Euclidean: one arithmetic, so the results must be equal) -- a net for performance cliffs off the bench's operating point.
usage: python tools/ivf_param_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from hnsw_clj_amd import engine

dev = torch.device("cuda", 0)
for nlist in (1024, 4096, 128):
    x, Qa = bench.ivf_dataset(dev, 1_000_000, 1024, 2048)
    idx = engine.Index(x, "l2", 0)
    del x
    idx.ivf_build(nlist, 3, 42)
    for k, nprobe in [(10, 32), (1, 32), (100, 32), (256, 32), (10, 1), (10, 4), (10, 128)]:
        if nprobe > nlist:
            continue
        out = {}
        for mode in (1, 0):
            idx.set_rejection_test(mode)
            res = {}
            for nq in (1, 32, 1024):
                q = Qa[:nq].contiguous()
                for _ in range(2):
                    idx.ivf_search_dev(q, k, nprobe)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    r = idx.ivf_search_dev(q, k, nprobe)
                torch.cuda.synchronize()
                res[nq] = ((time.perf_counter() - t0) / 3 * 1e3, r[0].cpu().numpy(), r[1].cpu().numpy())
            out[mode] = res
        line = []
        for nq in (1, 32, 1024):
            a, b = out[1][nq], out[0][nq]
            same = np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
            line.append("nq %d: %.3f vs %.3f ms%s" % (nq, a[0], b[0], "" if same else " MISMATCH"))
        print("nlist %d k %d nprobe %d: stream vs plain f32: %s" % (nlist, k, nprobe, "; ".join(line)), flush=True)
    idx.close()
