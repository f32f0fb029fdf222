"""Developer tool: survivors of the bounds pass (candidates that reach the exact f32 pass) and the search time on the
bench index.  usage: [HG_DIAG build + hnswgpu_debug_set_ablation(0, 2] python tools/ivf_survivors.py <metric> [nq ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from hnsw_clj_amd import engine

metric = sys.argv[1] if len(sys.argv) > 1 else "cosine"
dev = torch.device("cuda", 0)
x, Qa = bench.ivf_dataset(dev, 1_000_000, 1024, 4096)
idx = engine.Index(x, metric, 0)
del x
idx.ivf_build(1024, 10, 42)
for nq in [int(a) for a in sys.argv[2:]] or [32, 256, 1024, 4096]:
    Q = Qa[:nq].contiguous()
    for _ in range(3):
        idx.ivf_search_dev(Q, 10, 32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        idx.ivf_search_dev(Q, 10, 32)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10
    idx.set_profiling(True)
    idx.rejection_stats(reset=True)
    idx.ivf_search_dev(Q, 10, 32)
    torch.cuda.synchronize()
    surv, cand = idx.rejection_stats(reset=True)
    idx.set_profiling(False)
    print("%s batch %5d: %.3f ms; candidates/query %.0f, survivors/query %.1f (%.3f %%)" % (
        metric, nq, wall * 1e3, cand / nq, surv / nq, 100.0 * surv / max(cand, 1)), flush=True)
