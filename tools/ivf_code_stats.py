"""Developer tool: how many candidates of a batched IVF search survive the int8 bounds pass (code_kernels.hpp) on the
bench index (1M x 768, k-means lists, nprobe 32).  usage: python tools/ivf_code_stats.py [nq ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from hnsw_clj_amd import engine

dev = torch.device("cuda", 0)
x, Qa = bench.ivf_dataset(dev, 1_000_000, 1024, 4096)
idx = engine.Index(x, os.environ.get("METRIC", "cosine"), 0)
del x
idx.ivf_build(1024, 10, 42)
idx.set_profiling(True)
for nq in [int(a) for a in sys.argv[1:]] or [32, 1024]:
    Q = Qa[:nq].contiguous()
    idx.ivf_search_dev(Q, 10, 32)
    torch.cuda.synchronize()
    idx.rejection_stats(reset=True)
    idx.ivf_search_dev(Q, 10, 32)
    torch.cuda.synchronize()
    surv, cand = idx.rejection_stats(reset=True)
    print("batch %d: %d candidates, %d survivors (%.2f %%, %.0f per query)" % (nq, cand, surv, 100.0 * surv / max(cand, 1), surv / nq), flush=True)
