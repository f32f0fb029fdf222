"""Developer tool: a few batch-N IVF searches on an L2-metric index (k-means lists), to run under rocprofv3."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from hnsw_clj_amd import engine

dev = torch.device("cuda", 0)
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x, Qa = bench.ivf_dataset(dev, 1_000_000, 1024, 4096)
idx = engine.Index(x, os.environ.get("METRIC", "l2"), 0)
del x
idx.ivf_build(1024, 3, 42)
Q = Qa[:nq].contiguous()
for _ in range(4):
    idx.ivf_search_dev(Q, 10, 32)
torch.cuda.synchronize()
