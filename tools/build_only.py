"""Developer tool: ONE device HNSW build at configs[4]'s per-GPU size (1.25M x 1536 clustered-normalised rows, heuristic builder)
with the builder's own timing line -- meant to run under `rocprofv3 --kernel-trace --stats` for the per-kernel totals.
usage: python tools/build_only.py [rows] [dim] [builder]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from hnsw_clj_amd import engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
builder = sys.argv[3] if len(sys.argv) > 3 else "heuristic"
engine.set_tuning("BUILD_TIMING", 1)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
cen = torch.randn(1024, dim, generator=g, device=dev)
x = torch.empty(n, dim, device=dev)
for i in range(0, n, 250_000):
    m = min(250_000, n - i)
    y = cen[torch.randint(0, 1024, (m,), generator=g, device=dev)] + 0.3 * torch.randn(m, dim, generator=g, device=dev)
    x[i:i + m] = y / y.norm(dim=1, keepdim=True)
idx = engine.Index(x, "cosine", 0)
torch.cuda.synchronize()
graphs = []
for keep in [int(v) for v in os.environ.get("KEEP", "1").split(",")]:   # KEEP=0,1,0,1: the selection kernel's two forms on one box
    engine.set_tuning("BUILD_KEEP_ROWS", keep)
    t = time.time()
    idx.hnsw_build(16, 200, 42, **bench.BUILDERS[builder])
    print("HNSW build %d x %d (%s builder, taken rows in registers: %d): %.2f s" % (n, dim, builder, keep, time.time() - t), flush=True)
    g = idx.get_graph()
    graphs.append(g.l0_adj.copy() if hasattr(g, "l0_adj") else None)
if len(graphs) > 1 and graphs[0] is not None:
    import numpy as np
    print("layer-0 adjacency equal across the builds: %s" % all(np.array_equal(graphs[0], h) for h in graphs[1:]), flush=True)
