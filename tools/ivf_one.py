"""Developer tool: ONE batched IVF search configuration on the bench index, repeated -- meant to run under
`rocprofv3 --kernel-trace --stats` (tools/ivf_trace.sh) for the per-kernel times of one search.
usage: [TUNE=STREAM_HOME=0,MID_WIDE=0] python tools/ivf_one.py <metric> <nq> [reps] [n] [nlist]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from hnsw_clj_amd import engine

metric = sys.argv[1] if len(sys.argv) > 1 else "cosine"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
n = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
nlist = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
for kv in filter(None, os.environ.get("TUNE", "").split(",")):     # tuning-table overrides (hnswgpu_set_tuning)
    engine.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
dev = torch.device("cuda", 0)
x, Qa = bench.ivf_dataset(dev, n, nlist, max(nq, 64))
idx = engine.Index(x, metric, 0)
del x
idx.ivf_build(nlist, 10, 42)
Q = Qa[:nq].contiguous()
for _ in range(5):
    idx.ivf_search_dev(Q, 10, 32)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    idx.ivf_search_dev(Q, 10, 32)
torch.cuda.synchronize()
print("%s nq=%d: %.3f ms per search" % (metric, nq, (time.perf_counter() - t0) / reps * 1e3), flush=True)
