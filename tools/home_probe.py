"""Developer tool: the home-list kernel (ivf_home_kernel) alone over the WHOLE half-precision copy of the bench index --
hnswgpu_ivf_home_bounds on list rows [0, n) with a handful of queries -- meant to run under rocprofv3 --kernel-trace.
usage: [TUNE=HOME_CHUNK=512,HOME_DEPTH=4] python tools/home_probe.py [nq] [reps] [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from hnsw_clj_amd import engine

for kv in filter(None, os.environ.get("TUNE", "").split(",")):
    engine.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
dev = torch.device("cuda", 0)
x, Qa = bench.ivf_dataset(dev, n, 1024, 64)
idx = engine.Index(x, "cosine", 0)
del x
idx.ivf_build(1024, 2, 42)
Q = Qa[:nq].cpu().numpy()
for _ in range(reps):
    lb, ub = idx.ivf_home_bounds(Q, 0, n)
print("ok", lb.shape, float(lb[0, :5].min()))
