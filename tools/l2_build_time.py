"""Developer tool: IVF build time with the Euclidean metric (k-means assignment on l2_group_kernel; HNSWGPU_TUNE=TILE=0
forces the GEMV scan).  usage: [HNSWGPU_TUNE=TILE=0] python tools/l2_build_time.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from hnsw_clj_amd import engine
dev = torch.device("cuda", 0)
x, Qa = bench.ivf_dataset(dev, 1_000_000, 1024, 64)
idx = engine.Index(x, "l2", 0)
idx.ivf_build(1024, 2, 42)
t = time.time(); idx.ivf_build(1024, 10, 42); torch.cuda.synchronize()
print("L2 IVF build 1M x 768 x 1024 lists, 10 Lloyd passes: %.2f s (HNSWGPU_TUNE=TILE=%s)" % (time.time() - t, os.environ.get("HNSWGPU_TILE", "default")))
