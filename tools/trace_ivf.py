"""Developer tool: a few batch-1024 IVF searches, to be run under rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hnsw_clj_amd import engine

dev = torch.device("cuda", 0)
n, nlist, nprobe, D, K = 1_000_000, 1024, 32, 768, 10
g = torch.Generator(device=dev)
g.manual_seed(42)
centers = torch.randn(nlist, D, generator=g, device=dev)
which = torch.randint(0, nlist, (n,), generator=g, device=dev)
x = centers[which] + 0.3 * torch.randn(n, D, generator=g, device=dev)
x /= x.norm(dim=1, keepdim=True)
g.manual_seed(43)
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
qw = torch.randint(0, nlist, (nq,), generator=g, device=dev)
Q = centers[qw] + 0.3 * torch.randn(nq, D, generator=g, device=dev)
Q /= Q.norm(dim=1, keepdim=True)
idx = engine.Index(x, "cosine", 0)
a, _ = idx.kmeans_assign(centers.cpu().numpy())
order = np.argsort(a, kind="stable").astype(np.int32)
off = np.zeros(nlist + 1, np.int64)
off[1:] = np.cumsum(np.bincount(a, minlength=nlist))
idx.set_ivf(centers.cpu().numpy(), off, order)
for _ in range(10):
    idx.ivf_search_dev(Q, K, nprobe)
torch.cuda.synchronize()
