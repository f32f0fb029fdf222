"""Developer tool: in-process, interleaved sweep of the scan kernel's workgroup-count target on k-means lists."""
import os
import sys
import time

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
qw = torch.randint(0, nlist, (1024,), generator=g, device=dev)
Qa = centers[qw] + 0.3 * torch.randn(1024, D, generator=g, device=dev)
Qa /= Qa.norm(dim=1, keepdim=True)
idx = engine.Index(x, "cosine", 0)
if os.environ.get("BALANCED"):
    a, _ = idx.kmeans_assign(centers.cpu().numpy())
    order = np.argsort(a, kind="stable").astype(np.int32)
    off = np.zeros(nlist + 1, np.int64)
    off[1:] = np.cumsum(np.bincount(a, minlength=nlist))
    idx.set_ivf(centers.cpu().numpy(), off, order)
else:
    idx.ivf_build(nlist, 10, 42)
idx.set_profiling(True)
for nq in (8, 16, 24, 32):
    Q = Qa[:nq].contiguous()
    res = {}
    for rep in range(4):
        for blocks in (1024, 2048, 3072, 4096, 6144, 8192, 16384, 32768, 65536):
            engine.set_tuning("SCAN_BLOCKS", blocks)
            idx.ivf_search_dev(Q, K, nprobe)
            idx.get_profile(engine.PROF_IVF_SCAN, reset=True)
            for _ in range(10):
                idx.ivf_search_dev(Q, K, nprobe)
            ms, cnt = idx.get_profile(engine.PROF_IVF_SCAN, reset=True)
            res.setdefault(blocks, []).append(ms / cnt)
    print("nq=%d " % nq + "  ".join("%d: %.3f" % (b, min(v)) for b, v in res.items()), flush=True)
