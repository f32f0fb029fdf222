"""Developer tool: time the IVF list scan (1M x 768, nlist 1024, nprobe 32) at several batch sizes.
usage: [HNSWGPU_TUNE=SCAN_BLOCKS=N] python tools/tune_ivf.py"""
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
# cheap lists for tuning: assign to the generating centres (same list-length distribution as k-means)
import os
if os.environ.get("REAL_KMEANS"):
    idx.ivf_build(nlist, 10, 42)
    cen_, off, order = idx.get_ivf()
    if os.environ.get("REAL_KMEANS") == "fresh":   # same lists installed into a fresh handle (fresh allocations)
        idx.close()
        idx = engine.Index(x, "cosine", 0)
        idx.set_ivf(cen_, off, order)
else:
  a, _ = idx.kmeans_assign(centers.cpu().numpy())
  if os.environ.get("MERGE_PAIRS"):   # skewed synthetic lists: clusters 2l and 2l+1 share list 2l, list 2l+1 is empty
      a = (a // 2) * 2
  order = np.argsort(a, kind="stable").astype(np.int32)
  off = np.zeros(nlist + 1, np.int64)
  off[1:] = np.cumsum(np.bincount(a, minlength=nlist))
  idx.set_ivf(centers.cpu().numpy(), off, order)
lens = np.diff(off)
print("list len mean %.0f max %d min %d" % (lens.mean(), lens.max(), lens.min()))
for nq in (1, 32, 256, 1024):
    Q = Qa[:nq].contiguous()
    if os.environ.get("SAMEQ"):          # every query identical: all pairs of a rank stream the same list at once
        Q = Qa[:1].repeat(nq, 1).contiguous()
    if os.environ.get("GROUPQ"):         # queries in groups of 4 identical ones
        Q = Qa[:max(1, nq // 4)].repeat_interleave(4, 0)[:nq].contiguous()
    _, _, probes = idx.ivf_search(Q.cpu().numpy(), K, nprobe, want_probes=True)
    alg = int(lens[probes.ravel()].sum()) * (4 * D + 4)
    for _ in range(3):
        idx.ivf_search_dev(Q, K, nprobe)
    steps = 20 if nq <= 32 else 5
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.ivf_search_dev(Q, K, nprobe)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps          # end-to-end, profiling events off
    idx.set_profiling(True)
    idx.get_profile(engine.PROF_IVF_SCAN, reset=True)
    for _ in range(steps):
        idx.ivf_search_dev(Q, K, nprobe)
    ms, cnt = idx.get_profile(engine.PROF_IVF_SCAN, reset=True)
    idx.set_profiling(False)
    print("blocks=%s nq=%4d scan %.4f ms  %.0f GB/s algorithmic   search wall %.4f ms  QPS %.0f" % (
        os.environ.get("HNSWGPU_SCAN_BLOCKS", "8192"), nq, ms / cnt, alg / (ms / cnt * 1e-3) / 1e9, wall * 1e3, nq / wall))
