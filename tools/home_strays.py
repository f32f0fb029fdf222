"""Developer tool: where do the rows of a query's own cluster sit?  On the bench index (1M x 768, 1024 generated clusters, 1024
k-means lists) prints, over 4096 queries, the share of the query's cluster that is in its nearest list, its second nearest,
and beyond -- what the home-list pass of large batches covers and what the bounds pass still has to append."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from hnsw_clj_amd import engine

dev = torch.device("cuda", 0)
n, nlist, nq = 1_000_000, 1024, 4096
g = torch.Generator(device=dev)
g.manual_seed(42)
centers = torch.randn(nlist, bench.DIM, generator=g, device=dev)
which = torch.randint(0, nlist, (n,), generator=g, device=dev).cpu().numpy()
g.manual_seed(43)
qw = torch.randint(0, nlist, (nq,), generator=g, device=dev).cpu().numpy()
x, Qa = bench.ivf_dataset(dev, n, nlist, nq)
idx = engine.Index(x, "cosine", 0)
del x
idx.ivf_build(nlist, 10, 42)
_, off, lids = idx.get_ivf()
row_list = np.empty(n, np.int32)
for l in range(nlist):
    row_list[lids[off[l]:off[l + 1]]] = l
_, _, pr = idx.ivf_search(Qa.cpu().numpy(), 10, 32, want_probes=True)
comp = np.zeros((nlist, nlist), np.int32)          # [cluster][list]
np.add.at(comp, (which, row_list), 1)
tot = comp.sum(1)
share = np.zeros((nq, 4))
for i in range(nq):
    c = comp[qw[i]]
    s0, s1 = c[pr[i, 0]], c[pr[i, 1]]
    rest32 = c[pr[i, 2:]].sum()
    share[i] = (s0, s1, rest32, tot[qw[i]] - s0 - s1 - rest32)
print("rows of the query's own cluster: nearest list %.1f, second %.1f, probes 3..32 %.1f, not probed %.1f (means over %d queries)" % (*share.mean(0), nq))
out = share[:, 1] + share[:, 2]
print("queries by rows of their cluster OUTSIDE the nearest list (but probed): 0: %d, 1-8: %d, 9-32: %d, 33-128: %d, 129+: %d" % (
    (out == 0).sum(), ((out > 0) & (out <= 8)).sum(), ((out > 8) & (out <= 32)).sum(), ((out > 32) & (out <= 128)).sum(), (out > 128).sum()))
print("... of those outside: in the second list %.1f%%" % (100.0 * share[:, 1].sum() / max(out.sum(), 1)))
lens = np.diff(off)
print("list lengths: min %d max %d; lists holding > 1200 rows: %d, < 600: %d" % (lens.min(), lens.max(), (lens > 1200).sum(), (lens < 600).sum()))
