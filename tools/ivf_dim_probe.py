"""Developer tool: one IVF configuration at another row length than the bench's (clustered rows, NOT normalised), repeated --
for tools/ivf_trace-style profiling and for the device counters.  usage: python tools/ivf_dim_probe.py <metric> <dim> <n> <nlist> <nq> [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hnsw_clj_amd import engine

metric, dim, n, nlist, nq = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
cen = torch.randn(nlist, dim, generator=g, device=dev)
x = cen[torch.randint(0, nlist, (n,), generator=g, device=dev)] + 0.3 * torch.randn(n, dim, generator=g, device=dev)
Q = cen[torch.randint(0, nlist, (nq,), generator=g, device=dev)] + 0.3 * torch.randn(nq, dim, generator=g, device=dev)
idx = engine.Index(x, metric, 0)
idx.set_rejection_test(2)
idx.ivf_build(nlist, 3, 42)
for _ in range(3):
    idx.ivf_search_dev(Q, 10, 16)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    idx.ivf_search_dev(Q, 10, 16)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps
idx.set_profiling(True)
idx.rejection_stats(reset=True)
idx.ivf_search_dev(Q, 10, 16)
torch.cuda.synchronize()
surv, cand = idx.rejection_stats(reset=True)
print("%s dim %d n %d nlist %d nq %d: %.3f ms; candidates/query %.0f, f32 rows/query %.1f" % (
    metric, dim, n, nlist, nq, wall * 1e3, cand / nq, surv / nq), flush=True)
