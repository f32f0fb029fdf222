"""Developer tool: list-scan kernel time and end-to-end time of batched IVF searches on the bench index
(1M x 768, k-means lists, nprobe 32).  usage: [METRIC=l2] [HG_DIAG build + hnswgpu_debug_set_ablation(1, 1] python tools/ivf_batch_time.py [nq ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from hnsw_clj_amd import engine

dev = torch.device("cuda", 0)
_nqs = [int(a) for a in sys.argv[1:]] or [256, 1024, 4096]
x, Qa = bench.ivf_dataset(dev, 1_000_000, 1024, max(4096, max(_nqs)))
metric = os.environ.get("METRIC", "cosine")   # METRIC=l2: no tile path, every batch size on the GEMV scan
idx = engine.Index(x, metric, 0)
del x
idx.ivf_build(1024, 10, 42)
if os.environ.get("ABLATE"):          # -DHG_DIAG builds only (tools/build_stamps.sh): the bounds pass with parts cut out, results WRONG
    import ctypes as C
    from hnsw_clj_amd import _native
    _native.lib().hnswgpu_debug_set_ablation(C.c_int32(0), C.c_int32(int(os.environ["ABLATE"])))
    print("ablation of the bounds pass: %s (1 no epilogue, 8 the test without appends, 4 no slot counter)" % os.environ["ABLATE"], flush=True)
import numpy as np
_, _off, _ = idx.get_ivf()
_lens = np.diff(_off)
_, _, _pr = idx.ivf_search(Qa[:32].cpu().numpy(), 10, 32, want_probes=True)
print("lists: max %d, mean %.0f; candidates per query at nprobe 32: %.0f; distinct lists probed by 32 queries: %d" % (
    _lens.max(), _lens.mean(), _lens[_pr.ravel()].sum() / 32.0, len(np.unique(_pr))), flush=True)
for nq in _nqs:
    Q = Qa[:nq].contiguous()
    for _ in range(3):
        idx.ivf_search_dev(Q, 10, 32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        idx.ivf_search_dev(Q, 10, 32)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10
    idx.set_profiling(True)
    idx.get_profile(engine.PROF_IVF_SCAN, reset=True)
    for _ in range(10):
        idx.ivf_search_dev(Q, 10, 32)
    ms, cnt = idx.get_profile(engine.PROF_IVF_SCAN, reset=True)
    idx.set_profiling(False)
    print("batch %5d: scan kernel %.3f ms, search %.3f ms = %.0f QPS" % (nq, ms / cnt, wall * 1e3, nq / wall), flush=True)
