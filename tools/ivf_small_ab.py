"""Developer tool: small IVF batches on the bench index (1M x 768, nlist 1024, nprobe 32) with this round's schedule
switches on and off in ONE process (same box, same index): FINISH_DIRECT (a key per survivor to the query's last workgroup),
WORKLIST_FOLD (the bounds pass's work list inside the routing tail's launch), SEED_HALF (the first threshold from half rows).  Prints us per search, back to back, and
checks that every variant returns the same ids and distance bits.
usage: python tools/ivf_small_ab.py [nq ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from hnsw_clj_amd import _native, engine

dev = torch.device("cuda", 0)
nqs = [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64, 128]
x, Qa = bench.ivf_dataset(dev, 1_000_000, 1024, max(256, max(nqs)))
idx = engine.Index(x, os.environ.get("METRIC", "cosine"), 0)
del x
idx.ivf_build(1024, 10, 42)
variants = [("all off", {"FINISH_DIRECT": 0, "WORKLIST_FOLD": 0, "SEED_HALF": 0}), ("direct only", {"FINISH_DIRECT": None, "WORKLIST_FOLD": 0, "SEED_HALF": 0}),
            ("fold only", {"FINISH_DIRECT": 0, "WORKLIST_FOLD": None, "SEED_HALF": 0}), ("half-precision seed only", {"FINISH_DIRECT": 0, "WORKLIST_FOLD": 0, "SEED_HALF": None}),
            ("all three (default)", {"FINISH_DIRECT": None, "WORKLIST_FOLD": None, "SEED_HALF": None})]
for nq in nqs:
    Q = Qa[:nq].contiguous()
    o = (torch.empty((nq, 10), dtype=torch.int32, device=dev), torch.empty((nq, 10), dtype=torch.float32, device=dev))
    ref = None
    for name, tv in variants:
        for k, v in tv.items():
            _native.set_tuning(k, v)
        for _ in range(10):
            idx.ivf_search_dev(Q, 10, 32, out=o)
        torch.cuda.synchronize()
        got = (o[0].cpu().numpy().copy(), o[1].cpu().numpy().view(np.uint32).copy())
        if ref is None:
            ref = got
        same = np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
        best = 1e9
        for rep in range(5):
            t = time.perf_counter()
            for i in range(200):
                idx.ivf_search_dev(Q, 10, 32, out=o)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) / 200 * 1e6)
        lat = []
        for i in range(60):
            t = time.perf_counter()
            idx.ivf_search_dev(Q, 10, 32, out=o)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t) * 1e6)
        lat = sorted(lat[10:])
        idx.set_profiling(True)
        idx.rejection_stats(reset=True)
        idx.ivf_search_dev(Q, 10, 32, out=o)
        torch.cuda.synchronize()
        f32_rows, cands = idx.rejection_stats(reset=True)
        idx.set_profiling(False)
        print("nq=%4d  %-26s %7.1f us back to back   call + sync p50 %7.1f us   f32 rows per query %7.1f   same bits as the first variant: %s" % (nq, name, best, lat[len(lat) // 2], f32_rows / nq, same), flush=True)
    for k in ("FINISH_DIRECT", "WORKLIST_FOLD", "SEED_HALF"):
        _native.set_tuning(k, None)

# the schedule knobs of a batch of 32 on this box: workgroup target of the bounds pass, finish workgroups per query
if os.environ.get("SWEEP", "1") != "0":
    nq = 32
    Q = Qa[:nq].contiguous()
    o = (torch.empty((nq, 10), dtype=torch.int32, device=dev), torch.empty((nq, 10), dtype=torch.float32, device=dev))
    for key, vals in (("STREAM_WGS", (1024, 1536, 2048, 3072, 4096, 6144, 8192)), ("FINISH_SLICES", (8, 16, 32, 64)), ("FINISH_SPAN", (16, 32, 64))):
        for v in vals:
            _native.set_tuning(key, v)
            for _ in range(10):
                idx.ivf_search_dev(Q, 10, 32, out=o)
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(5):
                t = time.perf_counter()
                for i in range(200):
                    idx.ivf_search_dev(Q, 10, 32, out=o)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t) / 200 * 1e6)
            print("nq=32  %s=%d: %.1f us back to back" % (key, v, best), flush=True)
        _native.set_tuning(key, None)
