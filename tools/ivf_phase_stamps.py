"""Developer tool: the latency chain of ONE small IVF search (routing -> bounds -> finish), from wall_clock64 stamps
(100 MHz) that the diagnostic build (tools/build_stamps.sh, -DHG_IVF_STAMPS) writes for query 0.
usage: HNSWGPU_LIBRARY=build_dbg/libhnswgpu_stamps.so python tools/ivf_phase_stamps.py [nq ...]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from hnsw_clj_amd import _native, engine

dev = torch.device("cuda", 0)
_nqs = [int(a) for a in sys.argv[1:]] or [1, 32]
x, Qa = bench.ivf_dataset(dev, 1_000_000, 1024, max(64, max(_nqs)))
idx = engine.Index(x, os.environ.get("METRIC", "cosine"), 0)
del x
idx.ivf_build(1024, 10, 42)
buf = torch.zeros(32, dtype=torch.int64, device=dev)
L = _native.lib()
L.hnswgpu_debug_set_tile_stamps.argtypes = [C.c_void_p]
L.hnswgpu_debug_set_tile_stamps(buf.data_ptr())
names = [(16, "routing kernel: first workgroup starts"), (17, "routing tail of query 0 begins (all centroid distances in)"),
         (18, "  nprobe nearest selected"), (19, "  probe table written, pairs filed"), (20, "  threshold seeded (tail ends)"),
         (22, "bounds kernel: first workgroup starts"), (24, "finish kernel: first workgroup starts"),
         (25, "  first workgroup has evaluated its survivors"), (26, "  last workgroup of query 0 begins the merge"),
         (27, "  results of query 0 written")]
for nq in _nqs:
    Q = Qa[:nq].contiguous()
    for _ in range(5):
        idx.ivf_search_dev(Q, 10, 32)
    torch.cuda.synchronize()
    rows = []
    for rep in range(20):
        buf.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx.ivf_search_dev(Q, 10, 32)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e6
        b = buf.cpu().numpy()
        ref = b[16] if b[16] else b[17]   # (two-launch routing: no stamp 16; relative to the tail of query 0)
        rows.append([wall] + [(b[s] - ref) * 1e-2 if b[s] else np.nan for s, _ in names])
    med = np.nanmedian(np.array(rows), axis=0)
    print("batch %d: call + sync %.1f us (median of 20); stamps relative to the routing kernel's start:" % (nq, med[0]))
    prev = 0.0
    for (s, n), v in zip(names, med[1:]):
        print("   %7.1f us  (+%5.1f)  %s" % (v, v - prev, n))
        prev = v if v == v else prev
