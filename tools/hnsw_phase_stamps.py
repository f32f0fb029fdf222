"""Developer tool: where does a single-query HNSW traversal spend its time?  Needs the diagnostic build
(tools/build_stamps.sh -> build_dbg/libhnswgpu_stamps.so, kernels compiled with -DHG_HNSW_STAMPS) selected with
HNSWGPU_LIBRARY.  Prints per-phase totals of wall_clock64 ticks (100 MHz) for query 0 of a launch.
usage: HNSWGPU_LIBRARY=build_dbg/libhnswgpu_stamps.so python tools/hnsw_phase_stamps.py [ef] [nq]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from hnsw_clj_amd import _native, engine

ef = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
base = bench.make_31k("manifold", 42, 31173)
queries = bench.make_31k("manifold", 43, max(nq, 64))
dev = torch.device("cuda", 0)
idx = engine.Index(base, "cosine", 0)
idx.hnsw_build(16, 200, 42)
Q = torch.from_numpy(queries).to(dev)
buf = torch.zeros(12, dtype=torch.int64, device=dev)
L = _native.lib()
L.hnswgpu_debug_set_tile_stamps.argtypes = [C.c_void_p]
L.hnswgpu_debug_set_tile_stamps(buf.data_ptr())
names = ["level set-up", "select+adjacency+visited", "rejection test (int8 rows)", "f32 row gather+distances", "merge 1 (rank)",
         "merge 2 (admit)", "merge 3 (scatter)", "epilogue"]
slots = [0, 1, 8, 2, 3, 4, 5, 6]
for rep in range(3):
    lat = []
    for i in range(20):
        t0 = time.perf_counter()
        idx.hnsw_search_dev(Q[:nq], 10, ef)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    b = buf.cpu().numpy()
    hops = int(b[7])
    tot = float(b[slots].sum()) * 10e-3  # us
    print("nq=%d ef=%d hops=%d kernel-side total %.1f us (%.2f us/hop)  wall p50 %.3f ms" % (
        nq, ef, hops, tot, tot / max(hops, 1), sorted(lat)[10]))
    print("   shader clock during the query: %.0f MHz (%d cycles in %.1f us)" % (b[10] / max(b[11] * 10e-3, 1e-9), b[10], b[11] * 10e-3))
    for n, v in zip(names, b[slots]):
        print("   %-28s %8.1f us  %5.1f %%   %.2f us/hop" % (n, v * 10e-3, 100.0 * v / max(b[slots].sum(), 1), v * 10e-3 / max(hops, 1)))
