"""Developer tool: single-query HNSW latency on the headline set (31,173 x 768 clustered, heuristic builder) -- the
host entry point (hnswgpu_hnsw_search: mapped pinned I/O, host-polled completion), the kernel by hipEvents, what the
traversal still gathered itself, and -- with the diagnostic build (tools/build_stamps.sh, HNSWGPU_LIBRARY) -- the phases.
usage: [HNSWGPU_LIBRARY=build_dbg/libhnswgpu_stamps.so] [HNSWGPU_TUNE="KEY=v,..."] python tools/sq_probe.py [ef ...]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from hnsw_clj_amd import _native, engine

efs = [int(x) for x in sys.argv[1:]] or [100, 640]
dist = os.environ.get("SQ_DIST", "clustered")
cache = "/tmp/sq_%s_%s.npz" % (dist, os.environ.get("SQ_QUERIES", "256"))
if os.path.exists(cache):
    z = np.load(cache)
    base, queries = z["base"], z["q"]
else:
    base = bench.make_31k(dist, 42, 31173)
    queries = bench.make_31k(dist, 43, int(os.environ.get("SQ_QUERIES", "256")))
    np.savez(cache, base=base, q=queries)
dev = torch.device("cuda", 0)
idx = engine.Index(base, "cosine", 0)
idx.hnsw_build(16, 200, 42, **bench.BUILDERS[os.environ.get("SQ_BUILDER", "heuristic")])
stamps = "stamps" in os.environ.get("HNSWGPU_LIBRARY", "")
L = _native.lib()
buf = torch.zeros(64, dtype=torch.int64, device=dev)
L.hnswgpu_debug_set_tile_stamps.argtypes = [C.c_void_p]
if stamps or os.environ.get("SQ_COUNTERS", "1") == "1":
    L.hnswgpu_debug_set_tile_stamps(buf.data_ptr())


def pct(a):
    a = sorted(a)
    return "p50 %.1f  min %.1f  p95 %.1f us" % (a[len(a) // 2], a[0], a[int(len(a) * 0.95)])


for ef in efs:
    for nq in [int(x) for x in os.environ.get("SQ_NQ", "1,20").split(",")]:
        ids0, d0, st0 = idx.hnsw_search(queries[:nq], 10, ef, want_stats=True)
        lat = []
        for i in range(70):
            t = time.perf_counter()
            idx.hnsw_search(queries[i:i + nq], 10, ef)
            lat.append((time.perf_counter() - t) * 1e6)
        idx.set_profiling(True)
        idx.get_profile(engine.PROF_HNSW, reset=True)
        idx.rejection_stats(reset=True)
        for i in range(40):
            idx.hnsw_search(queries[i:i + nq], 10, ef)
        ms, cnt = idx.get_profile(engine.PROF_HNSW, reset=True)
        own, tot = idx.rejection_stats(reset=True)
        idx.set_profiling(False)
        print("ef %d, %d queries per call: host entry %s; kernel %.1f us; hops/query %.0f evals/query %.0f; gathered by the "
              "traversal itself %.1f %%" % (ef, nq, pct(lat[10:]), ms / max(cnt, 1) * 1e3, st0[:, 1].mean(), st0[:, 0].mean(),
                                            100.0 * own / max(tot, 1)), flush=True)
        b = buf.cpu().numpy()
        if b[32] > 0:   # solo_kernels.hpp: how the level-0 expansions of the launches above were served
            print("   level-0 expansions %d: node in the LDS cache %.1f %%, nothing to fetch %.1f %%, served by one more look "
                  "%.1f %%, own gather %.1f %% (%.2f rows each); fetcher ring entries %.2f per expansion, slots stolen %d"
                  % (b[32], 100.0 * b[33] / b[32], 100.0 * b[34] / b[32], 100.0 * b[35] / b[32], 100.0 * b[36] / b[32],
                     b[37] / max(b[36], 1), b[38] / b[32], b[39]), flush=True)
            print("   ring entries evaluated %.2f per expansion, of them appended by helpers (chase) %.2f" % (b[59] / b[32], b[60] / b[32]))
            if b[40:50].sum() > 0:   # -DHG_SOLO_STAMPS (tools/build_solo_stamps.sh): shader cycles of the sequencer per expansion
                names = ["select", "cache / adjacency", "visited filter", "one more look", "own gather", "survivor test", "buffer merge",
                         "admission", "mirror", "short-cut tails"]
                print("   sequencer cycles per level-0 expansion: " + ", ".join(
                    "%s %.0f" % (names[i], b[40 + i] / b[32]) for i in range(10) if names[i] != "-") + "; total %.0f" % (b[40:50].sum() / b[32]))
                d = b[50:58].astype(float)
                print("   admissions with the list full: %.1f %% of the expansions, %.2f survivors each; while it fills: %.1f %%, %.1f "
                      "survivors; merges of the buffer into the list: %.2f per query" % (100 * d[0] / b[32], d[2] / max(d[0], 1),
                      100 * d[1] / b[32], d[3] / max(d[1], 1), b[58] / 110.0 / nq))
            buf[32:61] = 0
        if stamps and (nq == 1 or os.environ.get("SQ_STAMPS_ANY")):
            names = ["level set-up", "select+adjacency+visited", "rejection test (int8 rows)", "f32 row gather+distances",
                     "merge 1 (rank)", "merge 2 (admit)", "merge 3 (scatter)", "epilogue"]
            slots = [0, 1, 8, 2, 3, 4, 5, 6]
            b = buf.cpu().numpy()
            hops = int(b[7])
            for n, v in zip(names, b[slots]):
                print("   %-28s %8.1f us  %5.1f %%   %.2f us/hop" % (n, v * 10e-3, 100.0 * v / max(b[slots].sum(), 1),
                                                                      v * 10e-3 / max(hops, 1)))
