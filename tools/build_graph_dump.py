"""Developer tool: build an HNSW graph on the device and dump it (npz) -- to compare two builds of the library
(HNSWGPU_LIBRARY=...) or two linker thread counts (HNSWGPU_BUILD_THREADS=...) for identical graphs.
usage: python tools/build_graph_dump.py out.npz [n] [dim]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hnsw_clj_amd import datagen, engine

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 31173
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 128
base = datagen.generate_dataset(n, dim, "clustered", num_clusters=64, noise_level=0.5, seed=42)
with engine.Index(base, "cosine", 0) as idx:
    idx.hnsw_build(16, 200, 42)
    t = time.time()
    idx.hnsw_build(16, 200, 42)
    print("build %d x %d: %.3f s (%s, threads %s)" % (n, dim, time.time() - t, os.environ.get("HNSWGPU_LIBRARY", "product"),
                                                     os.environ.get("HNSWGPU_BUILD_THREADS", "default")))
    g = idx.get_graph()
np.savez(out, levels=g.levels, l0=g.l0_adj, up_off=g.up_off, up=g.up_adj, entry=g.entry, max_level=g.max_level)
