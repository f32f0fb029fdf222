"""Developer tool: time the MFMA tile kernel on the k-means assignment shape (n x dim rows against nlist centroids).
Ablations are compile-time variants of the library (tools/tile_variant.sh name=-DHG_TILE_ABLATE=<mask> builds
build_dbg/libhnswgpu_<name>.so with parts of the K-step removed); select one with HNSWGPU_LIBRARY.
usage: [HNSWGPU_LIBRARY=build_dbg/libhnswgpu_<name>.so] python tools/tile_ablate.py [n] [dim] [nlist]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hnsw_clj_amd import engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nlist = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
base = torch.randn((n, dim), device=dev, generator=g)
idx = engine.Index(base, "cosine", 0)
cen = base[:nlist].cpu().numpy()
idx.kmeans_assign(cen)
idx.set_profiling(True)
for _ in range(3):
    idx.kmeans_assign(cen)
ms, cnt = idx.get_profile(2)
fl = 2.0 * n * nlist * dim
print("lib=%s  n=%d dim=%d nlist=%d: tile kernel %.3f ms/launch -> %.1f TFLOP/s (%.0f %% of 157.3)" % (
    os.path.basename(os.environ.get("HNSWGPU_LIBRARY", "libhnswgpu.so")), n, dim, nlist, ms / cnt, fl / (ms / cnt * 1e-3) / 1e12,
    100 * fl / (ms / cnt * 1e-3) / 157.3e12))
