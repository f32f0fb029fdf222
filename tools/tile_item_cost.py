"""Developer tool: what does a work item of the batched IVF tile scan cost by the size of its query group?
Diagnostic stamps of tile_scan_kernel ({start, end, hw id, tiles | cnt << 32} per workgroup), one workgroup per item
(HNSWGPU_TUNE=TILE_PERSIST=0 is forced).  usage: python tools/tile_item_cost.py [nq]"""
import ctypes
import os
import sys

os.environ["HNSWGPU_TUNE"] = "TILE_PERSIST=0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from hnsw_clj_amd import _native, engine

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
x, Qa = bench.ivf_dataset(dev, 1_000_000, 1024, 4096)
idx = engine.Index(x, "cosine", 0)
del x
idx.ivf_build(1024, 10, 42)
Q = Qa[:nq].contiguous()
for _ in range(3):
    idx.ivf_search_dev(Q, 10, 32)
L = _native.lib()
L.hnswgpu_debug_set_tile_stamps.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(4 * (1 << 20), dtype=torch.int64, device=dev)
L.hnswgpu_debug_set_tile_stamps(stamps.data_ptr())
idx.ivf_search_dev(Q, 10, 32)
torch.cuda.synchronize()
L.hnswgpu_debug_set_tile_stamps(None)
st = stamps.cpu().numpy().reshape(-1, 4)
st = st[st[:, 1] > 0]
dur = (st[:, 1] - st[:, 0]) / 100.0
tiles = st[:, 3] & 0xffffffff
cnt = st[:, 3] >> 32
big = tiles >= 1                                  # the routing launch (1024 centroids = 4 tiles, full groups) is in here too
print("items %d, span %.0f us, sum of item time %.0f us (/256 CUs = %.0f us)" % (len(dur), (st[:, 1].max() - st[:, 0].min()) / 100.0,
                                                                           dur.sum(), dur.sum() / 256))
tot = dur.sum()
for lo, hi in ((1, 4), (5, 8), (9, 16), (17, 24), (25, 32)):
    m = (cnt >= lo) & (cnt <= hi)
    if m.any():
        print("group of %2d-%2d queries: %5d items, %4.1f us per tile, %5.1f %% of the item time, %5.1f %% of the (query, row) pairs" % (
            lo, hi, m.sum(), (dur[m] / np.maximum(tiles[m], 1)).mean(), 100 * dur[m].sum() / tot,
            100 * (cnt[m] * tiles[m]).sum() / (cnt * tiles).sum()))
