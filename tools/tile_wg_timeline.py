"""Developer tool: per-workgroup timeline of the MFMA tile kernel on the k-means assignment shape (one workgroup
per CU at a time: 136 KiB of LDS).  Uses the kernel's diagnostic stamps {start, end, hw id, tiles} to separate the
time a workgroup runs from the gap a CU sits idle between two workgroups (dispatch + LDS allocation).
usage: python tools/tile_wg_timeline.py [n] [dim] [nlist]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hnsw_clj_amd import _native, engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nlist = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
base = torch.randn((n, dim), device=dev, generator=g)
idx = engine.Index(base, "cosine", 0)
cen = base[:nlist].cpu().numpy()
idx.kmeans_assign(cen)
L = _native.lib()
L.hnswgpu_debug_set_tile_stamps.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(4 * 65536, dtype=torch.int64, device=dev)
L.hnswgpu_debug_set_tile_stamps(stamps.data_ptr())
idx.kmeans_assign(cen)
torch.cuda.synchronize()
L.hnswgpu_debug_set_tile_stamps(None)
st = stamps.cpu().numpy().reshape(-1, 4)
st = st[st[:, 1] > 0]
t0 = st[:, 0].min()
start, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0          # s_memrealtime: 100 MHz -> us
hw = st[:, 2]
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (((hw >> 32) & 0xf) << 8)
dur = end - start
cyc = (hw >> 36) * 16.0
print("shader clock while the kernel ran: %.0f MHz (cycles / 100 MHz wall stamps, mean over workgroups)" % (cyc / dur).mean())
print("workgroups %d on %d distinct CUs | span %.0f us | WG duration mean %.1f p50 %.1f p95 %.1f us" % (
    len(dur), len(np.unique(cu)), end.max() - start.min(), dur.mean(), np.median(dur), np.percentile(dur, 95)))
gaps = []
for c in np.unique(cu):
    m = cu == c
    o = np.argsort(start[m])
    s_, e_ = start[m][o], end[m][o]
    gaps.append(s_[1:] - e_[:-1])
gaps = np.concatenate(gaps)
print("gap between consecutive workgroups on one CU: mean %.1f p50 %.1f p95 %.1f us (negative = overlap)" % (
    gaps.mean(), np.median(gaps), np.percentile(gaps, 95)))
print("MFMA-only time of one workgroup at 2.4 GHz: %.1f us" % (2.0 * 32 * nlist * dim / 8 * 2 / 4096 * 64 / 2400.0))
