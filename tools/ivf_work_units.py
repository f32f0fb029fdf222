"""Developer tool: how much MFMA tile work the batched list scan has on k-means lists vs equally long lists."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hnsw_clj_amd import engine

dev = torch.device("cuda", 0)
n, nlist, nprobe, D, K = 1_000_000, 1024, 32, 768, 10
g = torch.Generator(device=dev)
g.manual_seed(42)
centers = torch.randn(nlist, D, generator=g, device=dev)
which = torch.randint(0, nlist, (n,), generator=g, device=dev)
x = centers[which] + 0.3 * torch.randn(n, D, generator=g, device=dev)
x /= x.norm(dim=1, keepdim=True)
g.manual_seed(43)
qw = torch.randint(0, nlist, (1024,), generator=g, device=dev)
Q = centers[qw] + 0.3 * torch.randn(1024, D, generator=g, device=dev)
Q /= Q.norm(dim=1, keepdim=True)
idx = engine.Index(x, "cosine", 0)
for kind in ("kmeans", "balanced"):
    if kind == "kmeans":
        idx.ivf_build(nlist, 10, 42)
    else:
        a, _ = idx.kmeans_assign(centers.cpu().numpy())
        order = np.argsort(a, kind="stable").astype(np.int32)
        off = np.zeros(nlist + 1, np.int64)
        off[1:] = np.cumsum(np.bincount(a, minlength=nlist))
        idx.set_ivf(centers.cpu().numpy(), off, order)
    _, off, _ = idx.get_ivf()
    lens = np.diff(off)
    import ctypes
    from hnsw_clj_amd import _native
    L = _native.lib()
    L.hnswgpu_debug_set_tile_stamps.argtypes = [ctypes.c_void_p]
    idx.ivf_search_dev(Q, K, nprobe)
    torch.cuda.synchronize()
    stamps = torch.zeros(4 * 65536, dtype=torch.int64, device=dev)
    L.hnswgpu_debug_set_tile_stamps(stamps.data_ptr())
    idx.ivf_search_dev(Q, K, nprobe)
    torch.cuda.synchronize()
    L.hnswgpu_debug_set_tile_stamps(None)
    st = stamps.cpu().numpy().reshape(-1, 4)
    st = st[st[:, 1] > 0]
    # the routing launch (1024 x 1024) also stamps: keep the list-scan launch = the later, larger time range
    t0 = st[:, 0].min()
    start, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0      # s_memrealtime ticks at 100 MHz -> us
    tiles, cntq = st[:, 3] & 0xffffffff, st[:, 3] >> 32
    late = start > 100                                                  # skip the short routing launch
    start, end, tiles, cntq, hw = start[late], end[late], tiles[late], cntq[late], st[late, 2]
    dur = end - start
    span = end.max() - start.min()
    print("%-9s stamped WGs %d | kernel span %.0f us | sum(dur) / (256 * span) = %.2f CU occupancy | per-tile us: mean %.1f p50 %.1f p95 %.1f | "
          "WG dur mean %.0f max %.0f | last start %.0f us" % (kind, len(dur), span, dur.sum() / (256 * span), (dur / tiles).mean(),
          np.median(dur / tiles), np.percentile(dur / tiles, 95), dur.mean(), dur.max(), start.max() - start.min()))
    for lo, hi in ((1, 2), (3, 4), (5, 8), (9, 64)):
        m = (tiles >= lo) & (tiles <= hi)
        if m.any():
            print("     tiles %d-%d: %d WGs, per-tile mean %.1f us" % (lo, hi, m.sum(), (dur[m] / tiles[m]).mean()))
    _, _, probes = idx.ivf_search(Q.cpu().numpy(), K, nprobe, want_probes=True)
    cnt = np.bincount(probes.ravel(), minlength=nlist)
    groups = -(-cnt // 32)
    tiles = -(-lens // 128)
    units = int((groups * tiles).sum())
    useful = float((cnt * lens).sum())
    print("%-9s lists: len min %d max %d | probes per list min %d max %d | empty lists %d | groups %d | group-tiles %d | "
          "fill %.3f | pairs*rows %.3g | corr(cnt,len) %.2f" % (kind, lens.min(), lens.max(), cnt.min(), cnt.max(), int((lens == 0).sum()),
          int(groups.sum()), units, useful / (units * 32 * 128), useful, np.corrcoef(cnt, lens)[0, 1]))
