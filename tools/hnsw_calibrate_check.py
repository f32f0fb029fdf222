"""Developer tool: what rejection mode 1's per-graph measurement (hnswgpu_hnsw_rejection_state) decides, and what it is worth:
configs[4]'s per-GPU shard (1.25M x 1536 clustered-normalised, heuristic builder) and the 31k x 768 headline set, each with
the traversal's int8 test forced on (mode 2), off (mode 0) and left to the measurement (mode 1).
usage: python tools/hnsw_calibrate_check.py [shard|headline ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from hnsw_clj_amd import engine

dev = torch.device("cuda", 0)


def run(name, x, Q, efs, dim):
    idx = engine.Index(x, "cosine", 0)
    t = time.time()
    idx.hnsw_build(16, 200, 42, **bench.BUILDERS["heuristic"])
    print("%s: %d x %d, heuristic builder %.1f s" % (name, idx.n, dim, time.time() - t), flush=True)
    ti, _ = idx.exact_knn_dev(Q[:512], 10)
    stats = torch.zeros((len(Q), 2), dtype=torch.int64, device=dev)
    code_row = 256 * ((dim + 255) // 256) + 16
    for ef in efs:
        ids, _ = idx.hnsw_search_dev(Q, 10, ef, stats=stats)
        torch.cuda.synchronize()
        ev, hp = float(stats[:, 0].double().mean()), float(stats[:, 1].double().mean())
        rec = bench.recall_at_k(ids[:512], ti)
        alg = (ev * 4 * dim + hp * 4 * 32) * len(Q) / 1e9
        for mode, label in ((2, "int8 test on (mode 2)"), (0, "off (mode 0)"), (1, "measured (mode 1, the default)")):
            idx.set_rejection_test(mode)
            for _ in range(3):                       # (mode 1: the first large launch measures, a later one reads the verdict)
                idx.hnsw_search_dev(Q, 10, ef)
            torch.cuda.synchronize()
            t = time.time()
            for _ in range(3):
                idx.hnsw_search_dev(Q, 10, ef)
            torch.cuda.synchronize()
            dt = (time.time() - t) / 3
            idx.set_profiling(True)
            idx.rejection_stats(reset=True)
            idx.hnsw_search_dev(Q, 10, ef)
            torch.cuda.synchronize()
            f32_rows, nb = idx.rejection_stats(reset=True)
            idx.set_profiling(False)
            tested = f32_rows < 0.98 * nb
            gb = ((nb * code_row if tested else 0) + f32_rows * (4 * dim + 4) + hp * len(Q) * 4 * 32) / 1e9
            st = idx.hnsw_rejection_state()
            print("  ef %3d recall@10 %.4f  %-32s %d queries in %.2f ms = %.0f QPS; f32 rows %.0f of %.0f neighbours per query; %.2f GB "
                  "requested (reference algorithm: %.2f GB) = %.0f GB/s = %.2f of 8 TB/s%s"
                  % (ef, rec, label, len(Q), dt * 1e3, len(Q) / dt, f32_rows / len(Q), nb / len(Q), gb, alg, gb / dt, gb / dt / 8000,
                     "; measurement: state %d, test off %s, %.2f of the rows left to fetch" % (st[0], st[1], st[2]) if mode == 1 else ""),
                  flush=True)
    idx.close()


which = sys.argv[1:] or ["shard", "headline"]
if "shard" in which:
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    n, dim = 1_250_000, 1536
    cen = torch.randn(1024, dim, generator=g, device=dev)
    x = torch.empty(n, dim, device=dev)
    for i in range(0, n, 250_000):
        y = cen[torch.randint(0, 1024, (250_000,), generator=g, device=dev)] + 0.3 * torch.randn(250_000, dim, generator=g, device=dev)
        x[i:i + 250_000] = y / y.norm(dim=1, keepdim=True)
    g.manual_seed(43)
    Q = cen[torch.randint(0, 1024, (4096,), generator=g, device=dev)] + 0.3 * torch.randn(4096, dim, generator=g, device=dev)
    Q = (Q / Q.norm(dim=1, keepdim=True)).contiguous()
    run("configs[4] per-GPU shard", x, Q, (128, 256), dim)
    del x
if "headline" in which:
    base = bench.make_31k("clustered", 42, bench.N31K)
    Q = torch.from_numpy(bench.make_31k("clustered", 43, 10000)).to(dev)
    run("31k headline set", torch.from_numpy(base).to(dev), Q, (640,), 768)
