"""Developer tool: BASELINE.json configs[3]/[4] at their per-GPU size on ONE MI355X.
  python tools/scale_check.py hnsw [clustered|manifold] [graph.clj|ultra_fast.clj]
                                     -> 1.25M x 1536 cosine HNSW (one GPU's shard of 10M x 1536), ef_search 256 and the
                                        first ef with recall@10 >= 0.98; clustered = SURVEY S4 (clustered-normalised,
                                        1024 centres per shard, noise 0.3; queries from the same mixture)
  python tools/scale_check.py ivf    -> 10M x 768 IVF-FLAT nlist 1024 nprobe 32, batch 1024"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from hnsw_clj_amd import engine

for kv in filter(None, os.environ.get("TUNE", "").split(",")):     # tuning-table overrides, e.g. TUNE=BUILD_TIMING=1
    engine.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)


def manifold(n, dim, r=48, noise=0.1, chunk=250_000):
    w = torch.randn(r, dim, generator=g, device=dev)
    out = torch.empty(n, dim, device=dev)
    for i in range(0, n, chunk):
        m = min(chunk, n - i)
        x = torch.randn(m, r, generator=g, device=dev) @ w / r ** 0.5 + noise * torch.randn(m, dim, generator=g, device=dev)
        out[i:i + m] = x / x.norm(dim=1, keepdim=True)
    return out, w


if sys.argv[1] == "hnsw":
    n, dim = 1_250_000, 1536
    data = sys.argv[2] if len(sys.argv) > 2 else "clustered"
    builder = sys.argv[3] if len(sys.argv) > 3 else "graph.clj"
    if data == "manifold":
        x, w = manifold(n, dim)
        Q = x[:4096] + 0.02 * torch.randn(4096, dim, generator=g, device=dev)
    else:
        cen = torch.randn(1024, dim, generator=g, device=dev)
        x = torch.empty(n, dim, device=dev)
        for i in range(0, n, 250_000):
            y = cen[torch.randint(0, 1024, (250_000,), generator=g, device=dev)] + 0.3 * torch.randn(250_000, dim, generator=g, device=dev)
            x[i:i + 250_000] = y / y.norm(dim=1, keepdim=True)
        g.manual_seed(43)
        Q = cen[torch.randint(0, 1024, (4096,), generator=g, device=dev)] + 0.3 * torch.randn(4096, dim, generator=g, device=dev)
    Q = (Q / Q.norm(dim=1, keepdim=True)).contiguous()
    idx = engine.Index(x, "cosine", 0)
    t = time.time()
    idx.hnsw_build(16, 200, 42, **bench.BUILDERS[builder])
    print("HNSW build %d x %d (%s rows, %s builder): %.1f s (%.0f vectors/s)" % (n, dim, data, builder, time.time() - t, n / (time.time() - t)), flush=True)
    ti, _ = idx.exact_knn_dev(Q[:512], 10)
    stats = torch.zeros((len(Q), 2), dtype=torch.int64, device=dev)
    reached = False
    for ef in (64, 128, 256, 384, 512, 768, 1024, 1536, 2048):
        if reached and ef > 256:
            break
        ids, _ = idx.hnsw_search_dev(Q, 10, ef, stats=stats)
        torch.cuda.synchronize()
        ev, hp = float(stats[:, 0].double().mean()), float(stats[:, 1].double().mean())
        t = time.time()
        for _ in range(3):
            ids, _ = idx.hnsw_search_dev(Q, 10, ef)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 3
        idx.set_profiling(True)                      # one more launch with the device counters of the rejection test
        idx.rejection_stats(reset=True)
        idx.hnsw_search_dev(Q, 10, ef)
        torch.cuda.synchronize()
        f32_rows, nb = idx.rejection_stats(reset=True)
        idx.set_profiling(False)
        code_row = 256 * ((dim + 255) // 256) + 16
        tested = f32_rows < 0.98 * nb
        gb = ((nb * code_row if tested else 0) + f32_rows * (4 * dim + 4) + hp * len(Q) * 4 * 32) / 1e9
        alg = (ev * 4 * dim + hp * 4 * 32) * len(Q) / 1e9
        rec = bench.recall_at_k(ids[:512], ti)
        reached = reached or rec >= 0.98
        print("  ef %3d  recall@10 %.4f  %d queries in %.2f ms = %.0f QPS; E %.0f H %.0f, f32 rows fetched %.0f per query -> %.2f GB "
              "requested = %.0f GB/s (%.2f of 8 TB/s); the reference algorithm's bytes: %.2f GB"
              % (ef, rec, len(Q), dt * 1e3, len(Q) / dt, ev, hp, f32_rows / len(Q), gb, gb / dt,
                 gb / dt / 8000, alg), flush=True)
    print("  the traversal's int8 test on this graph (rejection mode 1, measured on the first large launch): state %d, "
          "switched off: %s, f32 rows per neighbour with it on: %.3f" % idx.hnsw_rejection_state(), flush=True)
    # the same launches with the test forced on (mode 2) -- what the per-graph measurement saves or costs
    idx.set_rejection_test(2)
    for ef in (256,):
        for _ in range(2):
            idx.hnsw_search_dev(Q, 10, ef)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(3):
            idx.hnsw_search_dev(Q, 10, ef)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 3
        print("  ef %3d with the int8 test forced on (mode 2): %.2f ms = %.0f QPS" % (ef, dt * 1e3, len(Q) / dt), flush=True)
else:
    n, dim, nlist = 10_000_000, 768, 1024
    cen = torch.randn(nlist, dim, generator=g, device=dev)
    x = torch.empty(n, dim, device=dev)
    for i in range(0, n, 500_000):
        wch = torch.randint(0, nlist, (500_000,), generator=g, device=dev)
        y = cen[wch] + 0.3 * torch.randn(500_000, dim, generator=g, device=dev)
        x[i:i + 500_000] = y / y.norm(dim=1, keepdim=True)
    idx = engine.Index(x, "cosine", 0)
    del x
    t = time.time()
    idx.ivf_build(nlist, 10, 42)
    print("IVF build %d x %d nlist %d: %.1f s" % (n, dim, nlist, time.time() - t), flush=True)
    qw = torch.randint(0, nlist, (1024,), generator=g, device=dev)
    Q = cen[qw] + 0.3 * torch.randn(1024, dim, generator=g, device=dev)
    Q = (Q / Q.norm(dim=1, keepdim=True)).contiguous()
    for nq in (1, 32, 64, 1024):
        q = Q[:nq].contiguous()
        for _ in range(2):
            idx.ivf_search_dev(q, 10, 32)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(5):
            ids, _ = idx.ivf_search_dev(q, 10, 32)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 5
        print("  batch %4d: %.3f ms per batch = %.0f QPS" % (nq, dt * 1e3, nq / dt), flush=True)
    engine.set_tuning("STREAM_HOME", 0)                # the same batch without the home-list pass of large batches (A/B)
    ref_ids, ref_d = idx.ivf_search_dev(Q, 10, 32)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(5):
        idx.ivf_search_dev(Q, 10, 32)
    torch.cuda.synchronize()
    print("  batch 1024, home-list pass off: %.3f ms per batch" % ((time.time() - t) / 5 * 1e3), flush=True)
    engine.set_tuning("STREAM_HOME", None)
    ids, d = idx.ivf_search_dev(Q, 10, 32)
    torch.cuda.synchronize()
    print("  ids / distance bits equal with the pass on and off: %s / %s" % (
        bool((ids == ref_ids).all()), bool((d.view(torch.int32) == ref_d.view(torch.int32)).all())), flush=True)
    ti, _ = idx.exact_knn_dev(Q[:64].contiguous(), 10)
    ids, _ = idx.ivf_search_dev(Q[:64].contiguous(), 10, 32)
    torch.cuda.synchronize()
    print("  recall@10 vs exact: %.4f" % bench.recall_at_k(ids, ti))
