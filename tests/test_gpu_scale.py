"""BASELINE.json configs[3] and configs[4] at the size ONE GPU holds of them (10M rows over 8 GPUs = 1.25M per GPU):
size-independent properties at full per-GPU size, and bit-equality with the oracle on a 64-query subsample of the very
same index / graph.

configs[3]: hnsw.ivf-flat 10M x 768 sharded over 8 MI355X, nlist 1024, nprobe 32, batch 1024 (and the HBM-bound batch 32).
configs[4]: 10M x 1536 cosine HNSW, ef_search 256, one sub-graph per GPU.
Reference: src/hnsw/ann/partition/ivf_flat.clj:217-294, src/hnsw/ultra_fast.clj:151-212,346-374."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from util import assert_exact, assert_topk_parity, close  # noqa: E402

pytestmark = pytest.mark.gpu

N_SHARD = 1_250_000


@pytest.fixture(scope="module")
def eng(native_lib):
    from hnsw_clj_amd import engine

    assert engine.device_count() >= 1, "no GPU visible"
    return engine


def _check_topk(ids, d, n, k):
    assert (np.diff(d, axis=1) >= 0).all(), "distances not ascending"
    assert (ids >= 0).all() and (ids < n).all()
    assert all(len(set(r.tolist())) == k for r in ids), "duplicate ids in a result"


@pytest.fixture(scope="module")
def shard3(eng):
    """One GPU's share of configs[3]: 1.25M x 768 clustered-normalised rows (1024 true centres), nlist 1024 built on the
    device (k-means++ seed 42, 10 Lloyd passes), 16,384 held-out queries of the same mixture (in blocks of 4096 draws: the
    first 4096 are the ones rounds 1-4 used).  Shared by the tests below."""
    import torch

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    n, nlist = N_SHARD, 1024
    cen = torch.randn(nlist, 768, generator=g, device=dev)
    x = cen[torch.randint(0, nlist, (n,), generator=g, device=dev)] + 0.3 * torch.randn(n, 768, generator=g, device=dev)
    x /= x.norm(dim=1, keepdim=True)
    g.manual_seed(43)
    Q = torch.cat([cen[torch.randint(0, nlist, (4096,), generator=g, device=dev)] + 0.3 * torch.randn(4096, 768, generator=g, device=dev)
                   for _ in range(4)])
    Q = (Q / Q.norm(dim=1, keepdim=True)).contiguous()
    base = x.cpu().numpy()
    qh = Q.cpu().numpy()
    idx = eng.Index(x, "cosine")
    del x
    idx.ivf_build(nlist, 10, 42)
    cent, off, lids = idx.get_ivf()
    assert off[0] == 0 and off[-1] == n and np.array_equal(np.sort(lids), np.arange(n, dtype=np.int32))
    yield dict(idx=idx, base=base, Q=Q, qh=qh, cent=cent, off=off, lids=lids, n=n, nlist=nlist)
    idx.close()


def test_config3_ivf_per_gpu_shard(eng, oracle, shard3):
    """1.25M x 768, nlist 1024, nprobe 32, k 10, with the suite's pinned boundary (HNSWGPU_TILE_PAIRS = 12,
    tests/conftest.py): batch 1024 goes through the f32 MFMA tile scan (the kernel of handles without int8 rows and of
    k > 256), batch 32 through the survivor stream (GEMV order), each against the oracle's matching order."""
    import torch

    O = oracle
    s = shard3
    idx, base, Q, qh, cent, off, lids, n = s["idx"], s["base"], s["Q"], s["qh"], s["cent"], s["off"], s["lids"], s["n"]
    nprobe, k = 32, 10
    res = {}
    for nq in (1024, 32):
        Qb = Q[:nq].contiguous()
        i1, d1 = idx.ivf_search_dev(Qb, k, nprobe)
        i2, d2 = idx.ivf_search_dev(Qb, k, nprobe)
        torch.cuda.synchronize()
        assert torch.equal(i1, i2) and torch.equal(d1.view(torch.int32), d2.view(torch.int32)), "not idempotent"
        ids, d = i1.cpu().numpy(), d1.cpu().numpy()
        _check_topk(ids, d, n, k)
        res[nq] = (ids, d)
        # every returned distance is the true distance of that id (gather order: bit-equal on the GEMV path,
        # within the tolerance of the two summation orders on the MFMA path)
        for r in (0, nq // 2, nq - 1):
            true = idx.batch_distances(qh[r], ids[r])
            if nq == 32:
                np.testing.assert_array_equal(true.view(np.uint32), d[r].view(np.uint32))
            else:
                assert close(d[r], true).all()
    # the two kernels agree within tolerance; recall against GPU brute force over all 1.25M rows
    assert_topk_parity(res[32][0], res[32][1], res[1024][0][:32], res[1024][1][:32], "gemv vs tile at 1.25M")
    ei, _ = idx.exact_knn_dev(Q[:128].contiguous(), k)
    torch.cuda.synchronize()
    ei = ei.cpu().numpy()
    rec = np.mean([len(set(res[1024][0][r]) & set(ei[r])) / k for r in range(128)])
    assert rec >= 0.9, rec
    # oracle, same centroids and lists, 64-query subsample: the batch-1024 answers in MFMA order, a 64-query batch
    # (64 * 32 pairs = 2 per list: the survivor stream) in the GEMV order -- ids and distance bits
    oi, od, _ = O.ivf_search(base, cent, off, lids, qh[:64], k, nprobe, mode=O.MODE_MFMA)
    assert_exact(res[1024][0][:64], res[1024][1][:64], oi, od, "config3 batch 1024 vs oracle (MFMA order)")
    i64, d64 = idx.ivf_search(qh[:64], k, nprobe)
    oi, od, _ = O.ivf_search(base, cent, off, lids, qh[:64], k, nprobe, mode=O.MODE_DEV)
    assert_exact(i64, d64, oi, od, "config3 batch 64 vs oracle (GEMV order)")
    fi, fd, _ = O.ivf_search(base, cent, off, lids, qh[:16], k, nprobe)           # f64 reference order
    assert_topk_parity(res[1024][0][:16], res[1024][1][:16], fi, fd, "config3 vs f64 oracle")


@pytest.mark.parametrize("nq", [256, 1024, 4096, 8192, 16384])
def test_config3_ivf_production_path_against_oracle(eng, oracle, shard3, nq, tune):
    """The path a DEFAULT handle takes (what bench.py times and `profiles/` report): no pinned boundary, rejection mode 1
    with its first-search calibration -- int8 bounds on the matrix cores -> half-precision pass -> f32 finish, cosine --
    at 1.25M x 768 / nlist 1024 / nprobe 32 / k 10 and batches of 256 / 1024 / 4096 / 8192 / 16384 (ivf_flat.clj:217-294; from
    256 (query, list) pairs per list -- 8192 queries here -- the bounds pass meets a staged list row with TWO 32-query column
    blocks, stream_bounds_kernel<.., QB = 2>: the library's launch counter says that it ran).  A 64-query
    subsample spread over the batch is compared with the oracle: ids and distance bits against its device (GEMV) order,
    ids and distances within 1e-4 against its f64 reference order; the counters say that the stream and the
    half-precision pass really ran (a handle whose calibration had switched the stream off would take the f32 scans)."""
    import torch

    O = oracle
    s = shard3
    idx, base, Q, qh, cent, off, lids, n = s["idx"], s["base"], s["Q"], s["qh"], s["cent"], s["off"], s["lids"], s["n"]
    nprobe, k = 32, 10
    tune.unset("TILE_PAIRS")
    idx.set_rejection_test(1)                 # the default mode: calibrates at the next IVF search
    try:
        Qb = Q[:nq].contiguous()
        idx.ivf_search_dev(Qb, k, nprobe)     # (calibration + scratch growth happen here)
        idx.set_profiling(True)
        idx.rejection_stats(reset=True)
        wide0 = eng.debug_counter("bounds_two_column_blocks")
        i1, d1 = idx.ivf_search_dev(Qb, k, nprobe)
        torch.cuda.synchronize()
        f32_rows, cand = idx.rejection_stats(reset=True)
        assert (eng.debug_counter("bounds_two_column_blocks") > wide0) == (nq * nprobe >= 256 * s["nlist"]), "bounds pass: column blocks"
        idx.set_profiling(False)
        i2, d2 = idx.ivf_search_dev(Qb, k, nprobe)
        torch.cuda.synchronize()
        assert torch.equal(i1, i2) and torch.equal(d1.view(torch.int32), d2.view(torch.int32)), "not idempotent"
        ids, d = i1.cpu().numpy(), d1.cpu().numpy()
        _check_topk(ids, d, n, k)
        # the survivor stream ran (candidates counted by its finish kernel), and behind the half-precision pass:
        # a few dozen f32 rows per query remain of ~39,000 candidates (the int8 pass alone leaves ~1,200)
        assert cand >= nq * nprobe * 600, (f32_rows, cand)
        assert f32_rows < 0.005 * cand, "f32 rows per query %.1f of %.1f candidates: the half-precision pass did not run" % (
            f32_rows / nq, cand / nq)
        sub = np.unique(np.linspace(0, nq - 1, 64).astype(np.int64))
        oi, od, _ = O.ivf_search(base, cent, off, lids, qh[sub], k, nprobe, mode=O.MODE_DEV)
        assert_exact(ids[sub], d[sub], oi, od, "config3 production path, batch %d, vs oracle (GEMV order)" % nq)
        fi, fd, _ = O.ivf_search(base, cent, off, lids, qh[sub[:16]], k, nprobe)          # f64 reference order
        assert_topk_parity(ids[sub[:16]], d[sub[:16]], fi, fd, "config3 production path, batch %d, vs f64 oracle" % nq)
        for r in (0, nq // 2, nq - 1):        # returned distance == the row's true distance, bit for bit
            np.testing.assert_array_equal(idx.batch_distances(qh[r], ids[r]).view(np.uint32), d[r].view(np.uint32))
    finally:
        idx.set_profiling(False)
        idx.set_rejection_test(2)             # the suite's mode for the tests that share this index


@pytest.mark.parametrize("nq", [1024, 4096])
def test_config3_ivf_euclidean_large_batches_against_oracle(eng, oracle, shard3, nq, tune):
    """The Euclidean metric at configs[3]'s per-GPU size through the path a DEFAULT handle takes (int8 bounds on the matrix
    cores, the home-list pass through `|q - v'|^2 = |q|^2 - 2 q.v' + |v'|^2`, the per-survivor half-precision pass, f32
    finish; euclidean-distance-ultra, ultra_fast.clj:43-51, inside ivf_flat.clj:217-294): batches of 1024 and 4096 on the
    same 1.25M x 768 rows, own k-means lists, a 64-query subsample against the oracle -- ids and distance bits in its device
    (GEMV) order, ids and distances within 1e-4 against its f64 order.  (test_ivf_stream_equals_the_f32_scan_at_scale
    compares the stream with the engine's own f32 scan; this puts the oracle beside the published large-batch numbers.)"""
    import torch

    O = oracle
    s = shard3
    base, Q, qh, n = s["base"], s["Q"], s["qh"], s["n"]
    nprobe, k, nlist = 32, 10, 1024
    tune.unset("TILE_PAIRS")
    with eng.Index(torch.from_numpy(base).to(Q.device), "l2") as idx:
        idx.set_rejection_test(1)             # the default mode: calibrates at the first IVF search
        idx.ivf_build(nlist, 4, 42)
        cent, off, lids = idx.get_ivf()
        Qb = Q[:nq].contiguous()
        idx.ivf_search_dev(Qb, k, nprobe)     # (calibration + scratch growth happen here)
        idx.set_profiling(True)
        idx.rejection_stats(reset=True)
        i1, d1 = idx.ivf_search_dev(Qb, k, nprobe)
        torch.cuda.synchronize()
        f32_rows, cand = idx.rejection_stats(reset=True)
        idx.set_profiling(False)
        ids, d = i1.cpu().numpy(), d1.cpu().numpy()
        _check_topk(ids, d, n, k)
        assert cand >= nq * nprobe * 600, (f32_rows, cand)
        assert f32_rows < 0.005 * cand, "f32 rows per query %.1f of %.1f candidates: the half-precision pass did not run" % (
            f32_rows / nq, cand / nq)
        sub = np.unique(np.linspace(0, nq - 1, 64).astype(np.int64))
        oi, od, _ = O.ivf_search(base, cent, off, lids, qh[sub], k, nprobe, metric=O.L2, mode=O.MODE_DEV)
        assert_exact(ids[sub], d[sub], oi, od, "config3 Euclidean production path, batch %d, vs oracle (GEMV order)" % nq)
        fi, fd, _ = O.ivf_search(base, cent, off, lids, qh[sub[:16]], k, nprobe, metric=O.L2)   # f64 reference order
        assert_topk_parity(ids[sub[:16]], d[sub[:16]], fi, fd, "config3 Euclidean production path, batch %d, vs f64 oracle" % nq)


def test_config4_hnsw_per_gpu_shard(eng, oracle):
    """1.25M x 1536 cosine, clustered-normalised rows (1024 centres, noise 0.3; queries held out of the same
    mixture -- bench.py's sharded_hnsw leg), graph built on the device by the heuristic builder (M 16, ef_construction 200,
    graph.clj:162-232 selection; closest-m lists reach recall 0.09 on these rows), 1024 queries at ef_search 256: the
    traversal is HBM-resident (7.7 GB of rows) and keeps its visited stamps in HBM."""
    import torch

    O = oracle
    dev = torch.device("cuda", 0)
    n, dim, ef, nq, k = N_SHARD, 1536, 256, 1024, 10
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    cen = torch.randn(1024, dim, generator=g, device=dev)

    def mixture(m):
        out = torch.empty(m, dim, device=dev)
        for i in range(0, m, 250_000):
            c = min(250_000, m - i)
            y = cen[torch.randint(0, 1024, (c,), generator=g, device=dev)] + 0.3 * torch.randn(c, dim, generator=g, device=dev)
            out[i:i + c] = y / y.norm(dim=1, keepdim=True)
        return out

    g.manual_seed(2000)
    x = mixture(n)
    g.manual_seed(43)
    Q = mixture(nq)
    base = x.cpu().numpy()
    qh = Q.cpu().numpy()
    with eng.Index(x, "cosine") as idx:
        del x
        idx.hnsw_build(16, 200, 42, heuristic=True)
        stats = torch.zeros((nq, 2), dtype=torch.int64, device=dev)
        i1, d1 = idx.hnsw_search_dev(Q, k, ef, stats=stats)
        i2, d2 = idx.hnsw_search_dev(Q, k, ef)
        torch.cuda.synchronize()
        assert torch.equal(i1, i2) and torch.equal(d1.view(torch.int32), d2.view(torch.int32)), "not idempotent"
        ids, d, st = i1.cpu().numpy(), d1.cpu().numpy(), stats.cpu().numpy()
        _check_topk(ids, d, n, k)
        assert st[:, 0].min() > ef and st[:, 1].min() >= ef // 2, "counters: evaluations / expansions per query"
        for q in (0, 511, 1023):                     # returned distance == the row's true distance, bit for bit
            np.testing.assert_array_equal(idx.batch_distances(qh[q], ids[q]).view(np.uint32), d[q].view(np.uint32))
        ei, _ = idx.exact_knn_dev(Q[:256].contiguous(), k)
        torch.cuda.synchronize()
        ei = ei.cpu().numpy()
        rec = np.mean([len(set(ids[q]) & set(ei[q])) / k for q in range(256)])
        assert rec >= 0.97, rec
        # the oracle on the SAME graph, 64-query subsample: ids, distance bits, both traversal counters
        gr = idx.get_graph()
        og = O.Graph(gr.levels, gr.l0_adj, gr.up_off, gr.up_adj, gr.M, gr.entry, gr.max_level)
        oi, od, ost, _ = O.hnsw_search(base, og, qh[:64], k, ef=ef, mode=O.MODE_DEV, nthreads=8)
        assert_exact(ids[:64], d[:64], oi, od, "config4 vs oracle (device order)")
        np.testing.assert_array_equal(st[:64], ost)
        fi, fd, _, _ = O.hnsw_search(base, og, qh[:16], k, ef=ef, nthreads=8)          # f64 reference order
        assert_topk_parity(ids[:16], d[:16], fi, fd, "config4 vs f64 oracle")
