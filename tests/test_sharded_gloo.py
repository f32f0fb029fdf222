"""World-size-2 tests of the multi-GPU drivers (hnsw-clj_amd/sharded.py) on CPU ranks with gloo.

* ShardedIVF: ONE IVF index whose inverted lists are dealt to the ranks -- distributed build (seeding, Lloyd with
  all-reduced list sums, dealing, the all-to-all row exchange) and search (same routing on every rank, all-gather, merge
  by (distance, position in the whole index's candidate stream)) must equal the oracle's search of the UNSHARDED index
  with the same centroids and lists: ids, distances and tie order.
* ShardedSearcher: independent sub-indexes over row ranges == search over the whole base.

On CPU ranks the local operations are oracle-backed stand-ins (tests/sharded_util.py: there is no CPU path in the
product); the `gpu`-marked variant runs the same two ranks with the product's HIP kernels on cuda:0."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _merge_np(all_ids, all_d):
    """Stand-in for hnswgpu_merge_topk_dev with the same contract (ties: lower shard, then rank)."""
    ns, nq, k = all_ids.shape
    ids, d = all_ids.numpy(), all_d.numpy()
    oi = np.full((nq, k), -1, np.int32)
    od = np.full((nq, k), np.inf, np.float32)
    for q in range(nq):
        flat = [(d[s, q, r], s * k + r, ids[s, q, r]) for s in range(ns) for r in range(k) if ids[s, q, r] >= 0]
        flat.sort()
        for i, t in enumerate(flat[:k]):
            oi[q, i], od[q, i] = t[2], t[0]
    return torch.from_numpy(oi), torch.from_numpy(od)


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker(rank, world, port, n, dim, k, out):
    _init(rank, world, port)
    from hnsw_clj_amd.sharded import ShardedSearcher, shard_range
    from oracle import oracle as O

    base = O.generate_dataset(n, dim).astype(np.float32)
    base[5] = base[n - 3]                          # an exact cross-shard tie
    Q = O.generate_dataset(6, dim, seed=43).astype(np.float32)
    Q[0] = base[5]
    lo, hi = shard_range(n, rank, world)

    def local_search(Qt, kk):
        ids, d, _ = O.exact_knn(base[lo:hi], Qt.numpy(), kk)
        return torch.from_numpy(ids), torch.from_numpy(d.astype(np.float32))

    s = ShardedSearcher(local_search, lo, merge_fn=_merge_np)
    ids, d = s.search(torch.from_numpy(Q), k)
    if rank == 0:
        ei, ed, _ = O.exact_knn(base, Q, k)
        out["ok_ids"] = bool(np.array_equal(ids.numpy(), ei))
        out["ok_d"] = bool(np.allclose(d.numpy(), ed, rtol=1e-6))
        out["tie"] = ids.numpy()[0, :2].tolist()
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_search_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    n, dim, k = 101, 16, 7                          # 101 rows: uneven shards (51 + 50)
    mp.spawn(_worker, args=(2, _free_port(), n, dim, k, out), nprocs=2, join=True)
    assert out["ok_ids"] and out["ok_d"]
    assert out["tie"] == [5, n - 3]                 # equal distances: lower global row (lower shard) first


def _ivf_worker(rank, world, port, use_gpu, out):
    _init(rank, world, port)
    from hnsw_clj_amd.sharded import Comm, EngineOps, ShardedIVF, deal_lists, lists_from_assign, shard_range
    from oracle import oracle as O
    from sharded_util import OracleOps

    n, dim, nlist, nprobe, k = 1501, 24, 12, 5, 9   # uneven row ranges (751 + 750), 12 lists dealt to 2 ranks
    base = O.generate_dataset(n, dim, "clustered", num_clusters=6, noise_level=0.5).astype(np.float32)
    base[700:704] = base[900]                        # exact duplicates: ties inside one list ...
    base[np.arange(12) * 120 + 7, 0] = 7.0           # ... and rows all over the index that tie EXACTLY for the query
    Q = O.generate_dataset(7, dim, "clustered", num_clusters=6, noise_level=0.5, seed=43).astype(np.float32)
    Q[0] = base[900]
    Q[1] = 0.0
    Q[1, 0] = 4.0                                    #     4 e_0: -dot = -28.0 for each of them, in whatever list it lives
    lo, hi = shard_range(n, rank, world)
    dev = torch.device("cuda", 0) if use_gpu else torch.device("cpu")
    metric = O.DOT
    ops = EngineOps(0) if use_gpu else OracleOps(O, metric)
    x = torch.from_numpy(base[lo:hi]).to(dev)
    idx = ShardedIVF.build(x, "dot", nlist, 4, 42, comm=Comm(device=dev if use_gpu else None), ops=ops)
    ids, d = idx.search(torch.from_numpy(Q).to(dev), k, nprobe)
    ids, d = ids.cpu().numpy(), d.cpu().numpy()
    # the unsharded index every rank can rebuild: same centroids, the gathered assignment, lists in index order
    comm = idx.comm
    counts = comm.all_gather(torch.tensor([len(idx.assign)])).view(-1).tolist()
    pad = np.full(max(counts), -1, np.int32)
    pad[:len(idx.assign)] = idx.assign
    allp = comm.all_gather(torch.from_numpy(pad)).numpy()
    assign = np.concatenate([allp[r, :counts[r]] for r in range(world)])
    off, lids = lists_from_assign(assign, nlist)
    mode = O.MODE_DEV
    if use_gpu:                                       # the kernel the batch selects: 7 * 5 = 35 pairs <= 12 * 12 lists: the GEMV order
        mode = O.MODE_MFMA if len(Q) * nprobe > 12 * nlist else O.MODE_DEV
    oi, od, _ = O.ivf_search(base, idx.centroids, off, lids, Q, k, nprobe, metric=metric, mode=mode)
    out["ids_%d" % rank] = bool(np.array_equal(ids, oi))
    out["d_%d" % rank] = bool(np.array_equal(d.view(np.uint32), od.astype(np.float32).view(np.uint32)))
    out["held_%d" % rank] = int(idx.shard.rows.shape[0]) if not use_gpu else int(idx.shard.n)
    if rank == 0:
        owner = deal_lists(np.diff(off), world)
        out["owner_ok"] = bool(np.array_equal(owner, idx.owner))
        out["balanced"] = [int(np.diff(off)[owner == r].sum()) for r in range(world)]
        out["some_cross_list_tie"] = bool(any(len(np.unique(od[q])) < k for q in range(len(Q))))
    idx.close()
    dist.barrier()
    dist.destroy_process_group()


def _check_ivf(out, world):
    for r in range(world):
        assert out["ids_%d" % r], "rank %d: merged ids differ from the unsharded index" % r
        assert out["d_%d" % r], "rank %d: merged distances differ from the unsharded index" % r
    assert out["owner_ok"]
    assert sum(out["held_%d" % r] for r in range(world)) == 1501
    assert [out["held_%d" % r] for r in range(world)] == out["balanced"]
    assert max(out["balanced"]) - min(out["balanced"]) < 1501 // 4
    assert out["some_cross_list_tie"], "the data set is meant to hold exact ties"


def test_list_sharded_ivf_world2():
    """One IVF index, lists dealt to 2 CPU ranks (oracle-backed local ops): build + search == unsharded oracle."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ivf_worker, args=(2, _free_port(), False, out), nprocs=2, join=True)
    _check_ivf(out, 2)


@pytest.mark.gpu
def test_list_sharded_ivf_world2_hip_kernels(native_lib):
    """The same two ranks (gloo, both on cuda:0) with the product's local operations: HIP list scan on a shard handle
    (hnswgpu_set_ivf_shard / hnswgpu_ivf_search_shard_dev) and hnswgpu_merge_keyed_dev, against the oracle's
    device-order search of the unsharded index: ids and distance bits."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ivf_worker, args=(2, _free_port(), True, out), nprocs=2, join=True)
    _check_ivf(out, 2)


def test_deal_lists_is_balanced_and_deterministic():
    sys.path.insert(0, ROOT)
    from hnsw_clj_amd.sharded import deal_lists

    rng = np.random.default_rng(0)
    lens = rng.integers(0, 3000, 1024)
    for world in (1, 2, 3, 8):
        owner = deal_lists(lens, world)
        load = np.bincount(owner, weights=lens, minlength=world)
        assert load.max() - load.min() <= lens.max()
        assert np.array_equal(owner, deal_lists(lens.copy(), world))
    assert deal_lists([5, 5, 5, 5], 2).tolist() == [0, 1, 0, 1]        # ties: lower list first, lower rank first


def test_missing_merge_on_cpu_tensors_raises():
    """The product's merge is the HIP kernel; CPU tensors without an injected merge must not silently fall back."""
    sys.path.insert(0, ROOT)
    from hnsw_clj_amd.sharded import EngineOps, ShardedSearcher

    ids = torch.zeros((2, 3), dtype=torch.int32)
    d = torch.zeros((2, 3), dtype=torch.float32)
    s = ShardedSearcher(lambda Q, k: (ids, d), 0)            # no process group: the single-rank identity gather
    with pytest.raises(RuntimeError, match="HIP kernel"):
        s.search(torch.zeros((2, 4)), 3)
    with pytest.raises(RuntimeError, match="HIP kernel"):
        EngineOps.merge(ids[None], d[None], ids[None])
