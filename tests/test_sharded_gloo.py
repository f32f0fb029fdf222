"""World-size-2 test of the row-sharded search driver (hnsw-clj_amd/sharded.py) on CPU with gloo:
local top-k with global row ids -> all_gather -> merge == search over the unsharded index.
The local search and the merge are injected (the product's are HIP kernels); what is tested here is
the host logic of the N > 1 path: shard ranges, id offsetting, gather layout, tie order across shards."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def _worker(rank, world, port, n, dim, k, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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


def test_missing_merge_on_cpu_tensors_raises():
    """The product's merge is the HIP kernel; CPU tensors without an injected merge must not silently fall back."""
    sys.path.insert(0, ROOT)
    import inspect

    from hnsw_clj_amd import sharded

    assert "RuntimeError" in inspect.getsource(sharded.ShardedSearcher.search)
