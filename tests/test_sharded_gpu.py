"""GPU tests of the N > 1 data path on ONE GPU: the shards are separate index handles, the all-gather is a
torch.stack in rank order, everything else is the product's code (hnswgpu_set_ivf_shard,
hnswgpu_ivf_search_shard_dev, hnswgpu_merge_keyed_dev, hnswgpu_merge_topk_dev, sharded.deal_lists,
sharded.ShardedIVF).  The bar: sharded == unsharded == oracle -- ids and distance bits.

Reference: scatter / per-partition top-k / gather / sort / take k, src/hnsw/ann/partition/partitioned_hnsw.clj:149-196;
probed-list scan and merge, src/hnsw/ann/partition/ivf_flat.clj:261-294."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sharded_util import cut_into_shards  # noqa: E402
from util import assert_exact  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(native_lib):
    from hnsw_clj_amd import engine

    assert engine.device_count() >= 1, "no GPU visible"
    return engine


def _ivf_mode(O, metric, nq, nprobe, nlist):
    return O.MODE_MFMA if (metric != O.L2 and nq * min(nprobe, nlist) > 12 * nlist) else O.MODE_DEV


def _search_shards(eng, shards, cen, lens, metric, Qt, k, nprobe):
    import torch

    parts = []
    for rows, gid, loff in shards:
        with eng.Index(rows, metric) as sh:
            sh.set_ivf_shard(cen, loff, np.arange(len(rows), dtype=np.int32), lens)
            ids, d, order = sh.ivf_search_shard_dev(Qt, k, nprobe)
            g = torch.from_numpy(gid).to(Qt.device)
            gids = torch.where(ids >= 0, g[ids.clamp(min=0).long()], ids) if len(gid) else ids
            parts.append((gids, d, order))
            torch.cuda.synchronize()
    return eng.merge_keyed_dev(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]),
                               torch.stack([p[2] for p in parts]))


@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
def test_ivf_shards_equal_unsharded_and_oracle(eng, oracle, metric):
    """200k x 128, 256 lists: the index cut into 2 and 8 shards by whole lists.  Small batch (GEMV scan) and large batch
    (MFMA tile scan / L2 group scan): merged ids and distance bits == the unsharded hnswgpu_ivf_search == the oracle."""
    import torch

    O = oracle
    m = O.METRICS[metric]
    n, dim, nlist, nprobe, k = 200_000, 128, 256, 8, 10
    base = O.generate_dataset(n, dim, "clustered", num_clusters=40, noise_level=0.6).astype(np.float32)
    Q = O.generate_dataset(600, dim, "clustered", num_clusters=40, noise_level=0.6, seed=43).astype(np.float32)
    # exact ties in different lists (and so, mostly, in different shards): coordinate 0 = 9.0 in rows all over the index,
    # query 1 = 4 e_0 (dot: -36.0 for each of them exactly, whatever the summation order); query 0 = a duplicated row
    base[np.arange(40) * 4999 + 3, 0] = 9.0
    base[5000:5006] = base[123]
    Q[0] = base[123]
    Q[1] = 0.0
    Q[1, 0] = 4.0
    with eng.Index(base, metric) as idx:
        idx.ivf_build(nlist, 3, 42)
        cen, off, lids = idx.get_ivf()
        for nshard in (2, 8):
            shards, lens, owner = cut_into_shards(base, cen, off, lids, nshard)
            assert sum(len(s[0]) for s in shards) == n and len(np.unique(owner)) == nshard
            for nq in (5, 600):
                Qt = torch.from_numpy(Q[:nq]).cuda()
                ui, ud = idx.ivf_search_dev(Qt, k, nprobe)
                mi, md = _search_shards(eng, shards, cen, lens, metric, Qt, k, nprobe)
                torch.cuda.synchronize()
                what = "%s %d shards nq=%d" % (metric, nshard, nq)
                np.testing.assert_array_equal(mi.cpu().numpy(), ui.cpu().numpy(), err_msg=what + ": ids vs unsharded")
                np.testing.assert_array_equal(md.cpu().numpy().view(np.uint32), ud.cpu().numpy().view(np.uint32),
                                              err_msg=what + ": distance bits vs unsharded")
                if nshard == 2:
                    oi, od, _ = O.ivf_search(base, cen, off, lids, Q[:nq], k, nprobe, metric=m,
                                             mode=_ivf_mode(O, m, nq, nprobe, nlist))
                    assert_exact(mi.cpu().numpy(), md.cpu().numpy(), oi, od, what + " vs oracle")
        # the order column of an ordinary index is the position in its own candidate stream: strictly increasing among ties
        Qt = torch.from_numpy(Q[:5]).cuda()
        ids, d, order = idx.ivf_search_shard_dev(Qt, k, nprobe)
        d, order = d.cpu().numpy(), order.cpu().numpy().view(np.uint32)
        for q in range(5):
            for j in range(1, k):
                if d[q, j] == d[q, j - 1]:
                    assert order[q, j] > order[q, j - 1]


def test_sharded_ivf_class_single_rank_equals_ivf_build(eng, oracle):
    """ShardedIVF with one rank (no process group) runs the distributed build's own code path -- seeding, Lloyd with the
    f64 list sums, dealing, the (identity) exchange -- and must reproduce hnswgpu_ivf_build bit for bit: centroids, lists,
    search results."""
    import torch

    from hnsw_clj_amd.sharded import ShardedIVF, lists_from_assign

    O = oracle
    n, dim, nlist = 30_000, 96, 50
    base = O.generate_dataset(n, dim, "clustered", num_clusters=12, noise_level=0.5).astype(np.float32)
    Q = torch.from_numpy(O.generate_dataset(64, dim, seed=43).astype(np.float32)).cuda()
    x = torch.from_numpy(base).cuda()
    sh = ShardedIVF.build(x, "cosine", nlist, 5, 42)
    with eng.Index(base, "cosine") as idx:
        idx.ivf_build(nlist, 5, 42)
        cen, off, lids = idx.get_ivf()
        np.testing.assert_array_equal(sh.centroids.view(np.uint32), cen.view(np.uint32))
        soff, sids = lists_from_assign(sh.assign, nlist)
        np.testing.assert_array_equal(soff, off)
        np.testing.assert_array_equal(sids, lids)
        for nq in (3, 64):
            a = sh.search(Q[:nq], 10, 6)
            b = idx.ivf_search_dev(Q[:nq], 10, 6)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(a[0].cpu().numpy(), b[0].cpu().numpy())
            np.testing.assert_array_equal(a[1].cpu().numpy().view(np.uint32), b[1].cpu().numpy().view(np.uint32))
    sh.close()


def test_set_ivf_shard_rejects_bad_lengths(eng, oracle, tmp_path):
    base = oracle.generate_dataset(100, 8).astype(np.float32)
    cen = base[:4].copy()
    off = np.array([0, 25, 50, 75, 100], np.int64)
    with eng.Index(base) as idx:
        with pytest.raises(Exception, match="global length"):
            idx.set_ivf_shard(cen, off, np.arange(100, dtype=np.int32), np.array([25, 24, 25, 25], np.int64))
        idx.set_ivf_shard(cen, off, np.arange(100, dtype=np.int32), np.array([25, 30, 25, 1000], np.int64))
        ids, d = idx.ivf_search(base[:3], 5, 4)
        assert (ids[:, 0] == np.arange(3)).all()
        # the file format does not carry the whole index's list lengths: a shard refuses to be saved as if it were one
        with pytest.raises(Exception, match="shard"):
            idx.save(str(tmp_path / "shard.bin"))


@pytest.mark.parametrize("metric", ["cosine", "l2"])
def test_hnsw_subgraphs_equal_oracle_merge(eng, oracle, metric):
    """configs[4]'s split at test size: one HNSW sub-graph per shard over contiguous row ranges (=
    PartitionedHNSWIndex), each searched with the full k, hnswgpu_merge_topk_dev over the stacked per-shard results
    == the oracle searching every sub-graph (same graphs) and merging by a stable sort of the concatenation."""
    import torch

    from hnsw_clj_amd.sharded import shard_range

    O = oracle
    m = O.METRICS[metric]
    n, dim, k, ef = 24_000, 64, 10, 80
    base = O.generate_dataset(n, dim).astype(np.float32)
    base[13000] = base[100]                                         # a cross-shard exact tie
    Q = O.generate_dataset(200, dim, seed=43).astype(np.float32)
    Q[0] = base[100]
    Qt = torch.from_numpy(Q).cuda()
    for nshard in (2, 8):
        gi, gd, oi, od = [], [], [], []
        for s in range(nshard):
            lo, hi = shard_range(n, s, nshard)
            with eng.Index(base[lo:hi], metric) as idx:
                idx.hnsw_build(8, 60, 42 + s)
                g = idx.get_graph()
                ids, d = idx.hnsw_search_dev(Qt, k, ef)
                gi.append(torch.where(ids >= 0, ids + lo, ids))
                gd.append(d)
                torch.cuda.synchronize()
            og = O.Graph(g.levels, g.l0_adj, g.up_off, g.up_adj, g.M, g.entry, g.max_level)
            a, b, _, _ = O.hnsw_search(base[lo:hi], og, Q, k, ef=ef, metric=m, mode=O.MODE_DEV)
            oi.append(np.where(a >= 0, a + lo, a))
            od.append(b.astype(np.float32))
        mi, md = eng.merge_topk_dev(torch.stack(gi), torch.stack(gd))
        torch.cuda.synchronize()
        ci, cd = np.concatenate(oi, axis=1), np.concatenate(od, axis=1)          # [nq, nshard * k], shard-major
        key = np.where(ci >= 0, cd, np.inf)
        sel = np.argsort(key, axis=1, kind="stable")[:, :k]                       # Collections/sort: stable
        ei, ed = np.take_along_axis(ci, sel, 1), np.take_along_axis(cd, sel, 1)
        assert_exact(mi.cpu().numpy(), md.cpu().numpy(), ei, ed, "%s hnsw sub-graphs, %d shards" % (metric, nshard))
        row = mi.cpu().numpy()[0].tolist()
        if nshard == 2 and 100 in row and 13000 in row:                           # both copies found by the sub-graph searches:
            assert row.index(13000) == row.index(100) + 1                         # equal distances, the lower shard first
        if nshard == 2 and metric == "cosine":
            assert row[:2] == [100, 13000]
