"""GPU tests of the multi-GPU group behind the C ABI (include/hnswgpu.h: hnswgpu_group_*) on ONE GPU: `devices` names
GPU 0 several times, so the group holds several engine handles on the card -- the per-device searches on their own
streams, the peer copies of the partial lists and the merge on the first device are the product's code as it runs on
an 8-GPU node.  The bar: group == unsharded == oracle, ids and distance bits, tie order included.

Reference: scatter / per-partition top-k / gather / sort / take k, src/hnsw/ann/partition/partitioned_hnsw.clj:149-196;
probed-list scan and merge, src/hnsw/ann/partition/ivf_flat.clj:261-294."""
import os
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from util import assert_exact  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(native_lib):
    from hnsw_clj_amd import engine

    assert engine.device_count() >= 1, "no GPU visible"
    return engine


@pytest.mark.parametrize("metric", ["cosine", "l2", "dot"])
@pytest.mark.parametrize("ndev", [1, 3, 8])
def test_group_ivf_equals_unsharded_and_oracle(eng, oracle, metric, ndev):
    """ONE IVF-FLAT index over `ndev` handles: lists dealt by row count, every handle scans the probed lists it holds,
    partial top-k lists merged on the first device by (distance, position in the whole index's candidate stream)."""
    O = oracle
    m = O.METRICS[metric]
    n, dim, nlist, nprobe, k = 30_000, 96, 64, 6, 10
    base = O.generate_dataset(n, dim, "clustered", num_clusters=20, noise_level=0.5).astype(np.float32)
    base[5000:5040] = base[3]                 # exact ties that land in different lists' neighbourhoods
    base[20000:20020, 1:] = base[3, 1:]
    Q = np.concatenate([O.generate_dataset(70, dim, "clustered", num_clusters=20, noise_level=0.5, seed=43), base[3:4]]).astype(np.float32)
    with eng.Index(base, metric) as full:
        full.ivf_build(nlist, 4, 42)
        cen, off, lids = full.get_ivf()
        with eng.Group([0] * ndev, dim, metric) as g:
            g.set_ivf(base, cen, off, lids)
            info = g.info()
            assert info["kind"] == "ivf" and info["n"] == n and info["rows_per_device"].sum() == n
            if ndev > 1:
                assert info["rows_per_device"].max() - info["rows_per_device"].min() <= np.diff(off).max()   # balanced by rows
            for nq in (1, 9, len(Q)):
                gi, gd = g.ivf_search(Q[:nq], k, nprobe)
                ui, ud = full.ivf_search(Q[:nq], k, nprobe)
                np.testing.assert_array_equal(gi, ui)
                np.testing.assert_array_equal(gd.view(np.uint32), ud.view(np.uint32))
                mode = O.MODE_MFMA if (m != O.L2 and nq * nprobe > 12 * nlist) else O.MODE_DEV
                oi, od, _ = O.ivf_search(base, cen, off, lids, Q[:nq], k, nprobe, metric=m, mode=mode)
                assert_exact(gi, gd, oi, od, "group ivf %s ndev=%d nq=%d" % (metric, ndev, nq))
            with pytest.raises(Exception, match="HNSW"):
                g.hnsw_search(Q[:2], k, 50)


@pytest.mark.parametrize("metric", ["cosine", "l2"])
def test_group_ivf_with_the_home_list_pass_on_shards(eng, oracle, metric, tune):
    """Large batches send every row of a query's NEAREST list through the matrix cores in half precision and take the query's
    threshold from there (stream_kernels.hpp, step 1a) -- on a shard the nearest list of most queries lives elsewhere: those
    keep the sampled threshold, their home pair has no rows here, and the shard's answer must still merge into the unsharded
    one bit for bit.  Forced on for a batch of 150 through the tuning table (production: from 512 queries)."""
    O = oracle
    m = O.METRICS[metric]
    n, dim, nlist, nprobe, k = 30_000, 128, 64, 6, 10
    base = O.generate_dataset(n, dim, "clustered", num_clusters=20, noise_level=0.5).astype(np.float32)
    base[5000:5040] = base[3]
    Q = np.concatenate([O.generate_dataset(149, dim, "clustered", num_clusters=20, noise_level=0.5, seed=43), base[3:4]]).astype(np.float32)
    for key, v in (("TILE_PAIRS", 1 << 40), ("IVF_CODES", 1), ("STREAM_MID", 1), ("FINISH_ORDER", 1), ("STREAM_HOME", 1), ("ROUTE_MFMA", 1)):
        tune.set(key, str(v))
    with eng.Index(base, metric) as full:
        full.set_rejection_test(2)
        full.ivf_build(nlist, 4, 42)
        cen, off, lids = full.get_ivf()
        oi, od, _ = O.ivf_search(base, cen, off, lids, Q, k, nprobe, metric=m, mode=O.MODE_DEV)
        ui, ud = full.ivf_search(Q, k, nprobe)
        assert_exact(ui, ud, oi, od, "unsharded, home-list pass forced, %s" % metric)
        for ndev in (2, 5):
            with eng.Group([0] * ndev, dim, metric) as g:
                g.set_ivf(base, cen, off, lids)
                gi, gd = g.ivf_search(Q, k, nprobe)
                assert_exact(gi, gd, oi, od, "group of %d, home-list pass forced, %s" % (ndev, metric))


def test_group_shards_take_the_path_of_the_whole_index(eng, oracle):
    """A shard holds n / ndev rows over the SAME nlist: what depends on the mean list length (the largest k the survivor
    stream serves, hence -- past the tile boundary -- the summation order) and the first-search calibration verdict must be
    decided for the WHOLE index, or shards and unsharded handle compute different distance bits.  40,000 rows
    in 8 lists (mean 5,000: the stream serves k <= 64), k = 100, 64 queries x 4 probes (32 pairs per list: past the
    boundary of 12): the unsharded handle takes the f32 MFMA tile scan, and so must four shards of 10,000 rows each
    (whose own mean of 1,250 would let the stream serve k = 256)."""
    O = oracle
    n, dim, nlist, nprobe, k = 40_000, 136, 8, 4, 100
    base = O.generate_dataset(n, dim, "clustered", num_clusters=8, noise_level=0.5).astype(np.float32)
    Q = O.generate_dataset(64, dim, "clustered", num_clusters=8, noise_level=0.5, seed=43).astype(np.float32)
    with eng.Index(base, "cosine") as full:
        full.ivf_build(nlist, 4, 42)
        cen, off, lids = full.get_ivf()
        assert np.diff(off).mean() >= 4096
        ui, ud = full.ivf_search(Q, k, nprobe)
        oi, od, _ = O.ivf_search(base, cen, off, lids, Q, k, nprobe, mode=O.MODE_MFMA)
        assert_exact(ui, ud, oi, od, "unsharded, k beyond the stream's range: tile scan")
        for _ in range(1):
            with eng.Group([0] * 4, dim, "cosine") as g:
                g.set_ivf(base, cen, off, lids)
                gi, gd = g.ivf_search(Q, k, nprobe)
                np.testing.assert_array_equal(gi, ui)
                np.testing.assert_array_equal(gd.view(np.uint32), ud.view(np.uint32))
        # the verdict API: measuring on a member reports, installing overrides without measuring
        full.set_rejection_test(1)
        assert full.ivf_stream_state() in (0, 1)
        full.ivf_set_stream_state(1)
        assert full.ivf_stream_state() == 1
        full.set_rejection_test(2)


@pytest.mark.parametrize("metric", ["cosine", "l2"])
def test_group_hnsw_equals_oracle_merge(eng, oracle, metric):
    """One HNSW sub-graph per handle over contiguous row ranges, each searched with the full k, merged by distance with
    ties to the lower handle: equal to the oracle searching every exported sub-graph and stable-sorting the
    concatenation (partitioned_hnsw.clj:171-196)."""
    O = oracle
    m = O.METRICS[metric]
    n, dim, ndev, k, ef = 9000, 64, 3, 10, 80
    base = O.generate_dataset(n, dim).astype(np.float32)
    base[7000:7010] = base[11]                # ties across sub-graphs
    Q = np.concatenate([O.generate_dataset(40, dim, seed=43), base[11:12]]).astype(np.float32)
    with eng.Group([0] * ndev, dim, metric) as g:
        g.hnsw_build(base, 16, 100, 42)
        info = g.info()
        assert info["kind"] == "hnsw" and list(info["rows_per_device"]) == [3000, 3000, 3000]
        gi, gd = g.hnsw_search(Q, k, ef)
        parts_i, parts_d = [], []
        for r in range(ndev):
            r0 = n * r // ndev
            r1 = n * (r + 1) // ndev
            gr = g.member_graph(r)
            oi, od, _, _ = O.hnsw_search(base[r0:r1], gr, Q, k, ef=ef, metric=m, mode=O.MODE_DEV)
            parts_i.append(np.where(oi >= 0, oi + r0, oi))
            parts_d.append(od.astype(np.float32))
        ci, cd = np.concatenate(parts_i, axis=1), np.concatenate(parts_d, axis=1)
        cd = np.where(ci >= 0, cd, np.inf)
        order = np.argsort(cd, axis=1, kind="stable")[:, :k]
        wi, wd = np.take_along_axis(ci, order, 1), np.take_along_axis(cd, order, 1)
        np.testing.assert_array_equal(gi, wi)
        np.testing.assert_array_equal(gd.view(np.uint32), wd.view(np.uint32))


def test_group_from_plain_c(native_lib, tmp_path):
    """examples/group_demo.c: the group API used the way a JNI / Panama binding uses it -- from C, without Python in
    the process: an IVF index built on one handle, served by a group of four, the two answers compared bit for bit."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "group_demo")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "group_demo.c"),
                           "-L" + native_lib.PKG, "-lhnswgpu", "-Wl,-rpath," + native_lib.PKG, "-lm", "-o", exe])
    out = subprocess.run([exe, "4"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "group_demo ok" in out.stdout
