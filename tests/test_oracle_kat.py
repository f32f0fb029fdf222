"""Pins the CPU oracle (oracle/oracle.c) against every numeric known-answer test the reference's
own test suite holds for this path, against the JDK's java.util.Random, and against the committed
golden fixtures.  No GPU."""
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))


# ---- distance KATs: test/hnsw/core_test.clj:9-31 ----------------------------------------------------
def test_euclidean_kats_core_test(oracle):
    O = oracle
    assert O.euclidean_distance([1, 2, 3], [1, 2, 3]) == 0.0          # (is (= 0.0 ...)) :16
    assert O.euclidean_distance([0, 0], [3, 4]) == 5.0                # (is (= 5.0 ...)) :17
    assert abs(O.euclidean_distance([1, 2, 3], [4, 5, 6]) - 5.196152422706632) < 0.00001  # :18-20


def test_cosine_kats_core_test(oracle):
    O = oracle
    for f in (O.cosine_distance, O.cosine_distance_ultra):
        assert f([1, 2, 3], [1, 2, 3]) < 0.001                        # :29
        assert abs(f([1, 0], [-1, 0]) - 2.0) < 0.001                  # :30
        assert abs(f([1, 2, 3], [4, 5, 6]) - 0.0253) < 0.01           # :31
        assert f([1, 2, 3], [4, 5, 6]) == 0.025368153802923787        # exact f64 value of that formula


# ---- test/simple_test.clj:33-41 and test-functional.sh:53-71 ----------------------------------------
def test_kats_simple_test_and_shell(oracle):
    O = oracle
    assert abs(O.euclidean_distance([1.0, 2.0, 3.0], [4.0, 5.0, 6.0]) - 5.196152) < 0.001
    assert O.euclidean_distance([1.0, 2.0, 3.0], [1.0, 2.0, 3.0]) == 0.0
    assert O.cosine_distance([1.0, 2.0, 3.0], [1.0, 2.0, 3.0]) < 0.001


# ---- test/hnsw/graph_test.clj:11-22 -------------------------------------------------------------------
def test_kats_graph_test(oracle):
    O = oracle
    assert abs(O.euclidean_distance([1, 2, 3], [1, 2, 3])) < 0.001
    assert 1.73 < O.euclidean_distance([1, 2, 3], [2, 3, 4]) < 1.74
    assert abs(O.cosine_distance([1, 0], [1, 0])) < 0.001
    assert abs(O.cosine_distance([1, 0], [0, 1]) - 1.0) < 0.001


def test_zero_norm_guards(oracle):
    O = oracle
    assert O.cosine_distance_ultra([0, 0, 0], [1, 2, 3]) == 1.0   # ultra_fast.clj:92-95
    assert O.cosine_distance([0, 0, 0], [1, 2, 3]) == 1.0         # simd.clj:144-147
    assert O.dot_product([1, 2, 3], [4, 5, 6]) == 32.0


def test_summation_is_left_to_right(oracle):
    """A.1: strictly sequential f64 sums -- not pairwise (np.dot) order."""
    rs = np.random.RandomState(0)
    a, b = rs.randn(768), rs.randn(768)
    dot = 0.0
    for x, y in zip(a, b):
        dot = dot + x * y
    assert oracle.dot_product(a, b) == dot


# ---- java.util.Random (JDK specification) ------------------------------------------------------------
def test_java_random_known_answers(oracle):
    O = oracle
    assert O.JavaRandom(42).next_int() == -1170105035            # new Random(42).nextInt()
    assert [O.JavaRandom(42).next_int(10)] == [0]
    r = O.JavaRandom(42)
    assert [r.next_int(10) for _ in range(5)] == [0, 3, 8, 4, 0]
    r = O.JavaRandom(42)
    assert [r.next_gaussian() for _ in range(4)] == [1.1419053154730547, 0.9194079489827879,
                                                    -0.9498666368908959, -1.1069902863993377]
    assert O.JavaRandom(0).next_double() == 0.730967787376657    # new Random(0).nextDouble()


def test_fdlibm_log_close_to_libm(oracle):
    xs = np.random.RandomState(1).rand(20000)
    got = np.array([oracle.lib().orc_fdlibm_log(float(x)) for x in xs])
    ulp = np.abs(got.view(np.int64) - np.log(xs).view(np.int64))
    assert ulp.max() <= 1


@pytest.mark.parametrize("dist", ["gaussian", "uniform", "unit", "clustered"])
def test_generator_two_implementations_agree(oracle, dist):
    """oracle/oracle.c (scalar C) vs hnsw-clj_amd/datagen.py (vectorised numpy): bit-equal f64."""
    from hnsw_clj_amd import datagen

    for n, d in [(7, 5), (64, 33), (300, 128)]:
        a = datagen.generate_dataset(n, d, dist, num_clusters=3, dtype=np.float64)
        b = oracle.generate_dataset(n, d, dist, num_clusters=3)
        assert np.array_equal(a, b), (dist, n, d)


def test_generator_semantics(oracle):
    g = oracle.generate_dataset(3, 4)
    assert g[0, 0] == 1.1419053154730547           # row-major nextGaussian order (data_generator.clj:28-31)
    u = oracle.generate_dataset(100, 8, "unit")
    assert np.allclose(np.linalg.norm(u, axis=1), 1.0)
    un = oracle.generate_dataset(100, 8, "uniform")
    assert un.min() >= -1 and un.max() < 1


# ---- index behaviour: test/hnsw/core_test.clj:33-121, test/simple_test.clj:21-58 ----------------------
def test_ultra_fast_index_behaviour(oracle):
    O = oracle
    base = O.generate_dataset(100, 128).astype(np.float32)   # (gen/generate-dataset 100 128) :35
    g = O.hnsw_build(base)
    ids, d, _, _ = O.hnsw_search(base, g, base[0], 5)
    assert (ids[0] >= 0).sum() == 5 and d[0, 0] < 0.01       # :41-47
    assert ids[0, 0] == 0
    assert np.all(np.diff(d[0]) >= 0)                        # ascending


def test_empty_and_single_and_k_gt_n(oracle):
    O = oracle
    empty = O.Graph(np.zeros(0, np.int32), np.zeros((0, 32), np.int32), np.zeros(1, np.int64),
                    np.zeros(0, np.int32), 16, -1, 0)
    ids, d, _, _ = O.hnsw_search(np.zeros((0, 3), np.float32), empty, [1, 2, 3], 5)
    assert (ids == -1).all()                                 # core_test.clj:63-68
    one = np.array([[1.0, 2.0, 3.0, 4.0]], np.float32)
    g = O.hnsw_build(one)
    ids, d, _, _ = O.hnsw_search(one, g, one[0], 1)
    assert ids[0, 0] == 0 and d[0, 0] < 0.001                # :70-78
    five = O.generate_dataset(5, 64).astype(np.float32)
    g = O.hnsw_build(five)
    ids, d, _, _ = O.hnsw_search(five, g, five[0], 10)
    assert (ids[0] >= 0).sum() == 5                          # :90-96


def test_recall_on_1000x128(oracle):
    """integration_test.clj:138-157 intent: recall >= 0.8 @k=10 on 1000x128."""
    O = oracle
    base = O.generate_dataset(1000, 128).astype(np.float32)
    Q = O.generate_dataset(40, 128, seed=43).astype(np.float32)
    g = O.hnsw_build(base)
    ids, _, _, _ = O.hnsw_search(base, g, Q, 10)
    ex, _, _ = O.exact_knn(base, Q, 10)
    assert O.recall(ids, ex) >= 0.8


def test_device_order_mode_agrees_with_f64(oracle):
    """The f32 device-order mimic stays within the north-star tolerance of the f64 truth."""
    from util import assert_topk_parity, metric_scale

    O = oracle
    base = O.generate_dataset(800, 96).astype(np.float32)
    Q = O.generate_dataset(24, 96, seed=43).astype(np.float32)
    for metric in (O.COSINE, O.L2, O.DOT):
        g = O.hnsw_build(base, metric, M=8, ef_construction=64)
        a = O.hnsw_search(base, g, Q, 10, metric=metric)
        b = O.hnsw_search(base, g, Q, 10, metric=metric, mode=O.MODE_DEV)
        sc = metric_scale(metric, Q, base)
        assert_topk_parity(b[0], b[1], a[0], a[1], "hnsw metric %d" % metric, sc)
        ea = O.exact_knn(base, Q, 10, metric=metric)
        eb = O.exact_knn(base, Q, 10, metric=metric, mode=O.MODE_DEV)
        assert_topk_parity(eb[0], eb[1], ea[0], ea[1], "exact metric %d" % metric, sc)


def test_kmeans_semantics(oracle):
    O = oracle
    base = O.generate_dataset(500, 32, "clustered", num_clusters=6, noise_level=0.2).astype(np.float32)
    ch = O.kmeanspp(base, 8)
    assert ch[0] == O.JavaRandom(42).next_int(500)           # first centroid = Random(42).nextInt(n) :39
    assert len(set(ch.tolist())) == 8
    cen, assign = O.ivf_build(base, 8, 10)
    a2, d2 = O.kmeans_assign(base, cen)
    assert np.array_equal(assign, a2)
    # strict <: ties go to the lowest index (:86-89)
    dup = np.vstack([cen[:1], cen[:1], cen[1:3]])
    a3, _ = O.kmeans_assign(base[:50], dup)
    assert not np.any(a3 == 1)
    # ivf search with every list probed == exact kNN
    off, lids = O.lists_from_assign(assign, 8)
    Q = base[:10]
    ii, dd, _ = O.ivf_search(base, cen.astype(np.float32), off, lids, Q, 5, 8)
    ex, exd, _ = O.exact_knn(base, Q, 5)
    assert np.array_equal(np.sort(ii, 1), np.sort(ex, 1))


# ---- the reference's float32 Vector-API forms ------------------------------------
@pytest.mark.parametrize("lanes", [4, 8, 16])
def test_f32_vector_forms(oracle, lanes):
    """simd.clj:26-115 (chunked: f32 lane products, f32 lane reduce, f64 accumulate, f64 tail) and wip/vector.clj:21-86
    (f32 lane accumulators, one reduce): the reference's own KATs (exact on small integers, whatever the lane count:
    the 2- and 3-element vectors are all tail), and on random data every form within f32 round-off of the f64 form --
    the spread that makes north_star's tolerance 1e-4 and not bitwise."""
    O = oracle
    for f in (O.f32_vector, O.f32_lane_accumulate):
        assert f(O.L2, [1, 2, 3], [1, 2, 3], lanes) == 0.0                     # core_test.clj:9-31
        assert f(O.L2, [0, 0], [3, 4], lanes) == 5.0
        assert abs(f(O.L2, [1, 2, 3], [4, 5, 6], lanes) - 5.196152422706632) < 1e-12
        assert abs(f(O.COSINE, [1, 2, 3], [4, 5, 6], lanes) - 0.025368153802923787) < 1e-12
        assert f(O.COSINE, [1, 0], [-1, 0], lanes) == 2.0 and f(O.COSINE, [1, 0], [0, 1], lanes) == 1.0
        assert f(O.COSINE, [0, 0, 0], [1, 2, 3], lanes) == 1.0                 # both zero guards give 1.0 here
        assert f(O.DOT, [1, 2, 3], [4, 5, 6], lanes) == 32.0
        # integers are exact in every association: 40 elements = full chunks + (for 16 lanes) a tail of 8
        a, b = np.arange(40) % 7 - 3, np.arange(40) % 5 - 2
        for assoc in (0, 1):
            assert f(O.DOT, a, b, lanes, assoc) == float(np.dot(a, b))
            assert f(O.L2, a, b, lanes, assoc) == float(np.sqrt(((a - b) ** 2).sum()))
    rng = np.random.default_rng(lanes)
    for dim in (5, 100, 768, 1536):
        a = rng.standard_normal(dim).astype(np.float32)
        b = (0.5 * a + rng.standard_normal(dim)).astype(np.float32)
        scale = float(np.linalg.norm(a.astype(np.float64)) * np.linalg.norm(b.astype(np.float64)))
        for m, ref, tol in ((O.COSINE, O.distance(O.COSINE, a, b), 4e-6), (O.L2, O.distance(O.L2, a, b), 4e-6),
                            (O.DOT, O.dot_product(a.astype(np.float64), b.astype(np.float64)), 4e-6 * scale)):
            vals = [f(m, a, b, lanes, s) for f in (O.f32_vector, O.f32_lane_accumulate) for s in (0, 1)]
            assert all(abs(v - ref) <= tol * max(1.0, abs(ref)) for v in vals), (dim, m, ref, vals)
            if dim >= 768 and m == O.COSINE:
                assert len(set(vals)) > 1, "the forms are meant to differ in the last bits"


# ---- committed golden fixtures -----------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g256x64", "c1000x128"])
def test_oracle_reproduces_golden(oracle, name):
    import make_golden

    O = oracle
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    base, Q = make_golden.inputs(name)
    for mname, metric in (("cos", O.COSINE), ("l2", O.L2), ("dot", O.DOT)):
        g = O.hnsw_build(base, metric, M=8, ef_construction=64, seed=42)
        assert np.array_equal(g.l0_adj, gold[mname + "_l0"]) and g.entry == int(gold[mname + "_entry"])
        ids, d, st, _ = O.hnsw_search(base, g, Q, 10, ef=50, metric=metric)
        assert np.array_equal(ids, gold[mname + "_hnsw_ids"]) and np.array_equal(d, gold[mname + "_hnsw_d"])
        assert np.array_equal(st, gold[mname + "_hnsw_stats"])
        ex, exd, _ = O.exact_knn(base, Q, 10, metric=metric)
        assert np.array_equal(ex, gold[mname + "_exact_ids"]) and np.array_equal(exd, gold[mname + "_exact_d"])
    assert np.array_equal(O.kmeanspp(base, 16), gold["ivf_kpp"])
    cen, assign = O.ivf_build(base, 16, 10)
    assert np.array_equal(assign, gold["ivf_assign"]) and np.array_equal(cen.astype(np.float32), gold["ivf_cent"])
