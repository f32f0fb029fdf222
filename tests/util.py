"""Shared parity checks.  Tolerance is the one BASELINE.json's north_star states: identical top-k id
sets, distances within 1e-4 relative (plus 1e-6 absolute: f32 cosine of near-duplicates has an
absolute floor of ~1e-7)."""
import numpy as np

RTOL, ATOL = 1e-4, 1e-6


def close(a, b, scale=1.0):
    """|a - b| <= 1e-4 |b| + 1e-6 * scale.  scale = 1 for cosine distances (values in [0, 2]); for the
    un-normalised metrics (dot, L2) the absolute floor scales with |q||v| (resp. |q|+|v|): an f32 sum
    of D products cannot resolve a cancelled dot product better than eps * sum|q_i v_i|."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    with np.errstate(invalid="ignore"):
        both_inf = np.isinf(a) & np.isinf(b)
        return both_inf | (np.abs(a - b) <= RTOL * np.abs(b) + ATOL * scale)


def metric_scale(metric, q, base):
    """Absolute-tolerance scale of a metric for a query against a base (see close())."""
    if metric in (0, "cosine", "cos"):
        return 1.0
    qn = float(np.linalg.norm(np.asarray(q, np.float64), axis=-1).max())
    bn = float(np.linalg.norm(np.asarray(base, np.float64), axis=-1).max())
    return qn * bn if metric in (2, "dot") else qn + bn


def assert_topk_parity(gpu_ids, gpu_d, ora_ids, ora_d, what="", scale=1.0):
    """ids: set-equal per query, except swaps among candidates whose ORACLE distance ties the k-th
    boundary within tolerance; distances: rank-wise within tolerance."""
    gpu_ids, ora_ids = np.asarray(gpu_ids), np.asarray(ora_ids)
    assert gpu_ids.shape == ora_ids.shape, (what, gpu_ids.shape, ora_ids.shape)
    ok = close(gpu_d, ora_d, scale)
    assert ok.all(), "%s: distances differ beyond tolerance at %s" % (what, np.argwhere(~ok)[:5])
    for q in range(len(gpu_ids)):
        g, o = set(gpu_ids[q][gpu_ids[q] >= 0].tolist()), set(ora_ids[q][ora_ids[q] >= 0].tolist())
        assert len(g) == len(o), "%s q%d: result counts differ" % (what, q)
        if g == o:
            continue
        kth = np.asarray(ora_d[q], np.float64)[len(o) - 1]
        for i in o - g:  # an oracle id the GPU dropped must sit on the boundary
            d = float(np.asarray(ora_d[q])[list(ora_ids[q]).index(i)])
            assert abs(d - kth) <= RTOL * abs(kth) + ATOL * scale, "%s q%d: id %d missing (d=%g, kth=%g)" % (what, q, i, d, kth)
        for i in g - o:  # a GPU-only id must be a boundary tie as well
            d = float(np.asarray(gpu_d[q])[list(gpu_ids[q]).index(i)])
            assert abs(d - kth) <= RTOL * abs(kth) + ATOL * scale, "%s q%d: extra id %d (d=%g, kth=%g)" % (what, q, i, d, kth)


def assert_exact(gpu_ids, gpu_d, ora_ids, ora_d, what=""):
    """Against the oracle's device-order mode: ids identical, distances bit-identical."""
    np.testing.assert_array_equal(np.asarray(gpu_ids), np.asarray(ora_ids), err_msg=what + ": ids")
    g = np.asarray(gpu_d, np.float32)
    o = np.asarray(ora_d, np.float64).astype(np.float32)
    np.testing.assert_array_equal(g.view(np.uint32), o.view(np.uint32), err_msg=what + ": distance bits")
