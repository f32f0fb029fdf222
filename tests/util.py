"""Shared parity checks.  Tolerance is the one BASELINE.json's north_star states: identical top-k id
sets, distances within 1e-4 relative (plus 1e-6 absolute: f32 cosine of near-duplicates has an
absolute floor of ~1e-7)."""
import numpy as np

RTOL, ATOL = 1e-4, 1e-6


def close(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    both_inf = np.isinf(a) & np.isinf(b)
    return both_inf | (np.abs(a - b) <= RTOL * np.abs(b) + ATOL)


def assert_topk_parity(gpu_ids, gpu_d, ora_ids, ora_d, what=""):
    """ids: set-equal per query, except swaps among candidates whose ORACLE distance ties the k-th
    boundary within tolerance; distances: rank-wise within tolerance."""
    gpu_ids, ora_ids = np.asarray(gpu_ids), np.asarray(ora_ids)
    assert gpu_ids.shape == ora_ids.shape, (what, gpu_ids.shape, ora_ids.shape)
    ok = close(gpu_d, ora_d)
    assert ok.all(), "%s: distances differ beyond tolerance at %s" % (what, np.argwhere(~ok)[:5])
    for q in range(len(gpu_ids)):
        g, o = set(gpu_ids[q][gpu_ids[q] >= 0].tolist()), set(ora_ids[q][ora_ids[q] >= 0].tolist())
        assert len(g) == len(o), "%s q%d: result counts differ" % (what, q)
        if g == o:
            continue
        kth = np.asarray(ora_d[q], np.float64)[len(o) - 1]
        for i in o - g:  # an oracle id the GPU dropped must sit on the boundary
            d = float(np.asarray(ora_d[q])[list(ora_ids[q]).index(i)])
            assert abs(d - kth) <= RTOL * abs(kth) + ATOL, "%s q%d: id %d missing (d=%g, kth=%g)" % (what, q, i, d, kth)
        for i in g - o:  # a GPU-only id must be a boundary tie as well
            d = float(np.asarray(gpu_d[q])[list(gpu_ids[q]).index(i)])
            assert abs(d - kth) <= RTOL * abs(kth) + ATOL, "%s q%d: extra id %d (d=%g, kth=%g)" % (what, q, i, d, kth)


def assert_exact(gpu_ids, gpu_d, ora_ids, ora_d, what=""):
    """Against the oracle's device-order mode: ids identical, distances bit-identical."""
    np.testing.assert_array_equal(np.asarray(gpu_ids), np.asarray(ora_ids), err_msg=what + ": ids")
    g = np.asarray(gpu_d, np.float32)
    o = np.asarray(ora_d, np.float64).astype(np.float32)
    np.testing.assert_array_equal(g.view(np.uint32), o.view(np.uint32), err_msg=what + ": distance bits")
