"""Generates tests/golden/*.npz with the CPU oracle (oracle/oracle.c, f64 reference-order mode).

The reference (Clojure/JVM) cannot run in this image, so these fixtures are produced by the
restatement, whose arithmetic is pinned by the reference's own KATs (tests/test_oracle_kat.py).
Inputs are NOT stored: they are regenerated from the java.util.Random-compatible generator
(test/data_generator.clj semantics), seed 42 for the base, 43 for the queries.

Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {
    # name: (n, dim, distribution, nq)
    "g256x64": (256, 64, "gaussian", 16),
    "c1000x128": (1000, 128, "clustered", 16),
}


def inputs(name):
    n, dim, dist, nq = CASES[name]
    base = O.generate_dataset(n, dim, dist, num_clusters=8, noise_level=0.3, seed=42).astype(np.float32)
    Q = O.generate_dataset(nq, dim, dist, num_clusters=8, noise_level=0.3, seed=43).astype(np.float32)
    return base, Q


def main():
    for name in CASES:
        base, Q = inputs(name)
        out = {}
        for mname, metric in (("cos", O.COSINE), ("l2", O.L2), ("dot", O.DOT)):
            g = O.hnsw_build(base, metric, M=8, ef_construction=64, seed=42)
            ids, d, st, _ = O.hnsw_search(base, g, Q, 10, ef=50, metric=metric)
            ex, exd, _ = O.exact_knn(base, Q, 10, metric=metric)
            out.update({
                mname + "_levels": g.levels, mname + "_l0": g.l0_adj.astype(np.int16), mname + "_up_off": g.up_off,
                mname + "_up": g.up_adj.astype(np.int16), mname + "_entry": g.entry, mname + "_maxl": g.max_level,
                mname + "_hnsw_ids": ids.astype(np.int16), mname + "_hnsw_d": d, mname + "_hnsw_stats": st,
                mname + "_exact_ids": ex.astype(np.int16), mname + "_exact_d": exd,
            })
        cen, assign = O.ivf_build(base, nlist=16, max_iterations=10, metric=O.COSINE, seed=42)
        off, lids = O.lists_from_assign(assign, 16)
        cen32 = cen.astype(np.float32)
        iv_ids, iv_d, probes = O.ivf_search(base, cen32, off, lids, Q, 10, 4)
        out.update({"ivf_kpp": O.kmeanspp(base, 16), "ivf_assign": assign.astype(np.int16), "ivf_cent": cen32,
                    "ivf_ids": iv_ids.astype(np.int16), "ivf_d": iv_d, "ivf_probes": probes.astype(np.int16)})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "written", os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")


if __name__ == "__main__":
    main()
