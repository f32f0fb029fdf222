"""Mirror of ``hnsw.helper.index-io`` (src/hnsw/helper/index_io.clj): ``save_index(index, filepath)`` /
``load_index(filepath, distance_fn)`` / ``index_exists(filepath)``.  The reference writes one EDN string
of every node (492.9 MB for 31k vectors); here the vectors, graph and IVF lists go to one flat binary
file written by the engine, and the id table to ``<filepath>.ids.json`` next to it."""
import json
import os

from . import engine, ivf_flat, ultra_fast


def save_index(index, filepath):
    """index_io.clj:10-39 (works for UltraGraph and IVFFlatIndex)."""
    index.index.save(filepath)
    meta = {"ids": list(index.ids), "kind": "ivf" if isinstance(index, ivf_flat.IVFFlatIndex) else "hnsw",
            "M": getattr(index, "M", None), "ef_construction": getattr(index, "ef_construction", None),
            "num_partitions": getattr(index, "num_partitions", None)}
    with open(filepath + ".ids.json", "w") as f:
        json.dump(meta, f)
    return index


def load_index(filepath, distance_fn=ultra_fast.cosine_distance_ultra, device=0):
    """index_io.clj:41-80: ``None`` when the file does not exist; the metric is stored in the file, a
    ``distance_fn`` that disagrees with it is an error (the reference re-supplies it because a fn cannot be
    serialised, :41-43)."""
    if not index_exists(filepath):
        return None
    idx = engine.Index.load(filepath, device)
    want = ultra_fast._metric_of(distance_fn)
    if want != idx.metric:
        idx.close()
        raise ValueError("index was built with metric %d, distance_fn asks for %d" % (idx.metric, want))
    with open(filepath + ".ids.json") as f:
        meta = json.load(f)
    if meta["kind"] == "ivf":
        return ivf_flat.IVFFlatIndex(idx, meta["ids"], distance_fn, meta["num_partitions"])
    return ultra_fast.UltraGraph(idx, meta["ids"], meta["M"], meta["ef_construction"], distance_fn)


def index_exists(filepath):
    """index_io.clj:82-85"""
    return os.path.exists(filepath) and os.path.exists(filepath + ".ids.json")
