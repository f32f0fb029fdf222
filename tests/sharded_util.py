"""TEST INFRASTRUCTURE: oracle-backed stand-ins for the local operations of hnsw-clj_amd/sharded.py (the product's
are HIP kernels, EngineOps), so that the N > 1 control flow -- seeding, distributed Lloyd, dealing of lists, the row
exchange, the gather layout, the keyed merge -- runs on CPU ranks under gloo.  Also the helpers the GPU tests use
to cut an unsharded index into shards."""
import numpy as np
import torch


class _OracleHandle:
    def __init__(self, O, x, metric):
        self.O, self.metric = O, metric
        self.x = np.ascontiguousarray(x.cpu().numpy() if hasattr(x, "cpu") else x, np.float32)

    def kmeanspp(self, nlist, seed):
        return self.O.kmeanspp(self.x, nlist, self.metric, seed)

    def kmeans_assign(self, cen):
        return self.O.kmeans_assign_f32(self.x, cen, self.metric, self.O.MODE_DEV)

    def list_sums(self, off, ids):
        out = np.zeros((len(off) - 1, self.x.shape[1]), np.float64)
        for l in range(len(off) - 1):
            for i in ids[off[l]:off[l + 1]]:                   # index order, f64, one add at a time (ivf_flat.clj:70-75)
                out[l] += self.x[i].astype(np.float64)
        return out

    def close(self):
        pass


class _OracleShard:
    def __init__(self, O, rows, metric, cen, off, glen):
        self.O, self.metric = O, metric
        self.rows = np.ascontiguousarray(rows.cpu().numpy(), np.float32)
        self.cen, self.off, self.glen = cen, np.asarray(off, np.int64), np.asarray(glen, np.int64)


class OracleOps:
    """Same protocol as hnsw_clj_amd.sharded.EngineOps, computed by oracle/ (its device-order f32 mode)."""

    def __init__(self, O, metric=0):
        self.O, self.metric = O, metric

    def open(self, x, metric):
        return _OracleHandle(self.O, x, self.metric)

    def open_shard(self, rows, metric, cen, off, glen):
        return _OracleShard(self.O, rows, self.metric, cen, off, glen)

    def search(self, sh, Q, k, nprobe):
        O = self.O
        Qn = np.ascontiguousarray(Q.cpu().numpy(), np.float32)
        n = len(sh.rows)
        if n == 0:
            ids = np.full((len(Qn), k), -1, np.int32)
            return torch.from_numpy(ids), torch.full((len(Qn), k), np.inf), torch.from_numpy(ids.copy())
        # device-order f32 arithmetic: the distances are float32 values, so the f32 transport of the gather is lossless
        ids, d, probes = O.ivf_search(sh.rows, sh.cen, sh.off, np.arange(n, dtype=np.int32), Qn, k, nprobe,
                                      metric=self.metric, mode=O.MODE_DEV)
        order = np.full(ids.shape, -1, np.int64)
        for q in range(len(Qn)):
            gbase = np.concatenate(([0], np.cumsum(sh.glen[probes[q]])))       # candidate stream of the WHOLE index
            for j, i in enumerate(ids[q]):
                if i < 0:
                    continue
                l = int(np.searchsorted(sh.off, i, side="right") - 1)
                p = int(np.flatnonzero(probes[q] == l)[0])
                order[q, j] = gbase[p] + (i - sh.off[l])
        return (torch.from_numpy(ids), torch.from_numpy(d.astype(np.float32)),
                torch.from_numpy(order.astype(np.uint32).view(np.int32)))

    @staticmethod
    def merge(ids, dist_, order):
        ns, nq, k = ids.shape
        ids, d, o = ids.numpy(), dist_.numpy(), order.numpy().view(np.uint32)
        oi = np.full((nq, k), -1, np.int32)
        od = np.full((nq, k), np.inf, np.float32)
        for q in range(nq):
            flat = [(d[s, q, r], int(o[s, q, r]), ids[s, q, r]) for s in range(ns) for r in range(k) if ids[s, q, r] >= 0]
            flat.sort()
            for i, t in enumerate(flat[:k]):
                oi[q, i], od[q, i] = t[2], t[0]
        return torch.from_numpy(oi), torch.from_numpy(od)


def cut_into_shards(base, cen, off, lids, nshard):
    """An unsharded IVF index (get_ivf layout) -> per shard (rows, global ids, local list_off) with whole lists dealt
    by hnsw_clj_amd.sharded.deal_lists; rows list by list, index order inside a list."""
    from hnsw_clj_amd.sharded import deal_lists

    lens = np.diff(off)
    owner = deal_lists(lens, nshard)
    out = []
    for s in range(nshard):
        mine = np.flatnonzero(owner == s)
        gid = np.concatenate([lids[off[l]:off[l + 1]] for l in mine]) if len(mine) else np.zeros(0, np.int32)
        loc = np.zeros(len(lens), np.int64)
        loc[mine] = lens[mine]
        loff = np.concatenate(([0], np.cumsum(loc))).astype(np.int64)
        out.append((np.ascontiguousarray(base[gid]), gid.astype(np.int32), loff))
    return out, lens, owner
