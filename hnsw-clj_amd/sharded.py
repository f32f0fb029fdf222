"""Multi-GPU driver: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The reference shards with threads: split the rows into P partitions, search each, concatenate,
sort, take k (src/hnsw/ann/partition/partitioned_hnsw.clj:149-196).  Three splits are served here:

* ``ShardedIVF`` -- ONE IVF-FLAT index over all GPUs (BASELINE.json configs[3]): the centroid table is
  replicated, every WHOLE inverted list lives on exactly one GPU (lists dealt longest-first onto the
  least loaded rank), every rank routes a query to the same nprobe lists (ivf_flat.clj:261-269), scans
  the probed lists it holds (:281-288) and numbers its candidates by their position in the candidate
  stream of the whole index; ONE all-gather of nq*k*12 bytes per rank and a merge by (distance,
  position) give, bit for bit, the result of the unsharded index (ids, distances, tie order).
  The build is distributed as well: k-means++ seeds from rank 0's rows, Lloyd over ALL rows with an
  all-reduce of the per-list f64 sums (:92-131), then one all-to-all that moves every row to the rank
  that owns its list.
* ``ShardedSearcher`` -- independent sub-indexes, one per GPU, over contiguous row ranges (=
  PartitionedHNSWIndex, partitioned_hnsw.clj:23-27: configs[4], one HNSW sub-graph per GPU), searched
  with the full k (k' = k, so recall is not traded away like the reference's k-per-partition heuristic
  :158-162); ONE all-gather of nq*k*8 bytes per rank; ties keep the lower shard first, as the reference's
  stable sort of the concatenation does.
* replicas (the 31k x 768 config, which fits every GPU): the index is replicated, the QUERIES are
  sharded, no collective on the data path (``split_queries``).

The payloads (80-120 KB per rank at nq=1024, k=10) are latency-bound: a single all-gather beats
anything ring-pipelined on point-to-point xGMI links.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced row ranges: the first n % world shards get one extra row."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def split_queries(nq, rank, world):
    """replicated mode: rank's slice of a query batch."""
    return shard_range(nq, rank, world)


def deal_lists(global_len, world):
    """Whole inverted lists -> ranks, balanced by row count: longest list first onto the least loaded rank
    (ties: the lower list index first, the lower rank first -- every rank computes the same table).
    Returns owner[nlist] (int32)."""
    glen = np.asarray(global_len, np.int64)
    order = np.lexsort((np.arange(len(glen)), -glen))        # by length descending, then list index
    load = np.zeros(world, np.int64)
    owner = np.zeros(len(glen), np.int32)
    for l in order:
        r = int(np.argmin(load))                             # first minimum = lowest rank
        owner[l] = r
        load[r] += glen[l]
    return owner


def lists_from_assign(assign, nlist):
    """Inverted lists in index order (ivf_flat.clj:126-131): list_off (nlist + 1) int64, list_ids (n) int32."""
    assign = np.asarray(assign)
    ids = np.argsort(assign, kind="stable").astype(np.int32)
    off = np.zeros(nlist + 1, np.int64)
    off[1:] = np.cumsum(np.bincount(assign, minlength=nlist))
    return off, ids


class Comm:
    """The four collectives the sharded paths use, over torch.distributed.  RCCL ("nccl") moves CUDA tensors;
    under "gloo" (CPU rehearsals and tests: N ranks on one GPU or none) the same calls stage through host
    memory.  Without an initialised process group it is the single-rank identity."""

    def __init__(self, group=None, device=None, always_collective=False):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.host = (not self.on) or dist.get_backend(group) == "gloo"
        self.device = device
        # a one-rank group normally short-cuts every collective; `always_collective` issues them anyway (bench.py
        # --sharded on one GPU: the RCCL calls of the N > 1 path run, with one rank, on the hardware that is there)
        self.single = self.world == 1 and not (always_collective and self.on)

    def _wire(self, t):
        t = torch.as_tensor(t)
        return t.cpu().contiguous() if self.host else t.to(self.device).contiguous()

    def all_reduce_sum(self, a):
        """numpy in, numpy out (small tables: list sums and counts)."""
        if self.single:
            return np.asarray(a)
        t = self._wire(np.ascontiguousarray(a))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def broadcast(self, a, src=0):
        if self.single:
            return np.asarray(a)
        t = self._wire(np.ascontiguousarray(a))
        dist.broadcast(t, src, group=self.group)
        return t.cpu().numpy()

    def all_gather(self, t):
        """tensor [..] (same shape on every rank) -> [world, ..] on t's device."""
        if self.single:
            return t.unsqueeze(0)
        w = self._wire(t)
        out = torch.empty((self.world,) + tuple(w.shape), dtype=w.dtype, device=w.device)
        dist.all_gather_into_tensor(out.view(-1), w.view(-1), group=self.group)
        return out.to(t.device)

    def all_to_all_rows(self, send, send_counts, recv_counts):
        """send: [sum(send_counts), width] rows grouped by destination rank -> [sum(recv_counts), width] rows grouped
        by source rank (RCCL all-to-all; peer-to-peer sends under gloo)."""
        if self.single:
            return send
        w = self._wire(send)
        out = torch.empty((int(sum(recv_counts)),) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
        if not self.host:
            dist.all_to_all_single(out, w, [int(c) for c in recv_counts], [int(c) for c in send_counts], group=self.group)
        else:
            so = np.concatenate(([0], np.cumsum(send_counts))).astype(np.int64)
            ro = np.concatenate(([0], np.cumsum(recv_counts))).astype(np.int64)
            out[ro[self.rank]:ro[self.rank + 1]] = w[so[self.rank]:so[self.rank + 1]]
            ops = []
            for peer in range(self.world):
                if peer == self.rank:
                    continue
                if send_counts[peer]:
                    ops.append(dist.P2POp(dist.isend, w[so[peer]:so[peer + 1]], peer, self.group))
                if recv_counts[peer]:
                    ops.append(dist.P2POp(dist.irecv, out[ro[peer]:ro[peer + 1]], peer, self.group))
            if ops:
                for r in dist.batch_isend_irecv(ops):
                    r.wait()
        return out.to(send.device)


class EngineOps:
    """The product's local operations: every one is a HIP kernel behind the C ABI (no CPU path)."""

    def __init__(self, device):
        self.device = int(device)

    def open(self, x, metric):
        from . import engine

        return engine.Index(x, metric, self.device)

    def open_shard(self, rows, metric, centroids, list_off, global_len):
        from . import engine

        idx = engine.Index(rows, metric, self.device)
        idx.set_ivf_shard(centroids, list_off, np.arange(idx.n, dtype=np.int32), global_len)
        return idx

    @staticmethod
    def search(idx, Q, k, nprobe):
        return idx.ivf_search_shard_dev(Q, k, nprobe)

    @staticmethod
    def merge(ids, dist_, order):
        if not ids.is_cuda:
            raise RuntimeError("the top-k merge runs in the HIP kernel: tensors must live on the GPU")
        from . import engine

        return engine.merge_keyed_dev(ids, dist_, order)


class ShardedIVF:
    """One IVF-FLAT index whose inverted lists are dealt to the ranks of a process group (see the module text)."""

    def __init__(self, comm, ops, shard, gid, centroids, list_off, global_len, owner, assign, row_base):
        self.comm, self.ops, self.shard = comm, ops, shard
        self.gid = gid                    # tensor int32 [rows held]: local row -> global row id
        self.centroids, self.list_off, self.global_len, self.owner = centroids, list_off, global_len, owner
        self.assign, self.row_base = assign, row_base   # this rank's ORIGINAL rows: their list, their first global id

    @classmethod
    def build(cls, x, metric="cosine", nlist=24, max_iterations=10, seed=42, comm=None, ops=None):
        """x: this rank's rows ([n_r, dim] float32 tensor, any device).  Global row ids are rank-major: rank r's row i is
        sum(n_0 .. n_{r-1}) + i.  Mirrors build-ivf-flat-index :partition-method :kmeans (ivf_flat.clj:137-211)."""
        comm = comm or Comm(device=x.device if x.is_cuda else None)
        ops = ops or EngineOps(x.device.index or 0)
        world, rank = comm.world, comm.rank
        n, dim = int(x.shape[0]), int(x.shape[1])
        n_all = comm.all_gather(torch.tensor([n], dtype=torch.int64)).view(-1).numpy()
        assert int(n_all.sum()) < 2 ** 31, "row ids are int32"
        row_base = int(n_all[:rank].sum())
        h = ops.open(x, metric)
        try:
            # 1. k-means++ seeds (ivf_flat.clj:32-60) drawn from rank 0's rows: the reference's seeding when world = 1,
            #    the same procedure on a 1/world sample otherwise
            if rank == 0:
                assert n >= 1, "rank 0 holds no rows"
                rows = h.kmeanspp(nlist, seed)
                cen = x[torch.as_tensor(rows.astype(np.int64), device=x.device)].float().cpu().numpy()
            else:
                cen = np.zeros((nlist, dim), np.float32)
            cen = np.ascontiguousarray(comm.broadcast(cen, 0), np.float32)
            # 2. Lloyd over ALL rows (:100-117): local assignment, local f64 list sums, all-reduce, divide
            for it in range(max_iterations + 1):
                assign = h.kmeans_assign(cen)[0] if n else np.zeros(0, np.int32)
                off, ids = lists_from_assign(assign, nlist)
                if it == max_iterations:                              # the final assignment (:120-124)
                    break
                sums = h.list_sums(off, ids) if n else np.zeros((nlist, dim), np.float64)
                sums = comm.all_reduce_sum(sums)
                cnt = comm.all_reduce_sum(np.diff(off))
                keep = cnt == 0                                       # an empty cluster keeps its centroid (:112-114)
                new = (sums / np.maximum(cnt, 1)[:, None].astype(np.float64)).astype(np.float32)
                new[keep] = cen[keep]
                cen = new
        finally:
            h.close()
        # 3. deal the lists; 4. move every row to the owner of its list
        lens_all = comm.all_gather(torch.from_numpy(np.diff(off))).numpy()      # [world, nlist]
        glen = lens_all.sum(axis=0)
        owner = deal_lists(glen, world)
        mine = np.flatnonzero(owner == rank)
        # rows leave grouped by destination, inside a destination by (list, index order)
        dest_of_list_sorted = owner[assign[ids]] if n else np.zeros(0, np.int32)   # ids: rows in (list, index) order
        send_order = ids[np.argsort(dest_of_list_sorted, kind="stable")]
        send_counts = np.bincount(dest_of_list_sorted, minlength=world).astype(np.int64)
        recv_counts = lens_all[:, mine].sum(axis=1).astype(np.int64)
        sel = torch.as_tensor(send_order.astype(np.int64), device=x.device)
        rows_in = comm.all_to_all_rows(x[sel], send_counts, recv_counts)
        gid_in = comm.all_to_all_rows(torch.from_numpy((send_order.astype(np.int64) + row_base).astype(np.int32)).to(x.device)
                                      .unsqueeze(1), send_counts, recv_counts).squeeze(1)
        # a source's block holds my lists in ascending list order; row j of list l from source s goes to
        # loff[l] + sum_{s' < s} len[s'][l] + j: lists contiguous, rows in global index order inside a list
        llen = lens_all[:, mine]                                               # [world, mine]
        loff_m = np.concatenate(([0], np.cumsum(llen.sum(axis=0)))).astype(np.int64)
        before = np.cumsum(llen, axis=0) - llen                                # rows of the list from lower ranks
        src_off = np.concatenate(([0], np.cumsum(recv_counts))).astype(np.int64)
        pos = np.empty(int(recv_counts.sum()), np.int64)
        for s in range(world):
            if recv_counts[s] == 0:
                continue
            starts = loff_m[:-1] + before[s]                                   # first slot of (s, list) in the shard
            pos[src_off[s]:src_off[s + 1]] = np.repeat(starts - (np.cumsum(llen[s]) - llen[s]), llen[s]) + \
                np.arange(recv_counts[s])
        perm = torch.as_tensor(pos, device=x.device)
        rows = torch.empty_like(rows_in)
        rows[perm] = rows_in
        gid = torch.empty_like(gid_in)
        gid[perm] = gid_in
        del rows_in, gid_in
        list_off = np.zeros(nlist + 1, np.int64)
        lens_local = np.zeros(nlist, np.int64)
        lens_local[mine] = llen.sum(axis=0)
        list_off[1:] = np.cumsum(lens_local)
        shard = ops.open_shard(rows, metric, cen, list_off, glen)
        # ONE verdict of the first-search calibration for the whole index (include/hnswgpu.h: hnswgpu_ivf_stream_state):
        # every rank measures on its rows, the index takes the survivor stream off if any rank would
        if hasattr(shard, "ivf_stream_state"):
            off = int(shard.ivf_stream_state()) if shard.n > 0 else 0
            off = int(comm.all_reduce_sum(np.array([off], np.int64))[0] > 0)
            shard.ivf_set_stream_state(off)
        return cls(comm, ops, shard, gid, cen, list_off, glen, owner, assign, row_base)

    def search(self, Q, k, nprobe):
        """Q: the SAME batch on every rank -> (global ids int32 [nq, k], distances f32 [nq, k]) on every rank,
        identical to the unsharded index's answer."""
        ids, d, order = self.ops.search(self.shard, Q, k, nprobe)
        gids = torch.where(ids >= 0, self.gid[ids.clamp(min=0).long()], ids) if len(self.gid) else ids
        mine = torch.stack((gids, d.contiguous().view(torch.int32), order), dim=0).contiguous()    # [3, nq, k]
        both = self.comm.all_gather(mine)                                                         # [world, 3, nq, k]
        return self.ops.merge(both[:, 0].contiguous(), both[:, 1].contiguous().view(torch.float32),
                              both[:, 2].contiguous())

    def close(self):
        if self.shard is not None and hasattr(self.shard, "close"):
            self.shard.close()
        self.shard = None


class ShardedSearcher:
    """Independent sub-indexes over contiguous row ranges.
    local_search(Q, k) -> (ids int32 [nq,k] local row ids, -1 padded; dist float32 [nq,k])."""

    def __init__(self, local_search, row_offset, group=None, merge_fn=None, always_collective=False):
        self.local_search = local_search
        self.row_offset = int(row_offset)
        self.group = group
        self.merge_fn = merge_fn
        self.always_collective = always_collective

    def search(self, Q, k):
        ids, d = self.local_search(Q, k)
        gids = torch.where(ids >= 0, ids + self.row_offset, ids)  # local row -> global row id
        # ONE collective per batch: (id, distance bits) packed as int32 pairs, nq * k * 8 bytes per rank
        mine = torch.stack((gids, d.contiguous().view(torch.int32)), dim=0).contiguous()      # [2, nq, k]
        both = Comm(self.group, mine.device if mine.is_cuda else None, self.always_collective).all_gather(mine)  # [world, 2, nq, k]
        all_ids = both[:, 0].contiguous()                   # [world, nq, k]: the merge kernel's layout
        all_d = both[:, 1].contiguous().view(torch.float32)
        merge = self.merge_fn
        if merge is None:
            if not all_ids.is_cuda:
                raise RuntimeError("the top-k merge runs in the HIP kernel: tensors must live on the GPU")
            from . import engine

            merge = engine.merge_topk_dev
        return merge(all_ids, all_d)
