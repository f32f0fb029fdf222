"""Multi-GPU driver: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The reference shards with threads: split the rows into P partitions, search each, concatenate,
sort, take k (src/hnsw/ann/partition/partitioned_hnsw.clj:149-196).  Here rank r owns a contiguous
row range; every rank searches its shard for the full k (k' = k, so recall is not traded away like
the reference's k-per-partition heuristic :158-162), then ONE all-gather of nq*k*(4+4) bytes per
rank (ids and distance bits packed in one int32 tensor) and a merge kernel produce the global top-k on every rank.  The payload (80 KB per rank at
nq=1024, k=10) is latency-bound, so a single all-gather beats anything ring-pipelined.

``replicated`` mode (the 31k x 768 config, which fits every GPU): the index is replicated, the
QUERIES are sharded, and there is no collective on the data path at all.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced row ranges: the first n % world shards get one extra row."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedSearcher:
    """local_search(Q, k) -> (ids int32 [nq,k] local row ids, -1 padded; dist float32 [nq,k])."""

    def __init__(self, local_search, row_offset, group=None, merge_fn=None):
        self.local_search = local_search
        self.row_offset = int(row_offset)
        self.group = group
        self.merge_fn = merge_fn

    def search(self, Q, k):
        world = dist.get_world_size(self.group)
        ids, d = self.local_search(Q, k)
        gids = torch.where(ids >= 0, ids + self.row_offset, ids)  # local row -> global row id
        # ONE collective per batch: (id, distance bits) packed as int32 pairs, nq * k * 8 bytes per rank
        mine = torch.stack((gids, d.contiguous().view(torch.int32)), dim=0).contiguous()      # [2, nq, k]
        flat = torch.empty((world * 2,) + tuple(gids.shape), dtype=torch.int32, device=mine.device)
        dist.all_gather_into_tensor(flat, mine, group=self.group)     # rank r's pair lands at rows [2r, 2r + 2)
        both = flat.view((world, 2) + tuple(gids.shape))
        all_ids = both[:, 0].contiguous()                   # [world, nq, k]: the merge kernel's layout
        all_d = both[:, 1].contiguous().view(torch.float32)
        merge = self.merge_fn
        if merge is None:
            if not all_ids.is_cuda:
                raise RuntimeError("the top-k merge runs in the HIP kernel: tensors must live on the GPU")
            from . import engine

            merge = engine.merge_topk_dev
        return merge(all_ids, all_d)


def split_queries(nq, rank, world):
    """replicated mode: rank's slice of a query batch."""
    return shard_range(nq, rank, world)
