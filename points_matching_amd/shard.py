"""Multi-GPU plumbing of the path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

What shards and what is exchanged
  * matcher      : query rows are split over ranks, the train set is replicated; the per-rank
                   survivors of the filter are all-gathered as fixed-size padded blocks + counts
                   (N x nq x 16 B; at 8k rows 128 KiB per rank).
  * RANSAC       : hypothesis ids [H*r/N, H*(r+1)/N) per rank over ALL gathered correspondences;
                   ONE all-reduce(max) of the packed 8-byte key (inliers << 32 | ~id) names the
                   winner and every rank re-derives F + mask from the id (no model broadcast).
The key fits a signed int64 as long as the inlier count is < 2^31, so torch's int64 MAX is the
uint64 max the C ABI defines.
"""
import torch
import torch.distributed as dist


def hyp_shard(n_hyp, rank, world):
    """Contiguous id range of `rank`; the union over ranks is [0, n_hyp), ranges are disjoint."""
    return rank * n_hyp // world, (rank + 1) * n_hyp // world


def pair_shard(n_pairs, rank, world):
    """Image pairs of a batch owned by `rank` (config C5): pair p -> rank p mod world.  Independent
    units: no collective on the data path."""
    return list(range(rank, n_pairs, world))


def reduce_key(key):
    """key: int64 tensor of one element (this rank's best key).  In-place global max."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(key, op=dist.ReduceOp.MAX)
    return key


def gather_blocks(xy1, xy2, n, out_xy1, out_xy2, out_n):
    """All-gathers the padded survivor blocks (nq x 2 floats each) and their counts.
    out_* have a leading world dimension.  Device- and backend-agnostic."""
    # flat 1-D views: the concatenating form every backend (RCCL and gloo) accepts
    dist.all_gather_into_tensor(out_xy1.view(-1), xy1.view(-1))
    dist.all_gather_into_tensor(out_xy2.view(-1), xy2.view(-1))
    dist.all_gather_into_tensor(out_n.view(-1), n.view(-1))


def concat_blocks_reference(g_xy1, g_xy2, g_n):
    """Torch restatement of pm_concat_points_dev for backends without the HIP library (tests)."""
    parts = [(g_xy1[p, :int(g_n[p])], g_xy2[p, :int(g_n[p])]) for p in range(g_n.numel())]
    return torch.cat([a for a, _ in parts]), torch.cat([b for _, b in parts])
