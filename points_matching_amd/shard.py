"""Partitioning rules and exchange layout of the N-GPU form of the path for callers that run one
process per GPU with their own communicator (bench.py: torch.distributed, "nccl" = RCCL over xGMI; the CPU tests:
"gloo").  The single-process form of the same thing lives behind the C ABI (csrc/mgpu.cpp, pm_mgpu_*).

What shards and what is exchanged
  * matcher : query rows are split over ranks, the train set is replicated.  Each rank's filter writes its survivors
              into ONE contiguous block  [count int32 + 3 pad words | xy1: cap x 2 f32 | xy2: cap x 2 f32]
              (`survivor_block`), and exchange 1 is one all-gather of that block.
  * RANSAC  : hypothesis ids [H*r/N, H*(r+1)/N) per rank over ALL gathered correspondences, read in place through a
              pm_points_view of the gathered blocks (`view_of_blocks`); exchange 2 is one all-gather of the 80-byte
              (key, F) record per rank; every rank takes the record with the largest key (`pick_record`): the arg-max
              all-reduce with its payload.  Nobody re-solves, nothing is broadcast.
  * batch   : image pairs p -> rank p mod N, no collective (config C5).
"""
import numpy as np
import torch

from . import api


def hyp_shard(n_hyp, rank, world):
    """Contiguous id range of `rank`; the union over ranks is [0, n_hyp), ranges are disjoint."""
    return rank * n_hyp // world, (rank + 1) * n_hyp // world


def row_shard(n_rows, rank, world):
    """Contiguous query-row block of `rank` for the strong-scaling form (cap = ceil(n / world) rows per rank)."""
    cap = (n_rows + world - 1) // world
    return min(rank * cap, n_rows), min((rank + 1) * cap, n_rows)


def pair_shard(n_pairs, rank, world):
    """Image pairs of a batch owned by `rank` (config C5): pair p -> rank p mod world.  Independent units: no
    collective on the data path."""
    return list(range(rank, n_pairs, world))


def survivor_block(cap, device, into=None):
    """One rank's survivor block and the views the filter writes through: (block f32[4 + 4*cap], n int32[1],
    xy1 f32[cap, 2], xy2 f32[cap, 2]).  `into`: a row of the gathered buffer (`gathered_blocks(...)[rank]`) — the
    filter then writes where the all-gather expects this rank's contribution and the collective runs in place."""
    blk = torch.zeros(4 + 4 * cap, dtype=torch.float32, device=device) if into is None else into
    assert blk.numel() == 4 + 4 * cap and blk.is_contiguous()
    return blk, blk[0:1].view(torch.int32), blk[4:4 + 2 * cap].view(cap, 2), blk[4 + 2 * cap:].view(cap, 2)


def gathered_blocks(world, cap, device):
    """Receive buffer of exchange 1: f32[world, 4 + 4*cap]."""
    return torch.zeros((world, 4 + 4 * cap), dtype=torch.float32, device=device)


def view_of_blocks(g_blk, cap):
    """pm_points_view over `world` gathered survivor blocks (g_blk: f32[world, 4 + 4*cap], contiguous)."""
    world, words = g_blk.shape
    assert words == 4 + 4 * cap and g_blk.is_contiguous()
    base = g_blk.data_ptr()
    return api.PointsView(base + 16, base + 16 + 8 * cap, base, world, cap, words, words, 0)


def concat_blocks(g_blk, cap):
    """The concatenation a view stands for, as host arrays (checker / CPU tests only)."""
    g = g_blk.detach().cpu()
    cnt = g[:, 0:1].contiguous().view(torch.int32).reshape(-1).clamp(0, cap).tolist()
    xy1 = np.concatenate([g[p, 4:4 + 2 * cap].view(cap, 2)[:cnt[p]].numpy() for p in range(g.shape[0])])
    xy2 = np.concatenate([g[p, 4 + 2 * cap:].view(cap, 2)[:cnt[p]].numpy() for p in range(g.shape[0])])
    return xy1, xy2, cnt


def make_record(key, F):
    """pm_ransac_record as 10 float64 words (key bits in word 0)."""
    rec = np.zeros(10, np.float64)
    rec[:1].view(np.uint64)[0] = np.uint64(key)
    rec[1:] = np.asarray(F, np.float64).reshape(9)
    return rec


def pick_record(records):
    """records: float64[world, 10].  Returns (key, F(3x3)) of the record with the largest key — what
    pm_ransac_finish_parts_dev does on the device."""
    records = np.ascontiguousarray(records, np.float64).reshape(-1, 10)
    keys = records[:, 0].copy().view(np.uint64)
    w = int(np.argmax(keys))
    return int(keys[w]), records[w, 1:].reshape(3, 3).copy()
