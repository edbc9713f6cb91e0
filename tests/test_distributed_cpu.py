"""CPU, world_size 2, gloo: the N>1 plumbing of the path (points_matching_amd/shard.py) —
hypothesis-id sharding + the single 8-byte all-reduce(max), and the all-gather of the
query-row-sharded matcher's survivors.  The per-rank compute is the oracle here (the HIP kernels
need a GPU); what is under test is that the exchange reproduces the unsharded answer."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pm_oracle as O
        from points_matching_amd import shard, synth
        nq, nt, H = 512, 600, 400
        w = synth.pair_workload(nq, nt, 64, seed=31, rank=rank, planted=0.5, kind="surf")
        knn = O.bf_knn_l2(w["q"], w["t"], 2)
        good = O.filter_ratio(knn, 0.8)
        xy1 = np.zeros((nq, 2), np.float32)
        xy2 = np.zeros((nq, 2), np.float32)
        xy1[:good.size] = O.gather_points(w["kp1"], good["queryIdx"])
        xy2[:good.size] = O.gather_points(w["kp2"], good["trainIdx"])
        g1 = torch.zeros((world, nq, 2))
        g2 = torch.zeros((world, nq, 2))
        gn = torch.zeros(world, dtype=torch.int32)
        shard.gather_blocks(torch.from_numpy(xy1), torch.from_numpy(xy2),
                            torch.tensor([good.size], dtype=torch.int32), g1, g2, gn)
        a1, a2 = shard.concat_blocks_reference(g1, g2, gn)
        a1, a2 = a1.numpy(), a2.numpy()
        hb, he = shard.hyp_shard(H, rank, world)
        rc, _, _, _, key = O.ransac_fundamental(a1, a2, he, 1.0, 77, hyp_begin=hb)
        kt = torch.tensor([key], dtype=torch.int64)
        shard.reduce_key(kt)
        full = O.ransac_fundamental(a1, a2, H, 1.0, 77)
        rc2, F, mask, n = O.ransac_model_from_hyp(a1, a2, 0xFFFFFFFF - (int(kt) & 0xFFFFFFFF), 1.0, 77)
        q.put((rank, int(gn.sum()), int(gn[rank]) == good.size, int(kt) == full[4],
               bool((F == full[1]).all() and (mask == full[2]).all()), (hb, he), a1.tobytes()))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_sharded_path_equals_unsharded():
    world, port = 2, 29000 + os.getpid() % 2000
    ctxmp = mp.get_context("spawn")
    q = ctxmp.Queue()
    procs = [ctxmp.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0, 'worker failed'
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert res[0][1] == res[1][1] > 100            # same global correspondence count on both ranks
    assert all(r[2] and r[3] and r[4] for r in res)
    assert res[0][5] == (0, 200) and res[1][5] == (200, 400)
    assert res[0][6] == res[1][6]                  # identical gathered correspondences, in rank order


def test_hyp_shard_partitions():
    from points_matching_amd import shard
    for H in (1, 7, 10000, 100000):
        for world in (1, 2, 3, 8):
            r = [shard.hyp_shard(H, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == H
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))


def test_pair_shard_partitions():
    from points_matching_amd import shard
    for P in (1, 5, 256):
        for world in (1, 2, 3, 8):
            parts = [shard.pair_shard(P, g, world) for g in range(world)]
            assert sorted(p for part in parts for p in part) == list(range(P))
            assert max(len(x) for x in parts) - min(len(x) for x in parts) <= 1
