"""CPU, world_size 2, gloo: the N>1 plumbing of the path (points_matching_amd/shard.py) — query-row sharding with ONE
all-gather of the survivor blocks, hypothesis-id sharding with ONE all-gather of the 80-byte (key, F) records, winner =
largest key.  The per-rank compute is the oracle here (the HIP kernels need a GPU; tests/test_bench_gpu.py runs the same
exchange through them with two ranks on one GPU); what is under test is that the exchange reproduces the unsharded
answer."""
import os
import sys

import numpy as np
import pytest
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
        # exchange 1: the survivor block (count | xy1 | xy2), one all-gather
        g_blk = shard.gathered_blocks(world, nq, "cpu")
        blk, n, xy1, xy2 = shard.survivor_block(nq, "cpu", into=g_blk[rank])      # in place, as bench.py runs it
        n[0] = good.size
        xy1[:good.size] = torch.from_numpy(O.gather_points(w["kp1"], good["queryIdx"]))
        xy2[:good.size] = torch.from_numpy(O.gather_points(w["kp2"], good["trainIdx"]))
        dist.all_gather_into_tensor(g_blk.view(-1), blk)
        a1, a2, cnt = shard.concat_blocks(g_blk, nq)
        view = shard.view_of_blocks(g_blk, nq)
        # this rank's hypothesis ids over ALL correspondences -> its 80-byte record
        hb, he = shard.hyp_shard(H, rank, world)
        rc, F_r, _, _, key_r = O.ransac_fundamental(a1, a2, he, 1.0, 77, hyp_begin=hb)
        rec = torch.from_numpy(shard.make_record(key_r, F_r))
        # exchange 2: the records, one all-gather; every rank picks the largest key's model
        g_rec = torch.zeros((world, 10), dtype=torch.float64)
        dist.all_gather_into_tensor(g_rec.view(-1), rec)
        key, F = shard.pick_record(g_rec.numpy())
        full = O.ransac_fundamental(a1, a2, H, 1.0, 77)
        cnt_o, mask = O.score(F.astype(np.float32), a1, a2, 1.0)
        q.put((rank, sum(cnt), cnt[rank] == good.size, key == full[4],
               bool((F.view(np.uint64) == full[1].view(np.uint64)).all() and (mask == full[2]).all() and cnt_o == full[3]),
               (hb, he), a1.tobytes(), (view.parts, view.cap, view.pitch_xy)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 5])
def test_gloo_sharded_path_equals_unsharded(world):
    """world 2, and an odd world size whose hypothesis cuts are uneven (400 ids over 5 ranks, rows 512 per rank)."""
    port = 29000 + (os.getpid() * 7 + world) % 2000
    ctxmp = mp.get_context("spawn")
    q = ctxmp.Queue()
    procs = [ctxmp.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0, 'worker failed'
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert all(r[1] == res[0][1] for r in res) and res[0][1] > 100    # same global correspondence count on every rank
    assert all(r[2] and r[3] and r[4] for r in res)  # winner key, model bits, mask and count = the unsharded run's
    assert [r[5] for r in res] == [(400 * g // world, 400 * (g + 1) // world) for g in range(world)]
    assert all(r[6] == res[0][6] for r in res)     # identical gathered correspondences, in rank order
    assert res[0][7] == (world, 512, 4 + 4 * 512)


def _pipe_worker(rank, world, port, q):
    """Four image pairs through the TWO-SLOT pipeline bench.py runs for N > 1 (pair i in slot i % 2; pair i+1's matcher and
    its survivor all-gather start before pair i's RANSAC half has finished; a slot is re-used only when its pair is done).
    CPU analogue: the exchanges are asynchronous gloo collectives, the compute is the oracle."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pm_oracle as O
        from points_matching_amd import shard, synth
        nq, nt, H, P = 256, 300, 200, 4
        slots = []
        for _ in range(2):
            g_blk = shard.gathered_blocks(world, nq, "cpu")
            blk, n, xy1, xy2 = shard.survivor_block(nq, "cpu", into=g_blk[rank])
            slots.append({"g_blk": g_blk, "blk": blk, "n": n, "xy1": xy1, "xy2": xy2, "g_rec": torch.zeros((world, 10), dtype=torch.float64),
                          "w1": None, "pair": None})
        hb, he = shard.hyp_shard(H, rank, world)
        results = {}

        def match(i):
            S = slots[i & 1]
            assert S["pair"] is None                        # the slot's previous pair has finished
            w = synth.pair_workload(nq, nt, 64, seed=100 + i, rank=rank, planted=0.5, kind="surf")
            good = O.filter_ratio(O.bf_knn_l2(w["q"], w["t"], 2), 0.8)
            S["n"][0] = good.size
            S["xy1"][:good.size] = torch.from_numpy(O.gather_points(w["kp1"], good["queryIdx"]))
            S["xy2"][:good.size] = torch.from_numpy(O.gather_points(w["kp2"], good["trainIdx"]))
            S["w1"] = dist.all_gather_into_tensor(S["g_blk"].view(-1), S["blk"], async_op=True)   # exchange 1 in flight
            S["pair"] = i

        def rest(i):
            S = slots[i & 1]
            S["w1"].wait()
            a1, a2, _ = shard.concat_blocks(S["g_blk"], nq)
            rc, F_r, _, _, key_r = O.ransac_fundamental(a1, a2, he, 1.0, 77 + i, hyp_begin=hb)
            rec = torch.from_numpy(shard.make_record(key_r, F_r))
            dist.all_gather_into_tensor(S["g_rec"].view(-1), rec)                                   # exchange 2
            key, F = shard.pick_record(S["g_rec"].numpy())
            full = O.ransac_fundamental(a1, a2, H, 1.0, 77 + i)
            results[i] = (key == full[4], bool((F.view(np.uint64) == full[1].view(np.uint64)).all()), a1.tobytes())
            S["pair"] = None

        for i in range(P):
            match(i)                                         # pair i's matcher + exchange 1 ...
            if i > 0:
                rest(i - 1)                                  # ... overlap pair i-1's RANSAC half
        rest(P - 1)
        q.put((rank, [results[i][:2] for i in range(P)], [results[i][2] for i in range(P)]))
    finally:
        dist.destroy_process_group()


def test_gloo_pipelined_pairs_equal_unsharded():
    world = 2
    port = 31000 + (os.getpid() * 11) % 2000
    ctxmp = mp.get_context("spawn")
    q = ctxmp.Queue()
    procs = [ctxmp.Process(target=_pipe_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0, 'worker failed'
    res = sorted(q.get(timeout=10) for _ in range(world))
    for r in res:
        assert all(k and f for k, f in r[1])                 # every pair: winner key and model bits = the unsharded run's
    assert res[0][2] == res[1][2]                            # every pair: identical gathered correspondences on both ranks
    assert len(set(res[0][2])) == 4                          # ... and the four pairs did not share a slot's contents


def test_row_shard_partitions():
    from points_matching_amd import shard
    for n in (1, 7, 8192, 32768, 1001):
        for world in (1, 2, 3, 8):
            r = [shard.row_shard(n, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))


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
