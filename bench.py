#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path on MI355X (contract: see task brief).

Workload (BASELINE.json configs[2], "C3"): 8k x 8k SIFT-128 float descriptors, brute-force L2
2-NN + ratio test (0.8), then 10 000-hypothesis RANSAC-F (normalised 8-point, Sampson, tau = 1 px)
on the ~2.3k surviving matches.  It carries BOTH halves of BASELINE.json's metric
("descriptor-pair distances/s + RANSAC hypotheses/s") and fits one GPU; configs[1] (2k x 2k) is
launch-latency sized and is covered as a parity test and as `--workload c2`.

One step = one pass of the path over one image pair, everything resident in HBM:
  pm_bf_knn_l2_ratio_dev (prep + coarse matrix-core pass + refinement, then the ratio test + compaction + gather as its
  own launch) -> pm_ransac_run_dev (one launch: sample, solve, score, pick, mask).
N GPUs (weak scaling, one process per GPU): rank r matches its own 8k query rows against the replicated 8k train rows
(global problem = N*8k x 8k) straight into its slot of the gathered survivor buffer -> all-gather #1 of the survivor
blocks (RCCL, in place) -> pm_ransac_shard_parts_dev: the rank's shard of the 10k hypothesis ids over ALL gathered
correspondences, one 80-byte (key, F) record -> all-gather #2 of the records (the arg-max all-reduce with its payload) ->
pm_ransac_finish_parts_dev: every rank takes the record with the largest key and writes F + the inlier mask (nobody
re-solves, nothing is broadcast).  The N > 1 step exists in two forms: software-pipelined (pair i+1's matcher is enqueued on
a second stream while pair i sits in its two all-gathers and RANSAC) and serial (pairs strictly one after the other); a
trial in the warm-up picks the form whose slowest rank is faster (`step_form_trial`; `--pipeline 0/1` pins one), both are
reported (`ms_per_step_serial`, `ms_per_step_pipelined`), and every collective is also timed by itself (`collectives`).

`value` = descriptor-pair distances/s of the matching stage (N*M*ranks / match-stage time);
the RANSAC half of the metric is reported next to it (`ransac.hyp_per_s`, with N_m).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense f32-input MFMA = f32 vector peak
PEAK_F32_VALU_TFLOPS = 157.3
PEAK_F16_MFMA_TFLOPS = 2500.0    # dense f16/bf16 MFMA (spec; the 5 PF headline includes 2:1 sparsity)


def _kernel_source_sha():
    import hashlib
    h = hashlib.sha256()
    for f in ("knn_coarse.hip", "knn_coarse_kernels.hpp", "knn_shared.hpp", "knn_l2.hip", "knn_hamming.hip"):
        with open(os.path.join(ROOT, "points_matching_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(key, nq, nt):
    """HBM bytes per launch of the dominant kernel.  PMC counters cannot be read from inside this process: the figure
    comes from the committed rocprofv3 --pmc passes (profiles/r03_traffic.json: FETCH_SIZE / WRITE_SIZE in separate
    runs, gfx950 correction applied) and is used ONLY when that file is stamped with the hash of the kernel sources
    it was measured on and this is exactly the profiled workload; otherwise null."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "r03_traffic.json")))
        t = doc["kernels"]
        if doc.get("kernel_source_sha16") != _kernel_source_sha():
            return None
    except (OSError, ValueError, KeyError):
        return None
    want = {"c3:knn_l2_mfma_f16": (8192, 8192), "c3:knn_l2_mfma_u8": (8192, 8192), "c3:knn_l2_mfma_f16s": (8192, 8192), "c3_f32:knn_l2_mfma": (8192, 8192), "c4:knn_hamming_mfma_i8": (32768, 32768),
            "l32k:knn_l2_mfma_u8": (32768, 32768), "l32k:knn_l2_mfma_f16": (32768, 32768)}
    if key in t and want.get(key) == (nq, nt):
        return t[key]["traffic_bytes"]
    return None


def bench_c5(args, world, rank, local_rank, dev, multi, saved_stdout):
    """BASELINE configs[4]: a batch of 256 independent image pairs x 4096 SIFT-128 descriptors, H = 2048
    hypotheses per pair, streamed end to end INCLUDING H2D of the descriptors/keypoints and D2H of the
    results (pm_batch_run: `lanes` streams per GPU).  Pairs are sharded over the ranks (pair p -> rank
    p mod N, no collective: SURVEY.md 8e).  One step = the whole batch once."""
    import torch
    import torch.distributed as dist
    import points_matching_amd as pm
    from points_matching_amd import shard, synth

    n, dim, H, ratio, thresh, seed = 4096, 128, 2048, 0.8, 1.0, 0x5EED
    my_pairs = shard.pair_shard(args.pairs, rank, world)
    distinct = min(len(my_pairs), 16)          # distinct synthetic pairs held in pinned memory, cycled
    u8 = args.c5_desc == "u8"
    ddt = np.uint8 if u8 else np.float32

    def build_jobs(one_block):
        """pinned host copies of the distinct pairs and the job list: four separate arrays per pair, or ONE block
        desc1 | desc2 | kp1 | kp2 (256-byte aligned sections), which pm_batch_run sends in one copy"""
        hold, jobs = [], []
        for i in range(distinct):
            w = synth.pair_workload(n, n, dim, seed=0xC5 + my_pairs[i], kind="sift")
            parts = [np.ascontiguousarray(w["q"].astype(ddt)), np.ascontiguousarray(w["t"].astype(ddt)), w["kp1"], w["kp2"]]
            if one_block:
                offs, total = [], 0
                for p in parts:
                    total = (total + 255) // 256 * 256
                    offs.append(total)
                    total += p.nbytes
                blk = torch.zeros(total, dtype=torch.uint8).pin_memory()
                for p, o in zip(parts, offs):
                    blk[o:o + p.nbytes] = torch.from_numpy(p.view(np.uint8).reshape(-1))
                hold.append((blk, [blk.data_ptr() + o for o in offs]))
            else:
                ts = [torch.from_numpy(p).pin_memory() for p in parts]
                hold.append((ts, [t.data_ptr() for t in ts]))
        for i in range(len(my_pairs)):
            a = hold[i % distinct][1]
            jobs.append((a[0], n, a[1], n, a[2], a[3]))
        return hold, jobs
    hold, jobs = build_jobs(args.c5_layout == "block")
    batch = pm.api.PairBatch(local_rank, args.lanes, n, n, dim)
    batch.set_option(pm.api.PM_OPT_RANSAC_PATH, args.ransac_path)
    if args.ransac_wg_ids >= 0:
        batch.set_option(pm.api.PM_OPT_RANSAC_WG_IDS, args.ransac_wg_ids)
    if args.c5_filter_fusion:
        batch.set_option(pm.api.PM_OPT_FILTER_FUSION, args.c5_filter_fusion)
    batch.set_host_threads(args.c5_host_threads)
    batch.set_desc_u8(u8)
    arr = batch.make_jobs(jobs)
    flags = pm.api.PM_KNN_HINT_U8

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    for _ in range(max(1, args.warmup // 5)):
        res, _, _ = batch.run(arr, ratio, H, thresh, seed, knn_flags=flags)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, _, _ = batch.run(arr, ratio, H, thresh, seed, knn_flags=flags)
    fence()
    wall = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())

    parity = "skipped"
    if not args.no_verify and rank == 0:
        from oracle import pm_oracle as O
        ok = True
        for j in (0, min(1, len(my_pairs) - 1)):
            w = synth.pair_workload(n, n, dim, seed=0xC5 + my_pairs[j % distinct], kind="sift")
            knn = O.bf_knn_l2(w["q"], w["t"], 2, nthreads=8)            # (u8 rows hold the same values)
            good = O.filter_ratio(knn, ratio)
            rc, F_o, mask_o, ninl_o, key_o = O.ransac_fundamental(w["kp1"][good["queryIdx"]], w["kp2"][good["trainIdx"]],
                                                                  H, thresh, seed, nthreads=8)
            r = res[j]
            ok = ok and r.n_good == good.size and r.best_key == key_o and r.n_inliers == ninl_o and \
                (np.array(r.F[:]).view(np.uint64) == F_o.reshape(9).view(np.uint64)).all()
        parity = "ok" if ok else "MISMATCH"
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    P = args.pairs
    ms_per_step = wall / args.steps * 1e3
    pairs_per_s = P / (ms_per_step * 1e-3)
    h2d = (2 * n * dim * (1 if u8 else 4) + 2 * n * 8)
    out = {
        "metric": "descriptor-pair distances/s (BF-L2 2-NN + ratio stage); RANSAC hypotheses/s in `ransac`",
        "value": pairs_per_s * n * n, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u8" if args.c5_desc == "u8" else "f32",
        "dtype_note": ("u8 rows in (pm_batch_set_desc_type), f32 distances out" if args.c5_desc == "u8" else "f32 in/out") +
                      "; coarse pass i8 x i8 -> i32 MFMA on x - 128 (exact for u8-valued data, device-verified), integer "
                      "refinement on the same bytes, distances = canonical f32 bits",
        "data": "synthetic",
        "config": {"workload": "C5: batch of %d image pairs x (%d x %d SIFT-128 %s BF-L2 2-NN + ratio 0.8 + %d-hypothesis "
                               "RANSAC-F), end to end incl. H2D/D2H from pinned host memory, %d lanes per GPU, %s; pairs "
                               "sharded over ranks, no collective" % (P, n, n, "u8 rows" if u8 else "f32", H, args.lanes,
                                                                      "one copy per pair" if args.c5_layout == "block" else "four copies per pair"),
                   "descriptors": "sift", "descriptor_rows": args.c5_desc, "pair_layout": args.c5_layout, "k": 2,
                   "distinct_pairs_per_rank": distinct},
        "image_pairs_per_s": pairs_per_s,
        "ransac": {"hyp_per_s": pairs_per_s * H, "hypotheses": H, "n_matches": int(res[0].n_good),
                   "inliers": int(res[0].n_inliers)},
        "pcie": {"h2d_bytes_per_pair": h2d, "h2d_GBps": pairs_per_s / world * h2d / 1e9,
                 "note": "per-GPU host->device rate sustained by the pipeline; the inputs are %.1f MB per pair" % (h2d / 1e6)},
        "parity": parity,
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def launch_ranks(n):
    """Parent of a multi-GPU run: never initialises the GPU, never execs; children are fresh processes."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rc = subprocess.call(cmd, env=env)
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nq", type=int, default=None, help="query rows (default: the workload's)")
    ap.add_argument("--nt", type=int, default=None, help="train rows (default: the workload's)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--hyps", type=int, default=None, help="RANSAC hypotheses (default: the workload's)")
    ap.add_argument("--kind", default="sift", choices=["sift", "surf", "orb"],
                    help="descriptor family: sift (u8-valued f32, BASELINE C3), surf (general f32), orb (256-bit, C4)")
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE config: c3 (default, headline), c2 = 2k x 2k SIFT latency case, "
                         "c4 = 32k x 32k ORB-256 + 100k hypotheses (per GPU: the multi-GPU run shards it), "
                         "c5 = batch of 256 image pairs x 4k descriptors streamed end to end (pairs sharded over ranks)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every rank matches its own nq query rows (global problem N*nq x nt, the default); "
                         "strong = ONE nq x nt problem, query rows and hypothesis ids both cut N ways (BASELINE config 4 is "
                         "`--workload c4 --scaling strong`)")
    ap.add_argument("--pipeline", type=int, default=-1, choices=[-1, 0, 1],
                    help="software-pipeline consecutive pairs over two streams (pair i+1's matcher while pair i is in its "
                         "exchanges / RANSAC): -1 = on for N > 1 (and --exercise-exchange), off for one GPU; the serial figure is "
                         "reported next to it either way")
    ap.add_argument("--no-large", action="store_true", help="skip the 32k x 32k roofline leg of the matcher")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="after the K timed steps: back-to-back steps for at least this long (clock-sustained figure)")
    ap.add_argument("--pairs", type=int, default=256, help="c5: image pairs in the whole job")
    ap.add_argument("--lanes", type=int, default=6, help="c5: streams (lanes) per GPU")
    ap.add_argument("--c5-desc", default="f32", choices=["f32", "u8"],
                    help="c5: descriptor rows on the host: f32 (BASELINE's, the headline) or u8 (the same values as bytes: a quarter "
                         "of the link traffic, pm_batch_set_desc_type)")
    ap.add_argument("--c5-layout", default="block", choices=["block", "separate"],
                    help="c5: a pair's four host arrays in one pinned block (one copy per pair) or separate (four copies)")
    ap.add_argument("--exercise-exchange", action="store_true",
                    help="debugging: run the N>1 code path (all-gather, concat, key all-reduce, model from key) "
                         "even with one rank")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="debugging: gloo lets a multi-rank run share ONE GPU (with --single-device); the driver uses nccl")
    ap.add_argument("--single-device", action="store_true", help="debugging: every rank uses cuda:0")
    ap.add_argument("--c5-host-threads", type=int, default=0, choices=[0, 1, 2],
                    help="c5: host threads enqueueing the pairs (0: automatic = 2 with >= 4 lanes)")
    ap.add_argument("--c5-filter-fusion", type=int, default=0, choices=[0, 1, 2],
                    help="c5: PM_OPT_FILTER_FUSION on every lane (0 = the library's choice, 1 = filter as its own launch, 2 = inside the refinement)")
    ap.add_argument("--ransac-wg-ids", type=int, default=-1,
                    help="c5: hypothesis ids per RANSAC workgroup (-1: the batch's default, 32; 0: spread over all CUs)")
    ap.add_argument("--ransac-path", type=int, default=0, choices=[0, 1, 2],
                    help="A/B timing: PM_OPT_RANSAC_PATH (0 automatic, 1 solve + score launches, 2 one-launch kernel)")
    ap.add_argument("--hint", default="u8", choices=["u8", "int", "auto"],
                    help="SIFT descriptors: what the caller states about them (u8: integers in [0, 255] -> i8 matrix pass; "
                         "int: integers, |x| <= 361 -> f16 matrix pass; auto: nothing, the device decides)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the general-float leg and the forced f32-route leg (clean kernel traces of the headline path)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This parent has not
    # imported torch or touched HIP; it only spawns `torch.distributed.run` (one fresh process per GPU),
    # relays the children's stdout (rank 0 prints the JSON line) and exits with their status.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)

    # Only the JSON line may reach stdout: RCCL prints a version banner to fd 1 when it initialises,
    # so fd 1 points at stderr until the result is ready.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import points_matching_amd as pm
    from points_matching_amd import shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.exercise_exchange
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        elif world > 1:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if multi:
        world = dist.get_world_size()          # what the communicator actually saw: this is `n_gpus` in the JSON

    if args.workload == "c5":
        return bench_c5(args, world, rank, local_rank, dev, multi, saved_stdout)
    wl_n, wl_h = {"c2": (2048, 10000), "c3": (8192, 10000), "c4": (32768, 100000)}[args.workload]
    if args.workload == "c4":
        args.kind, args.dim = "orb", 32
    nq = args.nq if args.nq else wl_n
    nt = args.nt if args.nt else wl_n
    dim, H, K = args.dim, args.hyps if args.hyps else wl_h, 2
    hamming = args.kind == "orb"
    ratio, thresh, seed = 0.8, 1.0, 0x5EED
    # A SIFT matcher knows its descriptors are u8-valued floats: state it, so that only the exact
    # f16-MFMA coarse route is enqueued (the claim is verified on the device).  Other kinds: auto.
    knn_flags = {"u8": pm.api.PM_KNN_HINT_U8, "int": pm.api.PM_KNN_HINT_INTEGER, "auto": 0}[args.hint] if args.kind == "sift" else \
        (pm.api.PM_KNN_HINT_UNIT_NORM if args.kind == "surf" and args.hint != "auto" else 0)      # SURF rows are unit-norm
    wseed = 0xC4 if hamming else 0xC3
    nq_total = nq
    if args.scaling == "strong" and world > 1:
        # ONE problem: every rank builds it and keeps its contiguous block of query rows (train set replicated)
        if nq % world:
            raise SystemExit("bench.py: --scaling strong needs nq divisible by the number of ranks")
        w = synth.pair_workload(nq, nt, dim, seed=wseed, rank=0, kind=args.kind)
        nq = nq // world
        w["q"] = np.ascontiguousarray(w["q"][rank * nq:(rank + 1) * nq])
        w["kp1"] = np.ascontiguousarray(w["kp1"][rank * nq:(rank + 1) * nq])
    else:
        w = synth.pair_workload(nq, nt, dim, seed=wseed, rank=rank, kind=args.kind)
        nq_total = nq * world

    ctx = pm.Context(local_rank)
    ctx.set_option(pm.api.PM_OPT_RANSAC_PATH, args.ransac_path)
    stream = torch.cuda.Stream(device=dev)      # a real (non-null) stream shared by torch and the library
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    # a second context + stream: the exchange / RANSAC half of a pipelined step
    ctx_b, stream_b = [], []
    for _ in range(1):
        cb = pm.Context(local_rank)
        cb.set_option(pm.api.PM_OPT_RANSAC_PATH, args.ransac_path)
        sb = torch.cuda.Stream(device=dev)
        cb.set_stream(sb.cuda_stream)
        ctx_b.append(cb)
        stream_b.append(sb)
    pipelined = multi if args.pipeline < 0 else bool(args.pipeline)

    d_q = torch.from_numpy(np.ascontiguousarray(w["q"])).to(dev)
    d_t = torch.from_numpy(np.ascontiguousarray(w["t"])).to(dev)
    d_kp1 = torch.from_numpy(np.ascontiguousarray(w["kp1"])).to(dev)
    d_kp2 = torch.from_numpy(np.ascontiguousarray(w["kp2"])).to(dev)
    n_all_max = nq * world
    hb, he = shard.hyp_shard(H, rank, world)

    class Slot:
        """Everything one in-flight image pair owns between the matcher and the end of RANSAC.  One contiguous
        "survivor block" per rank: [count (int32) + 3 pad words | xy1: nq x 2 f32 | xy2: nq x 2 f32]; the filter writes
        straight into it.  N > 1: the block is this rank's row of the gathered buffer and the record this rank's row of
        the gathered records, so both all-gathers run in place (no send-side copy) and RANSAC reads the gathered blocks
        through a pm_points_view (no concatenation pass)."""
        def __init__(self):
            self.knn = torch.empty((nq, K, 4), dtype=torch.int32, device=dev)
            self.good = torch.empty((nq, 4), dtype=torch.int32, device=dev)
            self.g_blk = shard.gathered_blocks(world, nq, dev) if multi else None
            self.blk, self.n, self.xy1, self.xy2 = shard.survivor_block(nq, dev, into=self.g_blk[rank] if multi else None)
            self.key = torch.zeros(1, dtype=torch.int64, device=dev)
            self.F = torch.zeros(9, dtype=torch.float64, device=dev)
            self.mask = torch.zeros(n_all_max, dtype=torch.uint8, device=dev)
            self.ninl = torch.zeros(1, dtype=torch.int32, device=dev)
            self.ntot = torch.zeros(1, dtype=torch.int32, device=dev)
            if multi:
                self.g_rec = torch.zeros((world, 10), dtype=torch.float64, device=dev)
                self.rec = self.g_rec[rank]                                  # pm_ransac_record: key + F[9]
                self.view = shard.view_of_blocks(self.g_blk, nq)
            self.ev_match = torch.cuda.Event()
            self.ev_done = torch.cuda.Event()
            self.used = False

    slots = [Slot(), Slot()]
    S0 = slots[0]
    d_knn, d_good, d_n, d_xy1, d_xy2, d_key, d_F, d_mask, d_ninl, d_ntot = (S0.knn, S0.good, S0.n, S0.xy1, S0.xy2, S0.key, S0.F,
                                                                            S0.mask, S0.ninl, S0.ntot)
    g_blk = S0.g_blk

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    def match(c, S):
        if hamming:
            c.bf_knn_hamming_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, K, S.knn.data_ptr())
            c.filter_ratio_gather_dev(S.knn.data_ptr(), nq, K, ratio, d_kp1.data_ptr(), d_kp2.data_ptr(),
                                      S.good.data_ptr(), S.xy1.data_ptr(), S.xy2.data_ptr(), S.n.data_ptr())
        else:       # 2-NN + ratio test + compaction + keypoint gather: one call (matcher launches, then the filter launch)
            c.bf_knn_l2_ratio_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, knn_flags, ratio, d_kp1.data_ptr(),
                                  d_kp2.data_ptr(), S.knn.data_ptr(), S.good.data_ptr(), S.xy1.data_ptr(), S.xy2.data_ptr(),
                                  S.n.data_ptr())

    def rest(c, S):
        if multi:
            dist.all_gather_into_tensor(S.g_blk.view(-1), S.blk)             # exchange 1: survivor blocks
            c.ransac_shard_parts_dev(S.view, hb, he, thresh, seed, S.rec.data_ptr())
            dist.all_gather_into_tensor(S.g_rec.view(-1), S.rec)             # exchange 2: 80-byte (key, F) records
            c.ransac_finish_parts_dev(S.view, thresh, S.g_rec.data_ptr(), world, S.key.data_ptr(), S.F.data_ptr(),
                                      S.mask.data_ptr(), n_all_max, S.ninl.data_ptr(), S.ntot.data_ptr())
        else:
            c.ransac_run_dev(S.xy1.data_ptr(), S.xy2.data_ptr(), nq, S.n.data_ptr(), hb, he, thresh, seed,
                             S.key.data_ptr(), S.F.data_ptr(), S.mask.data_ptr(), S.ninl.data_ptr())

    def step(e=None):
        """One pair, strictly serial on one stream (the form the stage brackets are taken on)."""
        if e:
            e[0].record(stream)
        match(ctx, S0)
        if e:
            e[1].record(stream)
        rest(ctx, S0)
        if e:
            e[2].record(stream)

    def step_pipe(i):
        """Pair i of a pipelined run: its matcher on `stream`, its exchanges + RANSAC on a second stream; pair i+1's matcher
        is enqueued behind this one's on `stream` and runs while this pair is in its collectives.  Two slots; a slot is
        re-used only after the pair that held it has finished (ev_done).  (Three pairs in flight over three streams measured
        SLOWER from this Python harness — 83.7 against 72.6 us per step at world 1: every torch.distributed call costs the
        host ~20 us, so the deeper pipeline is host-bound; the C ABI's pm_mgpu_submit_dev has a host thread per device.)"""
        S = slots[i & 1]
        sb, cb = stream_b[0], ctx_b[0]
        if S.used:
            stream.wait_event(S.ev_done)
        match(ctx, S)
        S.ev_match.record(stream)
        with torch.cuda.stream(sb):
            sb.wait_event(S.ev_match)
            rest(cb, S)
            S.ev_done.record(sb)
        S.used = True

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n):
        fence()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        fence()
        return time.perf_counter() - t0

    # The contract-timed region: exactly K steps, nothing but the path's own launches on the stream(s).  (Event records
    # between the stages are not free on this hardware — each is a barrier packet that ends the overlap of one kernel's
    # launch with its predecessor's tail — so the stage split is taken in a second, serial pass of the same K steps, and
    # `value`, the matcher stage's rate, comes from that instrumented pass: the conservative one.)
    for i in range(args.warmup):
        step()
        step_pipe(i)
    # Which of the two step forms the timed region reports is decided HERE, in the warm-up, not after the fact: with
    # --pipeline -1 (default) and more than one rank (or --exercise-exchange) both forms run a trial of the same length
    # and every rank takes the form whose slowest rank was faster (the pipelined form depends on the host keeping two
    # streams fed; on some boxes the serial form wins).
    form_trial = None
    if multi and args.pipeline < 0:
        n_trial = max(20, args.warmup)
        tr = torch.tensor([timed(lambda i: step(), n_trial), timed(step_pipe, n_trial)], dtype=torch.float64, device=dev)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        pipelined = bool(tr[1].item() < tr[0].item())
        form_trial = {"steps": n_trial, "serial_ms_per_step": tr[0].item() / n_trial * 1e3,
                      "pipelined_ms_per_step": tr[1].item() / n_trial * 1e3}
    wall_serial = timed(lambda i: step(), args.steps)
    wall_pipe = timed(step_pipe, args.steps)
    for i in range(args.steps):
        step(ev[i])
    fence()
    match_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    rest_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    if multi:
        tt = torch.tensor([wall_serial, wall_pipe, match_ms, rest_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall_serial, wall_pipe, match_ms, rest_ms = [float(x) for x in tt.tolist()]
    wall = wall_pipe if pipelined else wall_serial
    ms_per_step = wall / args.steps * 1e3

    # ---- every collective by itself: back-to-back all-gathers of the step's own buffers between two events
    collectives = None
    if multi:
        def coll_us(fn, reps=100):
            for _ in range(5):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fence()
            a.record(stream)
            for _ in range(reps):
                fn()
            b.record(stream)
            fence()
            us = a.elapsed_time(b) / reps * 1e3
            tt = torch.tensor([us], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        collectives = {"allgather_survivor_blocks_us": coll_us(lambda: dist.all_gather_into_tensor(S0.g_blk.view(-1), S0.blk)),
                       "survivor_block_bytes_per_rank": int(S0.blk.numel() * 4),
                       "allgather_records_us": coll_us(lambda: dist.all_gather_into_tensor(S0.g_rec.view(-1), S0.rec)),
                       "record_bytes_per_rank": 80, "ranks": world, "backend": args.backend,
                       "note": "microseconds per collective, issued back to back on the step's stream (max over ranks)"}
        step()                                       # leave slot 0 in the state of a whole step
        fence()

    n_m = int((d_ntot if multi else d_n).item())
    key = int(d_key.item())
    n_inl = int(d_ninl.item())
    both = all(int(S.key.item()) == key and int(S.ninl.item()) == n_inl for S in slots if S.used)   # every slot saw the same pair

    # ---- sustained leg: the K timed steps above last a few milliseconds, during which the chip still holds its
    # boost clock; >= --sustain-seconds of back-to-back steps show the figure it sustains (DVFS give-back)
    sustained = None
    if args.sustain_seconds > 0:
        n_sus = max(args.steps, int(args.sustain_seconds / max(ms_per_step * 1e-3, 1e-6)) + 1)
        t_sus = timed(step_pipe if pipelined else (lambda i: step()), n_sus)
        es = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
        for e in es:                                 # stage split right behind the sustained run (clocks still settled)
            step(e)
        fence()
        m_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in es]))
        if multi:
            tt = torch.tensor([t_sus, m_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_sus, m_ms = [float(x) for x in tt.tolist()]
        sustained = {"seconds": t_sus, "steps": n_sus, "ms_per_step": t_sus / n_sus * 1e3, "match_ms": m_ms,
                     "value": float(nq_total) * nt / (m_ms * 1e-3), "pipelined": pipelined}

    # ---- per-kernel durations: instrumented replay of the same K steps (hipEvents on the stream
    # the kernels run on, recorded inside the library around each launch)
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(args.steps):
        step()
    fence()
    kern = {}
    for name in ("knn_l2_prep", "knn_l2_mfma_u8", "knn_l2_mfma_f16s", "knn_l2_mfma_f16", "knn_l2_mfma", "knn_l2_refine", "knn_hamming_expand",
                 "knn_hamming_mfma_i8", "knn_hamming_refine", "knn_hamming", "knn_hamming_merge", "filter_gather", "ransac_solve", "ransac_score",
                 "ransac_select", "ransac_final", "concat_points", "ransac_fused", "ransac_finish"):
        ms, cnt = ctx.timing_get(name)
        if cnt:
            kern[name] = round(ms * 1e3, 2)           # microseconds
    ctx.timing_enable(False)
    # diagnostics of the coarse/refine split (a kNN call on its own, so the arena still holds them)
    kstats, f32_route_us = None, 0.0
    if not hamming:
        ctx.knn_diag_enable(True)
        ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, K, d_knn.data_ptr(), knn_flags)
        kstats = ctx.knn_stats()
        ctx.knn_diag_enable(False)
        # the f32-input coarse pass (PM_KNN_FORCE_F32) on the same data, for its own roofline line
        if not args.headline_only:
            ctx.timing_enable(True)
            ctx.timing_reset()
            for _ in range(args.steps):
                ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, K, d_knn.data_ptr(),
                                  pm.api.PM_KNN_FORCE_F32)
            fence()
            f32_route_us = ctx.timing_get("knn_l2_mfma")[0] * 1e3
            ctx.timing_enable(False)
        ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, K, d_knn.data_ptr(), knn_flags)
        fence()

    # ---- the same matcher stage on GENERAL floats (SURF-like unit-norm descriptors, what main.cpp:37-40 produces):
    # automatic route -> the f16 matrix pass on f16-ROUNDED scaled copies, refinement window widened by the rounding
    # bound (docs/SPEC.md S1c; results canonical).  Reported next to `value` (which is the u8-valued SIFT case).
    general = None
    if not hamming and args.kind == "sift" and args.workload in ("c2", "c3") and not args.headline_only:
        wg = synth.pair_workload(nq, nt, dim, seed=wseed, rank=rank, kind="surf")
        g_q = torch.from_numpy(np.ascontiguousarray(wg["q"])).to(dev)
        g_t = torch.from_numpy(np.ascontiguousarray(wg["t"])).to(dev)
        g_n = torch.zeros(1, dtype=torch.int32, device=dev)

        def gstep(gflags=0):
            ctx.bf_knn_l2_ratio_dev(g_q.data_ptr(), nq, g_t.data_ptr(), nt, dim, gflags, ratio, d_kp1.data_ptr(), d_kp2.data_ptr(),
                                    d_knn.data_ptr(), d_good.data_ptr(), d_xy1.data_ptr(), d_xy2.data_ptr(), g_n.data_ptr())

        def gtime(gflags):
            for _ in range(args.warmup):
                gstep(gflags)
            fence()
            ge = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ge[0].record(stream)
            for _ in range(args.steps):
                gstep(gflags)
            ge[1].record(stream)
            fence()
            ms = ge[0].elapsed_time(ge[1]) / args.steps
            if multi:
                tt = torch.tensor([ms], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ms = float(tt.item())
            return ms
        g_ms = gtime(0)                                       # no hint: the automatic route (two prep launches)
        g_ms_unit = gtime(pm.api.PM_KNN_HINT_UNIT_NORM)       # the caller states ||t|| <= 1 (SURF is): one prep launch
        g_n_unit = int(g_n.item())
        ctx.knn_diag_enable(True)
        gstep()
        g_st = ctx.knn_stats()
        ctx.knn_diag_enable(False)
        general = {"value": float(nq_total) * nt / (g_ms * 1e-3), "match_ms": g_ms, "matches": int(g_n.item()),
                   "unit_norm_hint": {"value": float(nq_total) * nt / (g_ms_unit * 1e-3), "match_ms": g_ms_unit, "matches": g_n_unit,
                                      "note": "PM_KNN_HINT_UNIT_NORM: same records, one prep launch (the bound is verified on the device)"},
                   "descriptors": "surf (unit-norm general floats), automatic route",
                   "coarse_pass": {0: "f16 MFMA, exact integer copies", 1: "f16 MFMA on f16-rounded scaled copies (SPEC S1c)",
                                   2: "f32-input MFMA"}.get(g_st["route"], "?"),
                   "refine": {"rescans": g_st["rescans"], "nonfinite": g_st["nonfinite"]}}
        # leave the arena and the survivor block in the state of the headline workload
        step()
        fence()

    # ---- the same SIFT data with NO hint (automatic route: the device finds the data integer-valued and takes the f16 pass;
    # two more near-empty launches than the hinted call) — `value_auto_route`
    auto_route = None
    if not hamming and args.kind == "sift" and knn_flags and not args.headline_only:
        a_n = torch.zeros(1, dtype=torch.int32, device=dev)

        def astep():
            ctx.bf_knn_l2_ratio_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, 0, ratio, d_kp1.data_ptr(), d_kp2.data_ptr(),
                                    d_knn.data_ptr(), d_good.data_ptr(), d_xy1.data_ptr(), d_xy2.data_ptr(), a_n.data_ptr())
        for _ in range(args.warmup):
            astep()
        fence()
        ae = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ae[0].record(stream)
        for _ in range(args.steps):
            astep()
        ae[1].record(stream)
        fence()
        a_ms = ae[0].elapsed_time(ae[1]) / args.steps
        if multi:
            tt = torch.tensor([a_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            a_ms = float(tt.item())
        auto_route = {"value": float(nq_total) * nt / (a_ms * 1e-3), "match_ms": a_ms, "matches": int(a_n.item()),
                      "note": "same data, flags = 0: prep16 + prep16g (returns at once) + f16 pass + refinement + filter"}
        step()
        fence()

    # ---- the matcher at 32k x 32k SIFT-128 (the size of BASELINE config 4's matrices, on the L2 path): the coarse kernel's
    # roofline where launch and prologue no longer dominate
    large = None
    if not hamming and args.kind == "sift" and world == 1 and not args.no_large and not args.headline_only and args.workload in ("c2", "c3"):
        nL = 32768
        wl = synth.sift_like(nL, nL, dim, seed=0x32)
        l_q, l_t = torch.from_numpy(wl[0]).to(dev), torch.from_numpy(wl[1]).to(dev)
        l_out = torch.empty((nL, K, 4), dtype=torch.int32, device=dev)
        for _ in range(3):
            ctx.bf_knn_l2_dev(l_q.data_ptr(), nL, l_t.data_ptr(), nL, dim, K, l_out.data_ptr(), knn_flags)
        fence()
        le = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        reps = 20
        le[0].record(stream)
        for _ in range(reps):
            ctx.bf_knn_l2_dev(l_q.data_ptr(), nL, l_t.data_ptr(), nL, dim, K, l_out.data_ptr(), knn_flags)
        le[1].record(stream)
        fence()
        call_us = le[0].elapsed_time(le[1]) / reps * 1e3
        ctx.timing_enable(True)
        ctx.timing_reset()
        for _ in range(reps):
            ctx.bf_knn_l2_dev(l_q.data_ptr(), nL, l_t.data_ptr(), nL, dim, K, l_out.data_ptr(), knn_flags)
        fence()
        lk = {n: round(ctx.timing_get(n)[0] * 1e3, 2) for n in ("knn_l2_prep", "knn_l2_mfma_u8", "knn_l2_mfma_f16", "knn_l2_mfma_f16s", "knn_l2_refine")
              if ctx.timing_get(n)[1]}
        ctx.timing_enable(False)
        got = l_out[5000:5128].cpu().numpy().view(pm.MATCH_DTYPE).reshape(128, K)
        large = {"nq": nL, "nt": nL, "call_us": call_us, "kernels_us": lk, "rows_checked": None}
        if not args.no_verify:
            from oracle import pm_oracle as O
            want = O.bf_knn_l2(wl[0][5000:5128], wl[1], K, nthreads=8)
            large["rows_checked"] = 128
            large["parity"] = "ok" if ((got["trainIdx"] == want["trainIdx"]).all() and
                                       (got["distance"].view(np.uint32) == want["distance"].view(np.uint32)).all()) else "MISMATCH"
        del l_q, l_t, l_out
        ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, K, d_knn.data_ptr(), knn_flags)
        step()
        fence()

    # ---- parity spot check against the CPU oracle (untimed; checker only)
    parity = "skipped"
    if not args.no_verify and rank == 0:
        from oracle import pm_oracle as O
        got = d_knn.cpu().numpy().view(pm.MATCH_DTYPE).reshape(nq, K)
        rows = np.random.default_rng(0).permutation(nq)[:256]
        want = (O.bf_knn_hamming if hamming else O.bf_knn_l2)(w["q"][rows], w["t"], K, nthreads=8)
        ok = (got["trainIdx"][rows] == want["trainIdx"]).all() and \
             (got["distance"][rows].view(np.uint32) == want["distance"].view(np.uint32)).all()
        if multi:                                  # the concatenation the view stands for, done here for the checker only
            xs1, xs2, _ = shard.concat_blocks(g_blk, nq)
        else:
            xs1 = d_xy1[:n_m].cpu().numpy()
            xs2 = d_xy2[:n_m].cpu().numpy()
        rc, F_o, mask_o, ninl_o, key_o = O.ransac_fundamental(xs1, xs2, H, thresh, seed, nthreads=8)
        ok = ok and key_o == key and ninl_o == n_inl and (mask_o == d_mask[:n_m].cpu().numpy()).all() \
            and (F_o.reshape(9).view(np.uint64) == d_F.cpu().numpy().view(np.uint64)).all()
        parity = "ok" if ok else "MISMATCH"

    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)

    pairs = float(nq_total) * nt
    value = pairs / (match_ms * 1e-3)
    hyp_per_s = H / (rest_ms * 1e-3)

    out = {
        "metric": "descriptor-pair distances/s (BF-%s 2-NN + ratio stage); RANSAC hypotheses/s in `ransac`"
                  % ("Hamming" if hamming else "L2"),
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "u8" if hamming else "f32",
        "dtype_note": "u8 bit strings; coarse pass i8xi8->i32 MFMA on +-1 expanded bits (exact), refinement u32 xor/popcount"
                      if hamming else ("f32 in/out; coarse pass i8 x i8 -> i32 MFMA on x - 128 (exact for u8-valued data), integer "
                                       "refinement on the same byte copies, distances = canonical f32 bits"
                                       if knn_flags == pm.api.PM_KNN_HINT_U8 else
                                       "f32 in/out; coarse pass f16xf16->f32 MFMA (exact for u8-valued data), refinement f32"
                                       if knn_flags else "f32 in/out; coarse pass on f32-input MFMA (or the exact f16 one when the "
                                                         "device finds the data integer-valued), refinement f32"),
        "data": "synthetic",
        "config": {"workload": "%s: %dx%d %s BF-%s 2-NN + ratio 0.8 + %d-hypothesis RANSAC-F (8-point, "
                               "Sampson, tau=1px) per image pair; N>1: query rows and hypothesis ids sharded"
                               % (args.workload.upper(), nq_total if args.scaling == "strong" else nq, nt,
                                  "ORB-%d binary" % (8 * dim) if hamming else "SIFT-%d f32" % dim,
                                  "Hamming" if hamming else "L2", H), "descriptors": args.kind, "k": K,
                   "coarse_route": "n/a (Hamming)" if hamming else
                                   {pm.api.PM_KNN_HINT_U8: "i8-MFMA on x - 128 (u8 hint, device-verified)",
                                    pm.api.PM_KNN_HINT_INTEGER: "f16-MFMA (integer hint, device-verified)",
                                    pm.api.PM_KNN_HINT_UNIT_NORM: "f16-MFMA on rounded scaled copies (unit-norm hint, device-verified: one prep launch)"
                                    }.get(knn_flags, "auto")},
        "ms_per_step_serial": wall_serial / args.steps * 1e3, "ms_per_step_pipelined": wall_pipe / args.steps * 1e3,
        "step_form": ("pipelined: pair i+1's matcher on a second stream while pair i is in its exchanges / RANSAC" if pipelined
                      else "serial: one pair after the other on one stream"),
        "stage_ms": {"match": match_ms, "ransac_and_exchange": rest_ms,
                     "note": "hipEvent brackets in a second pass of the same K steps (the events themselves add "
                             "~4-5 us per bracket); ms_per_step is the un-instrumented pass"},
        "ransac": {"hyp_per_s": hyp_per_s, "hypotheses": H, "n_matches": n_m, "inliers": n_inl,
                   "best_hyp": pm.api.ransac_key_hyp(key) if key else None},
        "kernels_us": kern, "knn_refine": kstats, "parity": parity,
    }
    if form_trial:
        out["step_form_trial"] = form_trial
    if args.single_device and world > 1:
        out["config"]["timing_note"] = ("%d processes share ONE GPU (--single-device): the device time-slices between processes at "
                                        "millisecond granularity and event brackets include the other process's slices; this line "
                                        "demonstrates parity of the N > 1 path, its timings are not performance data" % world)
    if collectives:
        out["collectives"] = collectives
    if not both:
        out["parity"] = "MISMATCH (the pipeline slots disagree)"
    if general:
        out["value_general_floats"] = general["value"]
        out["value_general_floats_unit_norm_hint"] = general["unit_norm_hint"]["value"]
        out["general_floats"] = general
    if sustained:
        out["sustained"] = sustained
    # roofline of the dominant kernel (algorithmic 2*D flop per descriptor pair)
    flops = 2.0 * dim * nq * nt
    if hamming and "knn_hamming_mfma_i8" in kern:
        # +-1 byte expansion: dot = 256 - 2*hamming, 2*256 integer ops per pair on v_mfma_i32_32x32x32_i8
        # (dense I8 peak = 2x the BF16 peak: same cycles at twice the K, MI355X_MICROARCH.md MFMA table)
        ops = 2.0 * 8 * dim * nq * nt
        ach = ops / (kern["knn_hamming_mfma_i8"] * 1e-6) / 1e12
        out["roofline"] = {"kernel": "knn_hamming_mfma_i8 (knn_mfma_rows288<RouteI8>)", "bound": "mfma", "achieved": ach,
                           "peak": 2 * PEAK_F16_MFMA_TFLOPS, "unit": "TOP/s", "frac": ach / (2 * PEAK_F16_MFMA_TFLOPS),
                           "traffic": pmc_traffic("c4:knn_hamming_mfma_i8", nq, nt),
                           "algorithmic_bytes": (nq + nt) * dim + nq * K * 16,
                           "dtype": "i8 (+-1 expanded bits) x i8 -> i32 MFMA; exact",
                           "note": "ablation (DESIGN.md 2.1): MFMA skeleton 76% of the kernel, staging 12%, grouped (8-row) "
                                   "selection 3%; every issued MFMA is algorithmic work (no seed chunk)"}
    elif hamming and "knn_hamming" in kern:
        # 16 integer VALU ops per pair (8 x v_xor_b32 + 8 x v_bcnt_u32_b32 at 32 B); one wave64 VALU
        # instruction occupies its SIMD 4 cycles: 1024 SIMDs * 64 lanes * 2.4 GHz / 4 = 3.93e13 lane-ops/s
        ops = 16.0 * nq * nt
        ach = ops / (kern["knn_hamming"] * 1e-6) / 1e12
        out["roofline"] = {"kernel": "knn_hamming_scan", "bound": "valu-int", "achieved": ach, "peak": 39.3,
                           "unit": "Tlane-op/s", "frac": ach / 39.3, "traffic": None,
                           "dtype": "u32 xor + popcount on the VALU (v_xor_b32, v_bcnt_u32_b32)"}
    elif "knn_l2_mfma_u8" in kern:
        ach = flops / (kern["knn_l2_mfma_u8"] * 1e-6) / 1e12
        out["roofline"] = {"kernel": "knn_l2_mfma_u8 (knn_mfma_rows288<RouteU8>)", "bound": "mfma", "achieved": ach,
                           "peak": 2 * PEAK_F16_MFMA_TFLOPS, "unit": "TOP/s", "frac": ach / (2 * PEAK_F16_MFMA_TFLOPS),
                           "frac_of_f16_peak": ach / PEAK_F16_MFMA_TFLOPS,
                           "traffic": pmc_traffic("c3:knn_l2_mfma_u8", nq, nt),
                           "algorithmic_bytes": (nq + nt) * dim * 4 + nq * K * 16,
                           "dtype": "i8 x i8 -> i32 MFMA (v_mfma_i32_32x32x32_i8) on x - 128, exact for u8-valued data; every "
                                    "issued MFMA is algorithmic work (the row term starts the accumulators)"}
    elif "knn_l2_mfma_f16s" in kern:
        ach = flops / (kern["knn_l2_mfma_f16s"] * 1e-6) / 1e12
        out["roofline"] = {"kernel": "knn_l2_mfma_f16s (knn_mfma_rows288<RouteF16S>)", "bound": "mfma", "achieved": ach,
                           "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F16_MFMA_TFLOPS,
                           "traffic": pmc_traffic("c3:knn_l2_mfma_f16s", nq, nt),
                           "algorithmic_bytes": (nq + nt) * dim * 4 + nq * K * 16,
                           "dtype": "f16-input MFMA, f32 accumulate (v_mfma_f32_32x32x16_f16), exact for integer data"}
    elif kern.get("knn_l2_mfma_f16", 0) > kern.get("knn_l2_mfma", 0):
        ach = flops / (kern["knn_l2_mfma_f16"] * 1e-6) / 1e12
        out["roofline"] = {"kernel": "knn_l2_mfma_f16 (knn_mfma_rows288<RouteF16>)", "bound": "mfma", "achieved": ach,
                           "peak": PEAK_F16_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": ach / PEAK_F16_MFMA_TFLOPS,
                           "traffic": pmc_traffic("c3:knn_l2_mfma_f16", nq, nt),
                           "algorithmic_bytes": (nq + nt) * dim * 4 + nq * K * 16,
                           "dtype": "f16-input MFMA, f32 accumulate (v_mfma_f32_32x32x16_f16), exact for u8-valued data",
                           "note": "launch/prologue-sized at C3: 16.7-17.0 us of the 19 us (hipEvent) remain with selection, staging, "
                                   "barrier and LDS operand reads all removed (ablation builds, DESIGN.md 2.1); 9.2 us of that is "
                                   "the bare chain of 288 MFMAs per wave at the 2.0 GHz the chip holds"}
    elif "knn_l2_mfma" in kern:
        ach = flops / (kern["knn_l2_mfma"] * 1e-6) / 1e12
        out["roofline"] = {"kernel": "knn_l2_mfma", "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                           "traffic": pmc_traffic("c3_f32:knn_l2_mfma", nq, nt),
                           "algorithmic_bytes": (nq + nt) * dim * 4 + nq * K * 16,
                           "dtype": "f32-input MFMA (v_mfma_f32_32x32x2_f32)"}
    if out.get("roofline", {}).get("traffic") is not None:
        out["roofline"]["traffic_source"] = "profiles/r03_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this kernel " \
                                            "on this workload, separate passes, gfx950 correction), stamped with the hash of " \
                                            "the kernel sources it was measured on: " + _kernel_source_sha()
    if auto_route:
        out["value_auto_route"] = auto_route["value"]
        out["auto_route"] = auto_route
    if large:
        lflops = 2.0 * dim * large["nq"] * large["nt"]
        for kn, peak, unit in (("knn_l2_mfma_u8", 2 * PEAK_F16_MFMA_TFLOPS, "TOP/s"), ("knn_l2_mfma_f16", PEAK_F16_MFMA_TFLOPS, "TFLOP/s"),
                               ("knn_l2_mfma_f16s", PEAK_F16_MFMA_TFLOPS, "TFLOP/s")):
            if kn in large["kernels_us"]:
                ach = lflops / (large["kernels_us"][kn] * 1e-6) / 1e12
                out["roofline_large"] = {"kernel": kn, "workload": "32768 x 32768 SIFT-128 f32, BF-L2 2-NN (matcher call only)",
                                         "bound": "mfma", "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                                         "frac_of_f16_peak": ach / PEAK_F16_MFMA_TFLOPS,
                                         "kernel_us": large["kernels_us"][kn], "call_us": large["call_us"],
                                         "kernels_us": large["kernels_us"], "pairs_per_s": float(large["nq"]) * large["nt"] / (large["call_us"] * 1e-6),
                                         "parity": large.get("parity", "skipped"), "traffic": pmc_traffic("l32k:" + kn, large["nq"], large["nt"])}
                break
    if f32_route_us > 0:
        ach = flops / (f32_route_us * 1e-6) / 1e12
        out["roofline_f32_route"] = {"kernel": "knn_l2_mfma", "bound": "mfma", "achieved": ach,
                                     "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                                     "kernel_us": round(f32_route_us, 2),
                                     "traffic": pmc_traffic("c3_f32:knn_l2_mfma", nq, nt),
                                     "dtype": "f32-input MFMA (v_mfma_f32_32x32x2_f32): PM_KNN_FORCE_F32, and the fallback of the "
                                              "automatic route when the rounded-copy pass withdraws"}
    rk = "ransac_fused" if "ransac_fused" in kern else "ransac_score"
    if rk in kern and n_m:
        flops = 34.0 * n_m * (he - hb)
        ach = flops / (kern[rk] * 1e-6) / 1e12
        out["roofline_ransac"] = {"kernel": rk + (" (sample + solve + score + pick + mask in one launch; the fp64 solves are "
                                                  "not credited)" if rk == "ransac_fused" else ""),
                                  "bound": "valu-f32", "achieved": ach,
                                  "peak": PEAK_F32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_VALU_TFLOPS,
                                  "flops_per_pair": 34, "n_matches": n_m, "traffic": None}

    if world == 1 and not args.no_cpu_baseline:
        from oracle import pm_oracle as O
        knn_cpu = O.bf_knn_hamming if hamming else O.bf_knn_l2
        try:
            avail = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            avail = os.cpu_count() or 1
        # the CPU share of this job: the cgroup quota when there is one (a 1-GPU job on the pool gets 16 cores of a
        # 256-thread host; running 256 OpenMP threads inside that quota is 4x SLOWER than 16), else the affinity mask
        quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = max(1, int(float(q) / float(per) + 0.5))
        except (OSError, ValueError):
            try:
                q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    quota = max(1, int(q / per + 0.5))
            except (OSError, ValueError):
                pass
        share = min(avail, quota) if quota else avail
        # all-cores leg: the job's share, and (when the mask is wider than that and no quota is visible) a probe of a
        # few thread counts, keeping the fastest — the number reported is the best the host cores gave this job
        cands = sorted({share} | ({16, 32, 64} if quota is None and avail > 16 else set()))
        cands = [c for c in cands if c <= avail]
        nth = cands[0]

        def best_of(fn, reps):
            best = None
            for _ in range(reps):
                t0 = time.perf_counter()
                r = fn()
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
            return best, r
        # 1 thread: the FULL matcher and the full RANSAC run (bounded: a few seconds at C3; C4's 1e9 pairs are sampled)
        sample_q = nq if float(nq) * nt <= 1.5e8 * 20 else max(256, int(1.5e8 * 20 / nt))
        t_knn, _ = best_of(lambda: knn_cpu(w["q"][:sample_q], w["t"], K, nthreads=1), 1)
        t_knn_all, knn_all = None, None
        for c in cands:
            tc, rc_ = best_of(lambda c=c: knn_cpu(w["q"], w["t"], K, nthreads=c), 3)
            if t_knn_all is None or tc < t_knn_all:
                t_knn_all, knn_all, nth = tc, rc_, c
        good = O.filter_ratio(knn_all, ratio)
        xs1 = O.gather_points(w["kp1"], good["queryIdx"])
        xs2 = O.gather_points(w["kp2"], good["trainIdx"])
        Hs = H if float(H) * max(good.size, 1) <= 4e9 else max(1000, int(4e9 / max(good.size, 1)))
        t_r, _ = best_of(lambda: O.ransac_fundamental(xs1, xs2, Hs, thresh, seed, nthreads=1), 1)
        t_r_all, _ = best_of(lambda: O.ransac_fundamental(xs1, xs2, H, thresh, seed, nthreads=nth), 3)
        try:
            cpu_model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
        except (OSError, IndexError):
            cpu_model = "unknown"
        out["cpu_baseline"] = {
            "value": sample_q * nt / t_knn, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": "oracle (CPU restatement of the path, gcc -O3 AVX2, 1 thread): matcher on %d of %d query rows x %d "
                      "train rows; RANSAC-F on %d of %d hypotheses x %d matches"
                      % (sample_q, nq, nt, Hs, H, good.size),
            "ransac_hyp_per_s": Hs / t_r, "host_cpus": os.cpu_count(), "affinity_cpus": avail, "cgroup_cpu_quota": quota,
            "cpu_model": cpu_model,
            "all_cores": {"value": float(nq) * nt / t_knn_all, "ransac_hyp_per_s": H / t_r_all, "cores": nth,
                          "sample": "full %d x %d matcher and the full RANSAC run, OpenMP over query rows / hypothesis ids, "
                                    "best of 3, %d threads (the job's CPU share: cgroup quota if any, else the fastest of %s threads)"
                                    % (nq, nt, nth, cands)},
        }
    print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
