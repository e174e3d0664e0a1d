"""CPU, world_size 2 over gloo: the N>1 path of the eval harness (clip sharding + the single
all-reduce of metric sums).  The per-clip compute is stubbed with a deterministic function of the
clip index because the real compute is GPU-only; what is under test is that every clip is
processed exactly once and that the reduced means equal the single-process result."""
import json
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def fake_clip(c):
    # two counted frames per clip with clip-dependent metric values
    return [(30.0 + c, 31.0 + 0.5 * c), (29.0 + 0.25 * c, 30.5)]


def worker(rank, world, port, n_clips, q):
    sys.path.insert(0, ROOT)
    from crfp_amd import evalrig
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = evalrig.evaluate(fake_clip, n_clips, rank, world, dist)
    owned = evalrig.shard_clips(n_clips, rank, world)
    q.put((rank, res, owned))
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_eval_matches_single_process():
    from crfp_amd import evalrig
    n_clips, world = 7, 2
    single = evalrig.evaluate(fake_clip, n_clips)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = sorted(sum((g[2] for g in got), []))
    assert owned == list(range(n_clips))                      # every clip exactly once
    for _, res, _ in got:                                     # every rank holds the global result
        assert res["frames"] == single["frames"] == 2 * n_clips
        assert abs(res["psnr"] - single["psnr"]) < 1e-12
        assert abs(res["psnr_y"] - single["psnr_y"]) < 1e-12


def bench_worker(rank, world, port, q):
    """bench.py's multi-rank bookkeeping (crfp_amd.benchutil) with the compute stubbed: each rank 'measures' its own
    elapsed time and squared-error sums; rank results must combine exactly as bench.py reports them."""
    sys.path.insert(0, ROOT)
    from crfp_amd import benchutil
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    clips, t, steps = 4, 7, 5
    seeds = benchutil.rank_clip_seeds(rank, clips)
    elapsed = benchutil.reduce_elapsed(0.10 + 0.03 * rank, dist)            # the slower rank defines the step
    vec = benchutil.reduce_sums(torch.tensor([100.0 + rank, 10.0 * (rank + 1), float(t * clips)], dtype=torch.float64), dist)
    agg = benchutil.aggregate(world, steps, t * clips, elapsed)
    line = benchutil.contract_line(agg, world, steps, 2, "bf16", {"workload": "stub", "clips_per_gpu_per_step": clips}, dist)
    line["psnr_reduce"] = benchutil.psnr_reduce_record(vec, world)
    q.put((rank, seeds, elapsed, vec.tolist(), agg, line))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_reduction_path_two_ranks():
    from crfp_amd import benchutil
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_seeds = sum((g[1] for g in got), [])
    assert len(set(all_seeds)) == 8 and all_seeds == list(range(1234, 1242))      # 2 ranks x 4 clips, all distinct
    for _, _, elapsed, vec, agg, line in got:
        # the JSON line an N > 1 run prints: the driver's contract keys, whole-job value, RCCL's N visible
        assert all(k in line for k in benchutil.CONTRACT_KEYS) and json.dumps(line)
        assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] == agg["value"] and line["dtype"] == "bf16"
        assert line["collectives"] == {"backend": "gloo", "initialised": True, "world_size": 2, "ops": line["collectives"]["ops"]}
        assert line["psnr_reduce"]["ranks"] == 2 and line["psnr_reduce"]["frames"] == 56.0
        assert abs(elapsed - 0.13) < 1e-12                                          # MAX over ranks
        assert vec == [201.0, 30.0, 56.0]                                           # SUM over ranks
        assert abs(agg["value"] - 2 * 5 * 28 / 0.13) < 1e-9 and abs(agg["per_gpu_frames_per_sec"] - 5 * 28 / 0.13) < 1e-9
        assert abs(agg["ms_per_step"] - 26.0) < 1e-9
