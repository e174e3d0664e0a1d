"""CPU, world_size 2 over gloo: the N>1 path of the eval harness (clip sharding + the single
all-reduce of metric sums).  The per-clip compute is stubbed with a deterministic function of the
clip index because the real compute is GPU-only; what is under test is that every clip is
processed exactly once and that the reduced means equal the single-process result."""
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
