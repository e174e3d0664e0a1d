"""The multi-rank bookkeeping of bench.py, kept importable without a GPU so that the N > 1 path can be exercised with
world-size-2 gloo processes on CPU (tests/test_sharding_gloo.py): which clips a rank owns, the MAX-over-ranks step time, the
SUM all-reduce of the PSNR sums (the only collective of the path) and the whole-job aggregate."""
from __future__ import annotations

from typing import List, Optional

import torch


def rank_clip_seeds(rank: int, clips_per_gpu: int, base: int = 1234) -> List[int]:
    """Every rank (and every clip of a rank) gets its own synthetic clip: independent units, no data-path collective."""
    return [base + rank * clips_per_gpu + c for c in range(clips_per_gpu)]


def reduce_elapsed(elapsed: float, dist, device: Optional[torch.device] = None) -> float:
    """MAX over ranks of the timed region (the slowest rank defines the step)."""
    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item())


def reduce_sums(vec: torch.Tensor, dist) -> torch.Tensor:
    """SUM over ranks of [sum sq err, sum sq err (Y), frames] -- RCCL all-reduce over xGMI on the GPU box, gloo in the test."""
    if dist is not None:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return vec


def aggregate(world: int, steps: int, frames_per_step_per_gpu: int, elapsed: float) -> dict:
    """bench.py's headline numbers: `value` is the WHOLE-JOB rate over all ranks (weak scaling: per-GPU work fixed)."""
    return {"value": world * steps * frames_per_step_per_gpu / elapsed,
            "per_gpu_frames_per_sec": steps * frames_per_step_per_gpu / elapsed,
            "ms_per_step": 1e3 * elapsed / steps}


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config")


def contract_line(agg: dict, world: int, steps: int, warmup: int, storage: str, config: dict, dist=None) -> dict:
    """The driver-contract part of bench.py's JSON line, identical for N = 1 and N > 1 (rank 0 adds roofline / cpu_baseline /
    psnr_reduce on top at every world size)."""
    return {
        "metric": "sr_frames_per_sec", "value": agg["value"], "unit": "frames/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": agg["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if storage == "f32" else "bf16",
        "data": "synthetic", "config": config,
        "per_gpu_frames_per_sec": agg["per_gpu_frames_per_sec"],
        "collectives": {"backend": dist.get_backend() if dist is not None else None, "initialised": dist is not None,
                        "world_size": dist.get_world_size() if dist is not None else 1,
                        "ops": ["barrier", "all_reduce(MAX) of the step time", "all_reduce(SUM) of the PSNR sums"]},
    }


def psnr_reduce_record(vec, world: int) -> dict:
    """What the one data-path-free collective produced: SUMs over all ranks; `frames` = frames per step over the whole job."""
    return {"sum_sq_err": float(vec[0]), "sum_sq_err_y": float(vec[1]), "frames": float(vec[2]), "ranks": world}
