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
