"""Eval harness counterpart of the reference's ``Trainer.eval_basicvsr`` (trainer.py:295-413) and the
clip sharding used for multi-GPU runs.

Quirks reproduced from the reference: per-frame metrics with an all-ones mask (:348), frame 0 of
every 50th batch skipped (:350-351), RGB PSNR on the tensors as they are and PSNR-Y on the luma
``24.966*c0 + 128.553*c1 + 65.481*c2 + 16`` (utils.py:328-330, BGR weights applied to RGB-ordered data,
kept as is) after the data-dependent range rule of utils.py:244-250, arithmetic mean of per-frame
values.  Clips are independent: clip c runs on rank c mod world; the only collective is the final
sum of (sum_psnr, sum_ssim, sum_psnr_y, sum_ssim_y, n_frames).
"""
from __future__ import annotations

import math
from typing import Callable, Iterable, List, Optional, Sequence

import torch


def shard_clips(n_clips: int, rank: int, world: int) -> List[int]:
    """Clip indices owned by `rank` (round-robin)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_clips, world))


def range_divisor(hr: torch.Tensor) -> float:
    """utils.calc_psnr_and_ssim_cuda (utils.py:244-250): span > 2 -> /255; span > 1 -> (x+1)/2; else 1."""
    span = float(hr.max() - hr.min())
    if span > 2:
        return 255.0
    if span > 1:
        return 2.0
    return 1.0


def psnr_from_mse(mse: float, numel: int) -> float:
    """utils.psnr_cuda (utils.py:177-181)."""
    if mse == 0:
        return -20.0 * math.log10(math.sqrt((1 / 255.0) ** 2 / numel))
    return -20.0 * math.log10(math.sqrt(mse))


def frame_psnrs(sr: torch.Tensor, hr: torch.Tensor):
    """(PSNR, PSNR-Y) of one frame pair [1,3,H,W] the way eval_basicvsr logs them; squared-error sums
    come from the HIP reduction kernel."""
    from . import ops
    acc = ops.sq_err_sums(sr, hr).cpu()
    n_rgb = sr.numel()
    d_rgb = range_divisor(hr)
    p = psnr_from_mse(float(acc[0]) / n_rgb / d_rgb ** 2, n_rgb)
    wy = torch.tensor([24.966, 128.553, 65.481], device=hr.device).view(1, 3, 1, 1)
    y_hr = (hr * wy).sum(1, keepdim=True) + 16.0
    d_y = range_divisor(y_hr)
    n_y = n_rgb // 3
    py = psnr_from_mse(float(acc[1]) / n_y / d_y ** 2, n_y)
    return p, py


def frame_metrics(sr: torch.Tensor, hr: torch.Tensor):
    """(PSNR, SSIM, PSNR-Y, SSIM-Y) of one frame pair [1,3,H,W] exactly as eval_basicvsr logs them
    (trainer.py:348-369): all-ones mask, RGB on the tensors as they are, Y through bgr2ycbcr(y_only) on the
    RGB-ordered data; two passes of the fused HIP PSNR+SSIM kernel (crfp_amd/utils.py)."""
    from . import utils as U
    p, s = U.calc_psnr_and_ssim_cuda(sr, hr, None)
    ys = U.bgr2ycbcr(sr.permute(0, 2, 3, 1), y_only=True)
    yh = U.bgr2ycbcr(hr.permute(0, 2, 3, 1), y_only=True)
    py, sy = U.calc_psnr_and_ssim_cuda(ys, yh, None)
    return float(p), float(s), float(py), float(sy)


def evaluate(clip_fn: Callable[[int], Sequence[Sequence[float]]], n_clips: int, rank: int = 0, world: int = 1, dist=None,
             device: Optional[torch.device] = None):
    """Run `clip_fn(clip_index) -> [per counted frame: (psnr, psnr_y) or (psnr, ssim, psnr_y, ssim_y)]` on this rank's
    shard and reduce with ONE all-reduce of [sum psnr, sum ssim, sum psnr_y, sum ssim_y, n].
    Returns dict(psnr, psnr_y, frames[, ssim, ssim_y])."""
    sums = torch.zeros(5, dtype=torch.float64)
    have_ssim = False
    for c in shard_clips(n_clips, rank, world):
        for m in clip_fn(c):
            if len(m) == 2:
                sums += torch.tensor([m[0], 0.0, m[1], 0.0, 1.0], dtype=torch.float64)
            else:
                have_ssim = True
                sums += torch.tensor([m[0], m[1], m[2], m[3], 1.0], dtype=torch.float64)
    if dist is not None and world > 1:
        buf = sums.to(device) if device is not None else sums
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        sums = buf.cpu()
    n = float(sums[4])
    nan = float("nan")
    out = {"psnr": float(sums[0]) / n if n else nan, "psnr_y": float(sums[2]) / n if n else nan, "frames": int(n)}
    if have_ssim:
        out.update(ssim=float(sums[1]) / n if n else nan, ssim_y=float(sums[3]) / n if n else nan)
    return out


def rgb2yuv(rgb: torch.Tensor) -> torch.Tensor:
    """trainer.rgb2yuv(y_only=False) (trainer.py:19-36)."""
    r, g, b = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    return torch.stack([0.299 * r + 0.587 * g + 0.114 * b, -0.147 * r - 0.289 * g + 0.436 * b,
                        0.615 * r - 0.515 * g - 0.100 * b], dim=1)


def yuv2rgb(yuv: torch.Tensor) -> torch.Tensor:
    """trainer.yuv2rgb (trainer.py:38-48)."""
    y, u, v = yuv[:, 0], yuv[:, 1], yuv[:, 2]
    return torch.stack([y + 1.14 * v, y + -0.396 * u - 0.581 * v, y + 2.029 * u], 1)


def counted_frames(i_batch: int, n_frames: int) -> Iterable[int]:
    """trainer.py:349-351: frame 0 is skipped when i_batch % 50 == 0."""
    return (i for i in range(n_frames) if not (i == 0 and i_batch % 50 == 0))


def eval_clip(model, batch: dict, i_batch: int, with_ssim: bool = True, sr: Optional[torch.Tensor] = None):
    """One eval batch through the model (trainer.py:307-369): batch has LR, HR, Ref, Ref_sp on device.  sr: the model's output for
    this batch when the caller already ran it (eval_reds with clips_per_call > 1)."""
    if sr is None:
        with torch.no_grad():
            sr = model(lrs=batch["LR"], fvs=batch["Ref"], mks=batch["Ref_sp"])
    B, N, C, H, W = sr.shape
    sr = sr.view(B * N, C, H, W)
    hr = batch["HR"].view(B * N, -1, H, W)
    if C == 1:   # y_only model (trainer.py:331-335): chroma of the bicubic LR_sr under the predicted luma, back to RGB
        yuv = rgb2yuv(batch["LR_sr"].view(B * N, 3, H, W).to(sr.dtype))
        sr = yuv2rgb(torch.cat((sr[:, 0:1], yuv[:, 1:3]), dim=1)).contiguous()
    fn = frame_metrics if with_ssim else frame_psnrs
    return [fn(sr[i:i + 1], hr[i:i + 1]) for i in counted_frames(i_batch, N)]


def load_checkpoint(model, model_path: str):
    """Trainer.load (trainer.py:185-199): keep the checkpoint entries whose key (after 'basic_' -> 'basic_module.')
    exists in the model, overlay them on the model's own state and load strictly."""
    sd = model.state_dict()
    saved = {k.replace("basic_", "basic_module."): v for k, v in torch.load(model_path, map_location="cpu").items()
             if k.replace("basic_", "basic_module.") in sd}
    sd.update(saved)
    model.load_state_dict(sd, strict=True)
    return sorted(saved)


def eval_reds(model, args, rank: int = 0, world: int = 1, dist=None, device=None, with_ssim: bool = True, log=None,
              clips_per_call: int = 1):
    """Trainer.eval_basicvsr over dataset.reds.EvalSet (trainer.py:295-413): batch size 1, items sharded round-robin over
    ranks (each item is an independent clip window), per-frame metrics, one final all-reduce.
    clips_per_call > 1 (not in the reference, whose eval DataLoader is batch_size=1, dataset/dataloader.py:14): that many of a rank's
    windows go through ONE model call (crfp_dsv_forward_batch: lock-step launches over the windows, per window bit-identical to its
    own call), every window keeping its own `i_batch` for the frame-0 rule -- the same metrics, more frames per second."""
    from .dataset import reds
    ds = reds.EvalSet(args)
    dev = device if device is not None else next(model.parameters()).device
    if clips_per_call > 1:
        mine = shard_clips(len(ds), rank, world)
        cache = {}

        def batched(i_batch):
            if i_batch not in cache:
                cache.clear()                      # the shard is walked in order: one group alive at a time
                g0 = mine.index(i_batch)
                group = mine[g0:g0 + clips_per_call]
                items = [ds[i] for i in group]
                if any(it["LR"].shape != items[0]["LR"].shape for it in items):
                    group, items = group[:1], items[:1]
                batch = {k: (torch.stack([it[k] for it in items]).to(dev) if torch.is_tensor(items[0][k]) else items[0][k]) for k in items[0]}
                with torch.no_grad():
                    sr_all = model(lrs=batch["LR"], fvs=batch["Ref"], mks=batch["Ref_sp"])
                for b, i in enumerate(group):
                    one = {k: (v[b:b + 1] if torch.is_tensor(v) else v) for k, v in batch.items()}
                    cache[i] = eval_clip(model, one, i, with_ssim, sr=sr_all[b:b + 1])
            m = cache[i_batch]
            if log is not None:
                log(i_batch, m)
            return m

        return evaluate(batched, len(ds), rank, world, dist, dev)

    def clip_fn(i_batch):
        item = ds[i_batch]
        batch = {k: (v.unsqueeze(0).to(dev) if torch.is_tensor(v) else v) for k, v in item.items()}   # DataLoader(batch_size=1)
        m = eval_clip(model, batch, i_batch, with_ssim)
        if log is not None:
            log(i_batch, m)
        return m

    return evaluate(clip_fn, len(ds), rank, world, dist, dev)


def main(argv=None):
    """`python -m crfp_amd.evalrig --dataset_dir <REDS>/val_sharp-style-root [--model_path ckpt.pt]`; under torchrun every
    rank takes its shard (RCCL all-reduce of the five sums at the end)."""
    import argparse
    import json
    import os

    from .model import CRFP
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--dataset_dir", required=True)
    ap.add_argument("--model_path", default=None)
    ap.add_argument("--scale", type=int, default=8)
    ap.add_argument("--N_frames", type=int, default=7)
    ap.add_argument("--GT_size", type=int, default=256)
    ap.add_argument("--FV_size", type=int, default=96)
    ap.add_argument("--y_only", type=int, default=0)
    ap.add_argument("--clips_per_call", type=int, default=1, help="windows per model call (lock-step batch; same metrics)")
    a = ap.parse_args(argv)
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    model = CRFP.CRFP_DSV(device=dev, mid_channels=32, y_only=bool(a.y_only), hr_dcn=True, offset_prop=True).to(dev).eval()
    if a.model_path:
        load_checkpoint(model, a.model_path)
    res = eval_reds(model, a, rank, world, dist, dev, clips_per_call=a.clips_per_call)
    if rank == 0:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
