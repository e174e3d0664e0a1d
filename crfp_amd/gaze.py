"""Gaze-trajectory driver and region metrics for the streaming model: counterpart of the per-frame loop of the
reference's video rig (test_video.py:303-379; BASELINE config 3 = 100 streamed frames, sigma^T = 50).

Per frame n (one model call, state kept on the device between calls):
  * gaze centre (x, y) ~ N(frame centre, sigma^2), drawn up front as ``sigma*randn(N) + W/2`` then ``sigma*randn(N) + H/2``
    (test_video.py:310-311: x first, then y, from one NumPy stream); fovea window origin = int(centre) - fv_size//2
    (:335-336).  The reference does not clip it; a window that would leave the frame is clipped here (no-op otherwise).
  * fv = GT inside the window, 0 elsewhere; mk = 1 inside (:340-342), only from frame ``fv_start`` on;
    mk_fv = mk with the window forced to 1 (:344-345);
    mk_out = (10 x [3x3 ones conv, clamp]) dilation of mk_fv minus mk = the ring around the fovea (:346-349);
    mk_past = union of the last three mk_out, as it was BEFORE this frame's ring is pushed (:371-375);
    fg = the regional-DCN box of rg x rg pixels around the window centre, or all ones (:351-358).
  * metrics = utils.calc_psnr_and_ssim_cuda(sr, gt, mask) for whole / fovea / outskirt / past (:360-370), arithmetic mean.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def gaze_trajectory(n_frames: int, H: int, W: int, sigma: float, rng: np.random.RandomState) -> Tuple[np.ndarray, np.ndarray]:
    """test_video.py:310-311 (the reference draws from the global NumPy stream; pass a seeded RandomState)."""
    x_array = sigma * rng.randn(n_frames) + (W / 2)
    y_array = sigma * rng.randn(n_frames) + (H / 2)
    return x_array, y_array


def window_origin(x: float, y: float, fv_size: int, H: int, W: int) -> Tuple[int, int]:
    """(cur_y, cur_x) of test_video.py:335-336, clipped so that the fv_size window stays inside the frame."""
    cur_y = int(y) - fv_size // 2
    cur_x = int(x) - fv_size // 2
    return min(max(cur_y, 0), H - fv_size), min(max(cur_x, 0), W - fv_size)


def dilate10(mask: torch.Tensor, iterations: int = 10) -> torch.Tensor:
    """`iterations` x (3x3 ones convolution, clamp to [0,1]) on a {0,1} mask [*,1,H,W] (test_video.py:347-348) = one
    (2*iterations+1)^2 max filter."""
    k = 2 * iterations + 1
    return F.max_pool2d(mask.float(), kernel_size=k, stride=1, padding=iterations)


def regional_box(cur_y: int, cur_x: int, fv_size: int, rg_h: int, rg_w: int, H: int, W: int) -> Tuple[int, int, int, int]:
    """(y0, y1, x0, x1) of the regional-DCN box (test_video.py:351-354; the reference hard-codes 1920 x 1080 as W x H)."""
    x0 = max(cur_x + (fv_size // 2) - (rg_w // 2), 0)
    x1 = min(cur_x + (fv_size // 2) + (rg_w // 2), W)
    y0 = max(cur_y + (fv_size // 2) - (rg_h // 2), 0)
    y1 = min(cur_y + (fv_size // 2) + (rg_h // 2), H)
    return y0, y1, x0, x1


class RegionMasks:
    """The per-frame masks of the rig; keeps the three-frame history that mk_past needs."""

    def __init__(self, H: int, W: int, fv_size: int, device, fv_start: int = 0, regional_dcn: bool = False,
                 rg_h: int = 0, rg_w: int = 0):
        self.H, self.W, self.fv, self.dev = H, W, fv_size, device
        self.fv_start, self.regional, self.rg_h, self.rg_w = fv_start, regional_dcn, rg_h, rg_w
        self.history: List[torch.Tensor] = []
        self.past: Optional[torch.Tensor] = None

    def frame(self, n: int, cur_y: int, cur_x: int) -> Dict[str, torch.Tensor]:
        H, W, fv = self.H, self.W, self.fv
        mk = torch.zeros((1, 1, H, W), device=self.dev)
        if n >= self.fv_start:
            mk[:, :, cur_y:cur_y + fv, cur_x:cur_x + fv] = 1
        mk_fv = mk.clone()
        mk_fv[:, :, cur_y:cur_y + fv, cur_x:cur_x + fv] = 1
        mk_out = torch.logical_and(torch.logical_not(mk.bool()), dilate10(mk_fv).bool())
        if self.regional:
            y0, y1, x0, x1 = regional_box(cur_y, cur_x, fv, self.rg_h, self.rg_w, H, W)
            fg = torch.zeros((1, 1, H, W), device=self.dev)
            fg[:, :, y0:y1, x0:x1] = 1
        else:
            fg = torch.ones((1, 1, H, W), device=self.dev)
        out = {"mk": mk.bool(), "fovea": mk_fv.bool(), "outskirt": mk_out, "past": self.past, "fg": fg.bool()}
        # history update happens after the frame's metrics (test_video.py:371-375)
        self.history.append(mk_out)
        if len(self.history) > 3:
            self.history.pop(0)
        self.past = torch.stack(self.history, 0).any(0)
        return out


def run_gaze_video(model, lr: torch.Tensor, gt: torch.Tensor, sigma: float, fv_size: int = 96, seed: int = 1234,
                   fv_start: int = 0, regional_dcn: bool = False, rg: int = 0, metric_fn=None) -> Dict[str, object]:
    """Stream `lr [N,3,h,w]` / `gt [N,3,8h,8w]` (device tensors, range [0,1]) through `model` (MRCF_simple_v18 interface:
    ``model(lrs=, fvs=, mks=, fgs=)`` one frame per call, ``clear_states()``) along a gaussian gaze trajectory and collect
    the rig's region metrics.  Returns per-region mean PSNR / SSIM, the trajectory and the outputs' checksum."""
    regions_fn = None
    if metric_fn is None:
        from . import utils as U
        metric_fn, regions_fn = U.calc_psnr_and_ssim_cuda, U.calc_psnr_and_ssim_regions
    N, _, H, W = gt.shape
    xs, ys = gaze_trajectory(N, H, W, sigma, np.random.RandomState(seed))
    masks = RegionMasks(H, W, fv_size, gt.device, fv_start, regional_dcn, rg, rg)
    regions = ("whole", "fovea", "outskirt", "past")
    acc = {r: [] for r in regions}
    traj = []
    ones = torch.ones((1, 1, H, W), device=gt.device, dtype=torch.bool)
    model.clear_states()
    with torch.no_grad():
        for n in range(N):
            cur_y, cur_x = window_origin(xs[n], ys[n], fv_size, H, W)
            traj.append((cur_y, cur_x))
            m = masks.frame(n, cur_y, cur_x)
            g = gt[n:n + 1]
            fv = g * m["mk"]
            sr = model(lrs=lr[n:n + 1].unsqueeze(0), fvs=fv.unsqueeze(0), mks=m["mk"].unsqueeze(0), fgs=m["fg"].unsqueeze(0))
            sr = sr.reshape(1, -1, H, W)
            todo = [(r, mask) for r, mask in (("whole", ones), ("fovea", m["fovea"]), ("outskirt", m["outskirt"]),
                                              ("past", m["past"])) if mask is not None]   # frame 0 has no past ring
            if regions_fn is not None:   # one range probe and one host sync per frame
                for (r, _), (p, s) in zip(todo, regions_fn(sr, g, [mask for _, mask in todo])):
                    acc[r].append((float(p), float(s)))
            else:
                for r, mask in todo:
                    p, s = metric_fn(sr, g, mask)
                    acc[r].append((float(p), float(s)))
    out: Dict[str, object] = {"trajectory": traj, "frames": N}
    for r in regions:
        if acc[r]:
            out[f"psnr_{r}"] = float(np.mean([v[0] for v in acc[r]]))
            out[f"ssim_{r}"] = float(np.mean([v[1] for v in acc[r]]))
    out["per_frame"] = acc
    return out


def main(argv=None):
    """BASELINE config 3 shape: N streamed frames at h x w -> 8h x 8w, gaussian gaze (sigma^T), synthetic data and
    weights (no REDS / checkpoints on the box); prints one JSON line with frames/s and the region metrics."""
    import argparse
    import json
    import time

    from . import synth
    from .model import CRFP

    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--lr-h", type=int, default=180)
    ap.add_argument("--lr-w", type=int, default=320)
    ap.add_argument("--sigma", type=float, default=50.0)
    ap.add_argument("--fv-size", type=int, default=96)
    ap.add_argument("--regional-dcn", type=int, default=0, help="side of the regional-DCN box in HR pixels (0 = whole frame)")
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    sd = synth.make_state_dict(7)
    m = CRFP.MRCF_simple_v18(device=dev, mid_channels=32)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.to(dev).eval()
    chunk = 10   # synthetic frames are generated in chunks of correlated frames and the state carried across
    lrs = np.concatenate([synth.make_clip(a.seed + i, 1, min(chunk, a.frames - i), a.lr_h, a.lr_w, fv_size=a.fv_size)[0][0]
                          for i in range(0, a.frames, chunk)], 0)
    lr = torch.from_numpy(lrs).to(dev)
    gt = torch.clamp(F.interpolate(lr, scale_factor=8, mode="bilinear", align_corners=False), 0, 1)   # stand-in ground truth
    run = lambda: run_gaze_video(m, lr, gt, a.sigma, a.fv_size, a.seed, regional_dcn=a.regional_dcn > 0, rg=a.regional_dcn)
    run_gaze_video(m, lr[:3], gt[:3], a.sigma, a.fv_size, a.seed)   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the same stream without the metric kernels and their host syncs
    ones = torch.ones((1, 1, 1, gt.shape[2], gt.shape[3]), device=dev, dtype=torch.bool)
    m.clear_states()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for n in range(a.frames):
            m(lrs=lr[n:n + 1].unsqueeze(0), fvs=gt[n:n + 1].unsqueeze(0), mks=ones, fgs=ones)
    torch.cuda.synchronize()
    dt_model = time.perf_counter() - t0
    res.pop("per_frame"); res.pop("trajectory")
    print(json.dumps({"workload": f"BASELINE config 3 shape: {a.frames} streamed frames {a.lr_h}x{a.lr_w} -> x8, sigma_T={a.sigma}, fp32, synthetic",
                      "frames_per_sec_with_region_metrics": a.frames / dt, "frames_per_sec_model_only": a.frames / dt_model, **res}))


if __name__ == "__main__":
    main()
