"""Timing rig with the shape and the arithmetic of the reference's stand-alone speed test (test_runtime.py:81-99,
128-129,142-186): 1 x 5 frames, LR 135 x 240 -> 1080 x 1920, a 96 x 96 fovea crop, a 720 x 720 warp window, 30 repetitions
of which the first 10 are warm-up; prints the output shape and seconds per frame = time / (repeat - warm_up + 1) / t
exactly as the reference's last line does (its divisor counts one repetition too many; kept, so the number is comparable).

Two variants:
  regional (default)  what test_runtime.py itself runs: ``MRCF_runtime.MRCF_simple_v18(...)(lr, fv, warp_size=(720, 720))``
                      = crfp_amd/model/CRFP_runtime.py, the mirror of the reference's benchmark-only wiring
                      (model/CRFP_runtime.py:8364-8682), as ONE C-ABI call per clip (crfp_rt_forward_clip, csrc/engine_rt.hip).
                      With --stage-prints it runs the per-operator composition instead and prints the reference's
                      per-stage means (flow / enc / dcn / res / last / total) on every call (a host sync per stage).
  dsv                 the shipped CRFP_DSV engine (one C-ABI call per clip) over the WHOLE frame with the crop pasted into
                      the warp window -- at least the work of the regional wiring, on the fused path.

    python -m crfp_amd.runtime_rig [--variant regional|dsv] [--repeat 30 --warm-up 10 --t 5 --hr 1080 1920 --fv 96 --warp 720 720]
"""
from __future__ import annotations

import argparse

import torch


def build_inputs(lr: torch.Tensor, fv: torch.Tensor, warp_size):
    """(lr [1,t,3,h,w], fv crop [1,t,3,fh,fw]) -> (lrs, fvs, mks) of CRFP_DSV.forward: the crop sits at the centre of the
    top-left ``warp_size`` window of the 8x frame, mk = 1 there."""
    n, t, _, h, w = lr.shape
    fh, fw = fv.shape[-2:]
    H, W = 8 * h, 8 * w
    wy, wx = min(warp_size[0], H), min(warp_size[1], W)
    y0, x0 = max((wy - fh) // 2, 0), max((wx - fw) // 2, 0)
    fvs = torch.zeros(n, t, 3, H, W, device=lr.device, dtype=lr.dtype)
    mks = torch.zeros(n, t, 1, H, W, device=lr.device, dtype=torch.bool)
    fvs[..., y0:y0 + fh, x0:x0 + fw] = fv
    mks[..., y0:y0 + fh, x0:x0 + fw] = True
    return lr, fvs, mks


def run(repeat_time=30, warm_up=10, t=5, hr=(1080, 1920), fv_size=96, warp_size=(720, 720), device="cuda:0", seed=7,
        variant="regional", quiet=True):
    from . import synth
    from .model import CRFP, MRCF_runtime
    dev = torch.device(device)
    g = torch.Generator(device="cpu").manual_seed(seed)
    lr = torch.rand(1, t, 3, hr[0] // 8, hr[1] // 8, generator=g).to(dev)
    fv = torch.rand(1, t, 3, fv_size, fv_size, generator=g).to(dev)
    if variant == "regional":   # test_runtime.py:41,142
        net = MRCF_runtime.MRCF_simple_v18(mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True, split_ratio=3,
                                           spynet_pretrained='pretrained_models/fnet.pth', device=dev)
        sd = synth.make_state_dict_like({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
        net = net.to(dev).eval()
        net.print_timings = not quiet
        model = lambda lrs, fvs, mks: net(lrs, fvs, warp_size=warp_size)   # noqa: E731
        lrs, fvs, mks = lr, fv, None
    else:
        net = CRFP.CRFP_DSV(device=dev, mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(seed).items()}, strict=True)
        model = net.to(dev).eval()
        lrs, fvs, mks = build_inputs(lr, fv, warp_size)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    infer_time, y = 0.0, None
    with torch.no_grad():
        for idx in range(repeat_time):
            if idx < warm_up:
                infer_time = 0.0
            torch.cuda.synchronize()
            start.record()
            y = model(lrs=lrs, fvs=fvs, mks=mks)
            end.record()
            torch.cuda.synchronize()
            infer_time += start.elapsed_time(end) / 1000
    return y, infer_time / (repeat_time - warm_up + 1) / t


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--repeat", type=int, default=30)
    ap.add_argument("--warm-up", type=int, default=10)
    ap.add_argument("--t", type=int, default=5)
    ap.add_argument("--hr", type=int, nargs=2, default=(1080, 1920))
    ap.add_argument("--fv", type=int, default=96)
    ap.add_argument("--warp", type=int, nargs=2, default=(720, 720))
    ap.add_argument("--variant", choices=("regional", "dsv"), default="regional")
    ap.add_argument("--stage-prints", action="store_true", help="regional: print the per-stage means on every call, as the reference does")
    a = ap.parse_args(argv)
    y, s_per_frame = run(a.repeat, a.warm_up, a.t, tuple(a.hr), a.fv, tuple(a.warp), variant=a.variant, quiet=not a.stage_prints)
    print(y.shape, s_per_frame)   # test_runtime.py:186


if __name__ == "__main__":
    main()
