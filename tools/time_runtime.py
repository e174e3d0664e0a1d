#!/usr/bin/env python3
"""Per-launch-site table of the regional wiring (crfp_rt_forward_clip) at the reference's test_runtime.py geometry.

  python tools/time_runtime.py [--t 5 --hr 1080 1920 --fv 96 --warp 720 720 --steps 5]

Prints wall ms per clip and, from an instrumented pass (crfp_prof_*: every launch bracketed by hipEvents), launches x avg us per site."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--t", type=int, default=5)
    ap.add_argument("--hr", type=int, nargs=2, default=(1080, 1920))
    ap.add_argument("--fv", type=int, default=96)
    ap.add_argument("--warp", type=int, nargs=2, default=(720, 720))
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    from crfp_amd import _lib, synth
    from crfp_amd.model import MRCF_runtime
    dev = torch.device("cuda:0")
    net = MRCF_runtime.MRCF_simple_v18(mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True, split_ratio=3, device=dev)
    sd = synth.make_state_dict_like({k: tuple(v.shape) for k, v in net.state_dict().items()}, 7)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    net = net.to(dev).eval()
    net.print_timings = False
    g = torch.Generator(device="cpu").manual_seed(7)
    lr = torch.rand(1, a.t, 3, a.hr[0] // 8, a.hr[1] // 8, generator=g).to(dev)
    fv = torch.rand(1, a.t, 3, a.fv, a.fv, generator=g).to(dev)
    L = _lib.lib()
    with torch.no_grad():
        for _ in range(3):
            net(lr, fv, warp_size=tuple(a.warp))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            net(lr, fv, warp_size=tuple(a.warp))
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0) / a.steps
        enq = []
        for _ in range(a.steps):            # host time to enqueue one clip (the call returns before the GPU has finished)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            net(lr, fv, warp_size=tuple(a.warp))
            enq.append(1e3 * (time.perf_counter() - t1))
        torch.cuda.synchronize()
        print(f"host enqueue time per clip: min {min(enq):.3f} ms, median {sorted(enq)[len(enq) // 2]:.3f} ms")
        L.crfp_prof_reset(); L.crfp_prof_enable(1)
        for _ in range(a.steps):
            net(lr, fv, warp_size=tuple(a.warp))
        torch.cuda.synchronize()
        recs = _lib.prof_report(512)
        L.crfp_prof_enable(0)
    tot = sum(r["total_ms"] for r in recs) / a.steps
    print(f"wall {wall:.3f} ms per {a.t}-frame clip ({wall / a.t:.3f} ms per frame); kernel sum {tot:.3f} ms; {sum(r['launches'] for r in recs) / a.steps:.0f} launches")
    for r in sorted(recs, key=lambda r: -r["total_ms"]):
        n = r["launches"] / a.steps
        print(f"  {r['name']:44s} {n:6.1f} x {1e3 * r['total_ms'] / r['launches']:7.1f} us = {r['total_ms'] / a.steps:7.3f} ms  {100 * r['total_ms'] / a.steps / tot:5.1f} %")


if __name__ == "__main__":
    main()
