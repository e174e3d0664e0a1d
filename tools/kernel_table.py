#!/usr/bin/env python3
"""Per-launch-site kernel table (hipEvent timing inside libcrfp_hip.so) for one bench clip.
    python tools/kernel_table.py [--frames 7] [--lr-h 180] [--lr-w 320] [--reps 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crfp_amd import _lib, synth  # noqa: E402
from crfp_amd.model import CRFP  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=7)
ap.add_argument("--lr-h", type=int, default=180)
ap.add_argument("--lr-w", type=int, default=320)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
sd = synth.make_state_dict(7)
m = CRFP.CRFP_DSV(device=dev, mid_channels=32)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
m = m.to(dev).eval()
lrs, fvs, mks = (torch.from_numpy(x).to(dev) for x in synth.make_clip(1234, 1, a.frames, a.lr_h, a.lr_w))
eng = m.engine()
for _ in range(2):
    eng.forward(lrs, fvs, mks)
torch.cuda.synchronize()
L = _lib.lib()
L.crfp_prof_reset()
L.crfp_prof_enable(1)
for _ in range(a.reps):
    eng.forward(lrs, fvs, mks)
torch.cuda.synchronize()
recs = _lib.prof_report()
L.crfp_prof_enable(0)
tot = sum(r["total_ms"] for r in recs)
print(f"{'launch site':34s} {'n/clip':>7s} {'ms/clip':>8s} {'avg us':>8s} {'share':>6s} {'GB/s':>8s} {'TFLOP/s':>8s}")
for r in sorted(recs, key=lambda r: -r["total_ms"]):
    s = r["total_ms"] * 1e-3
    print(f"{r['name']:34s} {r['launches'] / a.reps:7.1f} {r['total_ms'] / a.reps:8.3f} "
          f"{1e3 * r['total_ms'] / r['launches']:8.1f} {r['total_ms'] / tot:6.3f} {r['bytes'] / s / 1e9:8.1f} {r['flops'] / s / 1e12:8.2f}")
print(f"total {tot / a.reps:.3f} ms per clip")
