#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline workload: clip inputs start in pinned host memory, the SR frames end there.
(The C-ABI takes device pointers, so this is context for DESIGN.md, never the bench value.)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crfp_amd import synth
from crfp_amd.model import CRFP

dev = torch.device("cuda:0")
sd = synth.make_state_dict(7)
m = CRFP.CRFP_DSV(device=dev, mid_channels=32)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
m = m.to(dev).eval()
eng = m.engine()
lrs, fvs, mks = synth.make_clip(1234, 1, 7, 180, 320, fv_size=96, sigma_t=10.0)
host = [torch.from_numpy(a).pin_memory() for a in (lrs, fvs, mks)]
out_host = torch.empty((1, 7, 3, 1440, 2560), dtype=torch.float32).pin_memory()
def step():
    d = [h.to(dev, non_blocking=True) for h in host]
    out = eng.forward(*d)
    out_host.copy_(out, non_blocking=True)
for _ in range(2): step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
mb = sum(h.numel() * h.element_size() for h in host) / 1e6 + out_host.numel() * 4 / 1e6
print(f"PCIe-inclusive: {7 / dt:.1f} frames/s ({dt * 1e3:.2f} ms per clip; {mb:.0f} MB over PCIe per clip = {mb / dt / 1e3:.1f} GB/s), single stream, no copy/compute overlap")
