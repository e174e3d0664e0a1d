#!/usr/bin/env python3
"""Micro-benchmark of the DCNv2 (32->32, dg=8) and flow_warp kernels with controllable offset
statistics (smooth vs noisy) through the engine-internal Q4 path is not exposed, so this times the
API op (NCHW convert + kernel) and reads the library's hipEvent records per kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crfp_amd import _lib, ops

dev = torch.device("cuda:0")
H, W = 360, 640
torch.manual_seed(0)
x = torch.randn(1, 32, H, W, device=dev)
wt = torch.randn(32, 32, 3, 3, device=dev) * 0.1
b = torch.zeros(32, device=dev)
msk = torch.rand(1, 72, H, W, device=dev)
L = _lib.lib()
for name, off in (("zero", torch.zeros(1, 144, H, W, device=dev)),
                  ("const(3.3,-2.7)", torch.zeros(1, 144, H, W, device=dev) + torch.tensor([3.3, -2.7], device=dev).repeat(72).view(1, 144, 1, 1)),
                  ("noise std1", torch.randn(1, 144, H, W, device=dev)),
                  ("noise std4", 4 * torch.randn(1, 144, H, W, device=dev)),
                  ("uniform +-10", 20 * torch.rand(1, 144, H, W, device=dev) - 10)):
    for _ in range(2):
        ops.dcnv2(x, off, msk, wt, b, 3, 1, 1, 8)
    torch.cuda.synchronize()
    L.crfp_prof_reset(); L.crfp_prof_enable(1)
    for _ in range(5):
        ops.dcnv2(x, off, msk, wt, b, 3, 1, 1, 8)
    torch.cuda.synchronize()
    r = [r for r in _lib.prof_report() if r["name"].startswith("dcnv2")][0]
    L.crfp_prof_enable(0)
    us = 1e3 * r["total_ms"] / r["launches"]
    print(f"offsets {name:18s}: dcn_g8 {us:7.1f} us  {r['bytes'] / r['launches'] / us / 1e3:7.1f} GB/s")
