#!/usr/bin/env python3
"""Wall time of crfp_spynet_forward (one pair) at 192x320 and 64x96: python tools/time_spynet.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crfp_amd.model import CRFP
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = CRFP.SPyNet(pretrained=None, device=dev).to(dev).eval()
for h, w in ((64, 96), (192, 320)):
    a, b = torch.rand(1, 3, h, w, device=dev), torch.rand(1, 3, h, w, device=dev)
    with torch.no_grad():
        for _ in range(3):
            m(a, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            f = m(a, b)
        torch.cuda.synchronize()
    print(f"spynet {h}x{w}: {1e3 * (time.perf_counter() - t0) / 10:.3f} ms per pair, flow mean |.| {float(f.abs().mean()):.4f}")
