#!/usr/bin/env python3
"""Diagnostic: per-phase cycle sums of one conv launch site (split kernel). Usage: stamp_conv.py <site>"""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
site = sys.argv[1] if len(sys.argv) > 1 else "conv_mfma:res.main0"
dev = torch.device("cuda:0")
buf = torch.zeros(16384 * 4, dtype=torch.int64, device=dev)
os.environ["CRFP_STAMP_PTR"] = str(buf.data_ptr()); os.environ["CRFP_STAMP_NAME"] = site
from crfp_amd import synth
from crfp_amd.model import CRFP
sd = synth.make_state_dict(7)
m = CRFP.CRFP_DSV(device=dev, mid_channels=32); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}); m = m.to(dev).eval()
lrs, fvs, mks = (torch.from_numpy(x).to(dev) for x in synth.make_clip(1234, 1, 2, 180, 320))
eng = m.engine(); eng.forward(lrs, fvs, mks); torch.cuda.synchronize()
b = buf.view(-1, 4).cpu().double()
nz = b[(b.sum(1) > 0)]
print(site, "blocks", len(nz))
for i, nm in enumerate(["A: barrier1 (wait prev MFMA + loads land)", "B: split + LDS write + barrier2", "C: issue next loads", "D: MFMA loop"]):
    print(f"  {nm:45s} mean {nz[:, i].mean():10.0f}  max {nz[:, i].max():10.0f}  (x100MHz ticks -> cycles: s_memtime is shader clock)")
print("  total mean", nz.sum(1).mean())
lb = buf.view(-1, 4)[8192:].cpu().double(); lnz = lb[(lb.sum(1) > 0)]
if len(lnz):
    print(" loader wave: blocks", len(lnz))
    for i, nm in enumerate(["issue loads", "split + write tile", "X..Y (weight image write)", "wait at X"]):
        print(f"  {nm:45s} mean {lnz[:, i].mean():10.0f}  max {lnz[:, i].max():10.0f}")
