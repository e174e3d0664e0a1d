#!/usr/bin/env python3
"""Diagnostic (library built with -DCRFP_PIPE_STAMPS, CRFP_SPLIT_PIPE=1): phase cycle sums of the pipelined conv."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
site = sys.argv[1] if len(sys.argv) > 1 else "conv_mfma:res.conv1"
dev = torch.device("cuda:0")
buf = torch.zeros(32768 * 4, dtype=torch.int64, device=dev)
os.environ["CRFP_STAMP_PTR"] = str(buf.data_ptr()); os.environ["CRFP_STAMP_NAME"] = site
os.environ["CRFP_SPLIT_PIPE"] = "1"; os.environ["CRFP_SIDE_STREAM"] = "0"
from crfp_amd import synth
from crfp_amd.model import CRFP
sd = synth.make_state_dict(7)
m = CRFP.CRFP_DSV(device=dev, mid_channels=32); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}); m = m.to(dev).eval()
lrs, fvs, mks = (torch.from_numpy(x).to(dev) for x in synth.make_clip(1234, 1, 2, 180, 320))
eng = m.engine(); eng.forward(lrs, fvs, mks); torch.cuda.synchronize()
b = buf.view(-1, 8)[:4096].cpu().double()
b = b[b[:, 4] > 0]
print(site, "workgroups", len(b), "items per workgroup", b[:, 4].mean().item())
for i, nm in enumerate(["wait for loads (vmcnt 0) at item start", "issue + tap stream (MFMA + split in shadow)", "epilogue (tile ends)", "barrier"]):
    print(f"  {nm:45s} per item mean {(b[:, i] / b[:, 4]).mean():9.0f}   total/WG {b[:, i].mean():10.0f}")
print(f"  total per workgroup {b[:, :4].sum(1).mean():.0f} cycles = {b[:, :4].sum(1).mean() / 2.05e3:.1f} us at 2.05 GHz; MFMA-only bound 3456/item")

w = buf.view(-1, 8)[8192:8192 + 8].cpu().double()
print("  per wave of workgroup 3 (taps / epilogue / barrier, cycles per item):")
for i in range(8):
    if w[i, 4] > 0:
        print(f"    wave {i}: taps {w[i,1]/w[i,4]:7.0f}  epilogue {w[i,2]/w[i,4]:7.0f}  barrier {w[i,3]/w[i,4]:7.0f}")
