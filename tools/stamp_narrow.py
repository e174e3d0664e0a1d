#!/usr/bin/env python3
"""Diagnostic (library built with -DCRFP_NARROW_STAMPS): phase cycles of one narrow-conv launch site."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
site = sys.argv[1] if len(sys.argv) > 1 else "conv_narrow:dcn3.block0"
dev = torch.device("cuda:0")
buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
os.environ["CRFP_STAMP_PTR"] = str(buf.data_ptr()); os.environ["CRFP_STAMP_NAME"] = site
os.environ["CRFP_SIDE_STREAM"] = "0"
from crfp_amd import synth
from crfp_amd.model import CRFP
sd = synth.make_state_dict(7)
m = CRFP.CRFP_DSV(device=dev, mid_channels=32); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}); m = m.to(dev).eval()
lrs, fvs, mks = (torch.from_numpy(x).to(dev) for x in synth.make_clip(1234, 1, 2, 180, 320))
eng = m.engine(); eng.forward(lrs, fvs, mks); torch.cuda.synchronize()
b = buf.view(-1, 8).cpu().double(); b = b[b[:, 5] > 0]
print(site, "workgroups", len(b))
for i, nm in enumerate(["issue loads + LDS write", "barrier wait", "compute (LDS reads + FMA)", "epilogue issue", "store drain"]):
    print(f"  {nm:28s} mean {b[:, i].mean():8.0f}  max {b[:, i].max():8.0f}")
print(f"  workgroup lifetime mean {b[:, :5].sum(1).mean():.0f} cycles")
