#!/usr/bin/env python3
"""Timeline of one launch site of conv3x3_split8_kernel (lab library): per workgroup, absolute s_memtime of wave 0 at entry, first
tile in LDS, end of chunk 0's MFMAs, end of the MFMA loop, stores issued, stores acknowledged.
  CRFP_HIP_LIB=crfp_amd/libcrfp_hip_lab.so python tools/stamp_conv8.py conv_mfma:res.conv1"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
site = sys.argv[1] if len(sys.argv) > 1 else "conv_mfma:res.conv1"
dev = torch.device("cuda:0")
buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
os.environ["CRFP_STAMP_PTR"] = str(buf.data_ptr()); os.environ["CRFP_STAMP_NAME"] = site
from crfp_amd import synth
from crfp_amd.model import CRFP
sd = synth.make_state_dict(7)
m = CRFP.CRFP_DSV(device=dev, mid_channels=32); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}); m = m.to(dev).eval()
lrs, fvs, mks = (torch.from_numpy(x).to(dev) for x in synth.make_clip(1234, 1, 2, 180, 320))
eng = m.engine()
for _ in range(3):
    eng.forward(lrs, fvs, mks)
torch.cuda.synchronize()
b = buf.view(-1, 8).cpu().double()
b = b[b[:, 7] > 0]          # the LAST launch of the site that ran (each launch overwrites the table)
t0 = b[:, 0].min()
names = ["entry", "first tile in LDS", "chunk 0 MFMAs done", "MFMA loop done", "stores issued", "stores acknowledged"]
print(f"{site}: {len(b)} workgroups; cycles relative to the first workgroup's entry (shader clock)")
for i, nm in enumerate(names):
    v = b[:, i] - t0
    print(f"  {nm:22s} mean {v.mean():9.0f}  p10 {v.quantile(0.1):9.0f}  p90 {v.quantile(0.9):9.0f}  max {v.max():9.0f}")
d = b[:, 1:6] - b[:, 0:5]
for i, nm in enumerate(["entry -> first tile in LDS (loads + split + 2 barriers)", "chunk 0 MFMAs (chunk 1's loads in flight)", "rest of the loop (split 1, barriers, chunk 1 MFMAs ...)", "scale + epilogue until the last store is issued", "store acknowledgement"]):
    print(f"  phase: {nm:58s} mean {d[:, i].mean():8.0f}  p90 {d[:, i].quantile(0.9):8.0f}")
print(f"  workgroup lifetime mean {(b[:, 5] - b[:, 0]).mean():.0f}; launch span (first entry -> last ack) {(b[:, 5].max() - t0):.0f} cycles")
