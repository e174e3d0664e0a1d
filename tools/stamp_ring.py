#!/usr/bin/env python3
"""Phase cycles of the LDS-DMA ring conv (-DCRFP_BF16_RING build): per workgroup, the loader wave's and consumer wave 0's s_memtime sums.
  CRFP_HIP_LIB=_ab/libcrfp_ring.so CRFP_BF16_RING=1 [CRFP_BF16_RING_WGS=64] python tools/stamp_ring.py [site] [clips]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
site = sys.argv[1] if len(sys.argv) > 1 else "conv_mfma:res.conv1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
buf = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
os.environ["CRFP_STAMP_PTR"] = str(buf.data_ptr()); os.environ["CRFP_STAMP_NAME"] = site
from crfp_amd import synth
from crfp_amd.engine import DSVEngine
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(7).items()}
clips = [synth.make_clip(100 + s, 1, 2, 180, 320, fv_size=96) for s in range(n)]
data = tuple(torch.from_numpy(np.concatenate([c[k] for c in clips], 0)).to(dev) for k in range(3))
eng = DSVEngine(sd, dev, storage="bf16")
with torch.no_grad():
    for _ in range(3):
        eng.forward(*data)
torch.cuda.synchronize()
b = buf.view(-1, 16).cpu().double()
b = b[b[:, 15] > 0]
print(f"{site}: {len(b)} workgroups, {b[:, 15].mean():.1f} units each; s_memtime ticks (2.4 per ns at the nominal clock)")
t0 = min(b[:, 0].min(), b[:, 8].min())
def show(nm, v): print(f"  {nm:58s} mean {v.mean():9.0f}  p10 {v.quantile(0.1):9.0f}  p90 {v.quantile(0.9):9.0f}  max {v.max():9.0f}")
show("loader: entry (after the first workgroup's)", b[:, 0] - t0)
show("loader: entry -> A fragments in LDS (preamble)", b[:, 1] - b[:, 0])
show("loader: per-lane set-up + first D - 1 units issued", b[:, 2] - b[:, 1])
show("loader: main loop", b[:, 3] - b[:, 2])
for i, nm in ((4, "landing wait (s_waitcnt vmcnt)"), (5, "fix + cursor"), (6, "barrier"), (7, "issue of unit u + D - 1")):
    show("  loader loop, per unit: " + nm, b[:, i] / b[:, 15])
show("consumer 0: lifetime", b[:, 9] - b[:, 8])
for i, nm in ((10, "barrier (waiting for the unit)"), (11, "operand reads + MFMAs")):
    show("  consumer, per unit: " + nm, b[:, i] / b[:, 15])
show("  consumer, per tile: epilogue", b[:, 12] / (b[:, 15] / 2))
print(f"  launch span (first entry -> last end) {(max(b[:, 3].max(), b[:, 9].max()) - t0):.0f} ticks")
