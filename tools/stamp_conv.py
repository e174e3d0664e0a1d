#!/usr/bin/env python3
"""Diagnostic: per-phase cycle sums of one conv launch site (split kernel). Usage: stamp_conv.py <site>"""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
site = sys.argv[1] if len(sys.argv) > 1 else "conv_mfma:res.main0"
dev = torch.device("cuda:0")
buf = torch.zeros(65536 * 4, dtype=torch.int64, device=dev)
os.environ["CRFP_STAMP_PTR"] = str(buf.data_ptr()); os.environ["CRFP_STAMP_NAME"] = site
from crfp_amd import synth
from crfp_amd.model import CRFP
sd = synth.make_state_dict(7)
m = CRFP.CRFP_DSV(device=dev, mid_channels=32); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}); m = m.to(dev).eval()
lrs, fvs, mks = (torch.from_numpy(x).to(dev) for x in synth.make_clip(1234, 1, 2, 180, 320))
eng = m.engine(); eng.forward(lrs, fvs, mks); torch.cuda.synchronize()
b8 = buf.view(-1, 8)[:4096].cpu().double()
b8 = b8[b8[:, 4] > 0]
b = b8[:, :4]
nz = b
print(site, "blocks", len(nz))
for i, nm in enumerate(["A: barrier1 (wait prev MFMA + loads land)", "B: split + LDS write + barrier2", "C: issue next loads", "D: MFMA loop"]):
    print(f"  {nm:45s} mean {nz[:, i].mean():10.0f}  max {nz[:, i].max():10.0f}  (x100MHz ticks -> cycles: s_memtime is shader clock)")
print("  total mean", nz.sum(1).mean())
if len(b8):
    t0 = b8[:, 4].min()
    ent, mm, end = b8[:, 4] - t0, b8[:, 5] - t0, b8[:, 6] - t0
    print(f"  entry->first stamp(t0) n/a; entry->main loop end mean {(mm - ent).mean():.0f}; epilogue+store drain mean {(end - mm).mean():.0f}; WG lifetime mean {(end - ent).mean():.0f}")
    iss = b8[:, 7] - t0
    e1, e2 = b8[:, 2] - t0, b8[:, 3] - t0
    print(f"  (epilogue split: MFMA drain + f16 combine {(e1 - mm).mean():.0f}; epi_ctx scalar loads {(e2 - e1).mean():.0f}; act + address + store issue {(iss - e2).mean():.0f})")
    print(f"  epilogue issue (bias, act, store issue) mean {(iss - mm).mean():.0f}; store drain (s_waitcnt vmcnt(0)) mean {(end - iss).mean():.0f}")
lb = buf.view(-1, 4)[16384:].cpu().double(); lnz = lb[(lb.sum(1) > 0)]
if len(lnz):
    print(" loader wave: blocks", len(lnz))
    for i, nm in enumerate(["issue loads", "split + write tile", "X..Y (weight image write)", "wait at X"]):
        print(f"  {nm:45s} mean {lnz[:, i].mean():10.0f}  max {lnz[:, i].max():10.0f}")

dbg = buf.view(-1, 8)[16384:16384 + 1024].cpu().double(); dbg = dbg[dbg[:, 3] > 0]
if len(dbg):
    print(f"  epilogue body (CRFP_EPI_DBG build): pixel tile 0 {(dbg[:,1]-dbg[:,0]).mean():.0f} cycles, pixel tile 1 + exit {(dbg[:,2]-dbg[:,1]).mean():.0f}")
