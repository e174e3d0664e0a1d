"""Per-kernel times of the gather kernels inside the engine at config A (hipEvent records of the instrumented pass).
  [CRFP_HIP_LIB=crfp_amd/libcrfp_hip_lab.so CRFP_DCN_PROBE=1] python tools/dcn_probe.py [--storage bf16]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crfp_amd import _lib, synth
from crfp_amd.model import CRFP
ap = argparse.ArgumentParser(); ap.add_argument("--storage", default="f32")
ap.add_argument("--offset-std", type=float, default=None, help="0.02 = SURVEY 8d weights (small residual offsets)")
a = ap.parse_args()
dev = torch.device("cuda:0")
sd = synth.make_state_dict(7, offset_std=a.offset_std)
m = CRFP.CRFP_DSV(device=dev, mid_channels=32)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
m.storage = a.storage
m = m.to(dev).eval()
lrs, fvs, mks = (torch.from_numpy(x).to(dev) for x in synth.make_clip(1234, 1, 7, 180, 320, fv_size=96))
L = _lib.lib()
with torch.no_grad():
    for _ in range(2):
        m(lrs=lrs, fvs=fvs, mks=mks)
    torch.cuda.synchronize()
    L.crfp_prof_reset(); L.crfp_prof_enable(1)
    for _ in range(3):
        m(lrs=lrs, fvs=fvs, mks=mks)
    torch.cuda.synchronize()
for r in _lib.prof_report(512):
    if r["name"].startswith(("dcnv2", "flow_warp", "conv_mfma:dcn.offset")):
        us = 1e3 * r["total_ms"] / r["launches"]
        print(f"{r['name']:28s} {us:8.1f} us  {r['bytes'] / r['launches'] / us / 1e3:8.1f} GB/s")
