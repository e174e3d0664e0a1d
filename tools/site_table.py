"""Per launch-site kernel table of one clip (hipEvent records, single-stream instrumented pass).
  python tools/site_table.py [--storage bf16] [--h 180 --w 320]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crfp_amd import _lib, synth
from crfp_amd.model import CRFP
ap = argparse.ArgumentParser()
ap.add_argument("--storage", default="f32"); ap.add_argument("--h", type=int, default=180); ap.add_argument("--w", type=int, default=320)
ap.add_argument("--fv", type=int, default=96)
ap.add_argument("--mid", type=int, default=32, help="16: the constructor-default width, embedded in the 32-channel schedule (same launches, half the channels exact zeros)")
a = ap.parse_args()
dev = torch.device("cuda:0")
m = CRFP.CRFP_DSV(device=dev, mid_channels=a.mid)
sd = synth.make_state_dict(7) if a.mid == 32 else synth.make_state_dict_like({k: tuple(v.shape) for k, v in m.state_dict().items()}, 7)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
m.storage = a.storage
m = m.to(dev).eval()
lrs, fvs, mks = (torch.from_numpy(x).to(dev) for x in synth.make_clip(1234, 1, 7, a.h, a.w, fv_size=a.fv))
L = _lib.lib()
N = 3
with torch.no_grad():
    for _ in range(2):
        m(lrs=lrs, fvs=fvs, mks=mks)
    torch.cuda.synchronize()
    L.crfp_prof_reset(); L.crfp_prof_enable(1)
    for _ in range(N):
        m(lrs=lrs, fvs=fvs, mks=mks)
    torch.cuda.synchronize()
recs = _lib.prof_report(512)
tot = sum(r["total_ms"] for r in recs) / N
print(f"storage {a.storage} {a.h}x{a.w}: {tot:.3f} ms of kernels per clip")
print(f"{'site':34s} {'n/clip':>6s} {'us':>8s} {'ms/clip':>8s} {'share':>6s} {'GB/s':>7s} {'TF':>7s}")
for r in sorted(recs, key=lambda r: -r["total_ms"]):
    us = 1e3 * r["total_ms"] / r["launches"]
    print(f"{r['name']:34s} {r['launches'] / N:6.1f} {us:8.1f} {r['total_ms'] / N:8.3f} {100 * r['total_ms'] / N / tot:5.1f}% "
          f"{r['bytes'] / r['launches'] / us / 1e3:7.0f} {r['flops'] / r['launches'] / us / 1e6:7.1f}")
