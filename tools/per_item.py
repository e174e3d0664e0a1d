"""Per-launch-site time table (hipEvent profiler inside libcrfp_hip.so) for one BASELINE configs[1] clip.
usage: python tools/per_item.py [steps]   (CRFP_SIDE_STREAM=0 is forced: per-kernel times need one stream)"""
import os, sys
os.environ.setdefault("CRFP_SIDE_STREAM", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crfp_amd import _lib, synth
from crfp_amd.model import CRFP

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
sd = synth.make_state_dict(7)
model = CRFP.CRFP_DSV(device=dev, mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True)
model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
eng = model.to(dev).eval().engine()
lrs, fvs, mks = (torch.from_numpy(a).to(dev) for a in synth.make_clip(1234, 1, 7, 180, 320, fv_size=96, sigma_t=10.0))
with torch.no_grad():
    eng.forward(lrs, fvs, mks); torch.cuda.synchronize()
    L = _lib.lib(); L.crfp_prof_reset(); L.crfp_prof_enable(1)
    for _ in range(steps): eng.forward(lrs, fvs, mks)
    torch.cuda.synchronize()
recs = _lib.prof_report(); L.crfp_prof_enable(0)
tot = sum(r["total_ms"] for r in recs)
print(f"{'site':44s} {'n/clip':>6s} {'avg us':>8s} {'ms/clip':>8s} {'share':>6s}")
for r in sorted(recs, key=lambda r: -r["total_ms"]):
    print(f"{r['name']:44s} {r['launches']/steps:6.1f} {1e3*r['total_ms']/r['launches']:8.1f} {r['total_ms']/steps:8.3f} {r['total_ms']/tot:6.3f}")
print(f"total {tot/steps:.3f} ms per clip")
