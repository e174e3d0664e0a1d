"""Frames/s of the per-operator compositions (model variants without a one-call engine) at the headline geometry.
usage: python tools/bench_composed.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crfp_amd import synth, _lib
from crfp_amd.model import CRFP

dev = torch.device("cuda:0")
lrs, fvs, mks = (torch.from_numpy(a).to(dev) for a in synth.make_clip(1234, 1, 7, 180, 320, fv_size=96))
for name, cls, kw in (("CRFP_DSV mid=16", CRFP.CRFP_DSV, dict(mid_channels=16)), ("CRFP_DSV mid=32 composed", CRFP.CRFP_DSV, dict(mid_channels=32)),
                      ("CRFP_DSV_CRA mid=32", CRFP.CRFP_DSV_CRA, dict(mid_channels=32)), ("CRFP_simple mid=32", CRFP.CRFP_simple, dict(mid_channels=32))):
    m = cls(device=dev, **kw)
    sd = synth.make_state_dict_like({k: tuple(v.shape) for k, v in m.state_dict().items()}, 3)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    fwd = m.forward_composed if "composed" in name else m
    with torch.no_grad():
        fwd(lrs, fvs, mks); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fwd(lrs, fvs, mks)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        L = _lib.lib(); L.crfp_prof_reset(); L.crfp_prof_enable(1)
        fwd(lrs, fvs, mks); torch.cuda.synchronize()
        recs = _lib.prof_report(512); L.crfp_prof_enable(0); L.crfp_prof_reset()
    ksum = sum(r["total_ms"] for r in recs)
    top = sorted(recs, key=lambda r: -r["total_ms"])[:5]
    print(f"{name:28s} {7 / dt:7.1f} frames/s  {1e3 * dt:7.1f} ms per clip; library kernels {ksum:6.1f} ms in {sum(r['launches'] for r in recs)} launches; top:",
          [(r["name"], round(r["total_ms"], 1)) for r in top], flush=True)
