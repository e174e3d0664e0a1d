"""Frames/s of the CRFP_DSV_CRA engine schedule (crfp_cra_forward_batch) at the headline geometry, beside the plain CRFP_DSV engine on the
same box, plus the per-site table of the wiring's own launches.
usage: python tools/bench_cra.py [--clips 1 4]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crfp_amd import synth, _lib
from crfp_amd.model import CRFP

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, nargs="+", default=[1, 4])
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")


def build(cls):
    m = cls(device=dev, mid_channels=32)
    sd = synth.make_state_dict_like({k: tuple(v.shape) for k, v in m.state_dict().items()}, 3)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev).eval()


models = {"CRFP_DSV": build(CRFP.CRFP_DSV), "CRFP_DSV_CRA": build(CRFP.CRFP_DSV_CRA)}
for n in a.clips:
    lrs, fvs, mks = (torch.from_numpy(x).to(dev) for x in synth.make_clip(1234, n, 7, 180, 320, fv_size=96))
    for storage in ("f32", "bf16"):
        for name, m in models.items():
            m.storage = storage
            with torch.no_grad():
                for _ in range(3):
                    m(lrs, fvs, mks)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    m(lrs, fvs, mks)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            print(f"{name:14s} {storage:4s} clips/call {n}: {7 * n / dt:8.1f} frames/s  {1e3 * dt / n:7.3f} ms per clip", flush=True)
m = models["CRFP_DSV_CRA"]
lrs, fvs, mks = (torch.from_numpy(x).to(dev) for x in synth.make_clip(1234, 1, 7, 180, 320, fv_size=96))
for storage in ("f32", "bf16"):
    m.storage = storage
    L = _lib.lib()
    with torch.no_grad():
        m(lrs, fvs, mks); torch.cuda.synchronize()
        L.crfp_prof_reset(); L.crfp_prof_enable(1)
        m(lrs, fvs, mks); torch.cuda.synchronize()
        recs = _lib.prof_report(512); L.crfp_prof_enable(0); L.crfp_prof_reset()
    tot = sum(r["total_ms"] for r in recs)
    print(f"-- CRFP_DSV_CRA {storage}: {tot:.3f} ms of kernels per clip (single-stream timing); the wiring's own launches:")
    for r in sorted(recs, key=lambda r: -r["total_ms"]):
        if "cra" in r["name"]:
            print(f"   {r['name']:36s} n {r['launches']:3d}  {1e3 * r['total_ms'] / r['launches']:7.1f} us  {r['total_ms']:6.3f} ms/clip")
