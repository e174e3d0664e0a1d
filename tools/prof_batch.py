"""Per-kernel time per clip, one-clip calls vs a lock-step batch of n (hipEvent-bracketed launches, single-stream schedule).
usage: python tools/prof_batch.py f32|bf16 n"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crfp_amd import synth, _lib
from crfp_amd.engine import DSVEngine

storage = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
h, w, T = 180, 320, 7
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(7).items()}
clips = [synth.make_clip(100 + s, 1, T, h, w, fv_size=96) for s in range(n)]
data = tuple(torch.from_numpy(np.concatenate([c[k] for c in clips], 0)).to(dev) for k in range(3))
eng = DSVEngine(sd, dev, storage=storage)
L = _lib.lib()
tab = {}
for mode in ("loop", "lockstep"):
    eng.batch_mode = mode
    with torch.no_grad():
        out = eng.forward(*data)
        torch.cuda.synchronize()
        import hashlib
        print(mode, "digest", hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], "finite", bool(torch.isfinite(out).all()), flush=True)
        L.crfp_prof_reset(); L.crfp_prof_enable(1)
        for _ in range(3):
            eng.forward(*data)
        torch.cuda.synchronize()
        recs = _lib.prof_report(512)
        L.crfp_prof_enable(0); L.crfp_prof_reset()
    for r in recs:
        tab.setdefault(r["name"], {})[mode] = (r["total_ms"] / 3 / n, r["launches"] / 3)
tot = {"loop": 0.0, "lockstep": 0.0}
print(f"{'kernel':44s} {'loop ms/clip':>12s} {'lock ms/clip':>12s} {'ratio':>6s} {'us/launch loop':>14s} {'us/launch/clip lock':>18s}")
for name, d in sorted(tab.items(), key=lambda kv: -kv[1].get("loop", (0, 0))[0]):
    a, la = d.get("loop", (0, 1)); b, lb = d.get("lockstep", (0, 1))
    tot["loop"] += a; tot["lockstep"] += b
    print(f"{name:44s} {a:12.4f} {b:12.4f} {b / a if a else 0:6.3f} {1e3 * a * n / max(la, 1):14.1f} {1e3 * b / max(lb, 1):18.1f}")
print("total", tot)
