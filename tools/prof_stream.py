"""Per-launch-site time of one streamed frame (crfp_dsv_stream_frame, single-stream instrumented pass).
usage: python tools/prof_stream.py f32|bf16"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crfp_amd import synth, _lib
from crfp_amd.engine import DSVEngine

storage = sys.argv[1] if len(sys.argv) > 1 else "bf16"
h, w, T = 180, 320, 12
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(7).items()}
lrs, fvs, mks = (torch.from_numpy(a).to(dev) for a in synth.make_clip(100, 1, T, h, w, fv_size=96, sigma_t=50.0))
mk8 = mks.view(torch.uint8)
eng = DSVEngine(sd, dev, storage=storage)
L = _lib.lib()
with torch.no_grad():
    eng.clear_states()
    for i in range(3):
        eng.stream_frame(lrs[0, i], fvs[0, i], mk8[0, i])
    torch.cuda.synchronize()
    L.crfp_prof_reset(); L.crfp_prof_enable(1)
    n = 0
    for i in range(3, T):
        eng.stream_frame(lrs[0, i], fvs[0, i], mk8[0, i]); n += 1
    torch.cuda.synchronize()
    recs = _lib.prof_report(512)
    L.crfp_prof_enable(0); L.crfp_prof_reset()
tot = 0.0
for r in sorted(recs, key=lambda r: -r["total_ms"]):
    tot += r["total_ms"] / n
    print(f"{r['name']:44s} {1e3 * r['total_ms'] / n:9.1f} us/frame  {r['launches'] / n:5.1f} launches  {1e3 * r['total_ms'] / r['launches']:8.1f} us/launch")
print("total us/frame", 1e3 * tot)
fn = sum(r["total_ms"] for r in recs if "fnet" in r["name"] or r["name"] in ("avgpool2_q4", "upsample_bilinear_q4")) / n
print("fnet convs + pool/resize us/frame", 1e3 * fn)
