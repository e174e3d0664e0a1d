"""Debug aid: does the engine (stream B) disturb an unrelated torch elementwise kernel chain on stream A?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CRFP_SIDE_STREAM"] = "0"
from crfp_amd import synth
from crfp_amd.engine import DSVEngine

t, h, w = int(os.environ.get("T", 2)), 180, 320
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(7).items()}
d = torch.device("cuda:0")
eng = DSVEngine(sd, d)
lrs, fvs, mks = synth.make_clip(100, 1, t, h, w, fv_size=96, sigma_t=10.0)
clip = [torch.from_numpy(x).to(d) for x in (lrs, fvs, mks)]
eng_ref = eng.forward(*clip).clone()
x = torch.randn(16 << 20, device=d)
def chain(x):
    y = x
    for _ in range(6):
        y = torch.sin(y) * 1.25 + x * 0.5
    return y
ref = chain(x).clone()
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
bad_v = bad_e = 0
for rep in range(10):
    with torch.cuda.stream(sb):
        outs = [eng.forward(*clip) for _ in range(3)]
    with torch.cuda.stream(sa):
        ys = [chain(x) for _ in range(12)]
    torch.cuda.synchronize()
    bad_v += sum(int((y != ref).sum().item()) for y in ys)
    bad_e += sum(int(((o - eng_ref).abs() > 0).sum().item()) for o in outs)
print("victim(torch elementwise) wrong elements:", bad_v, " engine wrong elements:", bad_e,
      {k: v for k, v in os.environ.items() if k.startswith("CRFP_")})
