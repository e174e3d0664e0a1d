"""Debug aid: compare the two-stream schedule of crfp_dsv_forward_clip with the single-stream one."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crfp_amd import synth
from crfp_amd.engine import DSVEngine

t = int(sys.argv[1]) if len(sys.argv) > 1 else 2
h, w = 180, 320
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(7).items()}
lrs, fvs, mks = synth.make_clip(1234, 1, t, h, w, fv_size=96, sigma_t=10.0)
d = torch.device("cuda:0")
eng = DSVEngine(sd, d)
L, Fv, M = [torch.from_numpy(x).to(d) for x in (lrs, fvs, mks)]
os.environ["CRFP_SIDE_STREAM"] = "0"
ref = eng.forward(L, Fv, M).clone()
torch.cuda.synchronize()
os.environ["CRFP_SIDE_STREAM"] = "1"
worst = 0.0
for rep in range(6):
    out = eng.forward(L, Fv, M)
    torch.cuda.synchronize()
    dif = (out - ref).abs()[0]
    worst = max(worst, dif.max().item())
    if os.environ.get("VERBOSE"):
        for i in range(t):
            m = dif[i].amax(0)
            nz = torch.nonzero(m > 0)
            if nz.numel():
                print(f"rep {rep} frame {i}: max {m.max().item():.3e} n={nz.shape[0]} rows {nz[:,0].min().item()}..{nz[:,0].max().item()} cols {nz[:,1].min().item()}..{nz[:,1].max().item()}")
            else:
                print(f"rep {rep} frame {i}: identical")
print("PARTS", os.environ.get("CRFP_SIDE_PARTS"), "LAG", os.environ.get("CRFP_SIDE_LAG"), "worst", worst)
