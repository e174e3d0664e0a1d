"""Debug aid: one clip on stream A while stream B runs unrelated torch work (copy / matmul); compare with serial."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CRFP_SIDE_STREAM"] = "0"
from crfp_amd import synth
from crfp_amd.engine import DSVEngine

mode = sys.argv[1] if len(sys.argv) > 1 else "copy"
t, h, w = 7, 180, 320
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(7).items()}
d = torch.device("cuda:0")
eng = DSVEngine(sd, d)
lrs, fvs, mks = synth.make_clip(100, 1, t, h, w, fv_size=96, sigma_t=10.0)
clip = [torch.from_numpy(x).to(d) for x in (lrs, fvs, mks)]
ref = eng.forward(*clip).clone()
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
big = torch.randn(64 << 20, device=d)
big2 = torch.empty_like(big)
ma = torch.randn(4096, 4096, device=d)
worst = 0.0
for rep in range(6):
    with torch.cuda.stream(sb):
        for _ in range(40 if mode == "copy" else 20):
            if mode == "copy":
                big2.copy_(big)
            elif mode == "matmul":
                mb = ma @ ma
            elif mode == "tiny":
                for _ in range(50): big2[:1024].add_(1.0)
    with torch.cuda.stream(sa):
        out = eng.forward(*clip)
    torch.cuda.synchronize()
    dd = (out - ref).abs().max().item()
    worst = max(worst, dd)
print("disturbance", mode, os.environ.get("CRFP_CONV_MODE"), "worst", worst)
