"""Debug aid: run K independent clips concurrently on K HIP streams (single-stream schedule inside each call) and
compare every output with the serial result -- separates 'kernels misbehave under concurrency' from 'bad event logic'."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CRFP_SIDE_STREAM"] = "0"
from crfp_amd import synth
from crfp_amd.engine import DSVEngine

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
t, h, w = int(os.environ.get("T", 7)), 180, 320
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(7).items()}
d = torch.device("cuda:0")
engs = [DSVEngine(sd, d) for _ in range(K)]
clips = []
for k in range(K):
    lrs, fvs, mks = synth.make_clip(100 + k, 1, t, h, w, fv_size=96, sigma_t=10.0)
    clips.append([torch.from_numpy(x).to(d) for x in (lrs, fvs, mks)])
refs = [engs[k].forward(*clips[k]).clone() for k in range(K)]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(K)]
worst = 0.0
for rep in range(int(os.environ.get("REPS", 6))):
    outs = [None] * K
    for k in range(K):
        with torch.cuda.stream(streams[k]):
            outs[k] = engs[k].forward(*clips[k])
    torch.cuda.synchronize()
    for k in range(K):
        dd = (outs[k] - refs[k]).abs().max().item()
        worst = max(worst, dd)
        if dd > 0 and os.environ.get("VERBOSE"):
            per = (outs[k] - refs[k]).abs()[0].amax(dim=(1, 2, 3)).tolist()
            print(f"rep {rep} clip {k}: per-frame max diff", " ".join(f"{x:.1e}" for x in per))
print("concurrent clips", K, "T", t, {k: v for k, v in os.environ.items() if k.startswith("CRFP_")}, "worst", worst)
