"""Does HIP-graph replay beat eager launches?  Clip forward (config 2 / 4 shape) and one streamed frame (config 3 shape),
each captured once with torch.cuda.CUDAGraph (the C-ABI calls are stream-ordered and capturable, tests/test_gpu_round2.py)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crfp_amd import synth
from crfp_amd.model import CRFP
ap = argparse.ArgumentParser(); ap.add_argument("--storage", default="f32"); a = ap.parse_args()
dev = torch.device("cuda:0")
sd = synth.make_state_dict(7)
m = CRFP.CRFP_DSV(device=dev, mid_channels=32)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
m.storage = a.storage
m = m.to(dev).eval()
lrs, fvs, mks = (torch.from_numpy(x).to(dev) for x in synth.make_clip(1234, 1, 7, 180, 320, fv_size=96))
eng = m.engine()


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


with torch.no_grad():
    eager = timeit(lambda: eng.forward(lrs, fvs, mks))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = eng.forward(lrs, fvs, mks)
    replay = timeit(g.replay)
    print(f"{a.storage} clip 7x180x320: eager {1e3 * eager:.3f} ms  graph replay {1e3 * replay:.3f} ms")
    # streaming: steady-state frame
    eng.clear_states()
    for i in range(3):
        eng.stream_frame(lrs[0, i], fvs[0, i], mks[0, i])
    lr, fv, mk = lrs[0, 3].clone(), fvs[0, 3].clone(), mks[0, 3].clone()
    e2 = timeit(lambda: eng.stream_frame(lr, fv, mk), 30)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        o2 = eng.stream_frame(lr, fv, mk)
    r2 = timeit(g2.replay, 30)
    print(f"{a.storage} stream frame 180x320: eager {1e3 * e2:.3f} ms  graph replay {1e3 * r2:.3f} ms")
