"""Frames/s of n sequences streamed in lock-step (crfp_dsv_stream_batch, one frame of each per call) against one sequence per engine.
usage: python tools/bench_stream_batch.py [f32|bf16]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crfp_amd import synth
from crfp_amd.engine import DSVEngine

storage = sys.argv[1] if len(sys.argv) > 1 else "bf16"
h, w, T = 180, 320, 24
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(7).items()}
clips = [synth.make_clip(100 + s, 1, T, h, w, fv_size=96, sigma_t=50.0) for s in range(4)]
for resident in (False, True):
    for n in (1, 2, 4):
        lrs, fvs, mks = (torch.from_numpy(np.concatenate([c[k] for c in clips[:n]], 0)).to(dev) for k in range(3))
        fr = [(lrs[:, i].contiguous(), fvs[:, i].contiguous(), mks[:, i].contiguous().view(torch.uint8)) for i in range(T)]
        eng = DSVEngine(sd, dev, storage=storage)
        eng.inputs_resident = resident
        torch.cuda.synchronize()
        with torch.no_grad():
            for rep in range(3):
                if rep == 1:
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                eng.clear_states()
                for i in range(T):
                    eng.stream_frame(*fr[i]) if n > 1 else eng.stream_frame(fr[i][0][0], fr[i][1][0], fr[i][2][0])
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        print(f"{storage} resident={resident} n={n}: {n * T / dt:8.1f} frames/s  ({1e3 * dt / T:.3f} ms per step)", flush=True)
        del eng
