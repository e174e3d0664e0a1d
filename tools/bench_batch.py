"""Lock-step batch (crfp_dsv_forward_batch) against one-clip calls: frames/s for n clips per call at config A's geometry.
usage: python tools/bench_batch.py [f32|bf16|both] [h w]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crfp_amd import synth
from crfp_amd.engine import DSVEngine

which = sys.argv[1] if len(sys.argv) > 1 else "both"
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (180, 320)
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(7).items()}
T = 7
clips = [synth.make_clip(100 + s, 1, T, h, w, fv_size=96) for s in range(8)]
for storage in (("f32", "bf16") if which == "both" else (which,)):
    eng = DSVEngine(sd, dev, storage=storage)
    for n in (1, 2, 4, 8):
        data = tuple(torch.from_numpy(np.concatenate([c[k] for c in clips[:n]], 0)).to(dev) for k in range(3))
        res = {}
        for mode in ("loop", "lockstep"):
            if n == 1 and mode == "lockstep":
                continue
            eng.batch_mode = mode
            with torch.no_grad():
                for _ in range(2):
                    eng.forward(*data)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                it = 6
                for _ in range(it):
                    eng.forward(*data)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / it
            res[mode] = n * T / dt
        print(storage, f"n={n}", {k: round(v, 1) for k, v in res.items()}, flush=True)
    del eng
    torch.cuda.empty_cache()

# two lock-step calls in flight on two streams (n clips each): do the tails of one call fill the other's?
if len(sys.argv) > 4 and sys.argv[4] == "flight":
    for storage in (("f32", "bf16") if which == "both" else (which,)):
        engs = [DSVEngine(sd, dev, storage=storage) for _ in range(2)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for n in (1, 2, 4):
            datas = [tuple(torch.from_numpy(np.concatenate([c[k] for c in clips[g * n:(g + 1) * n]], 0)).to(dev) for k in range(3)) for g in range(2)]
            def step():
                cur = torch.cuda.current_stream()
                for s in streams:
                    s.wait_stream(cur)
                for g in range(2):
                    with torch.cuda.stream(streams[g]):
                        engs[g].forward(*datas[g])
                for s in streams:
                    cur.wait_stream(s)
            with torch.no_grad():
                step(); step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(6):
                    step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 6
            print(storage, f"2 calls in flight x n={n}", round(2 * n * T / dt, 1), flush=True)
