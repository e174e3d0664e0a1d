"""FNet alone (crfp_fnet_forward) for 1 and 6 pairs at 180 x 320, and the one-frame-per-call rates it bounds.  usage: python tools/fnet_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crfp_amd import synth
from crfp_amd.engine import DSVEngine

dev = torch.device("cuda:0")
lrs, fvs, mks = (torch.from_numpy(a).to(dev) for a in synth.make_clip(3, 1, 40, 180, 320, fv_size=96, sigma_t=50.0))
mks = mks.contiguous()
for storage in ("f32", "bf16"):
    eng = DSVEngine({k: torch.from_numpy(v) for k, v in synth.make_state_dict(7).items()}, dev, storage=storage)
    row = {}
    for n in (1, 6):
        cur, prev = lrs[0, 1:1 + n].contiguous(), lrs[0, :n].contiguous()
        for _ in range(5):
            eng.compute_flow(cur, prev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            eng.compute_flow(cur, prev)
        torch.cuda.synchronize()
        row[f"fnet_{n}pair_us"] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
    for resident in (False, True):
        eng.inputs_resident = resident
        best = 0.0
        for _ in range(3):
            eng.clear_states()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(40):
                eng.stream_frame(lrs[0, i], fvs[0, i], mks[0, i])
            torch.cuda.synchronize()
            best = max(best, 40 / (time.perf_counter() - t0))
        row["stream_resident_fps" if resident else "stream_fps"] = round(best, 1)
    print(storage, row, flush=True)
