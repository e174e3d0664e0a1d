#!/usr/bin/env python3
"""Soak of the C-ABI's threading contract: K generations of W worker threads, each with its own torch stream and engine (fp32 and bf16
alternating), running clip forwards and streamed frames concurrently; every output is compared bit for bit with the sequential result; the side-stream
registry must stop growing after the first generation (leased tables are reused).  usage: python tools/soak_threads.py [--gens 8] [--workers 4] [--rounds 6]"""
import argparse, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from crfp_amd import _lib, synth
from crfp_amd.engine import DSVEngine

ap = argparse.ArgumentParser()
ap.add_argument("--gens", type=int, default=8)
ap.add_argument("--workers", type=int, default=4)
ap.add_argument("--rounds", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda:0")
T = torch.from_numpy
L = _lib.lib()
sd = {k: T(v.copy()) for k, v in synth.make_state_dict(7).items()}
geo = [(2, 3, 36, 64), (1, 4, 27, 45), (3, 2, 24, 40), (1, 3, 33, 47)]
jobs = []
with torch.no_grad():
    for k in range(a.workers):
        storage = "bf16" if k & 1 else "f32"
        n, t, h, w = geo[k % len(geo)]
        eng = DSVEngine(sd, dev, storage=storage)
        cs = [synth.make_clip(500 + 10 * k + i, 1, t, h, w, fv_size=64) for i in range(n)]
        clip = tuple(T(np.concatenate([c[q] for c in cs], 0)).to(dev) for q in range(3))
        want = eng.forward(*clip).clone()
        eng.clear_states()
        wstream = torch.stack([eng.stream_frame(clip[0][0, i], clip[1][0, i], clip[2][0, i]).clone() for i in range(t)])
        eng.clear_states()
        jobs.append((eng, clip, want, wstream))
torch.cuda.synchronize()
errors, calls = [], [0]
lock = threading.Lock()


def worker(k, bar):
    eng, (lrs, fvs, mks), want, wstream = jobs[k]
    st = torch.cuda.Stream(device=dev)
    try:
        with torch.no_grad(), torch.cuda.stream(st):
            for r in range(a.rounds):
                bar.wait(timeout=120)
                out = eng.forward(lrs, fvs, mks).clone()
                eng.clear_states()
                fr = torch.stack([eng.stream_frame(lrs[0, i], fvs[0, i], mks[0, i]).clone() for i in range(lrs.shape[1])])
                st.synchronize()
                if not torch.equal(out, want) or not torch.equal(fr, wstream) or eng.overflowed():
                    errors.append((k, r))
                with lock:
                    calls[0] += 1 + lrs.shape[1]
    except Exception as e:   # noqa: BLE001
        errors.append((k, repr(e)))
        bar.abort()


t0 = time.time()
tables = []
for g in range(a.gens):
    bar = threading.Barrier(a.workers)
    ths = [threading.Thread(target=worker, args=(k, bar)) for k in range(a.workers)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=900)
    torch.cuda.synchronize()
    tables.append(L.crfp_debug_side_tables())
print(f"{a.gens} generations x {a.workers} threads x {a.rounds} rounds: {calls[0]} library calls in {time.time() - t0:.1f} s, "
      f"{len(errors)} mismatches {errors[:4]}; fp32 side-stream tables after each generation: {tables}")
assert not errors
assert len(set(tables[1:])) == 1 and tables[-1] <= tables[0] + 0, tables   # no growth after the first generation
assert L.crfp_shutdown() == 0
with torch.no_grad():
    for eng, clip, want, _ in jobs:
        assert torch.equal(eng.forward(*clip), want)
print("after crfp_shutdown(): first calls identical")
