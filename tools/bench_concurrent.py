#!/usr/bin/env python3
"""Throughput with C independent clips in flight on C HIP streams of one GPU (BASELINE config 4 style:
several clips per GPU).  Each clip has its own engine workspace; kernels of different clips fill each
other's tails / pipeline bubbles.  python tools/bench_concurrent.py --clips 1 2 4"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crfp_amd import synth
from crfp_amd.engine import DSVEngine

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, nargs="+", default=[1, 2, 4])
ap.add_argument("--steps", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(7).items()}
for C in a.clips:
    engs = [DSVEngine(sd, dev) for _ in range(C)]
    streams = [torch.cuda.Stream() for _ in range(C)]
    data = [tuple(torch.from_numpy(x).to(dev) for x in synth.make_clip(1234 + i, 1, 7, 180, 320)) for i in range(C)]
    for _ in range(2):
        for e, s, d in zip(engs, streams, data):
            with torch.cuda.stream(s):
                e.forward(*d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        for e, s, d in zip(engs, streams, data):
            with torch.cuda.stream(s):
                e.forward(*d)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"clips in flight {C}: {C * a.steps * 7 / dt:8.1f} frames/s  ({1e3 * dt / a.steps:.2f} ms per round of {C} clips)")
    del engs, data
