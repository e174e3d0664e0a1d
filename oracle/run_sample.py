"""ORACLE -- test infrastructure only.  Runs the CPU restatement on the first frames of the bench
clip in a separate process (so bench.py can bound it with a timeout) and saves output + wall time.

    python -m oracle.run_sample --frames 3 --h 180 --w 320 --threads 16 --out /tmp/o.npz
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crfp_amd import synth  # noqa: E402
from oracle import crfp_oracle as orc  # noqa: E402


def usable_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 quota: "<quota> <period>" or "max <period>"
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--h", type=int, default=180)
    ap.add_argument("--w", type=int, default=320)
    ap.add_argument("--fv-size", type=int, default=96)
    ap.add_argument("--sigma-t", type=float, default=10.0)
    ap.add_argument("--clip-seed", type=int, default=1234)
    ap.add_argument("--clip-frames", type=int, default=7)
    ap.add_argument("--weights-seed", type=int, default=7)
    ap.add_argument("--offset-std", type=float, default=None, help="synth.make_state_dict(offset_std=): 0.02 = SURVEY 8(d)'s N(0, 0.02) DCN heads")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--storage", choices=("f32", "bf16"), default="f32", help="bf16: the storage-rounding twin")
    ap.add_argument("--warmup", type=int, default=0, help="untimed passes before the timed one (BASELINE.md section 4: 1)")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    # every usable CPU (affinity / cgroup quota); beyond ~64 threads the small 180x320 convs stop scaling
    threads = a.threads or min(64, usable_cpus())
    torch.set_num_threads(threads)
    sd = synth.make_state_dict(a.weights_seed, offset_std=a.offset_std)
    lrs, fvs, mks = synth.make_clip(a.clip_seed, 1, a.clip_frames, a.h, a.w, fv_size=a.fv_size, sigma_t=a.sigma_t)
    P = orc.load_numpy_state(sd)
    T = torch.from_numpy
    import contextlib
    ctx = orc.bf16_storage() if a.storage == "bf16" else contextlib.nullcontext()
    if a.storage == "bf16":
        P = orc.bf16_weights(P)
    with torch.no_grad(), ctx:
        for _ in range(a.warmup):
            orc.crfp_dsv_forward(P, T(lrs[:, :a.frames]), T(fvs[:, :a.frames]), T(mks[:, :a.frames]))
        t0 = time.perf_counter()
        out = orc.crfp_dsv_forward(P, T(lrs[:, :a.frames]), T(fvs[:, :a.frames]), T(mks[:, :a.frames]))
        dt = time.perf_counter() - t0
    np.savez(a.out, out=out.numpy(), seconds=dt, threads=threads, usable_cpus=usable_cpus())


if __name__ == "__main__":
    main()
