#!/usr/bin/env python3
"""Same-box A/B of library variants on the BASELINE configs[1] clip (or --storage bf16 / other geometry).

  python tools/ab_sites.py [--storage f32] [--sites dcnv2_shared,flow_warp] [--rounds 2] NAME=ENV1=V1,ENV2=V2 ...

Every variant runs in its own child process (the library reads its switches once), alternating over --rounds so that box
drift hits all of them alike.  `lib=<path>` inside a variant selects another build of the C-ABI (CRFP_HIP_LIB); `lab`
is short for the lab library; `AB_RESIDENT=1` sets CRFP_DSV_INPUTS_RESIDENT in --mode stream.  Prints per variant: wall ms per clip (two-stream schedule), kernel-sum ms (single-stream
instrumented pass) and the avg us of every launch site whose name contains one of --sites, plus a digest of the output."""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(a):
    import time
    import torch
    sys.path.insert(0, ROOT)
    from crfp_amd import _lib, synth
    from crfp_amd.model import CRFP
    dev = torch.device("cuda:0")
    sd = synth.make_state_dict(7, offset_std=(a.offset_std if a.offset_std > 0 else None))
    m = CRFP.CRFP_DSV(device=dev, mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m.storage = a.storage
    eng = m.to(dev).eval().engine()
    lrs, fvs, mks = (torch.from_numpy(x).to(dev) for x in synth.make_clip(1234, 1, a.t, a.h, a.w, fv_size=a.fv, sigma_t=10.0))
    L = _lib.lib()
    eng.inputs_resident = bool(int(os.environ.get("AB_RESIDENT", "0")))   # CRFP_DSV_INPUTS_RESIDENT (stream mode)
    if a.mode == "stream":      # one frame per call (BASELINE config 3's call pattern), a.t calls per "step"
        mk8 = mks.contiguous()

        def run():
            eng.clear_states()
            o = None
            for i in range(a.t):
                o = eng.stream_frame(lrs[0, i], fvs[0, i], mk8[0, i])
            return o
        eng.forward = lambda *_: run()
    with torch.no_grad():
        for _ in range(3):
            out = eng.forward(lrs, fvs, mks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = eng.forward(lrs, fvs, mks)
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0) / a.steps
        L.crfp_prof_reset(); L.crfp_prof_enable(1)
        for _ in range(a.steps):
            eng.forward(lrs, fvs, mks)
        torch.cuda.synchronize()
        recs = _lib.prof_report(512)
        L.crfp_prof_enable(0)
    sites = {r["name"]: 1e3 * r["total_ms"] / r["launches"] for r in recs}
    counts = {r["name"]: r["launches"] / a.steps for r in recs}
    res = {"wall_ms": wall, "kernel_ms": sum(r["total_ms"] for r in recs) / a.steps, "sites": sites, "counts": counts,
           "digest": hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12], "finite": bool(torch.isfinite(out).all())}
    print("ABRESULT " + json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="*")
    ap.add_argument("--storage", default="f32")
    ap.add_argument("--sites", default="")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--h", type=int, default=180); ap.add_argument("--w", type=int, default=320)
    ap.add_argument("--t", type=int, default=7); ap.add_argument("--fv", type=int, default=96)
    ap.add_argument("--mode", default="clip", choices=("clip", "stream"))
    ap.add_argument("--all-sites", action="store_true", help="print every launch site (sorted by time) for the first variant")
    ap.add_argument("--offset-std", type=float, default=0.0, help="0.02 = SURVEY 8(d)'s N(0, 0.02) dcn_offset / dcn_mask heads (default: the stress weights)")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a)
    variants = []
    for v in a.variants or ["base="]:
        name, _, envs = v.partition("=")
        env = {}
        for kv in filter(None, envs.split(",")):
            k, _, val = kv.partition("=")
            if k == "lab":
                env["CRFP_HIP_LIB"] = os.path.join(ROOT, "crfp_amd", "libcrfp_hip_lab.so")
            elif k == "lib":
                env["CRFP_HIP_LIB"] = val if os.path.isabs(val) else os.path.join(ROOT, val)
            else:
                env[k] = val
        variants.append((name, env))
    want = [s for s in a.sites.split(",") if s]
    acc = {n: [] for n, _ in variants}
    for r in range(a.rounds):
        for name, env in variants:
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--storage", a.storage, "--steps", str(a.steps), "--h", str(a.h),
                   "--w", str(a.w), "--t", str(a.t), "--fv", str(a.fv), "--mode", a.mode, "--offset-std", str(a.offset_std)]
            p = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("ABRESULT ")]
            if not line:
                print(f"{name}: FAILED\n{p.stdout[-800:]}\n{p.stderr[-1500:]}")
                continue
            acc[name].append(json.loads(line[-1][9:]))
    for name, env in variants:
        rs = acc[name]
        if not rs:
            continue
        wall = min(r["wall_ms"] for r in rs)
        kern = min(r["kernel_ms"] for r in rs)
        sel = {}
        for s in rs[0]["sites"]:
            if any(w in s for w in want):
                sel[s] = min(r["sites"][s] for r in rs)
        if a.all_sites and name == variants[0][0]:
            tot = sum(rs[0]["sites"][k] * rs[0]["counts"][k] for k in rs[0]["sites"])
            for k in sorted(rs[0]["sites"], key=lambda k: -rs[0]["sites"][k] * rs[0]["counts"][k]):
                us = min(r["sites"][k] for r in rs)
                print(f"    {k:44s} {rs[0]['counts'][k]:6.1f} x {us:7.1f} us = {us * rs[0]['counts'][k] / 1e3:7.3f} ms  {100 * rs[0]['sites'][k] * rs[0]['counts'][k] / tot:5.1f} %")
        print(f"{name:22s} wall {wall:7.3f} ms  kernels {kern:7.3f} ms  digest {rs[0]['digest']} finite {rs[0]['finite']}  "
              + "  ".join(f"{k}={v:.1f}" for k, v in sorted(sel.items())), flush=True)


if __name__ == "__main__":
    main()
