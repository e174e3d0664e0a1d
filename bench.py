#!/usr/bin/env python3
"""Headline benchmark: SR frames/s of the CRFP_DSV recurrent x8 path on synthetic 7-frame
180x320 -> 1440x2560 clips (BASELINE.json configs[1]), one process per GPU.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus 8 --steps 20 --warmup 3

A step = one clip (7 frames) per rank through crfp_dsv_forward_clip with inputs resident in HBM.
Clips are independent, so ranks share nothing on the data path (weak scaling); the only collective
is one RCCL all-reduce of the PSNR sums after the timed region.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      dominant kernel family by GPU time: algorithmic bytes (or flops) / hipEvent-measured
                duration, from a second, instrumented pass of the same steps (every launch bracketed
                by hipEvents on its own stream inside libcrfp_hip.so: crfp_prof_*).
  warp_dcn      the same for the flow_warp + DCNv2 gather kernels (the north star's 60 % HBM target).
  kernels       per-kernel table (ms per clip, achieved GB/s and TFLOP/s).
  cpu_baseline  the oracle (CPU port of the reference path) timed on this host on a bounded sample.
  parity        max|HIP - oracle| and PSNR-Y delta on that same sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)
HBM_COPY_CEILING_GBS = 6290.0
F32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak
# The wide convs run as 3 fp16 MFMA products per algorithmic fp32 MAC (split-fp16 "f16x3", fp32-grade result:
# DESIGN.md section 3.1), or 6 bf16 products with CRFP_CONV_MODE=bf16x6; fp16 and bf16 MFMAs share the 2.5 PF dense
# peak, so the algorithmic-flop ceiling is that peak / 3 (/ 6).
SPLIT_F16_EQUIV_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0
SPLIT_BF16_EQUIV_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0


def kernel_family(name: str) -> str:
    if name.startswith("conv_mfma"):
        return "conv3x3_mfma"
    if name.startswith("conv_narrow"):
        return "conv3x3_narrow"
    return name


# hipEvent launch-site family -> rocprofv3 kernel names (for the PMC traffic lookup)
ROCPROF_NAMES = {"conv3x3_mfma": ("conv3x3_split_kernel", "conv3x3_mfma_kernel"), "conv3x3_narrow": ("conv3x3_narrow_kernel",),
                 "dcnv2_g8_c32": ("dcn_g8_kernel", "dcn_g8_pipe_kernel"), "dcnv2_shared_c4": ("dcn3_kernel",),
                 "flow_warp_q4_c4": ("flow_warp_p4_kernel",), "flow_warp_q4_c32": ("flow_warp_p4_kernel",),
                 "flow_warp_q4_c24": ("flow_warp_p4_kernel",), "flow_warp_q4_c32+c24": ("flow_warp_p4_dual_kernel",), "hr_prep_up8_blend": ("hr_prep_kernel",)}


def pmc_traffic(family: str):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (FETCH_SIZE x2 + WRITE_SIZE, per
    MI355X_MICROARCH.md), averaged over the family's kernels; None when no summary is present."""
    path = os.path.join(ROOT, "profiles", "pmc_summary_latest.json")
    if not os.path.exists(path) or family not in ROCPROF_NAMES:
        return None
    tot = calls = 0.0
    for r in json.load(open(path)):
        if any(r["kernel"].startswith(n) for n in ROCPROF_NAMES[family]) and r.get("hbm_MB_per_launch_corrected") is not None:
            tot += r["hbm_MB_per_launch_corrected"] * 1e6 * r["calls"]
            calls += r["calls"]
    return tot / calls if calls else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=7)
    ap.add_argument("--lr-h", type=int, default=180)
    ap.add_argument("--lr-w", type=int, default=320)
    ap.add_argument("--fv-size", type=int, default=96)
    ap.add_argument("--sigma-t", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-multi-stream", action="store_true")
    ap.add_argument("--cpu-sample-frames", type=int, default=3)
    ap.add_argument("--cpu-timeout", type=float, default=240.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")   # RCCL over xGMI
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from crfp_amd import _lib, synth
    from crfp_amd.model import CRFP

    t, h, w = args.frames, args.lr_h, args.lr_w
    sd = synth.make_state_dict(7)
    model = CRFP.CRFP_DSV(device=dev, mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    model = model.to(dev).eval()
    # every rank gets its own clip (seed offset by rank): independent units, no data-path collective
    lrs_np, fvs_np, mks_np = synth.make_clip(1234 + rank, 1, t, h, w, fv_size=args.fv_size, sigma_t=args.sigma_t)
    lrs, fvs, mks = (torch.from_numpy(a).to(dev) for a in (lrs_np, fvs_np, mks_np))
    eng = model.engine()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            out = eng.forward(lrs, fvs, mks)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = eng.forward(lrs, fvs, mks)
        barrier()
        elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # PSNR reduction (the only collective of the path): PSNR-Y of the SR frames against the synthetic HR
    # scene inside the fovea window is not meaningful without trained weights, so reduce the raw sums.
    from crfp_amd import ops
    acc = ops.sq_err_sums(out[0], fvs[0]).clone()
    vec = torch.cat([acc, torch.tensor([float(t)], dtype=torch.float64, device=dev)])
    if dist is not None:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)

    result = {
        "metric": "sr_frames_per_sec", "value": world * args.steps * t / elapsed, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: single MI355X, 7-frame 180x320 -> 1440x2560 x8 SR, "
                               "batch=1, fp32, sigma_T=10 (one clip per GPU per step; clips sharded over GPUs)",
                   "conv_arithmetic": "fp32 in / fp32 accumulate / fp32 out; products on the fp16 MFMA via an exact 2-term split "
                                      "(3 MFMAs per MAC, error at the fp32 summation-order floor, see `parity`)",
                   "frames_per_clip": t, "lr": [h, w], "sr": [8 * h, 8 * w], "fv_size": args.fv_size,
                   "sigma_t": args.sigma_t, "clips_per_gpu_per_step": 1, "parallelism": f"clip-sharded x{world}"},
        "per_gpu_frames_per_sec": args.steps * t / elapsed,
    }

    if rank == 0 and not args.no_kernel_profile:
        L = _lib.lib()
        L.crfp_prof_reset()
        L.crfp_prof_enable(1)
        psteps = min(args.steps, 5)
        with torch.no_grad():
            for _ in range(psteps):
                eng.forward(lrs, fvs, mks)
        torch.cuda.synchronize()
        recs = _lib.prof_report()
        L.crfp_prof_enable(0)
        L.crfp_prof_reset()
        fam = {}
        for r in recs:
            f = fam.setdefault(kernel_family(r["name"]), dict(launches=0, ms=0.0, bytes=0.0, flops=0.0))
            f["launches"] += r["launches"]; f["ms"] += r["total_ms"]; f["bytes"] += r["bytes"]; f["flops"] += r["flops"]
        total_ms = sum(f["ms"] for f in fam.values())
        table = []
        for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
            s = f["ms"] * 1e-3
            table.append({"kernel": name, "launches_per_clip": f["launches"] / psteps, "ms_per_clip": f["ms"] / psteps,
                          "avg_us": 1e3 * f["ms"] / f["launches"], "share": f["ms"] / total_ms,
                          "GBps": f["bytes"] / s / 1e9 if s > 0 else 0.0, "TFLOPs": f["flops"] / s / 1e12 if s > 0 else 0.0})
        result["kernels"] = table
        result["kernel_ms_per_clip"] = total_ms / psteps
        dom = table[0]
        domf = fam[dom["kernel"]]
        if dom["kernel"] == "conv3x3_mfma":
            mode = os.environ.get("CRFP_CONV_MODE", "f16x3")
            f32_mode = mode == "f32"
            peak = {"f32": F32_MFMA_PEAK_TFLOPS, "bf16x6": SPLIT_BF16_EQUIV_PEAK_TFLOPS}.get(mode, SPLIT_F16_EQUIV_PEAK_TFLOPS)
            note = {"f32": "fp32 MFMA",
                    "bf16x6": "algorithmic fp32 flops; executed as 6 bf16 MFMA products per MAC (split-bf16), so peak = 2.5 PF / 6"}.get(
                        mode, "algorithmic fp32 flops; executed as 3 fp16 MFMA products per MAC (split-fp16, fp32-grade), so peak = "
                              "2.5 PF dense fp16 / 3; PMC: MFMA pipe ~28 % busy (profiles/*_mfma_lds_util.txt), the rest is operand traffic through LDS (1.0 ds_read_b128 per MFMA), load wait, split and the tile-granularity tail")
            result["roofline"] = {"kernel": dom["kernel"], "bound": "mfma", "achieved": dom["TFLOPs"],
                                  "peak": peak, "unit": "TFLOP/s", "frac": dom["TFLOPs"] / peak,
                                  "traffic": pmc_traffic(dom["kernel"]), "avg_launch_us": dom["avg_us"],
                                  "algorithmic_flops_per_launch": domf["flops"] / domf["launches"],
                                  "frac_of_fp32_mfma_peak": dom["TFLOPs"] / F32_MFMA_PEAK_TFLOPS,
                                  "conv_scheme": mode, "note": note}
        else:
            result["roofline"] = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": dom["GBps"] / HBM_PEAK_GBS, "traffic": pmc_traffic(dom["kernel"]),
                                  "avg_launch_us": dom["avg_us"],
                                  "algorithmic_bytes_per_launch": domf["bytes"] / domf["launches"]}
        gat = [f for n, f in fam.items() if n.startswith("flow_warp") or n.startswith("dcnv2")]
        if gat:
            gb = sum(f["bytes"] for f in gat); gs = sum(f["ms"] for f in gat) * 1e-3
            # dcn_3 through the API moves 18 offset + 9 mask channels (the reference tiles 2+1 channels 9x); the kernel
            # reads the compact 2+1: SURVEY 8(d) asks for both figures
            d3 = fam.get("dcnv2_shared_c4")
            gb_api = gb + (d3["launches"] * (8 * h) * (8 * w) * 24 * 4.0 if d3 else 0.0)
            result["warp_dcn"] = {"bound": "hbm", "achieved": gb / gs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": gb / gs / 1e9 / HBM_PEAK_GBS, "frac_of_copy_ceiling": gb / gs / 1e9 / HBM_COPY_CEILING_GBS,
                                  "achieved_api_tensor_bytes": gb_api / gs / 1e9, "frac_api_tensor_bytes": gb_api / gs / 1e9 / HBM_PEAK_GBS,
                                  "ms_per_clip": 1e3 * gs / psteps,
                                  "traffic": {n: pmc_traffic(n) for n in fam if n.startswith("flow_warp") or n.startswith("dcnv2")},
                                  "note": "dcn_3 priced at its compact 2+1 offset/mask channels (162.2 MB/frame), "
                                          "not the 9x-replicated API tensors (516.1 MB/frame)"}

    if rank == 0 and world == 1 and not args.no_multi_stream:
        # extra (not the headline): independent clips in flight on separate HIP streams of the same GPU fill
        # each other's tails and pipeline bubbles (BASELINE config 4 runs 4 clips per GPU)
        from crfp_amd.engine import DSVEngine
        sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
        ms = {}
        exact = True
        for C in (2, 4):
            engs = [DSVEngine(sdt, dev) for _ in range(C)]
            streams = [torch.cuda.Stream(device=dev) for _ in range(C)]
            with torch.no_grad():
                for it in range(1 + min(args.steps, 5)):
                    if it == 1:
                        torch.cuda.synchronize()
                        tm = time.perf_counter()
                    outs = []
                    for e, st in zip(engs, streams):
                        with torch.cuda.stream(st):
                            outs.append(e.forward(lrs, fvs, mks))
                torch.cuda.synchronize()
            ms[str(C)] = C * min(args.steps, 5) * t / (time.perf_counter() - tm)
            # concurrent kernels must not disturb each other: every in-flight clip == the sequential result, bit for bit
            exact = exact and all(bool(torch.equal(o, out)) for o in outs)
            del engs, outs
        ms["bit_exact_vs_sequential"] = exact
        result["multi_stream_frames_per_sec"] = ms

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # The oracle (CPU port of the reference path; checker / baseline only, never on the product
        # path) runs in a child process so that a mis-sized host cannot stall the bench: bounded
        # sample, bounded threads, hard timeout.
        import subprocess
        import tempfile
        from oracle import crfp_oracle as orc
        ns = max(2, min(args.cpu_sample_frames, t))
        tmp = os.path.join(tempfile.mkdtemp(), "oracle_sample.npz")
        cmd = [sys.executable, "-m", "oracle.run_sample", "--frames", str(ns), "--h", str(h), "--w", str(w),
               "--fv-size", str(args.fv_size), "--sigma-t", str(args.sigma_t), "--clip-seed", "1234",
               "--clip-frames", str(t), "--out", tmp]
        try:
            subprocess.run(cmd, cwd=ROOT, timeout=args.cpu_timeout, check=True)
            z = np.load(tmp)
            ref, cpu_s = torch.from_numpy(z["out"]), float(z["seconds"])
            with torch.no_grad():
                got = eng.forward(lrs[:, :ns], fvs[:, :ns], mks[:, :ns]).cpu()
            d = float((got - ref).abs().max())
            hr = torch.from_numpy(np.clip(fvs_np[:, :ns], 0, 1))
            py_ref = np.mean([orc.psnr_rgb_and_y(ref[0, i:i + 1], hr[0, i:i + 1])[1] for i in range(ns)])
            py_got = np.mean([orc.psnr_rgb_and_y(got[0, i:i + 1], hr[0, i:i + 1])[1] for i in range(ns)])
            result["cpu_baseline"] = {"value": ns / cpu_s, "unit": "frames/s", "cores": int(z["threads"]), "kind": "port",
                                      "sample": f"first {ns} frames of the same 180x320 clip (1 first frame + {ns - 1} "
                                                f"steady-state frames), oracle/crfp_oracle.py on torch-CPU fp32, {cpu_s:.1f} s wall",
                                      "host_cpus": os.cpu_count(), "usable_cpus": int(z["usable_cpus"])}
            result["parity"] = {"max_abs_diff_vs_oracle": d, "tolerance": 1e-3, "frames": ns,
                                "psnr_y_delta_db": float(abs(py_ref - py_got))}
        except (subprocess.TimeoutExpired, subprocess.CalledProcessError) as e:
            result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": None, "kind": "port",
                                      "sample": f"oracle sample of {ns} frames did not finish: {type(e).__name__}"}

    if rank == 0:
        result["psnr_reduce"] = {"sum_sq_err": float(vec[0]), "frames": float(vec[2])}
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
