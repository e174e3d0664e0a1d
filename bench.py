#!/usr/bin/env python3
"""Benchmark of the CRFP_DSV recurrent x8 path on MI355X: SR frames/s (BASELINE.json `metric`), one process per GPU.

  python bench.py                                   # headline: BASELINE configs[1], N = 1
  python bench.py --config 3|4|5                    # the bf16 configs (streaming 100 frames / 4 clips per GPU / 4K)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus 8 --steps 20 --warmup 3 [--config 4]

Configs (BASELINE.json `configs`, 0-based index in brackets):
  2 [1]  single MI355X, 7-frame 180x320 -> 1440x2560, batch 1, fp32, sigma^T = 10        (default; the headline)
  3 [2]  100-frame 180x320 streaming recurrent inference (one frame per call), bf16 storage, sigma^T = 50
  4 [3]  32 independent 7-frame clips over 8 GPUs = 4 clips per GPU per step, bf16 storage
  5 [4]  270x480 -> 2160x3840 (4K), 7 frames, bf16 storage
A step = one pass of the hot path over one batch of synthetic input resident in HBM: config 2/5 one clip per rank,
config 4 four clips per rank, config 3 one 100-frame streamed sequence per rank.  Clips / sequences are independent,
so ranks share nothing on the data path (weak scaling); the only collective is one RCCL all-reduce of the PSNR sums
after the timed region.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      dominant kernel family by GPU time: algorithmic flops (or bytes) / hipEvent-measured duration, from a
                second, instrumented single-stream pass of the same steps (every launch bracketed by hipEvents on its
                own stream inside libcrfp_hip.so: crfp_prof_*).  `traffic` = HBM bytes per launch from the committed
                rocprofv3 PMC summary, tagged with its provenance (it is not re-measured in this run).
  warp_dcn      the same for the flow_warp + DCNv2 gather kernels (the north star's 60 % HBM target).
  kernels       per-kernel table (ms per step, achieved GB/s and TFLOP/s).
  strict_f32    (config 2) the same clip with CRFP_DSV_STRICT_F32: plain fp32 MFMA instead of the split-fp16 scheme.
  per_op_us     microseconds per call of the per-operator C-ABI entry points at this config's sizes.
  cpu_baseline  the oracle (CPU port of the reference path; its bf16 twin for the bf16 configs) timed on this host:
                1 warm-up + 1 timed pass over the config-2 clip (or a bounded sample of the bigger configs).
  parity        max|HIP - oracle| (bf16 configs: statistics against the twin) and PSNR-Y delta on that sample.
  other_configs (default run only) BASELINE configs[2..4] -- `--config 3|4|5` -- each as a short leg (3 steps) in a child process
                after the headline's timed region: frames/s, ms per step, parity against the bf16 twin on 3 frames, conv roofline.
  runtime_rig   (default run only) the reference's test_runtime.py measurement: ms per 1080p frame of the regional wiring (one C-ABI call
                per 5-frame clip) and of the whole-frame CRFP_DSV engine on the same rig.
  warp_dcn_8d   SURVEY 8(d)'s figure un-re-scoped: API-tensor bytes of flow_warp x3 + DCNv2 x4 per steady-state frame divided by
                the time of ALL warp / DCN kernels of such a frame, the fused offset-head + DCN kernel included.
  lockstep_batch_frames_per_sec  (clip configs) n = 2 / 4 clips per crfp_dsv_forward_batch call (lock-step launches over the clips), bit-exactness against
                one-clip calls; warp_dcn_8d_spec_weights: the 8(d) figure, frames/s and parity with SURVEY 8(d)'s own N(0, 0.02) DCN heads.
  dcn_g8_alone  (default run only) the DCNv2 kernel of dcn_0/1/2 by itself: CRFP_DCN_FUSED=0 in a child process, SURVEY 8(d) API bytes / its time;
                north_star_per_kernel lists every warp / DCN kernel's fraction of 8 TB/s against the 0.60 target.
  mask_gate     (default run only) what the mask-gated launches skip on this workload and the headline with dense launches instead (child process,
                CRFP_MASK_GATE=0): same output bits, the dense rate is what rounds 1-3 measured.
  cra_engine    (config 2) CRFP_DSV_CRA, the reference's cross-resolution-fusion wiring, on its own one-call schedule: frames/s and the
                difference to the per-operator composition of the same model.
  collectives   which backend ran the barrier / MAX / SUM reductions (CRFP_FORCE_DIST=1 initialises RCCL even with one rank).
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

import benchlegs  # noqa: E402
from benchlegs import (CONFIGS, HBM_PEAK_GBS, ROCPROF_NAMES, kernel_family, pmc_traffic, time_op,  # noqa: E402,F401
                                warp_dcn_8d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--lr-h", type=int, default=None)
    ap.add_argument("--lr-w", type=int, default=None)
    ap.add_argument("--fv-size", type=int, default=None)
    ap.add_argument("--sigma-t", type=float, default=None)
    ap.add_argument("--storage", choices=("f32", "bf16"), default=None)
    ap.add_argument("--clips-per-gpu", type=int, default=None)
    ap.add_argument("--in-flight", type=int, default=None, help="clips of one rank in flight on separate HIP streams (clip mode; "
                    "default 1, config 4: 2 -- independent clips fill each other's kernel tails, bit-identical results)")
    ap.add_argument("--batch-clips", type=int, default=None, help="clips per library call (clip mode): n > 1 = crfp_dsv_forward_batch, the n clips "
                    "in lock-step, one launch per layer (config 4: all 4 clips of a rank; bit-identical per clip to one-clip calls)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-resident", action="store_true", help="stream mode: do not set CRFP_DSV_INPUTS_RESIDENT")
    ap.add_argument("--no-extras", action="store_true", help="skip strict_f32 / multi-stream / per-op / other-config legs")
    ap.add_argument("--no-other-configs", action="store_true", help="default run: skip the short legs of BASELINE configs 3 / 4 / 5")
    ap.add_argument("--cpu-sample-frames", type=int, default=None)
    ap.add_argument("--cpu-timeout", type=float, default=420.0)
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    for k, v in (("t", args.frames), ("h", args.lr_h), ("w", args.lr_w), ("fv", args.fv_size), ("sigma", args.sigma_t),
                 ("storage", args.storage), ("clips", args.clips_per_gpu)):
        if v is not None:
            cfg[k] = v
    custom = any(cfg[k] != CONFIGS[args.config][k] for k in cfg) or (args.in_flight is not None and args.in_flight != cfg.get("flight", 1)) or (
        args.batch_clips is not None and args.batch_clips != cfg.get("batch", 1))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    # CRFP_FORCE_DIST=1: initialise the RCCL group even with ONE rank, so that the barrier and the two all-reduces of this file
    # run on RCCL on a 1-GPU box exactly as they do with N ranks (tests/test_gpu_round3.py starts it under torch.distributed.run)
    if world > 1 or os.environ.get("CRFP_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl")   # RCCL over xGMI

    from crfp_amd import _lib, benchutil, ops, synth
    from crfp_amd.engine import DSVEngine
    from crfp_amd.model import CRFP

    t, h, w, fv, storage, clips, mode = cfg["t"], cfg["h"], cfg["w"], cfg["fv"], cfg["storage"], cfg["clips"], cfg["mode"]
    sd = synth.make_state_dict(7)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    model = CRFP.CRFP_DSV(device=dev, mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True)
    model.load_state_dict({k: v.clone() for k, v in sdt.items()}, strict=True)
    model.storage = storage
    model = model.to(dev).eval()
    # every rank (and every clip of a rank) gets its own clip: independent units, no data-path collective
    data_np = [synth.make_clip(seed, 1, t, h, w, fv_size=fv, sigma_t=cfg["sigma"]) for seed in benchutil.rank_clip_seeds(rank, clips)]
    # clips per library call: the calls of a step each take `bclips` clips as ONE [n, t, ...] batch (the reference's own tensor shape)
    bclips = max(1, min(args.batch_clips if args.batch_clips is not None else cfg.get("batch", 1), clips)) if mode == "clip" else 1
    if clips % bclips:
        raise SystemExit(f"--batch-clips {bclips} does not divide the {clips} clips of a step")
    data1 = [tuple(torch.from_numpy(a).to(dev) for a in d) for d in data_np]   # clip by clip (parity / extras legs)
    data = data1 if bclips == 1 else [tuple(torch.cat([data1[g * bclips + c][k] for c in range(bclips)], 0).contiguous() for k in range(3))
                                      for g in range(clips // bclips)]
    ncalls = len(data)
    eng = model.engine()
    in_flight = args.in_flight if args.in_flight is not None else cfg.get("flight", 1)
    n_flight = max(1, min(in_flight, ncalls)) if mode == "clip" else 1
    engs = [eng] + [DSVEngine(sdt, dev, storage=storage) for _ in range(n_flight - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_flight)] if n_flight > 1 else None
    if mode == "stream":
        mk8 = data[0][2].contiguous()
        # the frames already sit in HBM when a step starts (the bench contract), nothing on the stream produces them: CRFP_DSV_INPUTS_RESIDENT
        # lets the library run the state-independent part of frame i beside frame i - 1 (same bits; --no-resident: the plain call pattern)
        model.inputs_resident = eng.inputs_resident = not args.no_resident

    def step(e=None):
        """one step of this rank; returns the last output tensor(s)"""
        if mode == "stream":
            lrs, fvs, _ = data[0]
            eng.clear_states()
            o = None
            for i in range(t):
                o = eng.stream_frame(lrs[0, i], fvs[0, i], mk8[0, i])
            return [o]
        if n_flight == 1:
            return [(e or eng).forward(*d) for d in data]
        outs = [None] * ncalls
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for c, d in enumerate(data):
            with torch.cuda.stream(streams[c % n_flight]):
                outs[c] = engs[c % n_flight].forward(*d)
        for s in streams:
            cur.wait_stream(s)
        return outs

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            outs = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            outs = step()
        barrier()
        elapsed = time.perf_counter() - t0
    elapsed = benchutil.reduce_elapsed(elapsed, dist, dev)
    frames_per_step = t * clips
    assert not any(e.overflowed(stream=(mode == "stream")) for e in engs), "numerics guard fired on the benchmark clip"

    # PSNR reduction (the only collective of the path): raw squared-error sums of the last SR frame(s) against the synthetic
    # HR scene inside the fovea window (not a quality figure without trained weights: it exercises the reduction)
    last = outs[-1] if mode == "stream" else outs[-1][0, -1]
    acc = ops.sq_err_sums(last[None].contiguous(), data[-1][1][0, -1][None].contiguous()).clone()
    vec = benchutil.reduce_sums(torch.cat([acc, torch.tensor([float(frames_per_step)], dtype=torch.float64, device=dev)]), dist)
    agg = benchutil.aggregate(world, args.steps, frames_per_step, elapsed)

    arithmetic = ("fp32 in / fp32 accumulate / fp32 out; products on the fp16 MFMA via an exact 2-term split (3 MFMAs per MAC, "
                  "error at the fp32 summation-order floor, see `parity`)") if storage == "f32" else (
                  "bf16 activations + recurrent state in HBM, bf16 conv / DCN weights, one bf16 MFMA per MAC, fp32 accumulate; fp32 "
                  "flow / offsets / masks / API tensors (include/crfp_hip.h, 'bf16 storage')")
    result = benchutil.contract_line(agg, world, args.steps, args.warmup, storage, {
        "workload": cfg["name"] + (" [with command-line overrides]" if custom else ""), "baseline_config_index": cfg["index"],
        "conv_arithmetic": arithmetic, "mode": mode, "frames_per_clip": t, "lr": [h, w], "sr": [8 * h, 8 * w],
        "fv_size": fv, "sigma_t": cfg["sigma"], "clips_per_gpu_per_step": clips, "clips_per_call": bclips,
        "calls_in_flight_per_gpu": n_flight, "clips_in_flight_per_gpu": n_flight * bclips,
        "batching": ("crfp_dsv_forward_batch: the clips of a call in lock-step, one launch per layer over all of them; per clip "
                     "bit-identical to one-clip calls (tests/test_gpu_round4.py)") if bclips > 1 else "one clip per call",
        "storage": storage, "parallelism": f"clip-sharded x{world}",
        **({"inputs_resident": bool(eng.inputs_resident)} if mode == "stream" else {})}, dist)
    result["frames_per_step_per_gpu"] = frames_per_step

    # everything below adds objects to the line; the code of each leg lives in benchlegs.py
    import types
    c = types.SimpleNamespace(args=args, cfg=cfg, custom=custom, dev=dev, rank=rank, world=world, mode=mode, storage=storage, t=t, h=h, w=w, fv=fv,
                              clips=clips, bclips=bclips, n_flight=n_flight, data=data, data1=data1, data_np=data_np, mk8=mk8 if mode == "stream" else None,
                              model=model, eng=eng, sdt=sdt, agg=agg, frames_per_step=frames_per_step, step=step, result=result, fam=None, extras=False)
    if rank == 0 and not args.no_kernel_profile:
        benchlegs.profile_objects(c)

    extras = c.extras = rank == 0 and world == 1 and not args.no_extras
    if extras and mode == "clip":
        benchlegs.leg_side_stream(c)
    if rank == 0 and world == 1 and mode == "stream" and c.eng.inputs_resident:
        benchlegs.leg_stream_without_resident(c)
    if extras and mode == "clip" and storage == "f32":
        benchlegs.leg_strict_f32(c)

    if extras and mode == "clip" and args.config == 2 and not custom:
        benchlegs.leg_spec_weights(c)

    if extras and mode == "clip":
        benchlegs.leg_multi_stream(c)

    if extras and mode == "clip":
        benchlegs.leg_lockstep_batch(c)

    if extras and mode == "clip" and args.config == 2:
        benchlegs.leg_cra_engine(c)

    if extras and mode == "clip" and args.config == 2:
        benchlegs.leg_ablation_engines(c)

    if extras and storage == "f32":
        benchlegs.leg_per_op(c)

    if extras and storage == "f32" and args.config == 2 and not custom:
        benchlegs.leg_runtime_rig(c)

    if extras and args.config == 2 and not custom and not args.no_other_configs:
        benchlegs.leg_other_configs(c)

    if extras and mode == "clip" and args.config == 2 and not custom and not args.no_other_configs:
        benchlegs.leg_mask_gate(c)

    if extras and mode == "clip" and args.config == 2 and not custom and not args.no_other_configs and "warp_dcn" in result:
        benchlegs.leg_dcn_g8_alone(c)

    if rank == 0 and not args.no_cpu_baseline:
        benchlegs.leg_cpu_baseline(c)

    if rank == 0:
        for key, src, field in (("strict_f32_frames_per_sec", "strict_f32", "frames_per_sec"), ("warp_dcn_8d_frac", "warp_dcn_8d", "frac"),
                                ("warp_dcn_frac", "warp_dcn", "frac"), ("dcn_fused_avg_us", "dcn_fused", "avg_us"),
                                ("cra_engine_frames_per_sec", "cra_engine", "frames_per_sec"),
                                ("dense_launch_frames_per_sec", "mask_gate", "dense_frames_per_sec"),
                                ("dcn_g8_alone_frac", "dcn_g8_alone", "frac")):
            if src in result and field in result[src]:
                result[key] = result[src][field]
        result["psnr_reduce"] = benchutil.psnr_reduce_record(vec, world)
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
