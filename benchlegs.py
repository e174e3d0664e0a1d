"""The objects bench.py adds to its JSON line beside the contract fields: the instrumented kernel pass (`profile_objects`) and the extra legs
(`leg_*`), each taking the bench context `c` (a SimpleNamespace of bench.main's state) and writing into `c.result`.  Moved out of bench.py in
round 5 without change of behaviour; bench.py keeps the contract: arguments, the timed region, the reductions, the one line.
Lives beside bench.py, NOT inside the crfp_amd package: the cpu_baseline leg runs the oracle, which nothing under crfp_amd/ may import."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)
HBM_COPY_CEILING_GBS = 6290.0
F32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak
MFMA16_PEAK_TFLOPS = 2500.0   # dense bf16 / fp16 MFMA peak
# fp32 storage: the wide convs run 3 fp16 MFMA products per algorithmic fp32 MAC (split-fp16, fp32-grade result,
# DESIGN.md 3.1), so the algorithmic-flop ceiling is the 16-bit peak / 3; bf16 storage: one bf16 product per MAC.
SPLIT_F16_EQUIV_PEAK_TFLOPS = MFMA16_PEAK_TFLOPS / 3.0

CONFIGS = {
    2: dict(index=1, storage="f32", mode="clip", t=7, h=180, w=320, fv=96, sigma=10.0, clips=1,
            name="BASELINE configs[1]: single MI355X, 7-frame 180x320 -> 1440x2560 x8 SR, batch=1, fp32, sigma_T=10"),
    3: dict(index=2, storage="bf16", mode="stream", t=100, h=180, w=320, fv=96, sigma=50.0, clips=1,
            name="BASELINE configs[2]: single MI355X, 100-frame 180x320 streaming recurrent inference (one frame per call), bf16, sigma_T=50"),
    4: dict(index=3, storage="bf16", mode="clip", t=7, h=180, w=320, fv=96, sigma=10.0, clips=4, flight=1, batch=4,
            name="BASELINE configs[3]: 32 independent 7-frame 180x320 clips sharded over 8 GPUs (4 clips per GPU per step), bf16"),
    5: dict(index=4, storage="bf16", mode="clip", t=7, h=270, w=480, fv=144, sigma=10.0, clips=1,
            name="BASELINE configs[4]: single MI355X, 270x480 -> 2160x3840 (4K) x8 SR, 7 frames, bf16"),
}


def kernel_family(name: str) -> str:
    if name.startswith("conv_mfma"):
        return "conv3x3_mfma"
    if name.startswith("conv_narrow"):
        return "conv3x3_narrow"
    return name


# hipEvent launch-site family -> rocprofv3 kernel names (for the PMC traffic lookup)
ROCPROF_NAMES = {"conv3x3_mfma": ("conv3x3_split_kernel", "conv3x3_split8_kernel", "conv3x3_pair_kernel", "conv3x3_mfma_kernel",
                                  "conv3x3_bf16_kernel", "conv3x3_bf16x8_kernel", "conv3x3_q16_kernel"),
                 "conv3x3_narrow": ("conv3x3_narrow_kernel", "conv3x3_narrow_pair_kernel", "conv3x3_narrow_seq_kernel", "conv3x3_narrow_chain_kernel"),
                 "dcnv2_g8_c32": ("dcn_g8_kernel", "dcn_g8_pipe_kernel"), "dcnv2_shared_c4": ("dcn3_kernel<false",), "dcnv2_shared_c4_fused": ("dcn3_kernel<true",),
                 "flow_warp_q4_c4": ("flow_warp_p4_kernel",), "flow_warp_q4_c32": ("flow_warp_p4_kernel",),
                 "flow_warp_q4_c24": ("flow_warp_p4_kernel",), "flow_warp_q4_c32+c24": ("flow_warp_p4_dual_kernel", "flow_warp_p4_dual_split_kernel"),
                 "hr_prep_up8_blend": ("hr_prep_kernel",), "offset_mask_conv+dcnv2_g8_fused": ("dcn_fused_kernel",)}


def pmc_traffic(family: str, storage: str, lr=(180, 320), clips_per_call: int = 1):
    """{"bytes_per_launch", "source", "measured_in_run": False} from the committed rocprofv3 PMC summary (2 x FETCH_SIZE +
    WRITE_SIZE per MI355X_MICROARCH.md, separate --pmc passes, tools/collect_profiles.sh) of the same workload; None when
    no summary exists for this storage mode.  It is a constant as far as this run is concerned -- hence the tag."""
    # one committed summary per launch shape: fp32 / bf16 one-clip calls, and BASELINE config 4's lock-step call of 4 bf16 clips (its
    # launches carry 4 clips each, so bytes per launch are those of 4 clips)
    fname = ("pmc_summary_latest_c4.json" if (storage == "bf16" and clips_per_call == 4) else
             "pmc_summary_latest.json" if storage == "f32" else "pmc_summary_latest_bf16.json")
    if clips_per_call not in (1, 4) or (clips_per_call == 4 and storage != "bf16"):
        return None
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path) or family not in ROCPROF_NAMES or tuple(lr) != (180, 320):
        return None   # the committed counter passes ran the 180x320 geometry; bytes per launch do not transfer to another one
    tot = calls = 0.0
    for r in json.load(open(path)):
        kname = r["kernel"].split("::")[-1]   # crfp:: / crfp_bf16:: prefixes
        if any(kname.startswith(n) for n in ROCPROF_NAMES[family]) and r.get("hbm_MB_per_launch_corrected") is not None:
            tot += r["hbm_MB_per_launch_corrected"] * 1e6 * r["calls"]
            calls += r["calls"]
    if not calls:
        return None
    # the summary names the kernel sources it was collected with (tools/summarize_pmc.py): another digest = other kernels than the ones timed here
    from crfp_amd import _lib
    meta = os.path.join(ROOT, "profiles", fname.replace(".json", ".meta.json"))
    sha = json.load(open(meta)).get("kernels_src_sha") if os.path.exists(meta) else None
    return {"bytes_per_launch": tot / calls, "source": "profiles/" + fname, "measured_in_run": False,
            "kernels_src_sha": sha, "stale": None if sha is None else sha != _lib.kernel_source_digest()}


def warp_dcn_8d(fam, steady_frames, storage, h, w):
    """SURVEY 8(d) as written, NOT re-scoped: API-tensor bytes of the three flow_warps and the four DCNv2 calls of a steady-state
    frame (fp32 @A: 254.4 + 1290.4 = 1544.7 MB) over the time of ALL kernels that do that work in such a frame -- including the
    fused offset-head + dcn_g8 kernel (whose time also contains the 32 -> 216 head conv) and the dcn_3 kernel with its offset conv
    inside.  Comparable across rounds whatever gets fused.  fam: {launch-site family: {"ms": total}} of `steady_frames` frames."""
    sb = 4 if storage == "f32" else 2
    px2, px8 = (2 * h) * (2 * w), (8 * h) * (8 * w)
    api_bytes = ((66 + 50) * px2 + 10 * px8) * sb + (3 * (280 * px2 + 32 * 32 * 9) + 35 * px8) * sb
    names8d = [n for n in fam if n.startswith("flow_warp") or n.startswith("dcnv2") or n == "offset_mask_conv+dcnv2_g8_fused"]
    t8d = sum(fam[n]["ms"] for n in names8d) * 1e-3 / steady_frames
    return {"bound": "hbm", "api_tensor_bytes_per_steady_frame": api_bytes, "kernel_us_per_steady_frame": 1e6 * t8d,
            "achieved": api_bytes / t8d / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": api_bytes / t8d / 1e9 / HBM_PEAK_GBS, "kernels": sorted(names8d),
            "per_kernel_avg_us": {n: 1e3 * fam[n]["ms"] / fam[n]["launches"] for n in sorted(names8d)},
            "note": "SURVEY 8(d): (2C+2)HWs per flow_warp, (Cin + 2 dg K + dg K + Cout)HWs + weights per DCNv2 with the "
                    "offset / mask tensors as the reference's API passes them (144 + 72 channels, dcn_3: 18 + 9), divided "
                    "by the time of every warp / DCN kernel of a steady-state frame; the fused kernels' time includes "
                    "the offset / mask head convs they absorbed"}


def time_op(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / iters


def profile_objects(c):
    """kernels / roofline / dcn_fused / warp_dcn / warp_dcn_8d from an instrumented single-stream pass (crfp_prof_*); sets c.fam"""
    from crfp_amd import _lib
    args, mode, storage, t, h, w, clips, bclips, eng, step, result, fam = c.args, c.mode, c.storage, c.t, c.h, c.w, c.clips, c.bclips, c.eng, c.step, c.result, c.fam
    L = _lib.lib()
    L.crfp_prof_reset()
    L.crfp_prof_enable(1)
    psteps = min(args.steps, 5 if mode == "clip" else 1)
    with torch.no_grad():
        for _ in range(psteps):
            step(eng) if mode == "clip" else step()
    torch.cuda.synchronize()
    recs = _lib.prof_report(512)
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
        table.append({"kernel": name, "launches_per_step": f["launches"] / psteps, "ms_per_step": f["ms"] / psteps,
                      "avg_us": 1e3 * f["ms"] / f["launches"], "share": f["ms"] / total_ms,
                      "GBps": f["bytes"] / s / 1e9 if s > 0 else 0.0, "TFLOPs": f["flops"] / s / 1e12 if s > 0 else 0.0})
    result["kernels"] = table
    result["kernel_ms_per_step"] = total_ms / psteps
    result["kernel_ms_note"] = ("sum of hipEvent-bracketed kernel durations of the instrumented pass, which runs the SINGLE-stream "
                                "schedule (events bracket launches per stream); the timed region uses the two-stream schedule, "
                                "so ms_per_step can be smaller than this sum")
    dom = table[0]
    domf = fam[dom["kernel"]]
    if dom["kernel"] == "conv3x3_mfma":
        strict_env = os.environ.get("CRFP_PRECISION") == "f32"
        if storage == "bf16":
            peak, scheme = MFMA16_PEAK_TFLOPS, "bf16"
            note = "one v_mfma_f32_32x32x16_bf16 per MAC on bf16 operands, fp32 accumulate: peak = 2.5 PF dense bf16"
        elif strict_env:
            peak, scheme, note = F32_MFMA_PEAK_TFLOPS, "f32", "fp32 MFMA (CRFP_PRECISION=f32)"
        else:
            peak, scheme = SPLIT_F16_EQUIV_PEAK_TFLOPS, "f16x3"
            note = ("algorithmic fp32 flops; executed as 3 fp16 MFMA products per MAC (split-fp16, fp32-grade), so peak = 2.5 PF "
                    "dense fp16 / 3; see profiles/*_mfma_lds_util.txt for the MFMA / LDS pipe counters")
        # The right roof per layer: a launch can be no shorter than max(flops / MFMA peak, bytes / HBM peak).  Most
        # layers of this family (32 -> 32 and 64 -> 32 at 2x resolution: 72-96 FLOP/B against a balance point of 104 at 833 TF / 8 TB/s)
        # are bounded by their BYTES, so pricing the family against the MFMA peak alone flattered the roof and mislabelled the bound.
        t_mfma = t_hbm = t_roof = 0.0
        for r in recs:
            if kernel_family(r["name"]) != "conv3x3_mfma":
                continue
            tm, th = r["flops"] / (peak * 1e12), r["bytes"] / (HBM_PEAK_GBS * 1e9)
            t_mfma += tm; t_hbm += th; t_roof += max(tm, th)
        t_act = domf["ms"] * 1e-3
        bound = "hbm" if t_hbm >= t_mfma else "mfma"
        result["roofline"] = {"kernel": dom["kernel"], "bound": bound,
                              "achieved": dom["GBps"] if bound == "hbm" else dom["TFLOPs"],
                              "peak": HBM_PEAK_GBS if bound == "hbm" else peak, "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                              "frac": (dom["GBps"] / HBM_PEAK_GBS) if bound == "hbm" else dom["TFLOPs"] / peak,
                              "frac_of_per_layer_roof": t_roof / t_act,
                              "frac_mfma": dom["TFLOPs"] / peak, "frac_hbm": dom["GBps"] / HBM_PEAK_GBS,
                              "achieved_TFLOPs": dom["TFLOPs"], "mfma_peak_TFLOPs": peak, "achieved_GBps": dom["GBps"],
                              "traffic": pmc_traffic(dom["kernel"], storage, (h, w), bclips), "avg_launch_us": dom["avg_us"],
                              "algorithmic_flops_per_launch": domf["flops"] / domf["launches"],
                              "algorithmic_bytes_per_launch": domf["bytes"] / domf["launches"],
                              "frac_of_fp32_mfma_peak": dom["TFLOPs"] / F32_MFMA_PEAK_TFLOPS,
                              "conv_scheme": scheme,
                              "note": note + "; `bound` = the roof that binds the family's launches in sum (sum of flops / MFMA peak against "
                                      "sum of bytes / 8 TB/s), `frac` is against that roof, `frac_of_per_layer_roof` = sum over launch sites of "
                                      "max(MFMA time, HBM time) / measured time"}
    else:
        result["roofline"] = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": dom["GBps"] / HBM_PEAK_GBS, "traffic": pmc_traffic(dom["kernel"], storage, (h, w), bclips),
                              "avg_launch_us": dom["avg_us"],
                              "algorithmic_bytes_per_launch": domf["bytes"] / domf["launches"]}
    fz = fam.get("offset_mask_conv+dcnv2_g8_fused")
    if fz:
        # the 32 -> 216 offset / mask head and dcn_g8 as ONE kernel (gather.hip dcn_fused_kernel): the offsets and masks
        # (216 fp32 channels written and read back by the two-kernel path) never reach HBM
        fs = fz["ms"] * 1e-3
        peak = MFMA16_PEAK_TFLOPS if storage == "bf16" else SPLIT_F16_EQUIV_PEAK_TFLOPS
        px2 = (2 * h) * (2 * w)
        result["dcn_fused"] = {"kernel": "offset_mask_conv+dcnv2_g8_fused", "bound": "mfma", "launches_per_step": fz["launches"] / psteps,
                               "avg_us": 1e3 * fz["ms"] / fz["launches"], "achieved": fz["flops"] / fs / 1e12, "peak": peak,
                               "unit": "TFLOP/s", "frac": fz["flops"] / fs / 1e12 / peak,
                               "algorithmic_GBps": fz["bytes"] / fs / 1e9,
                               "traffic": pmc_traffic("offset_mask_conv+dcnv2_g8_fused", storage, (h, w), bclips),
                               "hbm_bytes_not_moved_per_launch": 2.0 * px2 * 216 * 4,
                               "note": "offset / mask head + dcn_g8 in one launch, bit-identical to the two-kernel path "
                                       "(CRFP_DCN_FUSED=0 restores it: conv_mfma:dcn.offset_mask + dcnv2_g8_c32); flops = conv + DCN GEMM + "
                                       "bilinear; MFMA and sampling VALU share each SIMD's issue port (an MFMA gap hides up to ~24 issue "
                                       "cycles of VALU, MI355X_MICROARCH.md; re-measured in profiles/r03_mfma_valu_overlap_microtest.txt)"}
    gat = {n: f for n, f in fam.items() if n.startswith("flow_warp") or n.startswith("dcnv2")}
    if gat:
        gb = sum(f["bytes"] for f in gat.values()); gs = sum(f["ms"] for f in gat.values()) * 1e-3
        # dcn_3 through the API moves 18 offset + 9 mask channels (the reference tiles 2+1 channels 9x); the kernel
        # reads the compact 2+1: SURVEY 8(d) asks for both figures
        d3 = fam.get("dcnv2_shared_c4") or fam.get("dcnv2_shared_c4_fused")
        gb_api = gb + (d3["launches"] * (8 * h) * (8 * w) * 24 * 4.0 if d3 else 0.0)
        result["warp_dcn"] = {"bound": "hbm", "achieved": gb / gs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": gb / gs / 1e9 / HBM_PEAK_GBS, "frac_of_copy_ceiling": gb / gs / 1e9 / HBM_COPY_CEILING_GBS,
                              "achieved_api_tensor_bytes": gb_api / gs / 1e9, "frac_api_tensor_bytes": gb_api / gs / 1e9 / HBM_PEAK_GBS,
                              "ms_per_step": 1e3 * gs / psteps,
                              "per_kernel": {n: {"avg_us": 1e3 * f["ms"] / f["launches"], "GBps": f["bytes"] / (f["ms"] * 1e-3) / 1e9,
                                                 "frac": f["bytes"] / (f["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 "traffic": pmc_traffic(n, storage, (h, w), bclips)} for n, f in gat.items()},
                              "note": "dcn_3 priced at its compact 2+1 offset/mask channels, not the 9x-replicated API tensors; "
                                      "bf16 storage: feature bytes halve, offsets / masks / flow stay fp32"
                                      + ("; dcn_g8 (dcn_0/1/2) runs inside the fused kernel reported under dcn_fused and is not "
                                         "part of this figure" if fz else "")}

    if gat:
        steady = max(1, (t - 1) * clips * psteps) if mode == "clip" else max(1, (t - 1) * psteps)
        result["warp_dcn_8d"] = warp_dcn_8d(fam, steady, storage, h, w)
    c.fam = fam


def leg_side_stream(c):
    """what the two-stream schedule hides"""
    args, n_flight, data, data1, eng, agg, result = c.args, c.n_flight, c.data, c.data1, c.eng, c.agg, c.result
    # what the two-stream schedule hides: the same steps with CRFP_DSV_SINGLE_STREAM (everything on the caller's stream)
    eng.single_stream = True
    with torch.no_grad():
        eng.forward(*data1[0])
        torch.cuda.synchronize()
        n_s = max(2, min(args.steps, 5))
        t0 = time.perf_counter()
        for _ in range(n_s):
            for d_ in data:
                eng.forward(*d_)
        torch.cuda.synchronize()
        single_ms = 1e3 * (time.perf_counter() - t0) / n_s
    eng.single_stream = False
    if n_flight == 1:
        result["side_stream"] = {"single_stream_ms_per_step": single_ms, "two_stream_ms_per_step": agg["ms_per_step"],
                                 "side_stream_hidden_ms": single_ms - agg["ms_per_step"],
                                 "note": "FNet + fovea blend + encoder_hr + upsample conv + flow up-sampling run on the library's side "
                                         "stream beside the recurrent chain; bit-identical results"}


def leg_stream_without_resident(c):
    """stream mode without CRFP_DSV_INPUTS_RESIDENT"""
    model, eng, frames_per_step, step, result = c.model, c.eng, c.frames_per_step, c.step, c.result
    # the same calls without CRFP_DSV_INPUTS_RESIDENT (every call waits for the previous frame before its flow network starts)
    model.inputs_resident = eng.inputs_resident = False
    with torch.no_grad():
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        plain_ms = 1e3 * (time.perf_counter() - t0) / 2
    model.inputs_resident = eng.inputs_resident = True
    eng.clear_states()
    result["stream_without_resident_flag"] = {"ms_per_step": plain_ms, "frames_per_sec": 1e3 * frames_per_step / plain_ms,
                                              "note": "same bits; the flag is a promise about the INPUT tensors (complete before the call), see include/crfp_hip.h"}


def leg_strict_f32(c):
    """the same clip in strict fp32 (CRFP_DSV_STRICT_F32)"""
    args, t, data1, model, eng, result = c.args, c.t, c.data1, c.model, c.eng, c.result
    # the same clip in strict fp32 (plain fp32 MFMA for every conv and the DCN GEMM): what the split-fp16 scheme buys
    model.precision = "f32"
    se = model.engine()
    with torch.no_grad():
        ref_strict = se.forward(*data1[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_s = max(2, min(args.steps, 5))
        for _ in range(n_s):
            se.forward(*data1[0])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    model.precision = "split"
    eng = model.engine()
    with torch.no_grad():
        fast = eng.forward(*data1[0])
    result["strict_f32"] = {"frames_per_sec": n_s * t / dt, "ms_per_clip": 1e3 * dt / n_s,
                            "max_abs_diff_split_vs_strict": float((fast - ref_strict).abs().max()),
                            "note": "CRFP_DSV_STRICT_F32: v_mfma_f32_32x32x2_f32 everywhere, no fp16 operands, no range guard"}
    c.eng = eng


def leg_spec_weights(c):
    """SURVEY 8(d)'s own N(0, 0.02) DCN heads: warp_dcn_8d, frames/s, parity"""
    from crfp_amd.engine import DSVEngine
    from crfp_amd import synth
    from crfp_amd import _lib
    args, cfg, dev, storage, t, h, w, fv, data1, result = c.args, c.cfg, c.dev, c.storage, c.t, c.h, c.w, c.fv, c.data1, c.result
    # SURVEY 8(d)'s own weight regime: N(0, 0.02) on the dcn_offset / dcn_mask convs (residual offsets of a few pixels, as a trained
    # network has them) instead of the "stress" heads of the headline (residuals filling the whole +-10 px of 10 tanh).  Same clip,
    # same random stream for every other weight; its own timed loop, instrumented pass and parity leg against the oracle.
    import subprocess
    import tempfile
    sd_spec = synth.make_state_dict(7, offset_std=0.02)
    es = DSVEngine({k: torch.from_numpy(v) for k, v in sd_spec.items()}, dev, storage=storage)
    L = _lib.lib()
    with torch.no_grad():
        got_spec = es.forward(*data1[0]).clone()
        torch.cuda.synchronize()
        n_s = max(2, min(args.steps, 5))
        t0 = time.perf_counter()
        for _ in range(n_s):
            es.forward(*data1[0])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        L.crfp_prof_reset(); L.crfp_prof_enable(1)
        for _ in range(3):
            es.forward(*data1[0])
        torch.cuda.synchronize()
        recs_s = _lib.prof_report(512)
        L.crfp_prof_enable(0); L.crfp_prof_reset()
    fam_s = {}
    for r in recs_s:
        f = fam_s.setdefault(kernel_family(r["name"]), dict(launches=0, ms=0.0))
        f["launches"] += r["launches"]; f["ms"] += r["total_ms"]
    leg = warp_dcn_8d(fam_s, 3 * (t - 1), storage, h, w)
    leg.update({"weights": "synth.make_state_dict(7, offset_std=0.02): SURVEY 8(d)'s N(0, 0.02) dcn_offset / dcn_mask heads",
                "frames_per_sec": n_s * t / dt, "ms_per_clip": 1e3 * dt / n_s, "overflowed": bool(es.overflowed())})
    if not args.no_cpu_baseline:
        tmp = os.path.join(tempfile.mkdtemp(), "oracle_spec.npz")
        cmd = [sys.executable, "-m", "oracle.run_sample", "--frames", "3", "--h", str(h), "--w", str(w), "--fv-size", str(fv),
               "--sigma-t", str(cfg["sigma"]), "--clip-seed", "1234", "--clip-frames", str(t), "--storage", storage,
               "--offset-std", "0.02", "--out", tmp]
        try:
            subprocess.run(cmd, cwd=ROOT, timeout=240, check=True)
            ref_s = torch.from_numpy(np.load(tmp)["out"])
            leg["parity"] = {"max_abs_diff_vs_oracle": float((got_spec[:, :3].cpu() - ref_s).abs().max()), "frames": 3, "tolerance": 1e-3}
        except (subprocess.TimeoutExpired, subprocess.CalledProcessError) as e:
            leg["parity"] = {"error": type(e).__name__}
    result["warp_dcn_8d_spec_weights"] = leg
    del es


def leg_multi_stream(c):
    """2 / 4 one-clip calls in flight on separate HIP streams"""
    from crfp_amd.engine import DSVEngine
    args, dev, storage, t, clips, data1, eng, sdt, result = c.args, c.dev, c.storage, c.t, c.clips, c.data1, c.eng, c.sdt, c.result
    # independent clips in flight on separate HIP streams of the same GPU fill each other's tails (config 4 runs 4 per GPU)
    ms = {}
    exact = True
    with torch.no_grad():
        ref_out = eng.forward(*data1[0]).clone()
    for C in (2, 4):
        es = [DSVEngine(sdt, dev, storage=storage) for _ in range(C)]
        sts = [torch.cuda.Stream(device=dev) for _ in range(C)]
        with torch.no_grad():
            for it in range(1 + min(args.steps, 5)):
                if it == 1:
                    torch.cuda.synchronize()
                    tm = time.perf_counter()
                os_ = []
                for e, st in zip(es, sts):
                    with torch.cuda.stream(st):
                        os_.append(e.forward(*data1[0]))
            torch.cuda.synchronize()
        ms[str(C)] = C * min(args.steps, 5) * t / (time.perf_counter() - tm)
        # concurrent kernels must not disturb each other: every in-flight clip == the sequential result, bit for bit
        exact = exact and all(bool(torch.equal(o, ref_out)) for o in os_)
        del es, os_
    ms["bit_exact_vs_sequential"] = exact
    result["multi_stream_frames_per_sec"] = ms


def leg_lockstep_batch(c):
    """n clips per crfp_dsv_forward_batch call"""
    from crfp_amd import synth
    from crfp_amd import benchutil
    args, cfg, dev, rank, t, h, w, fv, clips, eng, step, result = c.args, c.cfg, c.dev, c.rank, c.t, c.h, c.w, c.fv, c.clips, c.eng, c.step, c.result
    # the batch axis inside the library (crfp_dsv_forward_batch): n clips per call in lock-step, one launch per layer over all of them
    lb = {}
    exact_lb = True
    with torch.no_grad():
        for nclips in (2, 4):
            seeds = benchutil.rank_clip_seeds(rank, nclips, base=4321)
            cl = [synth.make_clip(sd_, 1, t, h, w, fv_size=fv, sigma_t=cfg["sigma"]) for sd_ in seeds]
            stack = tuple(torch.from_numpy(np.concatenate([c[k] for c in cl], 0)).to(dev) for k in range(3))
            eng.batch_mode = "loop"
            ref_b = eng.forward(*stack).clone()
            eng.batch_mode = "lockstep"
            got_b = eng.forward(*stack)
            exact_lb = exact_lb and bool(torch.equal(got_b, ref_b))
            torch.cuda.synchronize()
            n_s = max(2, min(args.steps, 4))
            t0 = time.perf_counter()
            for _ in range(n_s):
                eng.forward(*stack)
            torch.cuda.synchronize()
            lb[str(nclips)] = nclips * t * n_s / (time.perf_counter() - t0)
            del stack, ref_b, got_b
    lb["bit_exact_vs_one_clip_calls"] = exact_lb
    lb["note"] = "frames/s with n clips per crfp_dsv_forward_batch call (the reference's own [n, t, ...] batch axis); one clip per call is the headline"
    result["lockstep_batch_frames_per_sec"] = lb
    torch.cuda.empty_cache()


def leg_cra_engine(c):
    """CRFP_DSV_CRA on its one-call schedule"""
    from crfp_amd import synth
    from crfp_amd.model import CRFP
    args, dev, storage, t, data1, model, result = c.args, c.dev, c.storage, c.t, c.data1, c.model, c.result
    # the reference's other shipped-size wiring, CRFP_DSV_CRA (model/CRFP.py:2314; eval.sh's `_cra` run), on its own one-call schedule
    # (crfp_cra_forward_batch): frames/s on this config's clip, checked against the per-operator composition of the same model
    with torch.no_grad():
        cra = CRFP.CRFP_DSV_CRA(device=dev, mid_channels=32)
        cra_sd = synth.make_state_dict_like({k: tuple(v.shape) for k, v in cra.state_dict().items()}, 1)
        cra.load_state_dict({k: torch.from_numpy(v) for k, v in cra_sd.items()})
        cra = cra.to(dev).eval()
        cra.storage = storage
        d0 = data1[0]
        got_c = cra(*d0)
        ref_c = cra.forward_composed(*d0)
        torch.cuda.synchronize()
        n_s = max(2, min(args.steps, 5))
        t0 = time.perf_counter()
        for _ in range(n_s):
            cra(*d0)
        torch.cuda.synchronize()
        result["cra_engine"] = {"frames_per_sec": t * n_s / (time.perf_counter() - t0), "entry_point": "crfp_cra_forward_batch",
                                "max_abs_diff_vs_per_operator_composition": float((got_c - ref_c).abs().max()), "tolerance": 2e-4,
                                "note": "CRFP_DSV_CRA(mid_channels=32) on this config's clip, one clip per call"}
        assert result["cra_engine"]["max_abs_diff_vs_per_operator_composition"] < 2e-4, result["cra_engine"]
        del cra, got_c, ref_c
    torch.cuda.empty_cache()


def leg_ablation_engines(c):
    """CRFP_simple / CRFP (mid_channels 32) and the constructor-default mid_channels = 16 on their one-call schedules"""
    from crfp_amd import synth
    from crfp_amd.model import CRFP
    args, dev, storage, t, data1, result = c.args, c.dev, c.storage, c.t, c.data1, c.result
    # round 6: the reference's ablation wirings (model/CRFP.py:816-1385) and its constructor default width (:1388) no longer run as per-operator
    # compositions: frames/s on this config's clip through crfp_simple_forward_batch / crfp_dense_forward_batch / the embedded 16-channel table,
    # each checked against the per-operator composition of the same model
    out = {}
    with torch.no_grad():
        for name, cls, mid, entry in (("CRFP_simple", "CRFP_simple", 32, "crfp_simple_forward_batch"), ("CRFP", "CRFP", 32, "crfp_dense_forward_batch"),
                                      ("CRFP_DSV_mid16", "CRFP_DSV", 16, "crfp_dsv_forward_batch (embed_mid32)")):
            m = getattr(CRFP, cls)(device=dev, mid_channels=mid)
            sd = synth.make_state_dict_like({k: tuple(v.shape) for k, v in m.state_dict().items()}, 1)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            m = m.to(dev).eval()
            m.storage = storage
            d0 = data1[0]
            got = m(*d0)
            ref = m.forward_composed(*d0)
            torch.cuda.synchronize()
            n_s = max(2, min(args.steps, 5))
            t0 = time.perf_counter()
            for _ in range(n_s):
                m(*d0)
            torch.cuda.synchronize()
            out[name] = {"frames_per_sec": t * n_s / (time.perf_counter() - t0), "entry_point": entry, "mid_channels": mid,
                         "max_abs_diff_vs_per_operator_composition": float((got - ref).abs().max())}
            assert out[name]["max_abs_diff_vs_per_operator_composition"] < 2e-4, out
            del m, got, ref
            torch.cuda.empty_cache()
    out["tolerance"] = 2e-4
    out["note"] = "one clip per call on this config's clip; before round 6 these models ran as per-operator compositions (136-165 frames/s)"
    result["ablation_engines"] = out


def leg_per_op(c):
    """per-operator C-ABI entry points"""
    from crfp_amd.model import CRFP
    from crfp_amd import ops
    dev, t, h, w, data1, eng, result = c.dev, c.t, c.h, c.w, c.data1, c.eng, c.result
    # the per-operator C-ABI entry points (NCHW API tensors in and out: each call includes its layout conversions)
    H2, W2, H8, W8 = 2 * h, 2 * w, 8 * h, 8 * w
    g = torch.Generator(device="cpu").manual_seed(3)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x32, fl2 = rn(1, 32, H2, W2), rn(1, H2, W2, 2) * 3
    x4, fl8 = rn(1, 4, H8, W8), rn(1, H8, W8, 2) * 8
    off, msk = rn(1, 144, H2, W2) * 4, torch.sigmoid(rn(1, 72, H2, W2))
    wd, bd = rn(32, 32, 3, 3) * 0.1, rn(32)
    wc, bc = rn(32, 64, 3, 3) * 0.05, rn(32)
    x64 = rn(1, 64, H2, W2)
    spy = CRFP.SPyNet(pretrained=None, device=dev).to(dev).eval()
    spy_a, spy_b = torch.rand(1, 3, 192, 320, generator=g).to(dev), torch.rand(1, 3, 192, 320, generator=g).to(dev)
    with torch.no_grad():
        result["per_op_us"] = {
            "flow_warp_c32@2x": time_op(lambda: ops.flow_warp(x32, fl2)),
            "flow_warp_c4@8x": time_op(lambda: ops.flow_warp(x4, fl8)),
            "dcnv2_c32_dg8@2x": time_op(lambda: ops.dcnv2(x32, off, msk, wd, bd, 3, 1, 1, 8)),
            "conv3x3_64to32@2x": time_op(lambda: ops.conv3x3(x64, wc, bc, "lrelu")),
            "upsample_bilinear_x8_c3": time_op(lambda: ops.upsample_bilinear(data1[0][0][0, :1], scale_factor=8)),
            "fnet_6pairs@lr": time_op(lambda: eng.compute_flow(data1[0][0][0, 1:7], data1[0][0][0, 0:6]), 10) if t >= 7 else None,
            "spynet_1pair@192x320": time_op(lambda: spy(spy_a, spy_b), 5),
            "note": "wall-clock per call incl. the NCHW <-> Q4 conversions the operator boundary needs (the engine pays none of them)"}


def leg_runtime_rig(c):
    """the reference's test_runtime.py measurement"""
    t, result = c.t, c.result
    # the reference's stand-alone speed test (test_runtime.py:81-99,142-186) on the regional wiring: 1 x 5 frames, 135 x 240 -> 1080p,
    # 96 x 96 fovea crop, 720 x 720 warp window; one crfp_rt_forward_clip per clip (csrc/engine_rt.hip)
    from crfp_amd import runtime_rig
    _, spf = runtime_rig.run(repeat_time=30, warm_up=10)
    _, spf_dsv = runtime_rig.run(repeat_time=12, warm_up=4, variant="dsv")
    result["runtime_rig"] = {"regional_ms_per_1080p_frame": round(1e3 * spf, 4), "dsv_whole_frame_ms_per_1080p_frame": round(1e3 * spf_dsv, 4),
                             "note": "seconds / (repeat - warm_up + 1) / t as test_runtime.py:186 prints it; regional = MRCF_runtime.MRCF_simple_v18 "
                                     "through crfp_rt_forward_clip (round 2: 3.0 ms composed of per-operator calls)"}


def leg_other_configs(c):
    """BASELINE configs[2..4] as short child-process legs"""
    args, frames_per_step, step, result, extras = c.args, c.frames_per_step, c.step, c.result, c.extras
    # BASELINE configs[2..4] (the bf16 configurations) as short legs in child processes, outside the headline's timed region:
    # frames/s, ms per step, parity against the bf16 twin on 3 frames, the conv family's roofline fraction
    import subprocess
    others = {}
    for c in (3, 4, 5):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(c), "--steps", "3", "--warmup", "1", "--no-extras",
               "--cpu-sample-frames", "3", "--cpu-timeout", "240"] + (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CRFP_FORCE_DIST")}
        try:
            r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
            j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            others[str(c)] = {"workload": j["config"]["workload"], "frames_per_sec": j["value"], "ms_per_step": j["ms_per_step"],
                              "frames_per_step": j["frames_per_step_per_gpu"], "steps": j["steps"], "dtype": j["dtype"],
                              "clips_in_flight": j["config"]["clips_in_flight_per_gpu"], "parity_vs_twin": j.get("parity"),
                              "conv": {k: j["roofline"].get(k) for k in ("kernel", "frac", "achieved", "peak", "unit", "avg_launch_us")}
                              if "roofline" in j else None,
                              "dcn_fused_avg_us": j.get("dcn_fused", {}).get("avg_us"),
                              "warp_dcn_frac": j.get("warp_dcn", {}).get("frac"),
                              "warp_dcn_8d_frac": j.get("warp_dcn_8d", {}).get("frac"),
                              "cpu_baseline": j.get("cpu_baseline")}
        except Exception as e:  # noqa: BLE001 -- a failed leg must not take the headline line down with it
            others[str(c)] = {"error": f"{type(e).__name__}: {e}"[:300]}
    result["other_configs"] = others


def leg_mask_gate(c):
    """dense launches (CRFP_MASK_GATE=0) in a child process"""
    args, storage, h, w, fv, result, extras = c.args, c.storage, c.h, c.w, c.fv, c.result, c.extras
    # The fovea blend is a select under the mask, and the engine skips, tile by tile, the work whose result the select discards
    # (DESIGN.md 3.3).  For the record: the same headline with dense launches (CRFP_MASK_GATE=0 is read once per process -> child process),
    # and what share of the frame this workload's fovea covers.
    import subprocess
    gate = {"enabled": os.environ.get("CRFP_MASK_GATE", "1") != "0",
            "fovea": f"{fv} x {fv} of {8 * h} x {8 * w} pixels per frame (the reference's eval.sh: --FV_size 96), {100.0 * fv * fv / (64.0 * h * w):.2f} % of the frame",
            "skipped_where_the_mask_is_clear": ["x8 frame stack (hr_prep)", "encoder_hr.slice1.0 / .2", "conv_tttf (the blend keeps lrelu(state) there)"],
            "outputs": "bit-identical to the dense launches (tests/test_gpu_gate.py: 46 mask x storage x wiring x schedule cases)"}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "2", "--steps", str(min(args.steps, 10)), "--warmup", "2", "--no-extras",
           "--no-cpu-baseline", "--no-kernel-profile", "--no-other-configs"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CRFP_FORCE_DIST")}
    env["CRFP_MASK_GATE"] = "0"
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        gate["dense_frames_per_sec"] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["value"]
    except Exception as e:  # noqa: BLE001 -- a failed leg must not take the headline line down with it
        gate["dense_frames_per_sec"] = None
        gate["error"] = f"{type(e).__name__}: {e}"[:300]
    result["mask_gate"] = gate


def leg_dcn_g8_alone(c):
    """DCNv2 of dcn_0/1/2 alone (CRFP_DCN_FUSED=0, child process) + north_star_per_kernel"""
    h, w, result, fam, extras = c.h, c.w, c.result, c.fam, c.extras
    # The north star's ">= 60 % of the HBM roofline on the flow_warp + DCNv2 kernels", kernel by kernel.  In the shipped
    # schedule DCNv2 of dcn_0/1/2 runs INSIDE the fused kernel, whose time also holds the 28.7 GFLOP offset / mask head conv, so the DCNv2
    # kernel itself is measured here on the two-kernel path (CRFP_DCN_FUSED=0, read once per process -> child process): SURVEY 8(d)'s API
    # bytes (258.1 MB fp32 per launch @A) over the time of dcn_g8_pipe_kernel alone.
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "2", "--steps", "3", "--warmup", "2", "--no-extras", "--no-cpu-baseline",
           "--no-other-configs"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CRFP_FORCE_DIST")}
    env["CRFP_DCN_FUSED"] = "0"
    alone = {"switch": "CRFP_DCN_FUSED=0 (two-kernel path: conv_mfma:dcn.offset_mask + dcn_g8_pipe_kernel), child process"}
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        pk = j["warp_dcn"]["per_kernel"]["dcnv2_g8_c32"]
        alone.update({"kernel": "dcn_g8_pipe_kernel", "avg_us": pk["avg_us"], "achieved": pk["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": pk["frac"], "bytes": "SURVEY 8(d) API bytes: (32 + 144 + 72 + 32) H W s + weights",
                      "two_kernel_path_frames_per_sec": j["value"]})
    except Exception as e:  # noqa: BLE001 -- a failed leg must not take the headline line down with it
        alone["error"] = f"{type(e).__name__}: {e}"[:300]
    result["dcn_g8_alone"] = alone
    pkm = result["warp_dcn"]["per_kernel"]
    ns = {n: {"avg_us": v["avg_us"], "frac": v["frac"]} for n, v in pkm.items()}
    if "frac" in alone:
        ns["dcnv2_g8_c32 (two-kernel path)"] = {"avg_us": alone["avg_us"], "frac": alone["frac"]}
    if "dcnv2_shared_c4_fused" in pkm and "frac_api_tensor_bytes" in result["warp_dcn"]:
        # dcn_3: the compact bytes the kernel moves (credit taken) and the 9x-replicated API tensors it never materialises (labelled)
        d3 = fam.get("dcnv2_shared_c4_fused")
        ns["dcnv2_shared_c4_fused"]["frac_at_api_tensor_bytes"] = (d3["bytes"] + d3["launches"] * (8 * h) * (8 * w) * 24 * 4.0) / (d3["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        # `frac` prices the fused kernel's own I/O (x + the offset conv's 4-channel input + flow + out = 14 values per pixel: the head conv runs
        # inside); SURVEY 8(d)'s compact DCNv2 bytes are (Cin + 2 + 1 + Cout) = 11 values per pixel (print both)
        esz = 4.0 if c.storage == "f32" else 2.0
        compact = d3["launches"] * (8.0 * h) * (8.0 * w) * c.bclips * ((4 + 4) * esz + 3 * 4.0)
        ns["dcnv2_shared_c4_fused"]["frac_survey_compact_bytes"] = compact / (d3["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        ns["dcnv2_shared_c4_fused"]["frac_is"] = "fused-kernel I/O (x, the offset conv's input, flow, out); frac_survey_compact_bytes = (Cin + 2 + 1 + Cout) H W s"
    result["north_star_per_kernel"] = {"target_frac": 0.60, "kernels": ns,
                                       "meets_target": sorted(n for n, v in ns.items() if v["frac"] >= 0.60),
                                       "below_target": sorted(n for n, v in ns.items() if v["frac"] < 0.60),
                                       "note": "fraction of 8 TB/s on each kernel's algorithmic bytes (SURVEY 8(d)); the fused head + DCNv2 kernel is an "
                                               "MFMA-bound kernel reported under dcn_fused, its DCNv2 part is priced here on the two-kernel path"}


def leg_cpu_baseline(c):
    """cpu_baseline + parity: the oracle in a child process"""
    args, cfg, custom, rank, world, mode, storage, t, h, w, fv, data1, data_np, mk8, eng, result = c.args, c.cfg, c.custom, c.rank, c.world, c.mode, c.storage, c.t, c.h, c.w, c.fv, c.data1, c.data_np, c.mk8, c.eng, c.result
    # The oracle (CPU port of the reference path; checker / baseline only, never on the product path) runs in a child
    # process so that a mis-sized host cannot stall the bench: bounded sample, hard timeout.
    import subprocess
    import tempfile
    from oracle import crfp_oracle as orc
    full = args.config == 2 and not custom and world == 1
    # N > 1: the other ranks wait in the closing barrier while rank 0 times the oracle, so the sample stays small (3 frames, no warm-up)
    ns = args.cpu_sample_frames or (t if full else 3)
    ns = max(2, min(ns, t))
    tmp = os.path.join(tempfile.mkdtemp(), "oracle_sample.npz")
    cmd = [sys.executable, "-m", "oracle.run_sample", "--frames", str(ns), "--h", str(h), "--w", str(w),
           "--fv-size", str(fv), "--sigma-t", str(cfg["sigma"]), "--clip-seed", "1234", "--clip-frames", str(t),
           "--storage", storage, "--warmup", "1" if full else "0", "--out", tmp]
    try:
        # N > 1: the other ranks sit in the closing barrier (RCCL's watchdog allows 10 minutes): bound the oracle sample well below that
        subprocess.run(cmd, cwd=ROOT, timeout=args.cpu_timeout if world == 1 else min(args.cpu_timeout, 240.0), check=True)
        z = np.load(tmp)
        ref, cpu_s = torch.from_numpy(z["out"]), float(z["seconds"])
        lrs, fvs, mks = data1[0]
        with torch.no_grad():
            if mode == "stream":
                eng.clear_states()
                got = torch.stack([eng.stream_frame(lrs[0, i], fvs[0, i], mk8[0, i]) for i in range(ns)])[None].cpu()
            else:
                got = eng.forward(lrs[:, :ns], fvs[:, :ns], mks[:, :ns]).cpu()
        dd = (got - ref).abs()
        hr = torch.from_numpy(np.clip(data_np[0][1][:, :ns], 0, 1))
        py_ref = np.mean([orc.psnr_rgb_and_y(ref[0, i:i + 1], hr[0, i:i + 1])[1] for i in range(ns)])
        py_got = np.mean([orc.psnr_rgb_and_y(got[0, i:i + 1], hr[0, i:i + 1])[1] for i in range(ns)])
        what = "oracle/crfp_oracle.py on torch-CPU fp32" if storage == "f32" else "oracle/crfp_oracle.py bf16-storage twin on torch-CPU"
        result["cpu_baseline"] = {"value": ns / cpu_s, "unit": "frames/s", "cores": int(z["threads"]), "kind": "port",
                                  "sample": (f"{'all' if ns == t else 'first'} {ns} frames of the same {h}x{w} clip, {what}, "
                                             f"{'1 warm-up pass + ' if full else ''}1 timed pass of {cpu_s:.1f} s wall"),
                                  "host_cpus": os.cpu_count(), "usable_cpus": int(z["usable_cpus"]),
                                  "torch_threads": int(z["threads"])}
        if storage == "f32":
            result["parity"] = {"max_abs_diff_vs_oracle": float(dd.max()), "tolerance": 1e-3, "frames": ns,
                                "psnr_y_delta_db": float(abs(py_ref - py_got))}
        else:
            result["parity"] = {"vs": "bf16-storage oracle twin (rounds where the engine stores)", "max_abs_diff": float(dd.max()),
                                "mean_abs_diff": float(dd.mean()), "psnr_db": float(-10 * torch.log10((dd.double() ** 2).mean())),
                                "tolerance": "mean <= 5e-4, max <= 3e-2 (tests/test_gpu_bf16.py; the twin itself sits ~1.3e-2 / 3.6e-4 from fp32)",
                                "frames": ns, "psnr_y_delta_db": float(abs(py_ref - py_got))}
    except (subprocess.TimeoutExpired, subprocess.CalledProcessError) as e:
        result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": None, "kind": "port",
                                  "sample": f"oracle sample of {ns} frames did not finish: {type(e).__name__}"}
