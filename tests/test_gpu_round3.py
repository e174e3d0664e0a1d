"""GPU (-m gpu), round 3: the shapes VERDICT r2 found untested -- one RCCL rank started the way the driver starts N ranks, bf16 streaming at the real
180x320 size, the command-line entry (`crfp_amd.main`, eval.sh's flags) end to end -- plus this round's kernels."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

T = torch.from_numpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def orc():
    from oracle import crfp_oracle
    return crfp_oracle


@pytest.fixture(autouse=True)
def _nograd():
    with torch.no_grad():
        yield


def _model(sd_np, y_only=False, cls="CRFP_DSV", storage="f32"):
    from crfp_amd.model import CRFP
    m = getattr(CRFP, cls)(device=dev(), mid_channels=32, y_only=y_only, hr_dcn=True, offset_prop=True)
    m.load_state_dict({k: T(v.copy()) for k, v in sd_np.items()}, strict=True)
    m.storage = storage
    return m.to(dev()).eval()


def _stats(got, ref):
    d = (got.detach().cpu().double() - ref.double()).abs()
    mse = float((d ** 2).mean())
    return float(d.max()), float(d.mean()), (99.0 if mse == 0 else -10 * np.log10(mse))


# ------------------------------------------------------------------------------------------------ config 4, one rank
# (round 5: the rank's 4 clips are ONE lock-step crfp_dsv_forward_batch call since round 4; the full-shape parity test of that call --
# bit-identity to one-clip calls + oracle / twin -- is tests/test_gpu_round5.py::test_config4_lockstep_batch_at_the_real_shape, which
# replaced round 3's two-calls-in-flight test here.  Two calls in flight on two caller streams stay covered by
# tests/test_gpu_parity.py::test_concurrent_clips_on_two_streams_are_bit_exact, tests/test_gpu_round4.py::test_two_batches_in_flight_on_two_streams_are_bit_exact and by round 5's two-host-thread test.)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_config4_under_torchrun_initialises_rccl_with_one_rank():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 --config 4` in a FRESH child process (the launcher
    starts before anything touches the GPU) with CRFP_FORCE_DIST=1: the `nccl` (= RCCL) group is initialised at world size 1,
    so barrier / MAX-reduce of the step time / SUM-reduce of the PSNR sums -- the whole collective path of bench.py -- runs
    on RCCL on hardware."""
    env = dict(os.environ, CRFP_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "4", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-kernel-profile"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["collectives"]["backend"] == "nccl" and j["collectives"]["initialised"] and j["collectives"]["world_size"] == 1
    assert j["n_gpus"] == 1 and j["dtype"] == "bf16" and j["config"]["clips_per_gpu_per_step"] == 4
    # round 4: the 4 clips of a rank go through ONE crfp_dsv_forward_batch call (lock-step), no longer two one-clip calls in flight
    assert j["config"]["clips_per_call"] == 4 and j["config"]["calls_in_flight_per_gpu"] == 1 and j["value"] > 0
    assert j["psnr_reduce"]["ranks"] == 1 and j["psnr_reduce"]["frames"] == 28.0 and np.isfinite(j["psnr_reduce"]["sum_sq_err"])


# ------------------------------------------------------------------------------------------------ config 3 at full size
def test_bf16_stream_180x320_12_calls_vs_twin(orc):
    """BASELINE configs[2] at its real geometry: one frame per call at 180x320 -> 1440x2560, bf16 storage, sigma^T = 50 gaze
    trajectory, 12 calls against the streaming twin (the 100-call run of test_gpu_bf16.py is at 45x80)."""
    from crfp_amd import gaze, synth
    sd = synth.make_state_dict(7)
    P = orc.bf16_weights(orc.load_numpy_state(sd))
    h, w, N, fv = 180, 320, 12, 96
    lrs = np.concatenate([synth.make_clip(4321 + i, 1, min(10, N - i), h, w, fv_size=fv)[0][0] for i in range(0, N, 10)], 0)
    lr = T(lrs)
    rs = np.random.RandomState(11)
    gt = torch.clamp(F.interpolate(lr, scale_factor=8, mode="bilinear", align_corners=False) +
                     T(rs.normal(0, 0.03, (N, 3, 8 * h, 8 * w)).astype(np.float32)), 0, 1)
    H, W = 8 * h, 8 * w
    xs, ys = gaze.gaze_trajectory(N, H, W, 50.0, np.random.RandomState(4321))
    masks = gaze.RegionMasks(H, W, fv, torch.device("cpu"))
    m = _model(sd, cls="MRCF_simple_v18", storage="bf16")
    m.clear_states()
    d = dev()
    means, maxs = [], []
    with orc.bf16_storage():
        so = orc.StreamOracle(P)
        for n in range(N):
            cy, cx = gaze.window_origin(xs[n], ys[n], fv, H, W)
            mk = masks.frame(n, cy, cx)["mk"]
            f = gt[n:n + 1] * mk
            ref = so(lr[n:n + 1].unsqueeze(0), f.unsqueeze(0), mk.unsqueeze(0))
            got = m(lrs=lr[n:n + 1].unsqueeze(0).to(d), fvs=f.unsqueeze(0).to(d), mks=mk.unsqueeze(0).to(d)).cpu()
            dd = (got - ref).abs()
            means.append(float(dd.mean())); maxs.append(float(dd.max()))
    print("bf16 stream @180x320, mean|HIP - twin| per call:", " ".join(f"{v:.2e}" for v in means), "| max:", f"{max(maxs):.2e}")
    assert max(means) <= 3e-4 and max(maxs) <= 3e-2
    assert np.mean(means[6:]) <= 2.0 * np.mean(means[1:6]) + 1e-6


# ------------------------------------------------------------------------------------------------ CLI end to end
def _write_reds_tree(tmp_path, rs, n_img=4, gh=128, gw=192):
    import PIL.Image
    from crfp_amd.dataset import reds
    gt_root = str(tmp_path / "REDS_sharp")
    lr_root = gt_root.replace("_sharp", "_sharp_BI_x8")
    for clip in reds.REDS4:
        base = rs.uniform(0, 255, (n_img, gh + 8, gw + 8, 3))
        for i in range(n_img):
            g = base[i % 2, i:i + gh, i:i + gw].astype(np.uint8)
            for root, img in ((gt_root, g), (lr_root, np.array(PIL.Image.fromarray(g).resize((gw // 8, gh // 8), PIL.Image.BICUBIC)))):
                dd = os.path.join(root, "val/val/val_sharp", clip)
                os.makedirs(dd, exist_ok=True)
                PIL.Image.fromarray(img).save(os.path.join(dd, f"{i:08d}.png"))
    return gt_root


def test_cli_main_eval_end_to_end_vs_oracle(orc, tmp_path):
    """`python -m crfp_amd.main` with eval.sh's flags (reference eval.sh:2-20 -> main.py:19-68): experiment directory, model
    factory, every checkpoint of sorted(os.listdir(model_path)) through Trainer.load's key rule (one checkpoint carries
    `module.`-free `basic_` keys and a stray key, trainer.py:185-199) and Trainer.eval_basicvsr on a synthetic REDS-shaped
    tree; the four logged means must equal the oracle driven through the same harness."""
    from crfp_amd import evalrig, main, option, synth
    from crfp_amd.dataset import reds
    rs = np.random.RandomState(5)
    gt_root = _write_reds_tree(tmp_path, rs)
    ckpt_dir = tmp_path / "ckpts"
    os.makedirs(ckpt_dir)
    sds = [synth.make_state_dict(7), synth.make_state_dict(8)]
    torch.save({k: T(v.copy()) for k, v in sds[0].items()}, str(ckpt_dir / "a_model.pt"))
    # second checkpoint: stray keys the model lacks (dropped by the key rule; one of them a `basic_` key that the rule renames
    # first) and one key MISSING -- Trainer.load overlays the checkpoint on the model's current state, so that tensor stays
    # what checkpoint a left there
    sd_b = {k: T(v.copy()) for k, v in sds[1].items() if k != "conv_last.bias"}
    sd_b["basic_stray.weight"] = torch.zeros(3)
    sd_b["module.not_in_model"] = torch.zeros(1)
    torch.save(sd_b, str(ckpt_dir / "b_model.pt"))
    sds[1] = dict(sds[1], **{"conv_last.bias": sds[0]["conv_last.bias"]})
    # eval.sh's flag list, with the tree / checkpoint paths and the (small) clip geometry substituted
    argv = ["--save_dir", str(tmp_path / "exp"), "--reset", "True", "--num_gpu", "1", "--gpu_id", "0", "--log_file_name", "eval.log",
            "--eval", "True", "--eval_save_results", "True", "--num_workers", "1", "--scale", "8", "--cra", "true", "--mrcf", "true",
            "--hr_dcn", "true", "--offset_prop", "true", "--N_frames", "3", "--FV_size", "32", "--GT_size", "128", "--dataset", "Reds",
            "--dataset_dir", gt_root, "--model_path", str(ckpt_dir), "--visdom_port", "8803", "--visdom_view", "cli_test"]
    results = main.main(argv)
    assert len(results) == 2
    log = open(tmp_path / "exp" / "eval.log").read()
    assert log.count("load_model_path") == 2 and log.count("Ref  PSNR (now)") == 2 and log.count("Ref  PSNR_Y (now)") == 2
    assert os.path.exists(tmp_path / "exp" / "args.txt")
    args = option.parse(argv)
    ds = reds.EvalSet(args)
    for sd, res in zip(sds, results):
        P = orc.load_numpy_state(sd)

        def oracle_frames(i_batch):
            item = ds[i_batch]
            sr = orc.crfp_dsv_forward(P, item["LR"][None], item["Ref"][None], item["Ref_sp"][None].float(), orc.DSVConfig())[0]
            ones = torch.ones(1, 1, *sr.shape[2:])
            out = []
            for i in evalrig.counted_frames(i_batch, sr.shape[0]):
                a, b = sr[i:i + 1], item["HR"][i:i + 1]
                p, s = orc.calc_psnr_and_ssim(a, b, ones)
                py, sy = orc.calc_psnr_and_ssim(orc.to_y(a.permute(0, 2, 3, 1)), orc.to_y(b.permute(0, 2, 3, 1)), ones)
                out.append((p, s, py, sy))
            return out

        ref = evalrig.evaluate(oracle_frames, len(ds))
        assert res["frames"] == ref["frames"] == 8 * 3 - 1
        for k in ("psnr", "psnr_y"):
            assert abs(res[k] - ref[k]) < 2e-3, (k, res[k], ref[k])
        for k in ("ssim", "ssim_y"):
            assert abs(res[k] - ref[k]) < 2e-5, (k, res[k], ref[k])
    assert abs(results[0]["psnr"] - results[1]["psnr"]) > 1e-6     # the second checkpoint really replaced the first


# ------------------------------------------------------------------------------------------------ numerics guards
def test_out_of_range_weight_is_named_at_pack_time(orc):
    """The split-fp16 scheme needs |w| < 32.  A weight outside that range must be reported as such, with the tensor's name,
    before any kernel runs (VERDICT r2 Weak #12: it used to surface as an activation overflow); precision='f32' runs it."""
    from crfp_amd import synth
    sd = synth.make_state_dict(7)
    sd = dict(sd)
    wkey = "forward_resblocks_1.main.2.0.conv1.weight"
    sd[wkey] = sd[wkey].copy()
    sd[wkey][3, 5, 1, 1] = 40.0
    lrs, fvs, mks = synth.make_clip(5, 1, 2, 16, 24, fv_size=48)
    d = dev()
    a = dict(lrs=T(lrs).to(d), fvs=T(fvs).to(d), mks=T(mks).to(d))
    m = _model(sd)
    with pytest.raises(ValueError, match="forward_resblocks_1.main.2.0.conv1.weight"):
        m(**a)
    ref = orc.crfp_dsv_forward(orc.load_numpy_state(sd), T(lrs), T(fvs), T(mks))
    m.on_overflow = "fallback"                      # falls back to strict fp32 by itself
    assert float((m(**a).cpu() - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max()))
    m.on_overflow, m.precision = "poison", "f32"
    assert float((m(**a).cpu() - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max()))
    mb = _model(sd, storage="bf16")                 # bf16 operands have fp32's exponent range: no limit there
    assert torch.isfinite(mb(**a)).all()


def test_overflow_flag_covers_every_clip_of_a_batch_and_fresh_workspaces():
    """overflowed() after forward() on n > 1 clips is the OR over the clips (each clip resets the workspace's status word,
    ADVICE r2), and a workspace allocated for another shape (compute_flow) starts clear."""
    from crfp_amd import synth
    sd = synth.make_state_dict(7)
    m = _model(sd)
    h, w, t = 16, 24, 2
    a = synth.make_clip(5, 1, t, h, w, fv_size=48)
    b = synth.make_clip(6, 1, t, h, w, fv_size=48)
    d = dev()
    lrs = torch.cat([T(a[0]) * 1.0e5, T(b[0])]).to(d)
    fvs = torch.cat([T(a[1]) * 1.0e5, T(b[1])]).to(d)
    mks = torch.cat([T(a[2]), T(b[2])]).to(d)
    out = m(lrs=lrs, fvs=fvs, mks=mks)
    assert m.engine().overflowed()                                  # clip 0 overflowed, clip 1 (the last) did not
    assert torch.isnan(out[0]).all() and torch.isfinite(out[1]).all()
    out = m(lrs=lrs[1:], fvs=fvs[1:], mks=mks[1:])
    assert not m.engine().overflowed()
    m.compute_flow(torch.cat([lrs[1:], lrs[1:, :1]], 1))            # 3 frames: another workspace shape, freshly allocated
    assert not m.engine().overflowed()


# ------------------------------------------------------------------------------------------------ round-3 kernels
def test_bf16_fused_conv_pairs_are_bit_identical():
    """bf16 build: dcn_block.0 -> .2 and res conv1 -> conv2(+x) of every level run as ONE launch each with the tensor between the
    two convolutions kept in LDS (conv3x3_bf16_pair_kernel: 62-column tiles, halo recompute, bf16 rounding of the intermediate
    exactly where the two-kernel path stores it).  Same arithmetic in the same order: the clip must not change by a bit --
    on the golden clip (two 62-column tiles per row, ragged last one) and at the real 180 x 320 geometry."""
    from test_gpu_parity import _golden_check
    a = _golden_check({"CRFP_CHECK_STORAGE": "bf16"}, want="DIGEST")
    b = _golden_check({"CRFP_CHECK_STORAGE": "bf16", "CRFP_CONV_PAIR": "0"}, want="DIGEST")
    assert a == b
    tool = os.path.join(ROOT, "tools", "ab_sites.py")
    r = subprocess.run([sys.executable, tool, "--storage", "bf16", "--rounds", "1", "--steps", "1", "--t", "3", "--sites", "pair",
                        "pair=", "single=CRFP_CONV_PAIR=0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = {ln.split()[0]: ln for ln in r.stdout.splitlines() if " digest " in ln}
    assert set(lines) == {"pair", "single"}, r.stdout + r.stderr[-2000:]
    dig = {k: v.split(" digest ")[1].split()[0] for k, v in lines.items()}
    assert dig["pair"] == dig["single"], lines
    assert "conv_mfma_pair" in lines["pair"] and "conv_mfma_pair" not in lines["single"]


def test_role_specialised_fused_dcn_kernel_is_bit_identical():
    """dcn_fused2_kernel (CRFP_DCN_FUSE_V=2: 8 conv waves hand the activated offsets / masks of every cout tile to 8 sampler waves
    through LDS, 16 waves per workgroup) against the shipped single-role kernel: same arithmetic, same order, same bits."""
    from test_gpu_parity import _golden_check
    # round 5: the 16-wave form lives in the lab library only (the product reads no CRFP_DCN_FUSE_V)
    assert _golden_check({}, want="DIGEST") == _golden_check({"CRFP_DCN_FUSE_V": "2"}, want="DIGEST", lab=True)
    tool = os.path.join(ROOT, "tools", "ab_sites.py")
    r = subprocess.run([sys.executable, tool, "--rounds", "1", "--steps", "1", "--t", "3", "v1=", "v2=CRFP_DCN_FUSE_V=2,lab"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    dig = {ln.split()[0]: ln.split(" digest ")[1].split()[0] for ln in r.stdout.splitlines() if " digest " in ln}
    assert set(dig) == {"v1", "v2"} and dig["v1"] == dig["v2"], r.stdout + r.stderr[-2000:]


# ------------------------------------------------------------------------------------------------ regional wiring as one C-ABI call (f4)
def _runtime_model(seed, y_only=False):
    from crfp_amd import synth
    from crfp_amd.model import MRCF_runtime
    m = MRCF_runtime.MRCF_simple_v18(mid_channels=32, y_only=y_only, hr_dcn=True, offset_prop=True, split_ratio=3,
                                     spynet_pretrained='pretrained_models/fnet.pth', device=dev())
    sd = synth.make_state_dict_like({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed)
    m.load_state_dict({k: T(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.to(dev()).eval()
    m.print_timings = False       # -> crfp_rt_forward_clip (csrc/engine_rt.hip)
    return m, sd


def test_runtime_engine_one_call_vs_reference_golden_and_oracle(orc):
    """``MRCF_runtime.MRCF_simple_v18(...)(lr, fv, warp_size=)`` through ``crfp_rt_forward_clip`` -- the whole wiring of
    model/CRFP_runtime.py:8469-8664 scheduled inside the library -- against the output of the reference class itself
    (tests/golden/runtime_small.npz), against the per-operator composition, and against the oracle at geometries the golden does
    not cover: window = whole frame, a window that is not a multiple of 64, ragged frame sizes, t = 1, y_only."""
    from oracle import runtime_oracle as ro
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "runtime_small.npz")))
    m, sd = _runtime_model(int(g["weights_seed"]))
    lrs, fvs, warp = T(g["lrs"]).to(dev()), T(g["fvs"]).to(dev()), tuple(int(v) for v in g["warp"])
    out = m(lrs, fvs, warp_size=warp)
    assert tuple(out.shape) == g["out"].shape and _stats(out, T(g["out"]))[0] < 2e-4
    assert not m.engine().overflowed()
    staged = m.forward_staged(lrs, fvs, warp)
    assert _stats(out, staged.cpu())[0] < 2e-4
    again = m(lrs, fvs, warp_size=warp)                     # same workspace, second clip: the recurrent state is reset per call
    assert torch.equal(out, again)
    m.engine().single_stream = True                         # the four-stream schedule changes no bit
    assert torch.equal(out, m(lrs, fvs, warp_size=warp))
    m.engine().single_stream = False
    P = orc.load_numpy_state(sd)
    rs = np.random.RandomState(5)
    for (t, h, w, fv, wp) in ((2, 16, 40, 48, (128, 192)),     # window = the whole frame in y
                              (2, 17, 27, 40, (136, 216)),     # window = the whole (ragged) frame
                              (3, 21, 30, 56, (72, 104)),      # window not a multiple of 64, FNet resize 8 x 8 -> 9 x 13
                              (1, 12, 20, 96, (64, 64))):      # one frame: no flow, no DCN; fovea = the whole frame height
        l = T(rs.rand(1, t, 3, h, w).astype(np.float32))
        f = T(rs.rand(1, t, 3, fv, fv).astype(np.float32))
        got = m(l.to(dev()), f.to(dev()), warp_size=wp)
        ref = ro.runtime_forward(P, l, f, wp)
        assert tuple(got.shape) == tuple(ref.shape) and _stats(got, ref)[0] < 2e-4, (t, h, w, fv, wp, _stats(got, ref))
    my, sdy = _runtime_model(3, y_only=True)
    l = T(rs.rand(2, 2, 3, 16, 24).astype(np.float32))
    f = T(rs.rand(2, 2, 3, 32, 32).astype(np.float32))
    got = my(l.to(dev()), f.to(dev()), warp_size=(96, 128))
    ref = ro.runtime_forward(orc.load_numpy_state(sdy), l, f, (96, 128), y_only=True)
    assert tuple(got.shape) == (2, 2, 1, 128, 192) and _stats(got, ref)[0] < 2e-4
    # geometry the one-call schedule cannot run: the ENGINE refuses it by name ...
    with pytest.raises(ValueError, match="geometry"):
        m.engine().forward(lrs, fvs, (256, 192))            # taller than the 192-row frame
    # ... and the module routes such calls through the per-operator composition, as the reference clamps its window by slicing
    # (model/CRFP_runtime.py:8487,8548): the default warp_size on a small frame == the clamped window
    big = m(lrs, fvs, warp_size=(1080, 1920))
    ref_big = ro.runtime_forward(P, lrs.cpu(), fvs.cpu(), (1080, 1920))
    assert _stats(big, ref_big)[0] < 2e-4
    assert _stats(big, m(lrs, fvs, warp_size=(8 * lrs.shape[-2], 8 * lrs.shape[-1])).cpu())[0] < 2e-4
    # `.data` writes + invalidate_packed, like CRFP_DSV
    m.conv_last.weight.data *= 0.5
    m.invalidate_packed()
    half = m(lrs, fvs, warp_size=warp)
    assert _stats(half, out.cpu())[0] > 1e-3


def test_runtime_engine_overflow_poisons_output():
    """an activation beyond the fp16 operand range inside the regional schedule: the frames come back NaN and overflowed() says why"""
    m, sd = _runtime_model(11)
    m.encoder_lr.slice1[0].bias.data.fill_(7.0e4)
    m.invalidate_packed()
    l = torch.rand(1, 2, 3, 16, 24, device=dev())
    f = torch.rand(1, 2, 3, 32, 32, device=dev())
    out = m(l, f, warp_size=(64, 64))
    assert m.engine().overflowed() and torch.isnan(out).all()


def test_runtime_engine_is_graph_capturable_and_replays_bit_exact():
    """crfp_rt_forward_clip (four streams, fork / join through events, three memsets) captured into a HIP graph once and replayed on
    new inputs equals the eager call bit for bit; graph replay is also how a caller removes the 1.1 ms of host enqueue time per clip."""
    m, _ = _runtime_model(5)
    rs = np.random.RandomState(8)
    clips = [(T(rs.rand(1, 3, 3, 24, 40).astype(np.float32)).to(dev()), T(rs.rand(1, 3, 3, 64, 64).astype(np.float32)).to(dev())) for _ in range(2)]
    eager = [m(l, f, warp_size=(128, 192)).clone() for l, f in clips]     # also creates the library's side streams and events
    L, Fv = clips[0][0].clone(), clips[0][1].clone()
    eng = m.engine()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = eng.forward(L, Fv, (128, 192))
    for i in (0, 1, 0):
        L.copy_(clips[i][0]); Fv.copy_(clips[i][1])
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager[i])


# ------------------------------------------------------------------------------------------------ streaming with resident inputs
@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_stream_with_resident_inputs_is_bit_identical(storage):
    """CRFP_DSV_INPUTS_RESIDENT (engine.inputs_resident): the library keeps the previous LR frame inside the workspace and starts the
    state-independent part of frame i on its side stream without waiting for the caller's stream, i.e. beside frame i - 1.  Same
    arithmetic on alternating buffer sets: every frame of a 9-call sequence with a clear_states() in the middle equals the plain
    one-frame-per-call path bit for bit, on two streams and on one; a sequence that does not start with the flag is refused."""
    from crfp_amd import _lib, synth
    sd = synth.make_state_dict(7)
    t, h, w = 9, 24, 40
    lrs, fvs, mks = (T(a).to(dev()) for a in synth.make_clip(77, 1, t, h, w, fv_size=64, sigma_t=10.0))
    mks = mks.bool() if mks.dtype != torch.bool else mks
    fg = torch.zeros(1, 8 * h, 8 * w, dtype=torch.bool, device=dev())
    fg[:, 40:150, 60:260] = True

    def run(resident, single=False):
        eng = _model(sd, storage=storage).engine()
        eng.inputs_resident, eng.single_stream = resident, single
        outs = []
        for i in range(t):
            if i == 5:
                eng.clear_states()
            outs.append(eng.stream_frame(lrs[0, i], fvs[0, i], mks[0, i], fg if i in (3, 7) else None).clone())   # + the regional mask on two frames
        torch.cuda.synchronize()
        return torch.stack(outs), eng

    base, _ = run(False)
    res, eng = run(True)
    assert torch.isfinite(base).all() and torch.equal(base, res)
    assert torch.equal(base, run(True, single=True)[0])
    # back-to-back sequences without host synchronisation, outputs checked at the end: the early side work of call i really overlaps call i - 1
    eng.clear_states()
    outs = [eng.stream_frame(lrs[0, i], fvs[0, i], mks[0, i], fg if i == 3 else None) for i in range(5)]
    assert torch.equal(torch.stack(outs), base[:5])
    # a float mask would need a conversion kernel on the caller's stream: refused instead of raced
    with pytest.raises(ValueError, match="inputs_resident"):
        eng.stream_frame(lrs[0, 5], fvs[0, 5], mks[0, 5].float())
    # the C-ABI refuses a flagged call on a workspace whose previous call did not keep its frame
    L = _lib.lib()
    sfx = "_bf16" if storage == "bf16" else ""
    nb = getattr(L, "crfp_dsv_workspace_bytes" + sfx)(1, h, w)
    ws = torch.zeros(nb + 512, dtype=torch.uint8, device=dev())[256:]   # an address no earlier sequence of this thread started at
    out = torch.empty(3, 8 * h, 8 * w, device=dev())
    mk8 = mks[0, 1].view(torch.uint8)
    rc = getattr(L, "crfp_dsv_stream_frame" + sfx)(eng.packed.data_ptr(), _lib.DSV_INPUTS_RESIDENT, lrs[0, 1].data_ptr(), None, fvs[0, 1].data_ptr(),
                                                   mk8.data_ptr(), None, out.data_ptr(), 0, h, w, ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
    assert rc < 0 and b"kept previous frame" in L.crfp_last_error_string()   # CRFP_E_BADARG
