"""GPU (-m gpu): parity of the HIP path against the oracle and the committed golden vectors.
Every call goes through the C-ABI of libcrfp_hip.so (crfp_amd.ops / crfp_amd.engine).
Tolerances: the north star asks |delta| < 1e-3 on the fp32 x8 SR frame; operator-level checks are
held much tighter (fp32 re-association only)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from conftest import GOLDEN  # noqa: E402

T = torch.from_numpy


def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def maxdiff(a, b):
    a = a.detach().cpu() if isinstance(a, torch.Tensor) else T(np.asarray(a))
    b = b.detach().cpu() if isinstance(b, torch.Tensor) else T(np.asarray(b))
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max())


@pytest.fixture(scope="module")
def orc():
    from oracle import crfp_oracle
    return crfp_oracle


@pytest.fixture(autouse=True)
def _nograd():
    with torch.no_grad():
        yield


def test_library_loaded_and_version():
    from crfp_amd import _lib
    assert _lib.lib().crfp_version() == 201
    assert os.path.exists(_lib.LIB_PATH)


@pytest.mark.parametrize("cin,cout,h,w,act", [(3, 32, 17, 70, "lrelu"), (6, 32, 8, 64, "relu"), (32, 32, 24, 40, "none"),
                                              (64, 32, 33, 65, "lrelu"), (66, 32, 9, 130, "relu"), (32, 216, 12, 20, "sigmoid"),
                                              (128, 256, 5, 9, "relu"), (32, 2, 16, 16, "tanh"), (24, 64, 8, 8, "none")])
def test_conv3x3_mfma(cin, cout, h, w, act):
    from crfp_amd import ops
    rs = np.random.RandomState(cin * 7 + cout)
    x = T(rs.standard_normal((2, cin, h, w)).astype(np.float32))
    wt = T((rs.standard_normal((cout, cin, 3, 3)) * (1.5 / np.sqrt(cin * 9))).astype(np.float32))
    b = T(rs.standard_normal(cout).astype(np.float32))
    ref = F.conv2d(x, wt, b, padding=1)
    ref = {"none": lambda v: v, "relu": F.relu, "lrelu": lambda v: F.leaky_relu(v, 0.1), "tanh": torch.tanh,
           "sigmoid": torch.sigmoid}[act](ref)
    got = ops.conv3x3(x.to(dev()), wt.to(dev()), b.to(dev()), act)
    assert maxdiff(got, ref) < 2e-5


@pytest.mark.parametrize("mode", ["zeros", "border"])
def test_flow_warp_golden(ops_golden, mode):
    from crfp_amd import ops
    g = ops_golden
    got = ops.flow_warp(T(g["warp_x"]).to(dev()), T(g["warp_flow"]).to(dev()), padding_mode=mode)
    assert maxdiff(got, g["warp_" + mode]) < 1e-5


@pytest.mark.parametrize("c,h,w", [(32, 36, 64), (4, 144, 256), (24, 37, 61)])
def test_flow_warp_random(orc, c, h, w):
    from crfp_amd import ops
    rs = np.random.RandomState(c + h)
    x = T(rs.standard_normal((1, c, h, w)).astype(np.float32))
    fl = T(rs.uniform(-9, 9, (1, h, w, 2)).astype(np.float32))
    got = ops.flow_warp(x.to(dev()), fl.to(dev()))
    assert maxdiff(got, orc.flow_warp(x, fl)) < 2e-5


def test_flow_warp_shape_error():
    from crfp_amd import ops
    with pytest.raises(ValueError):
        ops.flow_warp(torch.zeros(1, 2, 4, 5, device=dev()), torch.zeros(1, 4, 6, 2, device=dev()))


def test_cpu_tensor_rejected():
    from crfp_amd import ops
    with pytest.raises(RuntimeError):
        ops.flow_warp(torch.zeros(1, 4, 4, 4), torch.zeros(1, 4, 4, 2))


@pytest.mark.parametrize("C,O,dg,H,W", [(32, 32, 8, 23, 45), (32, 32, 8, 8, 32), (4, 4, 1, 40, 70), (8, 12, 2, 9, 11)])
def test_dcnv2(orc, C, O, dg, H, W):
    from crfp_amd import ops
    rs = np.random.RandomState(C * 31 + H)
    x = T(rs.standard_normal((2, C, H, W)).astype(np.float32))
    off = rs.uniform(-6, 6, (2, 2 * dg * 9, H, W)).astype(np.float32)
    off[0, :, 0, 0] = -40.0
    off[1, 1, 2, 2] = float(W + 5)
    off[1, 0, 1, 1] = -1.0
    off = T(off)
    msk = T(rs.uniform(0, 1, (2, dg * 9, H, W)).astype(np.float32))
    wt = T((rs.standard_normal((O, C, 3, 3)) * 0.2).astype(np.float32))
    b = T(rs.standard_normal(O).astype(np.float32))
    ref = orc.dcnv2(x, off, msk, wt, b, dg)
    got = ops.dcnv2(*[t.to(dev()) for t in (x, off, msk, wt, b)], 3, 1, 1, dg)
    assert maxdiff(got, ref) < 3e-5


def test_dcn_module_identity_kat(ops_golden):
    """reference known-answer: fresh DCN_module == 0.5 * flow_warp (model/CRFP.py:354-370)."""
    from crfp_amd.model import CRFP
    g = ops_golden
    for tag, m in (("n", CRFP.DCN_module(32, 8, 3, 10)), ("r", CRFP.DCN_module(4, 1, 3, 10, repeat=True))):
        m = m.to(dev())
        pre, flow = T(g[f"kat_{tag}_pre"]).to(dev()), T(g[f"kat_{tag}_flow"]).to(dev())
        a, _ = m(torch.randn_like(pre), pre, torch.randn_like(pre), flow)
        assert maxdiff(a, g[f"kat_{tag}_halfwarp"]) < 2e-5


def test_upsample(orc):
    from crfp_amd import ops
    rs = np.random.RandomState(5)
    x = T(rs.standard_normal((2, 3, 13, 21)).astype(np.float32))
    for r in (2, 4, 8):
        assert maxdiff(ops.upsample_bilinear(x.to(dev()), scale_factor=r), orc.up_bilinear(x, r)) < 1e-5
    ref = F.interpolate(x, size=(20, 36), mode="bilinear", align_corners=False)
    assert maxdiff(ops.upsample_bilinear(x.to(dev()), size=(20, 36)), ref) < 1e-5


def _load_sub(module, sd, prefix):
    module.load_state_dict({k[len(prefix):]: T(v.copy()) for k, v in sd.items() if k.startswith(prefix)}, strict=True)
    return module.to(dev())


def test_modules_vs_golden(ops_golden, weights_np):
    from crfp_amd.model import CRFP, LTE
    g, sd = ops_golden, weights_np
    d = dev()
    m = _load_sub(CRFP.PixelShufflePack(32, 24, 2, 3), sd, "upsample.")
    assert maxdiff(m(T(g["psp_x"]).to(d)), g["psp_y"]) < 2e-5
    m = _load_sub(CRFP.PixelUnShufflePack_v2(4, 32, 4, 3), sd, "downsample.")
    assert maxdiff(m(T(g["pusp_x"]).to(d)), g["pusp_y"]) < 2e-5
    m = _load_sub(CRFP.ResidualBlocksWithInputConv(64, 32, 1), sd, "forward_resblocks_1.")
    assert maxdiff(m(T(g["res_x"]).to(d)), g["res_y"]) < 2e-5
    m = _load_sub(LTE.LTE_simple_lr(32), sd, "encoder_lr.")
    assert maxdiff(m(T(g["enc_lr_x"]).to(d), islr=True)[2], g["enc_lr_y"]) < 2e-5
    m = _load_sub(LTE.LTE_simple_hr_single(4), sd, "encoder_hr.")
    assert maxdiff(m(T(g["enc_hr_x"]).to(d), islr=True)[2], g["enc_hr_y"]) < 2e-5
    m = _load_sub(CRFP.FNet(3), sd, "spynet.")
    for tag in "ab":
        assert maxdiff(m(T(g[f"fnet_{tag}_x1"]).to(d), T(g[f"fnet_{tag}_x2"]).to(d)), g[f"fnet_{tag}_y"]) < 1e-4
    m = _load_sub(CRFP.DCN_module(32, 8, 3, 10, pre_offset=True, interpolate="none"), sd, "dcn_1.")
    a, o = m(*[T(g[k]).to(d) for k in ("dcn1_cur", "dcn1_pre", "dcn1_prew", "dcn1_flow", "dcn1_poff")])
    assert maxdiff(a, g["dcn1_aligned"]) < 1e-4 and maxdiff(o, g["dcn1_offfeat"]) < 2e-5
    m = _load_sub(CRFP.DCN_module(4, 1, 3, 10, repeat=True, pre_offset=True, interpolate="pixelshuffle"), sd, "dcn_3.")
    a, o = m(*[T(g[k]).to(d) for k in ("dcn3_cur", "dcn3_pre", "dcn3_prew", "dcn3_flow", "dcn3_poff")])
    assert maxdiff(a, g["dcn3_aligned"]) < 1e-4 and maxdiff(o, g["dcn3_offfeat"]) < 2e-5


def _model(sd_np, y_only=False):
    from crfp_amd.model import CRFP
    m = CRFP.CRFP_DSV(device=dev(), mid_channels=32, y_only=y_only, hr_dcn=True, offset_prop=True)
    m.load_state_dict({k: T(v.copy()) for k, v in sd_np.items()}, strict=True)
    return m.to(dev()).eval()


@pytest.mark.parametrize("name", ["dsv_16x24_t3", "dsv_20x36_t4", "dsv_16x24_t2_yonly"])
def test_full_forward_golden(name):
    from crfp_amd import synth
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    y_only = bool(g["y_only"])
    sd = synth.make_state_dict(int(g["weights_seed"]), y_only=y_only)
    lrs, fvs, mks = synth.make_clip(int(g["clip_seed"]), 1, int(g["t"]), int(g["h"]), int(g["w"]),
                                    fv_size=int(g["fv_size"]), sigma_t=10.0)
    m = _model(sd, y_only)
    d = dev()
    flows, _ = m.compute_flow(T(lrs).to(d))
    assert maxdiff(flows, g["flows"]) < 2e-3          # flow is in pixels (tanh*256 amplifies round-off)
    out = m(lrs=T(lrs).to(d), fvs=T(fvs).to(d), mks=T(mks).to(d))
    assert maxdiff(out, g["out"]) < 1e-3              # the north-star tolerance
    assert maxdiff(out, g["out"]) < 2e-4              # what fp32 MFMA re-association actually gives


def test_bisect_intermediates_small(orc):
    """First frame + second frame intermediates against the oracle (localises a wiring error)."""
    from crfp_amd import synth
    sd = synth.make_state_dict(7)
    P = orc.load_numpy_state(sd)
    h, w, t = 16, 24, 2
    lrs, fvs, mks = synth.make_clip(3, 1, t, h, w, fv_size=48)
    m = _model(sd)
    d = dev()
    out = m(lrs=T(lrs).to(d), fvs=T(fvs).to(d), mks=T(mks).to(d))
    ref = orc.crfp_dsv_forward(P, T(lrs), T(fvs), T(mks))
    eng = m.engine()
    x_lr = eng.debug_fetch("x_lr", t, h, w)
    ref_xlr = orc.lrelu(orc.conv(P, "encoder_lr.slice1.2", orc.lrelu(orc.conv(P, "encoder_lr.slice1.0", T(lrs)[0]))))
    assert maxdiff(x_lr, ref_xlr) < 2e-5
    assert maxdiff(out[:, 0], ref[:, 0]) < 1e-4, "first frame (no warp / DCN) differs"
    assert maxdiff(out[:, 1], ref[:, 1]) < 2e-4, "second frame (warp + DCN) differs"


def test_streaming_equals_clip():
    from crfp_amd import synth
    sd = synth.make_state_dict(7)
    h, w, t = 16, 24, 4
    lrs, fvs, mks = synth.make_clip(5, 1, t, h, w, fv_size=48)
    m = _model(sd)
    d = dev()
    L, Fv, M = T(lrs).to(d), T(fvs).to(d), T(mks).to(d)
    clip = m(lrs=L, fvs=Fv, mks=M)
    m.clear_states()
    outs = [m.forward_stream(L[:, i:i + 1], Fv[:, i:i + 1], M[:, i:i + 1]) for i in range(t)]
    assert maxdiff(torch.cat(outs, dim=1), clip) == 0.0


def test_full_size_two_frames_vs_oracle(orc):
    """BASELINE config A geometry (180x320 -> 1440x2560), 2 frames, against the oracle."""
    from crfp_amd import synth
    sd = synth.make_state_dict(7)
    P = orc.load_numpy_state(sd)
    lrs, fvs, mks = synth.make_clip(1234, 1, 2, 180, 320, fv_size=96, sigma_t=10.0)
    m = _model(sd)
    d = dev()
    out = m(lrs=T(lrs).to(d), fvs=T(fvs).to(d), mks=T(mks).to(d))
    ref = orc.crfp_dsv_forward(P, T(lrs), T(fvs), T(mks))
    assert maxdiff(out, ref) < 1e-3
    p_ref = orc.psnr_rgb_and_y(ref[0, 1:2], T(np.clip(fvs[0, 1:2] * 0 + 0.5, 0, 1)))  # exercise metric code path
    assert np.isfinite(p_ref[0])


def test_psnr_sums(orc, ops_golden):
    from crfp_amd import ops
    g = ops_golden
    sr, hr = T(g["metric_sr"]), T(g["metric_hr"])
    acc = ops.sq_err_sums(sr.to(dev()), hr.to(dev())).cpu()
    n = sr.numel()
    psnr = -10.0 * np.log10(float(acc[0]) / n)
    psnr_y = -10.0 * np.log10(float(acc[1]) / (n / 3) / 255.0 ** 2)
    assert abs(psnr - float(g["metric_psnr"])) < 1e-3
    assert abs(psnr_y - float(g["metric_psnr_y"])) < 1e-3


def test_streaming_variant_golden():
    """reference model/CRFP_test.py MRCF_simple_v18 (one frame per call, fgs regional mask, clear_states)."""
    from crfp_amd import synth
    from crfp_amd.model import CRFP
    g = dict(np.load(os.path.join(GOLDEN, "stream_16x24_t7.npz")))
    sd = synth.make_state_dict(int(g["weights_seed"]))
    t, h, w = int(g["t"]), int(g["h"]), int(g["w"])
    lrs, fvs, mks = synth.make_clip(int(g["clip_seed"]), 1, t, h, w, fv_size=int(g["fv_size"]), sigma_t=10.0)
    d = dev()
    m = CRFP.MRCF_simple_v18(device=d, mid_channels=32, split_ratio=3)
    m.load_state_dict({k: T(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.to(d).eval()
    L, Fv, M, Fg = T(lrs).to(d), T(fvs).to(d), T(mks).to(d), T(g["fgs"]).to(d)
    outs = []
    for i in range(t):
        if i == int(g["clear_at"]):
            m.clear_states()
        outs.append(m(L[:, i:i + 1], Fv[:, i:i + 1], M[:, i:i + 1], Fg[:, i:i + 1]))
    assert maxdiff(torch.cat(outs, dim=1), g["out"]) < 2e-4


LAB_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "crfp_amd", "libcrfp_hip_lab.so")


def _golden_check(env, lab=False, want="MAXDIFF"):
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    if lab:
        if not os.path.exists(LAB_LIB):
            pytest.skip("lab library not built (make -C crfp_amd/csrc lab)")
        e["CRFP_HIP_LIB"] = LAB_LIB
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "run_golden_check.py")],
                         capture_output=True, text=True, env=e, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith(want)]
    assert line, out.stderr[-2000:]
    return line[0]


@pytest.mark.parametrize("env", [{"CRFP_PRECISION": "f32"}, {"CRFP_SIDE_STREAM": "0"}, {"CRFP_DCN_FUSED": "0"}, {"CRFP_MASK_GATE": "0"}])
def test_alternate_kernel_paths(env):
    """The process-wide switches the product library reads (strict fp32 MFMA for convs and the DCN GEMM, single-stream
    schedule, two-kernel DCN path, dense fovea-side launches) give the same clip within the parity tolerance."""
    assert float(_golden_check(env).split()[1]) < 2e-4


@pytest.mark.parametrize("env", [{"CRFP_SPLIT_WS": "1"}, {"CRFP_SPLIT_IS": "1"}, {"CRFP_SPLIT_RPW": "2"}, {"CRFP_SPLIT_PIPE": "1"},
                                 {"CRFP_CONV_MODE": "bf16x6"}, {"CRFP_CONV_MODE": "f32"}, {"CRFP_DCN_MODE": "f32"}])
def test_lab_kernel_paths(env):
    """The lab library (-DCRFP_LAB: every conv main loop that was tried -- split-bf16 single-role / input-stationary /
    warp-specialised / pipelined, 4- and 8-row tiles) stays correct, so the A/B numbers in DESIGN.md remain reproducible."""
    assert float(_golden_check(env, lab=True).split()[1]) < 2e-4


@pytest.mark.parametrize("h,w,t,fv", [(18, 26, 3, 48), (33, 47, 3, 64), (17, 65, 3, 48), (64, 16, 2, 48), (21, 130, 2, 64)])
def test_odd_geometries_vs_oracle(orc, h, w, t, fv):
    """Sizes that are no multiple of any tile (4x64 conv tiles, 16x64 stencil tiles, 32-pixel DCN rows, FNet's three
    poolings) against the CPU oracle."""
    from crfp_amd import synth
    sd = synth.make_state_dict(7)
    P = orc.load_numpy_state(sd)
    lrs, fvs, mks = synth.make_clip(1000 + h * w, 1, t, h, w, fv_size=fv, sigma_t=10.0)
    m = _model(sd)
    d = dev()
    out = m(lrs=T(lrs).to(d), fvs=T(fvs).to(d), mks=T(mks).to(d))
    ref = orc.crfp_dsv_forward(P, T(lrs), T(fvs), T(mks))
    assert maxdiff(out, ref) < 1e-4


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_small_offset_regime_vs_oracle(orc, storage):
    """SURVEY 8(d)'s offset / mask head scale (offset_std = 0.02: DCN residuals of a fraction of a pixel around the flow, as after
    training) instead of the stress weights every other test uses: the regular 4x4-neighbourhood path of dcn_3 and near-identity
    sampling in the fused DCN kernel, fp32 against the oracle and bf16 storage against its twin."""
    from crfp_amd import synth
    sd = synth.make_state_dict(7, offset_std=0.02)
    P = orc.load_numpy_state(sd)
    lrs, fvs, mks = synth.make_clip(77, 1, 3, 24, 40, fv_size=64, sigma_t=10.0)
    m = _model(sd)
    m.storage = storage
    d = dev()
    out = m(lrs=T(lrs).to(d), fvs=T(fvs).to(d), mks=T(mks).to(d))
    ref = orc.crfp_dsv_forward(P, T(lrs), T(fvs), T(mks))
    if storage == "f32":
        assert maxdiff(out, ref) < 1e-4
    else:   # the yardstick of tests/test_gpu_bf16.py: the twin (bf16 weights, rounds where the engine stores) and its distance to fp32
        from test_gpu_bf16 import _check_frame_stats, _twin
        _check_frame_stats(out, _twin(orc, sd, lrs, fvs, mks), ref, "small offsets 3x24x40")


def test_batch_of_two_clips(orc):
    """n = 2 (the reference's forward takes a batch; here the clips of a batch go through the C-ABI one by one)."""
    from crfp_amd import synth
    sd = synth.make_state_dict(7)
    P = orc.load_numpy_state(sd)
    lrs, fvs, mks = synth.make_clip(31, 2, 3, 16, 24, fv_size=48)
    m = _model(sd)
    d = dev()
    L, Fv, M = T(lrs).to(d), T(fvs).to(d), T(mks).to(d)
    out = m(lrs=L, fvs=Fv, mks=M)
    assert out.shape[0] == 2
    for b in range(2):
        one = m(lrs=L[b:b + 1], fvs=Fv[b:b + 1], mks=M[b:b + 1])
        assert maxdiff(out[b:b + 1], one) == 0.0
    ref = orc.crfp_dsv_forward(P, T(lrs), T(fvs), T(mks))
    assert maxdiff(out, ref) < 1e-4


def test_runtime_rig_smoke():
    """crfp_amd.runtime_rig (shape and arithmetic of the reference's test_runtime.py) on a small frame."""
    from crfp_amd import runtime_rig
    for variant in ("dsv", "regional"):
        y, spf = runtime_rig.run(repeat_time=3, warm_up=1, t=2, hr=(136, 200), fv_size=48, warp_size=(96, 96), variant=variant)
        assert tuple(y.shape) == (1, 2, 3, 136, 200) and spf > 0 and bool(torch.isfinite(y).all()), variant
    lr = torch.zeros(1, 2, 3, 17, 25)
    lrs, fvs, mks = runtime_rig.build_inputs(lr, torch.ones(1, 2, 3, 48, 48), (96, 96))
    assert int(mks.sum()) == 2 * 48 * 48 and float(fvs.sum()) == 2 * 3 * 48 * 48 and bool(mks[0, 0, 0, 24, 24])


def test_producer_split_is_bit_identical():
    """SRC_S3 (the producing conv writes the fp16 pair image) must not change a single bit: the
    producer applies the same split the consumer would.  (The switch exists in the lab library only.)"""
    digests = [_golden_check({"CRFP_CONV_S3": s3}, lab=True, want="DIGEST") for s3 in ("1", "0")]
    assert digests[0] == digests[1]
    assert _golden_check({}, want="DIGEST") == digests[0]      # and the product library computes exactly the same clip


def test_fused_offset_conv_dcn_is_bit_identical():
    """dcn_fused_kernel (offset / mask head + dcn_g8 in one launch, the offsets never leave the registers) applies
    the arithmetic of the two-kernel path in the same order: not one bit of the clip may differ (20x36 clip: partial tiles in x
    and y at 2x resolution)."""
    assert _golden_check({}, want="DIGEST") == _golden_check({"CRFP_DCN_FUSED": "0"}, want="DIGEST")


def test_config_b_geometry_vs_oracle(orc):
    """BASELINE config 5 geometry (270x480 -> 2160x3840): partial tiles in y at 2x and 8x resolution."""
    from crfp_amd import synth
    sd = synth.make_state_dict(7)
    P = orc.load_numpy_state(sd)
    lrs, fvs, mks = synth.make_clip(77, 1, 2, 270, 480, fv_size=144, sigma_t=10.0)
    m = _model(sd)
    d = dev()
    out = m(lrs=T(lrs).to(d), fvs=T(fvs).to(d), mks=T(mks).to(d))
    ref = orc.crfp_dsv_forward(P, T(lrs), T(fvs), T(mks))
    assert maxdiff(out, ref) < 1e-3


def _full_clip(seed, t):
    from crfp_amd import synth
    lrs, fvs, mks = synth.make_clip(seed, 1, t, 180, 320, fv_size=96, sigma_t=10.0)
    d = dev()
    return T(lrs).to(d), T(fvs).to(d), T(mks).to(d)


def test_two_stream_schedule_is_bit_exact():
    """crfp_dsv_forward_clip forks state-independent work onto an internal side stream; the result must not
    depend on the schedule (this caught the packed-FP32 / bf16-MFMA co-residency hazard)."""
    from crfp_amd import synth
    m = _model(synth.make_state_dict(7))
    clip = _full_clip(1234, 7)
    os.environ["CRFP_SIDE_STREAM"] = "0"
    try:
        ref = m(lrs=clip[0], fvs=clip[1], mks=clip[2]).clone()
    finally:
        os.environ.pop("CRFP_SIDE_STREAM")
    for _ in range(4):
        out = m(lrs=clip[0], fvs=clip[1], mks=clip[2])
        torch.cuda.synchronize()
        assert maxdiff(out, ref) == 0.0


@pytest.mark.parametrize("t", [1, 2, 3])
def test_two_stream_schedule_short_clips(t):
    """Clip lengths at which the side-stream schedule changes shape (no FNet at t=1, FNet behind frame 0's pre-work
    from t=2, buffer-set reuse from t=3): same bits as the single-stream schedule."""
    from crfp_amd import synth
    m = _model(synth.make_state_dict(7))
    lrs, fvs, mks = synth.make_clip(99 + t, 1, t, 36, 64, fv_size=96)
    d = dev()
    L, Fv, M = T(lrs).to(d), T(fvs).to(d), T(mks).to(d)
    os.environ["CRFP_SIDE_STREAM"] = "0"
    try:
        ref = m(lrs=L, fvs=Fv, mks=M).clone()
    finally:
        os.environ.pop("CRFP_SIDE_STREAM")
    for _ in range(3):
        out = m(lrs=L, fvs=Fv, mks=M)
        torch.cuda.synchronize()
        assert maxdiff(out, ref) == 0.0


def test_concurrent_clips_on_two_streams_are_bit_exact():
    """Two engines, two caller streams, clips in flight together == the same clips run one after the other."""
    from crfp_amd import synth
    sd = synth.make_state_dict(7)
    models = [_model(sd), _model(sd)]
    clips = [_full_clip(100, 4), _full_clip(101, 4)]
    refs = [models[k](lrs=clips[k][0], fvs=clips[k][1], mks=clips[k][2]).clone() for k in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(4):
        outs = [None, None]
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                outs[k] = models[k](lrs=clips[k][0], fvs=clips[k][1], mks=clips[k][2])
        torch.cuda.synchronize()
        for k in range(2):
            assert maxdiff(outs[k], refs[k]) == 0.0


def test_masked_psnr_ssim_golden_and_oracle(ops_golden, orc):
    """crfp_amd.utils.calc_psnr_and_ssim_cuda (one HIP pass) vs the reference's values (golden) and vs the oracle on a
    ragged size with a random mask and the [0,255] / [-1,1] range branches (utils.py:242-254)."""
    from crfp_amd import utils as U
    g, d = ops_golden, dev()
    sr, hr = T(g["metric_sr"]).to(d), T(g["metric_hr"]).to(d)
    ones = torch.ones(1, 1, *sr.shape[2:], device=d)
    for mask, tag in ((ones, ""), (T(g["metric_box"]).to(d), "_box"), (T(g["metric_ring"]).to(d), "_ring")):
        p, s = U.calc_psnr_and_ssim_cuda(sr, hr, mask)
        assert abs(float(p) - float(g["metric_psnr" + tag])) < 1e-4      # dB
        assert abs(float(s) - float(g["metric_ssim" + tag])) < 2e-6
    p, s = U.calc_psnr_and_ssim_cuda(sr * 255.0, hr * 255.0, T(g["metric_box"]).to(d))
    assert abs(float(p) - float(g["metric_psnr_box255"])) < 1e-4 and abs(float(s) - float(g["metric_ssim_box255"])) < 2e-6
    ys, yh = U.bgr2ycbcr(sr.permute(0, 2, 3, 1), y_only=True), U.bgr2ycbcr(hr.permute(0, 2, 3, 1), y_only=True)
    py, sy = U.calc_psnr_and_ssim_cuda(ys, yh, ones)
    assert abs(float(py) - float(g["metric_psnr_y"])) < 1e-4 and abs(float(sy) - float(g["metric_ssim_y"])) < 2e-6
    # ragged tiles, 2 images, random mask, all three range branches
    rs = np.random.RandomState(5)
    a = rs.uniform(0, 1, (2, 3, 70, 150)).astype(np.float32)
    b = np.clip(a + rs.normal(0, 0.05, a.shape), 0, 1).astype(np.float32)
    m = (rs.uniform(0, 1, (2, 1, 70, 150)) > 0.6)
    for scale, shift in ((1.0, 0.0), (255.0, 0.0), (2.0, -1.0)):
        A, B = T(a) * scale + shift, T(b) * scale + shift
        pr, sr_ = orc.calc_psnr_and_ssim(A, B, T(m).float())
        p, s = U.calc_psnr_and_ssim_cuda(A.to(d), B.to(d), T(m).to(d))
        assert abs(float(p) - pr) < 1e-4 and abs(float(s) - sr_) < 2e-6
    # identical images: the reference's mse == 0 special case
    p, s = U.calc_psnr_and_ssim_cuda(sr, sr, ones)
    assert abs(float(p) - orc.psnr(sr.cpu(), sr.cpu(), ones.cpu())) < 1e-4 and abs(float(s) - 1.0) < 1e-6


def test_evalrig_frame_metrics_vs_oracle(orc):
    """The four per-frame figures Trainer.eval_basicvsr logs (trainer.py:348-369), HIP vs oracle."""
    from crfp_amd import evalrig
    rs = np.random.RandomState(9)
    hr = rs.uniform(0, 1, (1, 3, 96, 160)).astype(np.float32)
    sr = (hr + rs.normal(0, 0.02, hr.shape)).astype(np.float32)
    p, s, py, sy = evalrig.frame_metrics(T(sr).to(dev()), T(hr).to(dev()))
    ones = torch.ones(1, 1, 96, 160)
    pr, sr_ = orc.calc_psnr_and_ssim(T(sr), T(hr), ones)
    pyr, syr = orc.calc_psnr_and_ssim(orc.to_y(T(sr).permute(0, 2, 3, 1)), orc.to_y(T(hr).permute(0, 2, 3, 1)), ones)
    assert abs(p - pr) < 1e-4 and abs(s - sr_) < 2e-6 and abs(py - pyr) < 1e-4 and abs(sy - syr) < 2e-6
    p2, py2 = evalrig.frame_psnrs(T(sr).to(dev()), T(hr).to(dev()))
    assert abs(p2 - p) < 1e-4 and abs(py2 - py) < 1e-4


def test_gaze_video_rig_vs_oracle(orc):
    """crfp_amd.gaze.run_gaze_video (streaming model + region metrics on the GPU, regional-DCN mask on) against the
    same loop driven through the oracle's StreamOracle and oracle metrics (test_video.py:303-379)."""
    from crfp_amd import gaze, synth
    from crfp_amd.model import CRFP
    sd = synth.make_state_dict(7)
    P = orc.load_numpy_state(sd)
    h, w, N, fv = 16, 24, 5, 32
    lrs, fvs, _ = synth.make_clip(21, 1, N, h, w, fv_size=fv)
    lr = T(lrs[0])
    rs = np.random.RandomState(4)
    gt = torch.clamp(F.interpolate(lr, scale_factor=8, mode="bilinear", align_corners=False) +
                     T(rs.normal(0, 0.02, (N, 3, 8 * h, 8 * w)).astype(np.float32)), 0, 1)
    m = CRFP.MRCF_simple_v18(device=dev(), mid_channels=32)
    m.load_state_dict({k: T(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.to(dev()).eval()
    res = gaze.run_gaze_video(m, lr.to(dev()), gt.to(dev()), sigma=6.0, fv_size=fv, seed=11, fv_start=1,
                              regional_dcn=True, rg=96)

    class OracleModel:
        def __init__(self):
            self.o = orc.StreamOracle(P)

        def clear_states(self):
            self.o.clear_states()

        def __call__(self, lrs, fvs, mks, fgs):
            return self.o(lrs, fvs, mks.float(), fgs.float())

    ref = gaze.run_gaze_video(OracleModel(), lr, gt, sigma=6.0, fv_size=fv, seed=11, fv_start=1, regional_dcn=True, rg=96,
                              metric_fn=lambda a, b, mk: orc.calc_psnr_and_ssim(a, b, mk.float()))
    assert res["trajectory"] == ref["trajectory"] and res["frames"] == N
    for r in ("whole", "fovea", "outskirt", "past"):
        assert abs(res[f"psnr_{r}"] - ref[f"psnr_{r}"]) < 2e-3, r          # dB
        assert abs(res[f"ssim_{r}"] - ref[f"ssim_{r}"]) < 2e-5, r
    assert len(res["per_frame"]["past"]) == N - 1


def test_avgpool2_and_fovea_head_ops():
    """C-ABI ops of SURVEY section 8b: crfp_avgpool2_f32 (odd sizes: floor mode) and the fused fovea head
    crfp_fovea_head_f32 (model/CRFP.py:1672-1684) against plain ATen on the CPU."""
    from crfp_amd import ops
    rs = np.random.RandomState(17)
    x = T(rs.normal(0, 1, (2, 5, 45, 81)).astype(np.float32))
    assert maxdiff(ops.avgpool2(x.to(dev())), F.avg_pool2d(x, 2, 2)) < 1e-6
    n, h, w = 2, 9, 13
    H, W = 8 * h, 8 * w
    state = T(rs.normal(0, 0.5, (n, 4, H, W)).astype(np.float32))
    x_hr = T(rs.normal(0, 0.5, (n, 4, H, W)).astype(np.float32))
    mask = T(rs.uniform(0, 1, (n, 1, H, W)) > 0.7)
    lr = T(rs.uniform(0, 1, (n, 3, h, w)).astype(np.float32))
    wt, bt = T(rs.normal(0, 0.2, (4, 8, 3, 3)).astype(np.float32)), T(rs.normal(0, 0.1, (4,)).astype(np.float32))
    for y_only in (False, True):
        co = 1 if y_only else 3
        wl, bl = T(rs.normal(0, 0.2, (co, 4, 3, 3)).astype(np.float32)), T(rs.normal(0, 0.1, (co,)).astype(np.float32))
        f2 = F.conv2d(torch.cat((state, x_hr), 1), wt, bt, padding=1)
        mk = mask.float()
        ns_ref = F.leaky_relu(mk * f2 + (1 - mk) * state, 0.1)
        base = F.interpolate(lr, scale_factor=8, mode="bilinear", align_corners=False)
        if y_only:
            base = 0.299 * base[:, 0:1] + 0.587 * base[:, 1:2] + 0.114 * base[:, 2:3]
        out_ref = F.conv2d(ns_ref, wl, bl, padding=1) + base
        d = dev()
        ns, out = ops.fovea_head(state.to(d), x_hr.to(d), mask.to(d), lr.to(d), wt.to(d), bt.to(d), wl.to(d), bl.to(d), y_only)
        assert maxdiff(ns, ns_ref) < 2e-5 and maxdiff(out, out_ref) < 2e-5


@pytest.mark.parametrize("y_only", [False, True])
def test_eval_reds_end_to_end_vs_oracle(orc, tmp_path, y_only):
    """dataset.reds.EvalSet -> model -> per-frame PSNR/SSIM/-Y -> means (Trainer.eval_basicvsr, trainer.py:295-413) on a
    synthetic REDS-shaped PNG tree: HIP path vs the oracle driven through the same harness code."""
    import types
    import PIL.Image
    from crfp_amd import evalrig, synth
    from crfp_amd.dataset import reds
    from crfp_amd.model import CRFP
    rs = np.random.RandomState(77)
    gt_root = str(tmp_path / "REDS_sharp")
    lr_root = gt_root.replace("_sharp", "_sharp_BI_x8")
    for clip in reds.REDS4:
        base = rs.uniform(0, 255, (4, 8 * 16 + 8, 8 * 24 + 8, 3))
        for i in range(4):
            g = base[i % 2, i:i + 128, i:i + 192].astype(np.uint8)
            for root, img in ((gt_root, g), (lr_root, np.array(PIL.Image.fromarray(g).resize((24, 16), PIL.Image.BICUBIC)))):
                d = os.path.join(root, "val/val/val_sharp", clip)
                os.makedirs(d, exist_ok=True)
                PIL.Image.fromarray(img).save(os.path.join(d, f"{i:08d}.png"))
    args = types.SimpleNamespace(dataset_dir=gt_root, scale=8, N_frames=3, GT_size=128, FV_size=32)
    sd = synth.make_state_dict(7, y_only=y_only)
    m = _model(sd, y_only)
    res = evalrig.eval_reds(m, args, device=dev())
    P = orc.load_numpy_state(sd)

    def oracle_frames(i_batch):
        item = reds.EvalSet(args)[i_batch]
        sr = orc.crfp_dsv_forward(P, item["LR"][None], item["Ref"][None], item["Ref_sp"][None].float(),
                                  orc.DSVConfig(y_only=y_only))[0]
        if y_only:   # trainer.py:331-335, restated with the trainer's own coefficients (:19-48)
            r, g_, b = item["LR_sr"][:, 0], item["LR_sr"][:, 1], item["LR_sr"][:, 2]
            u = -0.147 * r - 0.289 * g_ + 0.436 * b
            v = 0.615 * r - 0.515 * g_ - 0.100 * b
            y = sr[:, 0]
            sr = torch.stack([y + 1.14 * v, y + -0.396 * u - 0.581 * v, y + 2.029 * u], 1)
        out = []
        ones = torch.ones(1, 1, *sr.shape[2:])
        for i in evalrig.counted_frames(i_batch, sr.shape[0]):
            a, b = sr[i:i + 1], item["HR"][i:i + 1]
            p, s = orc.calc_psnr_and_ssim(a, b, ones)
            py, sy = orc.calc_psnr_and_ssim(orc.to_y(a.permute(0, 2, 3, 1)), orc.to_y(b.permute(0, 2, 3, 1)), ones)
            out.append((p, s, py, sy))
        return out

    # several windows per model call (lock-step batch inside the library): the very same numbers
    res3 = evalrig.eval_reds(m, args, device=dev(), clips_per_call=3)
    assert res3 == res
    ref = evalrig.evaluate(oracle_frames, len(reds.EvalSet(args)))
    assert res["frames"] == ref["frames"] == 8 * 3 - 1      # frame 0 of batch 0 is skipped (trainer.py:350-351)
    for k in ("psnr", "psnr_y"):
        assert abs(res[k] - ref[k]) < 2e-3, k
    for k in ("ssim", "ssim_y"):
        assert abs(res[k] - ref[k]) < 2e-5, k
