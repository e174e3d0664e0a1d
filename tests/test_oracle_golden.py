"""CPU: pin the oracle (oracle/crfp_oracle.py) against the golden vectors produced by the imported
reference (tests/golden/make_golden.py).  Same ATen ops => tolerance is float32 round-off only."""
import ctypes
import os

import numpy as np
import pytest
import torch

from crfp_amd import synth
from oracle import crfp_oracle as orc

from conftest import GOLDEN

T = torch.from_numpy
TOL = 2e-6


def sub(sd, prefix):
    return {k: T(v.copy()) for k, v in sd.items() if k.startswith(prefix)}


def close(a, b, tol=TOL):
    a = a.numpy() if isinstance(a, torch.Tensor) else a
    d = float(np.max(np.abs(a - b))) if a.size else 0.0
    assert a.shape == b.shape and d <= tol, f"shape {a.shape} vs {b.shape}, max|d|={d}"


@pytest.fixture(autouse=True)
def _nograd():
    with torch.no_grad():
        yield


def test_state_dict_table_matches_reference(weights_np):
    # make_golden.py asserted list(model.state_dict().keys()) == synth.state_dict_keys() on the reference
    assert len(weights_np) == 2 * len(synth.conv_spec())
    assert sum(v.size for v in weights_np.values()) == 2284352
    assert sum(v.size for k, v in weights_np.items() if k.startswith("spynet.")) == 1745506


@pytest.mark.parametrize("mode", ["zeros", "border"])
def test_flow_warp(ops_golden, mode):
    g = ops_golden
    close(orc.flow_warp(T(g["warp_x"]), T(g["warp_flow"]), padding_mode=mode), g["warp_" + mode], 0.0)


def test_flow_warp_size_mismatch_raises():
    with pytest.raises(ValueError):
        orc.flow_warp(torch.zeros(1, 2, 4, 5), torch.zeros(1, 4, 6, 2))


def test_flow_warp_c_restatement(ops_golden, oracle_c_lib):
    g = ops_golden
    x, fl = g["warp_x"], g["warp_flow"]
    n, c, h, w = x.shape
    fp = ctypes.POINTER(ctypes.c_float)
    for border, key in ((0, "warp_zeros"), (1, "warp_border")):
        out = np.empty_like(x)
        rc = oracle_c_lib.flow_warp_ref(x.ctypes.data_as(fp), fl.ctypes.data_as(fp), out.ctypes.data_as(fp),
                                        n, c, h, w, border)
        assert rc == 0
        close(out, g[key], 2e-5)


def test_pixel_unshuffle_equals_torch(ops_golden):
    g = ops_golden
    close(torch.nn.functional.pixel_unshuffle(T(g["unshuffle_x"]), 4), g["unshuffle_y"], 0.0)


def test_packs_and_blocks(ops_golden, weights_np):
    g, P = ops_golden, orc.load_numpy_state(weights_np)
    close(orc.pixel_shuffle_pack(P, "upsample.", T(g["psp_x"]), 2), g["psp_y"])
    close(orc.pixel_unshuffle_pack_v2(P, "downsample.", T(g["pusp_x"]), 4), g["pusp_y"])
    close(orc.resblocks_with_input_conv(P, "forward_resblocks_1.", T(g["res_x"])), g["res_y"])
    x = T(g["enc_lr_x"])
    close(orc.lrelu(orc.conv(P, "encoder_lr.slice1.2", orc.lrelu(orc.conv(P, "encoder_lr.slice1.0", x)))), g["enc_lr_y"])
    x = T(g["enc_hr_x"])
    close(orc.lrelu(orc.conv(P, "encoder_hr.slice1.2", orc.lrelu(orc.conv(P, "encoder_hr.slice1.0", x)))), g["enc_hr_y"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_fnet(ops_golden, weights_np, tag):
    g, P = ops_golden, orc.load_numpy_state(weights_np)
    # flows are 256 * tanh(.) pixels: one fp32 ulp at that magnitude is 3e-5, and torch's CPU convolutions change their summation
    # order with the thread partition (seen once: 2e-5 exceeded on a loaded host)
    close(orc.fnet(P, "spynet.", T(g[f"fnet_{tag}_x1"]), T(g[f"fnet_{tag}_x2"])), g[f"fnet_{tag}_y"], 1e-4)


def test_dcn_module_wiring(ops_golden, weights_np):
    g, P = ops_golden, orc.load_numpy_state(weights_np)
    a, o = orc.dcn_module(P, "dcn_1.", T(g["dcn1_cur"]), T(g["dcn1_pre"]), T(g["dcn1_prew"]), T(g["dcn1_flow"]),
                          T(g["dcn1_poff"]), dg=8, repeat=False, interpolate="none")
    close(a, g["dcn1_aligned"], 1e-5)
    close(o, g["dcn1_offfeat"])
    a, o = orc.dcn_module(P, "dcn_0.", T(g["dcn1_cur"]), T(g["dcn1_pre"]), T(g["dcn1_prew"]), T(g["dcn1_flow"]),
                          None, dg=8, repeat=False, interpolate="none")
    close(a, g["dcn0_aligned"], 1e-5)
    close(o, g["dcn0_offfeat"])
    a, o = orc.dcn_module(P, "dcn_3.", T(g["dcn3_cur"]), T(g["dcn3_pre"]), T(g["dcn3_prew"]), T(g["dcn3_flow"]),
                          T(g["dcn3_poff"]), dg=1, repeat=True, interpolate="pixelshuffle")
    close(a, g["dcn3_aligned"], 1e-5)
    close(o, g["dcn3_offfeat"])


def test_metrics(ops_golden):
    g = ops_golden
    p, py = orc.psnr_rgb_and_y(T(g["metric_sr"]), T(g["metric_hr"]))
    assert abs(p - float(g["metric_psnr"])) < 1e-5
    assert abs(py - float(g["metric_psnr_y"])) < 1e-5


@pytest.mark.parametrize("name", ["dsv_16x24_t3", "dsv_20x36_t4", "dsv_16x24_t2_yonly"])
def test_full_forward(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    y_only = bool(g["y_only"])
    sd = synth.make_state_dict(int(g["weights_seed"]), y_only=y_only)
    assert synth.state_dict_digest(sd) == str(g["weights_sha256"])
    lrs, fvs, mks = synth.make_clip(int(g["clip_seed"]), 1, int(g["t"]), int(g["h"]), int(g["w"]),
                                    fv_size=int(g["fv_size"]), sigma_t=10.0)
    assert synth.state_dict_digest({"lrs": lrs, "fvs": fvs, "mks": mks.astype(np.float32)}) == str(g["lrs_sha"])
    P = orc.load_numpy_state(sd)
    close(orc.compute_flow(P, T(lrs)), g["flows"], 1e-4)
    out = orc.crfp_dsv_forward(P, T(lrs), T(fvs), T(mks), orc.DSVConfig(y_only=y_only))
    close(out, g["out"], 2e-5)


def test_streaming_variant_with_regional_mask():
    """model/CRFP_test.py MRCF_simple_v18: one frame per call, fgs mask, clear_states mid-way."""
    g = dict(np.load(os.path.join(GOLDEN, "stream_16x24_t7.npz")))
    sd = synth.make_state_dict(int(g["weights_seed"]))
    t, h, w = int(g["t"]), int(g["h"]), int(g["w"])
    lrs, fvs, mks = synth.make_clip(int(g["clip_seed"]), 1, t, h, w, fv_size=int(g["fv_size"]), sigma_t=10.0)
    so = orc.StreamOracle(orc.load_numpy_state(sd))
    outs = []
    for i in range(t):
        if i == int(g["clear_at"]):
            so.clear_states()
        outs.append(so(T(lrs[:, i:i + 1]), T(fvs[:, i:i + 1]), T(mks[:, i:i + 1]), T(g["fgs"][:, i:i + 1])))
    close(torch.cat(outs, dim=1), g["out"], 2e-5)


def test_masked_psnr_ssim_against_reference(ops_golden):
    """utils.calc_psnr_and_ssim_cuda with the ones / fovea-box / dilated-ring masks of trainer.py:348 and
    test_video.py:340-369, and its [0,255] range branch."""
    g = ops_golden
    sr, hr = T(g["metric_sr"]), T(g["metric_hr"])
    ones = torch.ones(1, 1, *sr.shape[2:])
    for mask, tag in ((ones, ""), (T(g["metric_box"]).float(), "_box"), (T(g["metric_ring"]).float(), "_ring")):
        p, s = orc.calc_psnr_and_ssim(sr, hr, mask)
        assert abs(p - float(g["metric_psnr" + tag])) < 1e-4
        assert abs(s - float(g["metric_ssim" + tag])) < 1e-6
    p, s = orc.calc_psnr_and_ssim(sr * 255.0, hr * 255.0, T(g["metric_box"]).float())
    assert abs(p - float(g["metric_psnr_box255"])) < 1e-4 and abs(s - float(g["metric_ssim_box255"])) < 1e-6
    ys, yh = orc.to_y(sr.permute(0, 2, 3, 1)), orc.to_y(hr.permute(0, 2, 3, 1))
    py, sy = orc.calc_psnr_and_ssim(ys, yh, ones)
    assert abs(py - float(g["metric_psnr_y"])) < 1e-4 and abs(sy - float(g["metric_ssim_y"])) < 1e-6


def test_spynet_oracle_matches_reference_golden():
    """oracle.spynet against the flow the imported reference SPyNet produced
    (tests/golden/make_spynet_golden.py): same ATen ops, so the match is exact; also pins the seeded weight stream."""
    import os
    from conftest import GOLDEN
    from crfp_amd import synth
    from oracle import crfp_oracle as orc
    g = dict(np.load(os.path.join(GOLDEN, "spynet_small.npz")))
    sd = synth.make_spynet_state_dict(int(g["weights_seed"]))
    assert synth.state_dict_digest(sd) == str(g["weights_sha256"])
    P = orc.load_numpy_state(sd)
    for tag in "ab":
        f = orc.spynet(P, torch.from_numpy(g[tag + "_ref"]), torch.from_numpy(g[tag + "_supp"]))
        assert float((f - torch.from_numpy(g[tag + "_flow"])).abs().max()) < 1e-6
    # the mirror holds exactly the reference's state_dict (keys, order, shapes)
    from crfp_amd.model import CRFP
    m = CRFP.SPyNet(None, torch.device("cpu"))
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)


def test_runtime_variant_oracle_matches_reference_golden():
    """oracle.runtime_oracle (the regional-DCN benchmark wiring) against the output of
    the reference class itself, run on the CPU by tests/golden/make_runtime_golden.py; also pins the mirror's state_dict
    table (158 keys incl. the two-input residual blocks and their bottleneck residual branch) and the seeded weights."""
    import os
    from conftest import GOLDEN
    from crfp_amd import synth
    from crfp_amd.model import MRCF_runtime
    from oracle import crfp_oracle as orc, runtime_oracle as ro
    g = dict(np.load(os.path.join(GOLDEN, "runtime_small.npz")))
    m = MRCF_runtime.MRCF_simple_v18(mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True, split_ratio=3,
                                     spynet_pretrained='pretrained_models/fnet.pth', device=torch.device("cpu"))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert list(shapes) == list(g["keys"]) and [str(s) for s in shapes.values()] == list(g["shapes"])
    sd = synth.make_state_dict_like(shapes, int(g["weights_seed"]))
    assert synth.state_dict_digest(sd) == str(g["weights_sha256"])
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    with torch.no_grad():
        out = ro.runtime_forward(orc.load_numpy_state(sd), torch.from_numpy(g["lrs"]), torch.from_numpy(g["fvs"]),
                                 tuple(int(v) for v in g["warp"]))
    assert float((out - torch.from_numpy(g["out"])).abs().max()) < 1e-6
