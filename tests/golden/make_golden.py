#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference's Python (model/CRFP.py, model/LTE.py, utils.py) is imported from
/root/reference with three modules injected into ``sys.modules`` because they are not
installed and cannot be (no network):
  * ``dcn_v2``  - third-party CUDA op, un-vendored and un-pinned (reference README.md:26).  The
    stand-in ``DCNv2`` module owns ``weight``/``bias`` like the real one and evaluates
    ``oracle.crfp_oracle.dcnv2`` -> every vector that passes through DCNv2 pins the reference's
    *wiring* (offset/mask construction, flip, repeat, concat order) but not DCNv2's arithmetic,
    which the reference itself never pins ("parity unpinned", see oracle header + DESIGN.md).
  * ``cv2``     - only imported by utils.py, unused on the metric path.
Weights come from ``crfp_amd.synth.make_state_dict(seed)`` (numpy RandomState, bit-stable), are
loaded into the reference modules with ``strict=True`` (which pins the state_dict key/shape
table) and are NOT stored: fixtures keep the seed and a sha256 digest.
Only data (inputs + the reference's outputs) is written; no reference source is copied.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from crfp_amd import synth  # noqa: E402
from oracle import crfp_oracle as orc  # noqa: E402


def inject_stubs():
    class DCNv2(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1,
                     deformable_groups=1):
            super().__init__()
            assert stride == 1
            self.k, self.pad, self.dil, self.dg = kernel_size, padding, dilation, deformable_groups
            self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, kernel_size, kernel_size))
            self.bias = nn.Parameter(torch.zeros(out_channels))

        def forward(self, x, offset, mask):
            return orc.dcnv2(x, offset, mask, self.weight, self.bias, self.dg, self.k, self.pad, self.dil)

    m = types.ModuleType("dcn_v2")
    m.DCNv2 = DCNv2
    sys.modules["dcn_v2"] = m
    sys.modules["cv2"] = types.ModuleType("cv2")


def sub_state(sd, prefix):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in sd.items() if k.startswith(prefix)}


def main():
    assert os.path.isdir(REF), "reference not mounted: run this in the build container"
    inject_stubs()
    sys.path.insert(0, REF)
    from model import CRFP, LTE  # the reference
    import utils as ref_utils
    torch.manual_seed(0)
    torch.set_grad_enabled(False)

    SEED = 7
    sd = synth.make_state_dict(SEED)
    digest = synth.state_dict_digest(sd)
    rs = np.random.RandomState(123)
    f32 = lambda *s: rs.standard_normal(s).astype(np.float32)  # noqa: E731
    T = torch.from_numpy
    out = {"weights_seed": np.int64(SEED), "weights_sha256": np.array(digest)}

    # ---- flow_warp (model/CRFP.py:90-130): zeros and border padding, flows that leave the image
    x = f32(2, 5, 12, 17)
    fl = (rs.uniform(-4, 4, (2, 12, 17, 2))).astype(np.float32)
    fl[0, 0, 0] = (-30.0, 2.5)
    fl[1, 5, 5] = (40.0, -40.0)
    fl[0, 3, 4] = (0.0, 0.0)
    out["warp_x"], out["warp_flow"] = x, fl
    out["warp_zeros"] = CRFP.flow_warp(T(x), T(fl)).numpy()
    out["warp_border"] = CRFP.flow_warp(T(x), T(fl), padding_mode="border").numpy()

    # ---- pixel_unshuffle one-hot grouped conv (model/CRFP.py:28-42)
    x = f32(1, 4, 16, 24)
    out["unshuffle_x"] = x
    out["unshuffle_y"] = CRFP.pixel_unshuffle(T(x), 4).numpy()

    # ---- PixelShufflePack 32->24 r2 (model/CRFP.py:154-193; instance :1446)
    m = CRFP.PixelShufflePack(32, 24, 2, upsample_kernel=3)
    m.load_state_dict(sub_state(sd, "upsample."), strict=True)
    x = f32(1, 32, 9, 13)
    out["psp_x"], out["psp_y"] = x, m(T(x)).numpy()

    # ---- PixelUnShufflePack_v2 4->32 r4 (model/CRFP.py:239-279; instance :1438)
    m = CRFP.PixelUnShufflePack_v2(4, 32, 4, downsample_kernel=3)
    m.load_state_dict(sub_state(sd, "downsample."), strict=True)
    x = f32(1, 4, 24, 32)
    out["pusp_x"], out["pusp_y"] = x, m(T(x)).numpy()

    # ---- ResidualBlocksWithInputConv 64->32 (model/CRFP.py:516-552; instance :1424)
    m = CRFP.ResidualBlocksWithInputConv(64, 32, 1)
    m.load_state_dict(sub_state(sd, "forward_resblocks_1."), strict=True)
    x = f32(1, 64, 10, 14)
    out["res_x"], out["res_y"] = x, m(T(x)).numpy()

    # ---- LTE encoders (model/LTE.py:34-51,100-117)
    m = LTE.LTE_simple_lr(32)
    m.load_state_dict(sub_state(sd, "encoder_lr."), strict=True)
    x = rs.uniform(0, 1, (2, 3, 11, 15)).astype(np.float32)
    out["enc_lr_x"], out["enc_lr_y"] = x, m(T(x), islr=True)[2].numpy()
    m = LTE.LTE_simple_hr_single(4)
    m.load_state_dict(sub_state(sd, "encoder_hr."), strict=True)
    x = rs.uniform(0, 1, (1, 6, 16, 24)).astype(np.float32)
    out["enc_hr_x"], out["enc_hr_y"] = x, m(T(x), islr=True)[2].numpy()

    # ---- FNet (model/CRFP.py:743-814): a /8 size and a size that exercises the tail resize
    m = CRFP.FNet(3)
    m.load_state_dict(sub_state(sd, "spynet."), strict=True)
    for tag, (h, w) in (("a", (24, 40)), ("b", (20, 36))):
        x1 = rs.uniform(0, 1, (2, 3, h, w)).astype(np.float32)
        x2 = np.clip(np.roll(x1, (1, -1), axis=(2, 3)) + rs.uniform(-0.03, 0.03, x1.shape), 0, 1).astype(np.float32)
        out[f"fnet_{tag}_x1"], out[f"fnet_{tag}_x2"] = x1, x2
        out[f"fnet_{tag}_y"] = m(T(x1), T(x2)).numpy()

    # ---- DCN_module, normal mode with pre_offset (dcn_1 wiring, model/CRFP.py:1410)
    m = CRFP.DCN_module(32, 8, 3, 10, pre_offset=True, interpolate="none")
    m.load_state_dict(sub_state(sd, "dcn_1."), strict=True)
    H, W = 12, 16
    cur, pre, prew, poff = f32(1, 32, H, W), f32(1, 32, H, W), f32(1, 32, H, W), f32(1, 32, H, W)
    flow = rs.uniform(-3, 3, (1, 2, H, W)).astype(np.float32)
    a, o = m(T(cur), T(pre), T(prew), T(flow), T(poff))
    out.update(dcn1_cur=cur, dcn1_pre=pre, dcn1_prew=prew, dcn1_flow=flow, dcn1_poff=poff,
               dcn1_aligned=a.numpy(), dcn1_offfeat=o.numpy())
    # ---- DCN_module, first level (no pre_offset) (dcn_0 wiring, :1409)
    m = CRFP.DCN_module(32, 8, 3, 10)
    m.load_state_dict(sub_state(sd, "dcn_0."), strict=True)
    a, o = m(T(cur), T(pre), T(prew), T(flow))
    out.update(dcn0_aligned=a.numpy(), dcn0_offfeat=o.numpy())
    # ---- DCN_module, repeat mode + pixelshuffle pre_offset (dcn_3 wiring, :1413)
    m = CRFP.DCN_module(4, 1, 3, 10, repeat=True, pre_offset=True, interpolate="pixelshuffle")
    m.load_state_dict(sub_state(sd, "dcn_3."), strict=True)
    H8, W8 = 24, 32
    cur, pre, prew = f32(1, 4, H8, W8), f32(1, 4, H8, W8), f32(1, 4, H8, W8)
    flow = rs.uniform(-6, 6, (1, 2, H8, W8)).astype(np.float32)
    poff = f32(1, 32, H8 // 4, W8 // 4)
    a, o = m(T(cur), T(pre), T(prew), T(flow), T(poff))
    out.update(dcn3_cur=cur, dcn3_pre=pre, dcn3_prew=prew, dcn3_flow=flow, dcn3_poff=poff,
               dcn3_aligned=a.numpy(), dcn3_offfeat=o.numpy())
    # ---- known-answer: identity-initialised DCN_module == 0.5 * flow_warp (:354-370)
    for tag, m in (("n", CRFP.DCN_module(32, 8, 3, 10)), ("r", CRFP.DCN_module(4, 1, 3, 10, repeat=True))):
        C = 32 if tag == "n" else 4
        cur, pre, prew = f32(1, C, 10, 12), f32(1, C, 10, 12), f32(1, C, 10, 12)
        flow = rs.uniform(-3, 3, (1, 2, 10, 12)).astype(np.float32)
        a, _ = m(T(cur), T(pre), T(prew), T(flow))
        ref = 0.5 * CRFP.flow_warp(T(pre), T(flow).permute(0, 2, 3, 1))
        out[f"kat_{tag}_pre"], out[f"kat_{tag}_flow"] = pre, flow
        out[f"kat_{tag}_dcn"], out[f"kat_{tag}_halfwarp"] = a.numpy(), ref.numpy()

    # ---- metrics (utils.py:166-185,242-254,328-330)
    sr = rs.uniform(0, 1, (1, 3, 32, 48)).astype(np.float32)
    hr = np.clip(sr + rs.normal(0, 0.03, sr.shape), 0, 1).astype(np.float32)
    ones = torch.ones(1, 1, 32, 48)
    p, s = ref_utils.calc_psnr_and_ssim_cuda(T(sr), T(hr), ones)
    py, sy = ref_utils.calc_psnr_and_ssim_cuda(ref_utils.bgr2ycbcr(T(sr).permute(0, 2, 3, 1), y_only=True),
                                               ref_utils.bgr2ycbcr(T(hr).permute(0, 2, 3, 1), y_only=True), ones)
    out.update(metric_sr=sr, metric_hr=hr, metric_psnr=np.float64(float(p)), metric_ssim=np.float64(float(s)),
               metric_psnr_y=np.float64(float(py)), metric_ssim_y=np.float64(float(sy)))
    # region masks as test_video.py:340-369 passes them (fovea box, dilated ring), and the [0,255] range branch
    box = torch.zeros(1, 1, 32, 48); box[:, :, 6:22, 10:30] = 1
    ring = box.clone()
    for _ in range(3):
        ring = torch.clamp(torch.nn.functional.conv2d(ring, torch.ones(1, 1, 3, 3), padding=1), 0, 1)
    ring = torch.logical_and(torch.logical_not(box.bool()), ring.bool())
    pb, sb = ref_utils.calc_psnr_and_ssim_cuda(T(sr), T(hr), box)
    pr, sr_ = ref_utils.calc_psnr_and_ssim_cuda(T(sr), T(hr), ring)
    p255, s255 = ref_utils.calc_psnr_and_ssim_cuda(T(sr) * 255.0, T(hr) * 255.0, box)
    out.update(metric_box=box.numpy().astype(np.uint8), metric_ring=ring.numpy().astype(np.uint8),
               metric_psnr_box=np.float64(float(pb)), metric_ssim_box=np.float64(float(sb)),
               metric_psnr_ring=np.float64(float(pr)), metric_ssim_ring=np.float64(float(sr_)),
               metric_psnr_box255=np.float64(float(p255)), metric_ssim_box255=np.float64(float(s255)))
    np.savez_compressed(os.path.join(HERE, "ops_small.npz"), **out)
    print("ops_small.npz:", len(out), "arrays")

    # ---- full CRFP_DSV.forward (model/CRFP.py:1510-1686) through the model factory call of main.py:34
    tmp = tempfile.mkdtemp()
    fnet_path = os.path.join(tmp, "fnet.pth")
    torch.save(sub_state(sd, "spynet."), fnet_path)
    for name, (h, w, t, y_only, fv) in {"dsv_16x24_t3": (16, 24, 3, False, 48),
                                        "dsv_20x36_t4": (20, 36, 4, False, 64),
                                        "dsv_16x24_t2_yonly": (16, 24, 2, True, 48)}.items():
        sdm = synth.make_state_dict(SEED, y_only=y_only)
        model = CRFP.CRFP_DSV(mid_channels=32, y_only=y_only, hr_dcn=True, offset_prop=True,
                              spynet_pretrained=fnet_path, device=torch.device("cpu"))
        assert list(model.state_dict().keys()) == synth.state_dict_keys(y_only=y_only), "state_dict key table drifted"
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sdm.items()}, strict=True)
        model.eval()
        lrs, fvs, mks = synth.make_clip(seed=11, n=1, t=t, h=h, w=w, fv_size=fv, sigma_t=10.0)
        y = model(lrs=T(lrs), fvs=T(fvs), mks=T(mks))
        flows = model.compute_flow(T(lrs))[0]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), weights_seed=np.int64(SEED),
                            weights_sha256=np.array(synth.state_dict_digest(sdm)), y_only=np.bool_(y_only),
                            clip_seed=np.int64(11), fv_size=np.int64(fv), h=np.int64(h), w=np.int64(w),
                            t=np.int64(t), out=y.numpy(), flows=flows.numpy(),
                            lrs_sha=np.array(synth.state_dict_digest({"lrs": lrs, "fvs": fvs, "mks": mks.astype(np.float32)})))
        print(name, "out", tuple(y.shape), "mean %.6f" % float(y.mean()), "flow absmax %.3f" % float(flows.abs().max()))


    # ---- streaming variant (model/CRFP_test.py:2114-2478): 5 calls of one frame each with a regional
    #      mask fgs, then clear_states() and 2 more calls
    from model import CRFP_test
    sdm = synth.make_state_dict(SEED)
    sm = CRFP_test.MRCF_simple_v18(mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True, split_ratio=3,
                                   spynet_pretrained=fnet_path, device=torch.device("cpu"))
    assert list(sm.state_dict().keys()) == synth.state_dict_keys()
    sm.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sdm.items()}, strict=True)
    sm.eval()
    h, w, t = 16, 24, 7
    lrs, fvs, mks = synth.make_clip(seed=13, n=1, t=t, h=h, w=w, fv_size=48, sigma_t=10.0)
    rs2 = np.random.RandomState(99)
    fgs = np.zeros((1, t, 1, 8 * h, 8 * w), np.bool_)
    for i in range(t):
        y0, x0 = rs2.randint(0, 8 * h - 70), rs2.randint(0, 8 * w - 90)
        fgs[0, i, 0, y0:y0 + 70, x0:x0 + 90] = True
    outs = []
    for i in range(t):
        if i == 5:
            sm.clear_states()
        outs.append(sm(T(lrs[:, i:i + 1]), T(fvs[:, i:i + 1]), T(mks[:, i:i + 1]), T(fgs[:, i:i + 1])).numpy())
    np.savez_compressed(os.path.join(HERE, "stream_16x24_t7.npz"), weights_seed=np.int64(SEED), clip_seed=np.int64(13),
                        h=np.int64(h), w=np.int64(w), t=np.int64(t), fv_size=np.int64(48), clear_at=np.int64(5),
                        fgs=fgs, out=np.concatenate(outs, axis=1))
    print("stream_16x24_t7 out", np.concatenate(outs, axis=1).shape)


if __name__ == "__main__":
    main()
