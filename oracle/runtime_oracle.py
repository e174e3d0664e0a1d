"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU fp32 restatement of the reference's benchmark-only wiring
``model/CRFP_runtime.py::MRCF_simple_v18.forward(lrs, fvs, warp_size)`` (:8469-8664) with
``ResidualBlocksWithInputConv_v2`` (:511-556) and that file's ``DCN_module`` (:135-219, same arithmetic as
model/CRFP.py:324-352).  Functional over a flat ``{key: tensor}`` state dict; building blocks come from
``oracle.crfp_oracle``.

Pinning: ``tests/golden/runtime_small.npz`` holds the output of the REFERENCE class itself, run on the CPU in the build
container (``tests/golden/make_runtime_golden.py``: the module's import-time ``.to('cuda:0')`` of its pixel grid, its
``torch.cuda`` timers and the absent ``memory_profiler`` / ``dcn_v2`` packages are neutralised there); DCNv2 inside it is
the oracle's own, so the golden pins the wiring, like every other golden that passes through DCNv2."""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from . import crfp_oracle as o

Tensor = torch.Tensor


def rb_v2(P: Dict[str, Tensor], pre: str, feat1: Tensor, feat2: Tensor = None) -> Tensor:
    """ResidualBlocksWithInputConv_v2.forward (:541-556): conv1(feat1) pasted over conv2(feat2), LeakyReLU, one residual block."""
    if feat2 is not None:
        H, W = feat1.shape[-2:]
        feat = o.conv(P, pre + "conv2", feat2).clone()
        feat[:, :, :H, :W] = o.conv(P, pre + "conv1", feat1)
    else:
        feat = o.conv(P, pre + "conv1", feat1)
    x = o.lrelu(feat)
    return x + o.conv(P, pre + "main.1.0.conv2", F.relu(o.conv(P, pre + "main.1.0.conv1", x)))


def runtime_forward(P: Dict[str, Tensor], lrs: Tensor, fvs: Tensor, warp_size: Tuple[int, int], y_only: bool = False) -> Tensor:
    """MRCF_simple_v18.forward (:8469-8664), split_ratio 3, offset_prop True."""
    WP_h, WP_w = warp_size
    n, t, c, h, w = lrs.shape
    flows = o.compute_flow(P, lrs[:, :, :, :WP_h // 8, :WP_w // 8]) if t > 1 else None          # :8487
    lr2 = lrs.reshape(n * t, c, h, w)
    x_lr = o.lrelu(o.conv(P, "encoder_lr.slice1.2", o.lrelu(o.conv(P, "encoder_lr.slice1.0", lr2)))).view(n, t, -1, h, w)
    Hf, Wf = fvs.shape[-2:]
    fv2 = fvs.reshape(n * t, 3, Hf, Wf)
    x_hr = torch.cat((fv2, fv2), 1)                                                             # :8507
    x_hr = o.lrelu(o.conv(P, "encoder_hr.slice1.2", o.lrelu(o.conv(P, "encoder_hr.slice1.0", x_hr)))).view(n, t, -1, Hf, Wf)
    outs = []
    state = feat_lv = None
    for i in range(t):
        prop0 = o.pixel_shuffle_pack(P, "upsample.", x_lr[:, i], 2)                             # :8524
        if i > 0:
            flow = flows[:, i - 1]
            flow_lv3 = o.up_bilinear(flow, 2) * 2.0                                             # :8531
            flow_lv0 = o.up_bilinear(flow, 8) * 8.0
            state_w = o.flow_warp(state, flow_lv0.permute(0, 2, 3, 1))                          # :8534-8535
            prev2_w = o.pixel_unshuffle_pack_v2(P, "downsample.", state_w, 4)                   # :8536
            prev2 = o.pixel_unshuffle_pack_v2(P, "downsample.", state, 4)                       # :8537
            mix = torch.chunk(o.flow_warp(torch.cat(feat_lv, 1), flow_lv3.permute(0, 2, 3, 1)), 3, dim=1)   # :8538-8547
            feat_lv = list(mix)
            win = prop0[:, :, :WP_h // 4, :WP_w // 4]
            off = None
            for k in range(3):                                                                  # :8549-8599
                feat_temp = torch.cat((win, feat_lv[k]), 1)
                aligned, off = o.dcn_module(P, f"dcn_{k}.", feat_temp, prev2, prev2_w, flow_lv3, off,
                                            dg=8, repeat=False, interpolate="none")
                y = rb_v2(P, f"forward_resblocks_{k}.", torch.cat([feat_temp, aligned], 1), feat_temp)
                feat_lv[k] = y[:, 24:][:, :, :WP_h // 4, :WP_w // 4]
            up = o.lrelu(o.pixel_shuffle_pack(P, "upsample_post.", prop0, 4))                   # :8602
            upw = up[:, :, :WP_h, :WP_w]
            aligned, _ = o.dcn_module(P, "dcn_3.", upw, state, state_w, flow_lv0, off,
                                      dg=1, repeat=True, interpolate="pixelshuffle")            # :8603-8606
            feat = rb_v2(P, "forward_resblocks_3.", torch.cat([upw, aligned], 1), up)          # :8607-8609
        else:
            feat_lv = []
            for k in range(3):                                                                  # :8617-8633
                y = rb_v2(P, f"forward_resblocks_{k}_.", prop0)            # same two-input class, feat2 = None (:464-509)
                feat_lv.append(y[:, 24:][:, :, :WP_h // 4, :WP_w // 4])
                prop0 = y[:, :24]
            up = o.lrelu(o.pixel_shuffle_pack(P, "upsample_post.", prop0, 4))                   # :8636
            feat = rb_v2(P, "forward_resblocks_3_.", up)                                        # :8637
        fused = o.conv(P, "conv_tttf", torch.cat([feat[:, :, :Hf, :Wf], x_hr[:, i]], 1))        # :8645-8647
        feat = feat.clone()
        feat[:, :, :Hf, :Wf] = fused
        feat = o.lrelu(feat)                                                                    # :8648
        base = o.up_bilinear(o.rgb_to_y(lrs[:, i]), 8) if y_only else o.up_bilinear(lrs[:, i], 8)
        outs.append(o.conv(P, "conv_last", feat) + base)                                        # :8652-8654
        state = feat[:, :, :WP_h, :WP_w]                                                        # :8650
    return torch.stack(outs, 1)
