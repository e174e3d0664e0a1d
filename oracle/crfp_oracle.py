"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU fp32 restatement (torch-CPU ATen ops + our own DCNv2) of the reference's CRFP_DSV
recurrent inference path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; ``crfp_amd`` never does.

Written functionally over a flat ``{key: tensor}`` state dict with an explicit per-frame
state, so it shares no structure with the reference's nn.Module code; every function
cites the reference lines it restates (paths relative to /root/reference).

Pinning status
  * everything except DCNv2: pinned by ``tests/golden/*.npz`` generated from the imported
    reference (``tests/golden/make_golden.py``), max|delta| == 0 expected (same ATen ops).
  * DCNv2: the reference imports it from the un-vendored, un-pinned third-party package
    ``dcn_v2`` (README.md:26, github.com/jinfagang/DCNv2_latest) -> **parity unpinned** by
    the reference itself.  ``dcnv2`` below restates the published DCNv2 algorithm
    (modulated deformable im2col + GEMM); it is cross-checked against the independent C
    restatement ``oracle/dcnv2_ref.c`` and anchored on the reference's own call sites via
    the known-answer test "identity-initialised DCN_module == 0.5 * flow_warp"
    (model/CRFP.py:354-370 with :90-130) and conv / shifted-conv / mask-linearity properties.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------- bf16-storage twin
# BASELINE configs 3-5 run the path with bf16 activation storage.  The reference has no bf16 mode of its own (it would be
# ``model.bfloat16()`` / autocast, whose rounding points are an implementation detail of ATen), so the twin is defined by
# the build's storage contract (include/crfp_hip.h, "bf16 storage"): every activation tensor that is
# written to HBM and the recurrent state are rounded to bf16 (round-to-nearest-even) at the point where the HIP engine
# stores them; conv / DCN weights are bf16 values; accumulation, interpolation, activations, biases, the API tensors and
# everything that is a coordinate (flow, DCN offsets, masks) stay fp32.  ``R`` marks those store points below and is the
# identity unless ``bf16_storage()`` is active, so the fp32 oracle is untouched.
import contextlib  # noqa: E402

_BF16 = [False]


def R(x: Tensor) -> Tensor:
    """Storage rounding of an activation tensor (identity in the fp32 oracle)."""
    return x.to(torch.bfloat16).to(torch.float32) if _BF16[0] else x


@contextlib.contextmanager
def bf16_storage():
    _BF16[0] = True
    try:
        yield
    finally:
        _BF16[0] = False


_TAP = [None]


@contextlib.contextmanager
def tapping(store: dict):
    """Record named intermediates of dsv_frame (names = workspace buffers of the engine, crfp_dsv_debug_fetch) for bisecting."""
    _TAP[0] = store
    try:
        yield store
    finally:
        _TAP[0] = None


def tap(name: str, x: Tensor) -> Tensor:
    if _TAP[0] is not None:
        _TAP[0][name] = x.detach().clone()
    return x


def bf16_weights(P: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """The twin's parameters: every conv / DCN weight rounded to bf16, biases untouched."""
    return {k: (v.to(torch.bfloat16).to(torch.float32) if v.dim() == 4 else v) for k, v in P.items()}


# ----------------------------------------------------------------------------- primitives
def conv(P: Dict[str, Tensor], stem: str, x: Tensor) -> Tensor:
    """3x3 stride-1 pad-1 conv with bias (every conv on the path; e.g. model/CRFP.py:48-50)."""
    return F.conv2d(x, P[stem + ".weight"], P[stem + ".bias"], stride=1, padding=1)


def lrelu(x: Tensor) -> Tensor:
    """LeakyReLU(0.1) (model/CRFP.py:305,321,533,1481; model/LTE.py:41,107)."""
    return F.leaky_relu(x, 0.1)


def up_bilinear(x: Tensor, r: float) -> Tensor:
    """nn.Upsample(scale_factor=r, 'bilinear', align_corners=False) (model/CRFP.py:1471-1478,776)."""
    return F.interpolate(x, scale_factor=r, mode="bilinear", align_corners=False)


def flow_warp(x: Tensor, flow: Tensor, padding_mode: str = "zeros") -> Tensor:
    """model/CRFP.py:90-130.  x [n,c,h,w]; flow [n,h,w,2] (dx, dy) in pixels.
    Sample position = pixel index + flow, routed through the [-1,1] normalisation exactly
    as the reference does (same float32 operation order), then grid_sample(align_corners=True)."""
    n, c, h, w = x.shape
    if tuple(flow.shape[1:3]) != (h, w):
        raise ValueError(f"The spatial sizes of input ({(h, w)}) and flow ({tuple(flow.shape[1:3])}) are not the same.")
    gy, gx = torch.meshgrid(torch.arange(0, h), torch.arange(0, w), indexing="ij")
    base = torch.stack((gx, gy), 2).to(x.dtype)
    g = base + flow
    nx = 2.0 * g[..., 0] / max(w - 1, 1) - 1.0
    ny = 2.0 * g[..., 1] / max(h - 1, 1) - 1.0
    return F.grid_sample(x, torch.stack((nx, ny), dim=3), mode="bilinear",
                         padding_mode=padding_mode, align_corners=True)


def dcnv2(x: Tensor, offset: Tensor, mask: Tensor, weight: Tensor, bias: Tensor,
          deformable_groups: int, kernel_size: int = 3, padding: int = 1, dilation: int = 1) -> Tensor:
    """Modulated deformable convolution v2, stride 1 (third-party dcn_v2.DCNv2.forward; call site
    model/CRFP.py:350, ctor :318-320).  Published algorithm (DCNv2 ``modulated_deformable_im2col``):
      for tap k=(ky,kx), group g, channel c in g, output pixel (y,x):
        py = y - pad + ky*dil + offset[2*(g*K+k)  ];  px = x - pad + kx*dil + offset[2*(g*K+k)+1]
        v  = bilinear(x[c], py, px) with each out-of-range corner contributing 0
             (and v = 0 unless -1 < py < H and -1 < px < W)
        col[c,k] = v * mask[g*K+k]
      out[o] = bias[o] + sum_{c,k} weight[o,c,ky,kx] * col[c,k]
    """
    B, C, H, W = x.shape
    k = kernel_size
    K = k * k
    dg = deformable_groups
    cpg = C // dg
    O = weight.shape[0]
    assert offset.shape == (B, 2 * dg * K, H, W), (offset.shape, (B, 2 * dg * K, H, W))
    assert mask.shape == (B, dg * K, H, W), mask.shape
    assert weight.shape == (O, C, k, k)
    off = offset.reshape(B, dg, K, 2, H, W)
    msk = mask.reshape(B, dg, K, H, W)
    ys = torch.arange(H, dtype=x.dtype).view(1, 1, H, 1)
    xs = torch.arange(W, dtype=x.dtype).view(1, 1, 1, W)
    xf = x.reshape(B, dg, cpg, H * W)
    wk = weight.reshape(O, C, K)
    out = bias.view(1, O, 1).repeat(B, 1, H * W).to(x.dtype)
    for t in range(K):
        ky, kx = divmod(t, k)
        py = ys + float(ky * dilation - padding) + off[:, :, t, 0]      # [B,dg,H,W]
        px = xs + float(kx * dilation - padding) + off[:, :, t, 1]
        y0 = torch.floor(py)
        x0 = torch.floor(px)
        ly = py - y0
        lx = px - x0
        hy = 1.0 - ly
        hx = 1.0 - lx
        inside = (py > -1) & (px > -1) & (py < H) & (px < W)
        val = torch.zeros(B, dg, cpg, H * W, dtype=x.dtype)
        for dy, dx, wgt in ((0, 0, hy * hx), (0, 1, hy * lx), (1, 0, ly * hx), (1, 1, ly * lx)):
            yy = y0 + dy
            xx = x0 + dx
            ok = (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1) & inside
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long().reshape(B, dg, 1, H * W)
            g = torch.gather(xf, 3, idx.expand(B, dg, cpg, H * W))
            val += g * (wgt * ok.to(x.dtype)).reshape(B, dg, 1, H * W)
        col = (val * msk[:, :, t].reshape(B, dg, 1, H * W)).reshape(B, C, H * W)
        out += torch.matmul(wk[:, :, t], col)
    return out.reshape(B, O, H, W)


# ----------------------------------------------------------------------------- sub-networks
def fnet(P, pre: str, x1: Tensor, x2: Tensor) -> Tensor:
    """FNet.forward (model/CRFP.py:797-814; layers :747-795): flow from x1 to x2, [n,2,h,w]."""
    _, _, h, w = x1.shape
    o = torch.cat([x1, x2], dim=1)
    for blk in ("encoder1", "encoder2", "encoder3"):
        o = R(F.relu(conv(P, f"{pre}{blk}.0", o)))
        o = R(F.relu(conv(P, f"{pre}{blk}.2", o)))
        o = R(F.avg_pool2d(o, 2, 2))
    for blk in ("decoder1", "decoder2", "decoder3"):
        o = R(F.relu(conv(P, f"{pre}{blk}.0", o)))
        o = R(F.relu(conv(P, f"{pre}{blk}.2", o)))
        o = R(up_bilinear(o, 2))
    o = conv(P, pre + "flow.2", R(F.relu(conv(P, pre + "flow.0", o))))     # the flow itself stays fp32
    o = torch.tanh(o) * 256
    return F.interpolate(o, size=(h, w), mode="bilinear", align_corners=False)


def spynet(P, ref: Tensor, supp: Tensor, pre: str = "") -> Tensor:
    """SPyNet.forward(ref, supp) (model/CRFP.py:698-741) with compute_flow (:593-664), SPyNetBasicModule (:686-741) and the
    ``conv`` module whose ReLU precedes the convolution (:145-152).  P: ``basic_module.{L}.basic_module.{j}.conv.{weight,bias}``."""
    n, _, h, w = ref.shape
    w_up = w if w % 32 == 0 else 32 * (w // 32 + 1)                                         # :716-718
    h_up = h if h % 32 == 0 else 32 * (h // 32 + 1)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)                             # :586-591
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    pyr = []
    for x in (ref, supp):
        x = F.interpolate(x, size=(h_up, w_up), mode="bilinear", align_corners=False)       # :719-726
        levels = [(x - mean) / std]                                                         # :620-621
        for _ in range(5):
            levels.append(F.avg_pool2d(levels[-1], 2, 2, count_include_pad=False))          # :624-636
        pyr.append(levels[::-1])
    flow = ref.new_zeros(n, 2, h_up // 32, w_up // 32)                                      # :641
    for level in range(6):
        flow_up = flow if level == 0 else F.interpolate(flow, scale_factor=2, mode="bilinear", align_corners=True) * 2.0  # :643-652
        warped = flow_warp(pyr[1][level], flow_up.permute(0, 2, 3, 1), padding_mode="border")   # :655-657
        x = torch.cat([pyr[0][level], warped, flow_up], 1)                                  # :658
        for j in range(5):                                                                  # :693-734, :152
            k = f"{pre}basic_module.{level}.basic_module.{j}.conv."
            x = F.conv2d(F.relu(x), P[k + "weight"], P[k + "bias"], stride=1, padding=3)
        flow = flow_up + x                                                                  # :660
    flow = F.interpolate(flow, size=(h, w), mode="bilinear", align_corners=False)           # :728-733
    flow = flow.clone()
    flow[:, 0] *= float(w) / float(w_up)                                                    # :736-737
    flow[:, 1] *= float(h) / float(h_up)
    return flow


def compute_flow(P, lrs: Tensor) -> Tensor:
    """CRFP_DSV.compute_flow (model/CRFP.py:1483-1508): flows_forward[n,t-1,2,h,w] =
    FNet(frame i, frame i-1) for i = 1..t-1 (FNet input order [current | previous])."""
    n, t, c, h, w = lrs.shape
    prev = lrs[:, :-1].reshape(-1, c, h, w)
    cur = lrs[:, 1:].reshape(-1, c, h, w)
    return fnet(P, "spynet.", cur, prev).view(n, t - 1, 2, h, w)


def pixel_shuffle_pack(P, pre: str, x: Tensor, r: int) -> Tensor:
    """PixelShufflePack.forward (model/CRFP.py:184-193)."""
    return F.pixel_shuffle(conv(P, pre + "upsample_conv", x), r)


def pixel_unshuffle_pack_v2(P, pre: str, x: Tensor, r: int) -> Tensor:
    """PixelUnShufflePack_v2.forward (model/CRFP.py:270-279); the reference's one-hot grouped-conv
    unshuffle (:28-42) equals F.pixel_unshuffle (checked in tests against the golden vector)."""
    return conv(P, pre + "downsample_conv", F.pixel_unshuffle(x, r))


def resblocks_with_input_conv(P, pre: str, x: Tensor) -> Tensor:
    """ResidualBlocksWithInputConv(in,out,1).forward (model/CRFP.py:516-552 + :433-481):
    conv -> LReLU(0.1) -> one ResidualBlockNoBN (x + conv2(ReLU(conv1(x))), res_scale 1)."""
    x = R(lrelu(conv(P, pre + "main.0", x)))
    return R(x + conv(P, pre + "main.2.0.conv2", R(F.relu(conv(P, pre + "main.2.0.conv1", x)))))


def dcn_module(P, pre: str, cur: Tensor, prev: Tensor, prev_warped: Tensor, flow: Tensor,
               pre_offset: Optional[Tensor], *, dg: int, repeat: bool, interpolate: str,
               max_mag: float = 10.0, flow_is_mfma_operand: bool = True):
    """DCN_module.forward (model/CRFP.py:324-352).  Returns (aligned, offset_feature).
    bf16 twin: the flow channels of dcn_block.0's input are an operand of the bf16 MFMA conv at 2x resolution (rounded on
    the way into LDS); the 8x-resolution dcn_3 runs fp32 stencils on the float flow (flow_is_mfma_operand=False)."""
    f = torch.cat([cur, prev_warped, R(flow) if flow_is_mfma_operand else flow], dim=1)
    f = tap(pre + "block0", R(lrelu(conv(P, pre + "dcn_block.0", f))))
    f = tap(pre + "block2", R(lrelu(conv(P, pre + "dcn_block.2", f))))
    if pre_offset is not None:
        if interpolate == "pixelshuffle":
            pre_offset = R(pixel_shuffle_pack(P, pre + "upsample.", pre_offset, 4) * 2.0)
        elif interpolate == "bilinear":
            pre_offset = R(up_bilinear(pre_offset, 4) * 2.0)
        tap(pre + "pre_offset", pre_offset)
        f = tap(pre + "fuse", R(lrelu(conv(P, pre + "conv_fuse", torch.cat([f, pre_offset], dim=1)))))
    offset = max_mag * torch.tanh(conv(P, pre + "dcn_offset", f))
    mask = torch.sigmoid(conv(P, pre + "dcn_mask", f))
    flow_yx = flow.flip(1)
    if repeat:
        B, C2, H, W = offset.shape
        offset = offset.view(B, 2, C2 // 2, H, W) + flow_yx.unsqueeze(2)
        offset = offset.repeat(1, 9, 1, 1, 1).view(B, C2 * 9, H, W)
        mask = mask.repeat(1, 9, 1, 1)
    else:
        offset = offset + flow_yx.repeat(1, offset.shape[1] // 2, 1, 1)
    tap(pre + "offset", offset)
    tap(pre + "mask", mask)
    out = tap(pre + "aligned", R(dcnv2(prev, offset, mask, P[pre + "dcn.weight"], P[pre + "dcn.bias"], dg)))
    return out, f


# ----------------------------------------------------------------------------- the recurrent path
class DSVConfig:
    """Mirrors the ctor arguments of CRFP_DSV (model/CRFP.py:1388-1402)."""

    def __init__(self, mid_channels: int = 32, y_only: bool = False, hr_dcn: bool = True,
                 offset_prop: bool = True):
        if not (hr_dcn and offset_prop):
            raise NotImplementedError("oracle covers the eval.sh configuration (hr_dcn, offset_prop)")
        self.mid = mid_channels
        self.last = mid_channels // 8
        self.dg = 8
        self.split_ratio = 3
        self.y_only = y_only
        self.carry = (mid_channels * (4 - self.split_ratio)) // 4
        self.prop = mid_channels - self.carry


def new_state(cfg: DSVConfig, n: int, h: int, w: int, like: Tensor):
    """Zero recurrent state (model/CRFP.py:1529-1534): 8x feature + three 2x carry features."""
    return {"hr": like.new_zeros(n, cfg.last, 8 * h, 8 * w),
            "carry": [like.new_zeros(n, cfg.carry, 2 * h, 2 * w) for _ in range(3)],
            "first": True}


def rgb_to_y(rgb: Tensor) -> Tensor:
    """rgb2yuv (model/CRFP.py:12-26): luma only."""
    return (0.299 * rgb[:, 0] + 0.587 * rgb[:, 1] + 0.114 * rgb[:, 2]).unsqueeze(1)


def dsv_frame(P, cfg: DSVConfig, st, lr: Tensor, fv: Tensor, mk: Tensor, flow: Optional[Tensor],
              fg: Optional[Tensor] = None):
    """One iteration of the recurrent loop of CRFP_DSV.forward (model/CRFP.py:1555-1684) with the
    per-frame share of the up-front encoders (:1536-1553) folded in.  ``flow`` is None on the first
    frame of a clip (the i == 0 branch, :1634-1667).  ``mk`` is bool [n,1,8h,8w]."""
    n, _, h, w = lr.shape
    mkf = mk.float()
    lr8 = up_bilinear(lr, 8)                                                  # :1538
    x_lr = tap("x_lr", R(lrelu(conv(P, "encoder_lr.slice1.2", R(lrelu(conv(P, "encoder_lr.slice1.0", lr)))))))  # :1540
    fvb = fv * mkf + lr8 * (1 - mkf)                                          # :1544
    x_hr = torch.cat((R(fvb), R(lr8)), dim=1)                                 # :1547 (staged as conv input; the head keeps the fp32 lr8)
    tap("xin8", x_hr)
    x_hr = tap("x_hr", R(lrelu(conv(P, "encoder_hr.slice1.2", tap("enc_hr0", R(lrelu(conv(P, "encoder_hr.slice1.0", x_hr))))))))

    prop = tap("prop0", R(pixel_shuffle_pack(P, "upsample.", x_lr, 2)))       # :1560  [n,24,2h,2w]
    if flow is not None:
        flow2 = up_bilinear(flow, 2) * 2.0                                    # :1565
        flow8 = up_bilinear(flow, 8) * 8.0                                    # :1566
        prev_hr = st["hr"]                                                    # :1568
        tap("flow2", flow2.permute(0, 2, 3, 1))
        prev2 = tap("prev2", R(pixel_unshuffle_pack_v2(P, "downsample.", prev_hr, 4)))   # :1569 [n,32,2h,2w]
        prev2_w = tap("prev2w", R(flow_warp(prev2, flow2.permute(0, 2, 3, 1))))          # :1570
        prev_hr_w = tap("prevhrw", R(flow_warp(prev_hr, flow8.permute(0, 2, 3, 1))))     # :1571
        carry = torch.chunk(tap("carryw", R(flow_warp(torch.cat(st["carry"], dim=1), flow2.permute(0, 2, 3, 1)))), 3, dim=1)  # :1573-1582
        off_feat = None
        new_carry = []
        # streaming variant only (model/CRFP_test.py:2296-2298): regional mask, bilinear x0.25 at 2x res
        fg2 = None if fg is None else F.interpolate(fg.float(), scale_factor=0.25, mode="bilinear", align_corners=False)
        for lvl in range(3):                                                  # :1585-1622
            cur = torch.cat((prop, carry[lvl]), dim=1)
            aligned, off_feat = dcn_module(P, f"dcn_{lvl}.", cur, prev2, prev2_w, flow2, off_feat,
                                           dg=cfg.dg, repeat=False, interpolate="none")
            res_in = torch.cat([cur, aligned], dim=1)
            if fg2 is not None and lvl > 0:     # CRFP_test.py:2361,2375 (the level-0 product :2347 is a no-op)
                res_in = R(res_in * fg2)
            y = tap(f"res{lvl}", resblocks_with_input_conv(P, f"forward_resblocks_{lvl}.", res_in))
            prop, c_new = y[:, :cfg.prop], y[:, cfg.prop:]
            new_carry.append(c_new)
        up = tap("up", R(lrelu(pixel_shuffle_pack(P, "upsample_post.", prop, 4))))       # :1625
        aligned, _ = dcn_module(P, "dcn_3.", up, prev_hr, prev_hr_w, flow8, off_feat,
                                dg=1, repeat=True, interpolate="pixelshuffle", flow_is_mfma_operand=False)  # :1626
        res_in = torch.cat([up, aligned], dim=1)
        if fg is not None:                      # CRFP_test.py:2389
            res_in = R(res_in * fg.float())
        feat = tap("feat", resblocks_with_input_conv(P, "forward_resblocks_3.", res_in))   # :1629-1630
    else:
        zeros2 = lr.new_zeros(n, cfg.mid, 2 * h, 2 * w)
        new_carry = []
        for lvl in range(3):                                                  # :1637-1661
            y = resblocks_with_input_conv(P, f"forward_resblocks_{lvl}.",
                                          torch.cat([prop, zeros2, st["carry"][lvl]], dim=1))
            prop, c_new = y[:, :cfg.prop], y[:, cfg.prop:]
            new_carry.append(c_new)
        up = R(lrelu(pixel_shuffle_pack(P, "upsample_post.", prop, 4)))       # :1664
        feat = resblocks_with_input_conv(P, "forward_resblocks_3.", torch.cat([up, st["hr"]], dim=1))  # :1666-1667

    fused = conv(P, "conv_tttf", torch.cat([feat, x_hr], dim=1))              # :1672-1673
    feat = tap("state_hr", R(lrelu(mkf * fused + (1 - mkf) * feat)))          # :1674-1675 (the new recurrent state)
    out = conv(P, "conv_last", feat)                                          # :1678
    out = out + (up_bilinear(rgb_to_y(lr), 8) if cfg.y_only else lr8)         # :1679-1683
    return out, {"hr": feat, "carry": new_carry, "first": False}


def crfp_dsv_forward(P, lrs: Tensor, fvs: Tensor, mks: Tensor, cfg: Optional[DSVConfig] = None) -> Tensor:
    """CRFP_DSV.forward (model/CRFP.py:1510-1686): lrs[n,t,3,h,w], fvs[n,t,3,8h,8w], bool
    mks[n,t,1,8h,8w] -> [n,t,3|1,8h,8w]."""
    cfg = cfg or DSVConfig()
    n, t, c, h, w = lrs.shape
    flows = compute_flow(P, lrs) if t > 1 else None
    st = new_state(cfg, n, h, w, lrs)
    outs = []
    for i in range(t):
        out, st = dsv_frame(P, cfg, st, lrs[:, i], fvs[:, i], mks[:, i], flows[:, i - 1] if i > 0 else None)
        outs.append(out)
    return torch.stack(outs, dim=1)


class StreamOracle:
    """One-frame-per-call variant (model/CRFP_test.py:2114-2478, MRCF_simple_v18): recurrent state and
    the previous LR frame persist between calls; ``clear_states`` (:2473-2478) starts a new sequence.
    Flow of the first frame of a sequence is FNet(frame, last frame of the same call) (:2234-2239) but
    unused, because that frame takes the state-less branch (:2309 ``torch.is_tensor(self.feat_prop_lv3)``)."""

    def __init__(self, P, cfg: Optional[DSVConfig] = None):
        self.P, self.cfg = P, cfg or DSVConfig()
        self.clear_states()

    def clear_states(self):
        self.prev_lr = None
        self.st = None

    def __call__(self, lrs: Tensor, fvs: Tensor, mks: Tensor, fgs: Optional[Tensor] = None) -> Tensor:
        n, t, c, h, w = lrs.shape
        if self.st is None:
            self.st = new_state(self.cfg, n, h, w, lrs)
        outs = []
        for i in range(t):
            lr = lrs[:, i]
            flow = None
            if not self.st["first"]:
                flow = fnet(self.P, "spynet.", lr, self.prev_lr)
            out, self.st = dsv_frame(self.P, self.cfg, self.st, lr, fvs[:, i], mks[:, i], flow,
                                     None if fgs is None else fgs[:, i])
            self.prev_lr = lr
            outs.append(out)
        return torch.stack(outs, dim=1)


# ----------------------------------------------------------------------------- metrics
def psnr(img1: Tensor, img2: Tensor, mask: Tensor) -> float:
    """utils.psnr_cuda, batch_avg=False branch (utils.py:166-185)."""
    B, C, H, W = img1.shape
    mse = (((img1 - img2) ** 2) * mask).sum() / (mask.float().sum() * C)
    if mse == 0:
        return float(-20 * torch.log10(torch.sqrt((1 / 255.) ** 2 / torch.prod(torch.tensor(img1.size())))))
    return float(-20 * torch.log10(torch.sqrt(mse)))


def to_y(img_nhwc: Tensor) -> Tensor:
    """utils.bgr2ycbcr(y_only=True) (utils.py:328-330): BGR weights applied to whatever order arrives."""
    y = torch.matmul(img_nhwc, torch.tensor([24.966, 128.553, 65.481])) + 16.0
    return y.unsqueeze(3).permute(0, 3, 1, 2)


def range_normalise(sr: Tensor, hr: Tensor):
    """utils.calc_psnr_and_ssim_cuda's data-dependent range conversion (utils.py:244-250)."""
    span = hr.max() - hr.min()
    if span > 2:
        return sr / 255., hr / 255.
    if span > 1:
        return (sr + 1.) / 2., (hr + 1.) / 2.
    return sr, hr


def _gaussian_window(channel: int, window_size: int = 11, sigma: float = 1.5) -> Tensor:
    """utils.gaussian / create_window (utils.py:187-195)."""
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def ssim(img1: Tensor, img2: Tensor, mask: Tensor) -> float:
    """utils.ssim_cuda -> ssim -> _ssim, size_average branch (utils.py:197-240): masked mean of the SSIM map."""
    channel = img1.shape[1]
    window = _gaussian_window(channel).type_as(img1)
    pad = 5
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    C = ssim_map.shape[1]
    return float((ssim_map * mask).sum() / (mask.float().sum() * C))


def calc_psnr_and_ssim(sr: Tensor, hr: Tensor, mask: Tensor):
    """utils.calc_psnr_and_ssim_cuda (utils.py:242-254)."""
    a, b = range_normalise(sr, hr)
    return psnr(a, b, mask), ssim(a, b, mask)


def psnr_rgb_and_y(sr: Tensor, hr: Tensor):
    """The two PSNR figures Trainer.eval_basicvsr logs per frame (trainer.py:348-369), mask = ones."""
    ones = torch.ones((sr.shape[0], 1, sr.shape[2], sr.shape[3]))
    a, b = range_normalise(sr, hr)
    p_rgb = psnr(a, b, ones)
    ys, yh = to_y(sr.permute(0, 2, 3, 1)), to_y(hr.permute(0, 2, 3, 1))
    a, b = range_normalise(ys, yh)
    return p_rgb, psnr(a, b, ones)


def load_numpy_state(sd_np) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
