"""Host-side mirror of the reference's ``model/CRFP.py`` for the CRFP_DSV inference path.

Same public names, constructor signatures, attribute names and ``state_dict`` keys as the reference
(model/CRFP.py: flow_warp :90, PixelShufflePack :154, PixelUnShufflePack_v2 :239, DCN_module :281,
ResidualBlockNoBN :433, ResidualBlocksWithInputConv :516, FNet :743, CRFP_DSV :1387) so reference
checkpoints load with ``strict=True`` and ``main.py:34`` / ``trainer.py:318`` call it unchanged.
The modules only *hold parameters*; all arithmetic runs in the HIP library: ``CRFP_DSV.forward`` is
one C-ABI call per clip (crfp_amd.engine.DSVEngine), the smaller modules call per-operator entry
points.  Inference only (no autograd), CUDA/HIP tensors only.
"""
import os

import torch
import torch.nn as nn

from crfp_amd import ops
from crfp_amd.dcn_v2 import DCNv2
from crfp_amd.engine import CRAEngine, DenseEngine, DSVEngine, SimpleEngine
from . import LTE


def flow_warp(x, flow, interpolation='bilinear', padding_mode='zeros', align_corners=True):
    return ops.flow_warp(x, flow, interpolation, padding_mode, align_corners)


def pixel_unshuffle(input, downscale_factor):
    return torch.nn.functional.pixel_unshuffle(input, downscale_factor)


def conv3x3(in_channels, out_channels, stride=1):
    return nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=True)


def _run(conv: nn.Conv2d, x, act="none", post_scale=1.0):
    return ops.conv3x3(x, conv.weight, conv.bias, act, post_scale)


class PixelShufflePack(nn.Module):
    def __init__(self, in_channels, out_channels, scale_factor, upsample_kernel):
        super().__init__()
        assert upsample_kernel == 3
        self.in_channels, self.out_channels = in_channels, out_channels
        self.scale_factor, self.upsample_kernel = scale_factor, upsample_kernel
        self.upsample_conv = nn.Conv2d(in_channels, out_channels * scale_factor * scale_factor, 3, padding=1)

    def forward(self, x):
        c = self.upsample_conv
        if self.scale_factor in (2, 4):   # F.pixel_shuffle (model/CRFP.py:192) fused into the conv's store
            return ops.conv3x3_ex(x, c.weight, c.bias, shuffle=self.scale_factor)
        return torch.nn.functional.pixel_shuffle(_run(c, x), self.scale_factor)


class PixelUnShufflePack_v2(nn.Module):
    def __init__(self, in_channels, out_channels, scale_factor, downsample_kernel):
        super().__init__()
        assert downsample_kernel == 3 and out_channels % (scale_factor * scale_factor) == 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.scale_factor, self.downsample_kernel = scale_factor, downsample_kernel
        self.downsample_conv = nn.Conv2d(in_channels * scale_factor * scale_factor, out_channels, 3, padding=1)

    def forward(self, x):
        c = self.downsample_conv
        if self.scale_factor == 4 and x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0:   # pixel_unshuffle (:28-42) fused into the conv's load
            return ops.conv3x3_ex(x, c.weight, c.bias, unshuffle=4)
        return _run(c, pixel_unshuffle(x, self.scale_factor))


class DCN_module(nn.Module):
    def __init__(self, mid_channels=64, dg=16, dk=3, max_mag=10, repeat=False, pre_offset=False,
                 interpolate='none', offset_only=False):
        super().__init__()
        if offset_only:
            raise NotImplementedError("offset_only is never enabled by the reference's CRFP_DSV")
        self.mid_channels, self.dg_num, self.dk = mid_channels, dg, dk
        self.max_residue_magnitude = max_mag
        self.pre_offset, self.repeat, self.interpolate, self.offset_only = pre_offset, repeat, interpolate, offset_only
        if pre_offset:
            if interpolate == 'pixelshuffle':
                self.upsample = PixelShufflePack(mid_channels * 8, mid_channels, 4, upsample_kernel=3)
            elif interpolate == 'bilinear':
                self.upsample = nn.Upsample(scale_factor=4, mode='bilinear', align_corners=False)
            self.conv_fuse = nn.Conv2d(mid_channels * 2, mid_channels, 3, 1, 1)
        self.init_channels = mid_channels * 2 + 2
        self.dcn_block = nn.Sequential(nn.Conv2d(self.init_channels, mid_channels, 3, 1, 1),
                                       nn.LeakyReLU(0.1, inplace=True),
                                       nn.Conv2d(mid_channels, mid_channels, 3, 1, 1),
                                       nn.LeakyReLU(0.1, inplace=True))
        n_pos = 1 if repeat else dk * dk
        self.dcn_offset = nn.Conv2d(mid_channels, dg * 2 * n_pos, 3, 1, 1)
        self.dcn_mask = nn.Conv2d(mid_channels, dg * n_pos, 3, 1, 1)
        self.dcn = DCNv2(mid_channels, mid_channels, dk, stride=1, padding=(dk - 1) // 2, dilation=1,
                         deformable_groups=dg)
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)
        self.init_dcn()

    def init_dcn(self):
        with torch.no_grad():
            for m in (self.dcn_offset, self.dcn_mask):
                m.weight.zero_()
                m.bias.zero_()
            self.conv_identify(self.dcn.weight, self.dcn.bias)

    @staticmethod
    def conv_identify(weight, bias):
        with torch.no_grad():
            weight.zero_()
            bias.zero_()
            o, i, kh, kw = weight.shape
            d = torch.arange(min(o, i))
            weight[d, d, kh // 2, kw // 2] = 1.0

    def forward(self, cur_x, pre_x, pre_x_aligned, flow, pre_offset=None):
        # torch.cat (model/CRFP.py:331,336) -> two-input convs (cur_x | [pre_x_aligned | flow]; f | pre_offset)
        b0, b2 = self.dcn_block[0], self.dcn_block[2]
        f = ops.conv3x3_ex(cur_x, b0.weight, b0.bias, x2=torch.cat([pre_x_aligned, flow], dim=1), act="lrelu")
        f = _run(b2, f, "lrelu")
        if torch.is_tensor(pre_offset):
            if self.interpolate == 'pixelshuffle':
                pre_offset = self.upsample(pre_offset) * 2.
            elif self.interpolate == 'bilinear':
                pre_offset = ops.upsample_bilinear(pre_offset, scale_factor=4, mul=2.0)
            f = ops.conv3x3_ex(f, self.conv_fuse.weight, self.conv_fuse.bias, x2=pre_offset, act="lrelu")
        offset = _run(self.dcn_offset, f, "tanh", float(self.max_residue_magnitude))
        mask = _run(self.dcn_mask, f, "sigmoid")
        flow_yx = flow.flip(1)
        if self.repeat and self.dg_num == 1 and self.mid_channels == 4 and isinstance(self.dcn, DCNv2):
            # one (dy, dx) and one mask per pixel for all 9 taps (:341-347): never tiled 9x (crfp_dcnv2_shared_f32)
            return ops.dcnv2_shared(pre_x, offset + flow_yx, mask, self.dcn.weight, self.dcn.bias), f
        K = self.dk * self.dk
        if self.repeat:
            B, C2, H, W = offset.shape
            offset = (offset.view(B, 2, C2 // 2, H, W) + flow_yx.unsqueeze(2)).repeat(1, K, 1, 1, 1).view(B, C2 * K, H, W)
            mask = mask.repeat(1, K, 1, 1)
        else:
            offset = offset + flow_yx.repeat(1, offset.size(1) // 2, 1, 1)
        return self.dcn(pre_x, offset, mask), f


class ResidualBlockNoBN(nn.Module):
    def __init__(self, mid_channels=64, res_scale=1.0):
        super().__init__()
        self.res_scale = res_scale
        self.conv1 = nn.Conv2d(mid_channels, mid_channels, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(mid_channels, mid_channels, 3, 1, 1, bias=True)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        # x + conv2(relu(conv1(x))) * res_scale: scale and residual ride in conv2's epilogue
        return ops.conv3x3_ex(_run(self.conv1, x, "relu"), self.conv2.weight, self.conv2.bias, residual=x, post_scale=float(self.res_scale))


class ResidualBlocksWithInputConv(nn.Module):
    def __init__(self, in_channels, out_channels=64, num_blocks=30):
        super().__init__()
        self.main = nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, 1, 1, bias=True),
                                  nn.LeakyReLU(negative_slope=0.1, inplace=True),
                                  nn.Sequential(*[ResidualBlockNoBN(mid_channels=out_channels)
                                                  for _ in range(num_blocks)]))

    def forward(self, feat):
        x = _run(self.main[0], feat, "lrelu")
        for blk in self.main[2]:
            x = blk(x)
        return x


class FNet(nn.Module):
    """Optical-flow net (reference model/CRFP.py:743-814): same layer containers / parameter names."""

    def __init__(self, in_nc):
        super().__init__()

        def two(ci, co, tail):
            return nn.Sequential(nn.Conv2d(ci, co, 3, 1, 1, bias=True), nn.ReLU(inplace=True),
                                 nn.Conv2d(co, co, 3, 1, 1, bias=True), nn.ReLU(inplace=True), tail)

        up = lambda: nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False)  # noqa: E731
        self.encoder1 = two(2 * in_nc, 32, nn.AvgPool2d(2, 2))
        self.encoder2 = two(32, 64, nn.AvgPool2d(2, 2))
        self.encoder3 = two(64, 128, nn.AvgPool2d(2, 2))
        self.decoder1 = two(128, 256, up())
        self.decoder2 = two(256, 128, up())
        self.decoder3 = two(128, 64, up())
        self.flow = nn.Sequential(nn.Conv2d(64, 32, 3, 1, 1, bias=True), nn.ReLU(inplace=True),
                                  nn.Conv2d(32, 2, 3, 1, 1, bias=True))

    def forward(self, x1, x2):
        """Flow from x1 to x2, [n,2,h,w]; convs and resizes on the HIP kernels."""
        _, _, h, w = x1.shape
        o = None
        for blk in (self.encoder1, self.encoder2, self.encoder3):
            # the first conv takes [x1 | x2] as two inputs (torch.cat of the reference, :801)
            o = ops.conv3x3_ex(x1, blk[0].weight, blk[0].bias, x2=x2, act="relu") if o is None else _run(blk[0], o, "relu")
            o = ops.avgpool2(_run(blk[2], o, "relu"))
        for blk in (self.decoder1, self.decoder2, self.decoder3):
            o = _run(blk[2], _run(blk[0], o, "relu"), "relu")
            o = ops.upsample_bilinear(o, scale_factor=2)
        o = _run(self.flow[2], _run(self.flow[0], o, "relu"), "tanh", 256.0)
        return ops.upsample_bilinear(o, size=(h, w))


class conv(nn.Module):
    """Reference ``conv`` (model/CRFP.py:145-152): ReLU applied to the INPUT, then nn.Conv2d."""

    def __init__(self, in_channels, out_channels, kernel_size=7, stride=1, padding=3):
        super().__init__()
        if stride != 1 or padding != kernel_size // 2 or kernel_size not in (3, 5, 7):
            raise NotImplementedError("conv: stride 1, 'same' padding, kernel 3 / 5 / 7 (the reference's SPyNet uses 7)")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding, bias=True)
        self.act = nn.ReLU()

    def forward(self, x):
        return ops.convkxk(x, self.conv.weight, self.conv.bias, pre_relu=True)


class SPyNetBasicModule(nn.Module):
    """Reference SPyNetBasicModule (model/CRFP.py:686-741): five ``conv`` 7x7, 8 -> 32 -> 64 -> 32 -> 16 -> 2."""

    def __init__(self):
        super().__init__()
        self.basic_module = nn.Sequential(conv(8, 32), conv(32, 64), conv(64, 32), conv(32, 16), conv(16, 2))

    def forward(self, tensor_input):
        x = tensor_input
        for m in self.basic_module:
            x = m(x)
        return x


class SPyNet(nn.Module):
    """Reference SPyNet (model/CRFP.py:554-684): same ctor, ``basic_module.{0..5}.basic_module.{0..4}.conv`` parameters and
    ``mean`` / ``std`` buffers; ``forward(ref, supp)`` is one native call (crfp_spynet_forward).  CRFP_DSV itself uses
    FNet (the SPyNet line of the reference is commented out, :1405); this is the optional flow network."""

    def __init__(self, pretrained, device):
        super().__init__()
        self.device = device
        self.basic_module = nn.ModuleList([SPyNetBasicModule() for _ in range(6)])
        if isinstance(pretrained, str):
            sd = self.state_dict()
            sd.update({k: v for k, v in torch.load(pretrained, map_location="cpu").items()})
            self.load_state_dict(sd, strict=True)
        elif pretrained is not None:
            raise TypeError(f'[pretrained] should be str or None, but got {type(pretrained)}.')
        self.register_buffer('mean', torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer('std', torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def _params(self):
        return [p for lvl in self.basic_module for m in lvl.basic_module for p in (m.conv.weight, m.conv.bias)]

    @torch.no_grad()
    def forward(self, ref, supp):
        return ops.spynet_forward(self._params(), ref, supp)

    @torch.no_grad()
    def compute_flow(self, ref, supp):
        """Inputs already a multiple of 32 and normalised by the caller in the reference (:593-664); provided through the
        per-operator entry points (avg-pool pyramid, align_corners=True flow upsampling, border warp, 7x7 convs)."""
        n, _, h, w = ref.shape
        refs, supps = [(ref - self.mean) / self.std], [(supp - self.mean) / self.std]
        for _ in range(5):
            refs.append(ops.avgpool2(refs[-1]))
            supps.append(ops.avgpool2(supps[-1]))
        refs, supps = refs[::-1], supps[::-1]
        flow = ref.new_zeros(n, 2, h // 32, w // 32)
        for level in range(6):
            flow_up = flow if level == 0 else ops.upsample_bilinear_ac(flow, 2, mul=2.0)
            warped = flow_warp(supps[level], flow_up.permute(0, 2, 3, 1).contiguous(), padding_mode='border')
            flow = flow_up + self.basic_module[level](torch.cat([refs[level], warped, flow_up], 1))
        return flow


class CRFP_DSV(nn.Module):
    """Drop-in for the reference's CRFP_DSV (model/CRFP.py:1387-1706).  ``spynet_pretrained`` may be
    None (the reference would crash: it torch.load()s it unconditionally, :1407) -- weights then come
    from ``load_state_dict`` / ``init_weights``."""

    def __init__(self, device, mid_channels=16, y_only=False, hr_dcn=True, offset_prop=True, spynet_pretrained=None):
        super().__init__()
        if mid_channels < 8 or mid_channels % 8:
            raise ValueError(f"mid_channels = {mid_channels}: the reference's own table needs a multiple of 8 (last_channels = "
                             "mid_channels // 8, 8 deformable groups; model/CRFP.py:1393-1395)")
        self.device = device
        self.mid_channels, self.last_channels = mid_channels, mid_channels // 8
        self.dg_num, self.dk, self.max_residue_magnitude = 8, 3, 10
        self.y_only, self.hr_dcn, self.offset_prop, self.split_ratio = y_only, hr_dcn, offset_prop, 3
        m, l = mid_channels, mid_channels // 8

        self.spynet = FNet(in_nc=3)
        if spynet_pretrained is not None:
            self.spynet.load_state_dict(torch.load(spynet_pretrained, map_location="cpu"))
        # the module table of model/CRFP.py:1409-1435 for every flag combination (state_dict keys and shapes pinned against the imported
        # reference by tests/golden/make_flags_golden.py).  hr_dcn=False swaps dcn_3 / forward_resblocks_3 for mid-channel modules that the
        # reference's own forward then feeds last_channels-wide tensors -- it raises on the first frame there, and so does this class.
        self.dcn_0 = DCN_module(m, 8, 3, 10)
        self.dcn_1 = DCN_module(m, 8, 3, 10, pre_offset=offset_prop, interpolate='none')
        self.dcn_2 = DCN_module(m, 8, 3, 10, pre_offset=offset_prop, interpolate='none')
        if hr_dcn:
            self.dcn_3 = DCN_module(l, 1, 3, 10, repeat=True, pre_offset=offset_prop, interpolate='pixelshuffle')
        else:
            self.dcn_3 = DCN_module(m, 8, 3, 10, pre_offset=offset_prop, interpolate='none')
        self.encoder_lr = LTE.LTE_simple_lr(m)
        self.encoder_hr = LTE.LTE_simple_hr_single(l)
        self.conv_tttf = conv3x3(l * 2, l)
        self.forward_resblocks_0 = ResidualBlocksWithInputConv(m * 2, m, 1)
        self.forward_resblocks_1 = ResidualBlocksWithInputConv(m * 2, m, 1)
        self.forward_resblocks_2 = ResidualBlocksWithInputConv(m * 2, m, 1)
        self.forward_resblocks_3 = ResidualBlocksWithInputConv(l * 2, l, 1) if hr_dcn else ResidualBlocksWithInputConv(m * 2, m, 1)
        self.downsample = PixelUnShufflePack_v2(l, m, 4, downsample_kernel=3)
        self.upsample = PixelShufflePack(m, (m * self.split_ratio) // 4, 2, upsample_kernel=3)
        self.upsample_post = PixelShufflePack((m * self.split_ratio) // 4, l, 4, upsample_kernel=3)
        self.conv_last = nn.Conv2d(l, 1 if y_only else 3, 3, 1, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)
        self._engine = None
        self._engine_sig = None
        self._engine_sum = None
        # numerics policy of the HIP engine (crfp_amd.engine.DSVEngine): not part of the reference's interface
        self.precision = "split"      # "split": split-fp16 MFMA scheme (fp32-grade) | "f32": strict fp32 MFMA
        self.on_overflow = "poison"   # "poison" | "fallback" | "raise" when an activation leaves the fp16 operand range
        self.storage = "f32"          # "f32" | "bf16": activation / state storage in HBM (BASELINE configs 3-5 are bf16)
        self.inputs_resident = False  # streaming only: the frame tensors are complete before each call (CRFP_DSV_INPUTS_RESIDENT, engine.py)

    # ---- engine management: repack whenever a parameter was modified or moved
    def _signature(self):
        """(address, in-place version) of every parameter: changes under load_state_dict, optimizer steps, ``.to()`` and any
        in-place op on the parameter itself.  It does NOT see writes through ``param.data`` (a ``.data`` alias has its own
        version counter) -- after such writes call ``invalidate_packed()``.  ``CRFP_CHECK_PACKED=1`` in the environment makes
        every ``engine()`` call verify an on-device checksum of the parameters against the one taken at pack time (one host
        sync per call: a debugging aid) and raise if they differ."""
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _checksum(self):
        ps = [p.detach().reshape(-1) for p in self.parameters()]
        flat = torch.cat(ps).double()
        return torch.stack([flat.sum(), flat.abs().sum(), (flat * torch.arange(1, flat.numel() + 1, device=flat.device, dtype=torch.float64)).sum()])

    def invalidate_packed(self):
        """Drop the packed-weight image: the next forward repacks from the current parameter values.  Needed only after
        writing parameters through ``.data`` (see ``_signature``)."""
        self._engine_sig = None

    _engine_class = DSVEngine
    _engine_mids = (16, 32)   # mid_channels = 16 runs embedded in the 32-channel schedule (crfp_amd.engine.embed_mid32)

    def has_engine(self) -> bool:
        """The one-call C++ schedule (csrc/engine.hip) exists for the configuration the reference ships and evaluates (main.py:34 with
        eval.sh's flags: mid_channels=32, hr_dcn, offset_prop) and -- round 6 -- for mid_channels = 16 (the constructor default,
        model/CRFP.py:1388), which runs as the same function embedded in the 32-channel schedule; every other flag combination
        (hr_dcn / offset_prop off, mid_channels > 32) runs ``forward_composed``."""
        return self.mid_channels in self._engine_mids and bool(self.hr_dcn) and bool(self.offset_prop)

    def engine(self) -> DSVEngine:
        if not self.has_engine():
            raise RuntimeError(f"crfp_amd: no one-call engine for mid_channels={self.mid_channels}, hr_dcn={self.hr_dcn}, "
                               f"offset_prop={self.offset_prop}: this model runs through forward_composed (per-operator HIP calls)")
        dev = next(self.parameters()).device
        sig = self._signature()
        check = os.environ.get("CRFP_CHECK_PACKED") == "1"
        if (self._engine is None or self._engine_sig != sig or self._engine.device != dev
                or self._engine.storage != self.storage):
            self._engine = self._engine_class(self.state_dict(), dev, self.y_only, storage=self.storage, mid_channels=self.mid_channels)
            self._engine_sig = sig
            self._engine_sum = self._checksum() if check else None
        elif check and self._engine_sum is not None and not torch.equal(self._engine_sum, self._checksum()):
            raise RuntimeError("crfp_amd: parameters changed without their version counters moving (a write through `.data`?): "
                               "the packed weights are stale -- call model.invalidate_packed() after such writes")
        self._engine.precision, self._engine.on_overflow = self.precision, self.on_overflow
        self._engine.inputs_resident = bool(self.inputs_resident)
        return self._engine

    def compute_flow(self, lrs):
        n, t, c, h, w = lrs.shape
        cur = lrs[:, 1:].reshape(-1, c, h, w)
        prev = lrs[:, :-1].reshape(-1, c, h, w)
        if not self.has_engine():   # FNet has no mid_channels in it: the per-operator composition of the same network
            return self.spynet(cur.contiguous(), prev.contiguous()).view(n, t - 1, 2, h, w), None
        return self.engine().compute_flow(cur, prev).view(n, t - 1, 2, h, w), None

    @torch.no_grad()
    def forward(self, lrs, fvs, mks):
        if not self.has_engine():
            return self.forward_composed(lrs, fvs, mks)
        return self.engine().forward(lrs, fvs, mks)

    @torch.no_grad()
    def forward_composed(self, lrs, fvs, mks):
        """The recurrence of model/CRFP.py:1510-1686 as a composition of this file's module mirrors -- every convolution, warp, resize and
        deformable convolution a per-operator HIP call -- for the flag combinations without a one-call engine: any mid_channels (the
        ctor default is 16), ``offset_prop=False`` and ``hr_dcn=False``.  It inherits the reference's behaviour on the last two exactly:
        ``offset_prop=False`` builds dcn_1..3 without ``conv_fuse`` while forward() still hands the offset feature down, so clips of more
        than one frame end in AttributeError (:336); ``hr_dcn=False`` builds 32-channel dcn_3 / forward_resblocks_3 and feeds them
        last_channels-wide tensors, a channel-count RuntimeError on the first frame (:1668).  Tensors between operators are NCHW torch
        tensors on the device; concatenations ride as the second input of the consuming conv where the operator has one."""
        if lrs.dim() != 5 or not lrs.is_cuda:
            raise RuntimeError("crfp_amd: forward_composed needs CUDA/HIP tensors lrs[n,t,3,h,w], fvs[n,t,3,8h,8w], mks[n,t,1,8h,8w]")
        n, t, _, h, w = lrs.shape
        m, l = self.mid_channels, self.last_channels
        lrs = lrs.float().contiguous()
        mkf = mks.to(torch.float32)
        flows, _ = self.compute_flow(lrs) if t > 1 else (None, None)
        flat = lrs.reshape(n * t, 3, h, w)
        up8_all = ops.upsample_bilinear(flat, scale_factor=8)                               # :1538
        x_lr = self.encoder_lr(flat, islr=True)[2].view(n, t, m, h, w)                      # :1540
        fov = fvs.float() * mkf + up8_all.view(n, t, 3, 8 * h, 8 * w) * (1.0 - mkf)         # :1543-1544
        x_hr, side = self._encode_hr(torch.cat((fov.reshape(n * t, 3, 8 * h, 8 * w), up8_all), 1), n, t)
        levels = ((self.dcn_0, self.forward_resblocks_0), (self.dcn_1, self.forward_resblocks_1), (self.dcn_2, self.forward_resblocks_2))
        keep = (m * self.split_ratio) // 4                      # channels that propagate to the next level; the rest is carried over time
        state2 = lrs.new_zeros(n, m, 2 * h, 2 * w)              # the 2x-resolution view of the state (zero before the first frame)
        state8 = lrs.new_zeros(n, l, 8 * h, 8 * w)
        carry = [lrs.new_zeros(n, m - keep, 2 * h, 2 * w) for _ in levels]
        outs = []
        for i in range(t):
            cur = self.upsample(x_lr[:, i].contiguous())
            if i == 0:
                # no history yet: [features | zero state | zero carry] through each level's residual block (:1634-1667)
                for k, (_, block) in enumerate(levels):
                    y = self._after_level(k, block(torch.cat((cur, state2, carry[k]), 1)), side, i, mkf)
                    cur, carry[k] = y[:, :keep].contiguous(), y[:, keep:].contiguous()
                up = torch.nn.functional.leaky_relu(self.upsample_post(cur), 0.1)
                state = self.forward_resblocks_3(torch.cat((up, state8), 1))
            else:
                flow = flows[:, i - 1].contiguous()
                f2 = ops.upsample_bilinear(flow, scale_factor=2, mul=2.0).permute(0, 2, 3, 1).contiguous()     # [n, 2h, 2w, 2], (x, y)
                f8 = ops.upsample_bilinear(flow, scale_factor=8, mul=8.0).permute(0, 2, 3, 1).contiguous()
                prev8 = state
                prev2 = self.downsample(prev8)
                prev2_w, prev8_w = flow_warp(prev2, f2), flow_warp(prev8, f8)
                carry = list(torch.chunk(flow_warp(torch.cat(carry, 1), f2), 3, dim=1))
                f2c, f8c = f2.permute(0, 3, 1, 2).contiguous(), f8.permute(0, 3, 1, 2).contiguous()
                offset = None
                for k, (dcn, block) in enumerate(levels):
                    cur = torch.cat((cur, carry[k]), 1)
                    aligned, offset = dcn(cur, prev2, prev2_w, f2c) if k == 0 else dcn(cur, prev2, prev2_w, f2c, offset)
                    y = self._after_level(k, block(torch.cat((cur, aligned), 1)), side, i, mkf)
                    cur, carry[k] = y[:, :keep].contiguous(), y[:, keep:].contiguous()
                up = torch.nn.functional.leaky_relu(self.upsample_post(cur), 0.1)
                aligned, _ = self.dcn_3(up, prev8, prev8_w, f8c, offset)
                state = self.forward_resblocks_3(torch.cat((up, aligned), 1))
            # fovea fusion and output head (:1672-1684)
            fused = ops.conv3x3_ex(state, self.conv_tttf.weight, self.conv_tttf.bias, x2=x_hr[:, i].contiguous())
            mk = mkf[:, i]
            state = torch.nn.functional.leaky_relu(mk * fused + (1.0 - mk) * state, 0.1)
            lr = lrs[:, i]
            base = (0.299 * lr[:, 0] + 0.587 * lr[:, 1] + 0.114 * lr[:, 2]).unsqueeze(1) if self.y_only else lr
            outs.append(_run(self.conv_last, state) + ops.upsample_bilinear(base.contiguous(), scale_factor=8))
        return torch.stack(outs, dim=1)

    # hooks of forward_composed that the CRFP_DSV_CRA wiring overrides
    def _encode_hr(self, x6, n, t):
        """encoder_hr on the [n * t, 6, 8h, 8w] stack of (blended fovea | x8 LR) -> (x_hr [n, t, last, 8h, 8w], side inputs)"""
        f = self.encoder_hr(x6, islr=True)[2]
        return f.view(n, t, f.shape[1], f.shape[2], f.shape[3]), None

    def _after_level(self, k, y, side, i, mkf):
        return y

    # ---- streaming interface of the reference's one-frame-per-call variant (model/CRFP_test.py:2216-2478)
    def clear_states(self):
        if self._engine is not None:
            self._engine.clear_states()

    @torch.no_grad()
    def forward_stream(self, lrs, fvs, mks, fgs=None):
        """lrs[1,t,3,h,w], fvs[1,t,3,8h,8w], mks / fgs[1,t,1,8h,8w] -> [1,t,3|1,8h,8w]; recurrent state and
        the previous LR frame persist between calls (reference model/CRFP_test.py:2234-2239,2438-2441)."""
        if not self.has_engine():
            raise NotImplementedError("crfp_amd: the one-frame-per-call interface exists for mid_channels=32, hr_dcn=True, offset_prop=True "
                                      "(the configuration test_video.py / test_runtime.py build)")
        eng = self.engine()
        if lrs.shape[0] == 1:
            outs = [eng.stream_frame(lrs[0, i], fvs[0, i], mks[0, i], None if fgs is None else fgs[0, i])
                    for i in range(lrs.shape[1])]
            return torch.stack(outs, dim=0)[None]
        # n > 1 sequences advance in lock-step, one crfp_dsv_stream_batch call per frame (the reference's forward carries n through every op)
        if fgs is not None:
            raise NotImplementedError("crfp_amd: the regional mask `fgs` is supported for one sequence per call (n = 1)")
        outs = [eng.stream_frame(lrs[:, i].contiguous(), fvs[:, i].contiguous(), mks[:, i].contiguous()) for i in range(lrs.shape[1])]
        return torch.stack(outs, dim=1)

    def init_weights(self, pretrained=None, strict=True):
        if isinstance(pretrained, str):
            saved = torch.load(pretrained, map_location="cpu")
            sd = self.state_dict()
            sd.update(saved)
            self.load_state_dict(sd, strict=strict)
        elif pretrained is not None:
            raise TypeError(f'"pretrained" must be a str or None. But received {type(pretrained)}.')


class CRFP_DSV_CRA(CRFP_DSV):
    """The reference's CRFP_DSV_CRA (model/CRFP.py:2314-2664; eval.sh's run name ends in ``_cra``, main.py:35 holds its commented-out
    factory line): CRFP_DSV plus a cross-resolution fusion of the fovea into every 2x level.  ``encoder_hr`` is the four-level
    ``LTE_simple_hr_ps``; after each level's residual block the 32 features are replaced, under the x0.25-resampled fovea mask, by
    ``conv_tttf_k(cat(features, fovea level k))`` (:2533-2535,2549-2551,2565-2567 and the first-frame twins).  Same constructor,
    same state_dict table as the reference (tests/golden/dsv_flags.npz).  mid_channels = 32 with both flags on runs the one-call engine
    schedule of this wiring (``crfp_cra_forward_batch``, crfp_amd.engine.CRAEngine: clip forward, lock-step batches, both storage
    types); every other constructor combination runs ``forward_composed`` (per-operator HIP calls), as in CRFP_DSV."""

    _engine_class = CRAEngine

    def __init__(self, device, mid_channels=16, y_only=False, hr_dcn=True, offset_prop=True, spynet_pretrained=None):
        super().__init__(device, mid_channels, y_only, hr_dcn, offset_prop, spynet_pretrained)
        m, l = self.mid_channels, self.last_channels
        # the reference's registration order (:2347-2353): encoder_hr, conv_tttf, conv_tttf_0..2 sit between encoder_lr and the
        # propagation branches -- rebuild the module table in that order so that state_dict() enumerates the same key sequence
        mods = dict(self._modules)
        for k in list(self._modules):
            del self._modules[k]
        for k, v in mods.items():
            if k == "encoder_hr":
                v = LTE.LTE_simple_hr_ps(l)
            self._modules[k] = v
            if k == "conv_tttf":
                for j in range(3):
                    self._modules[f"conv_tttf_{j}"] = conv3x3(m + l * 4, m)

    def compute_flow(self, lrs):
        n, t, c, h, w = lrs.shape   # the flow network as per-operator calls (the engine computes its flows inside the clip call)
        return self.spynet(lrs[:, 1:].reshape(-1, c, h, w).contiguous(), lrs[:, :-1].reshape(-1, c, h, w).contiguous()).view(n, t - 1, 2, h, w), None

    def forward_stream(self, lrs, fvs, mks, fgs=None):
        raise NotImplementedError("crfp_amd: the one-frame-per-call interface belongs to the plain CRFP_DSV wiring (model/CRFP_test.py)")

    def _encode_hr(self, x6, n, t):
        lv0, lv1, lv2, lv3 = self.encoder_hr(x6)
        v = lambda f: f.view(n, t, f.shape[1], f.shape[2], f.shape[3])   # noqa: E731
        return v(lv3), (v(lv0), v(lv1), v(lv2))

    def _after_level(self, k, y, side, i, mkf):
        conv = (self.conv_tttf_0, self.conv_tttf_1, self.conv_tttf_2)[k]
        fused = ops.conv3x3_ex(y, conv.weight, conv.bias, x2=side[k][:, i].contiguous())
        mk2 = ops.upsample_bilinear(mkf[:, i].contiguous(), scale_factor=0.25)     # img_downsample_4x(mk.float()), :2501
        return mk2 * fused + (1.0 - mk2) * y


class CRFP_simple(nn.Module):
    """The reference's CRFP_simple ("v13", model/CRFP.py:816-1099) and, with ``dense = True``, its sibling CRFP ("v15", :1101-1385): the
    ablation wirings in front of CRFP_DSV -- no carried features (every level passes all mid_channels on), ``upsample`` keeps
    mid_channels, the previous state is warped at 8x FIRST and then brought to 2x (:1023-1026), and both constructor flags are live:
    ``hr_dcn=False`` runs dcn_3 / forward_resblocks_3 at 2x resolution in mid_channels and up-samples afterwards (:1066-1078),
    ``offset_prop=False`` drops the offset hand-down (:1032-1033).  The dense variant feeds each residual block the warped previous
    state as a third input (:1311,1316,1321).  Same constructor and state_dict table as the reference (tests/golden/dsv_flags.npz).
    mid_channels = 32 with both flags on runs the one-call engine schedule of the wiring (``crfp_simple_forward_batch`` /
    ``crfp_dense_forward_batch``, crfp_amd.engine.SimpleEngine / DenseEngine: clip forward, lock-step batches, both storage types);
    every other constructor combination runs ``forward_composed`` (per-operator HIP calls)."""

    dense = False
    _engine_class = SimpleEngine

    def __init__(self, device, mid_channels=16, y_only=False, hr_dcn=True, offset_prop=True, spynet_pretrained=None):
        super().__init__()
        if mid_channels < 8 or mid_channels % 8:
            raise ValueError(f"mid_channels = {mid_channels}: needs a multiple of 8 (last_channels = mid_channels // 8, 8 deformable groups)")
        self.device = device
        self.mid_channels, self.last_channels = mid_channels, mid_channels // 8
        self.dg_num, self.dk, self.max_residue_magnitude = 8, 3, 10
        self.y_only, self.hr_dcn, self.offset_prop = y_only, hr_dcn, offset_prop
        m, l, k = mid_channels, mid_channels // 8, (3 if self.dense else 2)
        self.spynet = FNet(in_nc=3)
        if spynet_pretrained is not None:
            self.spynet.load_state_dict(torch.load(spynet_pretrained, map_location="cpu"))
        self.dcn_0 = DCN_module(m, 8, 3, 10)
        self.dcn_1 = DCN_module(m, 8, 3, 10, pre_offset=offset_prop, interpolate='none')
        self.dcn_2 = DCN_module(m, 8, 3, 10, pre_offset=offset_prop, interpolate='none')
        self.dcn_3 = (DCN_module(l, 1, 3, 10, repeat=True, pre_offset=offset_prop, interpolate='pixelshuffle') if hr_dcn
                      else DCN_module(m, 8, 3, 10, pre_offset=offset_prop, interpolate='none'))
        self.encoder_lr = LTE.LTE_simple_lr(m)
        self.encoder_hr = LTE.LTE_simple_hr_single(l)
        self.conv_tttf = conv3x3(l * 2, l)
        self.forward_resblocks_0 = ResidualBlocksWithInputConv(m * k, m, 1)
        self.forward_resblocks_1 = ResidualBlocksWithInputConv(m * k, m, 1)
        self.forward_resblocks_2 = ResidualBlocksWithInputConv(m * k, m, 1)
        self.forward_resblocks_3 = ResidualBlocksWithInputConv(l * k, l, 1) if hr_dcn else ResidualBlocksWithInputConv(m * k, m, 1)
        self.downsample = PixelUnShufflePack_v2(l, m, 4, downsample_kernel=3)
        self.upsample = PixelShufflePack(m, m, 2, upsample_kernel=3)
        self.upsample_post = PixelShufflePack(m, l, 4, upsample_kernel=3)
        self.conv_last = nn.Conv2d(l, 1 if y_only else 3, 3, 1, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)
        self._engine = self._engine_sig = self._engine_sum = None
        # numerics policy of the HIP engine, as on CRFP_DSV (not part of the reference's interface)
        self.precision, self.on_overflow, self.storage, self.inputs_resident = "split", "poison", "f32", False

    def compute_flow(self, lrs):
        n, t, c, h, w = lrs.shape
        cur, prev = lrs[:, 1:].reshape(-1, c, h, w), lrs[:, :-1].reshape(-1, c, h, w)
        return self.spynet(cur.contiguous(), prev.contiguous()).view(n, t - 1, 2, h, w), None

    init_weights = CRFP_DSV.init_weights
    # engine management: CRFP_DSV's (repack whenever a parameter was modified or moved)
    _signature, _checksum, invalidate_packed = CRFP_DSV._signature, CRFP_DSV._checksum, CRFP_DSV.invalidate_packed
    has_engine, engine, _engine_mids = CRFP_DSV.has_engine, CRFP_DSV.engine, CRFP_DSV._engine_mids

    @torch.no_grad()
    def forward(self, lrs, fvs, mks):
        if not self.has_engine():
            return self.forward_composed(lrs, fvs, mks)
        return self.engine().forward(lrs, fvs, mks)

    @torch.no_grad()
    def forward_composed(self, lrs, fvs, mks):
        """The recurrence of model/CRFP.py:938-1079 (dense: :1223-1365) as a composition of per-operator HIP calls: every flag combination."""
        if lrs.dim() != 5 or not lrs.is_cuda:
            raise RuntimeError("crfp_amd: needs CUDA/HIP tensors lrs[n,t,3,h,w], fvs[n,t,3,8h,8w], mks[n,t,1,8h,8w]")
        n, t, _, h, w = lrs.shape
        m, l = self.mid_channels, self.last_channels
        lrelu = lambda x: torch.nn.functional.leaky_relu(x, 0.1)   # noqa: E731
        lrs = lrs.float().contiguous()
        mkf = mks.to(torch.float32)
        flows = self.compute_flow(lrs)[0] if t > 1 else None
        flat = lrs.reshape(n * t, 3, h, w)
        up8_all = ops.upsample_bilinear(flat, scale_factor=8)
        x_lr = self.encoder_lr(flat, islr=True)[2].view(n, t, m, h, w)
        fov = fvs.float() * mkf + up8_all.view(n, t, 3, 8 * h, 8 * w) * (1.0 - mkf)
        x_hr = self.encoder_hr(torch.cat((fov.reshape(n * t, 3, 8 * h, 8 * w), up8_all), 1), islr=True)[2].view(n, t, l, 8 * h, 8 * w)
        blocks = (self.forward_resblocks_0, self.forward_resblocks_1, self.forward_resblocks_2)
        dcns = (self.dcn_0, self.dcn_1, self.dcn_2)
        rep = 2 if self.dense else 1                 # the dense variant hands the (warped) previous state to every block once more
        state = None                                 # [n, last, 8h, 8w] after the first frame
        outs = []
        for i in range(t):
            cur = self.upsample(x_lr[:, i].contiguous())
            if i == 0:
                z2, z8 = lrs.new_zeros(n, m, 2 * h, 2 * w), lrs.new_zeros(n, l, 8 * h, 8 * w)
                for block in blocks:
                    cur = block(torch.cat([cur] + [z2] * rep, 1))
                if self.hr_dcn:
                    cur = lrelu(self.upsample_post(cur))
                    state = self.forward_resblocks_3(torch.cat([cur] + [z8] * rep, 1))
                else:
                    state = self.forward_resblocks_3(torch.cat([cur] + [z2] * rep, 1))
            else:
                flow = flows[:, i - 1].contiguous()
                f2c = ops.upsample_bilinear(flow, scale_factor=2, mul=2.0)
                f2 = f2c.permute(0, 2, 3, 1).contiguous()
                if self.hr_dcn:      # warp at 8x, then both versions of the state to 2x (:1021-1026)
                    f8c = ops.upsample_bilinear(flow, scale_factor=8, mul=8.0)
                    prev8, prev8_w = state, flow_warp(state, f8c.permute(0, 2, 3, 1).contiguous())
                    prev2_w, prev2 = self.downsample(prev8_w), self.downsample(prev8)
                else:
                    prev2 = self.downsample(state)
                    prev2_w = flow_warp(prev2, f2)
                offset = None
                for k in range(3):
                    aligned, offset = dcns[k](cur, prev2, prev2_w, f2c) if k == 0 else dcns[k](cur, prev2, prev2_w, f2c, offset)
                    if not self.offset_prop:
                        offset = None
                    cur = blocks[k](torch.cat([cur, aligned] + ([prev2_w] if self.dense else []), 1))
                if self.hr_dcn:
                    cur = lrelu(self.upsample_post(cur))
                    aligned, _ = self.dcn_3(cur, prev8, prev8_w, f8c, offset)
                    state = self.forward_resblocks_3(torch.cat([cur, aligned] + ([prev8_w] if self.dense else []), 1))
                else:
                    aligned, _ = self.dcn_3(cur, prev2, prev2_w, f2c, offset)
                    state = self.forward_resblocks_3(torch.cat([cur, aligned] + ([prev2_w] if self.dense else []), 1))
            if not self.hr_dcn:
                state = lrelu(self.upsample_post(state))
            fused = ops.conv3x3_ex(state, self.conv_tttf.weight, self.conv_tttf.bias, x2=x_hr[:, i].contiguous())
            mk = mkf[:, i]
            state = lrelu(mk * fused + (1.0 - mk) * state)
            lr = lrs[:, i]
            base = (0.299 * lr[:, 0] + 0.587 * lr[:, 1] + 0.114 * lr[:, 2]).unsqueeze(1) if self.y_only else lr
            outs.append(_run(self.conv_last, state) + ops.upsample_bilinear(base.contiguous(), scale_factor=8))
        return torch.stack(outs, dim=1)


class CRFP(CRFP_simple):
    """The reference's CRFP ("v15", model/CRFP.py:1101-1385): see CRFP_simple."""

    dense = True
    _engine_class = DenseEngine


class MRCF_simple_v18(CRFP_DSV):
    """The reference's one-frame-per-call model (model/CRFP_test.py:2114-2478; built by test_video.py and
    test_runtime.py through ``from model import MRCF_test / MRCF_runtime``): same parameters as CRFP_DSV,
    ``forward(lrs, fvs, mks, fgs)`` keeps its state between calls, ``clear_states()`` starts a new sequence."""

    def __init__(self, device, mid_channels=16, y_only=False, hr_dcn=True, offset_prop=True, split_ratio=3,
                 spynet_pretrained=None):
        if split_ratio != 3:
            raise NotImplementedError("split_ratio 3 only (the reference's shipped configuration)")
        super().__init__(device, mid_channels, y_only, hr_dcn, offset_prop, spynet_pretrained)

    def forward(self, lrs, fvs, mks, fgs=None):
        return self.forward_stream(lrs, fvs, mks, fgs)
