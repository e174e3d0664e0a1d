"""Mirror of the two encoder classes CRFP_DSV uses from the reference's model/LTE.py
(LTE_simple_lr :34-51, LTE_simple_hr_single :100-117): same ctor argument, same ``slice1.{0,2}``
parameter names, forward returns ``(None, None, features)``; plus the four-level encoder of the CRFP_DSV_CRA wiring
(LTE_simple_hr_ps :119-166, PixelUnshuffle :21-33)."""
import torch
import torch.nn as nn

from crfp_amd import ops


class _TwoConvEncoder(nn.Module):
    def __init__(self, cin, mid_channels):
        super().__init__()
        self.slice1 = nn.Sequential(nn.Conv2d(cin, mid_channels, 3, 1, 1), nn.LeakyReLU(0.1, inplace=True),
                                    nn.Conv2d(mid_channels, mid_channels, 3, 1, 1), nn.LeakyReLU(0.1, inplace=True))

    def forward(self, x, islr=False):
        a, b = self.slice1[0], self.slice1[2]
        x = ops.conv3x3(x, a.weight, a.bias, "lrelu")
        return None, None, ops.conv3x3(x, b.weight, b.bias, "lrelu")


class LTE_simple_lr(_TwoConvEncoder):
    def __init__(self, mid_channels):
        super().__init__(3, mid_channels)


class LTE_simple_hr_single(_TwoConvEncoder):
    def __init__(self, mid_channels):
        super().__init__(6, mid_channels)


class PixelUnshuffle(nn.Module):
    """reference model/LTE.py:21-33 (a grouped one-hot convolution there): out[c * k * k + i * k + j, y, x] = in[c, k y + i, k x + j]"""

    def __init__(self, downscale_factor):
        super().__init__()
        self.downscale_factor = downscale_factor

    def forward(self, input):
        return torch.nn.functional.pixel_unshuffle(input, self.downscale_factor)


class LTE_simple_hr_ps(nn.Module):
    """Four feature levels of the blended fovea frame for the cross-resolution fusion of CRFP_DSV_CRA: ``forward(x)`` returns
    ``(x_lv0, x_lv1, x_lv2, x_lv3)`` -- three 4 * mid-channel maps at a quarter of the input size and one mid-channel map at full
    size (reference :156-166).  Same container layout (``slice1..4``, ``conv_lv0..3``), so the state_dict keys match."""

    def __init__(self, mid_channels):
        super().__init__()
        m, q = mid_channels, mid_channels * 4

        def pair(ci, co, head=()):
            return nn.Sequential(*head, nn.Conv2d(ci, co, 3, 1, 1), nn.LeakyReLU(0.1, inplace=True),
                                 nn.Conv2d(co, co, 3, 1, 1), nn.LeakyReLU(0.1, inplace=True))

        self.slice1 = pair(6, m)
        self.slice2 = pair(m * 16, q, (PixelUnshuffle(4),))
        self.slice3 = pair(q, q)
        self.slice4 = pair(q, q)
        self.conv_lv0 = nn.Conv2d(q, q, 3, 1, 1)
        self.conv_lv1 = nn.Conv2d(q, q, 3, 1, 1)
        self.conv_lv2 = nn.Conv2d(q, q, 3, 1, 1)
        self.conv_lv3 = nn.Conv2d(m, m, 3, 1, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)

    @staticmethod
    def _c(conv, x, **kw):
        return ops.conv3x3_ex(x, conv.weight, conv.bias, act="lrelu", **kw)

    def forward(self, x):
        c = self._c
        x = c(self.slice1[2], c(self.slice1[0], x))
        lv3 = c(self.conv_lv3, x)
        a, b = self.slice2[1], self.slice2[3]
        # pixel_unshuffle(4) rides in the first conv's load when the size allows it (ops.conv3x3_ex), as in PixelUnShufflePack_v2
        x = c(a, x, unshuffle=4) if x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0 else c(a, self.slice2[0](x))
        x = c(b, x)
        lv2 = c(self.conv_lv2, x)
        x = c(self.slice3[2], c(self.slice3[0], x))
        lv1 = c(self.conv_lv1, x)
        x = c(self.slice4[2], c(self.slice4[0], x))
        return c(self.conv_lv0, x), lv1, lv2, lv3
