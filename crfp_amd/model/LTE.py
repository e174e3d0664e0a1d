"""Mirror of the two encoder classes CRFP_DSV uses from the reference's model/LTE.py
(LTE_simple_lr :34-51, LTE_simple_hr_single :100-117): same ctor argument, same ``slice1.{0,2}``
parameter names, forward returns ``(None, None, features)``."""
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
