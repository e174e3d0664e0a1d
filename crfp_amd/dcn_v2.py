"""Drop-in for the third-party ``dcn_v2`` package the reference imports (``from dcn_v2 import DCNv2``,
reference model/CRFP.py:6, test_runtime.py:11): same constructor, same ``weight``/``bias`` parameters
(which the reference mutates in place, model/CRFP.py:359-370), forward on the HIP kernels."""
import math

import torch
import torch.nn as nn

from crfp_amd import ops


class DCNv2(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        if isinstance(kernel_size, (tuple, list)):
            assert kernel_size[0] == kernel_size[1]
            kernel_size = kernel_size[0]
        if stride not in (1, (1, 1)):
            raise NotImplementedError("DCNv2: stride 1 only (every call site of CRFP)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = kernel_size, 1, padding, dilation
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.in_channels * self.kernel_size * self.kernel_size)
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
            self.bias.zero_()

    def forward(self, input, offset, mask):
        K = self.kernel_size * self.kernel_size
        assert 2 * self.deformable_groups * K == offset.shape[1]
        assert self.deformable_groups * K == mask.shape[1]
        return ops.dcnv2(input, offset, mask, self.weight, self.bias, self.kernel_size, self.padding, self.dilation,
                         self.deformable_groups)
