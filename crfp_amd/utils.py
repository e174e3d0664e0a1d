"""Host-side mirror of the metric functions of the reference's ``utils`` module that sit on the eval path
(utils.py:166-185 psnr_cuda, :187-240 ssim, :242-254 calc_psnr_and_ssim_cuda, :328-330 bgr2ycbcr(y_only)),
computed by libcrfp_hip.so (crfp_psnr_ssim_partial_f32: one pass over the image pair gives both figures).
Same names, argument meaning and quirks; CUDA/HIP tensors only (no CPU path in the product)."""
from __future__ import annotations

import math

import torch

from . import _lib
from .ops import _dev, _stream


def _mask_bytes(mask, n, h, w, device):
    """[n,1,h,w] (or broadcastable) bool / float mask -> contiguous uint8, None for 'all ones'."""
    if mask is None:
        return None
    m = mask.to(device)
    if m.dtype != torch.bool:
        m = m != 0
    return m.expand(n, 1, h, w).contiguous().view(torch.uint8)


def psnr_ssim_sums(a, b, mask=None, mul=1.0, add=0.0):
    """float64 tensor (sum m*(a'-b')^2 over channels, sum m*SSIM_map over channels, sum m) with x' = x*mul + add."""
    a, b = _dev(a, "a"), _dev(b, "b")
    n, c, h, w = a.shape
    if tuple(b.shape) != (n, c, h, w):
        raise ValueError(f"shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
    m8 = _mask_bytes(mask, n, h, w, a.device)
    acc = torch.zeros(3, dtype=torch.float64, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().crfp_psnr_ssim_partial_f32(a.data_ptr(), b.data_ptr(), None if m8 is None else m8.data_ptr(),
                                                         acc.data_ptr(), n, c, h, w, float(mul), float(add), _stream()),
                   "crfp_psnr_ssim_partial_f32")
    return acc


def _psnr_from(se, msum, shape):
    C = shape[1]
    mse = se / (msum * C)
    if mse == 0:   # utils.py:177-179
        return -20 * math.log10(math.sqrt((1 / 255.) ** 2 / float(torch.prod(torch.tensor(shape)))))
    return -20 * math.log10(math.sqrt(mse))


def psnr_cuda(img1, img2, mask, batch_avg=False):
    """utils.psnr_cuda, batch_avg=False branch.  Image range [0, 1]."""
    if batch_avg:
        raise NotImplementedError("batch_avg=True is a training-time branch of the reference (not on the eval path)")
    se, _, ms = (float(v) for v in psnr_ssim_sums(img1, img2, mask))
    return torch.tensor(_psnr_from(se, ms, tuple(img1.shape)))


def ssim_cuda(img1, img2, mask, batch_avg=False):
    """utils.ssim_cuda: masked mean of the 11x11-gaussian SSIM map.  Image range [0, 1]."""
    if batch_avg:
        raise NotImplementedError("batch_avg=True is a training-time branch of the reference (not on the eval path)")
    _, ss, ms = (float(v) for v in psnr_ssim_sums(img1, img2, mask))
    return torch.tensor(ss / (ms * img1.shape[1]))


def calc_psnr_and_ssim_cuda(sr, hr, mask, is_tensor=True, batch_avg=False):
    """utils.calc_psnr_and_ssim_cuda: range conversion chosen from hr's span (> 2: /255; > 1: (x+1)/2), then both."""
    if batch_avg:
        raise NotImplementedError("batch_avg=True is a training-time branch of the reference (not on the eval path)")
    sr, hr = _dev(sr, "sr"), _dev(hr, "hr")
    span = float(hr.max() - hr.min())
    mul, add = (1.0 / 255.0, 0.0) if span > 2 else ((0.5, 0.5) if span > 1 else (1.0, 0.0))
    se, ss, ms = (float(v) for v in psnr_ssim_sums(sr, hr, mask, mul, add))
    return torch.tensor(_psnr_from(se, ms, tuple(sr.shape))), torch.tensor(ss / (ms * sr.shape[1]))


def calc_psnr_and_ssim_regions(sr, hr, masks):
    """calc_psnr_and_ssim_cuda(sr, hr, m) for every mask m of one frame (the video rig's whole / fovea / outskirt / past
    regions, test_video.py:360-370) with ONE range probe and ONE host synchronisation instead of one per region; same
    kernel, same numbers.  Returns a list of (psnr, ssim) tensors."""
    sr, hr = _dev(sr, "sr"), _dev(hr, "hr")
    span = float(hr.max() - hr.min())
    mul, add = (1.0 / 255.0, 0.0) if span > 2 else ((0.5, 0.5) if span > 1 else (1.0, 0.0))
    sums = torch.stack([psnr_ssim_sums(sr, hr, m, mul, add).clone() for m in masks]).cpu()
    out = []
    for se, ss, ms in sums.tolist():
        out.append((torch.tensor(_psnr_from(se, ms, tuple(sr.shape))), torch.tensor(ss / (ms * sr.shape[1]))))
    return out


def bgr2ycbcr(img, y_only=False):
    """utils.bgr2ycbcr on an [N,H,W,3] tensor (BGR weights applied to whatever channel order arrives, as the
    reference does, trainer.py:362-363)."""
    if y_only:
        out = torch.matmul(img, torch.tensor([24.966, 128.553, 65.481], device=img.device)) + 16.0
        return out.unsqueeze(3).permute(0, 3, 1, 2)
    out = torch.matmul(img, torch.tensor([[24.966, 112.0, -18.214], [128.553, -74.203, -93.786], [65.481, -37.797, 112.0]],
                                         device=img.device)) + torch.tensor([16, 128, 128], device=img.device)
    return out.permute(0, 3, 1, 2)
