"""Operator-level wrappers over the C-ABI (device tensors in, device tensors out).

Every function requires float32 CUDA(=HIP) tensors; CPU tensors raise -- the CPU restatement of
the path lives in ``oracle/`` and is test infrastructure, never a fallback.
"""
from __future__ import annotations

import torch

from . import _lib

ACT = {"none": 0, "relu": 1, "lrelu": 2, "tanh": 3, "sigmoid": 4}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"crfp_amd: `{name}` must be a CUDA/HIP tensor (this build has no CPU path)")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _on(*tensors):
    """Context that makes the operands' device current (the C-ABI enqueues on a stream of the CURRENT device, so the
    stream handle must be taken inside this block); all operands must live on one device."""
    devs = {t.device for t in tensors if isinstance(t, torch.Tensor)}
    if len(devs) != 1:
        raise RuntimeError(f"crfp_amd: operands on different devices: {sorted(map(str, devs))}")
    return torch.cuda.device(devs.pop())


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# Weights of the per-operator entry points are repacked on EVERY call, on the caller's stream, right in front of the kernel that
# reads them (one ~3 us pack launch).  Round 2 cached the packed image on the weight tensor keyed by ``Tensor._version``; writes
# through ``.data`` (which the reference uses: ``m.weight.data *= scale``, conv_identify at model/CRFP.py:359-370) do not bump
# that counter, so the cache could serve stale weights, and an image packed on one stream could be read half-written from
# another.  Callers that own the immutability of their weights hoist the repack themselves: ``pack_conv3x3`` /
# ``conv3x3_packed`` and ``pack_dcnv2_g8`` / ``dcnv2_g8_packed`` (the C-ABI's *_pack_f32 / *_packed_f32 forms); the engine
# (crfp_amd/engine.py) packs once per ``DSVEngine`` and is rebuilt by ``CRFP_DSV.engine()`` / ``invalidate_packed()``.


def _pack_buf(nbytes, device):
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("crfp_amd.ops: per-call weight repacking allocates; inside CUDA-graph capture use the *_packed forms "
                           "with an image packed before the capture")
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def pack_conv3x3(weight, bias=None):
    """Packed image of a [cout, cin, 3, 3] weight (+ bias) for ``conv3x3_packed``; valid until the caller changes the weights.
    Produced on the current stream: use it on that stream, or order the consumer behind it."""
    weight = _dev(weight, "weight")
    cout, cin = weight.shape[:2]
    assert tuple(weight.shape) == (cout, cin, 3, 3), "conv3x3: weight must be [cout, cin, 3, 3]"
    bias = torch.zeros(cout, dtype=torch.float32, device=weight.device) if bias is None else _dev(bias, "bias")
    L = _lib.lib()
    with _on(weight, bias):
        pk = _pack_buf(L.crfp_conv3x3_packed_bytes(cin, cout), weight.device)
        _lib.check(L.crfp_conv3x3_pack_f32(weight.data_ptr(), bias.data_ptr(), cin, cout, pk.data_ptr(), pk.numel(), _stream()),
                   "crfp_conv3x3_pack_f32")
    pk._crfp_shape = (cin, cout)
    return pk


def conv3x3_packed(x, packed, act="none", post_scale=1.0):
    """conv3x3 with an image from ``pack_conv3x3``."""
    x = _dev(x, "x")
    cin, cout = packed._crfp_shape
    n, c, h, w = x.shape
    assert c == cin, f"conv3x3_packed: input has {c} channels, the packed weight {cin}"
    out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    with _on(x, packed):
        _lib.check(_lib.lib().crfp_conv3x3_packed_f32(x.data_ptr(), packed.data_ptr(), out.data_ptr(), n, cin, cout, h, w, ACT[act],
                                                      float(post_scale), _stream()), "crfp_conv3x3_packed_f32")
    return out


def pack_dcnv2_g8(weight):
    """Packed image of a DCNv2(32 -> 32, 3x3, 8 groups) weight for ``dcnv2_g8_packed``."""
    weight = _dev(weight, "weight")
    assert tuple(weight.shape) == (32, 32, 3, 3)
    L = _lib.lib()
    with _on(weight):
        pk = _pack_buf(L.crfp_dcnv2_g8_packed_bytes(), weight.device)
        _lib.check(L.crfp_dcnv2_g8_pack_f32(weight.data_ptr(), pk.data_ptr(), pk.numel(), _stream()), "crfp_dcnv2_g8_pack_f32")
    return pk


def dcnv2_g8_packed(x, offset, mask, packed, bias):
    """DCNv2(32 -> 32, 3x3, pad 1, 8 groups) with an image from ``pack_dcnv2_g8``."""
    x, offset, mask, bias = _dev(x, "input"), _dev(offset, "offset"), _dev(mask, "mask"), _dev(bias, "bias")
    n, cin, h, w = x.shape
    assert cin == 32 and tuple(offset.shape) == (n, 144, h, w) and tuple(mask.shape) == (n, 72, h, w)
    L = _lib.lib()
    out = torch.empty((n, 32, h, w), dtype=torch.float32, device=x.device)
    with _on(x, offset, mask, packed, bias):
        ws = _ws(L.crfp_dcnv2_workspace_bytes(n, 32, 32, h, w, 3, 8), x.device)
        _lib.check(L.crfp_dcnv2_g8_packed_f32(x.data_ptr(), offset.data_ptr(), mask.data_ptr(), packed.data_ptr(), bias.data_ptr(),
                                              out.data_ptr(), n, h, w, ws.data_ptr(), ws.numel(), _stream()), "crfp_dcnv2_g8_packed_f32")
    return out


def flow_warp(x, flow, interpolation="bilinear", padding_mode="zeros", align_corners=True):
    """Drop-in for the reference's ``flow_warp`` (model/CRFP.py:90-130): x[n,c,h,w], flow[n,h,w,2]."""
    if tuple(x.shape[-2:]) != tuple(flow.shape[1:3]):
        raise ValueError(f"The spatial sizes of input ({tuple(x.shape[-2:])}) and flow "
                         f"({tuple(flow.shape[1:3])}) are not the same.")
    if interpolation != "bilinear" or not align_corners:
        raise NotImplementedError("flow_warp: only bilinear / align_corners=True (all reference call sites)")
    if padding_mode not in ("zeros", "border"):
        raise NotImplementedError(f"flow_warp: padding_mode {padding_mode!r}")
    x, flow = _dev(x, "x"), _dev(flow, "flow")
    n, c, h, w = x.shape
    L = _lib.lib()
    out = torch.empty_like(x)
    nb = L.crfp_flow_warp_workspace_bytes(n, c, h, w)
    ws = _ws(nb, x.device)
    with _on(x, flow):
        _lib.check(L.crfp_flow_warp_f32(x.data_ptr(), flow.data_ptr(), out.data_ptr(), n, c, h, w,
                                        1 if padding_mode == "border" else 0, ws.data_ptr(), ws.numel(), _stream()),
                   "crfp_flow_warp_f32")
    return out


def dcnv2(x, offset, mask, weight, bias, kernel_size=3, padding=1, dilation=1, deformable_groups=1):
    """Drop-in for ``dcn_v2.DCNv2.forward`` (reference model/CRFP.py:350)."""
    x, offset, mask = _dev(x, "input"), _dev(offset, "offset"), _dev(mask, "mask")
    weight, bias = _dev(weight, "weight"), _dev(bias, "bias")
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    K = kernel_size * kernel_size
    assert offset.shape[1] == 2 * deformable_groups * K, "offset channels != 2*deformable_groups*k*k"
    assert mask.shape[1] == deformable_groups * K, "mask channels != deformable_groups*k*k"
    assert weight.shape[1] == cin and tuple(offset.shape[-2:]) == (h, w) and tuple(mask.shape[-2:]) == (h, w)
    L = _lib.lib()
    out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    ws = _ws(L.crfp_dcnv2_workspace_bytes(n, cin, cout, h, w, kernel_size, deformable_groups), x.device)
    if (cin, cout, deformable_groups, kernel_size, padding, dilation) == (32, 32, 8, 3, 1, 1):   # MFMA fast path
        return dcnv2_g8_packed(x, offset, mask, pack_dcnv2_g8(weight), bias)
    with _on(x, offset, mask, weight, bias):
        _lib.check(L.crfp_dcnv2_forward_f32(x.data_ptr(), offset.data_ptr(), mask.data_ptr(), weight.data_ptr(),
                                            bias.data_ptr(), out.data_ptr(), n, cin, cout, h, w, kernel_size, padding,
                                            dilation, deformable_groups, ws.data_ptr(), ws.numel(), _stream()),
                   "crfp_dcnv2_forward_f32")
    return out


def dcnv2_shared(x, offset, mask, weight, bias):
    """DCNv2 (4 -> 4 channels, one group) with one (dy, dx) [n,2,h,w] and one mask [n,1,h,w] per pixel shared by the 9 taps: equal to
    ``dcnv2(x, offset.repeat(1, 9, 1, 1), mask.repeat(1, 9, 1, 1), ...)`` (reference model/CRFP.py:341-350) without the tiled tensors."""
    x, offset, mask, weight, bias = _dev(x, "input"), _dev(offset, "offset"), _dev(mask, "mask"), _dev(weight, "weight"), _dev(bias, "bias")
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    assert tuple(offset.shape) == (n, 2, h, w) and tuple(mask.shape) == (n, 1, h, w), "dcnv2_shared: offset [n,2,h,w], mask [n,1,h,w]"
    L = _lib.lib()
    out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    with _on(x, offset, mask, weight, bias):
        nb = L.crfp_dcnv2_shared_workspace_bytes(n, cin, h, w)
        ws = _ws(max(nb, 256), x.device)
        _lib.check(L.crfp_dcnv2_shared_f32(x.data_ptr(), offset.data_ptr(), mask.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                           n, cin, cout, h, w, ws.data_ptr(), ws.numel(), _stream()), "crfp_dcnv2_shared_f32")
    return out


def conv3x3(x, weight, bias=None, act="none", post_scale=1.0):
    """3x3 stride-1 pad-1 convolution + bias + activation on the fp32 MFMA path (weights repacked by this call)."""
    x, weight = _dev(x, "x"), _dev(weight, "weight")
    if tuple(weight.shape) != (weight.shape[0], x.shape[1], 3, 3):   # nn.Conv2d's own error class and wording (the reference's callers see a RuntimeError)
        raise RuntimeError(f"conv3x3: weight of size {list(weight.shape)}, expected input{list(x.shape)} to have {weight.shape[1]} channels, "
                           f"but got {x.shape[1]} channels instead")
    with _on(x, weight):
        return conv3x3_packed(x, pack_conv3x3(weight, bias), act, post_scale)


def conv3x3_ex(x, weight, bias=None, x2=None, residual=None, act="none", post_scale=1.0, unshuffle=0, shuffle=0, out=None, out_c0=0):
    """crfp_conv3x3_ex_f32: conv3x3(cat([x, x2], 1)) + bias -> activation -> * post_scale (+ residual), with the reference's layout steps
    fused: ``unshuffle=4`` reads x [n, c, 4h, 4w] as pixel_unshuffle(x, 4) (model/CRFP.py:28-42), ``shuffle=r`` stores pixel_shuffle(., r)
    (:184-193), ``out`` / ``out_c0`` write the result into channels [out_c0, out_c0 + cout) of an existing [n, C, h, w] tensor."""
    x, weight = _dev(x, "x"), _dev(weight, "weight")
    n = x.shape[0]
    if unshuffle == 4:
        cin, h, w = 16 * x.shape[1], x.shape[2] // 4, x.shape[3] // 4
        assert x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0, "conv3x3_ex: pixel_unshuffle(4) needs H, W divisible by 4"
    else:
        cin, h, w = x.shape[1], x.shape[2], x.shape[3]
    cin2 = 0
    if x2 is not None:
        x2 = _dev(x2, "x2")
        cin2 = x2.shape[1]
        assert tuple(x2.shape) == (n, cin2, h, w), "conv3x3_ex: x2 must match x in n, h, w"
    cout = weight.shape[0]
    if tuple(weight.shape) != (cout, cin + cin2, 3, 3):
        raise RuntimeError(f"conv3x3_ex: weight of size {list(weight.shape)}, expected input to have {weight.shape[1]} channels, "
                           f"but got {cin + cin2} channels instead")
    bias = torch.zeros(cout, dtype=torch.float32, device=x.device) if bias is None else _dev(bias, "bias")
    if residual is not None:
        residual = _dev(residual, "residual")
        assert tuple(residual.shape) == (n, cout, h, w), "conv3x3_ex: residual must be [n, cout, h, w]"
    r = int(shuffle) if shuffle and shuffle > 1 else 0
    if r:
        assert out is None, "conv3x3_ex: a pixel-shuffle store writes its own tensor"
        out = torch.empty((n, cout // (r * r), h * r, w * r), dtype=torch.float32, device=x.device)
        ctot = cout
    elif out is None:
        out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
        ctot = cout
    else:
        # written in place: a copy made here would never reach the caller's tensor
        if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()):
            raise ValueError("conv3x3_ex: `out` must be a contiguous float32 CUDA/HIP tensor (it is written in place)")
        assert out.shape[0] == n and tuple(out.shape[2:]) == (h, w), "conv3x3_ex: out must be [n, C, h, w]"
        ctot = out.shape[1]
    L = _lib.lib()
    with _on(x, weight, bias, x2, residual, out):
        nb = L.crfp_conv3x3_ex_workspace_bytes(n, cin, cin2, cout, h, w, int(unshuffle), r, int(residual is not None))
        if nb == 0:
            _lib.check(-3, "crfp_conv3x3_ex_workspace_bytes")
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
        _lib.check(L.crfp_conv3x3_ex_f32(x.data_ptr(), cin, x2.data_ptr() if x2 is not None else None, cin2, weight.data_ptr(), bias.data_ptr(),
                                         residual.data_ptr() if residual is not None else None, out.data_ptr(), n, cout, h, w, ACT[act],
                                         float(post_scale), int(unshuffle), r, int(out_c0), int(ctot), ws.data_ptr(), nb, _stream()),
                   "crfp_conv3x3_ex_f32")
    return out


def conv3x3_unpacked(x, weight, bias, act="none", post_scale=1.0):
    """The one-call form (crfp_conv3x3_f32: repacks the weights inside the call) -- what a caller without a cache pays."""
    x, weight, bias = _dev(x, "x"), _dev(weight, "weight"), _dev(bias, "bias")
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    L = _lib.lib()
    out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    ws = _ws(L.crfp_conv3x3_workspace_bytes(n, cin, cout, h, w), x.device)
    with _on(x, weight, bias):
        _lib.check(L.crfp_conv3x3_f32(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(), n, cin, cout,
                                      h, w, ACT[act], float(post_scale), ws.data_ptr(), ws.numel(), _stream()),
                   "crfp_conv3x3_f32")
    return out


def upsample_bilinear(x, scale_factor=None, size=None, mul=1.0):
    """nn.Upsample(scale_factor=.., 'bilinear', align_corners=False) or F.interpolate(size=..)."""
    x = _dev(x, "x")
    n, c, h, w = x.shape
    if scale_factor is not None:
        oh, ow = int(h * scale_factor), int(w * scale_factor)
        sh = sw = 1.0 / float(scale_factor)
    else:
        oh, ow = size
        # PyTorch computes in/out in float32 when only a size is given
        sh = float(torch.tensor(h, dtype=torch.float32) / torch.tensor(oh, dtype=torch.float32))
        sw = float(torch.tensor(w, dtype=torch.float32) / torch.tensor(ow, dtype=torch.float32))
    out = torch.empty((n, c, oh, ow), dtype=torch.float32, device=x.device)
    with _on(x):
        _lib.check(_lib.lib().crfp_upsample_bilinear_f32(x.data_ptr(), out.data_ptr(), n, c, h, w, oh, ow, sh, sw,
                                                         float(mul), _stream()), "crfp_upsample_bilinear_f32")
    return out


def sq_err_sums(a, b):
    """(sum (a-b)^2, sum (Y(a)-Y(b))^2) as float64 -- raw material of psnr / psnr_y."""
    a, b = _dev(a, "a"), _dev(b, "b")
    n, c, h, w = a.shape
    acc = torch.zeros(2, dtype=torch.float64, device=a.device)
    with _on(a, b):
        _lib.check(_lib.lib().crfp_psnr_partial_f32(a.data_ptr(), b.data_ptr(), acc.data_ptr(), n, c, h, w, _stream()),
                   "crfp_psnr_partial_f32")
    return acc


def avgpool2(x):
    """nn.AvgPool2d(2, 2) (floor mode) on an NCHW tensor."""
    x = _dev(x, "x")
    n, c, h, w = x.shape
    out = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    with _on(x):
        _lib.check(_lib.lib().crfp_avgpool2_f32(x.data_ptr(), out.data_ptr(), n, c, h, w, _stream()), "crfp_avgpool2_f32")
    return out


def fovea_head(state, x_hr, mask, lr, w_tttf, b_tttf, w_last, b_last, y_only=False):
    """Fused fovea fusion + output head (model/CRFP.py:1672-1684) -> (new_state [n,4,8h,8w], out [n,3|1,8h,8w])."""
    state, x_hr, lr = _dev(state, "state"), _dev(x_hr, "x_hr"), _dev(lr, "lr")
    w_tttf, b_tttf, w_last, b_last = (_dev(t, "weights") for t in (w_tttf, b_tttf, w_last, b_last))
    n, _, h, w = lr.shape
    H, W = 8 * h, 8 * w
    if tuple(state.shape) != (n, 4, H, W) or tuple(x_hr.shape) != (n, 4, H, W):
        raise ValueError("state / x_hr must be [n,4,8h,8w]")
    m8 = (mask != 0 if mask.dtype != torch.bool else mask).to(state.device).expand(n, 1, H, W).contiguous().view(torch.uint8)
    L = _lib.lib()
    ws = torch.empty(L.crfp_fovea_head_workspace_bytes(n, h, w), dtype=torch.uint8, device=state.device)
    new_state = torch.empty_like(state)
    out = torch.empty((n, 1 if y_only else 3, H, W), dtype=torch.float32, device=state.device)
    with _on(state, x_hr, lr, w_tttf, b_tttf, w_last, b_last):
        _lib.check(L.crfp_fovea_head_f32(state.data_ptr(), x_hr.data_ptr(), m8.data_ptr(), lr.data_ptr(), w_tttf.data_ptr(),
                                         b_tttf.data_ptr(), w_last.data_ptr(), b_last.data_ptr(), new_state.data_ptr(),
                                         out.data_ptr(), n, h, w, int(bool(y_only)), ws.data_ptr(), ws.numel(), _stream()),
                   "crfp_fovea_head_f32")
    return new_state, out


def upsample_bilinear_ac(x, scale_factor, mul=1.0):
    """F.interpolate(x, scale_factor=.., mode='bilinear', align_corners=True) * mul (SPyNet's flow upsampling)."""
    x = _dev(x, "x")
    n, c, h, w = x.shape
    oh, ow = int(h * scale_factor), int(w * scale_factor)
    out = torch.empty((n, c, oh, ow), dtype=torch.float32, device=x.device)
    with _on(x):
        _lib.check(_lib.lib().crfp_upsample_bilinear_ac_f32(x.data_ptr(), out.data_ptr(), n, c, h, w, oh, ow, float(mul), _stream()),
                   "crfp_upsample_bilinear_ac_f32")
    return out


def convkxk(x, weight, bias, pre_relu=False):
    """k x k stride-1 'same' convolution (k in 3, 5, 7), optional ReLU on the input (reference `conv`, model/CRFP.py:145-152)."""
    x, weight, bias = _dev(x, "x"), _dev(weight, "weight"), _dev(bias, "bias")
    n, cin, h, w = x.shape
    cout, cin_w, k, k2 = weight.shape
    assert cin_w == cin and k == k2
    out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    with _on(x, weight, bias):
        _lib.check(_lib.lib().crfp_convkxk_f32(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(), n, cin, cout, h, w,
                                               k, int(bool(pre_relu)), _stream()), "crfp_convkxk_f32")
    return out


def spynet_forward(params, ref, supp):
    """SPyNet.forward(ref, supp) (reference model/CRFP.py:698-741) in one native call; params = the 60 conv tensors in
    state_dict order."""
    import ctypes as C
    ref, supp = _dev(ref, "ref"), _dev(supp, "supp")
    if len(params) != 60:
        raise ValueError(f"SPyNet has 60 conv parameters, got {len(params)}")
    keep = [_dev(p.detach(), "params") for p in params]
    n, c, h, w = ref.shape
    assert c == 3 and tuple(supp.shape) == tuple(ref.shape)
    L = _lib.lib()
    ptrs = (C.c_void_p * 60)(*[t.data_ptr() for t in keep])
    ws = _ws(L.crfp_spynet_workspace_bytes(n, h, w), ref.device)
    flow = torch.empty((n, 2, h, w), dtype=torch.float32, device=ref.device)
    with _on(ref, supp, *keep):
        _lib.check(L.crfp_spynet_forward(ptrs, ref.data_ptr(), supp.data_ptr(), flow.data_ptr(), n, h, w, ws.data_ptr(), ws.numel(),
                                         _stream()), "crfp_spynet_forward")
    return flow
