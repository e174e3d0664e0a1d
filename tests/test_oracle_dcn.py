"""CPU: anchor the oracle's DCNv2 (third-party op, un-pinned by the reference) three ways:
  1. the reference's own known-answer: an identity-initialised DCN_module returns 0.5*flow_warp
     (reference model/CRFP.py:354-370 with :90-130) -- vectors produced by the imported reference;
  2. agreement with the independent plain-C restatement oracle/dcnv2_ref.c;
  3. algebraic properties of DCNv2: zero offset + unit mask == conv2d, integer offsets == shifted
     conv, linearity in the mask."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import crfp_oracle as orc

T = torch.from_numpy


@pytest.fixture(autouse=True)
def _nograd():
    with torch.no_grad():
        yield


def identity_dcn_weight(C):
    w = torch.zeros(C, C, 3, 3)
    w[torch.arange(C), torch.arange(C), 1, 1] = 1.0
    return w


@pytest.mark.parametrize("tag,C,dg", [("n", 32, 8), ("r", 4, 1)])
def test_identity_dcn_is_half_warp(ops_golden, tag, C, dg):
    g = ops_golden
    pre, flow = T(g[f"kat_{tag}_pre"]), T(g[f"kat_{tag}_flow"])
    # what the reference wiring feeds DCNv2 when offset/mask convs are zero (CRFP.py:337-349)
    K = 9
    offset = flow.flip(1).repeat(1, dg * K, 1, 1)
    mask = torch.full((1, dg * K, pre.shape[2], pre.shape[3]), 0.5)
    out = orc.dcnv2(pre, offset, mask, identity_dcn_weight(C), torch.zeros(C), dg)
    assert float((out - T(g[f"kat_{tag}_dcn"])).abs().max()) == 0.0          # same as the generator's DCN
    assert float((out - T(g[f"kat_{tag}_halfwarp"])).abs().max()) < 2e-5      # == 0.5 * reference flow_warp


def c_dcn(lib, x, off, msk, w, b, dg):
    B, C, H, W = x.shape
    O = w.shape[0]
    out = np.empty((B, O, H, W), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    args = [np.ascontiguousarray(a, dtype=np.float32) for a in (x, off, msk, w, b)]
    rc = lib.dcnv2_ref_forward(*[a.ctypes.data_as(fp) for a in args], out.ctypes.data_as(fp),
                               B, C, O, H, W, 3, 1, 1, dg)
    assert rc == 0
    return out


@pytest.mark.parametrize("C,O,dg,H,W", [(32, 32, 8, 9, 13), (4, 4, 1, 17, 11), (8, 12, 2, 6, 7)])
def test_torch_vs_c_restatement(oracle_c_lib, C, O, dg, H, W):
    rs = np.random.RandomState(C * 100 + H)
    x = rs.standard_normal((2, C, H, W)).astype(np.float32)
    off = rs.uniform(-4, 4, (2, 2 * dg * 9, H, W)).astype(np.float32)
    off[0, :, 0, 0] = -25.0
    off[1, 3, 2, 2] = float(H + 3)
    off[1, 0, 1, 1] = -1.0            # lands exactly on the py == -1 boundary for ky=0,y=1
    msk = rs.uniform(0, 1, (2, dg * 9, H, W)).astype(np.float32)
    w = (rs.standard_normal((O, C, 3, 3)) * 0.2).astype(np.float32)
    b = rs.standard_normal(O).astype(np.float32)
    a = orc.dcnv2(T(x), T(off), T(msk), T(w), T(b), dg).numpy()
    c = c_dcn(oracle_c_lib, x, off, msk, w, b, dg)
    assert float(np.abs(a - c).max()) < 2e-5


def test_zero_offset_is_conv():
    rs = np.random.RandomState(3)
    x = T(rs.standard_normal((1, 8, 10, 12)).astype(np.float32))
    w = T(rs.standard_normal((6, 8, 3, 3)).astype(np.float32))
    b = T(rs.standard_normal(6).astype(np.float32))
    out = orc.dcnv2(x, torch.zeros(1, 2 * 2 * 9, 10, 12), torch.ones(1, 2 * 9, 10, 12), w, b, 2)
    assert float((out - F.conv2d(x, w, b, padding=1)).abs().max()) < 1e-5


def test_integer_offset_is_shifted_conv():
    rs = np.random.RandomState(4)
    x = T(rs.standard_normal((1, 4, 9, 11)).astype(np.float32))
    w = T(rs.standard_normal((4, 4, 3, 3)).astype(np.float32))
    b = torch.zeros(4)
    dy, dx = 2, -3
    off = torch.zeros(1, 18, 9, 11)
    off[:, 0::2] = dy
    off[:, 1::2] = dx
    out = orc.dcnv2(x, off, torch.ones(1, 9, 9, 11), w, b, 1)
    # conv of the image shifted so that sample (y+dy, x+dx) lands on (y, x), zero outside
    xs = torch.zeros(1, 4, 9 + 8, 11 + 8)
    xs[:, :, 4:13, 4:15] = x
    xs = xs[:, :, 4 + dy - 1: 4 + dy - 1 + 11, 4 + dx - 1: 4 + dx - 1 + 13]
    ref = F.conv2d(xs, w, b)
    assert float((out - ref).abs().max()) < 1e-5


def test_mask_linearity():
    rs = np.random.RandomState(5)
    x = T(rs.standard_normal((1, 4, 7, 8)).astype(np.float32))
    w = T(rs.standard_normal((4, 4, 3, 3)).astype(np.float32))
    off = T(rs.uniform(-2, 2, (1, 18, 7, 8)).astype(np.float32))
    m1 = T(rs.uniform(0, 1, (1, 9, 7, 8)).astype(np.float32))
    m2 = T(rs.uniform(0, 1, (1, 9, 7, 8)).astype(np.float32))
    z = torch.zeros(4)
    a = orc.dcnv2(x, off, m1, w, z, 1) + 2.0 * orc.dcnv2(x, off, m2, w, z, 1)
    bth = orc.dcnv2(x, off, m1 + 2.0 * m2, w, z, 1)
    assert float((a - bth).abs().max()) < 1e-4
