"""Seeded synthetic weights and clips for the CRFP_DSV hot path.

No pretrained weights or REDS data exist in either box (reference
``.MISSING_LARGE_BLOBS``), so parity fixtures, GPU tests, ``smoke()`` and
``bench.py`` all share this generator.  Everything is drawn from
``numpy.random.RandomState`` (bit-stable stream by numpy policy), so a fixture
only has to store a seed plus a checksum instead of 9 MB of weights.

The key/shape table restates the reference's ``state_dict`` for
``CRFP_DSV(mid_channels=32)`` (reference ``model/CRFP.py:1388-1481`` and
``model/LTE.py:34-51,100-117``); ``tests/golden/make_golden.py`` loads the
result into the imported reference with ``strict=True``, which pins it.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict

import numpy as np

FNET_LAYERS = [  # (key stem, cout, cin)   reference model/CRFP.py:747-795
    ("encoder1.0", 32, 6), ("encoder1.2", 32, 32),
    ("encoder2.0", 64, 32), ("encoder2.2", 64, 64),
    ("encoder3.0", 128, 64), ("encoder3.2", 128, 128),
    ("decoder1.0", 256, 128), ("decoder1.2", 256, 256),
    ("decoder2.0", 128, 256), ("decoder2.2", 128, 128),
    ("decoder3.0", 64, 128), ("decoder3.2", 64, 64),
    ("flow.0", 32, 64), ("flow.2", 2, 32),
]


def conv_spec(mid: int = 32, y_only: bool = False):
    """Ordered (key stem, cout, cin) for every 3x3 conv of CRFP_DSV (hr_dcn, offset_prop)."""
    last = mid // 8
    dg, K = 8, 9
    prop = (mid * 3) // 4
    spec = [("spynet." + k, o, i) for k, o, i in FNET_LAYERS]
    for lvl in range(3):
        p = f"dcn_{lvl}."
        if lvl > 0:
            spec.append((p + "conv_fuse", mid, 2 * mid))
        spec += [(p + "dcn_block.0", mid, 2 * mid + 2), (p + "dcn_block.2", mid, mid),
                 (p + "dcn_offset", dg * 2 * K, mid), (p + "dcn_mask", dg * K, mid),
                 (p + "dcn", mid, mid)]
    p = "dcn_3."
    spec += [(p + "upsample.upsample_conv", last * 16, last * 8), (p + "conv_fuse", last, 2 * last),
             (p + "dcn_block.0", last, 2 * last + 2), (p + "dcn_block.2", last, last),
             (p + "dcn_offset", 2, last), (p + "dcn_mask", 1, last), (p + "dcn", last, last)]
    spec += [("encoder_lr.slice1.0", mid, 3), ("encoder_lr.slice1.2", mid, mid),
             ("encoder_hr.slice1.0", last, 6), ("encoder_hr.slice1.2", last, last),
             ("conv_tttf", last, 2 * last)]
    for lvl in range(3):
        p = f"forward_resblocks_{lvl}.main."
        spec += [(p + "0", mid, 2 * mid), (p + "2.0.conv1", mid, mid), (p + "2.0.conv2", mid, mid)]
    p = "forward_resblocks_3.main."
    spec += [(p + "0", last, 2 * last), (p + "2.0.conv1", last, last), (p + "2.0.conv2", last, last)]
    spec += [("downsample.downsample_conv", mid, last * 16),
             ("upsample.upsample_conv", prop * 4, mid),
             ("upsample_post.upsample_conv", last * 16, prop),
             ("conv_last", 1 if y_only else 3, last)]
    return spec


def state_dict_keys(mid: int = 32, y_only: bool = False):
    keys = []
    for stem, _, _ in conv_spec(mid, y_only):
        keys += [stem + ".weight", stem + ".bias"]
    return keys


def make_state_dict(seed: int = 0, mid: int = 32, y_only: bool = False, offset_std: float = None) -> "OrderedDict[str, np.ndarray]":
    """Numerically healthy random weights: activations stay O(1) over a clip, FNet emits
    flows of a few LR pixels, DCN offsets/masks are non-degenerate (the reference's
    zero-init of dcn_offset/dcn_mask, model/CRFP.py:354-358, would collapse DCN to 0.5*warp).
    Default ("stress"): dcn_offset weights of std 0.053 (32-channel levels), i.e. residual offsets that fill the whole
    +-10 px range of 10*tanh -- every fixture, test and the headline benchmark use it.  ``offset_std=0.02`` rescales
    the dcn_offset / dcn_mask weights to the N(0, 0.02) SURVEY.md section 8d prescribes (residuals of a few pixels, closer
    to a trained network); same random stream, so everything else is identical."""
    rs = np.random.RandomState(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for stem, cout, cin in conv_spec(mid, y_only):
        fan_in = cin * 9
        std = np.sqrt(2.0 / fan_in) * 0.8
        w = (rs.standard_normal((cout, cin, 3, 3)) * std).astype(np.float32)
        b = (rs.standard_normal((cout,)) * 0.02).astype(np.float32)
        if stem.endswith("conv1") or stem.endswith("conv2"):
            w *= 0.3  # residual branches stay small (reference scales them 0.1 at init)
        if stem == "spynet.flow.2":
            w *= 0.12
            b *= 0.1
        if stem.endswith("dcn_offset"):
            w *= 0.8 if offset_std is None else offset_std / (std * 1.0)
        if stem.endswith("dcn_mask"):
            w *= 1.5 if offset_std is None else offset_std / (std * 1.0)
        if stem.endswith(".dcn"):
            w *= 0.25
            c = min(cout, cin)
            w[np.arange(c), np.arange(c), 1, 1] += 1.0
        sd[stem + ".weight"] = np.ascontiguousarray(w)
        sd[stem + ".bias"] = np.ascontiguousarray(b)
    return sd


def make_state_dict_like(shapes, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Seeded, numerically healthy weights for any conv-only state_dict given as ``{key: shape}`` in state_dict order (used
    for the regional-DCN runtime wiring, whose key table comes from the module itself): the scaling rules of
    ``make_state_dict`` applied by key name."""
    rs = np.random.RandomState(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape in shapes.items():
        shape = tuple(int(v) for v in shape)
        stem = key.rsplit(".", 1)[0]
        if key.endswith(".bias"):
            b = rs.standard_normal(shape) * 0.02
            if stem == "spynet.flow.2":
                b *= 0.1
            sd[key] = b.astype(np.float32)
            continue
        cout, cin = shape[0], shape[1]
        w = rs.standard_normal(shape) * (np.sqrt(2.0 / (cin * 9)) * 0.8)
        if stem.endswith("conv1") or stem.endswith("conv2"):
            w *= 0.3 if ".main." in stem else 1.0     # residual branches small; the v2 blocks' input convs are full size
        if stem == "spynet.flow.2":
            w *= 0.12
        if stem.endswith("dcn_offset"):
            w *= 0.8
        if stem.endswith("dcn_mask"):
            w *= 1.5
        if stem.endswith(".dcn"):
            w *= 0.25
            c = min(cout, cin)
            w[np.arange(c), np.arange(c), 1, 1] += 1.0
        sd[key] = np.ascontiguousarray(w.astype(np.float32))
    return sd


SPYNET_CHANNELS = (8, 32, 64, 32, 16, 2)   # reference model/CRFP.py:693-734


def make_spynet_state_dict(seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Seeded weights for the reference's SPyNet (model/CRFP.py:554-741): keys in state_dict order (buffers ``mean`` /
    ``std`` first, then ``basic_module.{L}.basic_module.{j}.conv.{weight,bias}``).  Scaled so that every pyramid level
    adds a flow residual of a fraction of a pixel to a few pixels."""
    rs = np.random.RandomState(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    sd["mean"] = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
    sd["std"] = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
    for lvl in range(6):
        for j in range(5):
            cin, cout = SPYNET_CHANNELS[j], SPYNET_CHANNELS[j + 1]
            std = np.sqrt(2.0 / (cin * 49)) * (0.5 if j == 4 else 0.9)
            sd[f"basic_module.{lvl}.basic_module.{j}.conv.weight"] = (rs.standard_normal((cout, cin, 7, 7)) * std).astype(np.float32)
            sd[f"basic_module.{lvl}.basic_module.{j}.conv.bias"] = (rs.standard_normal((cout,)) * 0.05).astype(np.float32)
    return sd


def state_dict_digest(sd) -> str:
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k], dtype=np.float32).tobytes())
    return h.hexdigest()


def _smooth_field(rs, c, H, W, n_waves=10):
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    img = np.zeros((c, H, W), np.float64)
    for ch in range(c):
        for _ in range(n_waves):
            fy, fx = rs.uniform(-0.08, 0.08, 2)
            ph = rs.uniform(0, 2 * np.pi)
            amp = rs.uniform(0.03, 0.12)
            img[ch] += amp * np.cos(2 * np.pi * (fy * yy + fx * xx) + ph)
        img[ch] += 0.5
    return img


def make_clip(seed: int, n: int, t: int, h: int, w: int, fv_size: int = 96, sigma_t: float = 10.0,
              max_shift_hr: int = 16):
    """Synthetic foveated clip: an HR scene translated by a few HR pixels
    per frame, box-downsampled x8 to the LR frames; ``fvs`` is the HR frame inside a
    ``fv_size`` window whose centre follows a gaze walk N(centre, sigma_t) (reference
    test_video.py:309-343), zero elsewhere (reference dataset/reds.py:196-203); ``mks`` is the
    bool window mask.  Returns float32 lrs[n,t,3,h,w], fvs[n,t,3,8h,8w], bool mks[n,t,1,8h,8w]."""
    rs = np.random.RandomState(seed)
    H, W = 8 * h, 8 * w
    fv = min(fv_size, H, W)
    lrs = np.zeros((n, t, 3, h, w), np.float32)
    fvs = np.zeros((n, t, 3, H, W), np.float32)
    mks = np.zeros((n, t, 1, H, W), np.bool_)
    for b in range(n):
        scene = _smooth_field(rs, 3, H, W)
        scene += rs.uniform(-0.05, 0.05, scene.shape)
        sy = sx = 0
        for i in range(t):
            sy += int(rs.randint(-max_shift_hr, max_shift_hr + 1))
            sx += int(rs.randint(-max_shift_hr, max_shift_hr + 1))
            hr = np.roll(scene, (sy, sx), axis=(1, 2))
            lr = hr.reshape(3, h, 8, w, 8).mean(axis=(2, 4))
            lr = lr + rs.uniform(-0.02, 0.02, lr.shape)
            lrs[b, i] = np.clip(lr, 0, 1).astype(np.float32)
            cy = int(np.clip(rs.normal(H / 2, sigma_t), fv // 2, H - (fv - fv // 2)))
            cx = int(np.clip(rs.normal(W / 2, sigma_t), fv // 2, W - (fv - fv // 2)))
            y0, x0 = cy - fv // 2, cx - fv // 2
            mks[b, i, 0, y0:y0 + fv, x0:x0 + fv] = True
            fvs[b, i, :, y0:y0 + fv, x0:x0 + fv] = np.clip(hr[:, y0:y0 + fv, x0:x0 + fv], 0, 1)
    return lrs, fvs, mks
