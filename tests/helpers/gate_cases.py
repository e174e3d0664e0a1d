"""Outputs of the engines for a set of fovea masks, written to an .npz: run once with CRFP_MASK_GATE=0 (dense launches) and once with the
default (mask-gated launches) by tests/test_gpu_gate.py, which compares the two files bit for bit.
usage: python tests/helpers/gate_cases.py out.npz"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from crfp_amd import synth
from crfp_amd.model import CRFP

dev = torch.device("cuda:0")
T = torch.from_numpy


def masks(t, h, w, seed):
    """[name, mks[1, t, 1, 8h, 8w] bool] -- the shapes a gate must get right: nothing set, everything set, single pixels in the corners and
    on tile seams (tiles are 64 x 16), a window straddling four tiles, scattered pixels, a different mask every frame."""
    H, W = 8 * h, 8 * w
    g = np.random.default_rng(seed)
    z = lambda: np.zeros((1, t, 1, H, W), dtype=bool)   # noqa: E731
    out = [("none", z())]
    m = z(); m[:] = True; out.append(("all", m))
    m = z(); m[..., 0, 0] = True; m[..., H - 1, W - 1] = True; out.append(("corners", m))
    m = z(); m[..., 15, 63] = True; m[..., 16, 64] = True; m[..., 31, 127] = True; out.append(("seams", m))
    m = z(); m[..., 10:40, 50:140] = True; out.append(("window", m))
    m = z(); m[..., g.integers(0, H, 12), g.integers(0, W, 12)] = True; out.append(("scattered", m))
    m = z()
    for i in range(t):
        y, x = int(g.integers(0, H - 24)), int(g.integers(0, W - 24))
        m[0, i, 0, y:y + 24, x:x + 24] = True
    out.append(("moving", m))
    return out


def model(cls, storage, **kw):
    m = cls(device=dev, mid_channels=32, **kw)
    sd = synth.make_state_dict_like({k: tuple(v.shape) for k, v in m.state_dict().items()}, 11)
    m.load_state_dict({k: T(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    m.storage = storage
    return m


res = {}
t, h, w = 3, 24, 40
lrs, fvs, _ = (T(a).to(dev) for a in synth.make_clip(5, 1, t, h, w, fv_size=64))
with torch.no_grad():
    for storage in ("f32", "bf16"):
        for cname, cls in (("dsv", CRFP.CRFP_DSV), ("cra", CRFP.CRFP_DSV_CRA)):
            m = model(cls, storage)
            for name, mk in masks(t, h, w, 3):
                res[f"{storage}.{cname}.{name}"] = m(lrs, fvs, T(mk).to(dev)).cpu().numpy()
            if cname == "dsv":
                ms = masks(t, h, w, 4)
                # lock-step batch: a different mask per clip; single-stream schedule; y_only
                mk3 = torch.cat([T(ms[4][1]), T(ms[0][1]), T(ms[6][1])], 0).to(dev)
                res[f"{storage}.batch"] = m(lrs.expand(3, -1, -1, -1, -1).contiguous(), fvs.expand(3, -1, -1, -1, -1).contiguous(), mk3).cpu().numpy()
                m.engine().single_stream = True
                res[f"{storage}.single_stream"] = m(lrs, fvs, T(ms[6][1]).to(dev)).cpu().numpy()
                m.engine().single_stream = False
                # one frame per call, with and without the resident-inputs promise
                for resident in (False, True):
                    m.clear_states()
                    m.inputs_resident = resident
                    mk = T(ms[6][1]).to(dev).contiguous()
                    torch.cuda.synchronize()
                    res[f"{storage}.stream.{int(resident)}"] = m.forward_stream(lrs, fvs, mk).cpu().numpy()
                m.inputs_resident = False
                res[f"{storage}.y_only"] = model(cls, storage, y_only=True)(lrs, fvs, T(ms[4][1]).to(dev)).cpu().numpy()
    # partial tiles on the right / bottom edge (8h, 8w not multiples of 16 / 64) and two sequences streamed in lock-step with different masks
    for storage in ("f32", "bf16"):
        m = model(CRFP.CRFP_DSV, storage)
        t2, h2, w2 = 2, 25, 36
        lr2, fv2, _ = (T(a).to(dev) for a in synth.make_clip(6, 1, t2, h2, w2, fv_size=64))
        for name, mk in masks(t2, h2, w2, 8):
            if name in ("corners", "window", "moving"):
                res[f"{storage}.edge.{name}"] = m(lr2, fv2, T(mk).to(dev)).cpu().numpy()
        ms = masks(t, h, w, 9)
        mk2 = torch.cat([T(ms[6][1]), T(ms[5][1])], 0).to(dev)
        m.clear_states()
        res[f"{storage}.stream_batch"] = m.forward_stream(lrs.expand(2, -1, -1, -1, -1).contiguous(), fvs.expand(2, -1, -1, -1, -1).contiguous(), mk2).cpu().numpy()
np.savez(sys.argv[1], **res)
print("wrote", len(res), "cases")
