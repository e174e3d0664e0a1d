"""Bisect the engine against the oracle: run a 2-frame clip, fetch named workspace intermediates of frame 1
(crfp_dsv_debug_fetch) and print max / mean |delta| per tensor against the oracle's taps.
  python tools/bisect_engine.py [--storage bf16] [--h 24 --w 40]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CRFP_MASK_GATE", "0")   # dense launches: xin8 / enc_hr0 / x_hr hold values away from the fovea as well (read once by the library)
from crfp_amd import synth  # noqa: E402
from crfp_amd.model import CRFP  # noqa: E402
from oracle import crfp_oracle as orc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--storage", default="f32")
ap.add_argument("--h", type=int, default=24)
ap.add_argument("--w", type=int, default=40)
ap.add_argument("--fv", type=int, default=64)
a = ap.parse_args()
T = torch.from_numpy
dev = torch.device("cuda:0")
sd = synth.make_state_dict(7)
t, h, w = 2, a.h, a.w
lrs, fvs, mks = synth.make_clip(3, 1, t, h, w, fv_size=a.fv)
m = CRFP.CRFP_DSV(device=dev, mid_channels=32)
m.load_state_dict({k: T(v.copy()) for k, v in sd.items()}, strict=True)
m.storage = a.storage
m = m.to(dev).eval()
with torch.no_grad():
    out = m(lrs=T(lrs).to(dev), fvs=T(fvs).to(dev), mks=T(mks).to(dev)).cpu()
    eng = m.engine()
    P = orc.load_numpy_state(sd)
    taps = {}
    ctx = orc.bf16_storage() if a.storage == "bf16" else orc.tapping({})
    if a.storage == "bf16":
        P = orc.bf16_weights(P)
    with ctx, orc.tapping(taps):
        ref = orc.crfp_dsv_forward(P, T(lrs), T(fvs), T(mks))


def show(name, got, want):
    got, want = got.float().cpu(), want.float()
    if got.shape != want.shape:
        print(f"{name:24s} SHAPE {tuple(got.shape)} vs {tuple(want.shape)}")
        return
    d = (got - want).abs()
    print(f"{name:24s} max {float(d.max()):.3e} mean {float(d.mean()):.3e}  (|ref| max {float(want.abs().max()):.2e})")


f = lambda n: eng.debug_fetch(n, t, h, w)  # noqa: E731
# frame 1 lives in parity set 1
show("x_lr (frame 1)", f("x_lr")[1:2], taps["x_lr"])
show("xin8.1", f("xin8.1")[:, [0, 1, 2, 4, 5, 6]], taps["xin8"])
show("enc_hr0.1", f("enc_hr0.1"), taps["enc_hr0"])
show("x_hr.1", f("x_hr.1"), taps["x_hr"])
show("prop0.1", f("prop0.1"), taps["prop0"])
show("flow2.1", f("flow2.1"), taps["flow2"])
show("prev2", f("prev2"), taps["prev2"])
show("prev2w", f("prev2w"), taps["prev2w"])
show("carryw", f("carryw"), taps["carryw"])
show("prevhrw", f("prevhrw"), taps["prevhrw"])
if a.storage == "bf16":   # fp32 build: these hold the producer-split S3 image, not a Q4 tensor
    show("offfeat0 (lvl0 block2)", f("offfeat0"), taps["dcn_0.block2"])
    show("offfeat1 (lvl1 fuse)", f("offfeat1"), taps["dcn_1.fuse"])
    show("offfeat2 (lvl2 fuse)", f("offfeat2"), taps["dcn_2.fuse"])
show("dcn.fa (lvl2 block0)", f("dcn.fa"), taps["dcn_2.block0"])
show("dcn.fb (lvl2 block2)", f("dcn.fb"), taps["dcn_2.block2"])
om = f("offmask")
show("offmask: offsets (lvl2)", om[:, :144], taps["dcn_2.offset"])
show("offmask: masks (lvl2)", om[:, 144:216], taps["dcn_2.mask"])
show("aligned (lvl2)", f("aligned"), taps["dcn_2.aligned"])
show("res.y0/y1 -> prop (res2)", torch.cat([f("prop_a")[:, :24], f("carry")[:, 16:24]], 1), taps["res2"])
show("up", f("up"), taps["up"])
show("poff", f("poff"), taps["dcn_3.pre_offset"])
show("dcn3.g0", f("dcn3.g0"), taps["dcn_3.block0"])
show("dcn3.g1", f("dcn3.g1"), taps["dcn_3.block2"])
show("dcn3.g2", f("dcn3.g2"), taps["dcn_3.fuse"])
o3 = f("om3")
show("om3 offsets", o3[:, :2], taps["dcn_3.offset"][:, :2])
show("om3 mask", o3[:, 2:3], taps["dcn_3.mask"][:, :1])
show("aligned3", f("aligned3"), taps["dcn_3.aligned"])
show("feat", f("feat"), taps["feat"])
show("state_hr", f("state_hr"), taps["state_hr"])
show("out frame 0", out[:, 0], ref[:, 0])
show("out frame 1", out[:, 1], ref[:, 1])
