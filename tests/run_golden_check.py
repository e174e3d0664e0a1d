"""Helper for test_alternate_kernel_paths: run one golden clip through the HIP path in a fresh process
(the kernel-selection environment variables are read once per process) and print the max abs error."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crfp_amd import synth  # noqa: E402
from crfp_amd.model import CRFP  # noqa: E402

g = dict(np.load(os.path.join(ROOT, "tests", "golden", "dsv_20x36_t4.npz")))
sd = synth.make_state_dict(int(g["weights_seed"]))
geom = os.environ.get("CRFP_CHECK_GEOM")   # "h,w,t": a synthetic clip of another geometry (DIGEST only; MAXDIFF is then not against a golden)
if geom:
    gh, gw, gt = (int(v) for v in geom.split(","))
    lrs, fvs, mks = synth.make_clip(4321, 1, gt, gh, gw, fv_size=48)
    g["out"] = np.zeros((1, gt, 3, 8 * gh, 8 * gw), np.float32)
else:
    lrs, fvs, mks = synth.make_clip(int(g["clip_seed"]), 1, int(g["t"]), int(g["h"]), int(g["w"]), fv_size=int(g["fv_size"]))
dev = torch.device("cuda:0")
m = CRFP.CRFP_DSV(device=dev, mid_channels=32)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
m = m.to(dev).eval()
if os.environ.get("CRFP_CHECK_STORAGE"):   # bf16: MAXDIFF is then the bf16 rounding noise, DIGEST still identifies the clip
    m.storage = os.environ["CRFP_CHECK_STORAGE"]
out = m(lrs=torch.from_numpy(lrs).to(dev), fvs=torch.from_numpy(fvs).to(dev), mks=torch.from_numpy(mks).to(dev)).cpu().numpy()
print("MAXDIFF %.6e" % float(np.abs(out - g["out"]).max()))
import hashlib  # noqa: E402
print("DIGEST " + hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest())
