#!/usr/bin/env python3
"""Golden vectors for SPyNet by IMPORTING the reference's ``model.CRFP.SPyNet``
(/root/reference/model/CRFP.py:554-741) in the build container:

    python tests/golden/make_spynet_golden.py

Weights: ``crfp_amd.synth.make_spynet_state_dict(seed)`` loaded with strict=True (pins the state_dict key / shape
table); the fixture stores seeds, the two input frames and the reference's flow -- data only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden  # noqa: E402  (stub injection for dcn_v2 / cv2)
from crfp_amd import synth  # noqa: E402


def main():
    make_golden.inject_stubs()
    sys.path.insert(0, make_golden.REF)
    from model import CRFP
    torch.set_grad_enabled(False)
    SEED = 11
    sd = synth.make_spynet_state_dict(SEED)
    net = CRFP.SPyNet(pretrained=None, device=torch.device("cpu"))
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    net.eval()
    out = {"weights_seed": np.int64(SEED), "weights_sha256": np.array(synth.state_dict_digest(sd))}
    for tag, (h, w) in (("a", (40, 72)), ("b", (64, 96))):     # a: resized up to 64 x 96; b: already a multiple of 32
        clip = synth.make_clip(300 + h, 1, 2, h, w, fv_size=32)[0][0]
        ref, supp = torch.from_numpy(clip[1:2]), torch.from_numpy(clip[0:1])
        out[f"{tag}_ref"], out[f"{tag}_supp"] = ref.numpy(), supp.numpy()
        out[f"{tag}_flow"] = net(ref, supp).numpy()
    # SPyNetBasicModule alone (ReLU-before-conv quirk) on a signed input
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.standard_normal((1, 8, 20, 28)).astype(np.float32))
    out["bm_x"], out["bm_y"] = x.numpy(), net.basic_module[3](x).numpy()
    np.savez_compressed(os.path.join(HERE, "spynet_small.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
    print("flow magnitudes:", float(np.abs(out["a_flow"]).max()), float(np.abs(out["b_flow"]).max()))


if __name__ == "__main__":
    main()
