"""frames/s of the one-call engine schedules of the reference's three recurrent wirings (CRFP_DSV, CRFP_simple, CRFP), mid_channels 32 and 16
(16 runs embedded in the 32-channel schedule), at BASELINE configs[1]'s shape (7 x 180 x 320 -> 1440 x 2560, one clip per call), beside the per-operator composition the ablation models ran through before round 6.
usage: python tools/ablation_fps.py [steps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from crfp_amd import synth  # noqa: E402
from crfp_amd.model import CRFP  # noqa: E402


def rate(fn, steps, frames):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return steps * frames / (time.perf_counter() - t0)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    lrs, fvs, mks = (torch.from_numpy(a).to(dev) for a in synth.make_clip(3, 1, 7, 180, 320, fv_size=96))
    for cls, mid in (("CRFP_DSV", 32), ("CRFP_simple", 32), ("CRFP", 32), ("CRFP_DSV", 16), ("CRFP_simple", 16), ("CRFP", 16)):
        torch.manual_seed(1)
        m = getattr(CRFP, cls)(dev, mid_channels=mid).to(dev).eval()
        row = {}
        with torch.no_grad():
            for storage in ("f32", "bf16"):
                m.storage = storage
                row[f"engine_{storage}"] = round(rate(lambda: m(lrs, fvs, mks), steps, 7), 1)
            m.storage = "f32"
            row["composed_f32"] = round(rate(lambda: m.forward_composed(lrs, fvs, mks), max(2, steps // 4), 7), 1)
        print(cls, f"mid_channels={mid}", row, flush=True)


if __name__ == "__main__":
    main()
