#!/usr/bin/env python3
"""Golden for the regional-DCN benchmark wiring by running the REFERENCE class
``model/CRFP_runtime.py::MRCF_simple_v18`` on the CPU in the build container:

    python tests/golden/make_runtime_golden.py

That module cannot be imported as it is on a CPU-only torch: at import time it moves a 1080 x 1920 pixel grid to
'cuda:0' (:59-61) and imports the absent ``memory_profiler`` (:6) and ``dcn_v2`` (:7); its forward brackets every stage
with ``torch.cuda`` events (:8483-8652).  Here, for the duration of the import / call only: ``Tensor.to`` ignores CUDA
targets, ``torch.cuda.synchronize`` is a no-op, the module's two event objects are replaced by dummies,
``memory_profiler.profile`` is the identity decorator and ``dcn_v2.DCNv2`` evaluates oracle.dcnv2 (as in make_golden.py).
None of that touches the arithmetic of the wiring, which is what the golden pins.  Data only is written."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden  # noqa: E402
from crfp_amd import synth  # noqa: E402


class _DummyEvent:
    def record(self):
        pass

    def elapsed_time(self, other):
        return 0.0


def import_reference_runtime():
    make_golden.inject_stubs()
    mp = types.ModuleType("memory_profiler")
    mp.profile = lambda f=None, **k: f if f is not None else (lambda g: g)
    sys.modules["memory_profiler"] = mp
    sys.path.insert(0, make_golden.REF)
    orig_to = torch.Tensor.to

    def cpu_to(self, *a, **k):
        a = tuple(x for x in a if not (isinstance(x, torch.device) and x.type == "cuda") and not (isinstance(x, str) and x.startswith("cuda")))
        if isinstance(k.get("device"), (torch.device, str)) and str(k["device"]).startswith("cuda"):
            k.pop("device")
        return orig_to(self, *a, **k) if (a or k) else self

    torch.Tensor.to = cpu_to
    try:
        from model import CRFP_runtime as ref
    finally:
        torch.Tensor.to = orig_to
    ref.start, ref.end = _DummyEvent(), _DummyEvent()
    torch.cuda.synchronize = lambda *a, **k: None
    return ref


def main():
    ref = import_reference_runtime()
    torch.set_grad_enabled(False)
    SEED = 17
    net = ref.MRCF_simple_v18(mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True, split_ratio=3,
                              spynet_pretrained='pretrained_models/fnet.pth', device=torch.device("cpu"))
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.make_state_dict_like(shapes, SEED)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    net.eval()
    t, h, w, fv, warp = 3, 24, 40, 64, (128, 192)
    lrs = torch.from_numpy(synth.make_clip(400, 1, t, h, w, fv_size=32)[0])
    rs = np.random.RandomState(6)
    fvs = torch.from_numpy(rs.uniform(0, 1, (1, t, 3, fv, fv)).astype(np.float32))
    out = net(lrs, fvs, warp_size=warp)
    np.savez_compressed(os.path.join(HERE, "runtime_small.npz"), weights_seed=np.int64(SEED),
                        weights_sha256=np.array(synth.state_dict_digest(sd)), keys=np.array(list(shapes)),
                        shapes=np.array([str(s) for s in shapes.values()]), lrs=lrs.numpy(), fvs=fvs.numpy(),
                        warp=np.array(warp), out=out.numpy())
    print(out.shape, float(out.mean()), float(out.abs().max()))
    # round 4: the calls the one-call engine does not take and the module mirror routes through its per-operator composition --
    # a model built with offset_prop=False (no conv_fuse / dcn_3.upsample parameters) and the default, oversized warp_size on a small
    # frame (the reference clamps its window by slicing, :8487,8548)
    extra = {}
    net2 = ref.MRCF_simple_v18(mid_channels=32, y_only=False, hr_dcn=True, offset_prop=False, split_ratio=3,
                               spynet_pretrained='pretrained_models/fnet.pth', device=torch.device("cpu"))
    shapes2 = {k: tuple(v.shape) for k, v in net2.state_dict().items()}
    sd2 = synth.make_state_dict_like(shapes2, SEED + 1)
    net2.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd2.items()}, strict=True)
    net2.eval()
    lrs2 = torch.from_numpy(synth.make_clip(401, 1, 2, 16, 24, fv_size=32)[0])
    fvs2 = torch.from_numpy(rs.uniform(0, 1, (1, 2, 3, 48, 48)).astype(np.float32))
    extra["noprop.keys"] = np.array([f"{k}:{','.join(map(str, v))}" for k, v in shapes2.items()])
    extra["noprop.weights_seed"] = np.int64(SEED + 1)
    extra["noprop.lrs"], extra["noprop.fvs"], extra["noprop.warp"] = lrs2.numpy(), fvs2.numpy(), np.array((96, 128))
    extra["noprop.out"] = net2(lrs2, fvs2, warp_size=(96, 128)).numpy()
    extra["oversize.lrs"], extra["oversize.fvs"] = lrs2.numpy(), fvs2.numpy()
    extra["oversize.out"] = net(lrs2, fvs2).numpy()          # warp_size = the signature's default (1080, 1920) on a 128 x 192 frame
    np.savez_compressed(os.path.join(HERE, "runtime_flags.npz"), **extra)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in extra.items()})


if __name__ == "__main__":
    main()
