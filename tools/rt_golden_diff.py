"""max |crfp_rt_forward_clip - reference class output| on tests/golden/runtime_small.npz (diagnostic)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crfp_amd import synth
from crfp_amd.model import MRCF_runtime
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "runtime_small.npz")))
dev = torch.device("cuda:0")
m = MRCF_runtime.MRCF_simple_v18(mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True, split_ratio=3, device=dev)
sd = synth.make_state_dict_like({k: tuple(v.shape) for k, v in m.state_dict().items()}, int(g["weights_seed"]))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
m = m.to(dev).eval()
with torch.no_grad():
    out = m(torch.from_numpy(g["lrs"]).to(dev), torch.from_numpy(g["fvs"]).to(dev), warp_size=tuple(int(v) for v in g["warp"])).cpu().numpy()
print("max abs diff vs the reference class's output:", float(np.abs(out - g["out"]).max()), "output range", float(g["out"].min()), float(g["out"].max()))
