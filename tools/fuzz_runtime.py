"""Random-geometry check of crfp_rt_forward_clip against oracle/runtime_oracle.py (diagnostic; the fixed cases live in tests/test_gpu_round3.py)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crfp_amd import synth
from crfp_amd.model import MRCF_runtime
from oracle import crfp_oracle as orc, runtime_oracle as ro
dev = torch.device("cuda:0")
m = MRCF_runtime.MRCF_simple_v18(mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True, split_ratio=3, device=dev)
sd = synth.make_state_dict_like({k: tuple(v.shape) for k, v in m.state_dict().items()}, 13)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
m = m.to(dev).eval()
P = orc.load_numpy_state(sd)
rs = np.random.RandomState(0)
worst = 0
with torch.no_grad():
    for case in range(14):
        t = int(rs.randint(1, 4)); h = int(rs.randint(8, 26)); w = int(rs.randint(8, 34))
        wph = 8 * int(rs.randint(8, h + 1)); wpw = 8 * int(rs.randint(8, w + 1))
        fh = int(rs.randint(1, 8 * h + 1)); fw = int(rs.randint(1, 8 * w + 1))
        if case % 3 == 0: fh, fw = min(fh, 40), min(fw, 56)
        l = torch.from_numpy(rs.rand(1, t, 3, h, w).astype(np.float32)); f = torch.from_numpy(rs.rand(1, t, 3, fh, fw).astype(np.float32))
        got = m(l.to(dev), f.to(dev), warp_size=(wph, wpw)).cpu()
        ref = ro.runtime_forward(P, l, f, (wph, wpw))
        d = float((got - ref).abs().max()); worst = max(worst, d)
        print(case, (t, h, w, fh, fw, wph, wpw), f"{d:.2e}", "OVF" if m.engine().overflowed() else "", flush=True)
print("worst", worst)
assert worst < 2e-4
