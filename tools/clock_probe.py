"""Does the engine's speed depend on the VALUES it computes on?  Same library, same launches, same bytes: CRFP_DSV at mid_channels 32 against
mid_channels 16 embedded in the 32-channel schedule (half of every wide tensor exact zeros), fp32 and bf16 storage, while a sampler thread reads the
GPU's shader clock and power from sysfs (hwmon freq1_input / power1_average|input) every 10 ms.
usage: python tools/clock_probe.py [seconds per case]"""
import glob
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crfp_amd import synth  # noqa: E402
from crfp_amd.model import CRFP  # noqa: E402


def sensors():
    out = {}
    # the hwmon directory of THE device torch runs on (a node exposes every GPU's sysfs entries, also those this process cannot use)
    pr = torch.cuda.get_device_properties(0)
    addr = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    cands = glob.glob(f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*")
    print("device", pr.name, "pci", addr, "hwmon", cands, flush=True)
    for hw in cands:
        for key, names in (("sclk_mhz", ("freq1_input",)), ("power_w", ("power1_average", "power1_input")), ("temp_c", ("temp2_input", "temp1_input"))):
            for n in names:
                p = os.path.join(hw, n)
                if os.path.exists(p) and key not in out:
                    out[key] = p
    return out


SCALE = {"sclk_mhz": 1e-6, "power_w": 1e-6, "temp_c": 1e-3}


class Sampler(threading.Thread):
    def __init__(self, paths):
        super().__init__(daemon=True)
        self.paths, self.rows, self.stop = paths, [], False

    def run(self):
        while not self.stop:
            row = {}
            for k, p in self.paths.items():
                try:
                    row[k] = float(open(p).read().strip()) * SCALE[k]
                except (OSError, ValueError):
                    pass
            self.rows.append(row)
            time.sleep(0.01)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    dev = torch.device("cuda:0")
    paths = sensors()
    print("sensors:", paths if paths else "none readable (sysfs hwmon not exposed to this user)", flush=True)
    lrs, fvs, mks = (torch.from_numpy(a).to(dev) for a in synth.make_clip(3, 1, 7, 180, 320, fv_size=96))
    for mid in (32, 16, 32, 16):
        for storage in ("f32", "bf16"):
            torch.manual_seed(1)
            m = CRFP.CRFP_DSV(dev, mid_channels=mid).to(dev).eval()
            m.storage = storage
            with torch.no_grad():
                for _ in range(3):
                    m(lrs, fvs, mks)
                torch.cuda.synchronize()
                s = Sampler(paths)
                s.start()
                t0 = time.perf_counter()
                n = 0
                while time.perf_counter() - t0 < secs:
                    for _ in range(10):
                        m(lrs, fvs, mks)
                    torch.cuda.synchronize()
                    n += 10
                dt = time.perf_counter() - t0
                s.stop = True
                s.join()
            rows = s.rows[len(s.rows) // 4:]      # the last three quarters: past the ramp
            avg = {k: round(sum(r[k] for r in rows if k in r) / max(1, sum(k in r for r in rows)), 1) for k in paths}
            print(f"mid_channels {mid:2d} {storage:4s}: {7 * n / dt:7.1f} frames/s  {avg}  ({len(rows)} samples)", flush=True)
            del m
            torch.cuda.empty_cache()
            time.sleep(1.0)


if __name__ == "__main__":
    main()
