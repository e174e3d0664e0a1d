import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the torch-CPU oracle: one thread per core this process may really use (the GPU box shows 256 host cores to a 16-core
    # allotment; torch's default of one thread per visible core makes the small oracle convs several times slower)
    try:
        import torch
        torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    except Exception:
        pass


@pytest.fixture(scope="session")
def ops_golden():
    return dict(np.load(os.path.join(GOLDEN, "ops_small.npz")))


@pytest.fixture(scope="session")
def weights_np(ops_golden):
    from crfp_amd import synth
    sd = synth.make_state_dict(int(ops_golden["weights_seed"]))
    assert synth.state_dict_digest(sd) == str(ops_golden["weights_sha256"]), "synthetic weight stream drifted"
    return sd


@pytest.fixture(scope="session")
def oracle_c_lib():
    """Build (if needed) and load our plain-C DCNv2 / flow_warp restatement."""
    import ctypes
    so = os.path.join(ROOT, "oracle", "_build", "libdcnv2_ref.so")
    src = os.path.join(ROOT, "oracle", "dcnv2_ref.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return ctypes.CDLL(so)
