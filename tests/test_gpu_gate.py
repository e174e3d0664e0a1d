"""Mask-gated launches (the x8 frame stack, encoder_hr and conv_tttf computed only where the fovea mask can select them) against the dense
launches of the same library (CRFP_MASK_GATE=0, read once per process -> two child processes): bit-identical outputs for every mask shape."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_mask_gated_launches_change_no_bit(tmp_path):
    files = {}
    for gate in ("0", "1"):
        files[gate] = str(tmp_path / f"gate{gate}.npz")
        env = dict(os.environ, CRFP_MASK_GATE=gate)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "gate_cases.py"), files[gate]], env=env, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
    dense, gated = np.load(files["0"]), np.load(files["1"])
    assert sorted(dense.files) == sorted(gated.files) and len(dense.files) >= 46
    for k in dense.files:
        a, b = dense[k], gated[k]
        assert np.isfinite(a).all(), k
        assert a.shape == b.shape and np.array_equal(a, b), (k, float(np.abs(a - b).max()))
    # the masks matter: the cases are not all the same picture
    assert not np.array_equal(dense["f32.dsv.none"], dense["f32.dsv.all"])
    assert not np.array_equal(dense["f32.dsv.window"], dense["f32.dsv.moving"])
