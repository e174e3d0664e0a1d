"""CPU: crfp_amd.dataset.reds (EvalSet + Evenscan schedule) against the batch dicts the REFERENCE's dataset.reds.EvalSet
produced for the same synthetic REDS-shaped PNG tree (tests/golden/make_reds_golden.py)."""
import os
import types

import numpy as np
import PIL.Image
import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope="module")
def golden():
    return dict(np.load(os.path.join(GOLDEN, "reds_evalset.npz")))


def test_evenscan_schedule(golden):
    from crfp_amd.dataset import reds
    for key in [k for k in golden if k.startswith("evenscan_")]:
        hw, fv, n = key.split("_")[1:]
        H, W = (int(v) for v in hw.split("x"))
        assert reds.evenscan(int(n), H, W, int(fv), int(fv)) == golden[key].tolist()


def test_evalset_items_match_reference(golden, tmp_path):
    from crfp_amd.dataset import reds
    gt_root = str(tmp_path / "REDS_sharp")
    lr_root = gt_root.replace("_sharp", "_sharp_BI_x8")
    for key, img in golden.items():
        if key[:3] in ("gt_", "lr_"):
            kind, clip, i = key.split("_")
            d = os.path.join(gt_root if kind == "gt" else lr_root, "val/val/val_sharp", clip)
            os.makedirs(d, exist_ok=True)
            PIL.Image.fromarray(img).save(os.path.join(d, f"{int(i):08d}.png"))
    args = types.SimpleNamespace(dataset_dir=gt_root, scale=8, N_frames=3, GT_size=64, FV_size=16)
    ds = reds.EvalSet(args)
    assert len(ds) == int(golden["n_items"]) == 4 * (5 - 3 + 1)
    for idx in (0, 7):
        item = ds[idx]
        assert os.path.relpath(ds.GT_imgfiles[idx][0], gt_root) == str(golden[f"item{idx}_first_gt_file"])
        assert sorted(item) == ["FV_sp", "HR", "LR", "LR_sr", "Ref", "Ref_sp"]
        for k, v in item.items():
            ref = golden[f"item{idx}_{k}"]
            assert tuple(v.shape) == ref.shape and v.numpy().dtype == ref.dtype, k
            assert np.array_equal(v.numpy(), ref), k       # same PIL, same NumPy arithmetic: bit for bit
    assert item["Ref_sp"].dtype == torch.bool and item["FV_sp"].dtype == torch.int64
