#!/usr/bin/env python3
"""Golden for the REDS eval data path: builds a tiny synthetic REDS-shaped PNG tree (4 clips x 5 frames, 64x96 ground
truth, x8 low resolution), runs the REFERENCE's dataset.reds.EvalSet on it (imported from /root/reference with the
torchvision import stubbed) and stores the tree's pixels plus the reference's batch dict for two items.
Run in the build container only:  python tests/golden/make_reds_golden.py"""
import os
import sys
import tempfile
import types

import numpy as np
import PIL.Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def make_tree(root, rs):
    gt_root = os.path.join(root, "REDS_sharp")
    lr_root = gt_root.replace("_sharp", "_sharp_BI_x8")
    frames = {}
    for clip in ("000", "001", "006", "017"):
        for tree, (h, w) in ((gt_root, (64, 96)), (lr_root, (8, 12))):
            d = os.path.join(tree, "val/val/val_sharp", clip)
            os.makedirs(d)
            for i in range(5):
                img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
                PIL.Image.fromarray(img).save(os.path.join(d, f"{i:08d}.png"))
                frames[f"{'gt' if tree == gt_root else 'lr'}_{clip}_{i}"] = img
    return gt_root, frames


def main():
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})
    sys.path.insert(0, REF)
    from dataset import reds as ref_reds
    rs = np.random.RandomState(2024)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        gt_root, frames = make_tree(td, rs)
        out.update(frames)
        args = types.SimpleNamespace(dataset_dir=gt_root, scale=8, N_frames=3, GT_size=64, FV_size=16)
        ds = ref_reds.EvalSet(args)
        out["n_items"] = np.int64(len(ds))
        for idx in (0, 7):
            item = ds[idx]
            for k, v in item.items():
                out[f"item{idx}_{k}"] = v.numpy()
            out[f"item{idx}_first_gt_file"] = np.array(os.path.relpath(ds.GT_imgfiles[idx][0], gt_root))
    # the schedule alone at the benchmark geometries (720x1280 REDS ground truth, 96-pixel fovea; 1440x2560)
    for (H, W, FV, n) in ((720, 1280, 96, 7), (1440, 2560, 96, 7), (64, 96, 16, 30)):
        imgs = [np.zeros((H, W, 3), np.uint8)] * n
        _, _, fv_sp = ref_reds.fovea_generator(imgs, method="Evenscan", FV_HW=(FV, FV))
        out[f"evenscan_{H}x{W}_{FV}_{n}"] = fv_sp.numpy()
    np.savez_compressed(os.path.join(HERE, "reds_evalset.npz"), **out)
    print("reds_evalset.npz:", len(out), "arrays,", out["n_items"], "items")


if __name__ == "__main__":
    main()
