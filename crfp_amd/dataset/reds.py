"""REDS eval data path: counterpart of the reference's ``dataset.reds.EvalSet`` and the 'Evenscan' fovea schedule
(dataset/reds.py:158-168 schedule, :186-227 mask / Ref construction, :339-427 EvalSet).

On-disk layout (reference :347-359): ``<dataset_dir>/val/val/val_sharp/<clip>/*.png`` for the ground truth and the
same tree under ``<dataset_dir with '_sharp' -> '_sharp_BI_x8'>`` (x8; '_sharp_BI' for x4) for the low-resolution
frames, clips 000 / 001 / 006 / 017 (REDS4).  Items are sliding windows of ``N_frames`` consecutive frames of a clip.
Each item holds, as the reference's batch dict does (keys and dtypes identical):
  LR [N,3,h,w], LR_sr [N,3,H,W] (PIL bicubic of LR), HR [N,3,H,W], Ref = HR inside the fovea window and 0 outside,
  Ref_sp bool [N,1,H,W] = the window, FV_sp [N,2] = (y, x) window origins; values /255 as float32.
Pure host code (PIL + NumPy): it feeds ``crfp_amd.evalrig.eval_clip`` when a REDS tree is mounted; nothing here runs on
the GPU box's synthetic benchmark path.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np
import torch

REDS4 = ("000", "001", "006", "017")


def evenscan(len_sp: int, GT_H: int, GT_W: int, FV_H: int, FV_W: int, idx: int = 20) -> List[List[int]]:
    """The 'Evenscan' window origins (reference :158-168): the frame is cut into N_H x N_W cells that each hold one
    fovea window, visited row-major starting at cell `idx`; the window is centred in its cell."""
    N_H, N_W = GT_H // FV_H, GT_W // FV_W
    SP_H, SP_W = GT_H / N_H, GT_W / N_W
    out = []
    for i in range(idx, idx + len_sp):
        x_i = i % N_W
        y_i = (i // N_W) % N_H
        out.append([int((1 + y_i) * SP_H - (SP_H + FV_H) / 2), int((1 + x_i) * SP_W - (SP_W + FV_W) / 2)])
    return out


def fovea_generator(GT_imgs: Sequence[np.ndarray], method: str = "Evenscan", FV_HW: Tuple[int, int] = (32, 32)):
    """NumPy branch of the reference's fovea_generator (:17, :212-227) for the eval schedule: per frame the [H,W,1]
    window mask, Ref = GT * mask (float64, as NumPy promotes it), and the origins as a LongTensor."""
    if method != "Evenscan":
        raise NotImplementedError("the eval path uses method='Evenscan' (dataset/reds.py:391); the training-time "
                                  "schedules are out of scope")
    GT_H, GT_W, _ = GT_imgs[0].shape
    FV_H, FV_W = FV_HW
    fv_sp = evenscan(len(GT_imgs), GT_H, GT_W, FV_H, FV_W)
    FV_imgs, Ref_sps = [], []
    for t, img in enumerate(GT_imgs):
        H, W, _ = img.shape
        Ref_sp = np.zeros((H, W, 1))
        Ref_sp[fv_sp[t][0]:fv_sp[t][0] + FV_H, fv_sp[t][1]:fv_sp[t][1] + FV_W, :] = 1
        FV_imgs.append(img * Ref_sp)
        Ref_sps.append(Ref_sp)
    return FV_imgs, Ref_sps, torch.tensor(fv_sp)


class EvalSet(torch.utils.data.Dataset):
    """``args`` needs dataset_dir, scale, N_frames, GT_size (unused on this path, kept for parity), FV_size."""

    def __init__(self, args, clips: Sequence[str] = REDS4):
        super().__init__()
        self.args = args
        scale = args.scale
        if scale == 8:
            LR_root = args.dataset_dir.replace("_sharp", "_sharp_BI_x8")
        elif scale == 4:
            LR_root = args.dataset_dir.replace("_sharp", "_sharp_BI")
        else:
            raise ValueError(f"scale {scale}: the reference ships x8 and x4 trees only")
        self.GT_dir_list = sorted(os.path.join(args.dataset_dir, "val/val/val_sharp", name) for name in clips)
        self.LR_dir_list = sorted(os.path.join(LR_root, "val/val/val_sharp", name) for name in clips)
        N = args.N_frames
        self.GT_imgfiles = self._windows(self.GT_dir_list, N)
        self.LR_imgfiles = self._windows(self.LR_dir_list, N)
        if len(self.GT_imgfiles) != len(self.LR_imgfiles):
            raise RuntimeError(f"{len(self.GT_imgfiles)} ground-truth windows vs {len(self.LR_imgfiles)} low-resolution windows")

    @staticmethod
    def _windows(dirs, N):
        out = []
        for d in dirs:
            files = sorted(os.listdir(d))
            for i in range(0, len(files) - N + 1):
                out.append([os.path.join(d, f) for f in files[i:i + N]])
        return out

    def __len__(self):
        return len(self.GT_imgfiles)

    def __getitem__(self, index):
        import PIL.Image
        FV_size = self.args.FV_size
        GT_imgs = [np.array(PIL.Image.open(f)) for f in self.GT_imgfiles[index]]
        H_, W_, _ = GT_imgs[0].shape
        LR_imgs = [np.array(PIL.Image.open(f)) for f in self.LR_imgfiles[index]]
        LR_sr_imgs = [np.array(PIL.Image.fromarray(img).resize((W_, H_), PIL.Image.BICUBIC)) for img in LR_imgs]
        Ref, Ref_sp, fv_sp = fovea_generator(GT_imgs, method="Evenscan", FV_HW=(FV_size, FV_size))

        def chw(stack, dtype=np.float32, div=255.0):
            a = np.stack(stack, axis=0).astype(dtype)
            if div:
                a = a / div
            return torch.from_numpy(np.ascontiguousarray(np.transpose(a, (0, 3, 1, 2))))

        return {"LR": chw(LR_imgs).float(), "LR_sr": chw(LR_sr_imgs).float(), "HR": chw(GT_imgs).float(),
                "Ref": chw(Ref).float(), "Ref_sp": chw(Ref_sp, np.bool_, 0), "FV_sp": fv_sp}
