"""Generates tests/golden/gaze_masks.npz by EXECUTING the mask / trajectory statements of the reference's video rig
(/root/reference/test_video.py, the per-frame loop at :303-375) -- they are inline script code, not importable functions, so this
script reads the reference file where it lies, keeps the statements that build the gaze trajectory and the masks (drops the model
call, the metrics and the image I/O) and runs them on a small synthetic video.  Nothing of the reference is copied into the repo:
only the resulting masks are stored.  Run in the build container (needs /root/reference): python tests/golden/make_gaze_golden.py"""
import os
import textwrap

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/test_video.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gaze_masks.npz")
DROP = ("print(", "lr = LR_imgs", "lrsr = LRSR_imgs", "model(", "calc_psnr_and_ssim", "_list.append(psnr", "_list.append(ssim", "if n > 0:",
        "white_paper", "model.eval()", "with torch.no_grad():")


def extract():
    lines = open(REF).read().splitlines()
    i_setup = next(i for i, l in enumerate(lines) if l.strip().startswith("kernel = np.array("))
    i_traj = next(i for i, l in enumerate(lines) if l.strip() == "traj_list = []" and i > i_setup)
    i_loop = next(i for i, l in enumerate(lines) if l.strip() == "for n in range(N):" and i > i_traj)
    i_end = next(i for i, l in enumerate(lines) if "mk_past = torch.sum(torch.cat(mk_list" in l and i > i_loop)
    keep = lambda l: l.strip() and not l.strip().startswith("#") and not any(d in l for d in DROP)
    setup = textwrap.dedent("\n".join(l for l in lines[i_setup:i_traj + 1] if keep(l)))
    body = textwrap.dedent("\n".join(l for l in lines[i_loop + 1:i_end + 1] if keep(l)))
    return setup, body


def run(seed, N, H, W, fv_size, sigma, regional_dcn, rg, fv_start):
    setup, body = extract()
    np.random.seed(seed)
    env = dict(np=np, torch=torch, F=F, device="cpu", N=N, H=H, W=W, C=3, sigma=sigma, fv_size=fv_size, fv_st_idx=[fv_start], v_idx=0,
               rg_w=rg, rg_h=rg, regional_dcn=regional_dcn, GT_imgs=torch.rand(N, 3, H, W))
    exec(setup, env)
    rec = {k: [] for k in ("cur", "mk", "fovea", "outskirt", "past", "fg")}
    for n in range(N):
        env["n"] = n
        past_before = env.get("mk_past")
        exec(body, env)
        rec["cur"].append((env["cur_y"], env["cur_x"]))
        rec["mk"].append(env["mk"].reshape(H, W).bool().numpy())
        rec["fovea"].append(env["mk_fv"].reshape(H, W).bool().numpy())
        rec["outskirt"].append(env["mk_out"].reshape(H, W).bool().numpy())
        rec["past"].append(np.zeros((H, W), bool) if past_before is None else past_before.reshape(H, W).bool().numpy())
        rec["fg"].append(env["fg"].reshape(H, W).bool().numpy())
    return rec


if __name__ == "__main__":
    out = {}
    for tag, kw in (("a", dict(seed=11, N=8, H=160, W=256, fv_size=48, sigma=10.0, regional_dcn=False, rg=0, fv_start=0)),
                    ("b", dict(seed=12, N=8, H=160, W=256, fv_size=32, sigma=14.0, regional_dcn=True, rg=96, fv_start=2))):
        r = run(**kw)
        for k, v in kw.items():
            out[f"{tag}_{k}"] = np.asarray(v)
        out[f"{tag}_cur"] = np.asarray(r["cur"], dtype=np.int64)
        for k in ("mk", "fovea", "outskirt", "past", "fg"):
            out[f"{tag}_{k}"] = np.packbits(np.stack(r[k]), axis=-1)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith("_mk") or k.endswith("_cur")})
