"""Debug aid: T=1 clips on 2 streams; find the first workspace intermediate that differs from the serial run."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CRFP_SIDE_STREAM"] = "0"
from crfp_amd import synth
from crfp_amd.engine import DSVEngine

K, t, h, w = 2, 1, 180, 320
names = ["enc_lr0", "x_lr", "xin8", "enc_hr0", "x_hr", "prop0", "prop_b", "prop_a", "res.y0", "res.y1", "up", "res3.z0",
         "res3.z1", "feat"]
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(7).items()}
d = torch.device("cuda:0")
engs = [DSVEngine(sd, d) for _ in range(K)]
clips = []
for k in range(K):
    lrs, fvs, mks = synth.make_clip(100 + k, 1, t, h, w, fv_size=96, sigma_t=10.0)
    clips.append([torch.from_numpy(x).to(d) for x in (lrs, fvs, mks)])
refs = []
for k in range(K):
    out = engs[k].forward(*clips[k]).clone()
    torch.cuda.synchronize()
    refs.append({n: engs[k].debug_fetch(n, t, h, w).clone() for n in names} | {"out": out})
streams = [torch.cuda.Stream() for _ in range(K)]
first = None
for rep in range(8):
    outs = [None] * K
    for k in range(K):
        with torch.cuda.stream(streams[k]):
            outs[k] = engs[k].forward(*clips[k])
    torch.cuda.synchronize()
    for k in range(K):
        if (outs[k] - refs[k]["out"]).abs().max().item() == 0:
            continue
        print(f"rep {rep} clip {k}: output differs")
        for n in names:
            cur = engs[k].debug_fetch(n, t, h, w)
            dif = (cur - refs[k][n]).abs()
            if dif.max().item() > 0:
                m = dif[0].amax(0)
                nz = torch.nonzero(m > 0)
                chans = torch.nonzero(dif[0].amax(dim=(1, 2)) > 0).flatten().tolist()
                if n == "res3.z0" or n == first:
                    first = n
                    ys, xs = nz[:, 0].tolist(), nz[:, 1].tolist()
                    # cluster into bounding boxes of connected-ish groups (gap > 8 px starts a new one)
                    pts = sorted(zip(ys, xs))
                    boxes = []
                    for y, x in pts:
                        for b in boxes:
                            if b[0] - 4 <= y <= b[1] + 4 and b[2] - 4 <= x <= b[3] + 4:
                                b[0] = min(b[0], y); b[1] = max(b[1], y); b[2] = min(b[2], x); b[3] = max(b[3], x); b[4] += 1
                                break
                        else:
                            boxes.append([y, y, x, x, 1])
                    for b in boxes[:3]:
                        for xx in (b[2], b[2] + 7, b[3]):
                            print("      at", b[0], xx, "ref", [f"{v:.5f}" for v in refs[k][n][0, :, b[0], xx].tolist()],
                                  "got", [f"{v:.5f}" for v in cur[0, :, b[0], xx].tolist()],
                                  "ref(row-1)", [f"{v:.5f}" for v in refs[k][n][0, :, b[0] - 1, xx].tolist()],
                                  "ref(row+1)", [f"{v:.5f}" for v in refs[k][n][0, :, b[0] + 1, xx].tolist()])
                    for b in boxes[:6]:
                        print(f"      box rows {b[0]}..{b[1]} cols {b[2]}..{b[3]} n={b[4]}  (row%16={b[0]%16} col%256={b[2]%256} col%64={b[2]%64})")
                print(f"   {n}: max {dif.max().item():.3e} n={nz.shape[0]} rows {nz[:,0].min().item()}..{nz[:,0].max().item()} "
                      f"cols {nz[:,1].min().item()}..{nz[:,1].max().item()} chans {chans[:12]}")
        break
    else:
        continue
    break
