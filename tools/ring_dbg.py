"""Bit-for-bit check of the LDS-DMA ring conv build (make -C crfp_amd/csrc EXTRA=-DCRFP_BF16_RING; CRFP_BF16_RING=1) against the shipped
kernels: runs a 2-frame bf16 clip in child processes with and without the switch and compares workspace intermediates."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from crfp_amd import synth
    from crfp_amd.engine import DSVEngine
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(7).items()}
    h, w, t = int(sys.argv[3]), int(sys.argv[4]), 2
    eng = DSVEngine(sd, dev, storage="bf16")
    eng.single_stream = True
    lrs, fvs, mks = (torch.from_numpy(a).to(dev) for a in synth.make_clip(5, 1, t, h, w, fv_size=48))
    out = eng.forward(lrs, fvs, mks)
    res = {"out": out.cpu().numpy()}
    for name in ("enc_lr0", "x_lr", "fnet.a0", "fnet.a1", "res.y0", "dcn.fa", "dcn.fb", "offfeat0", "offfeat1", "offfeat2", "res.y0", "res.y1", "prop_a", "prop_b", "aligned"):
        res[name] = eng.debug_fetch(name, t, h, w).cpu().numpy()
    np.savez(sys.argv[2], **res)
else:
    import numpy as np
    for (h, w) in ((64, 64),):
        for tag, env in (("base", {}), ("ring", {"CRFP_BF16_RING": "1"}), ("ring2", {"CRFP_BF16_RING": "1"})):
            subprocess.run([sys.executable, __file__, "child", f"/tmp/{tag}.npz", str(h), str(w)], env=dict(os.environ, **env), check=True)
        a, b, c = np.load("/tmp/base.npz"), np.load("/tmp/ring.npz"), np.load("/tmp/ring2.npz")
        print("size", h, w)
        for k in a.files:
            d = np.abs(a[k] - b[k]); d2 = np.abs(b[k] - c[k])
            bad = np.argwhere(d > 0)
            if k == "x_lr":
                dd = d[0]   # [c, h, w]
                print("   rows bad (c0):", (dd[0] > 0).sum(1).tolist())
                print("   cols bad (c0):", (dd[0] > 0).sum(0).tolist())
                print("   chans bad:", (dd > 0).reshape(dd.shape[0], -1).sum(1).tolist())
                print("   base c0 r2 :", np.round(a[k][0, 0, 2, :8], 4).tolist()); print("   ring c0 r2 :", np.round(b[k][0, 0, 2, :8], 4).tolist())
                print("   base c0 r3 :", np.round(a[k][0, 0, 3, :8], 4).tolist()); print("   ring c0 r1 :", np.round(b[k][0, 0, 1, :8], 4).tolist())
            print(f"  {k:10s} max|base-ring| {d.max():.3e}  n_bad {len(bad)}  ring-vs-ring {d2.max():.3e}", (bad[:3].tolist() if len(bad) else ""), (bad[-2:].tolist() if len(bad) else ""))
