"""CRFP_DSV's constructor flags (mid_channels, hr_dcn, offset_prop, y_only) against tests/golden/dsv_flags.npz, which holds what the
imported reference does for each combination (tests/golden/make_flags_golden.py): key / shape tables, outputs where its forward
runs, exception classes where the reference itself fails."""
import ast
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

T = torch.from_numpy


@pytest.fixture(scope="module")
def flags():
    return dict(np.load(os.path.join(GOLDEN, "dsv_flags.npz")))


def _kwargs(g, name):
    return dict(ast.literal_eval(str(g[f"{name}.kwargs"])))


def _table(g, name):
    tab = {}
    for item in g[f"{name}.keys"]:
        key, shp = str(item).split(":")
        tab[key] = tuple(int(v) for v in shp.split(",")) if shp else ()
    return tab


def _cls(name):
    from crfp_amd.model import CRFP
    return {"cra": CRFP.CRFP_DSV_CRA, "simple": CRFP.CRFP_simple, "dense": CRFP.CRFP}.get(name.split("_")[0], CRFP.CRFP_DSV)


def _model(g, name, device):
    from crfp_amd import synth
    m = _cls(name)(device=device, **_kwargs(g, name))
    sd = synth.make_state_dict_like(_table(g, name), int(g["weights_seed"]))
    assert synth.state_dict_digest(sd) == str(g[f"{name}.weights_sha256"])
    m.load_state_dict({k: T(v.copy()) for k, v in sd.items()}, strict=True)
    return m.to(device).eval()


def test_state_dict_tables_match_the_reference_for_every_flag_combination(flags):
    """No GPU: constructing the mirror only builds parameters.  Same keys, same order, same shapes as the reference's module table."""
    from crfp_amd.model import CRFP
    for name in map(str, flags["cases"]):
        kw = _kwargs(flags, name)
        if f"{name}.ctor_error" in flags:
            with pytest.raises(AssertionError if str(flags[f"{name}.ctor_error"]) == "AssertionError" else Exception):
                _cls(name)(device=torch.device("cpu"), **kw)
            continue
        m = _cls(name)(device=torch.device("cpu"), **kw)
        mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        ref = _table(flags, name)
        assert list(mine) == list(ref), name
        assert mine == ref, name
        if hasattr(m, "has_engine"):
            mids = (16, 32)   # mid_channels = 16 runs embedded in the 32-channel schedule
            assert m.has_engine() == (kw.get("mid_channels", 16) in mids and kw.get("hr_dcn", True) and kw.get("offset_prop", True))


@pytest.mark.gpu
def test_flag_combinations_behave_like_the_reference(flags):
    """Outputs within 2e-4 of the reference's where it runs; where the reference raises, the same exception class at the same place
    (hr_dcn=False: channel-count RuntimeError on the first frame; offset_prop=False: AttributeError from the second frame on)."""
    from crfp_amd import synth
    dev = torch.device("cuda:0")
    h, w, fv = int(flags["h"]), int(flags["w"]), int(flags["fv"])
    ran = failed = 0
    for name in map(str, flags["cases"]):
        if f"{name}.ctor_error" in flags:
            continue
        m = _model(flags, name, dev)
        # the cases with a one-call engine schedule run it here (their composed twins: test_gpu_cra.py, test_gpu_ablation_engines.py)
        assert getattr(m, "has_engine", lambda: False)() == (name in ("cra_mid32", "cra_mid16_yonly", "simple_mid32", "dense_mid32", "mid16_default", "mid16_yonly"))
        lrs, fvs, mks = (T(a).to(dev) for a in synth.make_clip(int(flags[f"{name}.clip_seed"]), 1, int(flags[f"{name}.t"]), h, w, fv_size=fv))
        if f"{name}.forward_error" in flags:
            cls = {"RuntimeError": RuntimeError, "AttributeError": AttributeError}[str(flags[f"{name}.forward_error"])]
            with pytest.raises(cls) as ei:
                m(lrs=lrs, fvs=fvs, mks=mks)
            if cls is AttributeError:
                assert "conv_fuse" in str(ei.value)
            else:
                assert "channels" in str(ei.value)
            failed += 1
            continue
        with torch.no_grad():
            got = m(lrs=lrs, fvs=fvs, mks=mks).cpu()
        ref = T(flags[f"{name}.out"])
        assert got.shape == ref.shape, name
        d = float((got - ref).abs().max())
        assert d < 2e-4, (name, d)
        ran += 1
    assert ran >= 11 and failed >= 3


# ---- the regional runtime wiring (model/CRFP_runtime.py): calls the one-call engine does not take
@pytest.fixture(scope="module")
def rt_flags():
    return dict(np.load(os.path.join(GOLDEN, "runtime_flags.npz")))


def _rt_model(g, device, offset_prop, seed):
    from crfp_amd import synth
    from crfp_amd.model import MRCF_runtime
    m = MRCF_runtime.MRCF_simple_v18(mid_channels=32, y_only=False, hr_dcn=True, offset_prop=offset_prop, split_ratio=3,
                                     spynet_pretrained='pretrained_models/fnet.pth', device=device)
    sd = synth.make_state_dict_like({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed)
    m.load_state_dict({k: T(v.copy()) for k, v in sd.items()}, strict=True)
    return m.to(device).eval()


def test_runtime_mirror_without_offset_prop_has_the_reference_table(rt_flags):
    m = _rt_model(rt_flags, torch.device("cpu"), False, 1)
    mine = [f"{k}:{','.join(map(str, v.shape))}" for k, v in m.state_dict().items()]
    assert mine == [str(s) for s in rt_flags["noprop.keys"]]
    assert not any("conv_fuse" in k or k.startswith("dcn_3.upsample") for k in m.state_dict())


@pytest.mark.gpu
def test_runtime_mirror_routes_what_the_engine_does_not_take(rt_flags):
    """offset_prop=False and the oversized default warp_size used to raise from the one-call path (KeyError at pack time / ValueError):
    both now run through the per-operator composition and match the reference class's own output."""
    dev = torch.device("cuda:0")
    g = rt_flags
    m = _rt_model(g, dev, False, int(g["noprop.weights_seed"]))
    lrs, fvs = T(g["noprop.lrs"]).to(dev), T(g["noprop.fvs"]).to(dev)
    assert not m._engine_takes(lrs, fvs, tuple(int(v) for v in g["noprop.warp"]))
    with torch.no_grad():
        got = m(lrs, fvs, warp_size=tuple(int(v) for v in g["noprop.warp"])).cpu()
    assert float((got - T(g["noprop.out"])).abs().max()) < 2e-4
    m2 = _rt_model(g, dev, True, 17)                         # make_runtime_golden.py's SEED
    lrs, fvs = T(g["oversize.lrs"]).to(dev), T(g["oversize.fvs"]).to(dev)
    assert not m2._engine_takes(lrs, fvs, (1080, 1920)) and m2._engine_takes(lrs, fvs, (128, 192))
    with torch.no_grad():
        got = m2(lrs, fvs).cpu()                             # the signature's default warp_size on a 128 x 192 frame
        clamped = m2(lrs, fvs, warp_size=(128, 192)).cpu()   # the one-call engine on the window the reference's slicing ends up with
    assert float((got - T(g["oversize.out"])).abs().max()) < 2e-4
    assert float((clamped - T(g["oversize.out"])).abs().max()) < 2e-4


@pytest.mark.parametrize("cls,wiring", [("CRFP_DSV", "dsv"), ("CRFP_DSV_CRA", "cra"), ("CRFP_simple", "simple"), ("CRFP", "dense")])
@pytest.mark.parametrize("y_only", [False, True])
def test_embedded_narrow_tables_have_the_32_channel_shapes_and_keep_every_weight(cls, wiring, y_only, mid=16):
    """No GPU: crfp_amd.engine.embed_mid32 places a narrow model's parameters inside the 32-channel table the engine packs -- same keys,
    the 32-channel shapes, every value kept exactly once, everything else zero."""
    from crfp_amd import engine
    from crfp_amd.model import CRFP
    torch.manual_seed(mid)
    narrow = getattr(CRFP, cls)(torch.device("cpu"), mid_channels=mid, y_only=y_only).state_dict()
    wide = getattr(CRFP, cls)(torch.device("cpu"), mid_channels=32, y_only=y_only).state_dict()
    emb = engine.embed_mid32(narrow, mid, wiring, y_only=y_only)
    assert list(emb) == list(wide)
    for k, v in emb.items():
        assert tuple(v.shape) == tuple(wide[k].shape), k
        a, b = narrow[k].flatten(), v.flatten()
        assert int((b != 0).sum()) == int((a != 0).sum()), k
        assert torch.equal(torch.sort(b[b != 0])[0], torch.sort(a[a != 0])[0]), k
