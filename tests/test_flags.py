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


def _model(g, name, device):
    from crfp_amd import synth
    from crfp_amd.model import CRFP
    m = CRFP.CRFP_DSV(device=device, **_kwargs(g, name))
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
                CRFP.CRFP_DSV(device=torch.device("cpu"), **kw)
            continue
        m = CRFP.CRFP_DSV(device=torch.device("cpu"), **kw)
        mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        ref = _table(flags, name)
        assert list(mine) == list(ref), name
        assert mine == ref, name
        assert m.has_engine() == (kw.get("mid_channels", 16) == 32 and kw.get("hr_dcn", True) and kw.get("offset_prop", True))


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
        assert not m.has_engine()
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
    assert ran >= 4 and failed >= 3
