"""The one-call engine schedule of the reference's CRFP_DSV_CRA wiring (crfp_cra_forward_batch) against the reference's own output
(tests/golden/dsv_flags.npz, case cra_mid32), against the per-operator composition of the same model, and against itself across the
schedules that must not change a bit (lock-step batch vs one clip per call, one stream vs two)."""
import numpy as np
import pytest
import torch

from test_flags import T, _model, flags  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _clip(seed, n, t, h, w, fv=None):
    from crfp_amd import synth
    return tuple(T(a).cuda() for a in synth.make_clip(seed, n, t, h, w, fv_size=fv or 8 * min(h, w) // 2))


def test_cra_engine_matches_the_reference_golden(flags):
    dev = torch.device("cuda:0")
    m = _model(flags, "cra_mid32", dev)
    assert m.has_engine()
    h, w, fv = int(flags["h"]), int(flags["w"]), int(flags["fv"])
    lrs, fvs, mks = (T(a).to(dev) for a in __import__("crfp_amd.synth", fromlist=["x"]).make_clip(int(flags["cra_mid32.clip_seed"]), 1, int(flags["cra_mid32.t"]), h, w, fv_size=fv))
    ref = T(flags["cra_mid32.out"])
    with torch.no_grad():
        got = m(lrs=lrs, fvs=fvs, mks=mks).cpu()
        comp = m.forward_composed(lrs, fvs, mks).cpu()
        m.precision = "f32"
        strict = m(lrs=lrs, fvs=fvs, mks=mks).cpu()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < 2e-4
    assert float((comp - ref).abs().max()) < 2e-4
    assert float((strict - ref).abs().max()) < 2e-4
    assert not m.engine().overflowed()


@pytest.mark.parametrize("y_only", [False, True])
def test_cra_engine_equals_its_composed_twin_and_is_schedule_invariant(flags, y_only):
    """Random weights of the golden's seed, a 3-clip batch of 4 frames at 24 x 40: engine within 2e-4 of the per-operator composition;
    lock-step == one clip per call and one stream == two streams, bit for bit; first frames differ from later ones (the fusion runs in both)."""
    from crfp_amd.model import CRFP
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    m = CRFP.CRFP_DSV_CRA(dev, mid_channels=32, y_only=y_only).to(dev).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.5)      # default init is small: make the fusion convs matter
    lrs, fvs, mks = _clip(77, 3, 4, 24, 40, 96)
    with torch.no_grad():
        out = m(lrs, fvs, mks)
        comp = m.forward_composed(lrs, fvs, mks)
        eng = m.engine()
        eng.batch_mode = "loop"
        loop = m(lrs, fvs, mks).clone()
        eng.batch_mode = "lockstep"
        eng.single_stream = True
        single = m(lrs, fvs, mks).clone()
        eng.single_stream = False
    assert out.shape == (3, 4, 1 if y_only else 3, 192, 320)
    assert float((out - comp).abs().max()) < 2e-4 * max(1.0, float(comp.abs().max()))
    assert torch.equal(out, loop) and torch.equal(out, single)
    # the wiring is not the plain one: same weights through CRFP_DSV's schedule give another picture
    plain = CRFP.CRFP_DSV(dev, mid_channels=32, y_only=y_only).to(dev).eval()
    plain.load_state_dict({k: v for k, v in m.state_dict().items() if k in plain.state_dict()}, strict=True)
    with torch.no_grad():
        assert float((plain(lrs, fvs, mks) - out).abs().max()) > 1e-3


def test_cra_engine_bf16_storage_and_long_clips(flags):
    """bf16 storage: lock-step == loop bit for bit, and within bf16 noise of the fp32-storage result; a clip longer than the flat limit
    (chunked clip-level stages) equals the same frames run as a flat job."""
    dev = torch.device("cuda:0")
    m = _model(flags, "cra_mid32", dev)
    lrs, fvs, mks = _clip(78, 2, 5, 24, 40, 96)
    with torch.no_grad():
        ref = m(lrs, fvs, mks).clone()
        m.storage = "bf16"
        out = m(lrs, fvs, mks).clone()
        m.engine().batch_mode = "loop"
        loop = m(lrs, fvs, mks).clone()
    assert torch.equal(out, loop)
    assert float((out - ref).abs().max()) < 0.06 and float((out - ref).abs().mean()) < 4e-3
    m.storage = "f32"
    lrs, fvs, mks = _clip(79, 1, 35, 16, 24, 64)      # 35 frames > 32: chunks of 8
    with torch.no_grad():
        long = m(lrs, fvs, mks)
        head = m(lrs[:, :20].contiguous(), fvs[:, :20].contiguous(), mks[:, :20].contiguous())
    assert torch.equal(long[:, :20], head)


def test_cra_engine_refuses_what_it_does_not_have(flags):
    dev = torch.device("cuda:0")
    m = _model(flags, "cra_mid32", dev)
    lrs, fvs, mks = _clip(80, 1, 2, 16, 24, 64)
    with pytest.raises(NotImplementedError):
        m.forward_stream(lrs, fvs, mks)
    with pytest.raises(NotImplementedError):
        m.engine().stream_frame(lrs[0, 0], fvs[0, 0], mks[0, 0])
    # a CRFP_DSV state_dict lacks the wiring's 26 extra parameters
    from crfp_amd import engine
    from crfp_amd.model import CRFP
    with pytest.raises(KeyError):
        engine.CRAEngine(CRFP.CRFP_DSV(dev, mid_channels=32).state_dict(), dev)
