"""CPU: the drop-in boundary as INTEGRATION.md documents it -- the `shim/` directory on PYTHONPATH makes the reference's
entry scripts bind `model`, `dcn_v2` and `pytorch_memlab` to this build even when started from the reference's own
checkout; `crfp_amd.option` / `crfp_amd.main` mirror option.py / main.py; the independent DCNv2 restatement agrees with
the oracle.  Tests that need /root/reference skip when it is absent (it does not exist on the GPU box)."""
import os
import shlex
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from conftest import ROOT

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout only exists in the build container")
SHIM = os.path.join(ROOT, "shim")


def _run_in(cwd, code, extra_env=None):
    env = dict(os.environ, PYTHONPATH=SHIM)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=cwd, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


@needs_ref
def test_shim_wins_from_the_reference_checkout(tmp_path):
    """`cd <reference> && PYTHONPATH=<repo>/shim python3 main.py` situation: sys.path[0] is the reference, which holds its
    own `model/`; the shim's finder must still win for model / dcn_v2 / pytorch_memlab and leave utils / dataset alone.
    Then the exact constructor call of main.py:34 (on the CPU device: modules only hold parameters) and the
    nn.DataParallel wrap of main.py:37-38."""
    fnet = tmp_path / "fnet.pth"
    out = _run_in(REF, f"""
        import os, sys, importlib.util
        assert os.getcwd() == {REF!r} and sys.path[0] == ''
        import model, dcn_v2, pytorch_memlab
        from model import CRFP, MRCF_runtime, MRCF_test
        from dcn_v2 import DCNv2
        from pytorch_memlab import LineProfiler, MemReporter
        shim = {SHIM!r}
        assert model.__file__.startswith(shim) and dcn_v2.__file__.startswith(shim) and pytorch_memlab.__file__.startswith(shim)
        assert CRFP.__file__.endswith('crfp_amd/model/CRFP.py'), CRFP.__file__
        import model.CRFP as again
        assert again is CRFP
        # not shadowed: the reference's own modules
        for name in ('utils', 'option', 'trainer'):
            spec = importlib.util.find_spec(name)
            assert spec is not None and spec.origin.startswith({REF!r}), (name, spec)
        import torch, torch.nn as nn
        torch.save(CRFP.FNet(in_nc=3).state_dict(), {str(fnet)!r})
        device = torch.device('cpu')
        _model = CRFP.CRFP_DSV(mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True,
                               spynet_pretrained={str(fnet)!r}, device=device).to(device)
        assert len(_model.state_dict()) == 118
        assert any('spynet' in n for n, _ in _model.named_parameters())      # trainer.py:131-141
        dp = nn.DataParallel(_model, list(range(2)))
        assert dp.module is _model and len(dp.state_dict()) == 118
        m = MRCF_runtime.MRCF_simple_v18(mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True, split_ratio=3,
                                         spynet_pretrained={str(fnet)!r}, device=device)
        import inspect
        assert 'warp_size' in inspect.signature(m.forward).parameters and len(m.state_dict()) == 158   # test_runtime.py:142
        assert hasattr(MRCF_test.MRCF_simple_v18, 'clear_states')                                         # test_video.py
        d = DCNv2(32, 32, 3, stride=1, padding=1, dilation=1, deformable_groups=8)
        assert tuple(d.weight.shape) == (32, 32, 3, 3)
        with LineProfiler(m.forward) as prof:
            pass
        MemReporter(m).report()
        print('OK')
    """)
    assert out.strip().endswith("OK")


def test_shim_is_inert_without_pythonpath():
    r = subprocess.run([sys.executable, "-c", "import sys; import importlib.util as u; print(u.find_spec('dcn_v2'))"],
                       cwd="/tmp", env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"}, capture_output=True, text=True)
    assert r.stdout.strip() == "None"


def test_cpu_forward_is_refused_not_emulated():
    """No CPU fallback in the product: a CPU tensor reaches the C-ABI wrapper and raises."""
    from crfp_amd.model import CRFP
    m = CRFP.CRFP_DSV(device=torch.device("cpu"), mid_channels=32)
    with pytest.raises(RuntimeError, match="CUDA/HIP"):
        m(lrs=torch.zeros(1, 2, 3, 8, 8), fvs=torch.zeros(1, 2, 3, 64, 64), mks=torch.zeros(1, 2, 1, 64, 64, dtype=torch.bool))


@needs_ref
def test_option_flags_match_the_reference():
    """Every flag of option.py:10-119 exists here with the same default, and eval.sh's argument list parses."""
    code = "import sys, json; sys.argv=['x']; import option; print(json.dumps(vars(option.args)))"
    r = subprocess.run([sys.executable, "-c", code], cwd=REF, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import json
    ref = json.loads(r.stdout)
    from crfp_amd import option
    mine = vars(option.parse([]))
    assert mine == ref
    eval_sh = open(os.path.join(REF, "eval.sh")).read()
    argv = shlex.split(eval_sh.split("main.py", 1)[1].replace("\\\n", " "))
    a = option.parse(argv)
    assert a.eval and a.reset and a.FV_size == 96 and a.scale == 8 and a.N_frames == 15 and a.dataset == "Reds"
    assert set(option.USED) | set(option.IGNORED) == set(ref)


def test_main_dispatch_guards(tmp_path):
    from crfp_amd import main, option
    with pytest.raises(SystemExit, match="eval path"):
        main.main(["--save_dir", str(tmp_path / "a")])
    with pytest.raises(SystemExit, match="torchrun"):
        main.main(["--eval", "True", "--num_gpu", "4", "--save_dir", str(tmp_path / "b")])
    args = option.parse(["--cpu", "True"])
    with pytest.raises(SystemExit, match="no CPU path"):
        main.select_device(args)
    # mkExpDir semantics (utils.py:41-64)
    args = option.parse(["--eval", "True", "--eval_save_results", "True", "--save_dir", str(tmp_path / "exp")])
    main.mk_exp_dir(args)
    assert os.path.isdir(tmp_path / "exp" / "save_results") and os.path.exists(tmp_path / "exp" / "args.txt")
    with pytest.raises(SystemExit, match="already exists"):
        main.mk_exp_dir(args)
    args.reset = True
    main.mk_exp_dir(args)


@pytest.mark.parametrize("C,O,dg,H,W", [(32, 32, 8, 9, 11), (4, 4, 1, 7, 13), (8, 12, 2, 6, 5)])
def test_oracle_dcnv2_vs_paper_equations(C, O, dg, H, W):
    """oracle.dcnv2 (and the C twin's convention) against the float64 restatement written from the DCN papers' equations,
    at sampling positions straddling -1, 0, H-1 and H."""
    from dcn_paper_ref import boundary_offsets, dcnv2_paper
    from oracle import crfp_oracle as orc
    rs = np.random.RandomState(C + H)
    x = rs.standard_normal((2, C, H, W)).astype(np.float32)
    off = boundary_offsets(rs, 2, dg, H, W)
    m = rs.uniform(0, 1, (2, dg * 9, H, W)).astype(np.float32)
    w = (rs.standard_normal((O, C, 3, 3)) * 0.2).astype(np.float32)
    b = rs.standard_normal(O).astype(np.float32)
    ref = dcnv2_paper(x, off, m, w, b, dg)
    got = orc.dcnv2(*[torch.from_numpy(a) for a in (x, off, m, w, b)], dg).numpy()
    assert np.abs(ref - got).max() < 2e-5
