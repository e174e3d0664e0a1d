"""CPU: host logic and the C-ABI surface (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

from crfp_amd import synth


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ROOT, "crfp_amd", "libcrfp_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "crfp_amd", "csrc"), "-j8"])
    from crfp_amd import _lib
    return _lib.lib()


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "crfp_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(crfp_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(lib):
    names = declared_symbols()
    assert len(names) >= 20
    raw = ctypes.CDLL(os.path.join(ROOT, "crfp_amd", "libcrfp_hip.so"))
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/crfp_hip.h but not exported"


def test_binding_table_covers_header():
    from crfp_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_version_and_param_table(lib):
    from crfp_amd import engine
    assert lib.crfp_version() == 201
    names = engine.param_names()
    assert names == synth.state_dict_keys()
    sd = synth.make_state_dict(1)
    for i, k in enumerate(names):
        assert lib.crfp_dsv_param_numel(i, 0) == sd[k].size
    sd1 = synth.make_state_dict(1, y_only=True)
    for i, k in enumerate(names):
        assert lib.crfp_dsv_param_numel(i, 1) == sd1[k].size
    assert lib.crfp_dsv_param_name(-1) is None and lib.crfp_dsv_param_name(118) is None


def test_sizes_are_sane(lib):
    assert 9e6 < lib.crfp_dsv_packed_weight_bytes(0) < 40e6      # fp32 pack + split-bf16 pack (1.5x) + padding
    a = lib.crfp_dsv_workspace_bytes(7, 180, 320)
    b = lib.crfp_dsv_workspace_bytes(7, 270, 480)
    assert 1e9 < a < 4e9 and 2.0 < b / a < 2.5                    # scales with pixels
    assert lib.crfp_dsv_workspace_bytes(0, 180, 320) == 0
    assert lib.crfp_dsv_workspace_bytes(7, 4, 4) == 0
    assert lib.crfp_flow_warp_workspace_bytes(1, 32, 360, 640) == 2 * 32 * 360 * 640 * 4


def test_argument_errors_do_not_touch_the_gpu(lib):
    from crfp_amd import _lib
    rc = lib.crfp_dsv_forward_clip(None, 0, None, None, None, None, 7, 180, 320, None, 0, None)
    assert rc == -1 and b"null" in lib.crfp_last_error_string()
    rc = lib.crfp_flow_warp_f32(None, None, None, 1, 4, 8, 8, 0, None, 0, None)
    assert rc == -1
    rc = lib.crfp_dcnv2_forward_f32(*([ctypes.c_void_p(16)] * 6), 1, 32, 32, 8, 8, 5, 2, 1, 8, None, 0, None)
    assert rc == -3 and b"kernel 3" in lib.crfp_last_error_string()
    rc = lib.crfp_dcnv2_forward_f32(*([ctypes.c_void_p(16)] * 6), 1, 30, 32, 8, 8, 3, 1, 1, 8, None, 0, None)
    assert rc == -1
    rc = lib.crfp_conv3x3_f32(*([ctypes.c_void_p(16)] * 4), 1, 3, 8, 8, 8, 9, 1.0, None, 0, None)
    assert rc == -1
    with pytest.raises(RuntimeError):
        _lib.check(rc, "x")


def test_module_mirror_keys_and_shapes():
    from crfp_amd.model import CRFP
    for y_only in (False, True):
        m = CRFP.CRFP_DSV(device=torch.device("cpu"), mid_channels=32, y_only=y_only)
        sd = synth.make_state_dict(3, y_only=y_only)
        assert list(m.state_dict().keys()) == list(sd.keys())
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
        assert any("spynet" in k for k, _ in m.named_parameters())        # trainer.py:131-141 relies on it
    # round 4: every flag combination constructs (tests/test_flags.py pins the tables); round 6: the shipped width and the constructor default
    # (16, embedded in the 32-channel schedule) have the one-call engine, wider models and the flags-off combinations do not
    assert CRFP.CRFP_DSV(device=torch.device("cpu"), mid_channels=16).has_engine()
    assert CRFP.CRFP_DSV(device=torch.device("cpu"), mid_channels=32).has_engine()
    assert not CRFP.CRFP_DSV(device=torch.device("cpu"), mid_channels=64).has_engine()
    assert not CRFP.CRFP_DSV(device=torch.device("cpu"), mid_channels=16, offset_prop=False).has_engine()
    with pytest.raises(ValueError):
        CRFP.CRFP_DSV(device=torch.device("cpu"), mid_channels=20)


def test_dcn_module_init_matches_reference_contract():
    from crfp_amd.model import CRFP
    m = CRFP.DCN_module(32, 8, 3, 10)
    assert float(m.dcn_offset.weight.abs().max()) == 0 and float(m.dcn_mask.bias.abs().max()) == 0
    w = m.dcn.weight.detach()
    assert float(w.sum()) == 32 and all(float(w[i, i, 1, 1]) == 1 for i in range(32))
    r = CRFP.DCN_module(4, 1, 3, 10, repeat=True, pre_offset=True, interpolate="pixelshuffle")
    assert r.dcn_offset.out_channels == 2 and r.dcn_mask.out_channels == 1
    assert r.upsample.upsample_conv.weight.shape == (64, 32, 3, 3)


def test_product_refuses_cpu_tensors():
    from crfp_amd import ops
    from crfp_amd.model import CRFP
    with pytest.raises(RuntimeError):
        ops.conv3x3(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3))
    m = CRFP.CRFP_DSV(device=torch.device("cpu"), mid_channels=32)
    with pytest.raises(RuntimeError):
        m(lrs=torch.zeros(1, 2, 3, 8, 8), fvs=torch.zeros(1, 2, 3, 64, 64), mks=torch.zeros(1, 2, 1, 64, 64, dtype=torch.bool))


def test_product_never_imports_oracle():
    """The oracle is a checker: nothing under crfp_amd/ may import, call or load it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "crfp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "dcnv2_ref" not in txt, f


def test_missing_library_fails_loudly(tmp_path):
    code = ("import sys; sys.path.insert(0, %r); import crfp_amd._lib as L; L.LIB_PATH = %r\n"
            "try:\n    L.lib()\nexcept RuntimeError as e:\n    print('RAISED', 'not built' in str(e))\n") % (ROOT, str(tmp_path / "nope.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout
    assert "RAISED True" in out


def test_eval_quirks():
    from crfp_amd import evalrig
    assert list(evalrig.counted_frames(0, 4)) == [1, 2, 3]         # trainer.py:350-351
    assert list(evalrig.counted_frames(7, 4)) == [0, 1, 2, 3]
    assert list(evalrig.counted_frames(50, 3)) == [1, 2]
    assert evalrig.range_divisor(torch.tensor([16.0, 235.0])) == 255.0
    assert evalrig.range_divisor(torch.tensor([-1.0, 0.9])) == 2.0
    assert evalrig.range_divisor(torch.tensor([0.0, 1.0])) == 1.0
    assert abs(evalrig.psnr_from_mse(0.01, 10) - 20.0) < 1e-9
    assert evalrig.shard_clips(32, 3, 8) == [3, 11, 19, 27]
    assert sorted(sum((evalrig.shard_clips(10, r, 4) for r in range(4)), [])) == list(range(10))


def test_synth_clip_properties():
    lrs, fvs, mks = synth.make_clip(5, 1, 3, 16, 24, fv_size=48, sigma_t=10.0)
    assert lrs.shape == (1, 3, 3, 16, 24) and fvs.shape == (1, 3, 3, 128, 192) and mks.dtype == np.bool_
    assert int(mks[0, 0].sum()) == 48 * 48
    assert float(np.abs(fvs[~np.broadcast_to(mks, fvs.shape)]).max()) == 0.0   # zero outside the fovea (reds.py:196-203)
    assert 0.0 <= lrs.min() and lrs.max() <= 1.0


def test_device_code_has_no_packed_fp32_ops():
    """Build property (crfp_amd/csrc/Makefile): v_pk_{fma,mul,add}_f32 returns wrong lanes on
    MI355X while another dispatch issues bf16 MFMAs on the same SIMD, so the library is built without them; the
    bf16 MFMA the split convolution relies on must still be there (guards against a no-op disassembly)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    if not os.path.exists(os.path.join(check_isa.LLVM, "llvm-objdump")):
        pytest.skip("ROCm LLVM tools not installed")
    from crfp_amd import _lib
    n_pk, total = check_isa.count(_lib.LIB_PATH, r"\bv_pk_(fma|mul|add)_f32\b")
    n_mfma, _ = check_isa.count(_lib.LIB_PATH, r"v_mfma_f32_32x32x16_(f16|bf16)")   # the aggressor class of the hazard
    assert total > 10000 and n_mfma > 0
    assert n_pk == 0, f"{n_pk} packed-FP32 VALU instructions in {_lib.LIB_PATH}"


def test_gaze_rig_masks_and_trajectory():
    """crfp_amd.gaze vs a literal restatement of test_video.py:303-379 (iterated 3x3 dilation, mask algebra, history)."""
    from crfp_amd import gaze
    import torch.nn.functional as F
    H, W, fv, N, sigma = 72, 96, 16, 6, 9.0
    xs, ys = gaze.gaze_trajectory(N, H, W, sigma, np.random.RandomState(3))
    rs = np.random.RandomState(3)
    assert np.array_equal(xs, sigma * rs.randn(N) + W / 2) and np.array_equal(ys, sigma * rs.randn(N) + H / 2)
    assert gaze.window_origin(W / 2 + 0.7, H / 2 - 0.2, fv, H, W) == (int(H / 2 - 0.2) - fv // 2, int(W / 2 + 0.7) - fv // 2)
    assert gaze.window_origin(-5.0, 1e6, fv, H, W) == (H - fv, 0)                 # clipped instead of wrapping
    assert gaze.regional_box(10, 20, 16, 40, 40, H, W) == (0, 38, 8, 48)
    kern = torch.ones(1, 1, 3, 3)
    masks = gaze.RegionMasks(H, W, fv, torch.device("cpu"), fv_start=1)
    hist = []
    past_ref = None
    for n in range(N):
        cy, cx = gaze.window_origin(xs[n], ys[n], fv, H, W)
        m = masks.frame(n, cy, cx)
        mk = torch.zeros(1, 1, H, W)
        if n >= 1:
            mk[:, :, cy:cy + fv, cx:cx + fv] = 1
        mk_fv = mk.clone(); mk_fv[:, :, cy:cy + fv, cx:cx + fv] = 1
        mk_out = mk_fv.clone()
        for _ in range(10):
            mk_out = torch.clamp(F.conv2d(mk_out, kern, padding=(1, 1)), 0, 1)
        mk_out = torch.logical_and(torch.logical_not(mk), mk_out)
        assert torch.equal(m["mk"], mk.bool()) and torch.equal(m["fovea"], mk_fv.bool()) and torch.equal(m["outskirt"], mk_out)
        assert (m["past"] is None and past_ref is None) or torch.equal(m["past"], past_ref)
        hist.append(mk_out)
        if len(hist) > 3:
            hist.pop(0)
        past_ref = torch.sum(torch.cat(hist, dim=1), dim=1, keepdim=True).clip(0, 1).bool()
        assert bool(m["fg"].all())
    assert int(m["outskirt"].sum()) > 0 and not bool((m["outskirt"] & m["mk"]).any())


@pytest.mark.parametrize("tag", ["a", "b"])
def test_gaze_rig_masks_match_the_reference_loop(tag):
    """crfp_amd.gaze against masks produced by EXECUTING the reference's per-frame loop (test_video.py:303-375 through
    tests/golden/make_gaze_golden.py): trajectory, window origin, mk / fovea / outskirt ring / three-frame past union / regional box,
    with and without regional DCN and a late fovea start -- bit for bit."""
    from crfp_amd import gaze
    g = np.load(os.path.join(ROOT, "tests", "golden", "gaze_masks.npz"))
    kw = {k: g[f"{tag}_{k}"].item() for k in ("seed", "N", "H", "W", "fv_size", "sigma", "regional_dcn", "rg", "fv_start")}
    N, H, W = kw["N"], kw["H"], kw["W"]
    xs, ys = gaze.gaze_trajectory(N, H, W, kw["sigma"], np.random.RandomState(kw["seed"]))
    masks = gaze.RegionMasks(H, W, kw["fv_size"], torch.device("cpu"), fv_start=kw["fv_start"], regional_dcn=bool(kw["regional_dcn"]),
                             rg_h=kw["rg"], rg_w=kw["rg"])
    unpack = lambda name, n: np.unpackbits(g[f"{tag}_{name}"][n], axis=-1)[:, :W].astype(bool)
    for n in range(N):
        cy, cx = gaze.window_origin(xs[n], ys[n], kw["fv_size"], H, W)
        assert (cy, cx) == tuple(g[f"{tag}_cur"][n]), n
        f = masks.frame(n, cy, cx)
        for name in ("mk", "fovea", "outskirt", "fg"):
            assert np.array_equal(f[name].reshape(H, W).numpy(), unpack(name, n)), (name, n)
        past = np.zeros((H, W), bool) if f["past"] is None else f["past"].reshape(H, W).numpy()
        assert np.array_equal(past, unpack("past", n)), ("past", n)


def test_product_library_carries_no_lab_kernels():
    """The experiments that lose to the default conv main loop (pipelined / input-stationary / warp-specialised / bf16x6)
    and the s_memtime stamp code are compiled only into the lab library (`make lab`, -DCRFP_LAB)."""
    so = os.path.join(ROOT, "crfp_amd", "libcrfp_hip.so")
    blob = open(so, "rb").read()
    for name in (b"conv3x3_split_pipe_kernel", b"conv3x3_split_is_kernel", b"conv3x3_split_ws_kernel"):
        assert name not in blob, name
    assert b"conv3x3_split_kernel" in blob
    assert b"dcn_fused2_kernel" not in blob          # round 5: the role-specialised fused DCN (lost its A/B) is lab-only too
    assert os.path.getsize(so) < 4 << 20


def test_product_library_reads_only_the_documented_switches():
    """VERDICT r4 item 7 / SURVEY 8b ("no mutable global state after first-call init"): the product .so changes kernels only under
    the five documented process-wide switches of INTEGRATION.md section 4.  Every tuning / probe / stamp knob is compiled only with
    -DCRFP_LAB, so no other CRFP_* name may appear in the product's string table -- the lab library must still carry them."""
    import re
    documented = {b"CRFP_PRECISION", b"CRFP_SIDE_STREAM", b"CRFP_MASK_GATE", b"CRFP_DCN_FUSED", b"CRFP_CONV_PAIR"}
    names = lambda path: set(re.findall(rb"(?<![A-Za-z0-9_])CRFP_[A-Z0-9_]+(?![A-Za-z0-9_])", open(path, "rb").read()))
    prod = names(os.path.join(ROOT, "crfp_amd", "libcrfp_hip.so"))
    # CRFP_E_* are the error-code names inside messages, CRFP_DSV_* the flag names inside messages: not environment variables
    env_like = {n for n in prod if not n.startswith((b"CRFP_E_", b"CRFP_DSV_", b"CRFP_NS", b"CRFP_ACT_"))}
    assert env_like == documented, sorted(env_like ^ documented)
    lab = names(os.path.join(ROOT, "crfp_amd", "libcrfp_hip_lab.so"))
    for knob in (b"CRFP_CONV_MODE", b"CRFP_DCN_MODE", b"CRFP_DCN_FUSE_V", b"CRFP_DCN_FUSE_NW", b"CRFP_F32_S8_MAX_WGS", b"CRFP_DCN_VARIANT"):
        assert knob in lab and knob not in prod, knob


def test_regional_engine_tables_and_argument_errors(lib):
    """crfp_rt_* (the one-call schedule of model/CRFP_runtime.py::MRCF_simple_v18): the parameter table is the mirror's (= the
    reference class's, tests/golden/runtime_small.npz `keys`) state_dict in order; geometry the schedule cannot run is refused by
    the size query and by the forward call before anything touches the GPU."""
    from crfp_amd import _lib
    from crfp_amd.engine import RuntimeEngine
    from crfp_amd.model import MRCF_runtime
    from conftest import GOLDEN
    m = MRCF_runtime.MRCF_simple_v18(mid_channels=32, y_only=False, hr_dcn=True, offset_prop=True, split_ratio=3, device=torch.device("cpu"))
    names = RuntimeEngine.param_names()
    assert len(names) == _lib.RT_NUM_PARAMS == 158 and names == list(m.state_dict().keys())
    assert names == [str(k) for k in np.load(os.path.join(GOLDEN, "runtime_small.npz"))["keys"]]
    for i, (k, v) in enumerate(m.state_dict().items()):
        assert lib.crfp_rt_param_numel(i, 0) == v.numel(), k
    assert lib.crfp_rt_param_numel(names.index("conv_last.weight"), 1) == 1 * 4 * 9
    assert lib.crfp_rt_param_name(-1) is None and lib.crfp_rt_param_name(158) is None
    assert 9e6 < lib.crfp_rt_packed_weight_bytes(0) < 60e6
    rig = lib.crfp_rt_workspace_bytes(5, 135, 240, 96, 96, 720, 720)          # test_runtime.py's geometry
    assert 3e8 < rig < 3e9
    for bad in ((5, 135, 240, 96, 96, 724, 720),       # window not a multiple of 8
                (5, 135, 240, 96, 96, 720, 1928),      # wider than the 1920-pixel frame
                (5, 135, 240, 96, 96, 56, 720),        # FNet needs an 8 x 8 LR window
                (5, 135, 240, 2000, 96, 720, 720),     # fovea crop taller than the frame
                (0, 135, 240, 96, 96, 720, 720), (5, 4, 240, 96, 96, 720, 720)):
        assert lib.crfp_rt_workspace_bytes(*bad) == 0, bad
    assert b"warp_size" in lib.crfp_last_error_string() or b"bad clip" in lib.crfp_last_error_string()
    p16 = ctypes.c_void_p(16)
    assert lib.crfp_rt_forward_clip(p16, 0, p16, p16, p16, 5, 135, 240, 96, 96, 724, 720, p16, 1 << 40, None) == -1
    assert lib.crfp_rt_forward_clip(p16, 0, None, p16, p16, 5, 135, 240, 96, 96, 720, 720, p16, 1 << 40, None) == -1
    assert lib.crfp_rt_forward_clip(p16, 0, p16, p16, p16, 5, 135, 240, 96, 96, 720, 720, p16, 1024, None) == -2     # workspace too small
    assert lib.crfp_rt_forward_clip(p16, 2, p16, p16, p16, 5, 135, 240, 96, 96, 720, 720, p16, 1 << 40, None) == -3   # CRFP_DSV_STRICT_F32
    assert lib.crfp_rt_pack_weights(None, 0, p16, 1 << 30, None) == -1


def _build_c_host(tmp_path):
    exe = str(tmp_path / "c_host_smoke")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           os.path.join(ROOT, "examples", "c_host_smoke.c"), "-L", os.path.join(ROOT, "crfp_amd"), "-lcrfp_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.join(ROOT, "crfp_amd"), "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_c99_and_a_plain_c_host_links(tmp_path):
    """The drop-in boundary is a C ABI: include/crfp_hip.h compiles as C99 (-Wpedantic clean) and examples/c_host_smoke.c -- a host
    without Python or torch -- builds with gcc against libcrfp_hip.so (it runs in the -m gpu suite)."""
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", "-Wall", "-Wpedantic", "-Werror", os.path.join(ROOT, "include", "crfp_hip.h")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(_build_c_host(tmp_path))


@pytest.mark.gpu
def test_plain_c_host_runs_on_the_gpu(tmp_path):
    """examples/c_host_smoke.c: hipMalloc'ed buffers, crfp_flow_warp_f32 / crfp_upsample_bilinear_f32 / an argument error, from C."""
    r = subprocess.run([_build_c_host(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c_host_smoke: OK" in r.stdout, r.stdout + r.stderr


def test_bench_pmc_traffic_is_tagged_with_the_kernel_source_digest(tmp_path, monkeypatch):
    """bench.py joins the committed PMC summary by kernel-name prefix and tags it stale when the summary's kernel-source digest is not
    the one of the library being timed."""
    import json
    import bench
    from crfp_amd import _lib
    prof = tmp_path / "profiles"
    prof.mkdir()
    rows = [{"kernel": "crfp::dcn3_kernel<true, 3>", "calls": 24, "hbm_MB_per_launch_corrected": 231.5},
            {"kernel": "conv3x3_split8_kernel<8>", "calls": 492, "hbm_MB_per_launch_corrected": 81.2},
            {"kernel": "conv3x3_split_kernel<1, 1, 2>", "calls": 120, "hbm_MB_per_launch_corrected": 58.8}]
    (prof / "pmc_summary_latest.json").write_text(json.dumps(rows))
    import benchlegs
    monkeypatch.setattr(benchlegs, "ROOT", str(tmp_path))   # round 5: pmc_traffic lives in benchlegs.py (bench.py re-exports it)
    tr = bench.pmc_traffic("dcnv2_shared_c4_fused", "f32")
    assert tr and abs(tr["bytes_per_launch"] - 231.5e6) < 1 and tr["stale"] is None          # no digest file: unknown
    (prof / "pmc_summary_latest.meta.json").write_text(json.dumps({"kernels_src_sha": _lib.kernel_source_digest()}))
    assert bench.pmc_traffic("dcnv2_shared_c4_fused", "f32")["stale"] is False
    conv = bench.pmc_traffic("conv3x3_mfma", "f32")
    assert abs(conv["bytes_per_launch"] - (81.2e6 * 492 + 58.8e6 * 120) / 612) < 1       # call-weighted over the family's kernels
    (prof / "pmc_summary_latest.meta.json").write_text(json.dumps({"kernels_src_sha": "0" * 12}))
    assert bench.pmc_traffic("conv3x3_mfma", "f32")["stale"] is True
    assert bench.pmc_traffic("conv3x3_mfma", "f32", (270, 480)) is None                   # another geometry: bytes do not transfer
    assert bench.pmc_traffic("conv3x3_mfma", "f32", (180, 320), 4) is None                # no fp32 lock-step summary


def test_committed_pmc_summaries_belong_to_the_shipped_kernels():
    """The summaries bench.py reads must have been collected with the kernel sources in the tree (else the driver's line says stale)."""
    import json
    from crfp_amd import _lib
    for name in ("pmc_summary_latest", "pmc_summary_latest_bf16", "pmc_summary_latest_c4"):
        meta = json.load(open(os.path.join(ROOT, "profiles", name + ".meta.json")))
        assert meta["kernels_src_sha"] == _lib.kernel_source_digest(), f"profiles/{name}.json is stale: re-run tools/measure_round.sh and copy the summaries"


def test_main_builds_the_wiring_the_environment_names(monkeypatch):
    """main.py:34-35: the reference switches between CRFP_DSV and CRFP_DSV_CRA by (un)commenting a factory line; here CRFP_MODEL names it."""
    import types
    from crfp_amd import main
    args = types.SimpleNamespace(y_only=False, hr_dcn=True, offset_prop=True)
    monkeypatch.delenv("CRFP_MODEL", raising=False)
    assert type(main.build_model(args, torch.device("cpu"), None)).__name__ == "CRFP_DSV"
    monkeypatch.setenv("CRFP_MODEL", "CRFP_DSV_CRA")
    m = main.build_model(args, torch.device("cpu"), None)
    assert type(m).__name__ == "CRFP_DSV_CRA" and m.has_engine() and "conv_tttf_2.weight" in m.state_dict()
    monkeypatch.setenv("CRFP_MODEL", "BasicVSR")
    with pytest.raises(SystemExit):
        main.build_model(args, torch.device("cpu"), None)


def test_host_logic_under_address_and_ub_sanitizers():
    """SURVEY.md section 5 / VERDICT r5 item 8: the library's host logic -- sizing, Layout arenas, parameter tables, argument checks, and the host
    side of whole engine calls (launch-argument tables, side-stream fork / join, crfp_shutdown) -- compiled host-only with
    -fsanitize=address,undefined and driven by tools/asan_host/host_check.cpp on a stub HIP runtime (no device code, no GPU; CPU build
    container only).  `make asan` fails on any sanitizer report or failed expectation."""
    import shutil
    import subprocess
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "crfp_amd", "csrc"), "-j8", "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:] + r.stderr[-3000:])
    assert "0 failed expectations" in r.stdout and "launches enqueued on the stub runtime" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr
