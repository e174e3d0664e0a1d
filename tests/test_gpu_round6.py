"""GPU (-m gpu), round 6: the persistent form of the fused offset-head + DCNv2 kernel (gather.hip dcn_fused_kernel<8, true>: launches with more
tiles than CUs) and the other round-6 kernel changes, bit for bit against the paths they replace."""
import pytest

from test_gpu_parity import _golden_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_persistent_fused_dcn_is_bit_identical_to_the_two_kernel_path(storage):
    """A 101 x 170 clip is 202 x 340 at 2x resolution = 26 x 11 = 286 tiles of 8 x 32 pixels (ragged in x and y) > 256 workgroups: the persistent
    form of dcn_fused_kernel (one workgroup per CU walks its XCD band of the tile list, the next tile's halo tile and weight stage 0
    requested in the current tile's tail) gives every workgroup of XCDs 0-5 a second tile.  Same arithmetic in the same order as conv3x3 + dcn_g8_pipe (CRFP_DCN_FUSED=0): not one bit
    of the clip may differ.  (The lock-step batch at 4 x 180 x 320 -- tile ids that run across batch items -- is
    test_gpu_round5.py::test_config4_lockstep_batch_at_the_real_shape.)"""
    env = {"CRFP_CHECK_GEOM": "101,170,3"}
    if storage == "bf16":
        env["CRFP_CHECK_STORAGE"] = "bf16"
    two_kernel = _golden_check(dict(env, CRFP_DCN_FUSED="0"), want="DIGEST")
    # the product takes the XCD-banded one-tile form here (the persistent form starts at six rounds of the chip) ...
    assert _golden_check(env, want="DIGEST") == two_kernel
    # ... and the lab library is told to take the persistent form for every launch with more tiles than workgroups
    assert _golden_check(dict(env, CRFP_DF_PS_MIN_TILES="257"), lab=True, want="DIGEST") == two_kernel


@pytest.mark.parametrize("env", [{"CRFP_NARROW_CHAIN": "0"}, {"CRFP_NARROW_CHAIN": "3"}, {"CRFP_NARROW_SEQ": "0", "CRFP_NARROW_CHAIN": "0"},
                                 {"CRFP_STATE_FROM_EPILOGUE": "0"}])
@pytest.mark.parametrize("geom", [None, "33,47,3"])
@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_round6_stencil_forms_are_bit_identical_to_the_launches_they_replace(env, geom, storage):
    """The 8x stencils of round 6 -- the quad-sequential kernels, forward_resblocks_3 (and, lab only, dcn_3's conv chain) as one launch, the new
    state written by the last conv's epilogue -- apply each conv's arithmetic in its single kernel's order: switching any of them off in the lab
    library (the product compiles the choice in) must not change one bit of the clip.  Geometries: the 20 x 36 golden clip (160 x 288 at 8x:
    ragged 60- and 64-pixel tiles, every tile touches a border) and 33 x 47 (264 x 376: interior tiles take the fast paths)."""
    base = {"CRFP_CHECK_GEOM": geom} if geom else {}
    if storage == "bf16":   # the bf16 build ships both chains and the quad-sequential form of the multi-quad stencils
        base["CRFP_CHECK_STORAGE"] = "bf16"
    assert _golden_check(base, lab=True, want="DIGEST") == _golden_check(dict(base, **env), lab=True, want="DIGEST")
    assert _golden_check(base, want="DIGEST") == _golden_check(base, lab=True, want="DIGEST")   # and the product computes the same clip


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_k_sliced_fnet_layers(storage):
    """Lab library, CRFP_CONV_KSPLIT=2: FNet's small maps run their convs in K slices (conv_auto_ksplit / launch_conv_ksplit; the slices are added by
    the pool / resize pass behind the layer or by launch_ksplit_reduce) -- measured and not shipped (conv_mfma.hip).  The sliced layers move the golden
    clip by summation order only: inside the golden's tolerance of the reference, but not the bits of the layers in one piece (so the slices did run);
    the product library never slices: its clip is the lab default's, bit for bit."""
    env = {"CRFP_CHECK_STORAGE": "bf16"} if storage == "bf16" else {}
    tol = 6e-2 if storage == "bf16" else 2e-4
    assert float(_golden_check(dict(env, CRFP_CONV_KSPLIT="2"), lab=True).split()[1]) < tol
    whole = _golden_check(env, lab=True, want="DIGEST")
    if storage == "f32":   # (bf16 storage rounds the summation order away on a clip this small)
        assert _golden_check(dict(env, CRFP_CONV_KSPLIT="2"), lab=True, want="DIGEST") != whole
    assert _golden_check(dict(env, CRFP_CONV_KSPLIT="2"), want="DIGEST") == whole
