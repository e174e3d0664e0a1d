"""ctypes binding of the C-ABI in include/crfp_hip.h (libcrfp_hip.so, built in-tree by
``crfp_amd/csrc/Makefile``).  There is no fallback: if the library is missing every op raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRFP_HIP_LIB points at an alternative build of the same C-ABI (diagnostic variants); default: the in-tree build
LIB_PATH = os.environ.get("CRFP_HIP_LIB") or os.path.join(_HERE, "libcrfp_hip.so")

NUM_PARAMS = 118
RT_NUM_PARAMS = 158   # CRFP_RT_NUM_PARAMS
DSV_Y_ONLY, DSV_STRICT_F32, DSV_SINGLE_STREAM = 1, 2, 4   # flags of crfp_dsv_forward_clip / crfp_dsv_stream_frame
DSV_INPUTS_RESIDENT = 8   # crfp_dsv_stream_frame only

c_float_p = C.POINTER(C.c_float)


class ProfRecord(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int), ("total_ms", C.c_double),
                ("bytes", C.c_double), ("flops", C.c_double)]


# name -> (restype, argtypes); mirrors include/crfp_hip.h one to one
SIGNATURES = {
    "crfp_version": (C.c_int, []),
    "crfp_last_error_string": (C.c_char_p, []),
    "crfp_shutdown": (C.c_int, []),
    "crfp_flow_warp_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "crfp_flow_warp_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_dcnv2_workspace_bytes": (C.c_size_t, [C.c_int] * 7),
    "crfp_dcnv2_forward_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 9 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_dcnv2_shared_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "crfp_dcnv2_shared_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_conv3x3_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "crfp_conv3x3_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_conv3x3_ex_workspace_bytes": (C.c_size_t, [C.c_int] * 9),
    "crfp_conv3x3_ex_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_float] + [C.c_int] * 4 +
                            [C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_conv3x3_packed_bytes": (C.c_size_t, [C.c_int] * 2),
    "crfp_conv3x3_pack_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 2 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_conv3x3_packed_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_float, C.c_void_p]),
    "crfp_dcnv2_g8_packed_bytes": (C.c_size_t, []),
    "crfp_dcnv2_g8_pack_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_size_t, C.c_void_p]),
    "crfp_dcnv2_g8_packed_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_upsample_bilinear_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 6 + [C.c_float] * 3 + [C.c_void_p]),
    "crfp_upsample_bilinear_ac_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 6 + [C.c_float, C.c_void_p]),
    "crfp_convkxk_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 7 + [C.c_void_p]),
    "crfp_spynet_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "crfp_spynet_forward": (C.c_int, [C.POINTER(C.c_void_p)] + [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_psnr_partial_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]),
    "crfp_avgpool2_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_void_p]),
    "crfp_fovea_head_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "crfp_fovea_head_f32": (C.c_int, [C.c_void_p] * 10 + [C.c_int] * 4 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_psnr_ssim_partial_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_float] * 2 + [C.c_void_p]),
    "crfp_dsv_param_name": (C.c_char_p, [C.c_int]),
    "crfp_dsv_param_numel": (C.c_int, [C.c_int, C.c_int]),
    "crfp_dsv_packed_weight_bytes": (C.c_size_t, [C.c_int]),
    "crfp_dsv_pack_weights": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_dsv_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "crfp_dsv_status_offset": (C.c_size_t, [C.c_int] * 3),
    "crfp_dsv_forward_clip": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 3 +
                              [C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_dsv_batch_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "crfp_dsv_batch_status_offset": (C.c_size_t, [C.c_int] * 4),
    "crfp_dsv_forward_batch": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 4 +
                               [C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_dsv_stream_frame": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 3 +
                              [C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_dsv_stream_batch": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 4 +
                              [C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_fnet_forward": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_rt_param_name": (C.c_char_p, [C.c_int]),
    "crfp_rt_param_numel": (C.c_int, [C.c_int, C.c_int]),
    "crfp_rt_packed_weight_bytes": (C.c_size_t, [C.c_int]),
    "crfp_rt_pack_weights": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_rt_workspace_bytes": (C.c_size_t, [C.c_int] * 7),
    "crfp_rt_forward_clip": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "crfp_prof_enable": (C.c_int, [C.c_int]),
    "crfp_prof_reset": (C.c_int, []),
    "crfp_debug_side_tables": (C.c_int, []),
    "crfp_prof_report": (C.c_int, [C.POINTER(ProfRecord), C.c_int]),
    "crfp_dsv_debug_fetch": (C.c_int, [C.c_char_p] + [C.c_int] * 3 + [C.c_void_p, C.c_void_p] +
                             [C.POINTER(C.c_int)] * 3 + [C.c_void_p]),
}
# the CRFP_DSV_CRA wiring: its own parameter table / packed weights / workspace, the forward call of crfp_dsv_forward_batch
CRA_NUM_PARAMS = 144
for _n in ("param_name", "param_numel", "packed_weight_bytes", "pack_weights", "batch_workspace_bytes", "batch_status_offset", "forward_batch"):
    SIGNATURES["crfp_cra_" + _n] = SIGNATURES["crfp_dsv_" + _n]
for _n in ("packed_weight_bytes", "pack_weights", "batch_workspace_bytes", "batch_status_offset", "forward_batch"):
    SIGNATURES["crfp_cra_" + _n + "_bf16"] = SIGNATURES["crfp_dsv_" + _n]
# the CRFP_simple / CRFP wirings (round 6): CRFP_DSV's parameter names, their own shapes / packed weights / workspace, the same forward call
for _w in ("simple", "dense"):
    for _n in ("param_numel", "packed_weight_bytes", "pack_weights", "batch_workspace_bytes", "batch_status_offset", "forward_batch"):
        SIGNATURES[f"crfp_{_w}_{_n}"] = SIGNATURES["crfp_dsv_" + _n]
        if _n != "param_numel":
            SIGNATURES[f"crfp_{_w}_{_n}_bf16"] = SIGNATURES["crfp_dsv_" + _n]
# bf16-storage twins of the engine entry points (same argument lists)
for _n in ("crfp_dsv_packed_weight_bytes", "crfp_dsv_pack_weights", "crfp_dsv_workspace_bytes", "crfp_dsv_status_offset",
           "crfp_dsv_forward_clip", "crfp_dsv_stream_frame", "crfp_fnet_forward", "crfp_dsv_debug_fetch",
           "crfp_dsv_batch_workspace_bytes", "crfp_dsv_batch_status_offset", "crfp_dsv_forward_batch", "crfp_dsv_stream_batch"):
    SIGNATURES[_n + "_bf16"] = SIGNATURES[_n]

_lib = None


def lib():
    """Load libcrfp_hip.so (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C crfp_amd/csrc`. "
                "crfp_amd has no CPU or PyTorch fallback path.")
        # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import torch first so
        # that ONE HIP runtime serves both torch and this library (streams and device pointers are
        # shared across the C-ABI); loading in the other order leaves the process with a runtime
        # torch's bundled HSA stack cannot drive ("no ROCm-capable device is detected").
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def kernel_source_digest() -> str:
    """sha256 (12 hex digits) over the kernel sources (csrc/*.hip, *.h, *.inc, Makefile): names the build a profile was taken with, so that
    bench.py can tell when the committed PMC summary belongs to other kernels than the ones it is timing (`stale`)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.inc")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().crfp_last_error_string().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def prof_report(cap: int = 256):
    arr = (ProfRecord * cap)()
    n = lib().crfp_prof_report(arr, cap)
    return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms,
                 bytes=arr[i].bytes, flops=arr[i].flops) for i in range(min(n, cap))]
