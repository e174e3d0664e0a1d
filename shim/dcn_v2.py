"""``from dcn_v2 import DCNv2`` (reference model/CRFP.py:6, test_runtime.py:11) -> the HIP-backed module."""
from crfp_amd.dcn_v2 import DCNv2  # noqa: F401
