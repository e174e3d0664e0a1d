"""Host-side mirror of the reference's ``model`` package for the CRFP_DSV inference path.
``from crfp_amd.model import CRFP`` (or put ``crfp_amd/`` on PYTHONPATH and keep the reference's
``from model import CRFP``)."""
from . import CRFP, LTE  # noqa: F401

# the reference's rigs import these module names (test_runtime.py:1, test_video.py:2); both alias the
# streaming-capable engine-backed model here
MRCF_runtime = CRFP
MRCF_test = CRFP
