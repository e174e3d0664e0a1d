"""Host-side mirror of the reference's ``model`` package for the CRFP_DSV inference path.
``from crfp_amd.model import CRFP`` (or put ``crfp_amd/`` on PYTHONPATH and keep the reference's
``from model import CRFP``)."""
from . import CRFP, CRFP_runtime, LTE  # noqa: F401

# the reference's rigs import these module names: test_video.py:2 drives the streaming model (model/CRFP_test.py) = the
# engine-backed CRFP mirror; test_runtime.py:1 drives the regional-DCN benchmark wiring (model/CRFP_runtime.py)
MRCF_runtime = CRFP_runtime
MRCF_test = CRFP
