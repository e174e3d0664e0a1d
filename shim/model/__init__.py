"""``model`` as the reference's entry scripts import it (main.py:7 ``from model import CRFP``; test_runtime.py:1
``from model import MRCF_runtime``; test_video.py:2 ``from model import MRCF_test``), bound to the MI355X build."""
import sys

from crfp_amd.model import CRFP, CRFP_runtime, LTE, MRCF_runtime, MRCF_test  # noqa: F401

# `import model.CRFP` / `from model.CRFP import flow_warp` style imports resolve to the same module objects
sys.modules.setdefault(__name__ + ".CRFP", CRFP)
sys.modules.setdefault(__name__ + ".LTE", LTE)
sys.modules.setdefault(__name__ + ".CRFP_runtime", CRFP_runtime)
sys.modules.setdefault(__name__ + ".MRCF_runtime", MRCF_runtime)
sys.modules.setdefault(__name__ + ".MRCF_test", MRCF_test)
