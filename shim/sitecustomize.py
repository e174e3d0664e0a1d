"""Activated by ``PYTHONPATH=<repo>/shim``: Python imports this module at start-up, BEFORE it puts the script's own
directory in front of sys.path.  It installs a meta-path finder so that the three top-level names the reference's entry
scripts bind for the inference path --

    from model import CRFP / MRCF_runtime / MRCF_test      (main.py:7, test_runtime.py:1, test_video.py:2)
    from dcn_v2 import DCNv2                                (model/CRFP.py:6, test_runtime.py:11)
    from pytorch_memlab import LineProfiler, MemReporter    (test_runtime.py:9-10)

-- resolve to this directory even when the script is started from the reference's checkout (``bash eval.sh`` runs
``python3 main.py`` there, and the script directory outranks PYTHONPATH).  Nothing else is shadowed: ``utils``,
``dataset``, ``trainer``, ``loss`` and ``option`` stay the reference's own files.
"""
import importlib.abc
import importlib.util
import os
import sys

_SHIM = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_SHIM)
_NAMES = {"model": os.path.join(_SHIM, "model", "__init__.py"),
          "dcn_v2": os.path.join(_SHIM, "dcn_v2.py"),
          "pytorch_memlab": os.path.join(_SHIM, "pytorch_memlab.py")}


class _ShimFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path=None, target=None):
        f = _NAMES.get(name)
        if f is None:
            return None
        search = [os.path.dirname(f)] if name == "model" else None
        return importlib.util.spec_from_file_location(name, f, submodule_search_locations=search)


if not any(isinstance(f, _ShimFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _ShimFinder())
if _REPO not in sys.path:
    sys.path.append(_REPO)   # makes `crfp_amd` importable; appended, so it shadows nothing

# chain to the interpreter's own sitecustomize (this file took its place on sys.path)
for _d in sys.path:
    _f = os.path.join(_d, "sitecustomize.py") if _d else ""
    if _f and os.path.isfile(_f) and os.path.abspath(_f) != os.path.abspath(__file__):
        try:
            exec(compile(open(_f).read(), _f, "exec"), {"__name__": "sitecustomize", "__file__": _f})
        except Exception:   # the system hook is best-effort in the stock file as well
            pass
        break
