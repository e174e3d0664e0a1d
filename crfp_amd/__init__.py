"""crfp_amd: MI355X-native (gfx950) implementation of the CRFP recurrent x8 foveated-VSR
inference path (reference eugenelet/CRFP: model/CRFP.py CRFP_DSV.forward + trainer.py eval).

Layout:
  csrc/     hand-written HIP kernels + the C-ABI (``libcrfp_hip.so``; header ``include/crfp_hip.h``)
  _lib.py   ctypes binding of that C-ABI (fails loudly when the library is missing)
  ops.py    per-operator wrappers (flow_warp, DCNv2, conv3x3, upsample, metrics)
  engine.py CRFP_DSV clip/stream engine: weight packing + one C-ABI call per clip or frame
  model/    drop-in mirror of the reference's ``model`` package (CRFP.py, LTE.py)
  dcn_v2.py drop-in mirror of the third-party ``dcn_v2.DCNv2`` module
  synth.py  seeded synthetic weights / clips shared by tests, bench and fixtures
"""

__version__ = "0.1.0"
