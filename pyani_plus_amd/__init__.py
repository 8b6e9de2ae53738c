"""MI355X-native compute backend for pyani-plus's all-vs-all sketch -> ANI path.

Layout:
  csrc/      hand-written HIP kernels (gfx950) + the C ABI of include/pyani_hip.h
  _capi.py   ctypes binding (fails loudly when the library or the GPU is missing)
  engine.py  buffers and call sequencing around the C ABI
  methods/   host-side mirror of the reference's method-plugin interface
"""

__version__ = "0.1.0"

from ._capi import HipBackendError  # noqa: F401
