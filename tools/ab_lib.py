"""Dev aid: SPKM_AB_LIB=<path to another libspkm.so> makes the Python host load that build (A/B across rounds); symbols an
older build lacks become stubs that raise when called.  Import before anything touches sparsifiedkmeans_amd._lib.lib()."""
import ctypes as _C
import os

from sparsifiedkmeans_amd import _lib

if os.environ.get("SPKM_AB_LIB"):
    _lib._SO = os.environ["SPKM_AB_LIB"]
    _orig = _C.CDLL.__getattr__

    def _tolerant(self, name):
        try:
            return _orig(self, name)
        except AttributeError:
            if name.startswith("spkm_"):
                class _Stub:
                    argtypes = None
                    restype = None

                    def __call__(self, *a):
                        raise RuntimeError(name + " is not in this build")
                return _Stub()
            raise
    _C.CDLL.__getattr__ = _tolerant
