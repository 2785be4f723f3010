"""MI355X-native Lloyd-iteration engine for sparsified K-means.

Drop-in for the native layer and host entry point of stephenbeckr/SparsifiedKMeans v2.1:
  * ops.*                  the five mex operators (same names / semantics), HIP kernels underneath
  * kmeans_sparsified()    the reference's entry point and options (kmeans_sparsified.m:1,130-155)
  * engine.LloydEngine     the fused device-resident iteration the reference does not have
The C ABI lives in include/spkm.h / libspkm.so; there is no CPU fallback anywhere in this package.
"""
from . import _lib  # noqa: F401
from .ops import (Context, SparseMatrixColumnNormSq, SparseMatrixInnerProduct,  # noqa: F401
                  SparseMatrixMinusCluster, default_context, hadamard, hadamard_pthreads)

__version__ = "0.1.0"


def __getattr__(name):
    # torch-dependent parts are imported on first use so that `import sparsifiedkmeans_amd`
    # stays cheap for pure host-array users
    if name in ("kmeans_sparsified", "findClusterAssignments"):
        from . import kmeans as _k

        return getattr(_k, name)
    if name in ("LloydEngine", "Shard", "torch_context"):
        from . import engine as _e

        return getattr(_e, name)
    raise AttributeError(name)
