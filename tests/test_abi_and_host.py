"""CPU: the C-ABI library loads and exports every symbol include/spkm.h declares (no compute calls
without a GPU), status texts mirror the reference, and the host-side logic (options, synthetic
pipeline, sharding) behaves."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp


def test_library_loads_and_exports_every_declared_symbol():
    from sparsifiedkmeans_amd import _lib

    L = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 24
    missing = [s for s in names if not hasattr(L, s)]
    assert not missing, missing
    assert L.spkm_version() >= 100
    assert os.path.basename(_lib._SO) == "libspkm.so" and os.path.dirname(_lib._SO).endswith("sparsifiedkmeans_amd")


def test_status_texts_mirror_reference_messages():
    from sparsifiedkmeans_amd import _lib

    L = _lib.lib()
    assert b"did not have p rows" in L.spkm_strerror(_lib.ERR_CENTER_ROWS)      # SparseMatrixMinusCluster.c:106
    assert b"beta" in L.spkm_strerror(_lib.ERR_BETA_K)                           # :120
    assert L.spkm_strerror(_lib.ERR_LEN_LE_1) == b"Vector length must be greater than 1."   # hadamard.c:101
    assert L.spkm_strerror(_lib.ERR_NOT_POW2) == b"Vector length must be power of 2."       # hadamard.c:109
    assert L.spkm_strerror(0) == b"ok"
    assert L.spkm_reduce_len(1024, 100) == 2 * 1024 * 100 + 100 + 1


def test_no_silent_cpu_fallback_without_gpu():
    """Without a HIP device the product path must fail loudly (never route to the oracle)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sparsifiedkmeans_amd import _lib
    from sparsifiedkmeans_amd.ops import Context

    with pytest.raises(_lib.SpkmError, match="no usable HIP device"):
        Context(0)
    from sparsifiedkmeans_amd.engine import torch_context

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        torch_context(0)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under sparsifiedkmeans_amd/ may import, link or
    execute it (docstrings may mention it)."""
    import re

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sparsifiedkmeans_amd")
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|liborc|oracle/|orc_[a-z_]+\s*\(", re.M)
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc", ".sh")):
                assert not bad.search(open(os.path.join(dp, f)).read()), f


def test_null_argument_statuses_without_gpu():
    from sparsifiedkmeans_amd import _lib

    L = _lib.lib()
    assert L.spkm_ctx_create(0, None, None) == _lib.ERR_NULL_ARG
    assert L.spkm_ctx_sync(None) == _lib.ERR_NULL_ARG
    assert L.spkm_shard_info(None, None, None, None, None) == _lib.ERR_NULL_ARG
    assert L.spkm_fwht_dev(None, 8, 1, None, None) == _lib.ERR_NULL_ARG


def test_option_parsing_matches_reference_defaults():
    from sparsifiedkmeans_amd.kmeans import _DEFAULTS, _parse

    o = _parse({})
    # kmeans_sparsified.m:130-155
    assert (o["Replicates"], o["Start"], o["MaxIter"], o["PrintEvery"], o["Tol"]) == (1, "Arthur", 100, 10, 1e-6)
    assert (o["Sparsify"], o["SparsityLevel"], o["SketchType"], o["EmptyAction"]) == (False, 0.01, "auto", "singleton")
    assert (o["ColumnSamples"], o["MLcorrection"], o["MB_limit"], o["denseCenters"]) == (False, True, 500, False)
    assert o["unbiasedDistance"] and o["unbiasedInitialization"] and o["tryBuiltinMex"] and not o["FORCE_BUG"]
    assert len(_DEFAULTS) == 21
    assert _parse({"sparsify": True, "TOL": 1e-3})["Sparsify"] is True          # inputParser is case-insensitive
    with pytest.raises(TypeError):
        _parse({"nope": 1})
    with pytest.raises(ValueError):
        _parse({"SparsityLevel": 0})
    with pytest.raises(ValueError):
        _parse({"EmptyAction": "explode"})
    with pytest.raises(ValueError):
        _parse({"Display": "loud"})


def test_small_p_rounding_and_gamma_quirk():
    from sparsifiedkmeans_amd import synth

    assert synth.small_p_of(0.05, 1024) == 51 and synth.small_p_of(0.05, 512) == 26
    assert synth.small_p_of(1e-9, 512) == 1                                      # max(1, round(.)) :324
    assert synth.small_p_of(2.5 / 8, 8) == 3                                     # MATLAB round: half away from zero


def test_synthetic_pipeline_properties(oracle):
    from sparsifiedkmeans_amd import synth

    d = synth.sparsified_gmm_host(p=100, n=200, K=4, gamma=0.1, seed=3, fwht=oracle.fwht)
    Y, p2, s = d["Y"], d["p2"], d["s"]
    assert p2 == 128 and s == 13 and Y.shape == (128, 200)
    assert np.all(np.diff(Y.indptr) == s)                        # exactly s entries per column
    for j in range(0, 200, 17):
        rows = Y.indices[Y.indptr[j]:Y.indptr[j + 1]]
        assert np.all(np.diff(rows) > 0)                          # ascending, distinct
        assert np.array_equal(Y.data[Y.indptr[j]:Y.indptr[j + 1]], d["Xmixed"][rows, j] / (np.float64(s) / p2))
    assert d["gamma"] == s / 100                                   # kmeans_sparsified.m:329 divides by p, not p2
    # the transform is orthogonal: column norms survive mixing (up to the (1+2eps) pre-scale)
    assert np.allclose(np.linalg.norm(d["Xmixed"], axis=0), np.linalg.norm(d["X"], axis=0), rtol=1e-12)
    # sampler: every row is picked about equally often
    rng = np.random.default_rng(0)
    rows = synth.sample_rows(rng, 64, 8, 4000)
    cnt = np.bincount(rows.ravel(), minlength=64)
    assert abs(cnt.mean() - 500) < 1e-9 and cnt.min() > 400 and cnt.max() < 600


def test_shard_ranges_cover_exactly():
    from sparsifiedkmeans_amd.distributed import reduce_layout, shard_range

    for n, w in [(10, 3), (100000000, 8), (7, 8), (1, 1)]:
        edges = [shard_range(n, r, w) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
    lay = reduce_layout(1024, 100)
    assert lay["length"] == 204901 and lay["nk"] == slice(204800, 204900)


def test_ops_reject_non_sparse_like_the_mex(oracle):
    import sparsifiedkmeans_amd as S

    with pytest.raises(TypeError, match="sparse"):
        S.SparseMatrixColumnNormSq(np.zeros((4, 4)))
    with pytest.raises(TypeError):
        S.hadamard(sp.csc_matrix(np.eye(4)))


def test_staging_copy_is_exact_and_leaves_the_thread_count_alone(monkeypatch):
    """engine._parallel_host_copy: the host-side staging copy of the streamed ingest runs torch's own copy with a bounded
    number of intra-op threads (SPKM_COPY_THREADS overrides) and restores the process's setting afterwards."""
    import numpy as np
    import torch

    from sparsifiedkmeans_amd import engine

    before = torch.get_num_threads()
    for dt in (torch.float64, torch.float32, torch.uint8, torch.int16):
        src = (torch.arange(4099 * 1031, dtype=torch.float64) % 251).to(dt).view(4099, 1031)       # > 8 MB for f64 / f32
        dst = torch.empty_like(src)
        engine._parallel_host_copy(dst, src)
        assert torch.equal(dst, src)
        assert torch.get_num_threads() == before
    monkeypatch.setenv("SPKM_COPY_THREADS", "3")
    assert engine._copy_threads() == 3
    monkeypatch.delenv("SPKM_COPY_THREADS")
    assert 1 <= engine._copy_threads() <= 16
    small = torch.ones(8, 8)
    out = torch.empty_like(small)
    engine._parallel_host_copy(out, small)
    assert np.array_equal(out.numpy(), small.numpy())


def test_mex_gateways_type_check_against_the_c_abi():
    """matlab/*.c (the five mex gateways with the reference's file names, and the engine gateway
    spkm_lloyd.c) go through a compiler -- `gcc -fsyntax-only -Wall -Wextra -Werror` against include/spkm.h and a
    DECLARATIONS-ONLY list of the documented mx*/mex* functions they call (tests/native/mex_decls.h: a compile check, never
    linked, never seen by the oracle or by any build).  Every spkm_* call in the gateways is thereby checked for argument
    count and types against the header the library is built from; the gateway entry point has the reference's signature
    (SparseMatrixMinusCluster.c:44-45)."""
    import glob
    import re
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    srcs = sorted(glob.glob(os.path.join(root, "matlab", "*.c")))
    assert {os.path.basename(f) for f in srcs} >= {"SparseMatrixMinusCluster.c", "SparseMatrixInnerProduct.c", "SparseMatrixColumnNormSq.c",
                                                   "hadamard.c", "hadamard_pthreads.c", "spkm_lloyd.c"}
    inc = ["-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "matlab"), "-I" + os.path.join(root, "tests", "native", "mexcheck")]
    hip_inc = "/opt/rocm/include"
    for f in srcs:
        text = open(f).read()
        if "hip/hip_runtime_api.h" in text and not os.path.isdir(os.path.join(hip_inc, "hip")):
            continue                                                     # (the engine gateway also needs the HIP runtime's header)
        r = subprocess.run([gcc, "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", *inc, "-isystem", hip_inc, f],
                           capture_output=True, text=True)
        assert r.returncode == 0, f"{os.path.basename(f)}:\n{r.stderr[:3000]}"
        assert re.search(r"void\s+mexFunction\s*\(\s*int\s+nlhs\s*,\s*mxArray\s*\*\s*plhs\[\]\s*,\s*int\s+nrhs\s*,\s*const\s+mxArray\s*\*\s*prhs\[\]\s*\)", text), f
        # every C-ABI name a gateway uses is one the header declares
        from sparsifiedkmeans_amd import _lib
        code = re.sub(r"/\*.*?\*/|//[^\n]*", " ", text, flags=re.S)         # (usage texts in comments name the MATLAB function)
        code = re.sub(r'"(?:\\.|[^"\\])*"', '""', code)
        used = set(re.findall(r"\b(spkm_[a-z0-9_]+)\s*\(", code)) - {"spkm_mex_ctx", "spkm_mex_atexit"}
        local = set(re.findall(r"^static[^\n(]*\b(spkm_[a-z0-9_]+)\s*\(", text, re.M))
        assert used - local <= set(_lib.declared_symbols()), sorted(used - local - set(_lib.declared_symbols()))
