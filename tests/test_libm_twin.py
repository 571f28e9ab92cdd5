"""CPU: the host twins of glibc expf/cosf/sinf/hypotf (oracle/libm_twin.c) equal libm bit for
bit over the argument ranges the hot path uses.  The HIP kernels implement the same fp64
sequences (openpano_amd/csrc/devmath.hpp); tests/test_gpu_sift.py closes the loop on device."""
import ctypes as C

import numpy as np


def _sweep(lo_bits, hi_bits, step):
    return np.arange(lo_bits, hi_bits, step, dtype=np.uint32).view(np.float32)


def test_twins_match_libm(oracle):
    lib = oracle.lib
    vec = {}
    for name in ("expf", "cosf", "sinf"):
        f = getattr(lib, f"orc_{name}_twin")
        vec[name] = np.vectorize(lambda v, f=f: f(float(v)), otypes=[np.float32])
    # strided sweeps over every binade of the ranges used (exhaustive runs: see DESIGN.md numerics)
    neg = -_sweep(np.float32(1e-12).view(np.uint32), np.float32(40.0).view(np.uint32), 40009)
    assert np.array_equal(vec["expf"](neg), oracle.libm(0, neg))
    ang = _sweep(np.float32(1e-6).view(np.uint32), np.float32(7.0).view(np.uint32), 30011)
    assert np.array_equal(vec["cosf"](ang), oracle.libm(1, ang))
    assert np.array_equal(vec["sinf"](ang), oracle.libm(2, ang))
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(20000) * 0.2).astype(np.float32)
    y = (rng.standard_normal(20000) * 0.2).astype(np.float32)
    hyp = np.array([lib.orc_hypotf_twin(float(a), float(b)) for a, b in zip(x, y)], np.float32)
    assert np.array_equal(hyp, oracle.libm(3, x, y))


def test_abi_header_symbols_exported():
    """Every entry point include/openpano_hip.h declares is exported by the built library
    (load + symbol lookup only: no GPU needed, no compute call)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "openpano_hip.h")).read()
    names = sorted(set(re.findall(r"\b(op_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) > 20
    so = os.path.join(root, "openpano_amd", "libopenpano_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(so)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.op_abi_version.restype = C.c_int
    assert lib.op_abi_version() >= 2


def test_host_abi_header_symbols_exported():
    """include/pano_host.h vs openpano_amd/libpano_host.so (host-only camera estimation entries)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "pano_host.h")).read()
    names = sorted(set(re.findall(r"\b(pano_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 7
    so = os.path.join(root, "openpano_amd", "libpano_host.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(so)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
