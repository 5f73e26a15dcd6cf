"""CPU: the C-ABI library loads and exports every symbol include/pycllp_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from pycllp_amd import _native


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pycllp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pycllp_hip_\w+)\s*\(", text)))


def test_header_and_loader_agree():
    assert declared_symbols() == sorted(_native.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(_native.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), name


def test_defaults_and_argument_errors_without_gpu():
    L = _native.lib()
    assert L.pycllp_hip_abi_version() == 1
    o = _native.default_opts()
    assert (o.eps, o.delta, o.r, o.pivot_floor, o.refine_tol) == (1e-10, 0.02, 0.9, 1e-6, 1e-11)
    assert (o.max_iter, o.max_refine, o.flags) == (200, -1, 0)     # -1 = PYCLLP_MAX_REFINE_AUTO: 5 plain / 20 HSD, resolved in C
    assert L.pycllp_hip_dense_max_rows() == 256 and L.pycllp_hip_dense_max_cols() == 1280
    h = ctypes.c_void_p()
    # NULL matrix / bad sizes are rejected before any HIP call
    assert L.pycllp_hip_dense_init(3, 3, None, None, ctypes.byref(h)) == -1
    assert L.pycllp_hip_dense_init(0, 3, ctypes.c_void_p(8), None, ctypes.byref(h)) == -1
    assert b"bad argument" in L.pycllp_hip_last_error()
    # sizes outside the compiled kernels -> PYCLLP_E_UNSUPPORTED, surfaced as NotImplementedError
    rc = L.pycllp_hip_dense_init(257, 40, ctypes.c_void_p(8), None, ctypes.byref(h))
    assert rc == -2
    with pytest.raises(NotImplementedError):
        _native.check(rc, "init")
    with pytest.raises(TypeError):
        _native.default_opts(bogus=1)


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_native.Opts) == 5 * 8 + 4 * 4
