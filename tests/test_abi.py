"""The C-ABI library loads on a CPU-only host and exports exactly what include/qsv.h declares.

No compute call is made here (there is no GPU in the build container): only dlopen, symbol lookup, the
version/last-error accessors, and the "no device" failure path of qsv_create, which must fail loudly.
"""
from __future__ import annotations

import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

from quantum_computations_amd import _lib

REPO = Path(__file__).resolve().parent.parent
HEADER = REPO / "include" / "qsv.h"


def declared_functions() -> list[str]:
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:int|const char \*)\s*\*?\s*(qsv_\w+)\s*\(", text, flags=re.M)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    assert len(names) >= 45
    for must in ("qsv_create", "qsv_apply_1q", "qsv_apply_2q", "qsv_apply_kq", "qsv_measure", "qsv_insert",
                 "qsv_apply_mode1", "qsv_apply_mode2", "qsv_upload", "qsv_download", "qsv_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"libqsv.so does not export {name}"
    # and nothing in the binding table that the header does not declare
    assert set(_lib.SIGNATURES) | {"qsv_version", "qsv_last_error"} == set(declared_functions())


def test_exports_are_plain_c_symbols():
    out = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True, check=True)
    exported = {line.split()[-1] for line in out.stdout.splitlines() if " T " in line}
    for name in declared_functions():
        assert name in exported, name        # unmangled: extern "C"


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.qsv_version() == 100
    assert isinstance(lib.qsv_last_error(), bytes)


def test_create_fails_loudly_without_a_device():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    handle = C.c_void_p()
    status = _lib.load().qsv_create(3, 0, C.byref(handle))
    assert status == _lib.QSV_EHIP and not handle.value
    with pytest.raises(_lib.QsvError):
        _lib.check(status)
    from quantum_computations_amd.dv_simulator import gates as G
    import numpy as np
    with pytest.raises(_lib.QsvError):          # the product path has no CPU fallback
        G.H(0).apply(np.array([1.0, 0.0]))


def test_null_handles_are_rejected_not_dereferenced():
    lib = _lib.load()
    assert lib.qsv_sync(None) == _lib.QSV_EINVAL
    assert lib.qsv_destroy(None) == _lib.QSV_OK
    n = C.c_int()
    assert lib.qsv_num_qubits(None, C.byref(n)) == _lib.QSV_EINVAL
