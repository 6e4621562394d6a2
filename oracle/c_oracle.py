"""TEST INFRASTRUCTURE ONLY — ctypes loader of oracle/libqsv_oracle.so (C restatement, OpenMP).

Used by tests (pinned to the golden vectors) and by ``bench.py``'s ``cpu_baseline`` leg.  Never imported by the
product package.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "libqsv_oracle.so"
_lib = None


def build(force: bool = False) -> Path:
    src = HERE / "csrc" / "qsv_oracle.c"
    if force or not LIB.exists() or LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "-B", "libqsv_oracle.so"], check=True, capture_output=True)
    return LIB


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB))
        _lib.oracle_apply_1q.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.oracle_apply_2q.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.oracle_dense_matvec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.oracle_norm2.argtypes = [C.c_void_p, C.c_int]
        _lib.oracle_norm2.restype = C.c_double
        _lib.oracle_threads.argtypes = [C.c_int]
        _lib.oracle_threads.restype = C.c_int
        for f in (_lib.oracle_apply_1q, _lib.oracle_apply_2q, _lib.oracle_dense_matvec):
            f.restype = None
    return _lib


def threads(requested: int = 0) -> int:
    """Set (if > 0) and return the OpenMP thread count of the C oracle."""
    return int(load().oracle_threads(int(requested)))


def apply_gate_inplace(state: np.ndarray, matrix: np.ndarray, indices: list[int]) -> None:
    """In-place 1- or 2-qubit gate on a contiguous complex128 ket."""
    assert state.dtype == np.complex128 and state.flags.c_contiguous
    n = state.size.bit_length() - 1
    m = np.ascontiguousarray(matrix, dtype=np.complex128)
    lib = load()
    if len(indices) == 1:
        lib.oracle_apply_1q(state.ctypes.data, n, int(indices[0]), m.ctypes.data)
    elif len(indices) == 2:
        lib.oracle_apply_2q(state.ctypes.data, n, int(indices[0]), int(indices[1]), m.ctypes.data)
    else:
        raise ValueError("the C oracle restates 1- and 2-qubit gates")


def run_circuit_inplace(ops: list[dict], state: np.ndarray) -> None:
    for op in ops:
        apply_gate_inplace(state, op["matrix"], op["indices"])


def dense_matvec(u: np.ndarray, ket: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(u, dtype=np.complex128)
    ket = np.ascontiguousarray(ket, dtype=np.complex128)
    out = np.empty_like(ket)
    load().oracle_dense_matvec(u.ctypes.data, ket.ctypes.data, out.ctypes.data, ket.size.bit_length() - 1)
    return out
