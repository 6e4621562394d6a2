"""ctypes binding of libqsv.so (include/qsv.h).  There is no CPU fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("QSV_LIBRARY", PKG_DIR / "libqsv.so"))

QSV_OK, QSV_EINVAL, QSV_ENOMEM, QSV_EHIP, QSV_ESTATE = 0, -1, -2, -3, -4
OPT_SPECIALIZE, OPT_UNROLL, OPT_GRID_CAP, OPT_NONTEMPORAL = 1, 2, 3, 4

_state_p = C.c_void_p
_dbl_p = C.POINTER(C.c_double)
_int_p = C.POINTER(C.c_int)
_u64_p = C.POINTER(C.c_uint64)

# name -> argument types; every function returns int except where noted.  Mirrors include/qsv.h one to one
# (tests/test_abi.py parses the header and checks this table and the library's exports against it).
SIGNATURES: dict[str, list] = {
    "qsv_device_count": [_int_p],
    "qsv_create": [C.c_int, C.c_int, C.POINTER(_state_p)],
    "qsv_create_view": [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(_state_p)],
    "qsv_destroy": [_state_p],
    "qsv_set_stream": [_state_p, C.c_void_p],
    "qsv_set_option": [_state_p, C.c_int, C.c_int64],
    "qsv_num_qubits": [_state_p, _int_p],
    "qsv_num_amps": [_state_p, _u64_p],
    "qsv_device_ptr": [_state_p, C.POINTER(C.c_void_p)],
    "qsv_sync": [_state_p],
    "qsv_set_basis": [_state_p, C.c_uint64],
    "qsv_upload": [_state_p, C.c_void_p, C.c_uint64, C.c_uint64],
    "qsv_download": [_state_p, C.c_void_p, C.c_uint64, C.c_uint64],
    "qsv_copy": [_state_p, _state_p],
    "qsv_fill_random": [_state_p, C.c_uint64, C.c_uint64, _dbl_p],
    "qsv_scale": [_state_p, C.c_double, C.c_double],
    "qsv_apply_1q": [_state_p, C.c_int, C.c_void_p],
    "qsv_apply_2q": [_state_p, C.c_int, C.c_int, C.c_void_p],
    "qsv_apply_diag_1q": [_state_p, C.c_int, C.c_void_p],
    "qsv_apply_diag_2q": [_state_p, C.c_int, C.c_int, C.c_void_p],
    "qsv_apply_cx": [_state_p, C.c_int, C.c_int],
    "qsv_apply_swap": [_state_p, C.c_int, C.c_int],
    "qsv_apply_controlled_1q": [_state_p, C.c_int, _int_p, C.c_int, C.c_void_p],
    "qsv_apply_mcphase": [_state_p, C.c_int, _int_p, C.c_double, C.c_double],
    "qsv_apply_kq": [_state_p, C.c_int, _int_p, C.c_void_p],
    "qsv_permute": [_state_p, _int_p],
    "qsv_measure": [_state_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, _int_p, _dbl_p, _dbl_p],
    "qsv_measure_probs": [_state_p, C.c_int, C.c_void_p, C.c_void_p, _dbl_p, _dbl_p],
    "qsv_collapse": [_state_p, C.c_int, C.c_void_p, C.c_double],
    "qsv_insert": [_state_p, C.c_int, C.c_void_p],
    "qsv_norm2": [_state_p, _dbl_p],
    "qsv_probabilities": [_state_p, _u64_p, C.c_int, _dbl_p],
    "qsv_inner": [_state_p, _state_p, _dbl_p, _dbl_p],
    "qsv_create_qudit": [C.c_int, C.c_int, C.c_int, C.POINTER(_state_p)],
    "qsv_create_qudit_view": [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(_state_p)],
    "qsv_qudit_shape": [_state_p, _int_p, _int_p],
    "qsv_apply_mode1": [_state_p, C.c_int, C.c_void_p],
    "qsv_apply_mode1_diag": [_state_p, C.c_int, C.c_void_p],
    "qsv_apply_mode2": [_state_p, C.c_int, C.c_int, C.c_void_p],
    "qsv_apply_mode2_diag": [_state_p, C.c_int, C.c_int, C.c_void_p],
    "qsv_tensor_apply_axis": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                              C.c_uint64, C.c_void_p],
    "qsv_timer_start": [_state_p],
    "qsv_timer_stop": [_state_p, C.POINTER(C.c_float)],
    "qsv_last_kernel": [_state_p, C.c_char_p, C.c_size_t],
    "qsv_event_record": [_state_p, C.c_int],
    "qsv_event_elapsed_ms": [_state_p, C.c_int, C.c_int, C.POINTER(C.c_float)],
}

_lib = None


class QsvError(RuntimeError):
    """HIP / device failure reported by libqsv.so (QSV_EHIP)."""


def load() -> C.CDLL:
    """Load libqsv.so once.  Raises if it has not been built -- the product has no other execution path."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP library first (python -m quantum_computations_amd.build, or "
            "__graft_entry__.build()).  quantum_computations_amd has no CPU fallback."
        )
    lib = C.CDLL(str(LIB_PATH))
    lib.qsv_version.restype = C.c_int
    lib.qsv_version.argtypes = []
    lib.qsv_last_error.restype = C.c_char_p
    lib.qsv_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status: int) -> None:
    """Map a status code to the exception class the reference raises for the same mistake."""
    if status == QSV_OK:
        return
    msg = load().qsv_last_error().decode("utf-8", "replace")
    if status == QSV_EINVAL:
        raise ValueError(msg)           # gates.py:9-19, numpy_quantum.py:229-232
    if status == QSV_ENOMEM:
        raise MemoryError(msg)
    if status == QSV_ESTATE:
        raise TypeError(msg)
    raise QsvError(msg)


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args))


def device_count() -> int:
    n = C.c_int(0)
    status = load().qsv_device_count(C.byref(n))
    return n.value if status == QSV_OK else 0
