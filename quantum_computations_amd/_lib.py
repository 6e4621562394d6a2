"""ctypes binding of libqsv.so (include/qsv.h).  There is no CPU fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("QSV_LIBRARY", PKG_DIR / "libqsv.so"))

QSV_OK, QSV_EINVAL, QSV_ENOMEM, QSV_EHIP, QSV_ESTATE = 0, -1, -2, -3, -4
OPT_SPECIALIZE, OPT_UNROLL, OPT_GRID_CAP, OPT_NONTEMPORAL, OPT_ITEM_STRIDE_BIT, OPT_TILE_REGIONS = 1, 2, 3, 4, 5, 6
OPT_KQ_VARIANT, OPT_PLANE_KERNEL, OPT_READOUT_VARIANT, OPT_COMPLEX_PRODUCT, OPT_SEQUENCE_WORK = 7, 8, 9, 10, 11
OPT_TILE_SEQUENCE_GATES = 12

RANK_NEEDS_OMEGA = (1 << 64) - 1     # qsv_tensor_rsvd_split without a test matrix: call again with one

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
    "qsv_rebind_view": [_state_p, C.c_int, C.c_void_p, C.c_uint64],
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
    "qsv_apply_sequence": [_state_p, C.c_int, _int_p, C.c_int, _int_p, _int_p, C.c_void_p, _int_p],
    "qsv_permute": [_state_p, _int_p],
    "qsv_measure": [_state_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, _int_p, _dbl_p, _dbl_p],
    "qsv_measure_probs": [_state_p, C.c_int, C.c_void_p, C.c_void_p, _dbl_p, _dbl_p],
    "qsv_collapse": [_state_p, C.c_int, C.c_void_p, C.c_double],
    "qsv_insert": [_state_p, C.c_int, C.c_void_p],
    "qsv_norm2": [_state_p, _dbl_p],
    "qsv_probabilities": [_state_p, _u64_p, C.c_int, _dbl_p],
    "qsv_inner": [_state_p, _state_p, _dbl_p, _dbl_p],
    "qsv_expect_pauli": [_state_p, C.c_int, _int_p, C.c_char_p, _dbl_p, _dbl_p],
    "qsv_reduced_density": [_state_p, C.c_int, _int_p, C.c_void_p],
    "qsv_expect_density": [_state_p, _state_p, _dbl_p, _dbl_p],
    "qsv_sample": [_state_p, C.c_int, C.c_void_p, C.c_void_p],
    "qsv_create_qudit": [C.c_int, C.c_int, C.c_int, C.POINTER(_state_p)],
    "qsv_create_qudit_view": [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(_state_p)],
    "qsv_qudit_shape": [_state_p, _int_p, _int_p],
    "qsv_apply_mode1": [_state_p, C.c_int, C.c_void_p],
    "qsv_apply_mode1_diag": [_state_p, C.c_int, C.c_void_p],
    "qsv_apply_mode2": [_state_p, C.c_int, C.c_int, C.c_void_p],
    "qsv_apply_mode2_diag": [_state_p, C.c_int, C.c_int, C.c_void_p],
    "qsv_apply_mode2_gather": [_state_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "qsv_apply_mode2_blocks": [_state_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "qsv_mode_marginal": [_state_p, C.c_int, C.c_void_p],
    "qsv_mode_project": [_state_p, C.c_int, C.c_int, C.c_double],
    "qsv_mode_insert": [_state_p, C.c_int, C.c_void_p],
    "qsv_tensor_apply_axis": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                              C.c_uint64, C.c_void_p],
    "qsv_tensor_apply_axis_dev": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                              C.c_uint64, C.c_void_p],
    "qsv_tensor_gemm": [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p],
    "qsv_tensor_svd_split": [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_uint64,
                             C.POINTER(C.c_uint64), C.c_void_p],
    "qsv_tensor_rsvd_split": [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double,
                              C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p],
    "qsv_tensor_skinny_gemm": [C.c_int, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "qsv_tensor_release_workspace": [C.c_int],
    "qsv_tensor_scale_axis": [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p],
    "qsv_tensor_plane_diag": [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p],
    "qsv_tensor_plane_gather": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p],
    "qsv_tensor_plane_phase": [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_double],
    "qsv_tensor_plane_affine": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.POINTER(C.c_double)],
    "qsv_tensor_outer": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int],
    "qsv_tensor_take_level": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double],
    "qsv_tensor_insert_axis": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p],
    "qsv_tensor_axis_overlap": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p],
    "qsv_run_programs": [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "qsv_timer_start": [_state_p],
    "qsv_timer_stop": [_state_p, C.POINTER(C.c_float)],
    "qsv_last_kernel": [_state_p, C.c_char_p, C.c_size_t],
    "qsv_event_record": [_state_p, C.c_int],
    "qsv_event_elapsed_ms": [_state_p, C.c_int, C.c_int, C.POINTER(C.c_float)],
}

_lib = None


class QsvError(RuntimeError):
    """HIP / device failure reported by libqsv.so (QSV_EHIP)."""


def _share_hip_runtime_with_torch() -> None:
    """Make libqsv.so and PyTorch use ONE HIP runtime, whatever the import order.

    The PyTorch-ROCm wheel bundles its own ``libamdhip64.so`` (SONAME ``libamdhip64.so.7``) and asks the loader for
    it by the unversioned file name, while libqsv.so needs ``libamdhip64.so.7``.  If libqsv.so is loaded first it
    pulls in /opt/rocm's copy, torch then loads its bundled copy as a second runtime, and that one finds no GPU
    ("No HIP GPUs are available").  Pre-loading torch's copy here (without importing torch) lets the SONAME match
    satisfy libqsv.so and the file identity match satisfy torch.  Without torch installed nothing happens and
    libqsv.so uses the system runtime.
    """
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    candidate = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
    if candidate.exists():
        try:
            C.CDLL(str(candidate), mode=C.RTLD_GLOBAL)
        except OSError:
            pass


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
    _share_hip_runtime_with_torch()
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
