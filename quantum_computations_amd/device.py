"""Device-resident registers: thin Python handles over ``qsv_state`` (include/qsv.h).

``DeviceState`` is what ``Gate.apply`` receives when the caller wants the register to stay in HBM between gates
(the reference passes a NumPy ket to every ``apply`` -- ``simulators/dv_simulator/simulator.py:47`` -- which
would cost one upload and one download per gate here).  Host arrays are only touched by ``from_numpy`` /
``to_numpy``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

_SEQUENCE_WORK_ENV = int(os.environ.get("QSV_SEQUENCE_WORK", "0") or 0)   # see DeviceState.apply_sequence
_PACKED_SEQUENCES: dict = {}     # id(sources list of a fused block) -> its packed gate list, see DeviceState.apply_sequence
_TILE_SEQUENCE_ENV = int(os.environ.get("QSV_TILE_SEQUENCE_GATES", "12") or 0)     # qsv_api.hip: qsv_apply_sequence


def _cbuf(a, n_complex: int | None = None) -> np.ndarray:
    """Contiguous complex128 copy of ``a`` (the interleaved layout the C ABI wants)."""
    arr = np.ascontiguousarray(np.asarray(a), dtype=np.complex128)
    if n_complex is not None and arr.size != n_complex:
        raise ValueError(f"expected {n_complex} complex entries, got {arr.size}")
    return arr


def _ptr(arr: np.ndarray) -> C.c_void_p:
    # data_as() keeps a reference to the array on the returned pointer object, so a temporary passed as
    # `_ptr(_cbuf(x))` stays alive for the duration of the foreign call (c_void_p(arr.ctypes.data) would not)
    return arr.ctypes.data_as(C.c_void_p)


def _ints(values) -> "C.Array[C.c_int]":
    values = [int(v) for v in values]
    return (C.c_int * max(len(values), 1))(*values)


class DeviceState:
    """An n-qubit complex128 register in HBM (qubit 0 = most significant bit, as in the reference)."""

    def __init__(self, handle: C.c_void_p, keepalive=None, device: int = 0):
        self._h = handle
        self._keepalive = keepalive   # e.g. the torch tensor backing a view
        self.device = device          # HIP device ordinal the register lives on

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def zeros(cls, n_qubits: int, device: int = 0) -> "DeviceState":
        """|0...0> on ``device`` (``n_qubits = 0`` is the empty register ``[1.]``, simulator.py:22)."""
        h = C.c_void_p()
        _lib.call("qsv_create", int(n_qubits), int(device), C.byref(h))
        return cls(h, device=int(device))

    @classmethod
    def from_numpy(cls, ket: np.ndarray, device: int = 0) -> "DeviceState":
        ket = np.asarray(ket)
        if ket.ndim != 1:
            raise ValueError("State has wrong dimensions.")
        size = ket.shape[0]
        if size == 0 or size & (size - 1):
            raise ValueError("Given array is not a qubit state nor operator")
        st = cls.zeros(size.bit_length() - 1, device)
        buf = _cbuf(ket)
        _lib.call("qsv_upload", st._h, _ptr(buf), 0, size)
        return st

    @classmethod
    def view(cls, n_qubits: int, dev_ptr: int, capacity_amps: int, device: int = 0, stream: int = 0,
             keepalive=None) -> "DeviceState":
        """Wrap caller-owned device memory (e.g. ``tensor.data_ptr()``) without copying."""
        h = C.c_void_p()
        _lib.call("qsv_create_view", int(n_qubits), int(device), C.c_void_p(dev_ptr), int(capacity_amps),
                  C.c_void_p(stream), C.byref(h))
        return cls(h, keepalive, int(device))

    def rebind(self, n_qubits: int, dev_ptr: int, capacity_amps: int, keepalive=None) -> "DeviceState":
        """Re-point a view register at another window of caller-owned device memory (``qsv_rebind_view``): no
        allocation, no synchronisation.  The sharded register walks the slices of an exchange with one handle."""
        _lib.call("qsv_rebind_view", self._h, int(n_qubits), C.c_void_p(dev_ptr), int(capacity_amps))
        self._keepalive = keepalive
        return self

    @classmethod
    def random(cls, n_qubits: int, seed: int, device: int = 0) -> "DeviceState":
        """Normalised pseudo-random ket generated on the device (counter-based, reproducible)."""
        st = cls.zeros(n_qubits, device)
        st.fill_random(seed)
        return st

    def fill_random(self, seed: int, index_offset: int = 0, normalise: bool = True) -> float:
        n2 = C.c_double()
        _lib.call("qsv_fill_random", self._h, int(seed), int(index_offset), C.byref(n2))
        if normalise:
            _lib.call("qsv_scale", self._h, 1.0 / np.sqrt(n2.value), 0.0)
        return n2.value

    def close(self) -> None:
        if self._h is not None and self._h.value:
            _lib.load().qsv_destroy(self._h)
        self._h = None
        self._keepalive = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inspection ---------------------------------------------------------------------------
    @property
    def num_qubits(self) -> int:
        n = C.c_int()
        _lib.call("qsv_num_qubits", self._h, C.byref(n))
        return n.value

    @property
    def num_amps(self) -> int:
        n = C.c_uint64()
        _lib.call("qsv_num_amps", self._h, C.byref(n))
        return n.value

    @property
    def shape(self) -> tuple[int]:
        return (self.num_amps,)

    ndim = 1

    @property
    def device_ptr(self) -> int:
        p = C.c_void_p()
        _lib.call("qsv_device_ptr", self._h, C.byref(p))
        return p.value or 0

    def to_numpy(self) -> np.ndarray:
        out = np.empty(self.num_amps, dtype=np.complex128)
        _lib.call("qsv_download", self._h, _ptr(out), 0, out.size)
        return out

    def download(self, offset: int, count: int) -> np.ndarray:
        out = np.empty(count, dtype=np.complex128)
        _lib.call("qsv_download", self._h, _ptr(out), int(offset), int(count))
        return out

    def upload(self, amplitudes: np.ndarray, offset: int = 0) -> None:
        buf = _cbuf(amplitudes)
        _lib.call("qsv_upload", self._h, _ptr(buf), int(offset), buf.size)

    def copy_into(self, other: "DeviceState") -> "DeviceState":
        """Overwrite ``other`` (a register of at least this size on the same device) with this register."""
        _lib.call("qsv_copy", other._h, self._h)
        return other

    def copy(self) -> "DeviceState":
        other = DeviceState.zeros(DeviceState.num_qubits.fget(self), self.device)     # register qubits, whatever the view
        _lib.call("qsv_copy", other._h, self._h)
        return other

    def sync(self) -> None:
        _lib.call("qsv_sync", self._h)

    def set_option(self, option: int, value: int) -> None:
        _lib.call("qsv_set_option", self._h, int(option), int(value))
        if int(option) == _lib.OPT_SEQUENCE_WORK:
            self._sequence_work = int(value) if int(value) >= 0 else _SEQUENCE_WORK_ENV
        if int(option) == _lib.OPT_TILE_SEQUENCE_GATES:
            self._tile_sequence_gates = int(value) if int(value) >= 0 else None

    def set_stream(self, stream: int) -> None:
        _lib.call("qsv_set_stream", self._h, C.c_void_p(stream))

    def set_basis(self, index: int) -> None:
        _lib.call("qsv_set_basis", self._h, int(index))

    # ---- gates --------------------------------------------------------------------------------
    def apply_matrix(self, matrix: np.ndarray, indices) -> "DeviceState":
        """In-place ``U_full @ ket`` for a 2^k x 2^k matrix on qubits ``indices`` (``Gate.apply``)."""
        indices = [int(i) for i in indices]
        k = len(indices)
        m = _cbuf(matrix, (1 << k) ** 2)
        if k == 1:
            _lib.call("qsv_apply_1q", self._h, indices[0], _ptr(m))
        elif k == 2:
            _lib.call("qsv_apply_2q", self._h, indices[0], indices[1], _ptr(m))
        else:
            _lib.call("qsv_apply_kq", self._h, k, _ints(indices), _ptr(m))
        return self

    def apply_sequence(self, indices, sources, matrix) -> "DeviceState":
        """A fused block (``fusion.fuse_circuit``): ``matrix`` on qubits ``indices`` is the product of the gates
        ``sources``.  6-qubit blocks made of at most 12 one- and two-qubit gates (``OPT_TILE_SEQUENCE_GATES``) are applied
        as that LIST on LDS-resident tiles in one pass over the register (``qsv_apply_sequence`` -> ``k_seq_tile``: a
        fraction of the dense block's arithmetic, 1.7-1.9 ms against 2.0-2.1); 5-qubit blocks only on request (tiles, or
        the register form behind ``OPT_SEQUENCE_WORK``: neither beats their dense product); everything else as the dense
        matrix."""
        indices = [int(i) for i in indices]
        requested = getattr(self, "_tile_sequence_gates", None)        # an explicit limit also admits 5-qubit blocks
        on_tiles = (len(indices) == 6 or (len(indices) == 5 and requested is not None)) and \
            len(sources) <= (_TILE_SEQUENCE_ENV if requested is None else requested)
        in_registers = getattr(self, "_sequence_work", _SEQUENCE_WORK_ENV) > 0 and len(indices) == 5
        if (on_tiles or in_registers) and all(getattr(g, "matrix", None) is not None and 1 <= len(g.indices) <= 2
                                     and np.shape(g.matrix) == (1 << len(g.indices),) * 2 for g in sources):
            packed = _PACKED_SEQUENCES.get(id(sources))
            if packed is None or packed[0] is not sources or packed[1] != indices or packed[6] != len(sources):
                arity = [len(g.indices) for g in sources]
                legs = []
                for g in sources:
                    pos = [indices.index(int(q)) for q in g.indices]
                    legs += pos + [0] * (2 - len(pos))
                mats = np.concatenate([np.ascontiguousarray(g.matrix, dtype=np.complex128).reshape(-1) for g in sources])
                packed = (sources, list(indices), _ints(indices), _ints(arity), _ints(legs), mats, len(sources))
                if len(_PACKED_SEQUENCES) >= 512:
                    _PACKED_SEQUENCES.clear()
                _PACKED_SEQUENCES[id(sources)] = packed       # a fused block is applied many times: pack its list once
            handled = C.c_int(0)
            _lib.call("qsv_apply_sequence", self._h, len(indices), packed[2], len(sources), packed[3], packed[4],
                      _ptr(packed[5]), C.byref(handled))
            if handled.value:
                return self
        return self.apply_matrix(matrix, indices)

    def apply_diagonal(self, diagonal, indices) -> "DeviceState":
        indices = [int(i) for i in indices]
        d = _cbuf(diagonal, 1 << len(indices))
        if len(indices) == 1:
            _lib.call("qsv_apply_diag_1q", self._h, indices[0], _ptr(d))
        elif len(indices) == 2:
            _lib.call("qsv_apply_diag_2q", self._h, indices[0], indices[1], _ptr(d))
        else:
            _lib.call("qsv_apply_kq", self._h, len(indices), _ints(indices), _ptr(_cbuf(np.diag(d))))
        return self

    def apply_cx(self, control: int, target: int) -> "DeviceState":
        _lib.call("qsv_apply_cx", self._h, int(control), int(target))
        return self

    def apply_swap(self, q0: int, q1: int) -> "DeviceState":
        _lib.call("qsv_apply_swap", self._h, int(q0), int(q1))
        return self

    def apply_controlled(self, matrix, controls, target: int) -> "DeviceState":
        m = _cbuf(matrix, 4)
        controls = list(controls)
        _lib.call("qsv_apply_controlled_1q", self._h, len(controls), _ints(controls), int(target), _ptr(m))
        return self

    def apply_mcphase(self, qubits, phase: complex) -> "DeviceState":
        qubits = list(qubits)
        phase = complex(phase)
        _lib.call("qsv_apply_mcphase", self._h, len(qubits), _ints(qubits), phase.real, phase.imag)
        return self

    def apply_scale(self, factor: complex) -> "DeviceState":
        factor = complex(factor)
        _lib.call("qsv_scale", self._h, factor.real, factor.imag)
        return self

    def permute(self, new_ordering) -> "DeviceState":
        _lib.call("qsv_permute", self._h, _ints(new_ordering))
        return self

    # ---- measurement / insertion --------------------------------------------------------------
    def measure(self, index: int, eig0, eig1, forced: int | None = None, u01: float = 0.0):
        """Collapse qubit ``index`` (register shrinks by one).  Returns ``(outcome, p0, p1)``."""
        e0, e1 = _cbuf(eig0, 2), _cbuf(eig1, 2)
        out, p0, p1 = C.c_int(), C.c_double(), C.c_double()
        _lib.call("qsv_measure", self._h, int(index), _ptr(e0), _ptr(e1), -1 if forced is None else int(forced),
                  float(u01), C.byref(out), C.byref(p0), C.byref(p1))
        return out.value, p0.value, p1.value

    def measure_probs(self, index: int, eig0, eig1) -> tuple[float, float]:
        e0, e1 = _cbuf(eig0, 2), _cbuf(eig1, 2)
        p0, p1 = C.c_double(), C.c_double()
        _lib.call("qsv_measure_probs", self._h, int(index), _ptr(e0), _ptr(e1), C.byref(p0), C.byref(p1))
        return p0.value, p1.value

    def collapse(self, index: int, eig, scale: float) -> "DeviceState":
        """Project qubit ``index`` on ``eig`` (unconjugated) and multiply by ``scale``; register shrinks."""
        _lib.call("qsv_collapse", self._h, int(index), _ptr(_cbuf(eig, 2)), float(scale))
        return self

    def insert(self, index: int, amplitudes) -> "DeviceState":
        a = _cbuf(amplitudes, 2)
        _lib.call("qsv_insert", self._h, int(index), _ptr(a))
        return self

    # ---- read-out -----------------------------------------------------------------------------
    def norm2(self) -> float:
        v = C.c_double()
        _lib.call("qsv_norm2", self._h, C.byref(v))
        return v.value

    def probabilities(self, indices) -> np.ndarray:
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        out = np.empty(idx.size, dtype=np.float64)
        _lib.call("qsv_probabilities", self._h, idx.ctypes.data_as(C.POINTER(C.c_uint64)), idx.size,
                  out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def expect_pauli(self, paulis: str, qubits) -> complex:
        """``<psi| P |psi>`` for the Pauli string ``paulis`` (letters I/X/Y/Z) on ``qubits``, computed on the device."""
        qubits = [int(q) for q in qubits]
        if len(paulis) != len(qubits):
            raise ValueError("one Pauli letter per qubit")
        re, im = C.c_double(), C.c_double()
        _lib.call("qsv_expect_pauli", self._h, len(qubits), _ints(qubits), paulis.encode(), C.byref(re), C.byref(im))
        return complex(re.value, im.value)

    def sample(self, shots: int, rng=None) -> np.ndarray:
        """``shots`` computational-basis outcomes drawn from |amplitude|^2 (inverse-CDF on the device; the uniforms
        come from ``rng``, a ``numpy.random.Generator``, default the global ``np.random`` state).  No collapse."""
        u = np.ascontiguousarray(rng.random(shots) if rng is not None else np.random.random_sample(shots))
        out = np.empty(shots, dtype=np.uint64)
        _lib.call("qsv_sample", self._h, int(shots), _ptr(u), out.ctypes.data_as(C.c_void_p))
        return out

    def inner(self, other: "DeviceState") -> complex:
        re, im = C.c_double(), C.c_double()
        _lib.call("qsv_inner", self._h, other._h, C.byref(re), C.byref(im))
        return complex(re.value, im.value)

    def reduced_density(self, qubits) -> np.ndarray:
        """Reduced density matrix of ``qubits`` (at most six; everything else traced out) as a host
        ``(2^k, 2^k)`` array, ``qubits[0]`` the most significant bit of both indices.  One read pass on the device."""
        qubits = [int(q) for q in qubits]
        if not 1 <= len(qubits) <= 6:
            raise ValueError("keep between 1 and 6 qubits")
        dim = 1 << len(qubits)
        out = np.empty((dim, dim), dtype=np.complex128)
        _lib.call("qsv_reduced_density", self._h, len(qubits), _ints(qubits), _ptr(out))
        return out

    def expect_density(self, rho: "DensityState") -> complex:
        """``<self| rho |self>`` for a density matrix held on the device (the ket / matrix branch of ``npq.fidelity``)."""
        re, im = C.c_double(), C.c_double()
        _lib.call("qsv_expect_density", self._h, rho._h, C.byref(re), C.byref(im))
        return complex(re.value, im.value)

    # ---- timing (HIP events on the register's stream) -----------------------------------------
    def timer_start(self) -> None:
        _lib.call("qsv_timer_start", self._h)

    def timer_stop(self) -> float:
        ms = C.c_float()
        _lib.call("qsv_timer_stop", self._h, C.byref(ms))
        return ms.value


    def last_kernel(self) -> str:
        """Name of the gate kernel the most recent apply launched, as rocprofv3 spells it."""
        buf = C.create_string_buffer(128)
        _lib.call("qsv_last_kernel", self._h, buf, 128)
        return buf.value.decode()

    def event_record(self, slot: int) -> None:
        """Non-blocking HIP event mark number ``slot`` on the register's stream."""
        _lib.call("qsv_event_record", self._h, int(slot))

    def event_elapsed_ms(self, slot_a: int, slot_b: int) -> float:
        ms = C.c_float()
        _lib.call("qsv_event_elapsed_ms", self._h, int(slot_a), int(slot_b), C.byref(ms))
        return ms.value


class DensityState(DeviceState):
    """An n-qubit density matrix in HBM: ``rho`` flattened row-major is a 2n-qubit register whose first n qubits index
    the row and whose last n the column -- the layout ``Gate.apply`` uses for ``U rho U^dagger`` (gates.py:51-52), kept
    on the device between gates.  ``ndim == 2`` like the ndarray it stands for."""

    ndim = 2

    @classmethod
    def from_numpy(cls, rho: np.ndarray, device: int = 0) -> "DensityState":
        rho = np.asarray(rho)
        if rho.ndim != 2 or rho.shape[0] != rho.shape[1]:
            raise ValueError("State has wrong dimensions.")
        flat = DeviceState.from_numpy(np.ascontiguousarray(rho).reshape(-1), device)
        out = cls(flat._h, device=flat.device)
        flat._h = None
        return out

    @classmethod
    def from_ket(cls, ket: np.ndarray, device: int = 0) -> "DensityState":
        ket = np.asarray(ket)
        return cls.from_numpy(np.multiply.outer(ket, ket.conj()), device)

    @property
    def num_qubits(self) -> int:
        return super().num_qubits // 2

    @property
    def shape(self) -> tuple[int, int]:
        dim = 1 << self.num_qubits
        return (dim, dim)

    def to_numpy(self) -> np.ndarray:
        return super().to_numpy().reshape(self.shape)

    def copy(self) -> "DensityState":
        twin = DeviceState.copy(self)
        out = DensityState(twin._h, device=twin.device)
        twin._h = None
        return out

    def apply_matrix(self, matrix: np.ndarray, indices) -> "DensityState":
        """``U rho U^dagger``: ``U`` on the row qubits, ``conj(U)`` on the column qubits."""
        n = self.num_qubits
        indices = [int(i) for i in indices]
        if any(not 0 <= q < n for q in indices):
            raise ValueError(f"qubit index out of range for a {n}-qubit register")
        DeviceState.apply_matrix(self, matrix, indices)
        DeviceState.apply_matrix(self, np.conjugate(matrix), [n + q for q in indices])
        return self

    def apply_sequence(self, indices, sources, matrix) -> "DensityState":
        return self.apply_matrix(matrix, indices)       # U rho U^dagger: both sides, as the dense block

    def purity(self) -> float:
        """``tr(rho rho)`` for a hermitian ``rho`` (``npq.purity``): the squared norm of the flattened register."""
        return self.norm2()

    def fidelity_with_ket(self, ket: DeviceState) -> float:
        return ket.expect_density(self).real


class QuditState:
    """``n_modes`` d-level modes, dense complex128, mode 0 slowest (cv_simulator-style mode indices)."""

    def __init__(self, handle: C.c_void_p, device: int = 0):
        self._h = handle
        self.device = device          # HIP device ordinal the register lives on

    @classmethod
    def zeros(cls, n_modes: int, d: int, device: int = 0) -> "QuditState":
        """All modes in level 0 (vacuum in a Fock basis)."""
        h = C.c_void_p()
        _lib.call("qsv_create_qudit", int(n_modes), int(d), int(device), C.byref(h))
        return cls(h, int(device))

    @classmethod
    def from_numpy(cls, tensor: np.ndarray, device: int = 0) -> "QuditState":
        tensor = np.asarray(tensor)
        d = tensor.shape[0]
        if any(s != d for s in tensor.shape):
            raise ValueError("all modes must share one local dimension")
        st = cls.zeros(tensor.ndim, d, device)
        buf = _cbuf(tensor)
        _lib.call("qsv_upload", st._h, _ptr(buf), 0, buf.size)
        return st

    def close(self) -> None:
        if self._h is not None and self._h.value:
            _lib.load().qsv_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def dims(self) -> tuple[int, int]:
        n, d = C.c_int(), C.c_int()
        _lib.call("qsv_qudit_shape", self._h, C.byref(n), C.byref(d))
        return n.value, d.value

    def to_numpy(self) -> np.ndarray:
        n, d = self.dims
        out = np.empty(d ** n, dtype=np.complex128)
        _lib.call("qsv_download", self._h, _ptr(out), 0, out.size)
        return out.reshape((d,) * n)

    def download(self, offset: int, count: int) -> np.ndarray:
        """``count`` consecutive amplitudes starting at flat index ``offset`` (mode 0 slowest)."""
        out = np.empty(count, dtype=np.complex128)
        _lib.call("qsv_download", self._h, _ptr(out), int(offset), int(count))
        return out

    def fill_random(self, seed: int) -> float:
        """Normalised pseudo-random amplitudes generated on the device (counter-based); returns the raw squared norm."""
        n2 = C.c_double()
        _lib.call("qsv_fill_random", self._h, int(seed), 0, C.byref(n2))
        _lib.call("qsv_scale", self._h, 1.0 / np.sqrt(n2.value), 0.0)
        return n2.value

    def upload(self, tensor: np.ndarray) -> None:
        """Overwrite the register with ``tensor`` (shape ``(d,) * n_modes`` or flat, mode 0 slowest)."""
        n, d = self.dims
        buf = _cbuf(tensor, d ** n)
        _lib.call("qsv_upload", self._h, _ptr(buf), 0, buf.size)

    def sync(self) -> None:
        _lib.call("qsv_sync", self._h)

    def apply_mode(self, matrix, mode: int) -> "QuditState":
        _, d = self.dims
        m = np.asarray(matrix)
        if m.ndim == 1:
            _lib.call("qsv_apply_mode1_diag", self._h, int(mode), _ptr(_cbuf(m, d)))
        else:
            _lib.call("qsv_apply_mode1", self._h, int(mode), _ptr(_cbuf(m, d * d)))
        return self

    def apply_two_mode(self, matrix, mode0: int, mode1: int) -> "QuditState":
        _, d = self.dims
        m = np.asarray(matrix)
        if m.ndim == 1 or m.shape == (d, d):   # diagonal given flat or as the (d, d) plane
            _lib.call("qsv_apply_mode2_diag", self._h, int(mode0), int(mode1), _ptr(_cbuf(m, d * d)))
        else:
            _lib.call("qsv_apply_mode2", self._h, int(mode0), int(mode1), _ptr(_cbuf(m, d ** 4)))
        return self

    def apply_two_mode_gather(self, cols, vals, mode0: int, mode1: int) -> "QuditState":
        """Sparse plane map: ``cols`` / ``vals`` of shape ``(d*d, nnz)`` (negative column = unused slot)."""
        _, d = self.dims
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        vals = _cbuf(vals)
        if cols.ndim != 2 or cols.shape[0] != d * d or vals.size != cols.size:
            raise ValueError("cols and vals must both have shape (d*d, nnz)")
        _lib.call("qsv_apply_mode2_gather", self._h, int(mode0), int(mode1), int(cols.shape[1]),
                  C.c_void_p(cols.ctypes.data), _ptr(vals))
        return self

    def apply_two_mode_blocks(self, blocks, mode0: int, mode1: int) -> "QuditState":
        """Block-diagonal plane operator: ``blocks`` = list of ``(plane_indices, matrix)`` with disjoint index sets
        (``j0 * d + j1`` in (mode0, mode1) order) and a dense ``s x s`` matrix each (s <= 32).  In place."""
        sizes = np.array([len(idx) for idx, _ in blocks], dtype=np.int32)
        indices = np.ascontiguousarray(np.concatenate([np.asarray(idx, dtype=np.int32) for idx, _ in blocks]))
        mats = np.ascontiguousarray(np.concatenate([_cbuf(m, len(idx) ** 2).reshape(-1) for idx, m in blocks]))
        _lib.call("qsv_apply_mode2_blocks", self._h, int(mode0), int(mode1), len(blocks),
                  sizes.ctypes.data_as(C.c_void_p), indices.ctypes.data_as(C.c_void_p), _ptr(mats))
        return self

    def marginal(self, mode: int) -> np.ndarray:
        """Sum of |amplitude|^2 over every other mode, per level of ``mode``."""
        _, d = self.dims
        out = np.empty(d, dtype=np.float64)
        _lib.call("qsv_mode_marginal", self._h, int(mode), C.c_void_p(out.ctypes.data))
        return out

    def project(self, mode: int, level: int, scale: float) -> "QuditState":
        """Keep level ``level`` of ``mode`` (times ``scale``) and remove the mode."""
        _lib.call("qsv_mode_project", self._h, int(mode), int(level), float(scale))
        return self

    def insert(self, mode: int, amplitudes) -> "QuditState":
        """New mode with the given ``d`` amplitudes at position ``mode`` (product with the rest)."""
        _, d = self.dims
        _lib.call("qsv_mode_insert", self._h, int(mode), _ptr(_cbuf(amplitudes, d)))
        return self

    def scale(self, factor: complex) -> "QuditState":
        factor = complex(factor)
        _lib.call("qsv_scale", self._h, factor.real, factor.imag)
        return self

    def copy(self) -> "QuditState":
        n, d = self.dims
        other = QuditState.zeros(n, d, self.device)
        _lib.call("qsv_copy", other._h, self._h)
        return other

    def norm2(self) -> float:
        v = C.c_double()
        _lib.call("qsv_norm2", self._h, C.byref(v))
        return v.value

    def timer_start(self) -> None:
        _lib.call("qsv_timer_start", self._h)

    def timer_stop(self) -> float:
        ms = C.c_float()
        _lib.call("qsv_timer_stop", self._h, C.byref(ms))
        return ms.value

    def set_option(self, option: int, value: int) -> None:
        _lib.call("qsv_set_option", self._h, int(option), int(value))

    def last_kernel(self) -> str:
        buf = C.create_string_buffer(128)
        _lib.call("qsv_last_kernel", self._h, buf, 128)
        return buf.value.decode()


def tensor_apply_axis(dev_in: int, dev_out: int, L: int, d_in: int, d_out: int, R: int, matrix, device: int = 0,
                      stream: int = 0) -> None:
    """``out[l, :, r] = M @ in[l, :, r]`` on raw device tensors (an MPS site: L = chi_l, R = chi_r).

    ``matrix`` is a host array (uploaded on every call) or, to keep an operator resident the way the reference's
    gates keep ``self.matrix``, the integer device address of a row-major complex128 ``(d_out, d_in)`` buffer
    (e.g. ``torch_tensor.data_ptr()``): then nothing is copied and the call is asynchronous on ``stream``."""
    if isinstance(matrix, (int, np.integer)):
        _lib.call("qsv_tensor_apply_axis_dev", int(device), C.c_void_p(stream), C.c_void_p(dev_in),
                  C.c_void_p(dev_out), int(L), int(d_in), int(d_out), int(R), C.c_void_p(int(matrix)))
        return
    m = _cbuf(matrix, d_in * d_out)
    _lib.call("qsv_tensor_apply_axis", int(device), C.c_void_p(stream), C.c_void_p(dev_in), C.c_void_p(dev_out),
              int(L), int(d_in), int(d_out), int(R), _ptr(m))
