"""Qubit gate classes: the operator API of the reference, backed by HIP kernels.

Same class names, constructor signatures, ``indices`` / ``matrix`` attributes and exceptions as
``simulators/dv_simulator/gates.py:7-194``.  The difference is what ``apply`` does: the reference expands the
gate to a dense 2^N x 2^N operator and multiplies (``gates.py:44-54`` -> ``numpy_quantum.py:243-247``); here
``apply`` hands the 2^k x 2^k matrix and the qubit indices to libqsv.so, which streams once over the register
in HBM.  ``apply`` accepts

* a NumPy ket / density matrix (reference behaviour: returns a NEW array, input untouched; costs one upload
  and one download -- use it for small registers and drop-in tests), or
* a :class:`~quantum_computations_amd.device.DeviceState` (updated IN PLACE and returned, so a circuit runs
  without the register ever leaving the GPU -- this is what ``Simulator.run`` uses).
"""
from __future__ import annotations

import numpy as np

import threading

from . import numpy_quantum as npq
from .states import State
from ..device import DeviceState

# ``Gate.apply(ndarray)`` -- the reference's literal calling convention -- needs a device register per call.  Small ones
# are kept between calls (one per size, <= 2^22 amplitudes = 64 MiB): allocating, zero-filling and freeing HBM costs more
# than the gate on a register of a few qubits (0.3 ms against 60 us per call at n <= 12).
_POOL: dict[int, DeviceState] = {}
_POOL_LOCK = threading.Lock()
_POOL_MAX_QUBITS = 22


def _borrow_register(ket: np.ndarray) -> DeviceState:
    size = ket.shape[0]
    if size == 0 or size & (size - 1):
        raise ValueError("Given array is not a qubit state nor operator")
    n = size.bit_length() - 1
    with _POOL_LOCK:
        dev = _POOL.pop(n, None)
    if dev is None:
        return DeviceState.from_numpy(ket)
    dev.upload(ket)
    return dev


def _return_register(dev: DeviceState, n: int) -> None:
    # only registers that still have the size they were borrowed with go back (matrix gates never resize)
    if n <= _POOL_MAX_QUBITS and DeviceState.num_qubits.fget(dev) == n:
        with _POOL_LOCK:
            if n not in _POOL:
                _POOL[n] = dev
                return
    dev.close()

REPR_DIGITS = 5


def is_device_register(state) -> bool:
    """A register that lives in HBM: a ``DeviceState`` (ket), a ``DensityState`` (density matrix: its ``apply_matrix``
    does ``U rho U^dagger``) or a ``distributed.ShardedState`` (same gate methods)."""
    return isinstance(state, DeviceState) or (not isinstance(state, np.ndarray) and hasattr(state, "apply_matrix"))


def _as_result_dtype(values: np.ndarray, *operands, padded: bool = True) -> np.ndarray:
    """Cast the complex128 device result to the dtype NumPy promotion gives the reference's ``@``.

    ``padded``: the reference's ``expand_gate`` pads with the float64 identity whenever the gate acts on fewer
    qubits than the register has (``numpy_quantum.py:245``), which promotes integer matrices to float64.
    """
    dtype = np.result_type(*operands, *([np.float64] if padded else []))
    if np.issubdtype(dtype, np.complexfloating):
        return values.astype(dtype, copy=False)
    return np.ascontiguousarray(values.real).astype(dtype, copy=False)


class Gate:
    """A 2^a x 2^k matrix acting on the qubits ``indices`` (first index = most significant leg)."""

    def __init__(self, indices: list[int], matrix: np.ndarray | None):
        _check_indices(indices)
        if matrix is not None:
            if matrix.ndim != 2:
                raise ValueError("Not a 2D array.")
            if not all(npq.is_power_of_two(size) for size in matrix.shape):
                raise ValueError("Given matrix is not a mapping between qubit spaces.")
            if matrix.shape[1] != 2 ** len(indices):
                raise ValueError("Dimensions of given matrix is not compatible with number of indices.")
        self.indices = indices
        self.matrix = matrix

    def __repr__(self):
        return type(self).__name__ + "_" + ",".join(str(i) for i in self.indices)

    def copy(self) -> "Gate":
        clone = type(self).__new__(type(self))
        clone.__dict__.update(self.__dict__)
        return clone

    def relabel(self, mapping: dict):
        """Rename qubits ``i -> mapping[i]`` in place (every index must be mapped)."""
        renamed = []
        for i in self.indices:
            if mapping.get(i) is None:
                raise ValueError(f"Index {i} does not map anywhere.")
            renamed.append(mapping[i])
        _check_indices(renamed)
        self.indices = renamed

    # ---- the hot path ---------------------------------------------------------------------------
    def apply(self, state):
        if self.matrix is None:
            raise ValueError(f"Matrix representation not given for {self}.")
        if self.matrix.shape[0] != self.matrix.shape[1]:
            raise ValueError("new_ordering must be a permutation of all qubits")  # as expand_gate would
        if is_device_register(state):
            sources = getattr(self, "sources", None)        # a fused block (fusion.py) knows the gates it was made of
            if sources and hasattr(state, "apply_sequence"):
                return state.apply_sequence(self.indices, sources, self.matrix)
            return state.apply_matrix(self.matrix, self.indices)
        state = np.asarray(state)
        if state.ndim == 1:
            dev = _borrow_register(state)
            n = DeviceState.num_qubits.fget(dev)
            try:
                dev.apply_matrix(self.matrix, self.indices)
                out = dev.to_numpy()
            except BaseException:
                dev.close()
                raise
            _return_register(dev, n)
            return _as_result_dtype(out, state, self.matrix, padded=len(self.indices) < npq.num_qubits(state))
        if state.ndim == 2:
            # U rho U^dagger: rho flattened row-major is a 2n-qubit ket; U acts on the row qubits and
            # conj(U) on the column qubits (gates.py:51-52).
            n = npq.num_qubits(state)
            dev = _borrow_register(np.ascontiguousarray(state).reshape(-1))
            try:
                dev.apply_matrix(self.matrix, self.indices)
                dev.apply_matrix(np.conjugate(self.matrix), [n + q for q in self.indices])
                out = dev.to_numpy().reshape(state.shape)
            except BaseException:
                dev.close()
                raise
            _return_register(dev, 2 * n)
            return _as_result_dtype(out, state, self.matrix, padded=len(self.indices) < n)
        raise ValueError("State has wrong dimensions.")


def _check_indices(indices) -> None:
    if len(set(indices)) != len(indices):
        raise ValueError("Indices must be distinct.")
    if min(indices) < 0:
        raise ValueError("Non-negative index")


class SingleQubitGate(Gate):
    def __init__(self, index: int, matrix):
        super().__init__([index], matrix)


class TwoQubitGate(Gate):
    def __init__(self, index1: int, index2: int, matrix):
        super().__init__([index1, index2], matrix)


def _fixed_1q(name: str, matrix_of, doc: str):
    def __init__(self, index):
        SingleQubitGate.__init__(self, index, matrix_of())
    return type(name, (SingleQubitGate,), {"__init__": __init__, "__doc__": doc, "__module__": __name__})


def _z_rotation(angle: float) -> np.ndarray:
    # the gate-class convention: exp(-i angle/2 Z) = diag(e^{-i angle/2}, e^{+i angle/2}) (gates.py:87-114),
    # NOT npq.P / npq.T, which differ by a global phase
    return npq.axis_rotation(angle, [0, 0, 1])


I = _fixed_1q("I", lambda: npq.IDTY, "Identity.")
X = _fixed_1q("X", lambda: npq.X, "Pauli X.")
Y = _fixed_1q("Y", lambda: npq.Y, "Pauli Y.")
Z = _fixed_1q("Z", lambda: npq.Z, "Pauli Z.")
H = _fixed_1q("H", lambda: npq.H, "Hadamard.")
P = _fixed_1q("P", lambda: _z_rotation(np.pi / 2), "Phase gate as RZ(pi/2).")
Pdg = _fixed_1q("Pdg", lambda: _z_rotation(-np.pi / 2), "Inverse phase gate, RZ(-pi/2).")
T = _fixed_1q("T", lambda: _z_rotation(np.pi / 4), "T gate as RZ(pi/4).")
Tdg = _fixed_1q("Tdg", lambda: _z_rotation(-np.pi / 4), "Inverse T gate, RZ(-pi/4).")


class RZ(SingleQubitGate):
    def __init__(self, index, angle: float):
        super().__init__(index, _z_rotation(angle))
        self.angle = angle

    def __repr__(self):
        return super().__repr__() + f"({round(self.angle, REPR_DIGITS)})"


def _fixed_2q(name: str, matrix_of, doc: str, **extra):
    def __init__(self, first, second):
        TwoQubitGate.__init__(self, first, second, matrix_of())
    return type(name, (TwoQubitGate,), {"__init__": __init__, "__doc__": doc, "__module__": __name__, **extra})


CX = _fixed_2q("CX", lambda: npq.CX, "Controlled X: ``CX(control, target)``; kept as a pure amplitude move on the GPU.",
               control=property(lambda self: self.indices[0]), target=property(lambda self: self.indices[1]))
CZ = _fixed_2q("CZ", lambda: npq.CZ, "Controlled Z (symmetric): touches a quarter of the register.")
SWAP = _fixed_2q("SWAP", lambda: npq.SWAP, "Exchange of two qubits.")


class Insert(SingleQubitGate):
    """Grow the register by one qubit in ``state`` at position ``index`` (``gates.py:136-153``)."""

    def __init__(self, index: int, state: State):
        super().__init__(index, state.get().reshape((1, 2)))
        self.state = state

    def __repr__(self):
        return super().__repr__() + f"({self.state})"

    def apply(self, state):
        new_qubit = self.matrix[0, :]
        index = self.indices[0]
        if is_device_register(state):
            return state.insert(index, new_qubit)
        state = np.asarray(state)
        dev = DeviceState.from_numpy(state)
        dev.insert(index, new_qubit)
        out = dev.to_numpy()
        dev.close()
        return _as_result_dtype(out, state, new_qubit, padded=False)


class M(SingleQubitGate):
    """Projective measurement along the Bloch direction (theta, phi); REMOVES the qubit (``gates.py:155-186``).

    Randomness is drawn exactly where the reference draws it -- the global ``np.random.choice`` with the two
    branch probabilities -- so a seeded script sees the same outcomes; ``result=`` forces the outcome.  The
    projector uses the eigenvectors unconjugated, as the reference does.
    """

    def __init__(self, index: int, theta: float, phi: float, *, result: int = None):
        super().__init__(index, None)
        if result is not None and result not in [0, 1]:
            raise ValueError(f"Measurement results must be from 0 or 1 but {result} was given.")
        self.theta = theta
        self.phi = phi
        self.result = result

    def eigenvectors(self) -> tuple[np.ndarray, np.ndarray]:
        rotation = npq.axis_rotation(self.phi, [0, 0, 1]) @ npq.axis_rotation(self.theta, [0, 1, 0])
        return rotation @ npq.ZERO, rotation @ npq.ONE

    def _measure(self, dev: DeviceState) -> int:
        eigs = self.eigenvectors()
        probs = dev.measure_probs(self.indices[0], *eigs)
        s = np.random.choice([0, 1], p=list(probs)) if self.result is None else self.result
        if self.result is None and hasattr(dev, "agree_on_outcome"):
            s = dev.agree_on_outcome(int(s))      # sharded registers: one generator per rank, one outcome for all
        dev.collapse(self.indices[0], eigs[s], 1.0 / np.sqrt(probs[s]))
        return int(s)

    def apply(self, state):
        if is_device_register(state):
            return state, self._measure(state)
        dev = DeviceState.from_numpy(np.asarray(state))
        s = self._measure(dev)
        out = dev.to_numpy()
        dev.close()
        return out, s


class MZ(M):
    def __init__(self, index, *, result=None):
        super().__init__(index, 0.0, 0.0, result=result)


class MX(M):
    def __init__(self, index, *, result=None):
        super().__init__(index, np.pi / 2, 0.0, result=result)
