"""Host-side constants and small linear algebra of the qubit simulator (the ``npq`` namespace).

Keeps the public names, argument meaning and error classes of ``simulators/dv_simulator/numpy_quantum.py`` so code
written against ``npq.*`` runs unchanged.  Everything here is O(small) host work: gate matrices, single-qubit states,
fidelities of kets that have already been downloaded.  The O(2^N) work the reference does in this module for
``Gate.apply`` -- ``expand_gate`` / ``permute_tensor_product`` / ``tensor`` (``numpy_quantum.py:169-170,212-247``) -- is
what the HIP kernels replace; the functions of those names below exist for small-N API compatibility only and are never
on the device path.
"""
from __future__ import annotations

import functools

import numpy as np

_INV_SQRT2 = 1.0 / np.sqrt(2)


def _ket(*amplitudes):
    return np.array(amplitudes)


# ---- single-qubit kets (dtypes as in the reference: integer computational states, float / complex superpositions)
ZERO, ONE = _ket(1, 0), _ket(0, 1)
PLUS, MINUS = _ket(1, 1) / np.sqrt(2), _ket(1, -1) / np.sqrt(2)
IPLUS, IMINUS = _ket(1, 1j) / np.sqrt(2), _ket(1, -1j) / np.sqrt(2)

# ---- gate matrices ----------------------------------------------------------------------------------------------
IDTY = np.eye(2)
X = np.array([[0, 1], [1, 0]])
Y = np.array([[0, -1j], [1j, 0]])
Z = np.array([[1, 0], [0, -1]])
PAULIS = [X, Y, Z]
H = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
CZ = np.diag([1.0, 1.0, 1.0, -1.0])
CX = np.eye(4)[[0, 1, 3, 2]]            # first index controls, second is the target (gates.py:116-126)
SWAP = np.eye(4)[[0, 2, 1, 3]]
P = np.diag([1.0, 1.0j])
T = np.diag([1.0, np.exp(0.25j * np.pi)])


# ---- Pauli bookkeeping --------------------------------------------------------------------------------------------
class PauliError(ValueError):
    """The identifier is not one of I, +-X, +-Y, +-Z (letter, signed number or unit axis)."""


_LETTERS = "ixyz"
_AXES = {(1, 0, 0): 1, (0, 1, 0): 2, (0, 0, 1): 3}


def get_pauli_number(pauli_identifier) -> int:
    """0 for I, +-1 / +-2 / +-3 for +-X / +-Y / +-Z."""
    ident = pauli_identifier
    if isinstance(ident, str) and 1 <= len(ident) <= 2:
        negative, letter = ident[:-1] == "-", ident[-1].lower()
        if letter in _LETTERS and ident[:-1] in ("", "-") and not (negative and letter == "i"):
            return (-1 if negative else 1) * _LETTERS.index(letter)
    elif isinstance(ident, (int, np.integer)) and not isinstance(ident, bool) and abs(int(ident)) <= 3:
        return int(ident)
    elif isinstance(ident, (list, tuple)) and len(ident) == 3:
        flipped = tuple(-c for c in ident)
        if tuple(ident) in _AXES:
            return _AXES[tuple(ident)]
        if flipped in _AXES:
            return -_AXES[flipped]
    raise PauliError(f'"{pauli_identifier}" could not be interpreted as a Pauli operator')


def get_pauli_identifier(pauli_identifier) -> str:
    number = get_pauli_number(pauli_identifier)
    return ("-" if number < 0 else "") + _LETTERS[abs(number)].upper()


def is_pauli(case) -> bool:
    try:
        get_pauli_number(case)
    except PauliError:
        return False
    return True


def get_pauli_operator(pauli_identifier) -> np.ndarray:
    return PAULIS[get_pauli_number(pauli_identifier) - 1]


def get_pauli_states(pauli_identifier):
    eigenbases = ([PLUS, MINUS], [IPLUS, IMINUS], [ZERO, ONE])
    return eigenbases[get_pauli_number(pauli_identifier) - 1]


def get_pauli_state(pauli_identifier, state_index: int) -> np.ndarray:
    return get_pauli_states(pauli_identifier)[state_index]


# ---- states -------------------------------------------------------------------------------------------------------
def basis_state(identifier, N: int = None) -> np.ndarray:
    """Computational basis ket from an index, a bit string or a bit sequence.

    (The reference's list / tuple branch forgets to pass ``N`` on, ``numpy_quantum.py:79``; here a bit sequence gives
    what the equivalent bit string gives.)
    """
    if isinstance(identifier, (list, tuple)):
        identifier = "".join(str(bit) for bit in identifier)
    if isinstance(identifier, str):
        identifier, N = int(identifier, 2), len(identifier)
    if not isinstance(identifier, (int, np.integer)):
        raise NotImplementedError(f"Could not generate basis state from identifier of type {type(identifier)}")
    ket = np.zeros(1 << N)
    ket[identifier] = 1
    return ket


def qubit_from_polar(theta: float, phi: float):
    return np.cos(theta / 2) * ZERO + np.exp(1j * phi) * np.sin(theta / 2) * ONE


def qubit_from_axis(axis) -> np.ndarray:
    length = np.sqrt(sum(component ** 2 for component in axis))
    return qubit_from_polar(np.arccos(axis[-1] / length), np.arctan2(axis[1], axis[0]))


def rand_ket(d=2) -> np.ndarray:
    return normalise(np.random.rand(d) + 1j * np.random.rand(d))


# ---- rotations ----------------------------------------------------------------------------------------------------
def phase_gate(theta: float) -> np.ndarray:
    return np.array([[1, 0], [0, np.exp(1j * theta)]])


def axis_rotation(theta: float, axis) -> np.ndarray:
    """``exp(-i theta/2 (a . sigma))`` for a unit axis ``a``."""
    a_dot_sigma = axis[0] * X + axis[1] * Y + axis[2] * Z
    return IDTY * np.cos(theta / 2) - 1j * a_dot_sigma * np.sin(theta / 2)


def euler_rotation(theta1, theta2, theta3) -> np.ndarray:
    x_axis, z_axis = [1, 0, 0], [0, 0, 1]
    return axis_rotation(theta3, x_axis) @ axis_rotation(theta2, z_axis) @ axis_rotation(theta1, x_axis)


# ---- kets, density matrices, overlaps -----------------------------------------------------------------------------
def dagger(array: np.ndarray) -> np.ndarray:
    return array.conj().T


def ket2dm(ket: np.ndarray) -> np.ndarray:
    if ket.ndim != 1:
        raise TypeError("state is not a ket")
    return np.multiply.outer(ket, ket.conj())


def is_hermitian(oper: np.ndarray) -> bool:
    return np.allclose(oper, dagger(oper))


def norm(ket: np.ndarray) -> float:
    return np.linalg.norm(ket)


def normalise(state: np.ndarray) -> np.ndarray:
    if state.ndim not in (1, 2):
        raise ValueError("State not ket nor density matrix.")
    return state / (np.linalg.norm(state) if state.ndim == 1 else np.trace(state))


def dm2ket(dm: np.ndarray, strict: bool = True) -> np.ndarray:
    """Dominant eigenvector of a density matrix; ``strict`` demands that it is the only one with weight."""
    if not is_hermitian(dm):
        raise TypeError("input is not a density matrix")
    weights, vectors = np.linalg.eigh(dm)
    if strict and not np.allclose(weights[:-1], 0):
        raise TypeError("density matrix does not represent a pure state")
    return normalise(vectors[:, -1])


def compare_kets(a: np.ndarray, b: np.ndarray) -> bool:
    return np.allclose(ket2dm(normalise(a)), ket2dm(normalise(b)))


def fidelity(a, b) -> float:
    """Fidelity between any mix of kets and (hermitian) density matrices (numpy_quantum.py:148-161).  Host arrays
    as in the reference; registers in HBM (``DeviceState`` kets, ``DensityState`` matrices) are reduced on the device
    -- ket/ket by one pass over both, ket/matrix by one pass over the matrix -- and never downloaded.  Two device
    density matrices need the spectrum of ``a @ b`` and go through the host (they are 4^n numbers: small n only)."""
    from ..device import DensityState, DeviceState
    if isinstance(a, DeviceState) or isinstance(b, DeviceState):
        if not (isinstance(a, DeviceState) and isinstance(b, DeviceState)):
            def lift(x, like):
                if isinstance(x, DeviceState):
                    return x
                x = np.asarray(x)
                return (DeviceState if x.ndim == 1 else DensityState).from_numpy(x, like.device)
            a, b = lift(a, b if isinstance(b, DeviceState) else a), lift(b, a if isinstance(a, DeviceState) else b)
        kinds = (a.ndim, b.ndim)
        if kinds == (1, 1):
            return abs(a.inner(b)) ** 2
        if kinds == (1, 2):
            return a.expect_density(b).real
        if kinds == (2, 1):
            return b.expect_density(a).real
        a, b = a.to_numpy(), b.to_numpy()
    kinds = (a.ndim, b.ndim)
    if kinds == (1, 1):
        return np.abs(np.vdot(a, b)).real ** 2
    if kinds == (1, 2):
        return np.vdot(a, b @ a).real
    if kinds == (2, 1):
        return np.vdot(b, a @ b).real
    spectrum = np.linalg.eigvals(a @ b).real.clip(min=0.0)        # (tr sqrt(a b))^2
    return np.sqrt(spectrum).sum() ** 2


def purity(rho) -> float:
    """``tr(rho rho)`` of a hermitian density matrix (numpy_quantum.py:164-166); a ``DensityState`` is reduced on
    the device."""
    if hasattr(rho, "purity"):
        return rho.purity()
    return np.trace(rho @ rho).real


def expect(oper: np.ndarray, state: np.ndarray):
    if not (is_qubit_operator(oper) and is_qubit_state(state) and oper.shape[0] == state.shape[0]):
        raise TypeError("incompatible operator and state vector")
    return np.vdot(state, oper @ state)


def expecth(oper: np.ndarray, state: np.ndarray):
    return expect(oper, state).real


# ---- sizes ----------------------------------------------------------------------------------------------------------
def is_power_of_two(n: int) -> bool:
    return n > 0 and n & (n - 1) == 0


def is_qubit_operator(oper: np.ndarray) -> bool:
    return oper.ndim == 2 and oper.shape[0] == oper.shape[1] and is_power_of_two(oper.shape[0])


def is_qubit_state(state: np.ndarray) -> bool:
    return state.ndim == 1 and is_power_of_two(len(state))


def num_qubits(arr) -> int:
    return int(np.log2(arr if isinstance(arr, int) else arr.shape[0]))


# ---- small-N tensor-product helpers (host arrays only; Gate.apply does NOT go through these) ------------------------
def tensor(*arrays) -> np.ndarray:
    """Kronecker product of all arguments, left to right."""
    return functools.reduce(np.kron, arrays, 1)


def permute_tensor_product(array: np.ndarray, new_ordering) -> np.ndarray:
    """Re-order the qubit factors of a ket or operator: the factor at position ``j`` moves to ``new_ordering[j]``."""
    dim = array.shape[0]
    if not is_power_of_two(dim):
        raise ValueError("Given array is not a qubit state nor operator")
    n = num_qubits(array)
    if sorted(new_ordering) != list(range(n)):
        raise ValueError("new_ordering must be a permutation of all qubits")
    came_from = list(np.argsort(new_ordering))

    def shuffle_rows(values):
        return values.reshape((2,) * n + (-1,)).transpose(came_from + [n]).reshape(dim, -1)

    if array.ndim == 1:
        return shuffle_rows(array).ravel()
    return shuffle_rows(shuffle_rows(array).T).T


def expand_gate(gate: np.ndarray, N: int, targets) -> np.ndarray:
    """Dense 2^N x 2^N operator of ``gate`` acting on ``targets`` (identity elsewhere); small N only."""
    targets = list(targets)
    spectators = [q for q in range(N) if q not in targets]
    padded = tensor(gate, *[IDTY for _ in spectators])
    return permute_tensor_product(padded, targets + spectators)


def add_control(gate: np.ndarray) -> np.ndarray:
    """``|0><0| x I + |1><1| x gate``."""
    dim = gate.shape[0]
    controlled = np.eye(2 * dim, dtype=np.result_type(gate, float))
    controlled[dim:, dim:] = gate
    return controlled
