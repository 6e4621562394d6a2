"""Host-side constants and small linear-algebra helpers of the qubit simulator.

Mirror of the reference module ``simulators/dv_simulator/numpy_quantum.py`` (same public names, argument
meaning and error behaviour) so code written against ``npq.*`` keeps working.  Everything here is O(small)
host work: gate matrices, single-qubit states, fidelities of already-downloaded kets.  The O(2^N) work the
reference does here -- ``expand_gate`` / ``permute_tensor_product`` / ``tensor`` feeding ``Gate.apply``
(``numpy_quantum.py:169-170,212-247``) -- is what the HIP kernels replace; the helpers of those names kept
below exist for small-N API compatibility only and are never on the device path.
"""
from __future__ import annotations

from functools import reduce

import numpy as np

# -- single-qubit kets and gate matrices (numpy_quantum.py:5-25), same dtypes as the reference -------------
_R2 = np.sqrt(2)
ZERO = np.array([1, 0])
ONE = np.array([0, 1])
PLUS = np.array([1, 1]) / _R2
MINUS = np.array([1, -1]) / _R2
IPLUS = np.array([1, 1j]) / _R2
IMINUS = np.array([1, -1j]) / _R2

IDTY = np.identity(2)
X = np.array([[0, 1], [1, 0]])
Y = np.array([[0, -1j], [1j, 0]])
Z = np.array([[1, 0], [0, -1]])
PAULIS = [X, Y, Z]
H = np.array([[1, 1], [1, -1]]) / _R2

CZ = np.diag([1.0, 1.0, 1.0, -1.0])
CX = np.identity(4)[[0, 1, 3, 2], :]          # control = first index, target = second (gates.py:116-126)
SWAP = np.identity(4)[[0, 2, 1, 3], :]

P = np.diag([1.0, 1.0j])
T = np.diag([1.0, np.exp(0.25j * np.pi)])


class PauliError(ValueError):
    pass


_PAULI_NAMES = {"i": 0, "x": 1, "y": 2, "z": 3}
_PAULI_AXES = {(1, 0, 0): 1, (0, 1, 0): 2, (0, 0, 1): 3, (-1, 0, 0): -1, (0, -1, 0): -2, (0, 0, -1): -3}


def get_pauli_number(pauli_identifier) -> int:
    """Signed Pauli number in {-3..3} from a letter, an int or a unit axis (``numpy_quantum.py:32-50``)."""
    p = pauli_identifier
    if isinstance(p, str):
        sign = -1 if p.startswith("-") else 1
        body = p.lstrip("-").lower()
        if body in _PAULI_NAMES and not (sign < 0 and body == "i") and len(p) <= 2:
            return sign * _PAULI_NAMES[body]
    elif isinstance(p, (int, np.integer)) and not isinstance(p, bool) and -3 <= int(p) <= 3:
        return int(p)
    elif isinstance(p, (list, tuple)) and tuple(p) in _PAULI_AXES:
        return _PAULI_AXES[tuple(p)]
    raise PauliError(f'"{pauli_identifier}" could not be interpreted as a Pauli operator')


def get_pauli_identifier(pauli_identifier) -> str:
    return ["-Z", "-Y", "-X", "I", "X", "Y", "Z"][get_pauli_number(pauli_identifier) + 3]


def is_pauli(case) -> bool:
    try:
        get_pauli_number(case)
    except PauliError:
        return False
    return True


def get_pauli_operator(pauli_identifier) -> np.ndarray:
    return PAULIS[get_pauli_number(pauli_identifier) - 1]


def get_pauli_states(pauli_identifier):
    return [[PLUS, MINUS], [IPLUS, IMINUS], [ZERO, ONE]][get_pauli_number(pauli_identifier) - 1]


def get_pauli_state(pauli_identifier, state_index: int) -> np.ndarray:
    return get_pauli_states(pauli_identifier)[state_index]


def basis_state(identifier, N: int = None) -> np.ndarray:
    """Computational basis ket from an int, a bit string or a bit sequence.

    The reference's list/tuple branch is broken (``numpy_quantum.py:79`` drops ``N``); here it works and
    yields what the string branch yields for the same bits.
    """
    if isinstance(identifier, (list, tuple)):
        identifier = "".join(str(b) for b in identifier)
    if isinstance(identifier, str):
        return basis_state(int(identifier, 2), len(identifier))
    if isinstance(identifier, (int, np.integer)):
        state = np.zeros(2 ** N)
        state[identifier] = 1
        return state
    raise NotImplementedError(f"Could not generate basis state from identifier of type {type(identifier)}")


def qubit_from_polar(theta: float, phi: float):
    return np.cos(theta / 2) * ZERO + np.exp(1j * phi) * np.sin(theta / 2) * ONE


def qubit_from_axis(axis) -> np.ndarray:
    theta = np.arccos(axis[-1] / np.sqrt(sum(a ** 2 for a in axis)))
    return qubit_from_polar(theta, np.arctan2(axis[1], axis[0]))


def phase_gate(theta: float) -> np.ndarray:
    return np.array([[1, 0], [0, np.exp(1j * theta)]])


def axis_rotation(theta: float, axis) -> np.ndarray:
    """``exp(-i theta/2 a.sigma)`` (``numpy_quantum.py:104-105``)."""
    generator = axis[0] * X + axis[1] * Y + axis[2] * Z
    return IDTY * np.cos(theta / 2) - 1j * generator * np.sin(theta / 2)


def euler_rotation(theta1, theta2, theta3) -> np.ndarray:
    return axis_rotation(theta3, [1, 0, 0]) @ axis_rotation(theta2, [0, 0, 1]) @ axis_rotation(theta1, [1, 0, 0])


def dagger(array: np.ndarray) -> np.ndarray:
    return np.conjugate(array.T)


def ket2dm(ket: np.ndarray) -> np.ndarray:
    if ket.ndim != 1:
        raise TypeError("state is not a ket")
    return np.outer(ket, np.conjugate(ket))


def is_hermitian(oper: np.ndarray) -> bool:
    return np.allclose(dagger(oper), oper)


def norm(ket: np.ndarray) -> float:
    return np.linalg.norm(ket)


def normalise(state: np.ndarray) -> np.ndarray:
    if state.ndim == 1:
        return state / np.linalg.norm(state)
    if state.ndim == 2:
        return state / np.trace(state)
    raise ValueError("State not ket nor density matrix.")


def dm2ket(dm: np.ndarray, strict: bool = True) -> np.ndarray:
    if not is_hermitian(dm):
        raise TypeError("input is not a density matrix")
    eigvals, eigvecs = np.linalg.eigh(dm)
    if strict and not np.allclose(eigvals[:-1], 0):
        raise TypeError("density matrix does not represent a pure state")
    return normalise(eigvecs[:, -1])


def compare_kets(a: np.ndarray, b: np.ndarray) -> bool:
    return np.allclose(ket2dm(normalise(a)), ket2dm(normalise(b)))


def fidelity(a: np.ndarray, b: np.ndarray) -> float:
    """Fidelity between kets and/or (hermitian) density matrices (``numpy_quantum.py:148-161``)."""
    a_ket, b_ket = a.ndim == 1, b.ndim == 1
    if a_ket and b_ket:
        return np.abs(a.conj() @ b).real ** 2
    if a_ket:
        return (a.conj() @ b @ a).real
    if b_ket:
        return (b.conj() @ a @ b).real
    eigvals = np.clip(np.linalg.eigvals(a @ b).real, 0.0, None)
    return np.sum(np.sqrt(eigvals)) ** 2


def purity(rho: np.ndarray) -> float:
    return np.trace(rho @ rho).real


def tensor(*arrays) -> np.ndarray:
    """Kronecker product of all arguments (``numpy_quantum.py:169-170``); small host arrays only."""
    return reduce(np.kron, arrays, 1)


def is_power_of_two(n: int) -> bool:
    return n != 0 and (n & (n - 1)) == 0


def is_qubit_operator(oper: np.ndarray) -> bool:
    return oper.ndim == 2 and oper.shape[0] == oper.shape[1] and is_power_of_two(oper.shape[0])


def is_qubit_state(state: np.ndarray) -> bool:
    return state.ndim == 1 and is_power_of_two(len(state))


def expect(oper: np.ndarray, state: np.ndarray):
    if not is_qubit_operator(oper) or not is_qubit_state(state) or oper.shape[0] != state.shape[0]:
        raise TypeError("incompatible operator and state vector")
    return np.conjugate(state) @ oper @ state


def expecth(oper: np.ndarray, state: np.ndarray):
    return expect(oper, state).real


def rand_ket(d=2) -> np.ndarray:
    return normalise(np.random.rand(d) + 1j * np.random.rand(d))


def num_qubits(arr) -> int:
    size = arr if isinstance(arr, int) else arr.shape[0]
    return int(np.log2(size))


def permute_tensor_product(array: np.ndarray, new_ordering) -> np.ndarray:
    """Qubit-axis permutation of a small host ket / operator (``numpy_quantum.py:227-240``).

    The factor at position ``j`` moves to position ``new_ordering[j]``.  Host arrays only; the device path
    uses ``qsv_permute`` / the gate kernels instead of materialising operators.
    """
    size = array.shape[0]
    if not is_power_of_two(size):
        raise ValueError("Given array is not a qubit state nor operator")
    n = num_qubits(array)
    if set(new_ordering) != set(range(n)):
        raise ValueError("new_ordering must be a permutation of all qubits")
    source = np.argsort(np.asarray(new_ordering))          # inverse permutation

    def rows(a):
        return a.reshape((2,) * n + (-1,)).transpose(*source, n).reshape(size, -1)

    if array.ndim == 2:
        return rows(rows(array).T).T
    return rows(array).reshape(-1)


def expand_gate(gate: np.ndarray, N: int, targets) -> np.ndarray:
    """Dense 2^N x 2^N operator of ``gate`` on ``targets`` (``numpy_quantum.py:243-247``).

    Kept for small-N host use (building fixtures, inspecting operators).  ``Gate.apply`` does NOT call it:
    the device kernels contract the 2^k x 2^k matrix directly.
    """
    targets = list(targets)
    rest = [q for q in range(N) if q not in targets]
    full = tensor(gate, *([IDTY] * len(rest)))
    return permute_tensor_product(full, targets + rest)


def add_control(gate: np.ndarray) -> np.ndarray:
    dim = gate.shape[0]
    out = np.zeros((2 * dim, 2 * dim), dtype=np.result_type(gate, float))
    out[:dim, :dim] = np.identity(dim)
    out[dim:, dim:] = gate
    return out
