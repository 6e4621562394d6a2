"""TEST INFRASTRUCTURE ONLY — O(2^n) NumPy restatement of the reference's qubit gate-application path.

The reference applies a gate by building the dense 2^N x 2^N operator and multiplying
(``simulators/dv_simulator/gates.py:44-54`` -> ``numpy_quantum.py:243-247``).  The functions here compute the
same results by contracting the small gate matrix against the addressed tensor legs of the state, which is
what the HIP kernels do.  Bit order follows the reference: qubit ``q`` of ``n`` is tensor axis ``q`` of the
C-ordered ``(2,)*n`` view, i.e. bit ``n-1-q`` of the flat index (qubit 0 = most significant bit).

Pinned by ``tests/golden/*.npz`` (generated from the reference by ``tests/golden/generate_golden.py``).
"""
from __future__ import annotations

import numpy as np


def num_qubits(state: np.ndarray) -> int:
    """Register size of a ket / density matrix (``numpy_quantum.py:254-258``)."""
    n = int(state.shape[0]).bit_length() - 1
    if state.shape[0] != 1 << n:
        raise ValueError("Given array is not a qubit state nor operator")
    return n


def _contract_legs(tensor: np.ndarray, matrix: np.ndarray, axes: list[int]) -> np.ndarray:
    """Apply a 2^k x 2^k ``matrix`` to the ``axes`` of a ``(2,)*m (+ trailing)`` tensor.

    Leg ``j`` of the matrix (most significant first, the ``kron(gate, I, ...)`` order of
    ``numpy_quantum.py:245``) acts on ``axes[j]`` -- the ``targets`` order of ``expand_gate``
    (``numpy_quantum.py:246``).
    """
    k = len(axes)
    g = np.asarray(matrix).reshape((2,) * (2 * k))
    out = np.tensordot(g, tensor, axes=(list(range(k, 2 * k)), list(axes)))
    # tensordot puts the k output legs first; move leg j back to position axes[j].
    return np.moveaxis(out, list(range(k)), list(axes))


def apply_gate(state: np.ndarray, matrix: np.ndarray, indices: list[int]) -> np.ndarray:
    """``Gate.apply`` (``gates.py:44-54``): ``U_full @ ket`` or ``U_full @ rho @ U_full^dagger``.

    Returns a new array; dtype follows NumPy promotion exactly as the reference's ``@`` does.
    """
    if matrix is None:
        raise ValueError("Matrix representation not given.")
    n = num_qubits(state)
    k = len(indices)
    if len(set(indices)) != k:
        raise ValueError("Indices must be distinct.")
    if min(indices) < 0 or max(indices) >= n:
        raise ValueError("index out of range for this register")
    if np.asarray(matrix).shape != (1 << k, 1 << k):
        raise ValueError("Dimensions of given matrix is not compatible with number of indices.")
    if k < n:
        # expand_gate pads with the float64 identity (numpy_quantum.py:245), which promotes integer gates
        matrix = np.asarray(matrix).astype(np.result_type(matrix, np.float64), copy=False)
    if state.ndim == 1:
        out = _contract_legs(state.reshape((2,) * n), matrix, list(indices))
        return np.ascontiguousarray(out).reshape(-1)
    if state.ndim == 2:
        rho = state.reshape((2,) * (2 * n))
        rho = _contract_legs(rho, matrix, list(indices))                       # U on the row legs
        rho = _contract_legs(rho, np.conjugate(matrix), [n + q for q in indices])  # conj(U) on the column legs
        return np.ascontiguousarray(rho).reshape(1 << n, 1 << n)
    raise ValueError("State has wrong dimensions.")


def permute_qubits(state: np.ndarray, new_ordering: list[int]) -> np.ndarray:
    """``permute_tensor_product`` for kets (``numpy_quantum.py:212-240``).

    The reference transposes with the *inverse* of ``new_ordering`` (``:234``): the qubit that was at
    position ``j`` ends up at position ``new_ordering[j]``.
    """
    n = num_qubits(state)
    if sorted(new_ordering) != list(range(n)):
        raise ValueError("new_ordering must be a permutation of all qubits")
    inv = [0] * n
    for j, p in enumerate(new_ordering):
        inv[p] = j
    return np.ascontiguousarray(state.reshape((2,) * n).transpose(inv)).reshape(-1)


def insert_qubit(state: np.ndarray, index: int, amplitudes: np.ndarray) -> np.ndarray:
    """``Insert.apply`` (``gates.py:145-153``): ``kron(state, new)`` then move the new qubit to ``index``."""
    n = num_qubits(state)
    if index < 0 or index > n:
        raise ValueError("new_ordering must be a permutation of all qubits")
    amplitudes = np.asarray(amplitudes).reshape(2)
    grown = np.multiply.outer(state.reshape((2,) * n), amplitudes)   # new qubit is the last axis
    grown = np.moveaxis(grown, n, index)
    return np.ascontiguousarray(grown).reshape(-1)


def reduced_density(state: np.ndarray, kept: list[int]) -> np.ndarray:
    """Reduced density matrix of the qubits ``kept`` of a ket: ``ket2dm`` (``numpy_quantum.py:110-113``) with every
    other qubit summed out, ``kept[0]`` the most significant bit of both indices -- computed as X X^H for the
    (2^k x 2^(n-k)) matrix X without forming the 2^n x 2^n matrix."""
    n = num_qubits(state)
    rest = [q for q in range(n) if q not in kept]
    x = np.transpose(state.reshape((2,) * n), list(kept) + rest).reshape(1 << len(kept), -1)
    return x @ x.conj().T


def axis_rotation(theta: float, axis) -> np.ndarray:
    """``numpy_quantum.py:104-105``: ``cos(theta/2) I - i sin(theta/2) (a.sigma)``."""
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    y = np.array([[0, -1j], [1j, 0]], dtype=complex)
    z = np.array([[1, 0], [0, -1]], dtype=complex)
    return np.identity(2) * np.cos(theta / 2) - 1j * (axis[0] * x + axis[1] * y + axis[2] * z) * np.sin(theta / 2)


def measurement_vectors(theta: float, phi: float) -> tuple[np.ndarray, np.ndarray]:
    """Eigenvectors used by ``M.apply`` (``gates.py:169-171``): columns of ``RZ(phi) RY(theta)``."""
    rot = axis_rotation(phi, [0, 0, 1]) @ axis_rotation(theta, [0, 1, 0])
    return rot[:, 0].copy(), rot[:, 1].copy()


def measure_branches(state: np.ndarray, index: int, theta: float, phi: float):
    """Both un-normalised branches of ``M.apply`` and their norms (``gates.py:173-181``).

    The reference puts the 1-D eigenvector itself in the Kronecker slot, so the projector row is ``eig``
    *unconjugated* (SURVEY.md appendix); we reproduce that.
    """
    n = num_qubits(state)
    psi = np.moveaxis(state.reshape((2,) * n), index, 0).reshape(2, -1)
    eig0, eig1 = measurement_vectors(theta, phi)
    res0 = eig0[0] * psi[0] + eig0[1] * psi[1]
    res1 = eig1[0] * psi[0] + eig1[1] * psi[1]
    return (res0, float(np.linalg.norm(res0))), (res1, float(np.linalg.norm(res1)))


def measure(state: np.ndarray, index: int, theta: float, phi: float, result: int):
    """``M.apply`` with a forced ``result`` (``gates.py:183-186``): normalised (n-1)-qubit ket and the bit."""
    branches = measure_branches(state, index, theta, phi)
    res, nrm = branches[result]
    return res / nrm, result


def run_circuit(ops: list[dict], state: np.ndarray) -> tuple[np.ndarray, list[int]]:
    """``Simulator.run`` (``simulator.py:36-53``) over a neutral op list (see ``workloads.py``).

    Each op is ``{"name", "indices", "matrix"}`` for gates, ``{"name": "M", "indices", "theta", "phi",
    "result"}`` for forced measurements, ``{"name": "Insert", "indices", "vector"}`` for insertions, with an
    optional ``"control": {"pos": [...], "neg": [...]}`` restating ``ClassicalControl.eval`` (``:16-17``).
    """
    results: list[int] = []
    for op in ops:
        ctl = op.get("control")
        if ctl is not None:
            fire = all(results[i] for i in ctl.get("pos", [])) and all(not results[i] for i in ctl.get("neg", []))
            if not fire:
                continue
        if op["name"] == "M":
            state, bit = measure(state, op["indices"][0], op["theta"], op["phi"], op["result"])
            results.append(bit)
        elif op["name"] == "Insert":
            state = insert_qubit(state, op["indices"][0], op["vector"])
        else:
            state = apply_gate(state, op["matrix"], op["indices"])
    return state, results
