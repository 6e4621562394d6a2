"""Conversions and logical read-out of GKP registers (``simulators/gkp_simulator/utils.py:1-106``).

``full_logical_density_mps`` is the expensive one: ``4^N`` Pauli strings, each an environment contraction through the
chain.  Here the four read-out operators are uploaded once, each site is dressed with them by one GEMM each
(``qsv_tensor_apply_axis_dev``), and a depth-first walk over the strings shares the environments of common prefixes, so
the device does ``(4^(N+1) - 4) / 3`` small GEMM pairs instead of ``N 4^N`` einsums.
"""
from __future__ import annotations

from functools import reduce
from itertools import product

import numpy as np

from ..cv_simulator.mps import MPS, tensor_svd  # noqa: F401  (re-exported, as upstream)
from ..cv_simulator.utils import *  # noqa: F401,F403
from ..dv_simulator import numpy_quantum as npq

PI = np.pi
SQPI = np.sqrt(np.pi)

_LOGICAL_PAULIS = (np.array([[1, 0], [0, 1]]), np.array([[0, 1], [1, 0]]), np.array([[0, -1j], [1j, 0]]),
                   np.array([[1, 0], [0, -1]]))


def eps2db(epsilon: float) -> float:
    """GKP damping parameter -> squeezing in dB."""
    return -10.0 * np.log10(2.0 * np.tanh(epsilon / 2.0))


def db2eps(db_squeezing: float) -> float:
    """Squeezing in dB -> GKP damping parameter (inverse of :func:`eps2db`)."""
    return 2.0 * np.arctanh(np.float_power(10.0, -db_squeezing / 10.0) / 2.0)


def decomp_result(s: float) -> tuple[int, float]:
    """``s = (n + r) sqrt(pi)`` with ``n`` the nearest integer."""
    in_units = s / SQPI
    n = np.round(in_units).astype(int)
    return n, in_units - n


def format_result(s: float, dec: int = 4) -> str:
    n, r = decomp_result(s * 2 ** 0.5)
    return f"({n}{r:+.{dec}f})√π"


def cv2dv_information(s: float) -> bool:
    """Logical bit carried by a homodyne value: parity of the nearest multiple of sqrt(pi)."""
    return np.round(s / SQPI) % 2 == 1


def syndrome_matrix(syndromes: list[tuple[int, int]]) -> np.ndarray:
    """Tensor product of ``Z^z X^x`` over the qubits, one ``(x, z)`` pair each."""
    factors = []
    for x, z in syndromes:
        m = npq.IDTY
        if x:
            m = npq.X @ m
        if z:
            m = npq.Z @ m
        factors.append(m)
    return npq.tensor(*factors)


_READOUT_OPERATORS: dict = {}      # grid -> [I, X, Y, Z] read-out operators, see pauli_readout_operators


def pauli_readout_operators(qs: np.ndarray) -> list[np.ndarray]:
    """Grid operators ``[I, X, Y, Z]`` whose expectation values are the logical Bloch components: truncated Fourier
    series of the sqrt(pi)-periodic sign functions -- odd multiples ``m`` of sqrt(pi), coefficients ``±2/(m pi)``,
    displacements realised by sinc interpolation and the momentum-like ones by cosines (utils.py:48-71; appendix D
    of Shaw et al., arXiv:2403.02396).  The quirks are kept: the spacing here is ``(q_max - q_min) / len(qs)``."""
    # the operators depend on the grid alone (0.2 s of sinc evaluations at d = 1000): one set per grid is kept, read-only,
    # which also lets a register keep ONE device copy of each (SiteRegister._keep goes by array identity)
    key = (qs.shape[0], qs.dtype.str, qs.tobytes()) if isinstance(qs, np.ndarray) and qs.ndim == 1 and qs.shape[0] <= 4096 else None
    if key is not None and key in _READOUT_OPERATORS:
        return _READOUT_OPERATORS[key]
    d = len(qs)
    dq = (qs[-1] - qs[0]) / d
    offsets = qs[:, None] - qs[None, :]
    shift, phase = np.zeros((d, d)), np.zeros((d, d))
    for n, m in enumerate(range(1, int((qs[-1] - qs[0]) / SQPI) + 1, 2)):
        weight = (-1) ** (n % 2) * 2 / (m * PI)
        shift += weight * (np.sinc((offsets - m * SQPI) / dq) + np.sinc((offsets + m * SQPI) / dq))
        phase += weight * np.diag(2 * np.cos(SQPI * m * qs))
    operators = [np.identity(d), shift, 1j * shift @ phase, phase]
    if key is not None:
        for op in operators:
            op.setflags(write=False)
        if len(_READOUT_OPERATORS) >= 2:
            _READOUT_OPERATORS.clear()
        _READOUT_OPERATORS[key] = operators
    return operators


def full_logical_density_mps(mps: MPS, normalised: bool = False) -> np.ndarray:
    """Logical ``2^N x 2^N`` density matrix of an N-mode GKP register: ``sum_P <P_readout> P_logical / 2^N`` with
    the grid measure ``(dq/2)^N`` of the reference (utils.py:82-96)."""
    qs, n = mps.domain, len(mps)
    dq = (qs[-1] - qs[0]) / len(qs)
    operators = pauli_readout_operators(qs)
    if mps.layout == "sites":
        coefficients = mps.reg.operator_string_coefficients(operators)
    else:  # dense register: contract on the host (small registers only)
        psi = mps.contract().reshape((len(qs),) * n)
        coefficients = np.zeros((4,) * n, dtype=np.complex128)
        for index in product(range(4), repeat=n):
            dressed = psi
            for axis, i in enumerate(index):
                dressed = np.moveaxis(np.tensordot(operators[i], dressed, [1, axis]), 0, axis)
            coefficients[index] = np.vdot(psi, dressed)
    rho = np.zeros((2 ** n, 2 ** n), dtype=complex)
    for index in product(range(4), repeat=n):
        rho += coefficients[index] * (dq / 2) ** n * reduce(np.kron, [_LOGICAL_PAULIS[i] for i in index], 1)
    if normalised:
        rho /= np.trace(rho)
    return rho


def full_logical_density(qs: np.ndarray, state: np.ndarray, normalised: bool = False, device: int = 0) -> np.ndarray:
    """Logical density matrix of a dense wavefunction ``state[q_0, ..., q_{N-1}]``.  (Upstream this helper stops after
    splitting the tensor into sites and returns nothing, utils.py:98-106; here it finishes the job.)"""
    rest = np.reshape(state, (1, *np.shape(state), 1))
    sites = []
    while rest.ndim > 3:
        head, rest = tensor_svd(rest, (0, 1), tuple(range(2, rest.ndim)), device=device)
        sites.append(head)
    sites.append(rest)
    return full_logical_density_mps(MPS(qs, sites, device=device, layout="sites"), normalised)
