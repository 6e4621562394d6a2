"""Single-mode wavefunctions on a position grid, for ``Insert`` and initial registers.

Host-side state preparation (O(d) per mode, runs once; SURVEY.md row 9 keeps it on the host).  Same ``State`` members
and ``eval(qs, gkp_epsilon)`` contract as ``simulators/cv_simulator/states.py:9-67``; the GKP comb is evaluated with a
plain NumPy theta-function sum instead of mpmath's ``jtheta`` (the reference's ``np.vectorize``d call is its slow spot).
"""
from __future__ import annotations

from enum import Enum, auto

import numpy as np

PI = np.pi
SQPI = np.sqrt(np.pi)


def theta3(u: np.ndarray, nome: float) -> np.ndarray:
    """Jacobi theta_3(u, q) = 1 + 2 sum_{n>=1} q^{n^2} cos(2 n u) for a real nome 0 <= q < 1."""
    if not 0 <= nome < 1:
        raise ValueError("nome must be in [0, 1)")
    out = np.ones_like(u, dtype=np.float64)
    n = 1
    while True:
        term = nome ** (n * n)
        if term < 1e-18:
            break
        out += 2 * term * np.cos(2 * n * u)
        n += 1
    return out


def _comb(qs: np.ndarray, epsilon: float, spacing: float, shift: float) -> np.ndarray:
    """Gaussian comb with Gaussian envelope ("symmetric" finite-energy form, states.py:117-119): teeth every
    ``spacing``, offset by ``shift`` teeth, envelope exp(-tanh(eps) q^2 / 2)."""
    t = np.tanh(epsilon)
    z = -qs / (spacing * np.cosh(epsilon)) + shift
    return np.exp(-t * qs ** 2 / 2) * theta3(PI * z, np.exp(-PI * t * (2 * PI / spacing ** 2)))


def gkp_sym(qs: np.ndarray, epsilon: float, coefficients=(1, 0)) -> np.ndarray:
    """Un-normalised square-lattice GKP state c_0 |0_L> + c_1 |1_L> (teeth at 2 sqrt(pi) (k + mu/2))."""
    return sum(c * _comb(qs, epsilon, 2 * SQPI, mu / 2) for mu, c in enumerate(coefficients))


def squeezed_coherent(q, alpha: complex, r: float, theta: float):
    alpha = complex(alpha)
    delta = np.sqrt((np.cos(theta) * np.exp(r)) ** 2 + (np.sin(theta) / np.exp(r)) ** 2)
    chirp = 1 - 1j * np.sinh(2 * r) * np.sin(2 * theta)
    return (PI * delta ** 2) ** -0.25 * np.exp(-0.5 * ((q - alpha.real) / delta) ** 2 * chirp + 1j * alpha.imag * q)


def vacuum(q):
    return squeezed_coherent(q, 0, 0, 0)


def coherent(q, alpha):
    return squeezed_coherent(q, alpha, 0, 0)


def squeezed_vac(q, r):
    return squeezed_coherent(q, 0, r, 0)


def fock_state(q, n: int):
    from scipy.special import factorial, hermite
    return hermite(n)(q) * np.exp(-q ** 2 / 2) * (2 ** n * factorial(n) * SQPI) ** -0.5


_GKP_COEFFICIENTS = {
    "GKP_ZERO": (1, 0), "GKP_ONE": (0, 1), "GKP_PLUS": (1, 1), "GKP_MINUS": (1, -1),
    "GKP_T": (1, np.exp(0.25j * PI)), "GKP_TDG": (1, np.exp(-0.25j * PI)),
    "GKP_H": (np.cos(PI / 8), np.sin(PI / 8)),
}


class State(Enum):
    GKP_ZERO = auto()
    GKP_ONE = auto()
    GKP_PLUS = auto()
    GKP_MINUS = auto()
    GKP_T = auto()
    GKP_TDG = auto()
    GKP_H = auto()
    VACUUM = auto()
    QUNAUGHT = auto()

    def __repr__(self):
        return self.name

    def __str__(self):
        return self.name

    def eval(self, qs: np.ndarray, gkp_epsilon: float = None) -> np.ndarray:
        """Wavefunction sampled on ``qs``, normalised with the grid measure (sum |psi|^2 dq = 1)."""
        if not isinstance(qs, np.ndarray) or qs.ndim != 1:
            raise TypeError("qs must be a 1D numpy array.")
        if not np.allclose(np.diff(qs, 2), 0, atol=np.finfo(qs.dtype).eps ** 0.5):
            raise ValueError("qs is not an arithmetic progression.")
        if gkp_epsilon is not None and gkp_epsilon <= 0:
            raise ValueError("epsilon must be a positive real number")
        if self is State.VACUUM:
            return vacuum(qs)
        if gkp_epsilon is None:
            raise ValueError("Evaluating gkp and qunaught states require a gkp_epsilon.")
        if self is State.QUNAUGHT:
            psi = _comb(qs, gkp_epsilon, np.sqrt(2 * PI), 0.0)
        else:
            psi = gkp_sym(qs, gkp_epsilon, _GKP_COEFFICIENTS[self.name])
        dq = abs(qs[-1] - qs[0]) / (len(qs) - 1)
        return psi / np.sqrt(np.real(np.sum(psi * np.conjugate(psi))) * dq)
