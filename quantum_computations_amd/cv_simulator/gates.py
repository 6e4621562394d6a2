"""CV gates on a position-grid register -- the operator API of ``simulators/cv_simulator/gates.py:13-269`` on HBM.

Each gate builds its small operator on the host (``d x d`` matrix, ``d``-vector of phases, or a 4-entries-per-row
resampling table for the (q1, q2) plane) and hands it to libqsv.so, which applies it along the addressed mode axes of
the dense register in one pass: ``qsv_apply_mode1`` (F, X, rotation), ``qsv_apply_mode1_diag`` (Z, P),
``qsv_apply_mode2_diag`` (CZ), ``qsv_apply_mode2_gather`` (BS, CX, SWAP), ``qsv_mode_marginal`` + ``qsv_mode_project``
(homodyne measurements), ``qsv_mode_insert`` (Insert).  Sign conventions (index order, ``dagger``) follow the reference.
"""
from __future__ import annotations

import logging

import numpy as np
from numpy.random import Generator as RNG

from .gate_abc import *  # noqa: F401,F403  (Gate, SingleModeGate, TwoModeGate, Measurement, MeasurementResult, REPR_DIGITS)
from .gate_abc import Gate, Measurement, MeasurementResult, REPR_DIGITS, SingleModeGate, TwoModeGate
from .mps import MPS
from .states import State
from .utils import fourier_matrix, plane_resample_table, rotation_matrix, sinc_matrix

logger = logging.getLogger(__name__)


def _sign(dagger: bool) -> int:
    return -1 if dagger else 1


class Insert(SingleModeGate):
    """Insert a mode in state ``state`` (a :class:`State` or a wavefunction sampled on the domain) at ``index``."""

    def __init__(self, index: int, state, *, gkp_epsilon: float = None, **kwargs):
        if kwargs.pop("dagger", None):
            logger.info(type(self).__name__ + "gates ignores adjoint/dagger.")
        super().__init__(index, arg=state, **kwargs)
        self.gkp_epsilon = gkp_epsilon

    def apply(self, mps: MPS, *, rng: RNG = None, **_):
        if self.index < 0 or self.index > len(mps):
            raise IndexError(f"Cannot insert mode at index {self.index} for MPS of length {len(mps)}")
        wave = self.arg.eval(mps.domain, self.gkp_epsilon) if hasattr(self.arg, "eval") else np.asarray(self.arg)
        mps.reg.insert(self.index, wave)


class SWAP(TwoModeGate):
    """Swap two neighbouring modes."""

    def apply(self, mps: MPS, *, rng: RNG = None, **_):
        d = len(mps.domain)
        i, j = np.meshgrid(np.arange(d), np.arange(d), indexing="ij")
        cols = (j * d + i).reshape(-1, 1)                      # new[i, j] = old[j, i]
        mps.reg.apply_two_mode_gather(cols, np.ones_like(cols, dtype=np.complex128), self.left_index, self.right_index)


class BS(TwoModeGate):
    """Beam splitter: rotation of the (q1, q2) plane by ``angle``, bilinear resampling (gates.py:58-84)."""

    def __init__(self, index1, index2, angle: float = np.pi / 4, **kwargs):
        super().__init__(index1, index2, arg=angle, **kwargs)

    def __repr__(self):
        return type(self).__name__ + f"({round(self.arg / np.pi, REPR_DIGITS)} * π)" + f"_{self.index1},{self.index2}"

    def apply(self, mps: MPS, *, rng: RNG = None, **_):
        angle = self.arg * (-1 if self.index1 > self.index2 else 1) * _sign(self.dagger)
        qs = mps.domain
        x, y = np.meshgrid(qs, qs, indexing="ij")
        c, s = np.cos(angle), np.sin(angle)
        cols, vals = plane_resample_table(qs, c * x + s * y, -s * x + c * y)
        mps.reg.apply_two_mode_gather(cols, vals, self.left_index, self.right_index)


class Mq(Measurement):
    """Homodyne measurement of q: samples (or takes the forced) grid point and removes the mode (gates.py:87-117)."""

    def apply(self, mps: MPS, rng: RNG = None, **_):
        qs, dq = mps.domain, mps.diff
        distribution = mps.marginal(self.index) * dq           # probability per grid cell
        if self.result is None:
            s_index = rng.choice(range(len(qs)), p=distribution / np.sum(distribution))
        else:
            s_index = int(np.argmin(np.abs(qs - self.result)))
        s = qs[s_index]
        p = distribution[s_index] / dq
        if len(mps) == 1:          # the reference returns the bare value here and leaves the state alone
            return s
        mps.reg.project(self.index, int(s_index), 1.0 / np.sqrt(p))
        return MeasurementResult(s, p)


class Mp(Mq):
    """Homodyne measurement of p: inverse Fourier gate, then Mq."""

    def apply(self, mps: MPS, **kwargs):
        mps.reg.apply_mode(fourier_matrix(mps.domain, inv=True), self.index)
        return super().apply(mps, **kwargs)


class Homodyne(Mq):
    """Homodyne measurement of the quadrature rotated by ``angle``."""

    def __init__(self, index, angle: float, result: float = None, **kwargs):
        super().__init__(index, result, arg=angle, **kwargs)

    def __repr__(self):
        return (type(self).__name__ + f"({round(self.arg / np.pi, REPR_DIGITS)} * π)" + f"_{self.index}"
                + (f" = {round(self.result, REPR_DIGITS)}" if self.result else ""))

    def apply(self, mps: MPS, **kwargs):
        if np.isclose(np.sin(self.arg), 0):
            logger.info("\tsin(angle) ≈ 0 detected: Using Mq gate instead.")
            result = super().apply(mps, **kwargs)
            result.result *= np.round(np.cos(self.arg))        # a -q measurement flips the sign
            return result
        mps.reg.apply_mode(rotation_matrix(mps.domain, -self.arg), self.index)
        return super().apply(mps, **kwargs)


class CZ(TwoModeGate):
    """Controlled p-displacement exp(i s q1 q2): a phase on every point of the (q1, q2) plane."""

    def __init__(self, index1, index2, s: float = 1.0, **kwargs):
        super().__init__(index1, index2, arg=s, **kwargs)

    def apply(self, mps: MPS, *, rng: RNG = None, **_):
        qs = mps.domain
        plane = np.exp(_sign(self.dagger) * 1j * self.arg * np.outer(qs, qs))
        mps.reg.apply_two_mode(plane, self.left_index, self.right_index)


class CX(TwoModeGate):
    """Controlled q-displacement: shear of the (q1, q2) plane, bilinear resampling (gates.py:166-192)."""

    def __init__(self, control, target, s: float = 1.0, **kwargs):
        super().__init__(control, target, arg=s, **kwargs)

    def __repr__(self):
        return Gate.__repr__(self) + f"_{self.index1},{self.index2}"

    def apply(self, mps: MPS, *, rng: RNG = None, **_):
        qs = mps.domain
        x, y = np.meshgrid(qs, qs, indexing="ij")
        if self.index1 < self.index2:          # the left mode is the control
            x, y = x, y - x * _sign(self.dagger)
        else:
            x, y = x - y * _sign(self.dagger), y
        cols, vals = plane_resample_table(qs, x, y)
        mps.reg.apply_two_mode_gather(cols, vals, self.left_index, self.right_index)


class F(SingleModeGate):
    """Fourier gate (FFT + sinc resampling on the grid, utils.py:41-84)."""

    def apply(self, mps: MPS, **_):
        mps.reg.apply_mode(fourier_matrix(mps.domain, inv=self.dagger), self.index)


class X(SingleModeGate):
    """q-displacement by ``s`` (band-limited shift: a sinc matrix)."""

    def __init__(self, index, s: float = 1.0, **kwargs):
        super().__init__(index, arg=s, **kwargs)

    def apply(self, mps: MPS, **_):
        qs = mps.domain
        mps.reg.apply_mode(sinc_matrix(qs, qs - _sign(self.dagger) * self.arg), self.index)


class Z(SingleModeGate):
    """p-displacement by ``s``: the phase exp(i s q)."""

    def __init__(self, index, s: float = 1.0, **kwargs):
        super().__init__(index, arg=s, **kwargs)

    def apply(self, mps: MPS, **_):
        mps.reg.apply_mode(np.exp(_sign(self.dagger) * 1j * self.arg * mps.domain), self.index)


class D(SingleModeGate):
    """Displacement by ``s = [s_q, s_p]``: X then Z."""

    def __init__(self, index, s, **kwargs):
        if len(s) != 2:
            raise ValueError("s must have exactly 2 elements.")
        super().__init__(index, arg=s, **kwargs)

    def apply(self, mps: MPS, **kwargs):
        X(self.index, _sign(self.dagger) * self.arg[0]).apply(mps, **kwargs)
        Z(self.index, _sign(self.dagger) * self.arg[1]).apply(mps, **kwargs)


class P(SingleModeGate):
    """Quadratic phase gate exp(i s q^2 / 2)."""

    def __init__(self, index, s: float = 1.0, **kwargs):
        super().__init__(index, arg=s, **kwargs)

    def apply(self, mps: MPS, **_):
        mps.reg.apply_mode(np.exp(_sign(self.dagger) * 0.5j * self.arg * mps.domain ** 2), self.index)


class Phase(SingleModeGate):
    """Phase-space rotation by ``angle``.  The reference declares this gate but raises ``NotImplementedError``
    (gates.py:261-269); here it applies the fractional-Fourier kernel the reference already uses for ``Homodyne``
    (``utils.rotation``), so the operator itself is pinned by the rotation fixtures."""

    def __init__(self, index, angle: float, **kwargs):
        super().__init__(index, arg=angle, **kwargs)

    def apply(self, mps: MPS, **_):
        mps.reg.apply_mode(rotation_matrix(mps.domain, _sign(self.dagger) * self.arg), self.index)


class S(SingleModeGate):
    """Single-mode squeezing by ``r`` along the direction ``angle``.  Declared but unimplemented in the reference
    (gates.py:249-258): **no reference counterpart, parity unpinned**.  Convention: ``S(r, 0)`` maps the vacuum to
    ``squeezed_vac(q, r)`` of ``states.py`` (width e^r in q): psi'(q) = e^{-r/2} psi(q e^{-r}), band-limited resampling."""

    def __init__(self, index, r: float, angle: float = 0.0, **kwargs):
        super().__init__(index, arg=r, **kwargs)
        self.angle = angle

    def matrix(self, qs: np.ndarray) -> np.ndarray:
        r = _sign(self.dagger) * self.arg
        m = np.exp(-r / 2) * sinc_matrix(qs, qs * np.exp(-r))
        if not np.isclose(np.sin(self.angle), 0):
            m = rotation_matrix(qs, self.angle) @ m @ rotation_matrix(qs, -self.angle)
        return m

    def apply(self, mps: MPS, **_):
        mps.reg.apply_mode(self.matrix(mps.domain), self.index)
