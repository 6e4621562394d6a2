"""CV gates on a position-grid register: the gate set of ``simulators/cv_simulator/gates.py:13-269`` on HBM.

Every gate reduces to one small host-built object handed to libqsv.so, which applies it along the addressed mode axes
of the dense register in a single pass:

=====================  =======================================  ===========================================
gate                   host-built operator                      kernel entry point
=====================  =======================================  ===========================================
F, X, Phase, S         ``d x d`` matrix (``operator(domain)``)   ``qsv_apply_mode1``
Z, P                   ``d`` phases                             ``qsv_apply_mode1_diag``
CZ                     ``(d, d)`` plane of phases               ``qsv_apply_mode2_diag``
BS, CX, SWAP           4- (1-) entry-per-point resampling table  ``qsv_apply_mode2_gather``
Mq, Mp, Homodyne       --                                       ``qsv_mode_marginal`` + ``qsv_mode_project``
Insert                 wavefunction on the grid                 ``qsv_mode_insert``
=====================  =======================================  ===========================================

Constructor signatures, ``repr`` strings, sign conventions (index order, ``dagger``) and the quirks of the reference
(a measurement of the last remaining mode returns the bare value and leaves the register alone) are kept.
"""
from __future__ import annotations

import logging
from collections import OrderedDict

import numpy as np

from .gate_abc import *  # noqa: F401,F403
from .gate_abc import Gate, Measurement, MeasurementResult, REPR_DIGITS, SingleModeGate, TwoModeGate
from .mps import MPS
from .states import State  # noqa: F401  (re-exported, as upstream)
from .utils import fourier_matrix, plane_resample_table, rotation_matrix, sinc_matrix

logger = logging.getLogger("simulators." + __name__.split(".", 1)[1])


def _truncation(gate, mps, rng=None) -> dict:
    """Keyword arguments for the two-site split: the gate's ``svd_options`` and the simulator's generator on a site
    register (the reference calls ``tensor_svd(..., **self.svd_options, rng_seed=rng)``), nothing on the dense one."""
    return dict(gate.svd_options, rng_seed=rng) if mps.layout == "sites" else {}


_OPERATORS: "OrderedDict[tuple, object]" = OrderedDict()      # host operators shared by equal gates (LRU)
_OPERATORS_KEPT = 16


def _cached(gate, name: str, domain: np.ndarray, build):
    """Host operators depend only on the gate's class, parameters and the grid: build once and hand every equal gate
    the same array object (which the registers keep resident on the device), so that a circuit of many ``F`` /
    ``X(sqrt(pi))`` gates pays for the ``d x d`` matrix once."""
    key = (type(gate).__name__, name, repr(gate.arg), bool(gate.dagger), getattr(gate, "angle", None),
           getattr(gate, "index1", 0) < getattr(gate, "index2", 1), float(domain[0]), float(domain[-1]), len(domain))
    if key in _OPERATORS:
        _OPERATORS.move_to_end(key)
    else:
        _OPERATORS[key] = build()
        while len(_OPERATORS) > _OPERATORS_KEPT:
            _OPERATORS.popitem(last=False)
    return _OPERATORS[key]


def _pi_fraction(angle: float) -> str:
    return f"({round(angle / np.pi, REPR_DIGITS)} * π)"


# ---- single-mode gates: subclasses provide the operator, the base class ships it ---------------------------------
class _AxisGate(SingleModeGate):
    """Single-mode gate defined by ``operator(domain)``: a ``(d, d)`` matrix or a length-``d`` diagonal."""

    def __init__(self, index, s: float = 1.0, **kwargs):
        super().__init__(index, arg=s, **kwargs)

    @property
    def _signed(self) -> float:
        return -self.arg if self.dagger else self.arg

    def operator(self, domain: np.ndarray) -> np.ndarray:
        raise NotImplementedError

    def apply(self, mps: MPS, **_):
        mps.reg.apply_mode(_cached(self, "op", mps.domain, lambda: self.operator(mps.domain)), self.index)


class F(SingleModeGate):
    """Fourier gate: FFT on the grid followed by sinc resampling onto the same grid (``utils.fourier``)."""

    def apply(self, mps: MPS, **_):
        mps.reg.apply_mode(_cached(self, "op", mps.domain, lambda: fourier_matrix(mps.domain, inv=self.dagger)),
                           self.index)


class X(_AxisGate):
    """Displacement in q by ``s``: a band-limited shift, i.e. a sinc matrix."""

    def operator(self, domain):
        return sinc_matrix(domain, domain - self._signed)


class Z(_AxisGate):
    """Displacement in p by ``s``: the phase ``exp(i s q)``."""

    def operator(self, domain):
        return np.exp(1j * self._signed * domain)


class P(_AxisGate):
    """Quadratic phase gate ``exp(i s q^2 / 2)``."""

    def operator(self, domain):
        return np.exp(0.5j * self._signed * domain ** 2)


class D(SingleModeGate):
    """Displacement by ``s = [s_q, s_p]`` = ``X(s_q)`` followed by ``Z(s_p)``."""

    def __init__(self, index, s, **kwargs):
        if len(s) != 2:
            raise ValueError("s must have exactly 2 elements.")
        super().__init__(index, arg=s, **kwargs)

    def apply(self, mps: MPS, **kwargs):
        sign = -1 if self.dagger else 1
        X(self.index, sign * self.arg[0]).apply(mps, **kwargs)
        Z(self.index, sign * self.arg[1]).apply(mps, **kwargs)


class Phase(_AxisGate):
    """Phase-space rotation by ``angle``.  Declared upstream but raising ``NotImplementedError``
    (``gates.py:261-269``); implemented here with the fractional-Fourier kernel the reference already uses for
    ``Homodyne`` (``utils.rotation``), so the operator is pinned by the rotation fixtures."""

    def __init__(self, index, angle: float, **kwargs):
        super().__init__(index, angle, **kwargs)

    def operator(self, domain):
        return rotation_matrix(domain, self._signed)


class S(_AxisGate):
    """Squeezing by ``r`` along ``angle``.  Declared upstream but unimplemented (``gates.py:249-258``): **no reference
    counterpart, parity unpinned**.  Convention: ``S(r, 0)`` maps the vacuum to ``squeezed_vac(q, r)`` (width ``e^r`` in
    q): ``psi'(q) = e^{-r/2} psi(q e^{-r})`` by band-limited resampling; other angles conjugate with rotations."""

    def __init__(self, index, r: float, angle: float = 0.0, **kwargs):
        super().__init__(index, r, **kwargs)
        self.angle = angle

    def operator(self, domain):
        r = self._signed
        stretch = np.exp(-r / 2) * sinc_matrix(domain, domain * np.exp(-r))
        if np.isclose(np.sin(self.angle), 0):
            return stretch
        return rotation_matrix(domain, self.angle) @ stretch @ rotation_matrix(domain, -self.angle)

    matrix = operator      # earlier name


# ---- two-mode gates ------------------------------------------------------------------------------------------------
class _PlaneResampling(TwoModeGate):
    """Two-mode gate that re-samples the (q_left, q_right) plane at ``source_points(x, y)`` with bilinear weights --
    the per-bond-pair ``RegularGridInterpolator`` loop of the reference (``gates.py:74-80,187-189``) as one table."""

    def affine_map(self) -> tuple[float, float, float, float]:
        """``(a00, a01, a10, a11)``: output point ``(x, y)`` reads the input plane at ``(a00 x + a01 y, a10 x + a11 y)``."""
        raise NotImplementedError

    def source_points(self, x: np.ndarray, y: np.ndarray):
        a00, a01, a10, a11 = self.affine_map()
        return a00 * x + a01 * y, a10 * x + a11 * y

    def apply(self, mps: MPS, rng=None, **_):
        grid = mps.domain
        if mps.layout == "sites":          # the site register evaluates the map in the kernel: no table
            mps.reg.apply_plane_affine(grid, self.affine_map(), self.left_index, **_truncation(self, mps, rng))
            return

        def table():
            x, y = np.meshgrid(grid, grid, indexing="ij")
            return plane_resample_table(grid, *self.source_points(x, y))

        cols, weights = _cached(self, "table", grid, table)
        mps.reg.apply_two_mode_gather(cols, weights, self.left_index, self.right_index, **_truncation(self, mps, rng))


class BS(_PlaneResampling):
    """Beam splitter: the plane is rotated by ``angle`` (sign flips with the index order and with ``dagger``)."""

    def __init__(self, index1, index2, angle: float = np.pi / 4, **kwargs):
        super().__init__(index1, index2, arg=angle, **kwargs)

    def __repr__(self):
        return f"{type(self).__name__}{_pi_fraction(self.arg)}_{self.index1},{self.index2}"

    def affine_map(self):
        theta = self.arg * (1 if self.index1 < self.index2 else -1) * (-1 if self.dagger else 1)
        return np.cos(theta), np.sin(theta), -np.sin(theta), np.cos(theta)


class CX(_PlaneResampling):
    """Controlled displacement in q: a shear of the plane along the target's axis."""

    def __init__(self, control, target, s: float = 1.0, **kwargs):
        super().__init__(control, target, arg=s, **kwargs)

    def __repr__(self):
        return Gate.__repr__(self) + f"_{self.index1},{self.index2}"

    def affine_map(self):
        sign = -1.0 if self.dagger else 1.0
        if self.index1 < self.index2:        # the left mode controls: (x, y - sign x)
            return 1.0, 0.0, -sign, 1.0
        return 1.0, -sign, 0.0, 1.0          # (x - sign y, y)


class SWAP(TwoModeGate):
    """Exchange two neighbouring modes (a transposition of the plane)."""

    def apply(self, mps: MPS, rng=None, **_):
        d = len(mps.domain)
        transposed = np.arange(d * d).reshape(d, d).T.reshape(-1, 1)          # new[i, j] = old[j, i]
        mps.reg.apply_two_mode_gather(transposed, np.ones(transposed.shape, dtype=np.complex128),
                                      self.left_index, self.right_index, **_truncation(self, mps, rng))


class CZ(TwoModeGate):
    """Controlled displacement in p: the phase ``exp(i s q1 q2)`` on every point of the plane."""

    def __init__(self, index1, index2, s: float = 1.0, **kwargs):
        super().__init__(index1, index2, arg=s, **kwargs)

    def apply(self, mps: MPS, rng=None, **_):
        strength = -self.arg if self.dagger else self.arg
        if mps.layout == "sites":          # phases evaluated in the kernel
            mps.reg.apply_plane_phase(mps.domain, strength, self.left_index, **_truncation(self, mps, rng))
            return
        plane = _cached(self, "plane", mps.domain, lambda: np.exp(1j * strength * np.outer(mps.domain, mps.domain)))
        mps.reg.apply_two_mode(plane, self.left_index, self.right_index, **_truncation(self, mps, rng))


# ---- measurements and insertion -----------------------------------------------------------------------------------
class Mq(Measurement):
    """Homodyne measurement of q: draw (``rng.choice`` over the grid, as upstream) or take the forced grid point
    nearest to ``result``, then keep that slice of the register, renormalised, and drop the mode."""

    def apply(self, mps: MPS, rng=None, **_):
        grid, dq = mps.domain, mps.diff
        weights = mps.marginal(self.index) * dq                       # probability of each grid cell
        if self.result is None:
            pick = rng.choice(range(len(grid)), p=weights / np.sum(weights))
        else:
            pick = int(np.argmin(np.abs(grid - self.result)))
        value, density = grid[pick], weights[pick] / dq
        if len(mps) == 1:
            return value            # upstream quirk: bare value, register untouched, nothing recorded
        mps.reg.project(self.index, int(pick), 1.0 / np.sqrt(density))
        return MeasurementResult(value, density)


class Mp(Mq):
    """Homodyne measurement of p: inverse Fourier gate, then ``Mq``."""

    def apply(self, mps: MPS, **kwargs):
        mps.reg.apply_mode(_cached(self, "inverse fourier", mps.domain, lambda: fourier_matrix(mps.domain, inv=True)), self.index)
        return Mq.apply(self, mps, **kwargs)


class Homodyne(Mq):
    """Homodyne measurement of the quadrature at ``angle``: rotate by ``-angle``, then ``Mq`` (angles that are
    multiples of pi reduce to ``Mq`` with the sign of cos(angle) applied to the outcome)."""

    def __init__(self, index, angle: float, result: float = None, **kwargs):
        super().__init__(index, result, arg=angle, **kwargs)

    def __repr__(self):
        forced = f" = {round(self.result, REPR_DIGITS)}" if self.result else ""
        return f"{type(self).__name__}{_pi_fraction(self.arg)}_{self.index}{forced}"

    def apply(self, mps: MPS, **kwargs):
        if np.isclose(np.sin(self.arg), 0):
            logger.info("\tsin(angle) ≈ 0 detected: Using Mq gate instead.")
            outcome = Mq.apply(self, mps, **kwargs)
            outcome.result *= np.round(np.cos(self.arg))
            return outcome
        mps.reg.apply_mode(_cached(self, "rotation", mps.domain, lambda: rotation_matrix(mps.domain, -self.arg)), self.index)
        return Mq.apply(self, mps, **kwargs)


class Insert(SingleModeGate):
    """Insert a mode prepared in ``state`` (a ``State`` member or a wavefunction sampled on the domain) at ``index``."""

    def __init__(self, index: int, state, *, gkp_epsilon: float = None, **kwargs):
        if kwargs.pop("dagger", False):
            logger.info("%s gates ignore adjoint/dagger.", type(self).__name__)
        super().__init__(index, arg=state, **kwargs)
        self.gkp_epsilon = gkp_epsilon

    def apply(self, mps: MPS, rng=None, **_):
        if not 0 <= self.index <= len(mps):
            raise IndexError(f"Cannot insert mode at index {self.index} for MPS of length {len(mps)}")
        prepared = self.arg.eval(mps.domain, self.gkp_epsilon) if hasattr(self.arg, "eval") else np.asarray(self.arg)
        mps.reg.insert(self.index, prepared, **_truncation(self, mps, rng))
