"""Grid-basis linear maps of the CV simulator, as explicit operators.

Mirror of ``simulators/cv_simulator/utils.py`` (same function names and argument meaning).  The reference applies each
map to an MPS site with ``np.tensordot`` on the host; here the ``*_matrix`` builders return the ``d x d`` operator
(host, O(d^2)) and the device applies it along a mode axis (``qsv_apply_mode1`` / ``qsv_tensor_apply_axis``).  The
array-in/array-out functions are kept for API compatibility and small host arrays.
"""
from __future__ import annotations

import logging

import numpy as np
import scipy.fft as fft

logger = logging.getLogger("simulators." + __name__.split(".", 1)[1])


def _spacing(xs: np.ndarray) -> float:
    return (xs[-1] - xs[0]) / (len(xs) - 1)


# ---- Whittaker-Shannon (sinc) interpolation: utils.py:9-18 ---------------------------------------------------
def sinc_matrix(xs: np.ndarray, new_xs: np.ndarray) -> np.ndarray:
    """``S[i, j] = sinc((new_xs[i] - xs[j]) / dx)``: band-limited resampling from grid ``xs`` onto ``new_xs``."""
    return np.sinc(np.subtract.outer(new_xs, xs) / _spacing(xs))


def _apply_along(matrix: np.ndarray, values: np.ndarray, axis: int) -> np.ndarray:
    return np.moveaxis(np.tensordot(matrix, values, [1, axis]), 0, axis)


def whittaker_shannon(xs: np.ndarray, ys: np.ndarray, new_xs: np.ndarray, axis: int = 0) -> np.ndarray:
    return _apply_along(sinc_matrix(xs, new_xs), ys, axis)


interpolate = whittaker_shannon


# ---- phase-space rotation (fractional Fourier transform): utils.py:22-39 --------------------------------------
def rotation_matrix(qs: np.ndarray, theta: float, new_qs: np.ndarray | None = None) -> np.ndarray:
    """``R[i, j] = dq * <new_qs[i]| e^{-i theta n} |qs[j]>`` sampled on the grid (rows: output points)."""
    new_qs = qs if new_qs is None else new_qs
    dq = (max(qs) - min(qs)) / (len(qs) - 1)
    s, c = np.sin(theta), np.cos(theta)
    quad = c * np.add.outer(new_qs ** 2, qs ** 2) / 2 - np.outer(new_qs, qs)
    return (2 * np.pi * abs(s)) ** -0.5 * np.exp(quad / (1j * s)) * dq


def rotation(qs: np.ndarray, tensor: np.ndarray, theta: float, axis: int = 0, new_qs: np.ndarray = None) -> np.ndarray:
    return _apply_along(rotation_matrix(qs, theta, new_qs), tensor, axis)


# ---- continuous Fourier transform by FFT: utils.py:41-84 -------------------------------------------------------
def CFT(qs: np.ndarray, tensor: np.ndarray, axis: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """``F(p) = (2 pi)^{-1/2} int dq f(q) e^{-ipq}`` on the FFT's momentum grid; returns ``(ps, values)``."""
    n = tensor.shape[axis]
    period = (qs[-1] - qs[0]) * n / (n - 1)
    ps = fft.fftshift(fft.fftfreq(n, d=period / (n * 2 * np.pi)))
    spectrum = fft.fftshift(fft.fft(tensor, axis=axis), axes=axis)
    weights = period / (n * np.sqrt(2 * np.pi)) * np.exp(-1j * ps * qs[0])
    shape = [1] * spectrum.ndim
    shape[axis] = -1
    return ps, spectrum * weights.reshape(shape)


def iCFT(qs: np.ndarray, tensor: np.ndarray, axis: int = 0) -> tuple[np.ndarray, np.ndarray]:
    ps, values = CFT(qs, tensor, axis=axis)
    return np.flip(-ps), np.flip(values, axis=axis)


def fourier(qs: np.ndarray, tensor: np.ndarray, axis: int = 0, ps: np.ndarray = None, inv: bool = False) -> np.ndarray:
    """The Fourier gate ``F`` (``inv``: its adjoint) evaluated at ``ps`` (default: the same grid) by sinc resampling."""
    ps = qs if ps is None else ps
    grid, values = CFT(qs, tensor, axis=axis) if inv else iCFT(qs, tensor, axis=axis)
    if ps[-1] - ps[0] > grid[-1] - grid[0]:
        logger.warning("Evaluation outside of the Nyquist bandwidth might have unintended sideeffects.")
    wrapped = (ps - grid[-1]) % (grid[-1] - grid[0]) + grid[0]      # the sampled transform is periodic
    return whittaker_shannon(grid, values, wrapped, axis=axis)


def fourier_matrix(qs: np.ndarray, inv: bool = False) -> np.ndarray:
    """The ``d x d`` operator of :func:`fourier` on grid ``qs`` (column j = image of the j-th grid basis vector)."""
    return fourier(qs, np.identity(len(qs), dtype=np.complex128), axis=0, inv=inv)


# ---- bilinear resampling of the (q1, q2) plane: what BS / CX do per bond pair (gates.py:74-80,187-189) ----------
def _cell(grid: np.ndarray, x: np.ndarray):
    """Lower cell index, fractional position and in-bounds mask of ``x`` on ``grid`` (linear interpolation with
    zero fill outside, as ``RegularGridInterpolator(method='linear', bounds_error=False, fill_value=0)``)."""
    i = np.clip(np.searchsorted(grid, x) - 1, 0, grid.size - 2)
    frac = (x - grid[i]) / (grid[i + 1] - grid[i])
    inside = (x >= grid[0]) & (x <= grid[-1])
    return i, frac, inside


def plane_resample_table(qs: np.ndarray, x_new: np.ndarray, y_new: np.ndarray):
    """Gather table of ``new[i, j] = interp(old)(x_new[i, j], y_new[i, j])``.

    Returns ``(cols, vals)`` of shape ``(d*d, 4)``: row ``i*d + j`` lists the four source plane points
    ``j0*d + j1`` and their bilinear weights (column ``-1`` / weight 0 where the point falls off the grid).
    """
    d = len(qs)
    i0, f0, in0 = _cell(qs, x_new.reshape(-1))
    i1, f1, in1 = _cell(qs, y_new.reshape(-1))
    inside = in0 & in1
    cols = np.stack([i0 * d + i1, i0 * d + i1 + 1, (i0 + 1) * d + i1, (i0 + 1) * d + i1 + 1], axis=1)
    vals = np.stack([(1 - f0) * (1 - f1), (1 - f0) * f1, f0 * (1 - f1), f0 * f1], axis=1)
    cols = np.where(inside[:, None], cols, -1).astype(np.int32)
    vals = np.where(inside[:, None], vals, 0.0)
    return cols, vals.astype(np.complex128)


def wigner(state, q, p):
    raise NotImplementedError("Evaluation of Wigner function not yet implemented")
