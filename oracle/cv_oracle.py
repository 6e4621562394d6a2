"""TEST INFRASTRUCTURE ONLY — NumPy restatement of the mode-axis contractions of the reference's cv_simulator.

The reference applies every single-mode grid operator as ``np.tensordot(M, T, [1, axis])`` followed by
``np.moveaxis(res, 0, axis)`` (``simulators/cv_simulator/utils.py:15-16`` for the sinc matrix,
``:37-38`` for the fractional-Fourier kernel) and two-mode maps on the ``(chi_l, d, d, chi_r)`` contraction
of two neighbouring sites (``gates.py:73,160,185``).  These are the same primitives; the dense registers the
HIP kernels work on are just MPS sites with trivial bonds.

Pinned by ``tests/golden/cv_operators.npz`` for the operators the reference defines (X, F, Z, P, rotation, CZ,
BS, CX, SWAP on a position grid).  Fock-basis squeezing / beam-splitter matrices have no reference counterpart
(``S`` / ``Phase`` raise ``NotImplementedError``, ``gates.py:249-269``): **parity unpinned** for those
matrices; only this contraction is pinned.
"""
from __future__ import annotations

import numpy as np


def apply_axis(tensor: np.ndarray, matrix: np.ndarray, axis: int) -> np.ndarray:
    """``out[..., i, ...] = sum_j M[i, j] T[..., j, ...]`` along ``axis`` (``utils.py:15-16``)."""
    res = np.tensordot(matrix, tensor, [1, axis])
    return np.moveaxis(res, 0, axis)


def apply_axis_diag(tensor: np.ndarray, diag: np.ndarray, axis: int) -> np.ndarray:
    """Diagonal phase along ``axis`` (the einsum of ``gates.py:222,246``)."""
    shape = [1] * tensor.ndim
    shape[axis] = -1
    return tensor * np.reshape(diag, shape)


def apply_two_axes(tensor: np.ndarray, matrix: np.ndarray, axis0: int, axis1: int) -> np.ndarray:
    """``d^2 x d^2`` operator on the ``(axis0, axis1)`` plane; row/column index = ``i0 * d + i1``."""
    d = tensor.shape[axis0]
    g = np.asarray(matrix).reshape(d, d, d, d)
    res = np.tensordot(g, tensor, axes=([2, 3], [axis0, axis1]))
    return np.moveaxis(res, [0, 1], [axis0, axis1])


def apply_two_axes_diag(tensor: np.ndarray, plane: np.ndarray, axis0: int, axis1: int) -> np.ndarray:
    """Elementwise phase on the (axis0, axis1) plane: the ``cz`` factor of ``gates.py:159-160``."""
    d = tensor.shape[axis0]
    plane = np.asarray(plane).reshape(d, d)
    if axis0 > axis1:
        plane, axis0, axis1 = plane.T, axis1, axis0
    shape = [1] * tensor.ndim
    shape[axis0], shape[axis1] = d, d
    return tensor * plane.reshape(shape)
