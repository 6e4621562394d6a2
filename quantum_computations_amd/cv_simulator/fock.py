"""Fock-truncated mode registers: BASELINE.json config 4 (6 modes x cutoff d = 32, squeezing + beam splitters).

The reference has no Fock-basis operators at all (its ``S`` / ``Phase`` raise ``NotImplementedError``,
``simulators/cv_simulator/gates.py:249-269``; the register is a position grid): **parity unpinned** for the matrices
built here.  They are derived from the truncated ladder operator with ``scipy.linalg.expm`` on the host; what IS pinned
is the contraction that applies them (``qsv_apply_mode1`` / ``qsv_apply_mode2`` == ``np.tensordot``).  The gate classes
keep the cv_simulator API shape: ``SingleModeGate(index, ...)``, ``TwoModeGate(i, j, ...)`` (neighbours), ``apply(state)``.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import expm

from ..device import QuditState
from .gate_abc import SingleModeGate, TwoModeGate


def annihilation(d: int) -> np.ndarray:
    return np.diag(np.sqrt(np.arange(1, d)), 1).astype(np.complex128)


def squeeze_matrix(d: int, r: float, angle: float = 0.0) -> np.ndarray:
    """exp((z* a^2 - z a^dagger^2) / 2), z = r e^{i angle}, truncated to d levels."""
    a = annihilation(d)
    z = r * np.exp(1j * angle)
    return expm(0.5 * (np.conj(z) * a @ a - z * a.conj().T @ a.conj().T))


def phase_matrix(d: int, angle: float) -> np.ndarray:
    return np.exp(-1j * angle * np.arange(d))          # diagonal: exp(-i angle n)


def beamsplitter_blocks(d: int, theta: float, phi: float = 0.0):
    """The beam splitter exp(theta (e^{i phi} a b^dagger - e^{-i phi} a^dagger b)) on two d-level modes, as its
    2d - 1 photon-number blocks ``[(plane_indices, block_matrix), ...]``, ``plane_indices = n_a * d + n_b`` with
    ``n_a + n_b = N``.  The truncated generator only connects |n_a, n_b> with |n_a -+ 1, n_b +- 1>, so it is block
    diagonal in N and its exponential is the exponential of each (at most d x d, tridiagonal) block -- 63 small
    ``expm`` calls for d = 32 instead of one on a 1024 x 1024 matrix (7 s)."""
    blocks = []
    for total in range(2 * d - 1):
        na = np.arange(max(0, total - d + 1), min(d, total + 1))
        nb = total - na
        gen = np.zeros((na.size, na.size), dtype=np.complex128)
        # a b^dagger |n_a, n_b> = sqrt(n_a (n_b + 1)) |n_a - 1, n_b + 1>: column j -> row j - 1
        up = np.sqrt(na[1:] * (nb[1:] + 1.0))
        gen[np.arange(na.size - 1), np.arange(1, na.size)] = theta * np.exp(1j * phi) * up
        gen[np.arange(1, na.size), np.arange(na.size - 1)] = -theta * np.exp(-1j * phi) * up
        blocks.append(([int(a * d + b) for a, b in zip(na, nb)], expm(gen)))
    return blocks


def beamsplitter_matrix(d: int, theta: float, phi: float = 0.0, *, dense_expm: bool = False) -> np.ndarray:
    """The same operator as a (d^2 x d^2) matrix, row / column index n_a * d + n_b.  It conserves n_a + n_b, so every
    row has at most d non-zero entries.  ``dense_expm`` exponentiates the full generator instead (cross-check)."""
    if dense_expm:
        a = np.kron(annihilation(d), np.identity(d))
        b = np.kron(np.identity(d), annihilation(d))
        return expm(theta * (np.exp(1j * phi) * a @ b.conj().T - np.exp(-1j * phi) * a.conj().T @ b))
    out = np.zeros((d * d, d * d), dtype=np.complex128)
    for idx, block in beamsplitter_blocks(d, theta, phi):
        out[np.ix_(idx, idx)] = block
    return out


def sparse_rows(matrix: np.ndarray, tol: float = 0.0):
    """Row-compressed (cols, vals) of a d^2 x d^2 matrix for ``qsv_apply_mode2_gather``; unused slots are -1 / 0."""
    rows = matrix.shape[0]
    mask = np.abs(matrix) > tol
    nnz = max(1, int(mask.sum(axis=1).max()))
    cols = np.full((rows, nnz), -1, dtype=np.int32)
    vals = np.zeros((rows, nnz), dtype=np.complex128)
    for r in range(rows):
        idx = np.nonzero(mask[r])[0]
        cols[r, : idx.size] = idx
        vals[r, : idx.size] = matrix[r, idx]
    return cols, vals


class FockState:
    """``n_modes`` modes truncated to ``d`` Fock levels each, dense complex128 in HBM, all starting in vacuum."""

    def __init__(self, n_modes: int, d: int, device: int = 0):
        self.reg = QuditState.zeros(n_modes, d, device)
        self.d = d

    def __len__(self):
        return self.reg.dims[0]

    def contract(self) -> np.ndarray:
        return self.reg.to_numpy()

    def norm(self) -> float:
        return float(np.sqrt(self.reg.norm2()))


class S(SingleModeGate):
    def __init__(self, index, r: float, angle: float = 0.0, **kwargs):
        super().__init__(index, arg=r, **kwargs)
        self.angle = angle

    def apply(self, state: FockState, **_):
        r = -self.arg if self.dagger else self.arg
        state.reg.apply_mode(squeeze_matrix(state.d, r, self.angle), self.index)


class Phase(SingleModeGate):
    def __init__(self, index, angle: float, **kwargs):
        super().__init__(index, arg=angle, **kwargs)

    def apply(self, state: FockState, **_):
        state.reg.apply_mode(phase_matrix(state.d, -self.arg if self.dagger else self.arg), self.index)


def photon_number_blocks(matrix: np.ndarray, d: int):
    """Split a photon-number-conserving d^2 x d^2 operator into its 2d-1 anti-diagonal blocks:
    ``[(plane_indices, block_matrix), ...]`` with ``plane_indices = n_a * d + n_b`` for ``n_a + n_b = N``."""
    blocks = []
    for total in range(2 * d - 1):
        idx = [na * d + (total - na) for na in range(max(0, total - d + 1), min(d, total + 1))]
        blocks.append((idx, matrix[np.ix_(idx, idx)]))
    return blocks


class BS(TwoModeGate):
    """Beam splitter in the Fock basis.  Default ``method="blocks"``: one small dense matrix per total photon
    number, applied in place (``qsv_apply_mode2_blocks``).  ``"gather"`` uses the row-sparse form (<= d entries per
    row), ``"dense"`` the full d^2 x d^2 matrix -- same result, kept for cross-checks."""

    def __init__(self, index1, index2, angle: float = np.pi / 4, *, method: str = "blocks", dense: bool = False,
                 **kwargs):
        super().__init__(index1, index2, arg=angle, **kwargs)
        self.method = "dense" if dense else method

    def apply(self, state: FockState, **_):
        theta = -self.arg if self.dagger else self.arg
        if self.method == "blocks":
            state.reg.apply_two_mode_blocks(beamsplitter_blocks(state.d, theta), self.index1, self.index2)
            return
        m = beamsplitter_matrix(state.d, theta)
        if self.method == "dense":
            state.reg.apply_two_mode(m, self.index1, self.index2)
        elif self.method == "gather":
            cols, vals = sparse_rows(m, tol=1e-300)
            state.reg.apply_two_mode_gather(cols, vals, self.index1, self.index2)
        else:
            raise ValueError(f"unknown method {self.method!r}")


def spot_check_register(reg, rng, fibres: int = 6, planes: int = 3, r: float = 0.3, theta: float = np.pi / 4) -> dict:
    """Size-independent amplitude check of the d-level gate kernels on a register of any size (config 4: 6 modes x
    d = 32 = 16 GiB, where no host copy of the register is wanted): one S(r) on every mode and one BS(theta) on every
    neighbouring pair, each compared on sampled fibres / (d, d) planes with ``U @ in`` evaluated from the sampled
    inputs alone -- ``np.tensordot(M, T, [1, axis])`` of cv_simulator/utils.py:15,37 restricted to those fibres, the
    plane map of cv_simulator/gates.py:58-84 restricted to those planes.  Returns the largest deviation per gate kind
    and the kernels that ran.  The register is overwritten with seeded pseudo-random amplitudes first."""
    n_modes, d = reg.dims
    reg.fill_random(int(rng.integers(1, 1 << 30)))
    out = {"S_max_abs_err": 0.0, "BS_max_abs_err": 0.0, "kernels": [], "fibres_per_mode": fibres, "planes_per_pair": planes}

    def gather(index_lists):
        return np.array([reg.download(int(i), 1)[0] for i in index_lists])

    for mode in range(n_modes):
        stride = d ** (n_modes - 1 - mode)
        m = squeeze_matrix(d, r, 0.0)
        bases = []
        for _ in range(fibres):
            left, right = int(rng.integers(0, d ** mode)), int(rng.integers(0, stride))
            bases.append(left * d * stride + right)
        idx = [b + a * stride for b in bases for a in range(d)]
        before = gather(idx).reshape(fibres, d)
        reg.apply_mode(m, mode)
        out["kernels"].append(reg.last_kernel())
        after = gather(idx).reshape(fibres, d)
        out["S_max_abs_err"] = max(out["S_max_abs_err"], float(np.max(np.abs(after - before @ m.T))))
    dense = beamsplitter_matrix(d, theta)
    blocks = beamsplitter_blocks(d, theta)
    for mode in range(n_modes - 1):
        stride = d ** (n_modes - 2 - mode)
        bases = []
        for _ in range(planes):
            left, right = int(rng.integers(0, d ** mode)), int(rng.integers(0, stride))
            bases.append(left * d * d * stride + right)
        if stride == 1:
            before = np.array([reg.download(b, d * d) for b in bases])
        else:
            before = gather([b + ab * stride for b in bases for ab in range(d * d)]).reshape(planes, d * d)
        reg.apply_two_mode_blocks(blocks, mode, mode + 1)
        out["kernels"].append(reg.last_kernel())
        if stride == 1:
            after = np.array([reg.download(b, d * d) for b in bases])
        else:
            after = gather([b + ab * stride for b in bases for ab in range(d * d)]).reshape(planes, d * d)
        out["BS_max_abs_err"] = max(out["BS_max_abs_err"], float(np.max(np.abs(after - before @ dense.T))))
    out["norm2_after"] = reg.norm2()
    out["kernels"] = sorted(set(out["kernels"]))
    return out

