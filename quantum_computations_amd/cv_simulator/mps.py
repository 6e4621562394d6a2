"""Register of the CV simulator behind the reference's ``MPS`` surface, in HBM, in one of two layouts.

``layout="sites"`` (default) is the reference's own data structure -- a chain of ``(chi_l, d, chi_r)`` site tensors
(``simulators/cv_simulator/mps.py:102-201``) re-compressed with a truncated SVD after every two-mode gate -- kept on the
device by ``site_register.SiteRegister``; this is what reaches the reference's d = 1000 grids, what
honours the truncation keywords, and what a script written for the reference gets when it calls ``MPS(qs, tensors)``.

``layout="dense"`` (opt-in) holds the *contracted* tensor ``psi[q_0, ..., q_{m-1}]`` as a ``QuditState``
(``d = len(domain)`` levels per mode): every gate is one bandwidth-bound pass, results equal the reference's with
truncation disabled (``rel_err = 0``), truncation keywords are ignored, and the size is limited by ``d^m * 16`` bytes
(the Fock-truncated configuration of BASELINE.json, ``cv_simulator.fock``, is this kind of register).

Either way the class keeps the constructor, ``domain`` / ``diff``, ``len``, ``copy``, ``validate``, ``contract``,
``norm`` and ``partial_density_mps`` of the reference class, and the gate classes drive both through the same calls.
"""
from __future__ import annotations

import inspect
from functools import reduce

import numpy as np

from ..device import QuditState


def tensor_svd(tensor, left_indices, right_indices, *, max_bond_dim: int = np.inf, abs_err: float = 0,
               rel_err: float = 1e-12, rng_seed: int = None, device: int = 0):
    """Split a host tensor by a truncated SVD on the device (``tensor_svd``, mps.py:52-97): returns ``m1`` with legs
    ``left_indices + [j]`` and ``m2`` with legs ``[j] + right_indices``; the singular values are shared as square
    roots and the kept rank follows the reference's rule (tail sum <= max(abs_err, rel_err * sum), then the cap)."""
    from .site_register import SiteRegister
    left_indices, right_indices = list(left_indices), list(right_indices)
    if sorted(left_indices + right_indices) != list(range(np.ndim(tensor))):
        raise IndexError("Output indices does not match indices of initial tensor")
    shape = np.shape(tensor)
    rows = int(np.prod([shape[i] for i in left_indices]))
    cols = int(np.prod([shape[i] for i in right_indices]))
    matrix = np.moveaxis(np.asarray(tensor), left_indices + right_indices, range(len(shape))).reshape(rows, cols)
    worker = SiteRegister([], 1, device)
    m1, m2, r = worker._split(worker._upload(matrix), rows, cols, max_bond_dim=max_bond_dim, abs_err=abs_err,
                              rel_err=rel_err, rng_seed=rng_seed)
    return (m1.cpu().numpy().reshape([shape[i] for i in left_indices] + [r]),
            m2.cpu().numpy().reshape([r] + [shape[i] for i in right_indices]))


# the keyword-only truncation options gates and the simulator accept (mps.py:99-100)
SVD_OPTIONS = {name: p for name, p in inspect.signature(tensor_svd).parameters.items()
               if p.kind == inspect.Parameter.KEYWORD_ONLY}


class MPS:
    def __init__(self, domain: np.ndarray, tensors: list[np.ndarray], *, device: int = 0, layout: str = "sites"):
        """``tensors``: one entry per mode -- a wavefunction on ``domain`` (1-D) or an MPS site ``(chi_l, d, chi_r)``.
        ``layout="sites"`` (the reference's behaviour): the sites are uploaded as they are and stay a matrix-product
        state; ``layout="dense"``: the sites are contracted on the host (small registers) and the dense tensor is
        uploaded."""
        if layout not in ("dense", "sites"):
            raise ValueError("layout must be 'dense' or 'sites'")
        self.domain: np.ndarray = domain
        self._check_domain()
        self.diff: float = abs(domain[-1] - domain[0]) / (len(domain) - 1)
        sites = [np.asarray(t).reshape(1, -1, 1) if np.ndim(t) == 1 else np.asarray(t) for t in tensors]
        self._check_sites(sites)
        d = len(domain)
        if layout == "sites":
            from .site_register import SiteRegister
            self.reg = SiteRegister(sites, d, device)
        elif sites:
            dense = np.squeeze(reduce(lambda a, b: np.tensordot(a, b, axes=1), sites), axis=(0, -1))
            self.reg = QuditState.from_numpy(np.ascontiguousarray(dense, dtype=np.complex128).reshape((d,) * len(sites)),
                                             device)
        else:
            self.reg = QuditState.zeros(0, d, device)

    @classmethod
    def _wrap(cls, domain: np.ndarray, reg: QuditState) -> "MPS":
        out = cls.__new__(cls)
        out.domain = domain
        out.diff = abs(domain[-1] - domain[0]) / (len(domain) - 1)
        out.reg = reg
        return out

    # ---- container protocol (the reference class is a list of site tensors, mps.py:102-134) ----------
    def __len__(self):
        return self.reg.dims[0]

    def _sites_only(self, what: str):
        if self.layout != "sites":
            raise AttributeError(f"a dense register has no site tensors ({what}): build the MPS with layout='sites'")

    def __getitem__(self, index):
        """Host copy of site ``index`` (or of a slice of sites)."""
        self._sites_only("indexing")
        picked = self.reg.sites[index]
        return [t.cpu().numpy() for t in picked] if isinstance(index, slice) else picked.cpu().numpy()

    def __setitem__(self, index: int, tensor: np.ndarray) -> None:
        """Replace site ``index`` (uploaded; shapes are the caller's responsibility until :meth:`validate`)."""
        self._sites_only("item assignment")
        tensor = np.asarray(tensor)
        if tensor.ndim != 3:
            raise ValueError(f"Tensor at index {index} does not have exactly three axes.")
        self.reg.sites[index] = self.reg._upload(tensor)

    def __iter__(self):
        self._sites_only("iteration")
        return iter(self.reg.site_arrays())

    def copy(self) -> "MPS":
        return MPS._wrap(self.domain.copy(), self.reg.copy())

    @property
    def layout(self) -> str:
        return getattr(self.reg, "layout", "dense")

    @property
    def tensors(self) -> list[np.ndarray]:
        """Host copies of the site tensors (``layout="sites"`` only; the dense register has no sites)."""
        self._sites_only("tensors")
        return self.reg.site_arrays()

    def shape(self):
        if self.layout == "sites":
            return self.reg.shape()
        n, d = self.reg.dims
        return ("dense",) + (d,) * n

    # ---- validation (mps.py:136-161) ----------------------------------------------------------------
    def _check_domain(self):
        if not isinstance(self.domain, np.ndarray) or self.domain.ndim != 1:
            raise TypeError("Domain must be a 1D numpy array.")
        if not np.allclose(np.diff(self.domain, 2), 0, atol=np.finfo(self.domain.dtype).eps ** 0.5):
            raise ValueError("Domain is not an arithmetic progression.")

    def _check_sites(self, sites):
        for idx, t in enumerate(sites):
            if t.ndim != 3:
                raise ValueError(f"Tensor at index {idx} does not have exactly three axes.")
            if t.shape[1] != len(self.domain):
                raise ValueError(f"Tensor at index {idx} does not have the right physical dimension.")
        if sites:
            if sites[0].shape[0] != 1:
                raise ValueError("Left-most tensor does not have a trivial left edge")
            if sites[-1].shape[2] != 1:
                raise ValueError("Right-most tensor does not have a trivial right edge")
        for idx, (a, b) in enumerate(zip(sites, sites[1:])):
            if a.shape[2] != b.shape[0]:
                raise ValueError(f"Tensors at indices {idx} and {idx + 1} do not have compatible bond dimensions.")

    def validate(self):
        self._check_domain()
        if not np.isclose(self.diff, abs(self.domain[-1] - self.domain[0]) / (len(self.domain) - 1)):
            raise ValueError("Stored difference does not match current domain.")
        if self.reg.dims[1] != len(self.domain):
            raise ValueError("Register does not have the right physical dimension.")

    # ---- read-out ---------------------------------------------------------------------------------------
    def contract(self) -> np.ndarray:
        """The dense tensor ``psi[q_0, ..., q_{m-1}]`` on the host (``MPS.contract``, mps.py:163-164)."""
        return self.reg.to_numpy()

    def norm(self) -> float:
        """``sqrt(int |psi|^2)`` with the grid measure ``diff^m`` (mps.py:166-170)."""
        return float(np.sqrt(self.reg.norm2() * self.diff ** len(self)))

    def marginal(self, axis: int) -> np.ndarray:
        """Diagonal of :meth:`partial_density_mps` -- the position density of mode ``axis`` -- computed on the device."""
        if axis < 0 or axis >= len(self):
            raise IndexError(f"axis={axis} out of bounds")
        return self.reg.marginal(axis) * self.diff ** (len(self) - 1)

    def partial_density_mps(self, axis: int) -> np.ndarray:
        """Reduced density matrix of mode ``axis`` (mps.py:176-190).  Host contraction of the downloaded tensor:
        for read-out of small registers; measurements use :meth:`marginal`, which stays on the GPU."""
        if axis < 0 or axis >= len(self):
            raise IndexError(f"axis={axis} out of bounds")
        if self.layout == "sites":
            return self.reg.reduced_density(axis) * self.diff ** (len(self) - 1)
        psi = np.moveaxis(self.contract(), axis, 0).reshape(len(self.domain), -1)
        return (psi @ psi.conj().T) * self.diff ** (len(self) - 1)

    @staticmethod
    def fidelity(a: "MPS", b: "MPS") -> float:
        """``|<a|b>|^2`` on a shared grid.  (The reference's version, mps.py:192-201, contracts ``a`` with itself
        and never reads ``b``; this one computes what its docstring says.)"""
        if len(a) != len(b):
            raise ValueError("registers of different sizes")
        overlap = np.vdot(a.contract(), b.contract()) * a.diff ** len(a)
        return float(np.abs(overlap) ** 2)
