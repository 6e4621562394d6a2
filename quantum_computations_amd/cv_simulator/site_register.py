"""Matrix-product-state register of the CV simulator, resident in HBM.

The reference keeps a CV state as a list of ``(chi_l, d, chi_r)`` site tensors (``simulators/cv_simulator/mps.py:102-201``)
and re-compresses after every two-mode gate: contract the two neighbours, map the ``(q_left, q_right)`` plane, split with
a truncated SVD (``cv_simulator/gates.py:48-84,151-192``).  ``SiteRegister`` is that data structure with every tensor in
device memory and every step done by libqsv.so on the GPU:

====================================  ==========================================================================
step                                  entry point
====================================  ==========================================================================
single-mode operator on a site        ``qsv_tensor_apply_axis_dev`` (rocBLAS zgemm for grids >= 64 points)
single-mode phases (Z, P)             ``qsv_tensor_scale_axis``
theta = site_l . site_r               ``qsv_tensor_gemm``
CZ phases / BS, CX resampling / SWAP   ``qsv_tensor_plane_phase`` / ``qsv_tensor_plane_affine`` / ``qsv_tensor_plane_gather``
split + truncate                      ``qsv_tensor_svd_split`` / ``qsv_tensor_rsvd_split`` (rocSOLVER + the reference's rule)
homodyne read-out                     ``qsv_tensor_gemm`` environments + ``qsv_tensor_axis_overlap``; ``qsv_tensor_take_level``
Insert in the middle of the chain     ``qsv_tensor_insert_axis`` + ``qsv_tensor_svd_split``
====================================  ==========================================================================

PyTorch only owns the device buffers (allocation, host<->device copies, the current stream); no torch operator touches
the amplitudes.  The interface is the one ``QuditState`` offers the gate classes, so ``cv_simulator.gates`` drive either
register; the truncation keywords of the gates (``max_bond_dim``, ``abs_err``, ``rel_err``) arrive as keyword arguments.

Like the reference, the split is exact (rocSOLVER ``zgesvd``) unless ``max_bond_dim * 10 < min(matrix shape)``, where it
switches to the randomized range finder of ``mps.py:5-50`` (``qsv_tensor_rsvd_split``: GEMMs + Householder QR + a small
SVD), fed with the same Gaussian test matrix a seeded reference run would draw.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .. import _lib

OP_NONE, OP_TRANSPOSE, OP_CONJ_TRANSPOSE = 0, 1, 2
_DRAW_POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix="qsv-omega")     # see SiteRegister._split


_TORCH = None


def _torch():
    """The torch module, after checking once that a HIP device is there (buffers only: allocation, copies, streams)."""
    global _TORCH
    if _TORCH is None:
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("SiteRegister needs a HIP device: there is no CPU fallback")
        _TORCH = torch
    return _TORCH


class SiteRegister:
    layout = "sites"

    def __init__(self, sites: list[np.ndarray], d: int, device: int = 0):
        torch = _torch()
        self.device = int(device)
        self._dev = torch.device("cuda", self.device)
        self.d = int(d)
        self.sites = [self._upload(np.asarray(s).reshape(1, -1, 1) if np.ndim(s) == 1 else s) for s in sites]
        self._resident: OrderedDict[int, tuple] = OrderedDict()    # host operator id -> (host array, device copy)
        self._scratch_buffers: dict = {}
        self.last_singular_values: np.ndarray | None = None
        self.split_counts = {"exact": 0, "randomized": 0}          # which branch of tensor_svd the splits took

    # ---- plumbing ----------------------------------------------------------------------------------------
    def _upload(self, array, dtype=np.complex128):
        torch = _torch()
        host = np.ascontiguousarray(array, dtype=dtype)
        if not host.flags.writeable:
            host = host.copy()          # torch.from_numpy insists on a writable buffer (cached read-only operators)
        return torch.from_numpy(host).to(self._dev)

    def _empty(self, *shape):
        torch = _torch()
        return torch.empty(shape, dtype=torch.complex128, device=self._dev)

    def _scratch(self, slot: str, *shape):
        """A complex128 tensor of ``shape`` carved from this register's persistent scratch buffer ``slot`` (grow-only).
        The two-site temporaries (theta, its mapped image, the capacity-sized split outputs) are gigabytes at the
        reference's sizes and die within the gate: re-using one buffer per role keeps hipMalloc / hipFree -- which
        synchronise the device -- out of the gate loop."""
        torch = _torch()
        count = int(np.prod(shape))
        held = self._scratch_buffers.get(slot)
        if held is None or held.numel() < count:
            self._scratch_buffers.pop(slot, None)
            held = torch.empty(count + count // 8, dtype=torch.complex128, device=self._dev)
            self._scratch_buffers[slot] = held
        return held[:count].view(*shape)

    def _stream(self) -> C.c_void_p:
        torch = _torch()
        return C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)

    def _keep(self, host: np.ndarray, dtype=np.complex128):
        """Device copy of a host operator, cached while the caller keeps handing in the same array object."""
        key = id(host)
        hit = self._resident.get(key)
        if hit is not None and hit[0] is host:
            self._resident.move_to_end(key)
            return hit[1]
        dev = self._upload(host, dtype)
        self._resident[key] = (host, dev)
        while len(self._resident) > 16:
            self._resident.popitem(last=False)
        return dev

    @staticmethod
    def _p(t) -> C.c_void_p:
        return C.c_void_p(t.data_ptr())

    def _gemm(self, a, op_a: int, b, op_b: int, m: int, n: int, k: int, scratch: str | None = None):
        out = self._scratch(scratch, m, n) if scratch else self._empty(m, n)
        _lib.call("qsv_tensor_gemm", self.device, self._stream(), op_a, op_b, m, n, k, self._p(a), self._p(b),
                  self._p(out))
        return out

    # ---- register protocol shared with QuditState -----------------------------------------------------------
    @property
    def dims(self) -> tuple[int, int]:
        return len(self.sites), self.d

    def bond_dims(self) -> list[int]:
        return [int(s.shape[2]) for s in self.sites[:-1]]

    def shape(self):
        return tuple(tuple(int(x) for x in s.shape) for s in self.sites)

    def sync(self) -> None:
        _torch().cuda.synchronize(self._dev)

    def close(self) -> None:
        self.sites = []
        self._resident.clear()
        self._scratch_buffers.clear()
        _lib.call("qsv_tensor_release_workspace", self.device)

    def copy(self) -> "SiteRegister":
        out = SiteRegister([], self.d, self.device)
        out.sites = [s.clone() for s in self.sites]     # device-to-device copies
        return out

    def site_arrays(self) -> list[np.ndarray]:
        """The site tensors on the host (``MPS.tensors`` of the reference)."""
        return [s.cpu().numpy() for s in self.sites]

    def to_numpy(self) -> np.ndarray:
        """Contract the chain on the device (``MPS.contract``, mps.py:163-164) and download ``psi[q_0, ..., q_{m-1}]``."""
        if not self.sites:
            return np.ones((), dtype=np.complex128)
        acc = self.sites[0]
        rows = int(acc.shape[0] * acc.shape[1])
        for site in self.sites[1:]:
            chi, cols = int(site.shape[0]), int(site.shape[1] * site.shape[2])
            acc = self._gemm(acc, OP_NONE, site, OP_NONE, rows, cols, chi)
            rows = rows * self.d
        return acc.cpu().numpy().reshape((self.d,) * len(self.sites))

    # ---- environments (mps.py:166-190) ------------------------------------------------------------------------
    def _left_environment(self, stop: int):
        """conj of ``reduce("ab,aci,bcj -> ij")`` over the sites before ``stop`` (the conjugate is what chains through
        GEMMs with a conjugate-transposed first operand: X = G^H T, G' = X^H T)."""
        g = self._upload(np.ones((1, 1)))
        for t in self.sites[:stop]:
            cl, d, cr = (int(x) for x in t.shape)
            x = self._gemm(g, OP_CONJ_TRANSPOSE, t, OP_NONE, cl, d * cr, cl)          # X[b, (c, i)]
            g = self._gemm(x, OP_CONJ_TRANSPOSE, t, OP_NONE, cr, cr, cl * d)          # conj(res')[i, j]
        return g

    def _right_environment(self, start: int):
        """``reduce("ica,jcb,ab -> ij")`` over the sites after ``start``, from the right end."""
        r = self._upload(np.ones((1, 1)))
        for t in reversed(self.sites[start + 1:]):
            cl, d, cr = (int(x) for x in t.shape)
            y = self._gemm(t, OP_NONE, r, OP_NONE, cl * d, cr, cr)                    # Y[(i, c), b]
            r = self._gemm(y, OP_NONE, t, OP_CONJ_TRANSPOSE, cl, cl, d * cr)          # right'[i, j]
        return r

    def norm2(self) -> float:
        g = self._left_environment(len(self.sites))
        return float(g.cpu().numpy()[0, 0].real)

    def _dressed_site(self, axis: int):
        """``Z[b, i, d] = sum_{a, c} left[a, b] t[a, i, c] right[c, d]`` for the site at ``axis``."""
        t = self.sites[axis]
        cl, d, cr = (int(x) for x in t.shape)
        g, r = self._left_environment(axis), self._right_environment(axis)
        w = self._gemm(t, OP_NONE, r, OP_NONE, cl * d, cr, cr)
        return self._gemm(g, OP_CONJ_TRANSPOSE, w, OP_NONE, cl, d * cr, cl), t

    def marginal(self, axis: int) -> np.ndarray:
        """``sum over the other modes of |psi|^2`` per grid point of mode ``axis`` (no grid-measure factors)."""
        torch = _torch()
        z, t = self._dressed_site(axis)
        cl, d, cr = (int(x) for x in t.shape)
        out = torch.empty(d, dtype=torch.float64, device=self._dev)
        _lib.call("qsv_tensor_axis_overlap", self.device, self._stream(), self._p(z), self._p(t), cl, d, cr, self._p(out))
        return out.cpu().numpy()

    def reduced_density(self, axis: int) -> np.ndarray:
        """Full ``(d, d)`` reduced density matrix (mps.py:176-190): environments on the device, the last small
        contraction on the host."""
        z, t = self._dressed_site(axis)
        host_t = t.cpu().numpy()
        return np.einsum("bid,bjd -> ij", z.cpu().numpy().reshape(host_t.shape), np.conj(host_t), optimize=True)

    # ---- gates ------------------------------------------------------------------------------------------------
    def apply_mode(self, operator: np.ndarray, mode: int) -> None:
        t = self.sites[mode]
        cl, d, cr = (int(x) for x in t.shape)
        op = self._keep(operator)
        if operator.ndim == 1:
            _lib.call("qsv_tensor_scale_axis", self.device, self._stream(), self._p(t), cl, d, cr, self._p(op))
            return
        out = self._empty(cl, int(operator.shape[0]), cr)
        _lib.call("qsv_tensor_apply_axis_dev", self.device, self._stream(), self._p(t), self._p(out), cl, d,
                  int(operator.shape[0]), cr, self._p(op))
        self.sites[mode] = out

    def _two_site(self, left: int):
        a, b = self.sites[left], self.sites[left + 1]
        cl, d, chi = (int(x) for x in a.shape)
        cr = int(b.shape[2])
        theta = self._gemm(a, OP_NONE, b, OP_NONE, cl * d, d * cr, chi, scratch="theta")
        return theta, cl, d, cr

    def _split(self, theta, rows: int, cols: int, *, max_bond_dim=np.inf, abs_err: float = 0, rel_err: float = 1e-12,
               rng_seed=None):
        """``tensor_svd`` (mps.py:52-97) of the device matrix ``theta``: exact SVD, or -- exactly when the reference does,
        ``max_bond_dim * 10 < min(rows, cols)`` -- the randomized range finder with the Gaussian test matrix drawn
        from ``rng_seed`` the way ``randomized_range_finder`` draws it (mps.py:14-15)."""
        full = min(rows, cols)
        capped = np.isfinite(max_bond_dim)
        cap = max(0, min(full, int(max_bond_dim))) if capped else full
        m1, m2 = self._scratch("m1", rows, max(cap, 1)), self._scratch("m2", max(cap, 1), cols)
        rank = C.c_uint64(0)
        if capped and max_bond_dim * 10 < full and cap >= 1:
            k = int(max_bond_dim)
            probes, power_iterations = k + 10, (7 if k < 0.1 * full else 4)
            s = np.empty(k, dtype=np.float64)
            # The first call goes without the Gaussian test matrix: under a loose tolerance the library's verified
            # low-rank route decides the split without reading it.  The reference draws it from default_rng(rng_seed)
            # (mps.py:14-15): a Generator handed down by the simulator (gates.py passes its own) is advanced by that draw,
            # so it is made in any case -- on a helper thread while the library works (both release the GIL) -- and only
            # the conversion and upload wait for the library to ask; a seed or None creates a generator nobody else sees,
            # and the draw itself waits too.
            draw = lambda: np.random.default_rng(rng_seed).normal(0, 1, size=(full, probes))
            pending = _DRAW_POOL.submit(draw) if isinstance(rng_seed, np.random.Generator) else None
            dev_omega = None
            try:
                for attempt in range(2):
                    _lib.call("qsv_tensor_rsvd_split", self.device, self._stream(), self._p(theta), rows, cols, k, probes,
                              power_iterations, self._p(dev_omega) if dev_omega is not None else None, float(abs_err),
                              float(rel_err), self._p(m1), self._p(m2), cap, C.byref(rank), s.ctypes.data_as(C.c_void_p))
                    if rank.value != _lib.RANK_NEEDS_OMEGA:
                        break
                    omega = pending.result() if pending is not None else draw()
                    pending = None
                    dev_omega = self._upload(np.asfortranarray(omega).T)      # the transpose's C order = column-major omega
            finally:
                if pending is not None:
                    pending.result()          # the caller's generator must have moved on before anyone else draws
            self.split_counts["randomized"] += 1
        else:
            s = np.empty(full, dtype=np.float64)
            _lib.call("qsv_tensor_svd_split", self.device, self._stream(), self._p(theta), rows, cols,
                      int(max_bond_dim) if capped else -1, float(abs_err), float(rel_err), self._p(m1), self._p(m2),
                      cap, C.byref(rank), s.ctypes.data_as(C.c_void_p))
            self.split_counts["exact"] += 1
        r = int(rank.value)
        self.last_singular_values = s
        # the library wrote compact (rows x r) and (r x cols) matrices at the start of the buffers; copy them out so that
        # the sites do not pin buffers sized for the untruncated rank (theta-sized when there is no cap)
        return (m1.view(-1)[: rows * r].view(rows, r).clone(), m2.view(-1)[: r * cols].view(r, cols).clone(), r)

    def _store_pair(self, left: int, m1, m2, cl: int, cr: int, r: int) -> None:
        self.sites[left] = m1.reshape(cl, self.d, r)
        self.sites[left + 1] = m2.reshape(r, self.d, cr)

    def apply_two_mode(self, plane: np.ndarray, mode0: int, mode1: int, **truncation) -> None:
        """``theta[a, j, l, b] *= plane[j, l]`` (legs in the order ``mode0, mode1``), then split: CZ (gates.py:151-163)."""
        left = min(mode0, mode1)
        if mode0 > mode1:
            plane = np.ascontiguousarray(plane.T)
        theta, cl, d, cr = self._two_site(left)
        dev_plane = self._keep(plane)
        _lib.call("qsv_tensor_plane_diag", self.device, self._stream(), self._p(theta), cl, d, cr, self._p(dev_plane))
        m1, m2, r = self._split(theta, cl * d, d * cr, **truncation)
        self._store_pair(left, m1, m2, cl, cr, r)

    def apply_two_mode_gather(self, cols: np.ndarray, weights: np.ndarray, mode0: int, mode1: int, **truncation) -> None:
        """Resample every ``(q_left, q_right)`` plane of theta with the row-sparse table, then split: BS, CX
        (gates.py:58-84,166-192) and SWAP (gates.py:48-55; the table is the transposition)."""
        if mode0 > mode1:
            raise ValueError("plane tables are built for (left, right) order")
        left = mode0
        theta, cl, d, cr = self._two_site(left)
        per_point = int(cols.shape[-1]) if cols.ndim > 1 else 1
        dev_cols = self._keep(cols, np.int32)
        dev_vals = self._keep(weights, np.complex128)
        mapped = self._scratch("mapped", cl * d, d * cr)
        _lib.call("qsv_tensor_plane_gather", self.device, self._stream(), self._p(theta), self._p(mapped), cl, d, cr,
                  per_point, self._p(dev_cols), self._p(dev_vals))
        m1, m2, r = self._split(mapped, cl * d, d * cr, **truncation)
        self._store_pair(left, m1, m2, cl, cr, r)

    def apply_plane_phase(self, grid: np.ndarray, strength: float, left: int, **truncation) -> None:
        """CZ with the phases ``exp(i strength q_j q_l)`` evaluated in the kernel (no ``(d, d)`` table)."""
        theta, cl, d, cr = self._two_site(left)
        _lib.call("qsv_tensor_plane_phase", self.device, self._stream(), self._p(theta), cl, d, cr,
                  self._p(self._keep(grid, np.float64)), float(strength))
        m1, m2, r = self._split(theta, cl * d, d * cr, **truncation)
        self._store_pair(left, m1, m2, cl, cr, r)

    def apply_plane_affine(self, grid: np.ndarray, coefficients, left: int, **truncation) -> None:
        """BS / CX: resample every plane at ``(a00 x + a01 y, a10 x + a11 y)``, bilinear, computed in the kernel."""
        theta, cl, d, cr = self._two_site(left)
        mapped = self._scratch("mapped", cl * d, d * cr)
        a = (C.c_double * 4)(*[float(v) for v in coefficients])
        _lib.call("qsv_tensor_plane_affine", self.device, self._stream(), self._p(theta), self._p(mapped), cl, d, cr,
                  self._p(self._keep(grid, np.float64)), a)
        m1, m2, r = self._split(mapped, cl * d, d * cr, **truncation)
        self._store_pair(left, m1, m2, cl, cr, r)

    def project(self, mode: int, level: int, scale: float) -> None:
        """Keep grid point ``level`` of ``mode`` (times ``scale``) and absorb the bond matrix into a neighbour, on the
        side the reference picks (gates.py:108-115): into the left neighbour iff the slice is at least as tall as wide
        (``np.argmax(mode.shape) == 0``) and there is a left neighbour."""
        t = self.sites[mode]
        cl, d, cr = (int(x) for x in t.shape)
        bond = self._empty(cl, cr)
        _lib.call("qsv_tensor_take_level", self.device, self._stream(), self._p(t), self._p(bond), cl, d, cr, int(level),
                  float(scale))
        if cl >= cr and mode != 0:
            nb = self.sites[mode - 1]
            ncl, nd, _ = (int(x) for x in nb.shape)
            self.sites[mode - 1] = self._gemm(nb, OP_NONE, bond, OP_NONE, ncl * nd, cr, cl).reshape(ncl, nd, cr)
        else:
            nb = self.sites[mode + 1]
            _, nd, ncr = (int(x) for x in nb.shape)
            self.sites[mode + 1] = self._gemm(bond, OP_NONE, nb, OP_NONE, cl, nd * ncr, cr).reshape(cl, nd, ncr)
        self.sites.pop(mode)

    def insert(self, mode: int, vec: np.ndarray, **truncation) -> None:
        """New mode at ``mode``: a free-standing site at either end, otherwise attached to the site now at ``mode``
        and split off again (gates.py:24-45)."""
        n = len(self.sites)
        if mode == 0 or mode == n:
            self.sites.insert(mode, self._upload(np.asarray(vec).reshape(1, -1, 1)))
            return
        t = self.sites[mode]
        cl, d, cr = (int(x) for x in t.shape)
        dev_vec = self._upload(vec)
        joined = self._scratch("theta", cl * d, d * cr)          # [a, i, (j, b)] = vec[i] * t[a, (j, b)]
        _lib.call("qsv_tensor_insert_axis", self.device, self._stream(), self._p(t), self._p(joined), cl, d, d * cr,
                  self._p(dev_vec))
        m1, m2, r = self._split(joined, cl * d, d * cr, **truncation)
        self.sites[mode] = m2.reshape(r, d, cr)
        self.sites.insert(mode, m1.reshape(cl, d, r))

    # ---- GKP layer (gkp_simulator) ----------------------------------------------------------------------------
    def _outer(self, p, q, x: int, y: int, z: int, w: int, swap_last: bool):
        out = self._scratch("theta", x * y, z * w)
        _lib.call("qsv_tensor_outer", self.device, self._stream(), self._p(p), self._p(q), self._p(out), x, y, z, w,
                  int(swap_last))
        return out

    def insert_bond_pair(self, mode: int, first: np.ndarray, second: np.ndarray, **truncation) -> None:
        """Insert two new modes at ``mode`` that share a bond: ``first`` is ``(d, chi)``, ``second`` ``(chi, d)`` (a GKP
        Bell pair has chi = 2).  At the ends of the chain they are simply attached; inside, each half is multiplied into
        its neighbour and split off again (``InsertBell.apply``, gkp_simulator/insert_bell.py:60-96)."""
        d, chi = first.shape
        if mode == 0 or mode == len(self.sites):
            pair = [self._upload(first.reshape(1, d, chi)), self._upload(second.reshape(chi, d, 1))]
            self.sites[mode:mode] = pair
            return
        t1, t2 = self.sites[mode - 1], self.sites[mode]
        a, _, b = (int(v) for v in t1.shape)
        c = int(t2.shape[2])
        # [(a, i), k, b, dd] = t1[(a, i), b] * first[k, dd]  ->  rows (a, i), columns (k, b, dd)
        joined = self._outer(t1, self._keep(first), a * d, d, b, chi, swap_last=False)       # read only: a cached copy will do
        m1, m2, r1 = self._split(joined, a * d, d * b * chi, **truncation)
        new_t1, new_first = m1.reshape(a, d, r1), m2.reshape(r1, d, b * chi)
        # [b, dd, l, (j, c)] = second[dd, l] * t2[b, (j, c)]  ->  rows (b, dd, l), columns (j, c)
        joined = self._outer(t2, self._keep(second), b, chi, d * c, d, swap_last=True)
        m1, m2, r2 = self._split(joined, b * chi * d, d * c, **truncation)
        new_second, new_t2 = m1.reshape(b * chi, d, r2), m2.reshape(r2, d, c)
        self.sites[mode - 1:mode + 1] = [new_t1, new_first, new_second, new_t2]

    def operator_string_coefficients(self, operators: list[np.ndarray]) -> np.ndarray:
        """``C[i_0, ..., i_{m-1}] = <psi| O_{i_0} (x) ... (x) O_{i_{m-1}} |psi>`` (no grid-measure factors) for every
        string over the given single-mode operators: the contraction loop of ``full_logical_density_mps``
        (gkp_simulator/utils.py:82-90) as a depth-first walk that shares the environments of common prefixes."""
        n, k = len(self.sites), len(operators)
        dressed = []
        for t in self.sites:
            cl, d, cr = (int(v) for v in t.shape)
            row = []
            for op in operators:
                out = self._empty(cl, d, cr)
                _lib.call("qsv_tensor_apply_axis_dev", self.device, self._stream(), self._p(t), self._p(out), cl, d, d, cr,
                          self._p(self._keep(op)))
                row.append(out)
            dressed.append(row)
        result = np.zeros((k,) * n, dtype=np.complex128)

        def walk(level: int, g, prefix: tuple):
            if level == n:
                result[prefix] = np.conj(g.cpu().numpy()[0, 0])
                return
            bra = self.sites[level]
            cl, d, cr = (int(v) for v in bra.shape)
            for i in range(k):
                x = self._gemm(g, OP_CONJ_TRANSPOSE, dressed[level][i], OP_NONE, cl, d * cr, cl)
                walk(level + 1, self._gemm(x, OP_CONJ_TRANSPOSE, bra, OP_NONE, cr, cr, cl * d), prefix + (i,))

        walk(0, self._upload(np.ones((1, 1))), ())
        return result
