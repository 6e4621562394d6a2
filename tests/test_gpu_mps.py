"""GPU parity of the matrix-product-state register against the reference's own MPS run with truncation.

``tests/golden/cv_mps.npz`` holds, for three truncation settings, what the reference produced gate by gate for
``fixture_io.cv_mps_program``: site shapes (i.e. the kept bond dimensions), norms, checkpoints of the contracted state,
measurement records and reduced densities.  The ``cap5`` setting drives the reference onto its randomized-SVD branch for
the interior bonds, with the simulator's seeded generator supplying the test matrices.
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
from fixture_io import cv_mps_program  # noqa: E402

from quantum_computations_amd.cv_simulator import gates as CV
from quantum_computations_amd.cv_simulator.gate_abc import MeasurementResult
from quantum_computations_amd.cv_simulator.mps import MPS, tensor_svd
from quantum_computations_amd.cv_simulator.states import State


def maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))


def _cases(golden, kind=None):
    g = golden["cv_mps"]
    cases = json.loads(str(g["cases"]))
    return g, [c for c in cases if (c.get("kind") == kind if kind else "label" in c)]


def test_tensor_svd_matches_reference(golden):
    g, cases = _cases(golden, "tensor_svd")
    assert len(cases) == 4
    for c in cases:
        t = g[f"svd_in_{c['index']}"]
        m1, m2 = tensor_svd(t, c["left"], c["right"], **c["options"])
        assert m1.shape[-1] == m2.shape[0] == c["rank"]
        assert m1.shape[:-1] == tuple(t.shape[i] for i in c["left"])
        assert m2.shape[1:] == tuple(t.shape[i] for i in c["right"])
        assert maxdiff(np.tensordot(m1, m2, axes=1), g[f"svd_product_{c['index']}"]) < 1e-10
    with pytest.raises(IndexError):
        tensor_svd(np.zeros((2, 2, 2)), [0], [1])


@pytest.mark.parametrize("label,tol", [("rel1e-6", 1e-9), ("abs1e-3", 1e-9), ("cap5", 1e-7)])
def test_truncated_mps_program_matches_reference(golden, label, tol):
    g, cases = _cases(golden)
    case = next(c for c in cases if c["label"] == label)
    mps = MPS(g["qs"], [], layout="sites")
    rng = np.random.default_rng(5)
    results = []
    for position, gate in enumerate(cv_mps_program(CV, State, case["options"])):
        out = gate.apply(mps, rng=rng)
        if isinstance(out, MeasurementResult):
            results.append([position, out.result, out.probability])
        assert [list(s) for s in mps.shape()] == case["shapes"][position], (position, gate)
        assert abs(mps.norm() - g[f"{label}_norms"][position]) < tol, (position, gate)
        key = f"{label}_state_{position}"
        if key in g:
            assert maxdiff(mps.contract(), g[key]) < tol, (position, gate)
    assert np.allclose(np.array(results), g[f"{label}_results"], rtol=0, atol=tol)
    assert maxdiff(mps.marginal(1), g[f"{label}_marginal"]) < tol
    assert maxdiff(mps.partial_density_mps(0), g[f"{label}_rho0"]) < tol
    # only the capped run reaches max_bond_dim * 10 < min(shape), on its interior bonds
    assert (mps.reg.split_counts["randomized"] > 0) == (label == "cap5")


def test_site_layout_agrees_with_dense_layout_when_nothing_is_truncated(golden):
    g = golden["cv_mps"]
    qs = g["qs"]
    exact = {"rel_err": 0.0, "abs_err": 0.0}
    sites, dense = MPS(qs, [], layout="sites"), MPS(qs, [], layout="dense")
    for gate in cv_mps_program(CV, State, exact)[:14]:
        gate.apply(sites, rng=None)
        gate.apply(dense, rng=None)
    assert maxdiff(sites.contract(), dense.contract()) < 1e-10
    assert abs(sites.norm() - dense.norm()) < 1e-10
    assert maxdiff(sites.marginal(2), dense.marginal(2)) < 1e-10
    clone = sites.copy()
    CV.F(0).apply(clone)
    assert maxdiff(sites.contract(), dense.contract()) < 1e-10      # the copy owns its tensors


def test_randomized_split_on_a_low_rank_matrix():
    """rank-6 matrix, max_bond_dim = 8 < min(shape) / 10: the randomized branch must recover it to rounding."""
    rng = np.random.default_rng(2)
    a = (rng.standard_normal((180, 6)) + 1j * rng.standard_normal((180, 6))) @ \
        (rng.standard_normal((6, 130)) + 1j * rng.standard_normal((6, 130)))
    for matrix in (a, np.ascontiguousarray(a.T)):
        t = matrix.reshape(matrix.shape[0], 1, 1, matrix.shape[1])
        m1, m2 = tensor_svd(t, [0, 1], [2, 3], max_bond_dim=8, rng_seed=11)
        assert m1.shape[-1] == 6        # the tail of the 8 kept values is below rel_err * sum
        assert maxdiff(np.tensordot(m1, m2, axes=1).reshape(matrix.shape), matrix) < 1e-9 * np.abs(matrix).max()


def test_randomized_split_asks_for_its_test_matrix_only_when_it_reads_it():
    """``qsv_tensor_rsvd_split`` without a test matrix (``dev_omega = NULL``): a split the verified low-rank route decides
    (loose tolerance, more than 64 probes asked for) is answered at once; one it cannot decide -- or a panel of at most 64
    probes, which goes straight to the caller's matrix -- reports ``QSV_RANK_NEEDS_OMEGA`` and computes nothing; the second
    call with the matrix gives what a call with the matrix from the start gives.  A ``Generator`` handed down as
    ``rng_seed`` is advanced by the draw in either case (the reference's ``default_rng(generator)`` is the generator)."""
    import ctypes as C

    import torch

    from quantum_computations_amd import _lib
    rng = np.random.default_rng(5)
    rows, cols, k, probes = 1200, 900, 70, 80
    u, _ = np.linalg.qr(rng.standard_normal((rows, cols)) + 1j * rng.standard_normal((rows, cols)))
    v, _ = np.linalg.qr(rng.standard_normal((cols, cols)) + 1j * rng.standard_normal((cols, cols)))
    low = (u * np.exp(-np.arange(cols) / 2.0)) @ v.conj().T                   # numerical rank ~ 30 at rel_err = 1e-3
    flat = (u * np.linspace(1.0, 0.5, cols)) @ v.conj().T                     # nothing to truncate: no verified answer
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    omega = np.random.default_rng(9).normal(0, 1, size=(cols, probes))
    dev_omega = torch.from_numpy(np.ascontiguousarray(np.asfortranarray(omega).T, dtype=np.complex128)).cuda()

    def split(matrix, with_omega, rel_err, n_probes=probes, keep=k):
        dev = torch.from_numpy(matrix).cuda()
        m1 = torch.empty(rows * keep, dtype=torch.complex128, device="cuda")
        m2 = torch.empty(keep * cols, dtype=torch.complex128, device="cuda")
        rank = C.c_uint64(0)
        s = np.empty(keep)
        _lib.call("qsv_tensor_rsvd_split", 0, stream, C.c_void_p(dev.data_ptr()), rows, cols, keep, n_probes, 4,
                  C.c_void_p(dev_omega.data_ptr()) if with_omega else None, 0.0, rel_err, C.c_void_p(m1.data_ptr()),
                  C.c_void_p(m2.data_ptr()), keep, C.byref(rank), s.ctypes.data_as(C.c_void_p))
        torch.cuda.synchronize()
        r = int(rank.value)
        if r == _lib.RANK_NEEDS_OMEGA:
            return r, None
        return r, (m1[: rows * r].view(rows, r) @ m2[: r * cols].view(r, cols)).cpu().numpy()

    r, product = split(low, False, 1e-3)
    assert r != _lib.RANK_NEEDS_OMEGA and 10 < r < k and maxdiff(product, low) < 2e-3 * np.abs(low).max() * r
    r, product = split(flat, False, 1e-3)
    assert r == _lib.RANK_NEEDS_OMEGA and product is None
    r_again, with_matrix = split(flat, True, 1e-3)
    r_direct, direct = split(flat, True, 1e-3)
    assert r_again == r_direct == k and np.array_equal(with_matrix, direct)
    # the host side: tensor_svd with a Generator as rng_seed leaves it where the reference's draw would
    t = low.reshape(rows, 1, 1, cols)
    gen, twin = np.random.default_rng(21), np.random.default_rng(21)
    tensor_svd(t, [0, 1], [2, 3], max_bond_dim=k, rel_err=1e-3, rng_seed=gen)
    twin.normal(0, 1, size=(cols, k + 10))
    assert gen.random() == twin.random()


def test_library_qr_path_of_the_randomized_split_still_works():
    """``QSV_RSVD=rocsolver`` selects rocSOLVER's Householder QR / gesvd instead of the fused panel kernels; the choice is
    read once per process, so the check runs in a child process."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np\n"
        "from quantum_computations_amd.cv_simulator.mps import tensor_svd\n"
        "rng = np.random.default_rng(4)\n"
        "a = (rng.standard_normal((150, 5)) + 1j * rng.standard_normal((150, 5))) @ "
        "(rng.standard_normal((5, 170)) + 1j * rng.standard_normal((5, 170)))\n"
        "m1, m2 = tensor_svd(a.reshape(150, 1, 1, 170), [0, 1], [2, 3], max_bond_dim=7, rng_seed=3)\n"
        "assert m1.shape[-1] == 5, m1.shape\n"
        "err = np.max(np.abs(np.tensordot(m1, m2, axes=1).reshape(a.shape) - a))\n"
        "assert err < 1e-9 * np.abs(a).max(), err\n"
        "print('ok')\n")
    env = dict(os.environ, QSV_RSVD="rocsolver")
    repo = str(Path(__file__).resolve().parent.parent)
    done = subprocess.run([sys.executable, "-c", code], cwd=repo, env=env, capture_output=True, text=True, timeout=300)
    assert done.returncode == 0 and "ok" in done.stdout, done.stderr[-2000:]


@pytest.mark.parametrize("n,m,l", [(100, 70, 5), (257, 129, 16), (1000, 333, 26), (2049, 515, 42), (130, 1027, 64),
                                   (64, 32, 1), (4096, 4096, 33), (700, 450, 110), (333, 900, 200)])
def test_mfma_tall_skinny_products(n, m, l):
    """``qsv_tensor_skinny_gemm``: Y = op(A) Q for op in {A, A^H, A^T, conj A} (column-major) on the f64 matrix cores,
    ragged shapes, with and without the split of the k range."""
    import ctypes as C

    import torch

    from quantum_computations_amd import _lib
    rng = np.random.default_rng(n + m + l)
    a = rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m))
    dev_a = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()             # row-major A^T == column-major A
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    forms = {0: lambda x: x, 1: lambda x: x.conj().T, 2: lambda x: x.T, 3: lambda x: x.conj()}
    for op, form in forms.items():
        rows_q, rows_y = (n, m) if op in (1, 2) else (m, n)
        q = rng.standard_normal((rows_q, l)) + 1j * rng.standard_normal((rows_q, l))
        dev_q = torch.from_numpy(np.ascontiguousarray(q.T)).cuda()
        dev_y = torch.empty(l, rows_y, dtype=torch.complex128, device="cuda")
        _lib.call("qsv_tensor_skinny_gemm", 0, stream, op, n, m, l, C.c_void_p(dev_a.data_ptr()),
                  C.c_void_p(dev_q.data_ptr()), C.c_void_p(dev_y.data_ptr()))
        torch.cuda.synchronize()
        want = form(a) @ q
        assert maxdiff(dev_y.cpu().numpy().T, want) < 1e-12 * np.abs(want).max(), op
    with pytest.raises(ValueError):
        _lib.call("qsv_tensor_skinny_gemm", 0, stream, 0, n, m, 257, C.c_void_p(dev_a.data_ptr()),
                  C.c_void_p(dev_a.data_ptr()), C.c_void_p(dev_a.data_ptr()))


@pytest.mark.parametrize("m,n,k", [(1, 1, 1000), (2, 2, 2000), (1, 7, 333), (8, 8, 4097), (3, 2, 256), (2, 2, 255), (9, 9, 500)])
def test_tensor_gemm_with_few_outputs_and_a_long_inner_dimension(m, n, k):
    """``qsv_tensor_gemm`` on the shapes of the MPS environment matrices (``site_register.py``: chi x chi outputs over an
    inner dimension of chi d): at most 64 outputs over k >= 256 take the reduction kernel ``k_gemm_few_outputs``, the
    others rocBLAS -- every combination of plain / transposed / conjugate-transposed operands against NumPy."""
    import ctypes as C

    import torch

    from quantum_computations_amd import _lib
    rng = np.random.default_rng(m * 100 + n * 10 + k)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    form = {0: lambda x: x, 1: lambda x: x.T, 2: lambda x: x.conj().T}
    for op_a in (0, 1, 2):
        for op_b in (0, 1, 2):
            shape_a = (m, k) if op_a == 0 else (k, m)
            shape_b = (k, n) if op_b == 0 else (n, k)
            a = rng.standard_normal(shape_a) + 1j * rng.standard_normal(shape_a)
            b = rng.standard_normal(shape_b) + 1j * rng.standard_normal(shape_b)
            dev_a, dev_b = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
            dev_c = torch.empty(m, n, dtype=torch.complex128, device="cuda")
            _lib.call("qsv_tensor_gemm", 0, stream, op_a, op_b, m, n, k, C.c_void_p(dev_a.data_ptr()),
                      C.c_void_p(dev_b.data_ptr()), C.c_void_p(dev_c.data_ptr()))
            torch.cuda.synchronize()
            want = form[op_a](a) @ form[op_b](b)
            assert maxdiff(dev_c.cpu().numpy(), want) < 1e-13 * k, (op_a, op_b)


@pytest.mark.parametrize("shape", [(700, 520), (300, 1100), (1024, 1024)])
@pytest.mark.parametrize("options", [{"rel_err": 1e-2}, {"rel_err": 1e-3, "max_bond_dim": 40}, {"abs_err": 5e-3, "rel_err": 0.0},
                                     {"rel_err": 1e-5}])
def test_exact_split_shortcuts_keep_the_reference_rank(shape, options):
    """Under loose tolerances the exact branch of ``tensor_svd`` may be served by the verified low-rank route or by
    rocSOLVER's Gram-based ``zgesdd`` instead of ``zgesvd`` (see qsvg_svd_split): the kept rank must be the one the
    truncation rule gives on LAPACK's spectrum, and the product must match the exactly truncated one."""
    from oracle import mps_oracle as MO
    rng = np.random.default_rng(shape[0] + shape[1])
    rows, cols = shape
    full = min(rows, cols)
    u, _ = np.linalg.qr(rng.standard_normal((rows, full)) + 1j * rng.standard_normal((rows, full)))
    v, _ = np.linalg.qr(rng.standard_normal((cols, full)) + 1j * rng.standard_normal((cols, full)))
    spectrum = np.exp(-np.arange(full) / 6.0) + 1e-9 * rng.random(full)       # graded, with a noise floor
    a = (u * spectrum) @ v.conj().T
    want1, want2 = MO.split(a, **options)
    m1, m2 = tensor_svd(a.reshape(rows, 1, 1, cols), [0, 1], [2, 3], **options)
    assert m1.shape[-1] == want1.shape[1]
    got = np.tensordot(m1, m2, axes=1).reshape(rows, cols)
    assert maxdiff(got, want1 @ want2) < 1e-9


@pytest.mark.parametrize("rows,cols,cap", [(1500, 1300, 70), (1000, 2100, 90), (2600, 2600, 150)])
def test_randomized_split_with_wide_panels(rows, cols, cap):
    """max_bond_dim > 54 means more than 64 probe columns: the panels are orthonormalised in 64-column blocks (block
    Gram-Schmidt around CholeskyQR3) and the projected factor goes to the library SVD.  Same random stream as the
    reference's randomized branch, so the product must agree with the CPU restatement."""
    from oracle import mps_oracle as MO
    rng = np.random.default_rng(rows + cap)
    full = min(rows, cols)
    assert cap * 10 < full
    u, _ = np.linalg.qr(rng.standard_normal((rows, full)) + 1j * rng.standard_normal((rows, full)))
    v, _ = np.linalg.qr(rng.standard_normal((cols, full)) + 1j * rng.standard_normal((cols, full)))
    a = (u * np.exp(-np.arange(full) / 9.0)) @ v.conj().T
    want1, want2 = MO.split(a, max_bond_dim=cap, rng_seed=17)
    m1, m2 = tensor_svd(a.reshape(rows, 1, 1, cols), [0, 1], [2, 3], max_bond_dim=cap, rng_seed=17)
    assert m1.shape[-1] == want1.shape[1]
    assert maxdiff(np.tensordot(m1, m2, axes=1).reshape(rows, cols), want1 @ want2) < 1e-9


@pytest.mark.parametrize("rows,cols,rank,decay", [(1400, 1100, 12, 0.7), (1000, 2000, 45, 0.25), (1800, 1800, 100, 0.12),
                                                 (1300, 1500, 230, 0.05), (1200, 1200, 700, 0.02)])
def test_exact_split_at_the_default_tolerance(rows, cols, rank, decay):
    """The reference's default ``rel_err = 1e-12`` on numerically low-rank theta of the sizes its GKP runs produce
    (1000..2000 on a side): the verified route evaluates what its projection misses entry by entry (resolution
    ~1e-15 ||theta||, not the ~1e-8 of a difference of norms) and widens its probe panel 64 -> 128 -> 256 before the
    library SVD gets the matrix.  Whatever route answers, the kept rank must be the one the truncation rule gives on
    LAPACK's spectrum and the product must match the exactly truncated one -- including spectra that do NOT fit 256
    probes (the last case: the library decides)."""
    import time
    from oracle import mps_oracle as MO
    rng = np.random.default_rng(rows + rank)
    full = min(rows, cols)
    u, _ = np.linalg.qr(rng.standard_normal((rows, full)) + 1j * rng.standard_normal((rows, full)))
    v, _ = np.linalg.qr(rng.standard_normal((cols, full)) + 1j * rng.standard_normal((cols, full)))
    spectrum = np.exp(-decay * np.arange(full))
    spectrum[rank:] = 1e-16 * rng.random(full - rank)                       # exactly low rank up to rounding dust
    a = (u * spectrum) @ v.conj().T
    want1, want2 = MO.split(a)                                               # rel_err = 1e-12, no cap, no abs_err
    tensor_svd(a.reshape(rows, 1, 1, cols), [0, 1], [2, 3])                  # warm-up: library loads, probe upload
    t0 = time.perf_counter()
    m1, m2 = tensor_svd(a.reshape(rows, 1, 1, cols), [0, 1], [2, 3])
    seconds = time.perf_counter() - t0
    assert m1.shape[-1] == want1.shape[1], (m1.shape[-1], want1.shape[1])
    got = np.tensordot(m1, m2, axes=1).reshape(rows, cols)
    assert maxdiff(got, want1 @ want2) < 1e-11
    print(f"split {rows}x{cols} rank {rank}: {seconds * 1e3:.1f} ms")
    # tens of milliseconds through the verified low-rank route, ~0.2 s through the Jacobi sweeps over the whole matrix
    # (rank 700 does not fit 256 probes); rocSOLVER's zgesvd takes 1-14 s on these
    assert seconds < (2.0 if rank <= 230 else 1.0), seconds


@pytest.mark.parametrize("rows,cols", [(35, 7), (61, 61), (120, 333), (333, 120), (401, 400), (640, 900)])
@pytest.mark.parametrize("options", [{}, {"rel_err": 0.0}, {"max_bond_dim": "a third"}, {"rel_err": 1e-7, "abs_err": 1e-9}])
def test_exact_split_of_full_rank_matrices(rows, cols, options):
    """Round 3: a theta that is NOT numerically low-rank takes one-sided Jacobi sweeps over the whole matrix, seeded with
    the factors of rocSOLVER's Gram-based ``zgesdd`` (``jacobi_full_split`` in csrc/qsv_decomp.hip) -- tall, wide, square
    and odd sizes, spectra graded over eight decades with rounding dust below, every truncation option: the kept rank
    must be the one the rule gives on LAPACK's spectrum, the singular values must agree with LAPACK's, and the product
    must match the exactly truncated one."""
    from oracle import mps_oracle as MO
    rng = np.random.default_rng(rows * 1000 + cols)
    full = min(rows, cols)
    if "max_bond_dim" in options:
        options = {"max_bond_dim": full // 3 + 1}          # a cap that keeps tensor_svd on its exact branch (cap * 10 >= full)
    u, _ = np.linalg.qr(rng.standard_normal((rows, full)) + 1j * rng.standard_normal((rows, full)))
    v, _ = np.linalg.qr(rng.standard_normal((cols, full)) + 1j * rng.standard_normal((cols, full)))
    spectrum = np.exp(-18.0 * np.arange(full) / full)
    dust = full // 5
    if dust:
        spectrum[-dust:] = 1e-17 * rng.random(dust)
    a = (u * spectrum) @ v.conj().T
    want1, want2 = MO.split(a, **options)
    m1, m2 = tensor_svd(a.reshape(rows, 1, 1, cols), [0, 1], [2, 3], **options)
    if options.get("rel_err", 1.0) == 0.0 and "abs_err" not in options:
        # nothing may be truncated: LAPACK keeps every value above exactly zero, dust included
        assert m1.shape[-1] == want1.shape[1] == full
    else:
        assert m1.shape[-1] == want1.shape[1], (m1.shape[-1], want1.shape[1])
    got = np.tensordot(m1, m2, axes=1).reshape(rows, cols)
    assert maxdiff(got, want1 @ want2) < 1e-13
    # the factors carry sqrt(S) each: column norms of m1 squared are the singular values
    kept = m1.shape[-1]
    values = np.sum(np.abs(m1.reshape(rows, kept)) ** 2, axis=0)
    lapack = np.linalg.svd(a, compute_uv=False)[:kept]
    resolved = lapack > 1e-12
    assert np.max(np.abs(values[resolved] - lapack[resolved]) / lapack[resolved]) < 1e-9
