"""GPU parity of the CV path: gate classes -> C ABI -> mode kernels, against what the reference's cv_simulator
produced with truncation disabled (tests/golden/cv_operators.npz, cv_extra.npz) and against the tensordot oracle."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cv_oracle as CO
from quantum_computations_amd.cv_simulator import fock
from quantum_computations_amd.cv_simulator import gates as CV
from quantum_computations_amd.cv_simulator.mps import MPS
from quantum_computations_amd.cv_simulator.simulator import Simulator
from quantum_computations_amd.cv_simulator.states import State, squeezed_vac, vacuum
from quantum_computations_amd.device import QuditState, tensor_apply_axis

TOL = 1e-11


def maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))


def dense_mps(domain, psi):
    """Register holding the dense tensor ``psi`` (bypasses the site contraction of the constructor)."""
    return MPS._wrap(domain, QuditState.from_numpy(psi))


def test_gate_sequence_matches_reference(golden):
    g = golden["cv_operators"]
    reg = dense_mps(g["qs12"], g["cv_seq_in"])
    seq = [CV.F(0), CV.CZ(0, 1, 0.6), CV.X(1, 0.5), CV.P(2, 0.3), CV.CZ(2, 1, 0.4, dagger=True), CV.Z(0, 1.1),
           CV.F(2, dagger=True)]
    Simulator(seq).run(reg)
    assert maxdiff(reg.contract(), g["cv_seq_out"]) < TOL
    assert abs(reg.norm() - np.sqrt(np.sum(np.abs(g["cv_seq_out"]) ** 2) * reg.diff ** 3)) < TOL


def test_two_mode_gates_on_basis_states_reproduce_reference_operators(golden):
    g = golden["cv_operators"]
    qs = g["qs8"]
    d = len(qs)
    makers = {
        "cv2_CZ_0.8": lambda: CV.CZ(0, 1, 0.8), "cv2_CZ_0.8_dag": lambda: CV.CZ(0, 1, 0.8, dagger=True),
        "cv2_BS_pi4": lambda: CV.BS(0, 1, np.pi / 4), "cv2_BS_0.3_rev": lambda: CV.BS(1, 0, 0.3),
        "cv2_BS_0.3_dag": lambda: CV.BS(0, 1, 0.3, dagger=True), "cv2_CX_1.0": lambda: CV.CX(0, 1, 1.0),
        "cv2_CX_1.0_rev": lambda: CV.CX(1, 0, 1.0), "cv2_SWAP": lambda: CV.SWAP(0, 1),
    }
    rng = np.random.default_rng(3)
    psi = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    for key, make in makers.items():
        reg = dense_mps(qs, psi)
        make().apply(reg, rng=None)
        want = (g[key] @ psi.reshape(-1)).reshape(d, d)
        assert maxdiff(reg.contract(), want) < 1e-9 if key == "cv2_SWAP" else maxdiff(reg.contract(), want) < TOL, key
    # the same plane maps embedded in a 3-mode register, on both neighbour pairs
    psi3 = rng.standard_normal((d, d, d)) + 1j * rng.standard_normal((d, d, d))
    for pair, axes in [((0, 1), (0, 1)), ((1, 2), (1, 2)), ((2, 1), (1, 2))]:
        reg = dense_mps(qs, psi3)
        CV.BS(*pair, 0.3).apply(reg)
        op = g["cv2_BS_0.3_rev"] if pair[0] > pair[1] else None
        if op is None:
            # BS(i, i+1, 0.3) is the inverse rotation of the "rev" fixture: build it from the dagger fixture's adjoint
            c, s = np.cos(0.3), np.sin(0.3)
            x, y = np.meshgrid(qs, qs, indexing="ij")
            from quantum_computations_amd.cv_simulator.utils import plane_resample_table
            cols, vals = plane_resample_table(qs, c * x + s * y, -s * x + c * y)
            op = np.zeros((d * d, d * d), dtype=complex)
            for row in range(d * d):
                for cc, vv in zip(cols[row], vals[row]):
                    if cc >= 0:
                        op[row, cc] += vv
        want = CO.apply_two_axes(psi3, op, *axes)
        assert maxdiff(reg.contract(), want) < TOL, pair


def test_homodyne_measurements_match_reference(golden):
    g = golden["cv_extra"]
    qs = g["qs16"]
    makers = {"Mq": lambda i, r: CV.Mq(i, r), "Mp": lambda i, r: CV.Mp(i, r),
              "Hom0.7": lambda i, r: CV.Homodyne(i, 0.7, r), "Hompi": lambda i, r: CV.Homodyne(i, np.pi, r)}
    for case in golden.cases("cv_extra"):
        if case["kind"] != "measure":
            continue
        reg = dense_mps(qs, g[f"meas_in_{case['n_modes']}"])
        res = makers[case["gate"]](case["index"], case["forced"]).apply(reg, rng=None)
        assert abs(res.result - case["result"]) < 1e-12, case
        assert abs(res.probability - case["probability"]) < 1e-10, case
        assert len(reg) == case["n_modes"] - 1
        assert maxdiff(reg.contract(), g[case["key"]]) < 1e-10, case


def test_sampled_measurement_uses_the_simulator_rng(golden):
    g = golden["cv_extra"]
    qs = g["qs16"]
    outcomes = []
    for _ in range(2):
        reg = dense_mps(qs, g["meas_in_3"])
        sim = Simulator([CV.Mq(1), CV.Mp(0)], rng_seed=11)
        sim.run(reg)
        outcomes.append([(r.result, r.probability) for r in sim.results])
        assert len(reg) == 1
    assert outcomes[0] == outcomes[1] and len(outcomes[0]) == 2
    # the last mode: the reference returns the bare value and leaves the register alone
    reg = dense_mps(qs, g["meas_in_2"])
    Simulator([CV.Mq(0, 0.1)]).run(reg)
    out = CV.Mq(0, 0.1).apply(reg, rng=None)
    assert isinstance(out, float) and len(reg) == 1


def test_simulator_run_with_inserts_matches_reference(golden):
    g = golden["cv_extra"]
    qs = g["qs16"]
    gates = [CV.Insert(0, State.VACUUM), CV.Insert(1, State.GKP_PLUS, gkp_epsilon=0.3),
             CV.Insert(1, State.GKP_ZERO, gkp_epsilon=0.3), CV.X(0, 0.4), CV.CZ(0, 1, 0.5), CV.F(2), CV.BS(1, 2, 0.6),
             CV.P(1, 0.2), CV.CX(1, 0, 0.5), CV.SWAP(0, 1), CV.Homodyne(2, 0.4, 0.8), CV.D(0, [0.3, -0.2]),
             CV.Mp(0, -0.5)]
    sim = Simulator(gates, rng_seed=3, svd_options={"rel_err": 0.0, "typo": 1})
    out = sim.run(MPS(qs, []))
    assert len(out) == 1
    assert maxdiff(out.contract(), g["sim_out"]) < 1e-9
    got = np.array([[r.result, r.probability] for r in sim.results])
    assert maxdiff(got, g["sim_results"]) < 1e-9


def test_mps_facade(golden):
    g = golden["cv_extra"]
    qs = g["qs16"]
    psi = g["meas_in_3"]
    reg = dense_mps(qs, psi)
    assert len(reg) == 3 and abs(reg.norm() - 1.0) < 1e-12
    rho = reg.partial_density_mps(1)
    m = np.moveaxis(psi, 1, 0).reshape(16, -1)
    assert maxdiff(rho, (m @ m.conj().T) * reg.diff ** 2) < 1e-13
    assert maxdiff(reg.marginal(1), np.real(np.diagonal(rho))) < 1e-13
    clone = reg.copy()
    CV.Z(0, 0.3).apply(reg)
    assert maxdiff(clone.contract(), psi) == 0.0 and abs(MPS.fidelity(clone, clone) - 1.0) < 1e-12
    # constructor from site tensors (vectors and (chi_l, d, chi_r) sites)
    a, b = State.VACUUM.eval(qs), State.GKP_ZERO.eval(qs, 0.3)
    prod = MPS(qs, [a, b])
    assert prod.layout == "sites"                      # the reference's data structure is the default
    assert maxdiff(prod.contract(), np.multiply.outer(a, b)) < 1e-15
    # ... and behaves like the reference's list of site tensors
    assert prod[0].shape == (1, 16, 1) and [t.shape for t in prod] == [(1, 16, 1), (1, 16, 1)]
    assert [t.shape for t in prod[0:2]] == [t.shape for t in prod.tensors]
    prod[1] = (2 * b).reshape(1, -1, 1)
    assert maxdiff(prod.contract(), 2 * np.multiply.outer(a, b)) < 1e-15
    with pytest.raises(AttributeError):
        MPS(qs, [a, b], layout="dense")[0]
    with pytest.raises(ValueError):
        MPS(qs, [np.ones((2, 16, 1))])
    with pytest.raises(TypeError):
        MPS(list(qs), [a])
    with pytest.raises(IndexError):
        CV.Insert(5, State.VACUUM).apply(prod)


def test_squeezing_and_phase_on_the_grid():
    qs = np.linspace(-12, 12, 192)
    reg = MPS(qs, [vacuum(qs)])
    CV.S(0, 0.4).apply(reg)
    assert maxdiff(reg.contract(), squeezed_vac(qs, 0.4)) < 1e-9      # convention check (parity unpinned)
    CV.S(0, 0.4, dagger=True).apply(reg)
    assert maxdiff(reg.contract(), vacuum(qs)) < 1e-9
    # the vacuum is invariant under phase rotation up to a global phase (the reference's kernel, utils.py:33-34,
    # omits the zero-point factor, so only the ray is checked)
    CV.Phase(0, 0.7).apply(reg)
    overlap = np.vdot(vacuum(qs), reg.contract()) * reg.diff
    assert abs(abs(overlap) - 1.0) < 1e-8 and abs(reg.norm() - 1.0) < 1e-8


@pytest.mark.parametrize("L,d_in,d_out,R", [(3, 40, 40, 100), (1, 64, 48, 7), (5, 1000, 1000, 2), (2, 130, 130, 70),
                                          # grids large enough for the rocBLAS route (qsv_gemm.hip): R > 1, R == 1, rectangular
                                          (20, 256, 256, 20), (4096, 128, 128, 1), (6, 200, 72, 300),
                                          # thin fibres (bonds of 1-2 on d = 1000 grids): a wave per output row, k_axis_rows
                                          (1, 1000, 1000, 1), (1, 1000, 1000, 2), (2, 1000, 1000, 1), (3, 257, 129, 5),
                                          (2, 128, 200, 8), (1, 130, 1, 3)])
def test_tensor_apply_axis_on_mps_sites(L, d_in, d_out, R):
    """out[l, :, r] = M @ in[l, :, r] on a raw (chi_l, d, chi_r) site, as utils.py:15-16 does with tensordot."""
    import torch
    rng = np.random.default_rng(L + d_in)
    site = rng.standard_normal((L, d_in, R)) + 1j * rng.standard_normal((L, d_in, R))
    m = rng.standard_normal((d_out, d_in)) + 1j * rng.standard_normal((d_out, d_in))
    t_in = torch.from_numpy(site).cuda()
    t_out = torch.empty((L, d_out, R), dtype=torch.complex128, device="cuda")
    torch.cuda.synchronize()
    tensor_apply_axis(t_in.data_ptr(), t_out.data_ptr(), L, d_in, d_out, R, m,
                      stream=torch.cuda.current_stream().cuda_stream)
    want = CO.apply_axis(site, m, 1)
    assert maxdiff(t_out.cpu().numpy(), want) < 1e-9 * np.sqrt(d_in)
    # operator resident on the device (no upload, asynchronous)
    t_m = torch.from_numpy(m).cuda()
    t_out.zero_()
    tensor_apply_axis(t_in.data_ptr(), t_out.data_ptr(), L, d_in, d_out, R, t_m.data_ptr(),
                      stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert maxdiff(t_out.cpu().numpy(), want) < 1e-9 * np.sqrt(d_in)


def test_fock_path_cfg4_small():
    """BASELINE config 4 at a size the oracle handles: squeezing + beam splitters on Fock-truncated modes."""
    n_modes, d = 3, 12
    st = fock.FockState(n_modes, d)
    want = np.zeros((d,) * n_modes, dtype=complex)
    want[(0,) * n_modes] = 1
    for k in range(6):
        mode = k % n_modes
        r = 0.1 * k % 0.5 + 0.05
        fock.S(mode, r).apply(st)
        want = CO.apply_axis(want, fock.squeeze_matrix(d, r), mode)
        i = k % (n_modes - 1)
        pair = (i, i + 1) if k % 2 else (i + 1, i)
        bs = fock.BS(*pair, np.pi / 4, method=("blocks", "gather", "dense")[k % 3])
        bs.apply(st)
        want = CO.apply_two_axes(want, fock.beamsplitter_matrix(d, np.pi / 4), *pair)   # first leg <-> pair[0]
        fock.Phase(mode, 0.3).apply(st)
        want = CO.apply_axis_diag(want, fock.phase_matrix(d, 0.3), mode)
    assert maxdiff(st.contract(), want) < 1e-12
    assert abs(st.norm() - np.linalg.norm(want)) < 1e-12


@pytest.mark.parametrize("plane_kernel", [1, 0])
@pytest.mark.parametrize("n_modes", [3, 4])
def test_fock_path_cfg4_at_cutoff_32(n_modes, plane_kernel):
    """BASELINE config 4 at its own cutoff d = 32 (3 and 4 of its 6 modes: 32 Ki / 1 Mi amplitudes), on a random
    register.  d = 32 makes the register a 5-bit-per-mode qubit register, so this runs the instantiations the
    6-mode configuration launches: the real S(r) through the workgroup-tile kernel ``k_dense_tile<5, 8, true, ...>`` (every
    mode whose five bits start at bit 3 or higher) and through the line-granular kernel ``k_dense_lds<5, 3, ...>`` (last
    mode: all target bits inside a wavefront); the Fock beam splitter through ``k_mode2_blocks<256>`` (interior pairs, R =
    32 and 1024), and on the last pair (R = 1: the plane is contiguous) through ``k_mode2_plane`` / ``k_mode2_blocks<64>``,
    with the legs in either order.  The assertions below name the kernels that must have run."""
    d = 32
    rng = np.random.default_rng(320 + n_modes)
    psi = rng.standard_normal((d,) * n_modes) + 1j * rng.standard_normal((d,) * n_modes)
    psi /= np.linalg.norm(psi)
    from quantum_computations_amd import _lib
    st = fock.FockState(n_modes, d)
    st.reg.upload(psi)
    st.reg.set_option(_lib.OPT_PLANE_KERNEL, plane_kernel)   # last pair: workgroup-per-plane (1) / thread-per-plane (0)
    want = psi
    bs = fock.beamsplitter_matrix(d, np.pi / 4)
    for mode in range(n_modes):
        r = 0.1 * (mode + 1)
        fock.S(mode, r).apply(st)
        want = CO.apply_axis(want, fock.squeeze_matrix(d, r), mode)
    assert maxdiff(st.contract(), want) < 1e-12
    pairs = [(i, i + 1) for i in range(n_modes - 1)] + [(n_modes - 1, n_modes - 2), (1, 0)]
    for pair in pairs:
        fock.BS(*pair, np.pi / 4).apply(st)
        want = CO.apply_two_axes(want, bs, *pair)
        assert maxdiff(st.contract(), want) < 1e-12, pair
        if max(pair) == n_modes - 1:
            assert st.reg.last_kernel().startswith("k_mode2_plane<" if plane_kernel else "k_mode2_blocks<64, true")
        else:
            assert st.reg.last_kernel().startswith("k_mode2_blocks<256, true, true>"), st.reg.last_kernel()
    # a beam splitter with a phase has complex blocks (the real-matrix fast paths must not be taken for it)
    bs_c = fock.beamsplitter_blocks(d, 0.4, 0.9)
    st.reg.apply_two_mode_blocks(bs_c, n_modes - 2, n_modes - 1)
    want = CO.apply_two_axes(want, fock.beamsplitter_matrix(d, 0.4, 0.9), n_modes - 2, n_modes - 1)
    st.reg.apply_two_mode_blocks(bs_c, 0, 1)
    want = CO.apply_two_axes(want, fock.beamsplitter_matrix(d, 0.4, 0.9), 0, 1)
    fock.S(n_modes - 1, 0.3, 0.7).apply(st)                      # complex squeezing parameter on the lane-bit mode
    want = CO.apply_axis(want, fock.squeeze_matrix(d, 0.3, 0.7), n_modes - 1)
    assert maxdiff(st.contract(), want) < 1e-12
    assert abs(st.norm() - np.linalg.norm(want)) < 1e-12


@pytest.mark.parametrize("n_modes,d", [(4, 8), (5, 4), (3, 16), (3, 12)])
def test_block_operator_on_the_last_two_modes(n_modes, d):
    """``qsv_apply_mode2_blocks`` on the last pair (R = 1) at cutoffs where one workgroup of ``k_mode2_plane`` holds
    several planes (d = 8: 16 planes, d = 4: 64, d = 16: 4; d = 12 does not divide and takes the general kernel), with
    only some of the blocks given (the other plane elements must stay untouched), legs in either order, and a block set
    whose elements are not equally spaced (general kernel again)."""
    rng = np.random.default_rng(d * 10 + n_modes)
    psi = rng.standard_normal((d,) * n_modes) + 1j * rng.standard_normal((d,) * n_modes)
    st = QuditState.from_numpy(psi)
    want = psi
    a, b = n_modes - 2, n_modes - 1
    full = fock.beamsplitter_blocks(d, 0.7)
    kernels = []
    for blocks, legs in [(full, (a, b)), (full, (b, a)), (full[2:d], (a, b)), (full[d - 1:], (b, a))]:
        st.apply_two_mode_blocks(blocks, *legs)
        kernels.append(st.last_kernel())
        m = np.identity(d * d, dtype=complex)
        for idx, block in blocks:
            m[np.ix_(idx, idx)] = block
        want = CO.apply_two_axes(want, m, *legs)
        assert maxdiff(st.to_numpy(), want) < 1e-12, (legs, len(blocks))
    expect = "k_mode2_plane<" if d != 12 else "k_mode2_blocks<64, true"
    assert all(k.startswith(expect) for k in kernels), kernels
    # unequal spacing inside a block: not the plane kernel's case
    idx = [0, 1, d + 2, 2 * d + 1][: min(4, d)]
    q = np.linalg.qr(rng.standard_normal((len(idx), len(idx))))[0]
    st.apply_two_mode_blocks([(idx, q)], a, b)
    assert st.last_kernel().startswith("k_mode2_blocks<64")
    m = np.identity(d * d, dtype=complex)
    m[np.ix_(idx, idx)] = q
    want = CO.apply_two_axes(want, m, a, b)
    assert maxdiff(st.to_numpy(), want) < 1e-12


@pytest.mark.parametrize("real", [True, False])
def test_block_operator_with_unequally_spaced_elements_on_an_interior_pair(real):
    """``qsv_apply_mode2_blocks`` away from the last modes (R >= 8: 256-thread form) with index sets that are not a
    constant step apart (offset-table addressing instead of first + c * step), sizes 1..7 (rows padded to 4 and 8),
    real and complex block matrices, both leg orders."""
    d, n_modes = 8, 3
    rng = np.random.default_rng(88 + real)
    psi = rng.standard_normal((d,) * n_modes) + 1j * rng.standard_normal((d,) * n_modes)
    st = QuditState.from_numpy(psi)
    want = psi
    cells = rng.permutation(d * d)
    blocks, at = [], 0
    for size in (7, 5, 4, 3, 2, 1, 6):
        idx = [int(c) for c in cells[at:at + size]]
        at += size
        m = rng.standard_normal((size, size))
        if not real:
            m = m + 1j * rng.standard_normal((size, size))
        blocks.append((idx, np.linalg.qr(m)[0]))
    for legs in [(0, 1), (1, 0)]:
        st.apply_two_mode_blocks(blocks, *legs)
        assert st.last_kernel() == f"k_mode2_blocks<256, {'true' if real else 'false'}, false>", st.last_kernel()
        m = np.identity(d * d, dtype=complex)
        for idx, block in blocks:
            m[np.ix_(idx, idx)] = block
        want = CO.apply_two_axes(want, m, *legs)
        assert maxdiff(st.to_numpy(), want) < 1e-12, legs


def test_cfg4_full_size_spot_check():
    """BASELINE config 4 at its own size (6 modes x cutoff 32 = 2^30 amplitudes, 16 GiB): S(r) on every mode and
    BS(pi/4) on every neighbouring pair, sampled fibres / planes of the register against U @ in evaluated from their own
    inputs (no host copy of the register) -- the same check bench.py reports as ``secondary.cfg4.max_abs_err``."""
    from quantum_computations_amd.cv_simulator import fock
    st = fock.FockState(6, 32)
    spot = fock.spot_check_register(st.reg, np.random.default_rng(7), fibres=8, planes=3)
    assert spot["S_max_abs_err"] < 1e-14 and spot["BS_max_abs_err"] < 1e-14, spot
    assert abs(spot["norm2_after"] - 1.0) < 1e-9          # exponentials of anti-hermitian truncated generators
    names = " ".join(spot["kernels"])
    assert "k_dense_tile<5, 8" in names and "k_mode2_blocks" in names and "k_mode2_plane" in names, names
    del st

