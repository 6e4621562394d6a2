"""Pin the CPU oracle (oracle/dv_oracle.py) to the reference's own outputs (tests/golden/*.npz).

Every fixture was produced by running /root/reference in the build container (tests/golden/generate_golden.py).
Tolerance: 1e-13 max-abs on normalised kets (fp64; the oracle contracts the small matrix, the reference
multiplies a dense 2^N x 2^N operator, so summation order differs); permutation-only gates must be bit-exact.
"""
from __future__ import annotations

import numpy as np
import pytest

from fixture_io import unpack_ops
from oracle import dv_oracle as O
from quantum_computations_amd import workloads as W

TOL = 1e-13


def test_single_gates_every_position(golden):
    g = golden["dv_single_gates"]
    for case in golden.cases("dv_single_gates"):
        n = case["n"]
        ket = g[f"in_n{n}"]
        op = W.op(case["name"], *case["indices"], **({"angle": case["angle"]} if "angle" in case else {}))
        got = O.apply_gate(ket, op["matrix"], op["indices"])
        want = g[f"out_n{n}"][case["row"]]
        assert np.max(np.abs(got - want)) < TOL, case
        if case["name"] in ("I", "X", "CX", "SWAP"):
            assert np.array_equal(got, want), case


def test_dtype_promotion_matches_reference(golden):
    g = golden["dv_single_gates"]
    real = g["real_in"]
    h = O.apply_gate(real, W.op("H", 1)["matrix"], [1])
    assert h.dtype == g["real_H1"].dtype == np.float64
    assert np.allclose(h, g["real_H1"], atol=TOL)
    t = O.apply_gate(real, W.op("T", 0)["matrix"], [0])
    assert t.dtype == g["real_T0"].dtype == np.complex128
    x = O.apply_gate(np.array([1, 0, 0, 0]), W.op("X", 0)["matrix"], [0])
    assert x.dtype == g["int_X0"].dtype and np.array_equal(x, g["int_X0"])


def test_expand_gate_equivalence(golden):
    """Applying the small matrix == multiplying by the reference's expanded operator, column by column."""
    g = golden["dv_expand_gate"]
    names = {"CX": W.op("CX", 0, 1)["matrix"], "SWAP": W.op("SWAP", 0, 1)["matrix"], "u2": g["u2"], "u4": g["u4"]}
    for case in golden.cases("dv_expand_gate"):
        if "targets" not in case:
            continue
        full = g[case["label"]]
        N = case["N"]
        eye = np.identity(1 << N)
        cols = np.stack([O.apply_gate(eye[:, c], names[case["matrix"]], case["targets"]) for c in range(1 << N)], axis=1)
        assert np.max(np.abs(cols - full)) < TOL, case
    # the measured permutation table of SURVEY.md 8c(2): CX on targets [2, 0], N = 3
    full = g["cx_n3_t20"]
    assert [int(np.argmax(full[:, c])) for c in range(8)] == [0, 5, 2, 7, 4, 1, 6, 3]


def test_permute_tensor_product(golden):
    g = golden["dv_expand_gate"]
    for case in golden.cases("dv_expand_gate"):
        if "order" in case:
            assert np.array_equal(O.permute_qubits(g["perm_in"], case["order"]), g[case["label"]]), case


def test_cfg1_clifford_circuits(golden, state_vectors):
    g = golden["dv_clifford_n4"]
    for seed in g["seeds"]:
        ops = unpack_ops(g[f"meta_{seed}"], g[f"mats_{seed}"])
        # fixture circuit == what the generator produces today
        regenerated = W.random_clifford_circuit(4, 20, int(seed))
        assert [(o["name"], o["indices"]) for o in ops] == [(o["name"], o["indices"]) for o in regenerated]
        start = np.zeros(16)
        start[0] = 1
        final, _ = O.run_circuit(ops, start)
        assert np.max(np.abs(final - g[f"final_{seed}"])) < TOL


def test_random_circuits_depth100(golden):
    g = golden["dv_random_circuits"]
    for case in golden.cases("dv_random_circuits"):
        tag = case["tag"]
        ops = unpack_ops(g[f"meta_{tag}"], g[f"mats_{tag}"])
        regenerated = W.random_circuit(case["n"], case["depth"], case["seed"])
        assert len(ops) == len(regenerated) == case["depth"]
        for a, b in zip(ops, regenerated):
            assert a["indices"] == b["indices"] and np.array_equal(a["matrix"], b["matrix"])
        assert np.array_equal(g[f"init_{tag}"], W.random_ket(case["n"], case["seed"]))
        final, _ = O.run_circuit(ops, g[f"init_{tag}"])
        assert np.max(np.abs(final - g[f"final_{tag}"])) < 1e-12, tag


def test_measure_insert_control_density(golden, state_vectors):
    g = golden["dv_measure_insert"]
    for case in golden.cases("dv_measure_insert"):
        kind, key = case["kind"], case["key"]
        if kind == "measure":
            ket = g[f"ket_n{case['n']}"]
            branches = O.measure_branches(ket, case["q"], case["theta"], case["phi"])
            assert abs(branches[case["result"]][1] - case["norm"]) < TOL, case
            out, s = O.measure(ket, case["q"], case["theta"], case["phi"], case["result"])
            assert s == case["s"] and np.max(np.abs(out - g[key])) < TOL, case
        elif kind == "insert_chain":
            state = np.ones(1)
            for q, name in case["chain"]:
                state = O.insert_qubit(state, q, state_vectors[name])
            assert np.max(np.abs(state - g[key])) < TOL, case
        elif kind == "insert":
            out = O.insert_qubit(g["ket_n3"], case["q"], state_vectors[case["state"]])
            assert np.max(np.abs(out - g[key])) < TOL, case
        elif kind == "control":
            ops = unpack_ops(g[f"{key}_meta"], g[f"{key}_mats"], state_vectors)
            out, results = O.run_circuit(ops, np.ones(1))
            assert results == case["results"]
            assert np.max(np.abs(out - g[key])) < TOL, case
        elif kind == "density":
            (op,) = unpack_ops(g[f"{key}_meta"], g[f"{key}_mats"])
            out = O.apply_gate(g[f"rho_n{case['n']}"], op["matrix"], op["indices"])
            assert np.max(np.abs(out - g[key])) < TOL, case
    # the measured insert example of SURVEY.md 8c(5): amplitudes 1/sqrt2 at indices 2, 3
    a = g["ins_a"]
    assert np.allclose(np.abs(a), [0, 0, 2 ** -0.5, 2 ** -0.5, 0, 0, 0, 0])


def test_unconjugated_projector_is_reproduced(golden):
    """phi != 0: the reference projects on eig^T, not eig^dagger; the conjugated projector would differ."""
    g = golden["dv_measure_insert"]
    case = next(c for c in golden.cases("dv_measure_insert") if c["kind"] == "measure" and c["key"].startswith("Mgen_n3_q1_r0"))
    ket = g["ket_n3"]
    e0, _ = O.measurement_vectors(case["theta"], case["phi"])
    psi = np.moveaxis(ket.reshape(2, 2, 2), 1, 0).reshape(2, -1)
    textbook = np.conj(e0[0]) * psi[0] + np.conj(e0[1]) * psi[1]
    textbook /= np.linalg.norm(textbook)
    assert np.max(np.abs(textbook - g[case["key"]])) > 1e-3


def test_grover3_anchor(golden, state_vectors):
    g = golden["dv_grover3"]
    for case in golden.cases("dv_grover3"):
        ops = W.grover3_ops(case["tagged"])
        out, _ = O.run_circuit(ops, np.ones(1))
        assert np.max(np.abs(out - g[case["key"]])) < TOL
        probs = np.abs(out) ** 2
        assert abs(probs[case["tagged"]].sum() - 1.0) < 1e-12          # success probability 1 - 2e-15
        assert np.allclose(probs[case["tagged"]], 0.5, atol=1e-12)
    ccz = np.stack([O.run_circuit(W.ccz_ops(), np.identity(8)[:, c])[0] for c in range(8)], axis=1)
    assert np.max(np.abs(ccz - g["ccz_operator"])) < TOL
    phase = ccz[0, 0]
    assert np.allclose(ccz / phase, np.diag([1, 1, 1, 1, 1, 1, 1, -1]), atol=1e-12)


@pytest.mark.parametrize("n", [2, 5, 9])
def test_oracle_is_linear_and_unitary(n):
    rng = np.random.default_rng(n)
    a = W.random_ket(n, 1)
    b = W.random_ket(n, 2)
    ops = W.random_circuit(n, 30, 3)
    fa, _ = O.run_circuit(ops, a)
    fb, _ = O.run_circuit(ops, b)
    fab, _ = O.run_circuit(ops, 0.3 * a + (0.1 - 0.7j) * b)
    assert np.max(np.abs(fab - (0.3 * fa + (0.1 - 0.7j) * fb))) < 1e-13
    assert abs(np.linalg.norm(fa) - 1) < 1e-13
    assert abs(np.vdot(fa, fb) - np.vdot(a, b)) < 1e-13


# ---- the C restatement (oracle/csrc/qsv_oracle.c) is pinned to the same vectors ------------------------------
def test_c_oracle_random_circuits(golden):
    from oracle import c_oracle
    g = golden["dv_random_circuits"]
    for case in golden.cases("dv_random_circuits"):
        tag = case["tag"]
        ops = unpack_ops(g[f"meta_{tag}"], g[f"mats_{tag}"])
        state = np.array(g[f"init_{tag}"], dtype=np.complex128)
        c_oracle.run_circuit_inplace(ops, state)
        assert np.max(np.abs(state - g[f"final_{tag}"])) < 1e-12, tag


def test_c_oracle_every_position_vs_numpy_oracle():
    from oracle import c_oracle
    rng = np.random.default_rng(0)
    n = 7
    ket = W.random_ket(n, 70)
    state = ket.copy()
    want = ket
    for q0 in range(n):
        u = W.haar_unitary(2, rng)
        c_oracle.apply_gate_inplace(state, u, [q0])
        want = O.apply_gate(want, u, [q0])
        for q1 in range(n):
            if q1 != q0:
                u = W.haar_unitary(4, rng)
                c_oracle.apply_gate_inplace(state, u, [q0, q1])
                want = O.apply_gate(want, u, [q0, q1])
    assert np.max(np.abs(state - want)) < 1e-12


def test_c_oracle_dense_matvec_is_the_literal_algorithm(golden):
    from oracle import c_oracle
    g = golden["dv_expand_gate"]
    full = g["u4_n4_t31"]
    ket = g["perm_in"]
    assert np.max(np.abs(c_oracle.dense_matvec(full, ket) - full @ ket)) < TOL
    assert np.max(np.abs(c_oracle.dense_matvec(full, ket) - O.apply_gate(ket, g["u4"], [3, 1]))) < TOL


def test_readout_functions_match_the_reference(golden):
    """ket2dm / fidelity (all four branches) / purity of numpy_quantum.py:110-166 as the reference evaluated them, and
    reduced density matrices derived from its ket2dm: pins the package's host mirror and the oracle's reduced_density."""
    from quantum_computations_amd.dv_simulator import numpy_quantum as npq
    g = golden["dv_readout"]
    for case in golden.cases("dv_readout"):
        tag, v = case["tag"], case["values"]
        a, b, rho, sigma = (g[f"{tag}_{k}"] for k in ("a", "b", "rho", "sigma"))
        assert abs(npq.fidelity(a, b) - v["ket_ket"]) < 1e-14
        assert abs(npq.fidelity(a, rho) - v["ket_dm"]) < 1e-14
        assert abs(npq.fidelity(sigma, b) - v["dm_ket"]) < 1e-14
        assert abs(npq.fidelity(rho, sigma) - v["dm_dm"]) < 1e-10
        assert abs(npq.purity(rho) - v["purity_rho"]) < 1e-14 and abs(npq.purity(sigma) - v["purity_sigma"]) < 1e-14
        assert abs(npq.purity(npq.ket2dm(a)) - v["purity_pure"]) < 1e-14
        for kept in case["kept"]:
            want = g[f"{tag}_rdm_{'_'.join(map(str, kept))}"]
            assert np.max(np.abs(O.reduced_density(a, kept) - want)) < 1e-14, (tag, kept)
