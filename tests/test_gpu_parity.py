"""GPU parity: the HIP path (through the C ABI) against the reference's golden vectors and the CPU oracle.

Tolerance (stated by BASELINE.json's north_star as "a stated fp64 tolerance"): max-abs 1e-12 on normalised
kets after a depth-100 circuit, 1e-13 for a single gate; permutation-only gates (X, CX, SWAP) bit-exact.
The kernels use fp64 FMA, NumPy does not, and the summation orders differ -- hence not bit-exact for dense gates.
"""
from __future__ import annotations

import numpy as np
from pathlib import Path
import pytest

pytestmark = pytest.mark.gpu

from fixture_io import unpack_ops
from oracle import cv_oracle as CO
from oracle import dv_oracle as O
from quantum_computations_amd import _lib
from quantum_computations_amd import workloads as W
from quantum_computations_amd.device import DeviceState, QuditState
from quantum_computations_amd.dv_simulator import gates as G
from quantum_computations_amd.dv_simulator.simulator import ClassicalControl, Simulator
from quantum_computations_amd.dv_simulator.states import State

GATE_TOL = 1e-13
CIRCUIT_TOL = 1e-12


def maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))


# ---- golden vectors from the reference ---------------------------------------------------------------------
def test_golden_single_gates(golden):
    g = golden["dv_single_gates"]
    for case in golden.cases("dv_single_gates"):
        ket = g[f"in_n{case['n']}"]
        gate = W.to_gates([W.op(case["name"], *case["indices"], **({"angle": case["angle"]} if "angle" in case else {}))])[0]
        got = gate.apply(ket)
        want = g[f"out_n{case['n']}"][case["row"]]
        assert maxdiff(got, want) < GATE_TOL, case
        if case["name"] in ("I", "X", "CX", "SWAP"):
            assert np.array_equal(got, want), case
    # dtype behaviour of the ndarray API
    assert G.H(1).apply(g["real_in"]).dtype == np.float64
    assert maxdiff(G.H(1).apply(g["real_in"]), g["real_H1"]) < GATE_TOL
    assert G.T(0).apply(g["real_in"]).dtype == np.complex128
    x = G.X(0).apply(np.array([1, 0, 0, 0]))
    assert x.dtype == g["int_X0"].dtype and np.array_equal(x, g["int_X0"])
    # input untouched (the reference returns a new array)
    ket = g["in_n3"].copy()
    G.H(0).apply(ket)
    assert np.array_equal(ket, g["in_n3"])


def test_golden_cfg1_clifford(golden):
    g = golden["dv_clifford_n4"]
    for seed in g["seeds"]:
        ops = unpack_ops(g[f"meta_{seed}"], g[f"mats_{seed}"])
        final = Simulator(W.to_gates(ops)).run([State.ZERO] * 4)
        assert maxdiff(final, g[f"final_{seed}"]) < GATE_TOL * 20


def test_golden_random_circuits(golden):
    g = golden["dv_random_circuits"]
    for case in golden.cases("dv_random_circuits"):
        tag = case["tag"]
        ops = unpack_ops(g[f"meta_{tag}"], g[f"mats_{tag}"])
        final = Simulator(W.to_gates(ops)).run(g[f"init_{tag}"])
        assert maxdiff(final, g[f"final_{tag}"]) < CIRCUIT_TOL, tag
        # every gate through the dense kernels too (no diagonal / permutation shortcuts)
        dev = DeviceState.from_numpy(g[f"init_{tag}"])
        dev.set_option(_lib.OPT_SPECIALIZE, 0)
        for gate in W.to_gates(ops):
            gate.apply(dev)
        assert maxdiff(dev.to_numpy(), g[f"final_{tag}"]) < CIRCUIT_TOL, tag


def test_golden_measure_insert_control_density(golden, state_vectors):
    g = golden["dv_measure_insert"]
    for case in golden.cases("dv_measure_insert"):
        kind, key = case["kind"], case["key"]
        if kind == "measure":
            ket = g[f"ket_n{case['n']}"]
            gate = G.M(case["q"], case["theta"], case["phi"], result=case["result"])
            out, s = gate.apply(ket)
            assert s == case["s"] and maxdiff(out, g[key]) < GATE_TOL, case
            dev = DeviceState.from_numpy(ket)
            probs = dev.measure_probs(case["q"], *gate.eigenvectors())
            assert abs(np.sqrt(probs[case["result"]]) - case["norm"]) < GATE_TOL, case
        elif kind == "insert_chain":
            circuit = [G.Insert(q, State[name]) for q, name in case["chain"]]
            assert maxdiff(Simulator(circuit).run(), g[key]) < GATE_TOL, case
        elif kind == "insert":
            out = G.Insert(case["q"], State[case["state"]]).apply(g["ket_n3"])
            assert maxdiff(out, g[key]) < GATE_TOL, case
        elif kind == "control":
            ops = unpack_ops(g[f"{key}_meta"], g[f"{key}_mats"], state_vectors)
            sim = Simulator(W.to_gates(ops))
            out = sim.run()
            assert sim.results == case["results"] and maxdiff(out, g[key]) < GATE_TOL, case
        elif kind == "density":
            (op,) = unpack_ops(g[f"{key}_meta"], g[f"{key}_mats"])
            out = W.to_gates([op])[0].apply(g[f"rho_n{case['n']}"])
            assert out.shape == g[key].shape and maxdiff(out, g[key]) < GATE_TOL, case


def test_golden_grover3(golden):
    g = golden["dv_grover3"]
    for case in golden.cases("dv_grover3"):
        out = Simulator(W.to_gates(W.grover3_ops(case["tagged"]))).run()
        assert maxdiff(out, g[case["key"]]) < GATE_TOL * 10
        assert abs((np.abs(out) ** 2)[case["tagged"]].sum() - 1.0) < 1e-12


def test_sampled_measurement_uses_global_numpy_rng():
    ket = W.random_ket(5, 3)
    outcomes = []
    for trial in range(2):
        np.random.seed(1234)
        sim = Simulator([G.MZ(0), G.MX(1), G.M(0, 0.3, 0.9)])
        final = sim.run(ket)
        outcomes.append((tuple(sim.results), final))
    assert outcomes[0][0] == outcomes[1][0] and np.array_equal(outcomes[0][1], outcomes[1][1])
    # replay with forced results through the oracle
    ops = [{"name": "M", "indices": [0], "theta": 0.0, "phi": 0.0, "result": outcomes[0][0][0]},
           {"name": "M", "indices": [1], "theta": np.pi / 2, "phi": 0.0, "result": outcomes[0][0][1]},
           {"name": "M", "indices": [0], "theta": 0.3, "phi": 0.9, "result": outcomes[0][0][2]}]
    want, _ = O.run_circuit(ops, ket)
    assert maxdiff(outcomes[0][1], want) < GATE_TOL * 10


# ---- every target position, every kernel regime, against the oracle ----------------------------------------
@pytest.mark.parametrize("n", [6, 7, 11, 14])
@pytest.mark.parametrize("unroll", [0, 1, 2])
def test_dense_1q_every_position(n, unroll):
    rng = np.random.default_rng(n)
    ket = W.random_ket(n, n)
    dev = DeviceState.from_numpy(ket)
    dev.set_option(_lib.OPT_UNROLL, unroll)
    want = ket
    for q in range(n):
        u = W.haar_unitary(2, rng)
        dev.apply_matrix(u, [q])
        want = O.apply_gate(want, u, [q])
    assert maxdiff(dev.to_numpy(), want) < CIRCUIT_TOL


@pytest.mark.parametrize("n", [6, 9, 12])
def test_dense_2q_every_ordered_pair(n):
    rng = np.random.default_rng(100 + n)
    ket = W.random_ket(n, n)
    dev = DeviceState.from_numpy(ket)
    want = ket
    for q0 in range(n):
        for q1 in range(n):
            if q0 == q1:
                continue
            u = W.haar_unitary(4, rng)
            dev.apply_matrix(u, [q0, q1])
            want = O.apply_gate(want, u, [q0, q1])
            if (q0 * n + q1) % 7 == 0:   # renormalisation-free check as we go
                assert maxdiff(dev.to_numpy(), want) < CIRCUIT_TOL, (q0, q1)
    assert maxdiff(dev.to_numpy(), want) < CIRCUIT_TOL


@pytest.mark.parametrize("specialize", [1, 0])
@pytest.mark.parametrize("n", [6, 10, 13])
def test_named_gates_every_position(n, specialize):
    ket = W.random_ket(n, 40 + n)
    dev = DeviceState.from_numpy(ket)
    dev.set_option(_lib.OPT_SPECIALIZE, specialize)
    want = ket
    for q in range(n):
        for name in ("X", "Y", "Z", "H", "P", "Tdg"):
            gate = getattr(G, name)(q)
            gate.apply(dev)
            want = O.apply_gate(want, gate.matrix, [q])
        angle = 0.1 + q
        G.RZ(q, angle).apply(dev)
        want = O.apply_gate(want, G.RZ(q, angle).matrix, [q])
    for q0 in range(n):
        for q1 in range(n):
            if q0 == q1:
                continue
            for cls in (G.CX, G.CZ, G.SWAP):
                gate = cls(q0, q1)
                gate.apply(dev)
                want = O.apply_gate(want, gate.matrix, [q0, q1])
    assert maxdiff(dev.to_numpy(), want) < CIRCUIT_TOL


@pytest.mark.parametrize("n", [6, 12])
def test_permutation_gates_are_bit_exact(n):
    ket = W.random_ket(n, 77)
    dev = DeviceState.from_numpy(ket)
    want = ket
    for q0 in range(n):
        q1 = (q0 * 5 + 3) % n
        if q1 == q0:
            q1 = (q0 + 1) % n
        for gate in (G.X(q0), G.CX(q0, q1), G.SWAP(q1, q0), G.CX(q1, q0)):
            gate.apply(dev)
            want = O.apply_gate(want, gate.matrix, gate.indices)
    assert np.array_equal(dev.to_numpy(), want)


@pytest.mark.parametrize("n", [5, 8, 13])
def test_controlled_and_multicontrolled(n):
    rng = np.random.default_rng(n)
    ket = W.random_ket(n, 5)
    dev = DeviceState.from_numpy(ket)
    want = ket
    for trial in range(12):
        k = int(rng.integers(1, min(n, 5)))
        qs = [int(v) for v in rng.choice(n, size=k + 1, replace=False)]
        controls, target = qs[:-1], qs[-1]
        u = W.haar_unitary(2, rng)
        dev.apply_controlled(u, controls, target)
        full = np.identity(1 << (k + 1), dtype=complex)
        full[-2:, -2:] = u
        want = O.apply_gate(want, full, controls + [target])
        # multi-controlled phase on another random subset
        qs = [int(v) for v in rng.choice(n, size=int(rng.integers(1, min(n, 6) + 1)), replace=False)]
        phase = np.exp(1j * rng.uniform(0, 2 * np.pi))
        dev.apply_mcphase(qs, phase)
        d = np.ones(1 << len(qs), dtype=complex)
        d[-1] = phase
        want = O.apply_gate(want, np.diag(d), qs)
    assert maxdiff(dev.to_numpy(), want) < CIRCUIT_TOL


@pytest.mark.parametrize("n,k", [(3, 3), (6, 3), (9, 4), (10, 5), (11, 6)])
def test_generic_kq(n, k):
    rng = np.random.default_rng(k * 10 + n)
    ket = W.random_ket(n, 9)
    dev = DeviceState.from_numpy(ket)
    want = ket
    for trial in range(4):
        qs = [int(v) for v in rng.choice(n, size=k, replace=False)]
        u = W.haar_unitary(1 << k, rng)
        dev.apply_matrix(u, qs)
        want = O.apply_gate(want, u, qs)
        d = np.exp(1j * rng.uniform(0, 6.28, 1 << k))
        dev.apply_matrix(np.diag(d), qs)
        want = O.apply_gate(want, np.diag(d), qs)
    assert maxdiff(dev.to_numpy(), want) < CIRCUIT_TOL


def test_six_qubit_gates_on_the_matrix_cores():
    """k = 6 on registers of 16+ qubits runs on the f64 matrix cores (k_dense_mfma<6>): complex and real 64 x 64
    matrices, targets all high, mixed, with one to all six of them on the lowest index bits (lanes then run over the
    lowest free bits), legs in any order; against the oracle."""
    n = 16
    rng = np.random.default_rng(66)
    ket = W.random_ket(n, 66)
    dev = DeviceState.from_numpy(ket)
    want = ket
    cases = [[6, 8, 9, 11, 13, 15], [4, 5, 7, 10, 12, 14], [0, 6, 7, 9, 12, 15], [0, 1, 2, 3, 4, 5], [1, 3, 5, 8, 10, 14],
             [15, 14, 13, 12, 11, 10], [2, 3, 9, 4, 0, 15], [0, 1, 2, 9, 12, 14], [2, 7, 8, 10, 11, 13], [0, 1, 2, 3, 7, 15],
             [1, 2, 4, 5, 6, 8]]
    for i, bits in enumerate(cases):
        bits = list(bits)
        rng.shuffle(bits)
        qs = [n - 1 - b for b in bits]
        u = W.haar_unitary(64, rng) if i % 2 == 0 else np.linalg.qr(rng.standard_normal((64, 64)))[0]
        dev.apply_matrix(u, qs)
        assert dev.last_kernel().startswith("k_dense_mfma<6, "), dev.last_kernel()
        want = O.apply_gate(want, u, qs)
        assert maxdiff(dev.to_numpy(), want) < CIRCUIT_TOL, bits


@pytest.mark.parametrize("product", [3, 4])
def test_complex_blocks_in_three_and_four_multiplications(product):
    """QSV_OPT_COMPLEX_PRODUCT: complex 5- and 6-qubit blocks with three real multiplications per matrix entry (Ar xr,
    Ai xi, (Ar + Ai)(xr + xi); shipped on the matrix cores, a measurement variant on the vector kernels) and with four,
    in every kernel form that implements both, against the oracle."""
    n = 16
    rng = np.random.default_rng(30 + product)
    ket = W.random_ket(n, 5)
    dev = DeviceState.from_numpy(ket)
    dev.set_option(_lib.OPT_COMPLEX_PRODUCT, product)
    want = ket
    seen = set()
    for variant, cases in ((0, [[6, 8, 9, 11, 13], [0, 3, 7, 11, 15], [0, 1, 2, 7, 11], [3, 4, 5, 6, 7]]),
                           (3, [[6, 8, 9, 11, 13], [0, 7, 11, 13, 15], [1, 2, 7, 11, 12], [0, 1, 2, 3, 4]]),
                           (4, [[6, 8, 9, 11, 13], [3, 4, 5, 6, 7], [11, 12, 13, 14, 15]]),
                           (5, [[6, 8, 9, 11, 13], [0, 1, 2, 3, 4], [0, 3, 7, 11, 15]]),
                           (0, [[6, 8, 9, 11, 13, 15], [0, 1, 2, 3, 4, 5], [2, 3, 9, 4, 0, 15]])):
        dev.set_option(_lib.OPT_KQ_VARIANT, variant)
        for bits in cases:
            bits = list(bits)
            rng.shuffle(bits)
            qs = [n - 1 - b for b in bits]
            u = W.haar_unitary(1 << len(bits), rng)
            dev.apply_matrix(u, qs)
            seen.add(dev.last_kernel())
            want = O.apply_gate(want, u, qs)
            assert maxdiff(dev.to_numpy(), want) < CIRCUIT_TOL, (bits, dev.last_kernel())
    three = {"mfma6": any(x.startswith("k_dense_mfma<6, ") and x.endswith(", false, true>") for x in seen),
             "mfma5": any(x.startswith("k_dense_mfma<5, ") and x.endswith(", false, true>") for x in seen),
             "lds": any(x.startswith("k_dense_lds<5, ") and x.endswith(", 256, true>") for x in seen),
             "tile": any(x.startswith("k_dense_tile<5, 8, false, ") and x.count(",") == 4 for x in seen),
             "big": any(x.startswith("k_dense_big<5, 0, ") and x.count(",") == 3 for x in seen)}
    assert all(three.values()) if product == 3 else not any(three.values()), (three, seen)


def test_fused_blocks_as_gate_sequences_every_low_target_set():
    """Round 3: a fused 5-qubit block is applied as the sequence of its 1- and 2-qubit source gates in one pass
    (``qsv_apply_sequence`` -> k_seq_big / k_seq_lds<KB>) instead of its dense 32 x 32 product.  Every set of block qubits
    below bit 6 (all subsets of the six lane bits, up to five), the others on random high bits, legs in any order, gates
    on every leg / ordered pair of legs incl. CX, CZ, SWAP; against the oracle applying the source gates one by one."""
    import itertools

    from quantum_computations_amd.fusion import fuse_circuit
    n, k = 13, 5
    rng = np.random.default_rng(77)
    ket = W.random_ket(n, 32)
    dev = DeviceState.from_numpy(ket)
    dev.set_option(_lib.OPT_SEQUENCE_WORK, 1 << 20)          # every sequence as a sequence, whatever its length
    want = ket
    kernels = set()
    fixed = [G.CX, G.CZ, G.SWAP]
    for n_low in range(k + 1):
        for low_bits in itertools.combinations(range(6), n_low):
            high_bits = [int(b) for b in rng.choice(np.arange(6, n), size=k - n_low, replace=False)]
            bits = list(low_bits) + high_bits
            rng.shuffle(bits)
            qs = [n - 1 - b for b in bits]
            sources = []
            for _ in range(int(rng.integers(2, 9))):
                if rng.random() < 0.45:
                    sources.append(G.Gate([int(rng.choice(qs))], W.haar_unitary(2, rng)))
                else:
                    a, b = (int(q) for q in rng.choice(qs, size=2, replace=False))
                    pick = int(rng.integers(0, 5))
                    sources.append(fixed[pick](a, b) if pick < 3 else G.Gate([a, b], W.haar_unitary(4, rng)))
            sources.append(G.Gate([qs[0], qs[4]], W.haar_unitary(4, rng)))       # every block touches all five qubits:
            sources.append(G.Gate([qs[1], qs[2]], W.haar_unitary(4, rng)))       # legs 0-4, 1-2, 3
            sources.append(G.Gate([qs[3]], W.haar_unitary(2, rng)))
            fused = fuse_circuit(sources, 5, n_qubits=n)
            assert len(fused) == 1 and sorted(fused[0].indices) == sorted(qs) and len(fused[0].sources) == len(sources)
            fused[0].apply(dev)
            kernels.add(dev.last_kernel())
            for g in sources:
                want = O.apply_gate(want, np.asarray(g.matrix, dtype=complex), list(g.indices))
            assert maxdiff(dev.to_numpy(), want) < CIRCUIT_TOL, (bits, dev.last_kernel())
    assert kernels == {"k_seq_big<true>", "k_seq_lds<0, true>", "k_seq_lds<1, true>", "k_seq_lds<2, true>", "k_seq_lds<3, true>"}, kernels
    # the dense product gives the same state (to rounding); a block with a 3-qubit source gate has no sequence form
    block = fused[0]
    a = DeviceState.from_numpy(ket)
    b = DeviceState.from_numpy(ket)
    a.set_option(_lib.OPT_SEQUENCE_WORK, 1 << 20)
    block.apply(a)
    b.apply_matrix(block.matrix, block.indices)
    assert a.last_kernel().startswith("k_seq_") and not b.last_kernel().startswith("k_seq_")
    assert maxdiff(a.to_numpy(), b.to_numpy()) < GATE_TOL * 10
    # the work limit: by default a sequence costing more than the dense block's share goes as the dense block; 0 = never
    work = sum(256 if len(g.indices) == 1 else 512 for g in block.sources)
    a.set_option(_lib.OPT_SEQUENCE_WORK, work - 1)
    block.apply(a)
    assert not a.last_kernel().startswith("k_seq_")
    a.set_option(_lib.OPT_SEQUENCE_WORK, work)
    block.apply(a)
    assert a.last_kernel().startswith("k_seq_")
    a.set_option(_lib.OPT_SEQUENCE_WORK, 0)
    block.apply(a)
    assert not a.last_kernel().startswith("k_seq_")
    a.set_option(_lib.OPT_SEQUENCE_WORK, 1 << 20)
    block.sources = block.sources + [G.Gate([qs[0], qs[1], qs[2]], np.identity(8))]
    block.apply(a)
    assert a.last_kernel().startswith(("k_dense_big<5", "k_dense_lds<5")), a.last_kernel()


@pytest.mark.parametrize("k", [5, 6])
def test_fused_blocks_as_gate_lists_on_lds_tiles(k):
    """Round 3: a fused block of 5 or 6 qubits applied as the LIST of its 1- and 2-qubit source gates on LDS-resident
    4096-amplitude tiles (``qsv_apply_sequence`` -> ``k_seq_tile``) instead of its dense product: target sets with 0..k
    qubits below bit 6 (every count, random choices), the others anywhere above, legs in any order, gates on every kind
    of leg pair incl. CX / CZ / SWAP; against the oracle applying the source gates one by one, on registers of 12, 13 and
    17 qubits (1, 2 and 32 tiles).  An explicit limit admits 5-qubit blocks too; by default only 6-qubit blocks of at most
    12 gates take this form."""
    from quantum_computations_amd.fusion import fuse_circuit
    fixed = [G.CX, G.CZ, G.SWAP]
    for n in (12, 13, 17):
        rng = np.random.default_rng(100 * k + n)
        ket = W.random_ket(n, 33)
        dev = DeviceState.from_numpy(ket)
        dev.set_option(_lib.OPT_TILE_SEQUENCE_GATES, 48)
        want = ket
        for trial in range(14):
            n_low = trial % (k + 1)
            low_bits = [int(b) for b in rng.choice(6, size=min(n_low, 6), replace=False)]
            high_bits = [int(b) for b in rng.choice(np.arange(6, n), size=k - len(low_bits), replace=False)]
            bits = low_bits + high_bits
            rng.shuffle(bits)
            qs = [n - 1 - b for b in bits]
            sources = []
            for _ in range(int(rng.integers(1, 9))):
                if rng.random() < 0.45:
                    sources.append(G.Gate([int(rng.choice(qs))], W.haar_unitary(2, rng)))
                else:
                    a, b = (int(q) for q in rng.choice(qs, size=2, replace=False))
                    pick = int(rng.integers(0, 5))
                    sources.append(fixed[pick](a, b) if pick < 3 else G.Gate([a, b], W.haar_unitary(4, rng)))
            for i in range(0, k - 1, 2):                                       # every block touches all its qubits
                sources.append(G.Gate([qs[i], qs[i + 1]], W.haar_unitary(4, rng)))
            sources.append(G.Gate([qs[k - 1]], W.haar_unitary(2, rng)))
            fused = fuse_circuit(sources, k, n_qubits=n)
            assert len(fused) == 1 and sorted(fused[0].indices) == sorted(qs)
            fused[0].apply(dev)
            assert dev.last_kernel() == "k_seq_tile<true>", (bits, dev.last_kernel())
            for g in sources:
                want = O.apply_gate(want, np.asarray(g.matrix, dtype=complex), list(g.indices))
            assert maxdiff(dev.to_numpy(), want) < CIRCUIT_TOL, (n, bits)
        # the limit on the list length; 0 = never; registers below 12 qubits have no tile
        block = fused[0]
        dev.set_option(_lib.OPT_TILE_SEQUENCE_GATES, len(block.sources) - 1)
        block.apply(dev)
        assert not dev.last_kernel().startswith("k_seq_tile")
        dev.set_option(_lib.OPT_TILE_SEQUENCE_GATES, 0)
        block.apply(dev)
        assert not dev.last_kernel().startswith("k_seq_tile")
    # the default: 6-qubit blocks of at most 12 gates on the tiles, 5-qubit blocks as their dense product
    plain = DeviceState.from_numpy(W.random_ket(17, 2))
    block.apply(plain)
    assert plain.last_kernel().startswith("k_seq_tile") == (k == 6 and len(block.sources) <= 12), plain.last_kernel()
    small = DeviceState.from_numpy(W.random_ket(11, 1))
    small.set_option(_lib.OPT_TILE_SEQUENCE_GATES, 48)
    qs = list(range(k))
    sources = [G.Gate([qs[i], qs[i + 1]], W.haar_unitary(4, np.random.default_rng(i))) for i in range(k - 1)]
    fuse_circuit(sources, k, n_qubits=11)[0].apply(small)
    assert not small.last_kernel().startswith("k_seq_tile")


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("k", [3, 4, 5])
def test_register_blocked_kq_every_low_target_set(k, variant):
    """k = 3..5 dense gates with EVERY set of target bits below 6 (all subsets of the six lane bits, up to k of them),
    the other legs on random high bits, legs in any order, complex and real matrices.  ``variant`` 3 is the
    line-granular kernel (k_dense_lds: address arithmetic for lane bits 3..5, LDS for bits 0..2), 1 the wave-shuffle
    form (k_dense_big<K, KL>), 2 the no-exchange form, 4 the workgroup tile staged through LDS (k_dense_tile: k = 4, 5 with every target on bit 3
    or higher), 5 the matrix-core kernel (k_dense_mfma<5>; k = 3, 4 as 0), 0 the shipped per-case choice between them: all must agree with the oracle."""
    import itertools

    from quantum_computations_amd import _lib
    n = k + 8
    rng = np.random.default_rng(50 + k)
    ket = W.random_ket(n, 31)
    dev = DeviceState.from_numpy(ket)
    dev.set_option(_lib.OPT_KQ_VARIANT, variant)
    want = ket
    kernels = set()
    for n_low in range(k + 1):
        for low_bits in itertools.combinations(range(6), n_low):
            high_bits = [int(b) for b in rng.choice(np.arange(6, n), size=k - n_low, replace=False)]
            bits = list(low_bits) + high_bits
            rng.shuffle(bits)
            qs = [n - 1 - b for b in bits]
            u = W.haar_unitary(1 << k, rng)
            if len(kernels) % 3 == 2:
                u = np.linalg.qr(rng.standard_normal((1 << k, 1 << k)))[0]      # real: the two-FMA variant
            dev.apply_matrix(u, qs)
            kernels.add(dev.last_kernel())
            want = O.apply_gate(want, u, qs)
            assert maxdiff(dev.to_numpy(), want) < CIRCUIT_TOL, (bits, dev.last_kernel())
    if variant == 0:
        assert any(name.startswith("k_dense_lds<5") for name in kernels) == (k == 5), kernels
    elif variant == 3:
        assert all(name.startswith(f"k_dense_lds<{k}, ") for name in kernels), kernels
        assert {name.split(", ")[1] for name in kernels} == {"0", "1", "2", "3"}     # 0..3 targets inside a line
        assert {name.split(", ")[3] for name in kernels} == {"true>", "false>"}      # real and complex matrices
    elif variant == 1:
        assert {name.split(", ")[1] for name in kernels} == {str(j) for j in range(k + 1)}, kernels
    elif variant == 4:
        assert any(name.startswith(f"k_dense_tile<{k}, ") for name in kernels), kernels
    elif variant == 5:
        assert all(name.startswith("k_dense_mfma<5, ") for name in kernels) == (k == 5), kernels
    elif variant == 6:
        assert any(name.startswith("k_dense_mtile5<") for name in kernels) == (k == 5), kernels
    else:
        assert kernels == {f"k_dense_big<{k}, 0, false>"} or kernels == {f"k_dense_big<{k}, 0, true>",
                                                                        f"k_dense_big<{k}, 0, false>"}, kernels


@pytest.mark.parametrize("n", [1, 4, 9, 12])
def test_measure_insert_every_position(n):
    ket = W.random_ket(n, 21)
    for q in range(n):
        for theta, phi, result in [(0.0, 0.0, 0), (np.pi / 2, 0.0, 1), (0.9, 2.2, 1)]:
            out, s = G.M(q, theta, phi, result=result).apply(ket)
            want, _ = O.measure(ket, q, theta, phi, result)
            assert maxdiff(out, want) < GATE_TOL * 10, (q, theta, phi)
    for q in range(n + 1):
        out = G.Insert(q, State.TDG).apply(ket)
        assert maxdiff(out, O.insert_qubit(ket, q, State.TDG.get())) < GATE_TOL, q


@pytest.mark.parametrize("variant", [0, 1])
def test_streaming_readout_kernels_every_position(variant):
    """Registers of 2^14 amplitudes and more run the streaming forms of measurement / insertion / permutation /
    table diagonals (``variant`` 0; 1 forces the plain grid-stride forms the small-register tests above exercise):
    every qubit position -- lane bits resolved by wave shuffles and lane gathers, high bits by paired streams --
    against the oracle."""
    from quantum_computations_amd import _lib
    n = 15
    ket = W.random_ket(n, 77)
    rng = np.random.default_rng(7)
    names = set()
    for q in range(n):
        theta, phi, result = [(0.0, 0.0, 0), (np.pi / 2, 0.0, 1), (0.9, 2.2, 1)][q % 3]
        dev = DeviceState.from_numpy(ket)
        dev.set_option(_lib.OPT_READOUT_VARIANT, variant)
        eigs = G.M(q, theta, phi).eigenvectors()
        p = dev.measure_probs(q, *eigs)
        names.add(dev.last_kernel())
        want, _ = O.measure(ket, q, theta, phi, result)
        want_p = [nrm ** 2 for _, nrm in O.measure_branches(ket, q, theta, phi)]
        out, s = G.M(q, theta, phi, result=result).apply(dev)
        names.add(out.last_kernel())
        assert s == result and maxdiff(out.to_numpy(), want) < GATE_TOL * 10, (q, theta, phi)
        assert maxdiff(np.array(p), np.array(want_p)) < 1e-13, (q, p, want_p)
        out.insert(q, State.TDG.get())                        # back to n qubits at the same place
        names.add(out.last_kernel())
        assert maxdiff(out.to_numpy(), O.insert_qubit(want, q, State.TDG.get())) < GATE_TOL * 10, q
    dev = DeviceState.from_numpy(ket)
    dev.set_option(_lib.OPT_READOUT_VARIANT, variant)
    dev.insert(n, [0.6, 0.8j])                                # a new least significant qubit, and a new top one
    dev.insert(0, [0.8, -0.6])
    want = O.insert_qubit(O.insert_qubit(ket, n, np.array([0.6, 0.8j])), 0, np.array([0.8, -0.6]))
    assert maxdiff(dev.to_numpy(), want) < GATE_TOL
    for trial in range(8):
        order = [int(v) for v in rng.permutation(n)]
        if trial == 0:
            order = list(range(n))
        elif trial == 1:
            order = list(range(n - 3)) + [n - 1, n - 3, n - 2]          # only bits inside a 128-byte line move
        elif trial == 2:
            order = list(range(n))[::-1]
        elif trial == 3:
            order = list(range(3, n)) + [0, 1, 2]
        dev = DeviceState.from_numpy(ket)
        dev.set_option(_lib.OPT_READOUT_VARIANT, variant)
        dev.permute(order)
        names.add(dev.last_kernel())
        assert np.array_equal(dev.to_numpy(), O.permute_qubits(ket, order)), order
    for k in (3, 4, 6):
        qs = [int(v) for v in rng.choice(n, size=k, replace=False)]
        d = np.exp(1j * rng.uniform(0, 6.28, 1 << k))
        dev = DeviceState.from_numpy(ket)
        dev.set_option(_lib.OPT_READOUT_VARIANT, variant)
        dev.apply_matrix(np.diag(d), qs)
        names.add(dev.last_kernel())
        assert maxdiff(dev.to_numpy(), O.apply_gate(ket, np.diag(d), qs)) < GATE_TOL
    streaming = {"k_measure_probs_s<true>", "k_measure_probs_s<false>", "k_collapse_s<true>", "k_collapse_s<false>",
                 "k_insert_s<true>", "k_insert_s<false>", "k_permute_s", "k_diag_table_s"}
    plain = {"k_measure_probs", "k_collapse", "k_insert", "k_permute", "k_diag_table"}
    assert names == (streaming if variant == 0 else plain), names


def test_reduced_density_matrices_and_density_registers(golden):
    """f2: reduced density matrices in one read pass (k_rdm on the matrix cores from 14 qubits up, k_rdm_small below),
    fidelity / purity with the operands in HBM, density matrices as device registers."""
    from quantum_computations_amd.device import DensityState
    from quantum_computations_amd.dv_simulator import numpy_quantum as npq
    g = golden["dv_readout"]
    for case in golden.cases("dv_readout"):                       # what the reference itself computed
        tag, v = case["tag"], case["values"]
        a, b, rho, sigma = (g[f"{tag}_{k}"] for k in ("a", "b", "rho", "sigma"))
        da, db = DeviceState.from_numpy(a), DeviceState.from_numpy(b)
        drho, dsigma = DensityState.from_numpy(rho), DensityState.from_numpy(sigma)
        assert abs(npq.fidelity(da, db) - v["ket_ket"]) < 1e-13
        assert abs(npq.fidelity(da, drho) - v["ket_dm"]) < 1e-13 and abs(npq.fidelity(a, drho) - v["ket_dm"]) < 1e-13
        assert abs(npq.fidelity(dsigma, db) - v["dm_ket"]) < 1e-13 and abs(npq.fidelity(dsigma, b) - v["dm_ket"]) < 1e-13
        assert abs(npq.fidelity(drho, dsigma) - v["dm_dm"]) < 1e-10
        assert abs(npq.purity(drho) - v["purity_rho"]) < 1e-13 and abs(npq.purity(dsigma) - v["purity_sigma"]) < 1e-13
        for kept in case["kept"]:
            want = g[f"{tag}_rdm_{'_'.join(map(str, kept))}"]
            assert maxdiff(da.reduced_density(kept), want) < 1e-14, (tag, kept)
        assert da.last_kernel() == "k_rdm_small"
    # registers large enough for the matrix-core kernel: every k, kept qubits anywhere, in any order
    # (round 3: the LDS-staged workgroup tile k_rdm_tile is the shipped form; variant 2 keeps round 2's k_rdm)
    rng = np.random.default_rng(8)
    for n in (14, 16, 19):
        ket = W.random_ket(n, 80 + n)
        dev = DeviceState.from_numpy(ket)
        for variant, name in ((0, "k_rdm_tile"), (2, "k_rdm")):
            dev.set_option(_lib.OPT_READOUT_VARIANT, variant)
            for k in range(1, 7):
                for trial in range(5):
                    kept = [int(q) for q in rng.choice(n, size=k, replace=False)]
                    if trial == 0:
                        kept = list(range(n - k, n))                  # the lowest index bits: rows inside a cache line
                    elif trial == 1:
                        kept = list(range(k))[::-1]                   # the top bits, reversed
                    elif trial == 2:
                        kept = [n - 1 - b for b in sorted(rng.choice(np.arange(6, n), size=k, replace=False))]   # all from bit 6
                    got = dev.reduced_density(kept)
                    assert dev.last_kernel() == f"{name}<{1 if k <= 4 else 2 if k == 5 else 4}>", (variant, dev.last_kernel())
                    assert maxdiff(got, O.reduced_density(ket, kept)) < 1e-14, (n, kept, variant)
                    assert abs(np.trace(got).real - 1.0) < 1e-13 and maxdiff(got, got.conj().T) == 0.0
            assert np.array_equal(dev.reduced_density([3, 1]), dev.reduced_density([3, 1]))      # deterministic sums
        dev.set_option(_lib.OPT_READOUT_VARIANT, 0)
    with pytest.raises(ValueError):
        dev.reduced_density(list(range(7)))
    with pytest.raises(ValueError):
        dev.reduced_density([0, 0])
    # a density matrix kept on the device through a circuit: U rho U^dagger gate by gate (gates.py:51-52)
    n = 5
    kets = [W.random_ket(n, s) for s in (1, 2)]
    rho = 0.6 * npq.ket2dm(kets[0]) + 0.4 * npq.ket2dm(kets[1])
    dev = DensityState.from_numpy(rho)
    assert dev.ndim == 2 and dev.shape == (32, 32) and dev.num_qubits == n
    ops = W.random_circuit(n, 30, 5)
    out = dev
    for gate in W.to_gates(ops):
        out = gate.apply(out)
        assert out is dev
    want = 0.6 * npq.ket2dm(O.run_circuit(ops, kets[0])[0]) + 0.4 * npq.ket2dm(O.run_circuit(ops, kets[1])[0])
    assert maxdiff(dev.to_numpy(), want) < CIRCUIT_TOL
    assert abs(dev.purity() - npq.purity(want)) < 1e-12
    target = DeviceState.from_numpy(O.run_circuit(ops, kets[0])[0])
    assert abs(npq.fidelity(target, dev) - npq.fidelity(O.run_circuit(ops, kets[0])[0], want)) < 1e-12
    clone = dev.copy()
    G.X(0).apply(clone)
    assert isinstance(clone, DensityState) and maxdiff(dev.to_numpy(), want) < CIRCUIT_TOL


def test_permute_matches_reference_convention(golden):
    g = golden["dv_expand_gate"]
    for case in golden.cases("dv_expand_gate"):
        if "order" in case:
            dev = DeviceState.from_numpy(g["perm_in"])
            dev.permute(case["order"])
            assert np.array_equal(dev.to_numpy(), g[case["label"]]), case
    ket = W.random_ket(11, 4)
    order = [int(v) for v in np.random.default_rng(0).permutation(11)]
    dev = DeviceState.from_numpy(ket)
    dev.permute(order)
    assert np.array_equal(dev.to_numpy(), O.permute_qubits(ket, order))


def test_readout_helpers():
    ket = W.random_ket(12, 8)
    dev = DeviceState.from_numpy(ket * 1.5)
    assert abs(dev.norm2() - 2.25) < 1e-12
    idx = [0, 5, 4095, 1234]
    assert maxdiff(dev.probabilities(idx), np.abs(1.5 * ket[idx]) ** 2) < 1e-15
    other = DeviceState.from_numpy(W.random_ket(12, 9))
    assert abs(dev.inner(other) - np.vdot(1.5 * ket, W.random_ket(12, 9))) < 1e-13
    clone = dev.copy()
    G.H(3).apply(dev)
    assert np.array_equal(clone.to_numpy(), ket * 1.5)
    assert np.array_equal(dev.download(16, 32), dev.to_numpy()[16:48])


@pytest.mark.parametrize("n", [3, 7, 13])
def test_pauli_expectations_and_sampling(n):
    """Device-side read-out: <psi|P|psi> against the dense Pauli operator built from npq.PAULIS (the reference's
    npq.expect on a Kronecker product), and inverse-CDF sampling against the host cumulative distribution."""
    from quantum_computations_amd.dv_simulator import numpy_quantum as npq
    rng = np.random.default_rng(n)
    ket = W.random_ket(n, 60 + n)
    dev = DeviceState.from_numpy(ket)
    mats = {"I": npq.IDTY, "X": npq.X, "Y": npq.Y, "Z": npq.Z}
    for trial in range(12):
        k = int(rng.integers(1, min(n, 5) + 1))
        qubits = [int(q) for q in rng.choice(n, size=k, replace=False)]
        letters = "".join(rng.choice(list("IXYZ"), size=k))
        got = dev.expect_pauli(letters, qubits)
        phi = ket
        for letter, q in zip(letters, qubits):
            phi = O.apply_gate(phi, mats[letter], [q])
        want = np.vdot(ket, phi)
        assert abs(got - want) < 1e-13, (letters, qubits)
        assert abs(got.imag) < 1e-13                     # Pauli strings are hermitian
    with pytest.raises(ValueError):
        dev.expect_pauli("Q", [0])
    # sampling: identical uniforms -> identical outcomes as the host inverse CDF (away from rounding ties)
    probs = np.abs(ket) ** 2
    cdf = np.cumsum(probs)
    u = rng.random(2000)
    want_idx = np.searchsorted(cdf, u * cdf[-1], side="right")
    margin = np.minimum(np.abs(cdf[np.minimum(want_idx, len(cdf) - 1)] - u * cdf[-1]),
                        np.abs(np.concatenate([[0.0], cdf])[want_idx] - u * cdf[-1]))
    got_idx = dev.sample(len(u), rng=np.random.default_rng(0))           # API shape
    assert got_idx.dtype == np.uint64 and got_idx.max() < (1 << n)
    import ctypes as C
    out = np.empty(len(u), dtype=np.uint64)
    _lib.call("qsv_sample", dev._h, len(u), u.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    safe = margin > 1e-12
    assert safe.sum() > 1900 and np.array_equal(out[safe], want_idx[safe].astype(np.uint64))
    # statistics: many shots reproduce the distribution of a small register
    if n == 3:
        shots = dev.sample(200000, rng=np.random.default_rng(5))
        freq = np.bincount(shots.astype(np.int64), minlength=8) / 200000
        assert np.max(np.abs(freq - probs)) < 5e-3
    # a basis state is sampled with certainty
    dev.set_basis((1 << n) - 2)
    assert np.all(dev.sample(50, rng=np.random.default_rng(1)) == (1 << n) - 2)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def test_device_random_fill_is_counter_based():
    n, seed = 10, 12345
    dev = DeviceState.zeros(n)
    norm2 = dev.fill_random(seed, index_offset=0, normalise=False)
    got = dev.to_numpy()
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed))
        g = np.arange(1 << n, dtype=np.uint64)
        r1, r2 = _splitmix64(key ^ (np.uint64(2) * g)), _splitmix64(key ^ (np.uint64(2) * g + np.uint64(1)))
    u1 = ((r1 >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53
    u2 = ((r2 >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53
    rad = np.sqrt(-2.0 * np.log(u1))
    want = rad * np.cos(2 * np.pi * u2) + 1j * rad * np.sin(2 * np.pi * u2)
    assert maxdiff(got, want) < 1e-12
    assert abs(norm2 - np.sum(np.abs(want) ** 2)) < 1e-9
    # a shard filled with an offset equals the matching slice of the whole
    shard = DeviceState.zeros(n - 2)
    shard.fill_random(seed, index_offset=3 << (n - 2), normalise=False)
    assert np.array_equal(shard.to_numpy(), got[3 << (n - 2):])


def test_errors_map_to_reference_exception_classes():
    dev = DeviceState.zeros(3)
    with pytest.raises(ValueError):
        G.H(3).apply(dev)                       # index out of range
    with pytest.raises(ValueError):
        dev.apply_matrix(np.identity(4), [1, 1])  # duplicate
    with pytest.raises(ValueError):
        G.Gate([0, 0], np.identity(4))
    with pytest.raises(ValueError):
        G.Gate([-1], np.identity(2))
    with pytest.raises(ValueError):
        G.M(0, 0.0, 0.0).apply(np.ones((2, 2, 2)))
    with pytest.raises(ValueError):
        G.H(0).apply(np.ones((2, 2, 2)))
    with pytest.raises(ValueError):
        G.M(0, 0.0, 0.0, result=2)
    with pytest.raises(TypeError):
        Simulator([]).run("000")
    with pytest.raises(ValueError):
        DeviceState.from_numpy(np.ones(6))


# ---- full-size properties (BASELINE config 2: 28 qubits, 4 GiB) ----------------------------------------------
def test_full_size_28q_round_trip_and_known_answers():
    n = 28
    dev = DeviceState.random(n, seed=28)
    assert abs(dev.norm2() - 1.0) < 1e-10
    ref = dev.copy()
    ops = W.random_circuit(n, 24, 2028)
    gates = W.to_gates(ops)
    for gate in gates:
        gate.apply(dev)
    assert abs(dev.norm2() - 1.0) < 1e-10                # unitarity
    overlap = dev.inner(ref)
    assert abs(overlap) < 0.99                           # the circuit did something
    for o in reversed(ops):                              # U^dagger in reverse order restores the state
        dev.apply_matrix(np.conjugate(np.asarray(o["matrix"])).T, o["indices"])
    assert abs(dev.inner(ref) - 1.0) < 1e-10
    probe = np.random.default_rng(0).integers(0, 1 << n, 64)
    assert maxdiff(dev.probabilities(probe), ref.probabilities(probe)) < 1e-18
    # known answer: H on every qubit of |0...0> is the uniform superposition; one more layer returns |0...0>
    dev.set_basis(0)
    for q in range(n):
        G.H(q).apply(dev)
    assert maxdiff(dev.probabilities(probe), np.full(64, 2.0 ** -n)) < 1e-22
    for q in range(n):
        G.H(q).apply(dev)
    assert abs(dev.probabilities([0])[0] - 1.0) < 1e-12
    # GHZ: H(0) then a CX chain; only |0..0> and |1..1> are populated
    G.H(0).apply(dev)
    for q in range(n - 1):
        G.CX(q, q + 1).apply(dev)
    p = dev.probabilities([0, (1 << n) - 1, 12345])
    assert abs(p[0] - 0.5) < 1e-12 and abs(p[1] - 0.5) < 1e-12 and p[2] < 1e-24
    # measuring qubit 5 of the GHZ state with outcome 1 leaves |1...1> on 27 qubits
    out_state, s = G.MZ(5, result=1).apply(dev)
    assert s == 1 and out_state.num_qubits == n - 1
    assert abs(out_state.probabilities([(1 << (n - 1)) - 1])[0] - 1.0) < 1e-12


def test_full_size_28q_multi_qubit_kernels_round_trip_and_spot_check():
    """The k = 3..6 kernels at the benchmark's register size (k_dense_tile, k_dense_lds, k_dense_big, k_dense_mfma<6>;
    complex and real matrices; targets high, low and mixed): unitarity, U then U^dagger restores the state, and after
    the forward pass a sample of amplitudes agrees with the oracle formula out[r] = sum_c U[r, c] in[c] evaluated from
    the input amplitudes of the sampled groups alone (size-independent: 2^k inputs per sampled output)."""
    n = 28
    rng = np.random.default_rng(2828)
    dev = DeviceState.random(n, seed=29)
    ref = dev.copy()
    cases = [[8, 11, 14], [0, 1, 2], [3, 4, 5, 7], [0, 7, 11, 15], [20, 21, 22, 23], [8, 11, 14, 17, 20],
             [0, 1, 2, 3, 4], [3, 9, 15, 24, 26], [5, 6, 7, 8, 9], [0, 3, 7, 12, 19, 26], [8, 11, 14, 17, 20, 23]]
    applied, kernels = [], set()
    for i, bits in enumerate(cases):
        k = len(bits)
        qs = [n - 1 - b for b in bits]
        u = W.haar_unitary(1 << k, rng) if i % 2 == 0 else np.linalg.qr(rng.standard_normal((1 << k, 1 << k)))[0]
        # spot check: 6 random groups; input amplitudes before, outputs after
        others = [b for b in range(n) if b not in bits]
        bases = []
        for _ in range(6):
            v = int(rng.integers(0, 1 << (n - k)))
            bases.append(sum(((v >> j) & 1) << b for j, b in enumerate(others)))
        def members(base):
            # index of the basis state with the target qubits spelling c (qs[0] most significant, as in O.apply_gate)
            return [base | sum(((c >> (k - 1 - leg)) & 1) << bits[leg] for leg in range(k)) for c in range(1 << k)]
        idx = [j for base in bases for j in members(base)]
        before = np.array([dev.download(j, 1)[0] for j in idx])
        dev.apply_matrix(u, qs)
        kernels.add(dev.last_kernel().split("<")[0])
        after = np.array([dev.download(j, 1)[0] for j in idx])
        for g in range(len(bases)):
            sl = slice(g << k, (g + 1) << k)
            assert maxdiff(after[sl], u @ before[sl]) < 1e-15, (bits, dev.last_kernel())
        applied.append((u, qs))
    assert abs(dev.norm2() - 1.0) < 1e-10
    assert abs(dev.inner(ref)) < 0.99
    assert {"k_dense_tile", "k_dense_lds", "k_dense_big", "k_dense_mfma"} <= kernels, kernels
    for u, qs in reversed(applied):
        dev.apply_matrix(np.conjugate(u).T, qs)
    assert abs(dev.inner(ref) - 1.0) < 1e-10


def test_staging_ring_with_the_host_far_ahead_of_the_gpu():
    """k = 3..6 gates, diagonals and phases queued back to back on a 24-qubit register without any host wait: each
    kernel takes longer than the host needs to prepare the next gate, so the host laps the 8-slot staging ring of gate
    matrices many times while the GPU is still busy.  A slot overwritten before its kernel has read it would apply a
    wrong matrix: U_1 .. U_m followed by U_m^dagger .. U_1^dagger must restore the state, and a prefix must agree with
    the same gates applied one by one with a synchronisation after each."""
    n = 24
    rng = np.random.default_rng(2424)
    dev = DeviceState.random(n, seed=24)
    ref = dev.copy()
    seq = []
    for i in range(120):
        k = int(rng.integers(3, 7))
        qs = [int(q) for q in rng.choice(n, size=k, replace=False)]
        if i % 5 == 4:
            u = np.diag(np.exp(1j * rng.uniform(0, 2 * np.pi, 1 << k)))        # table diagonal
        elif i % 2:
            u = np.linalg.qr(rng.standard_normal((1 << k, 1 << k)))[0]
        else:
            u = W.haar_unitary(1 << k, rng)
        seq.append((u, qs))
    slow = ref.copy()
    for u, qs in seq[:24]:
        slow.apply_matrix(u, qs)
        slow.sync()
    for i, (u, qs) in enumerate(seq):
        dev.apply_matrix(u, qs)                         # no sync: the host runs ahead
        if i == 23:
            mid = dev.copy()
    assert abs(mid.inner(slow) - 1.0) < 1e-10
    for u, qs in reversed(seq):
        dev.apply_matrix(np.conjugate(u).T, qs)
    assert abs(dev.inner(ref) - 1.0) < 1e-9
    assert abs(dev.norm2() - 1.0) < 1e-10


def test_33_qubit_register_uses_64_bit_indices():
    """128 GiB register: amplitude indices beyond 2^32 (the 34-qubit config shards to 2^31 amplitudes per GPU; this
    exercises the same index arithmetic on one device).  Known answers only -- no host copy of the state exists."""
    n = 33
    try:
        dev = DeviceState.zeros(n)
    except MemoryError:
        pytest.skip("not enough free HBM for a 128 GiB register")
    top = 1 << (n - 1)
    G.X(0).apply(dev)                                   # |10...0>: index 2^32
    assert dev.probabilities([top])[0] == 1.0 and dev.probabilities([0])[0] == 0.0
    G.CX(0, n - 1).apply(dev)                           # control on bit 32 flips bit 0
    assert dev.probabilities([top + 1])[0] == 1.0
    G.SWAP(0, 5).apply(dev)                             # bit 32 <-> bit 27
    assert dev.probabilities([(1 << 27) + 1])[0] == 1.0
    G.CZ(5, n - 1).apply(dev)                           # both bits set: global phase -1, probabilities unchanged
    G.SWAP(5, 0).apply(dev)
    assert dev.probabilities([top + 1])[0] == 1.0
    for q in range(n):
        G.H(q).apply(dev)
    probe = [0, 1, top, top + 1, (1 << n) - 1, top + 123456789, 3 << 30]
    assert maxdiff(dev.probabilities(probe), np.full(len(probe), 2.0 ** -n)) < 1e-24
    assert abs(dev.norm2() - 1.0) < 1e-10
    u = W.haar_unitary(4, np.random.default_rng(3))
    dev.apply_matrix(u, [0, n - 1])                     # dense 2q on the highest and the lowest bit
    dev.apply_matrix(np.conjugate(u).T, [0, n - 1])
    assert maxdiff(dev.probabilities(probe), np.full(len(probe), 2.0 ** -n)) < 1e-24
    for q in range(n):
        G.H(q).apply(dev)
    assert abs(dev.probabilities([top + 1])[0] - 1.0) < 1e-10
    s_state, bit = G.MZ(0, result=1).apply(dev)          # 2^32 -> 2^31 amplitudes
    assert bit == 1 and s_state.num_qubits == n - 1 and abs(s_state.probabilities([1])[0] - 1.0) < 1e-10
    dev.close()


# ---- sharded registers with the real HIP engine (ranks share the one GPU; collectives staged over gloo) -------
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_state_on_one_gpu(world):
    from test_distributed_gloo import run_workers
    # staging pieces of 64 amplitudes: every exchange runs the multi-slice, double-buffered loop on device tensors
    # (slicing of the shard, two-slice staging buffer, copies into place); only the send/recv itself is host-staged
    # round 3: the same worker also holds low-bit gates back and applies them slice by slice inside the exchange steps
    # (one view handle re-pointed at slice after slice: qsv_rebind_view), 256-amplitude staging pieces
    out = run_workers(world, "--backend", "gloo-gpu", "--qubits", "12", "--chunk-amps", "64", "--overlap-chunk-amps", "256")
    assert f"dist_worker ok: world={world} backend=gloo-gpu" in out and "chunk_amps=64" in out
    line = [l for l in out.splitlines() if l.startswith("overlap ok")][0]
    assert int(line.split("gates_in_exchanges=")[1].split()[0]) > 0


def test_shipped_collectives_on_rccl_with_one_rank():
    """RCCL refuses two ranks on one device; a one-rank world still runs the shipped ``_p2p`` (grouped ncclSend /
    ncclRecv of (re, im) views of shard slices), all-reduce and all-gather on HBM tensors over the real backend."""
    import subprocess
    import sys
    worker = Path(__file__).resolve().parent / "rccl_self_worker.py"
    r = subprocess.run([sys.executable, str(worker)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl self ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("n,world", [(30, 2), (32, 2), (31, 4)])
def test_large_sharded_register_at_the_production_piece_size(n, world):
    """Two ranks on the one GPU with 8 GiB and 32 GiB shards (32 GiB is the shard of BASELINE config 3: 34 qubits on 8
    GPUs): the half shard travels in 4 / 16 pieces of 1 GiB through the two-piece staging buffer (device tensors beyond
    2^31 bytes, the default chunk size); with four ranks all rank bits are exchanged at once (three peers, 256 MiB
    slices).  Known answers and circuit + inverse; no CPU oracle can hold these registers."""
    import os
    import subprocess
    import sys
    from test_distributed_gloo import free_port
    env = dict(os.environ, OMP_NUM_THREADS="4", MASTER_ADDR="127.0.0.1")
    env.pop("QSV_EXCHANGE_CHUNK_AMPS", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(Path(__file__).resolve().parent / "dist_big_worker.py"), "--qubits", str(n)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-3000:] + "\n" + proc.stderr[-3000:]
    assert f"dist_big_worker ok: n={n}" in proc.stdout


# ---- d-level modes --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_modes,d", [(1, 32), (3, 8), (4, 5), (3, 32), (2, 70)])
def test_mode_gates_against_tensordot(n_modes, d):
    rng = np.random.default_rng(d)
    psi = rng.standard_normal((d,) * n_modes) + 1j * rng.standard_normal((d,) * n_modes)
    psi /= np.linalg.norm(psi)
    st = QuditState.from_numpy(psi)
    want = psi
    for mode in range(n_modes):
        m = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        m /= np.linalg.norm(m, 2)
        st.apply_mode(m, mode)
        want = CO.apply_axis(want, m, mode)
        diag = np.exp(1j * rng.uniform(0, 6.28, d))
        st.apply_mode(diag, mode)
        want = CO.apply_axis_diag(want, diag, mode)
    assert maxdiff(st.to_numpy(), want) < CIRCUIT_TOL
    if n_modes >= 2 and d <= 8:
        for m0, m1 in [(0, 1), (1, 0), (0, n_modes - 1), (n_modes - 1, 0)]:
            if m0 == m1:
                continue
            g = rng.standard_normal((d * d, d * d)) + 1j * rng.standard_normal((d * d, d * d))
            g /= np.linalg.norm(g, 2)
            st.apply_two_mode(g, m0, m1)
            want = CO.apply_two_axes(want, g, m0, m1)
            plane = np.exp(1j * rng.uniform(0, 6.28, (d, d)))
            st.apply_two_mode(plane, m0, m1)
            want = CO.apply_two_axes_diag(want, plane, m0, m1)
        assert maxdiff(st.to_numpy(), want) < CIRCUIT_TOL
