"""Gate fusion is an exact rewrite of the circuit: checked on the host against the oracle (no GPU needed), and on the
GPU through ``Simulator(fuse=k)`` against the reference's golden final states."""
from __future__ import annotations

import numpy as np
import pytest

from fixture_io import unpack_ops
from oracle import dv_oracle as O
from quantum_computations_amd import workloads as W
from quantum_computations_amd.dv_simulator import gates as G
from quantum_computations_amd.dv_simulator.simulator import ClassicalControl
from quantum_computations_amd.fusion import fuse_circuit, fusion_stats


def oracle_run(gates, ket):
    for g in gates:
        ket = O.apply_gate(ket, g.matrix, g.indices)
    return ket


@pytest.mark.parametrize("max_qubits", [2, 3, 4, 5])
@pytest.mark.parametrize("n,seed", [(5, 1), (9, 2), (12, 3)])
def test_fused_circuit_is_equivalent(n, seed, max_qubits):
    ops = W.random_circuit(n, 120, seed)
    gates = W.to_gates(ops)
    fused = fuse_circuit(gates, max_qubits)
    assert all(len(g.indices) <= max_qubits for g in fused)
    assert sum(len(getattr(g, "sources", [g])) for g in fused) == len(gates)      # nothing lost, nothing doubled
    ket = W.random_ket(n, seed)
    assert np.max(np.abs(oracle_run(fused, ket) - oracle_run(gates, ket))) < 1e-12
    stats = fusion_stats(gates, fused)
    assert stats["launches"] <= len(gates)
    if max_qubits >= 3 and n <= 9:
        assert stats["gates_per_launch"] > 1.5


def test_single_gate_blocks_keep_their_class():
    gates = [G.CZ(0, 1), G.H(5), G.CX(2, 3)]            # pairwise disjoint: nothing to merge below 4 qubits
    fused = fuse_circuit(gates, 2)
    assert sorted(type(g).__name__ for g in fused) == ["CX", "CZ", "H"]
    fused = fuse_circuit([G.H(0), G.T(0), G.CZ(0, 1)], 2)
    assert len(fused) == 1 and fused[0].indices == [0, 1] and len(fused[0].sources) == 3
    assert fuse_circuit(gates, 0) == gates and fuse_circuit(gates, 1) == gates


def test_barriers_split_blocks():
    circuit = [G.H(0), G.CX(0, 1), G.MZ(0, result=1), G.H(0), ClassicalControl(G.X(0), [0]), G.T(0), G.Z(0),
               G.Insert(1, G.State.PLUS), G.H(1)]
    fused = fuse_circuit(circuit, 4)
    kinds = [type(g).__name__ for g in fused]
    assert kinds == ["Gate", "MZ", "H", "ClassicalControl", "Gate", "Insert", "H"]
    assert [len(getattr(g, "sources", [g])) for g in fused] == [2, 1, 1, 1, 2, 1, 1]


def test_order_of_gates_sharing_a_qubit_is_kept():
    rng = np.random.default_rng(0)
    n = 6
    gates = []
    for _ in range(60):
        q = [int(v) for v in rng.choice(n, size=2, replace=False)]
        gates.append(G.Gate(q, W.haar_unitary(4, rng)))
        gates.append(G.Gate([int(rng.integers(n))], W.haar_unitary(2, rng)))
    ket = W.random_ket(n, 9)
    for k in (2, 3, 4):
        assert np.max(np.abs(oracle_run(fuse_circuit(gates, k), ket) - oracle_run(gates, ket))) < 1e-12


def test_rank_bits_a_block_only_conserves_are_free_and_never_pulled_in():
    """Fusion on a sharded register (``remote`` = qubits on rank bits): a remote control / diagonal leg does not count
    towards the block size, and a gate that mixes a remote qubit does not join a block that merely conserved it."""
    rng = np.random.default_rng(3)
    u4 = lambda: W.haar_unitary(4, rng)
    # qubit 0 is remote and only ever a control: 4 local legs + the free remote one fit a 4-qubit cap
    gates = [G.CZ(0, 3), G.Gate([3, 4], u4()), G.CX(0, 5), G.Gate([5, 6], u4()), G.Gate([4, 5], u4())]
    fused = fuse_circuit(gates, 4, remote=[0])
    assert len(fused) == 1 and sorted(fused[0].indices) == [0, 3, 4, 5, 6]
    assert len(fuse_circuit(gates, 4)) > 1                       # without the hint the union is 5 > 4 qubits
    # H(0) mixes the rank bit: it must not drag the block (which runs without any exchange) behind an exchange
    fused = fuse_circuit(gates + [G.H(0), G.CZ(0, 3)], 5, remote=[0])
    assert [sorted(g.indices) for g in fused] == [[0, 3, 4, 5, 6], [0, 3]]
    assert [len(getattr(g, "sources", [g])) for g in fused] == [5, 2]
    # never more legs than the library's k-qubit entry point takes, however many of them are free
    many = [G.CZ(r, 4 + r) for r in range(4)] + [G.Gate([4, 5], u4()), G.Gate([6, 7], u4()), G.Gate([5, 6], u4())]
    assert all(len(g.indices) <= 6 for g in fuse_circuit(many, 5, remote=[0, 1, 2, 3]))
    # and it stays an exact rewrite
    n = 8
    ops = W.random_circuit(n, 150, 17)
    gates = W.to_gates(ops)
    ket = W.random_ket(n, 17)
    for remote in ([0], [0, 1, 2]):
        fused = fuse_circuit(gates, 5, n_qubits=n, remote=remote)
        assert sum(len(getattr(g, "sources", [g])) for g in fused) == len(gates)
        assert np.max(np.abs(oracle_run(fused, ket) - oracle_run(gates, ket))) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [2, 3, 4, 5])
def test_simulator_with_fusion_matches_reference(golden, fuse):
    from quantum_computations_amd.dv_simulator.simulator import Simulator
    g = golden["dv_random_circuits"]
    for case in golden.cases("dv_random_circuits"):
        tag = case["tag"]
        ops = unpack_ops(g[f"meta_{tag}"], g[f"mats_{tag}"])
        sim = Simulator(W.to_gates(ops), fuse=fuse)
        final = sim.run(g[f"init_{tag}"])
        assert np.max(np.abs(final - g[f"final_{tag}"])) < 1e-12, (tag, fuse)
        assert len(sim.launch_list) < len(ops)
    # measurements, insertions and classical control still behave (they are fusion barriers)
    gm = golden["dv_measure_insert"]
    for case in golden.cases("dv_measure_insert"):
        if case["kind"] == "control":
            from quantum_computations_amd.dv_simulator.states import State
            ops = unpack_ops(gm[f"{case['key']}_meta"], gm[f"{case['key']}_mats"], {s.name: s.get() for s in State})
            sim = Simulator(W.to_gates(ops), fuse=fuse)
            out = sim.run()
            assert sim.results == case["results"] and np.max(np.abs(out - gm[case["key"]])) < 1e-12


@pytest.mark.gpu
def test_fused_28_qubit_circuit_agrees_with_unfused_and_with_the_oracle_block_by_block():
    from quantum_computations_amd.device import DeviceState
    from quantum_computations_amd.dv_simulator.simulator import Simulator
    n = 28
    gates = W.to_gates(W.random_circuit(n, 100, 100))
    a = DeviceState.random(n, 28)
    b = a.copy()
    Simulator(gates).run(a)
    sim = Simulator(gates, fuse=4)
    sim.run(b)
    assert len(sim.launch_list) < 60
    assert abs(a.inner(b) - 1.0) < 1e-10
    probe = np.random.default_rng(1).integers(0, 1 << n, 128)
    assert np.max(np.abs(a.probabilities(probe) - b.probabilities(probe))) < 1e-18
    # and against the ORACLE at full size, block by block (round 3; the two device paths above could share a mistake):
    # every fused block's matrix is rebuilt from its source gates with oracle.dv_oracle.apply_gate, and after the block
    # has run, sampled groups of the 4 GiB register must equal that matrix times the group's amplitudes before it
    from oracle import dv_oracle as O
    rng = np.random.default_rng(2)
    c = DeviceState.random(n, 29)
    for block in sim.launch_list:
        qs = list(block.indices)
        k = len(qs)
        sources = getattr(block, "sources", [block])
        m = np.identity(1 << k, dtype=complex)
        for col in range(1 << k):
            ket = m[:, col].copy()
            for g in sources:
                ket = O.apply_gate(ket, np.asarray(g.matrix, dtype=complex), [qs.index(q) for q in g.indices])
            m[:, col] = ket
        bits = [n - 1 - q for q in qs]
        others = [bit for bit in range(n) if bit not in bits]
        bases = []
        for _ in range(3):
            v = int(rng.integers(0, 1 << (n - k)))
            bases.append(sum(((v >> j) & 1) << bit for j, bit in enumerate(others)))
        members = [[base | sum(((col >> (k - 1 - leg)) & 1) << bits[leg] for leg in range(k)) for col in range(1 << k)]
                   for base in bases]
        before = [np.array([c.download(j, 1)[0] for j in group]) for group in members]
        block.apply(c)
        for group, x in zip(members, before):
            after = np.array([c.download(j, 1)[0] for j in group])
            assert np.max(np.abs(after - m @ x)) < 1e-15, (qs, c.last_kernel())


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [3, 5])
def test_fused_24_qubit_circuit_against_the_c_oracle(fuse):
    """Fusion at a size the NumPy oracle cannot reach, against an independent implementation: the C restatement
    (oracle/csrc/qsv_oracle.c) applies the 100 gates one by one on the host, the HIP path applies the fused blocks
    (dense 3..5-qubit kernels, incl. the line-granular one for low target bits); every one of the 2^24 amplitudes is
    compared.  (``bench.py`` repeats this at n = 28 in its cpu_baseline leg.)"""
    from oracle import c_oracle
    from quantum_computations_amd.dv_simulator.simulator import Simulator
    n = 24
    ops = W.random_circuit(n, 100, 100 + fuse)
    ket = W.random_ket(n, 24)
    want = ket.copy()
    for op in ops:
        c_oracle.apply_gate_inplace(want, op["matrix"], op["indices"])
    sim = Simulator(W.to_gates(ops), fuse=fuse)
    got = sim.run(ket)
    assert max(len(g.indices) for g in sim.launch_list) == fuse
    assert np.max(np.abs(got - want)) < 1e-12
