"""Host side of the single-launch executor (no GPU): the program encoding ``dv_simulator.program`` produces, what it
refuses, and that drawing one uniform number per measurement is what ``np.random.choice`` does (gates.py:183)."""
from __future__ import annotations

import numpy as np
import pytest

from quantum_computations_amd.dv_simulator import gates as G
from quantum_computations_amd.dv_simulator import program as P
from quantum_computations_amd.dv_simulator.simulator import ClassicalControl
from quantum_computations_amd.dv_simulator.states import State


def header(word):
    w = int(word)
    return {"op": w & 0xff, "k": (w >> 8) & 0xf, "len": (w >> 12) & 0xffff, "bits": [(w >> (28 + 6 * j)) & 0x3f for j in range(6)]}


def walk(words):
    pc, ops = 0, []
    while True:
        h = header(words[pc])
        ops.append((pc, h))
        if h["op"] == P.OP_END:
            return ops
        pc += h["len"]


def test_encoding_of_a_small_circuit():
    circuit = [G.H(0), G.CX(2, 0), G.MZ(1, result=1), G.Insert(0, State.PLUS), ClassicalControl(G.X(1), [0], []), G.M(2, 0.3, 0.9)]
    prog = P.compile_circuit(circuit, 3)
    assert (prog.n_initial, prog.n_final, prog.n_max, prog.measurements) == (3, 2, 3, 2)
    ops = walk(prog.words)
    assert [h["op"] for _, h in ops] == [P.OP_DENSE, P.OP_DENSE, P.OP_MEASURE, P.OP_INSERT, P.OP_CCTRL, P.OP_DENSE, P.OP_MEASURE, P.OP_END]
    assert ops[0][1]["k"] == 1 and ops[0][1]["bits"][0] == 2                     # qubit 0 of 3 = bit 2
    assert ops[1][1]["k"] == 2 and ops[1][1]["bits"][:2] == [0, 2]               # CX(2, 0): legs on bits 0 and 2
    assert ops[2][1]["bits"][0] == 1                                             # MZ(1) of 3 qubits = bit 1
    assert ops[3][1]["bits"][0] == 2                                             # Insert(0) into 2 qubits: new bit 2
    pc = ops[1][0]
    assert np.array_equal(prog.words[pc + 1:pc + 33].view(np.float64).view(np.complex128).reshape(4, 4), G.CX(2, 0).matrix)
    pc = ops[2][0]
    assert prog.words[pc + 9].view(np.int64) == 1                                # forced outcome
    pc = ops[4][0]
    assert (int(prog.words[pc + 1]), int(prog.words[pc + 2])) == (1, 0)           # fires iff result 0 is 1
    pc = ops[6][0]
    assert prog.words[pc + 9].view(np.int64) == -1 and prog.uniform_slots == [pc + 10]
    assert [n for _, _, n in prog.steps] == [3, 3, 3, 2, 3, 3]


def test_no_op_straddles_a_program_chunk():
    circuit = [G.CX(0, 1)] * 80 + [G.H(0)] * 9          # 33-word and 9-word ops: 2048 is not a multiple of either
    prog = P.compile_circuit(circuit, 2)
    for pc, h in walk(prog.words):
        assert pc // P.CHUNK_WORDS == (pc + h["len"] - 1) // P.CHUNK_WORDS, (pc, h)
    assert any(h["op"] == P.OP_NOP for _, h in walk(prog.words))


@pytest.mark.parametrize("circuit,n", [
    ([G.Gate([0, 1, 2, 3, 4], np.identity(32))], 5),                    # more legs than the executor takes
    ([G.Gate([0], None)], 2),                                           # no matrix: the gate-by-gate path raises
    ([G.H(3)], 3),                                                      # index outside the register
    ([ClassicalControl(G.X(0), [0], [])], 2),                           # control on a result that does not exist
    ([G.MZ(0, result=0), ClassicalControl(G.MZ(0), [0], [])], 2),       # control around a measurement
    ([G.Insert(0, State.ZERO)], 13),                                    # grows beyond 13 qubits
    ([G.Gate([0], np.ones((4, 2)))], 2),                                # not square
])
def test_what_the_executor_leaves_to_the_gate_by_gate_path(circuit, n):
    with pytest.raises(P.Unsupported):
        P.compile_circuit(circuit, n)


def test_one_uniform_number_per_measurement_is_what_numpy_choice_draws():
    rng = np.random.default_rng(0)
    for trial in range(200):
        p0 = float(rng.uniform(0, 1))
        p = [p0, 1.0 - p0]
        np.random.seed(trial)
        want = [int(np.random.choice([0, 1], p=p)) for _ in range(5)]
        state_after = np.random.get_state()[1][:8].copy()
        np.random.seed(trial)
        got = [0 if np.random.random_sample() < p[0] / (p[0] + p[1]) else 1 for _ in range(5)]
        assert got == want and np.array_equal(np.random.get_state()[1][:8], state_after)
