"""Whole circuits in one launch: the gate list of ``Simulator.run`` compiled to the word stream ``qsv_run_programs`` executes.

The reference runs ``for gate in self.circuit: ... gate.apply(state)`` (``simulators/dv_simulator/simulator.py:40-52``) on
registers of 4..12 qubits -- sizes at which a launch per gate costs far more than the gate -- and sweeps thousands of such
circuits through a process pool (``impact_.../randomised_benchmarking.py:60-76``).  Here a circuit becomes a *program*
(format: ``csrc/qsv_circuit.hip``): one workgroup keeps the register in LDS and walks the whole list, measurements,
insertions and classical control included; a batch of programs is one launch, one workgroup per instance.

``compile_circuit`` only *encodes*; anything it cannot express (gates on more than four qubits, objects without a
matrix, controls wrapped around measurements, registers beyond 13 qubits, any input the reference would reject) raises
``Unsupported`` and the caller takes the gate-by-gate path, which produces the reference's own exceptions.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from .. import _lib

MAX_QUBITS = 13
MAX_GATE_QUBITS = 4
CHUNK_WORDS = 2048
OP_END, OP_NOP, OP_DENSE, OP_MEASURE, OP_INSERT, OP_CCTRL = 0, 1, 2, 3, 4, 5


class Unsupported(Exception):
    """The circuit needs the gate-by-gate path."""


@dataclass
class Program:
    words: np.ndarray                      # uint64
    n_initial: int
    n_final: int
    n_max: int
    measurements: int
    uniform_slots: list[int] = field(default_factory=list)   # word index of u01 of every unforced measurement, in order
    steps: list = field(default_factory=list)                 # (gate, control or None, register qubits) per circuit entry


def _header(op: int, k: int, length: int, bits=()) -> int:
    h = op | (k << 8) | (length << 12)
    for j, b in enumerate(bits):
        h |= int(b) << (28 + 6 * j)
    return h


def _doubles(values) -> np.ndarray:
    return np.ascontiguousarray(values, dtype=np.float64).view(np.uint64)


def _complex_words(values) -> np.ndarray:
    return np.ascontiguousarray(values, dtype=np.complex128).reshape(-1).view(np.float64).view(np.uint64)


def compile_circuit(circuit, n_initial: int) -> Program:
    from .gates import Insert, M
    from .simulator import ClassicalControl

    n, n_max, measured = n_initial, n_initial, 0
    out: list[np.ndarray] = []
    length = 0
    slots: list[int] = []
    steps = []

    def emit(words: np.ndarray) -> int:
        nonlocal length
        room = CHUNK_WORDS - length % CHUNK_WORDS
        if len(words) > CHUNK_WORDS:
            raise Unsupported("op larger than a program chunk")
        if len(words) > room:                      # no op straddles a chunk: pad with one NOP
            pad = np.zeros(room, dtype=np.uint64)
            pad[0] = _header(OP_NOP, 0, room)
            out.append(pad)
            length += room
        out.append(words)
        length += len(words)
        return length - len(words)

    def encode_matrix_gate(gate) -> np.ndarray:
        matrix, indices = getattr(gate, "matrix", None), list(getattr(gate, "indices", []))
        if matrix is None or not isinstance(matrix, np.ndarray) or matrix.ndim != 2:
            raise Unsupported("no matrix")
        k = len(indices)
        if k < 1 or k > MAX_GATE_QUBITS or matrix.shape != (1 << k, 1 << k):
            raise Unsupported("not a square gate on 1..4 qubits")
        if len(set(indices)) != k or any((not isinstance(i, (int, np.integer))) or i < 0 or i >= n for i in indices):
            raise Unsupported("indices the reference rejects")
        if not (np.issubdtype(matrix.dtype, np.number) or matrix.dtype == bool):
            raise Unsupported("matrix dtype")
        words = np.empty(1 + 2 * (1 << k) ** 2, dtype=np.uint64)
        words[0] = _header(OP_DENSE, k, len(words), [n - 1 - int(i) for i in indices])
        words[1:] = _complex_words(matrix)
        return words

    for entry in circuit:
        if isinstance(entry, ClassicalControl):
            inner = entry.gate
            if isinstance(inner, (M, Insert)) or isinstance(inner, ClassicalControl):
                raise Unsupported("classical control around a size-changing gate")
            pos, neg = list(entry._pos), list(entry._neg)
            if any((not isinstance(i, (int, np.integer))) or i < 0 or i >= min(measured, 64) for i in pos + neg):
                raise Unsupported("control index outside the measurement record")
            body = encode_matrix_gate(inner)
            ctrl = np.empty(3, dtype=np.uint64)
            ctrl[0] = _header(OP_CCTRL, 0, 3)
            ctrl[1] = sum(1 << int(i) for i in set(pos))
            ctrl[2] = sum(1 << int(i) for i in set(neg))
            # the control word and the gate it guards stay in one chunk: emit them as one unit
            emit(np.concatenate([ctrl, body]))
            steps.append((inner, entry, n))
            continue
        if isinstance(entry, M):
            index = entry.indices[0]
            if n < 1 or not 0 <= index < n:
                raise Unsupported("measured qubit outside the register")
            e0, e1 = entry.eigenvectors()
            words = np.empty(11, dtype=np.uint64)
            words[0] = _header(OP_MEASURE, 1, 11, [n - 1 - index])
            words[1:5] = _complex_words(e0)
            words[5:9] = _complex_words(e1)
            words[9] = np.int64(-1 if entry.result is None else int(entry.result)).view(np.uint64)
            words[10] = _doubles([0.0])[0]
            at = emit(words)
            if entry.result is None:
                slots.append(at + 10)
            measured += 1
            steps.append((entry, None, n))
            n -= 1
            continue
        if isinstance(entry, Insert):
            index = entry.indices[0]
            if not 0 <= index <= n:
                raise Unsupported("insert position outside the register")
            if n + 1 > MAX_QUBITS:
                raise Unsupported("register grows beyond the executor's size")
            words = np.empty(5, dtype=np.uint64)
            words[0] = _header(OP_INSERT, 1, 5, [n - index])
            words[1:5] = _complex_words(entry.matrix[0, :])
            emit(words)
            steps.append((entry, None, n))
            n += 1
            n_max = max(n_max, n)
            continue
        if getattr(type(entry), "apply", None) is not _gate_apply():
            raise Unsupported("a gate with its own apply()")
        emit(encode_matrix_gate(entry))
        steps.append((entry, None, n))
    end = np.array([_header(OP_END, 0, 1)], dtype=np.uint64)
    emit(end)
    if n_max > MAX_QUBITS:
        raise Unsupported("register too large")
    return Program(np.concatenate(out), n_initial, n, n_max, measured, slots, steps)


def _gate_apply():
    from .gates import Gate
    return Gate.apply


def run_programs(programs: list[Program], kets: list[np.ndarray], device: int = 0):
    """Run every program on its ket in ONE launch; returns ``[(final ket, results, (p0, p1) per measurement)]``.
    Unforced measurements take their uniform numbers from the global ``np.random`` state, in order (instance by
    instance, measurement by measurement) -- exactly the draws ``np.random.choice`` would make (gates.py:183)."""
    count = len(programs)
    prog_off = np.zeros(count + 1, dtype=np.uint64)
    state_off = np.zeros(count + 1, dtype=np.uint64)
    out_off = np.zeros(count + 1, dtype=np.uint64)
    res_off = np.zeros(count + 1, dtype=np.uint64)
    words = []
    for i, prog in enumerate(programs):
        w = prog.words
        if prog.uniform_slots:
            w = w.copy()
            for slot in prog.uniform_slots:
                w[slot] = _doubles([np.random.random_sample()])[0]
        words.append(w)
        prog_off[i + 1] = prog_off[i] + len(w)
        state_off[i + 1] = state_off[i] + (1 << prog.n_initial)
        out_off[i + 1] = out_off[i] + (1 << prog.n_final)
        res_off[i + 1] = res_off[i] + prog.measurements
    all_words = np.concatenate(words)
    states_in = np.concatenate([np.ascontiguousarray(k, dtype=np.complex128).reshape(-1) for k in kets])
    if states_in.size != int(state_off[-1]):
        raise ValueError("State has wrong dimensions.")
    states_out = np.empty(int(out_off[-1]), dtype=np.complex128)
    n_meas = int(res_off[-1])
    results = np.zeros(max(n_meas, 1), dtype=np.int32)
    probs = np.zeros(max(2 * n_meas, 2), dtype=np.float64)
    n0 = np.array([p.n_initial for p in programs], dtype=np.int32)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.call("qsv_run_programs", int(device), count, max(p.n_max for p in programs), ptr(all_words), ptr(prog_off), ptr(n0),
              ptr(states_in), ptr(state_off), ptr(states_out), ptr(out_off), ptr(results), ptr(probs), ptr(res_off))
    out = []
    for i, prog in enumerate(programs):
        r = [int(x) for x in results[int(res_off[i]):int(res_off[i + 1])]]
        p = probs[2 * int(res_off[i]):2 * int(res_off[i + 1])].reshape(-1, 2)
        out.append((states_out[int(out_off[i]):int(out_off[i + 1])], r, p))
    return out
