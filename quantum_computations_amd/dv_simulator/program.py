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
    shift = 28
    for b in bits:
        h |= b << shift
        shift += 6
    return h


def _word(value: int) -> bytes:
    return value.to_bytes(8, "little")


_F64 = np.dtype(np.float64)
_C128 = np.dtype(np.complex128)
_CONSTANT_PAYLOADS: dict[int, tuple] = {}      # id(matrix) -> (matrix, bytes): the fixed matrices of numpy_quantum only


def _payload(matrix: np.ndarray) -> bytes:
    """Row-major complex128 bytes of a gate matrix.  The module-level constants of numpy_quantum (H, X, CX, ...: every
    gate object of those classes shares ONE array) are converted once."""
    hit = _CONSTANT_PAYLOADS.get(id(matrix))
    if hit is not None and hit[0] is matrix:
        return hit[1]
    if matrix.dtype != _C128 or not matrix.flags.c_contiguous:
        matrix = np.ascontiguousarray(matrix, dtype=_C128)
    return matrix.tobytes()


def _register_constants() -> None:
    from . import numpy_quantum as npq

    for value in vars(npq).values():
        if isinstance(value, np.ndarray) and value.ndim == 2 and value.shape[0] == value.shape[1] and value.shape[0] in (2, 4, 8, 16):
            _CONSTANT_PAYLOADS[id(value)] = (value, np.ascontiguousarray(value, dtype=_C128).tobytes())


def compile_circuit(circuit, n_initial: int) -> Program:
    """Encode ``circuit`` for a register of ``n_initial`` qubits.  Pure encoding, about a microsecond per gate: headers
    and payloads are appended to one byte buffer."""
    from .gates import Gate, Insert, M
    from .simulator import ClassicalControl

    if not _CONSTANT_PAYLOADS:
        _register_constants()
    gate_apply = Gate.apply
    n, n_max, measured = n_initial, n_initial, 0
    buf = bytearray()
    slots: list[int] = []
    steps = []
    chunk_bytes = 8 * CHUNK_WORDS

    def room_for(nbytes: int) -> None:
        if nbytes > chunk_bytes:
            raise Unsupported("op larger than a program chunk")
        room = chunk_bytes - len(buf) % chunk_bytes
        if nbytes > room:                          # no op straddles a chunk: pad with one NOP
            buf.extend(_word(_header(OP_NOP, 0, room // 8)))
            buf.extend(bytes(room - 8))

    def matrix_gate(gate, extra: int = 0):
        """(header, payload) of a square gate on 1..4 qubits; ``extra``: bytes emitted in front of it as one unit."""
        matrix = getattr(gate, "matrix", None)
        indices = getattr(gate, "indices", None)
        if type(matrix) is not np.ndarray or matrix.ndim != 2 or type(indices) is not list:
            raise Unsupported("no matrix")
        k = len(indices)
        if k < 1 or k > MAX_GATE_QUBITS or matrix.shape != (1 << k, 1 << k):
            raise Unsupported("not a square gate on 1..4 qubits")
        if matrix.dtype.kind not in "biufc":
            raise Unsupported("matrix dtype")
        bits = []
        for i in indices:
            if type(i) is not int:
                if not isinstance(i, (int, np.integer)):
                    raise Unsupported("indices the reference rejects")
                i = int(i)
            if i < 0 or i >= n or (n - 1 - i) in bits:
                raise Unsupported("indices the reference rejects")
            bits.append(n - 1 - i)
        payload = _payload(matrix)
        words = 1 + len(payload) // 8
        room_for(extra + 8 * words)
        return _word(_header(OP_DENSE, k, words, bits)), payload

    for entry in circuit:
        if type(entry) is ClassicalControl:
            inner = entry.gate
            if isinstance(inner, (M, Insert, ClassicalControl)) or getattr(type(inner), "apply", None) is not gate_apply:
                raise Unsupported("classical control around a size-changing gate")
            pos, neg = list(entry._pos), list(entry._neg)
            if any((not isinstance(i, (int, np.integer))) or i < 0 or i >= min(measured, 64) for i in pos + neg):
                raise Unsupported("control index outside the measurement record")
            head, payload = matrix_gate(inner, extra=24)     # the control word and the gate it guards: one unit
            buf.extend(_word(_header(OP_CCTRL, 0, 3)))
            buf.extend(_word(sum(1 << int(i) for i in set(pos))))
            buf.extend(_word(sum(1 << int(i) for i in set(neg))))
            buf.extend(head)
            buf.extend(payload)
            steps.append((inner, entry, n))
        elif isinstance(entry, M):
            index = entry.indices[0]
            if n < 1 or not 0 <= index < n:
                raise Unsupported("measured qubit outside the register")
            e0, e1 = entry.eigenvectors()
            room_for(88)
            buf.extend(_word(_header(OP_MEASURE, 1, 11, [n - 1 - index])))
            buf.extend(np.ascontiguousarray(e0, dtype=_C128).tobytes())
            buf.extend(np.ascontiguousarray(e1, dtype=_C128).tobytes())
            buf.extend((-1 if entry.result is None else int(entry.result)).to_bytes(8, "little", signed=True))
            if entry.result is None:
                slots.append(len(buf) // 8)
            buf.extend(bytes(8))
            measured += 1
            steps.append((entry, None, n))
            n -= 1
        elif isinstance(entry, Insert):
            index = entry.indices[0]
            if not 0 <= index <= n:
                raise Unsupported("insert position outside the register")
            if n + 1 > MAX_QUBITS:
                raise Unsupported("register grows beyond the executor's size")
            room_for(40)
            buf.extend(_word(_header(OP_INSERT, 1, 5, [n - index])))
            buf.extend(np.ascontiguousarray(entry.matrix[0, :], dtype=_C128).tobytes())
            steps.append((entry, None, n))
            n += 1
            n_max = max(n_max, n)
        else:
            if getattr(type(entry), "apply", None) is not gate_apply:
                raise Unsupported("a gate with its own apply()")
            head, payload = matrix_gate(entry)
            buf.extend(head)
            buf.extend(payload)
            steps.append((entry, None, n))
    room_for(8)
    buf.extend(_word(_header(OP_END, 0, 1)))
    if n_max > MAX_QUBITS:
        raise Unsupported("register too large")
    return Program(np.frombuffer(bytes(buf), dtype=np.uint64), n_initial, n, n_max, measured, slots, steps)


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
                w[slot] = np.float64(np.random.random_sample()).view(np.uint64)
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
