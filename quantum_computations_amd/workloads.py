"""Synthetic workloads of BASELINE.json's configs, as neutral op lists.

An *op* is a dict ``{"name": str, "indices": [int, ...], "matrix": ndarray | None, ...}``: ``name`` is a gate
class name of ``dv_simulator.gates`` (``"H"``, ``"CX"``, ...) or ``"U"`` for an arbitrary unitary carried in
``matrix``.  The same list drives the HIP path (``to_gates`` -> ``Simulator``), the CPU oracle
(``oracle.dv_oracle.run_circuit``) and -- in the build container only -- the reference itself
(``tests/golden/generate_golden.py``), so all three see identical circuits.  Generators are seeded and
documented in SURVEY.md section 8d.
"""
from __future__ import annotations

import numpy as np

from .dv_simulator import gates as G
from .dv_simulator import numpy_quantum as npq

_FIXED_1Q = ("I", "X", "Y", "Z", "H", "P", "Pdg", "T", "Tdg")
_FIXED_2Q = ("CX", "CZ", "SWAP")


def op(name: str, *indices: int, matrix: np.ndarray | None = None, **extra) -> dict:
    """Build one op; the matrix of a named gate is taken from the gate class."""
    indices = [int(i) for i in indices]
    if matrix is None and name in _FIXED_1Q + _FIXED_2Q:
        matrix = getattr(G, name)(*indices).matrix
    if name == "RZ":
        matrix = G.RZ(indices[0], extra["angle"]).matrix
    return {"name": name, "indices": indices, "matrix": matrix, **extra}


def to_gates(ops: list[dict]) -> list:
    """Instantiate ``dv_simulator.gates`` objects (what ``Simulator`` consumes) from ops."""
    from .dv_simulator.simulator import ClassicalControl
    from .dv_simulator.states import State

    out = []
    for o in ops:
        name, idx = o["name"], o["indices"]
        if name in _FIXED_1Q + _FIXED_2Q:
            gate = getattr(G, name)(*idx)
        elif name == "RZ":
            gate = G.RZ(idx[0], o["angle"])
        elif name == "M":
            gate = G.M(idx[0], o["theta"], o["phi"], result=o.get("result"))
        elif name == "Insert":
            gate = G.Insert(idx[0], State[o["state"]])
        else:
            gate = G.Gate(list(idx), np.asarray(o["matrix"]))
        ctl = o.get("control")
        if ctl is not None:
            gate = ClassicalControl(gate, list(ctl.get("pos", [])), list(ctl.get("neg", [])))
        out.append(gate)
    return out


def haar_unitary(dim: int, rng: np.random.Generator) -> np.ndarray:
    """Haar-random unitary: QR of a complex Ginibre draw with the phases of R's diagonal fixed."""
    z = (rng.standard_normal((dim, dim)) + 1j * rng.standard_normal((dim, dim))) / np.sqrt(2)
    q, r = np.linalg.qr(z)
    phases = np.diagonal(r) / np.abs(np.diagonal(r))
    return q * phases


def random_circuit(n: int, depth: int, seed: int) -> list[dict]:
    """cfg2/cfg3 generator: each gate is 1-qubit with p = 1/2 (Haar 2x2, uniform target), else
    2-qubit drawn uniformly from {CX, CZ, SWAP, Haar 4x4} on a uniform ordered pair of distinct qubits."""
    rng = np.random.default_rng(seed)
    ops = []
    for _ in range(depth):
        if n < 2 or rng.random() < 0.5:
            q = int(rng.integers(n))
            ops.append(op("U", q, matrix=haar_unitary(2, rng)))
        else:
            q0, q1 = (int(v) for v in rng.choice(n, size=2, replace=False))
            kind = int(rng.integers(4))
            if kind < 3:
                ops.append(op(_FIXED_2Q[kind], q0, q1))
            else:
                ops.append(op("U", q0, q1, matrix=haar_unitary(4, rng)))
    return ops


def random_clifford_circuit(n: int, depth: int, seed: int) -> list[dict]:
    """cfg1 generator, after ``random_circ`` of the reference's randomised-benchmarking script
    (``impact_.../randomised_benchmarking.py:27-49``): gates uniform from {I, H, P, Pdg, CZ, SWAP}; a 1-qubit gate
    lands on a uniform qubit, a 2-qubit gate on a uniform nearest-neighbour pair (i, i+1), i in [0, n-2].
    ``depth`` counts gates here (the reference counts GKP layers, which needs its transpiler)."""
    if n < 2:
        raise ValueError("At least 2 qubits required!")
    rng = np.random.default_rng(seed)
    names = ("I", "H", "P", "Pdg", "CZ", "SWAP")
    ops = []
    for _ in range(depth):
        name = names[int(rng.integers(len(names)))]
        if name in _FIXED_1Q:
            ops.append(op(name, int(rng.integers(n))))
        else:
            i = int(rng.integers(n - 1))
            ops.append(op(name, i, i + 1))
    return ops


def random_ket(n: int, seed: int, chunk_bits: int = 24) -> np.ndarray:
    """cfg2 initial state: normalised complex normal amplitudes from ``default_rng(seed)``, drawn in chunks of
    2^chunk_bits amplitudes (real parts of a chunk first, then its imaginary parts)."""
    rng = np.random.default_rng(seed)
    size = 1 << n
    chunk = min(size, 1 << chunk_bits)
    out = np.empty(size, dtype=np.complex128)
    for start in range(0, size, chunk):
        out.real[start:start + chunk] = rng.standard_normal(chunk)
        out.imag[start:start + chunk] = rng.standard_normal(chunk)
    out /= np.linalg.norm(out)
    return out


# ---- n-qubit Grover (cfg5).  Build-defined generalisation: the reference only has the 3-qubit circuit below ----
class MCPhase:
    """Multiply the amplitudes whose ``qubits`` are all 1 by ``phase`` (a multi-controlled Z for phase = -1), as an
    object with the gate protocol (``indices``, ``apply``) so that circuits containing it can be announced to a sharded
    register.  It conserves every qubit it touches: ``mixing`` is empty, it never moves data between GPUs."""

    matrix = None
    mixing = frozenset()

    def __init__(self, qubits, phase: complex = -1.0):
        self.indices = [int(q) for q in qubits]
        self.phase = complex(phase)

    def __repr__(self):
        return f"MCPhase_{len(self.indices)}({self.phase})"

    def apply(self, state):
        if hasattr(state, "apply_mcphase"):
            return state.apply_mcphase(self.indices, self.phase)
        from .device import DeviceState
        dev = DeviceState.from_numpy(np.asarray(state))
        dev.apply_mcphase(self.indices, self.phase)
        out = dev.to_numpy()
        dev.close()
        return out


def grover_circuit(n: int, marked: int, iterations: int) -> list:
    """``iterations`` Grover iterations as a gate list (what ``grover_iteration`` applies, in the same order)."""
    zeros = [q for q in range(n) if not (marked >> (n - 1 - q)) & 1]
    one = [G.X(q) for q in zeros] + [MCPhase(range(n))] + [G.X(q) for q in zeros]
    one += [G.H(q) for q in range(n)] + [G.X(q) for q in range(n)] + [MCPhase(range(n))]
    one += [G.X(q) for q in range(n)] + [G.H(q) for q in range(n)]
    return [g for _ in range(iterations) for g in one]



def grover_iteration(state, n: int, marked: int) -> None:
    """One Grover iteration on a device register (``DeviceState`` / ``ShardedState``): phase oracle on the basis
    state ``marked`` (X on its zero bits, multi-controlled Z, X back) followed by the diffuser H X (MCZ) X H.
    Same structure as ``dv_circuits.grover`` (``impact_.../dv_circuits.py:50-79``) with its hand-decomposed CCZ
    replaced by one multi-controlled phase on all n qubits (``qsv_apply_mcphase``)."""
    h, x = G.H(0).matrix, G.X(0).matrix
    zeros = [q for q in range(n) if not (marked >> (n - 1 - q)) & 1]
    for q in zeros:
        state.apply_matrix(x, [q])
    state.apply_mcphase(list(range(n)), -1.0)
    for q in zeros:
        state.apply_matrix(x, [q])
    for q in range(n):
        state.apply_matrix(h, [q])
    for q in range(n):
        state.apply_matrix(x, [q])
    state.apply_mcphase(list(range(n)), -1.0)
    for q in range(n):
        state.apply_matrix(x, [q])
    for q in range(n):
        state.apply_matrix(h, [q])


def grover_gate_count(n: int, marked: int) -> int:
    zeros = sum(1 for q in range(n) if not (marked >> (n - 1 - q)) & 1)
    return 2 * zeros + 4 * n + 2


def grover_success_probability(n: int, iterations: int, n_marked: int = 1) -> float:
    """sin^2((2k + 1) asin(sqrt(M / 2^n))): the analytic success probability after k iterations."""
    return float(np.sin((2 * iterations + 1) * np.arcsin(np.sqrt(n_marked / 2.0 ** n))) ** 2)


# ---- Grover (cfg5): the reference pins the 3-qubit instance (impact_.../dv_circuits.py:27-109) -------------
def ccz_ops() -> list[dict]:
    """The reference's 15-gate nearest-neighbour CCZ on qubits (0, 1, 2) (``dv_circuits.py:27-48``)."""
    seq = [("CX", 2, 1), ("Tdg", 1), ("CX", 0, 1), ("T", 1), ("CX", 2, 1), ("Tdg", 1), ("CX", 0, 1), ("T", 1),
           ("T", 2), ("SWAP", 1, 2), ("CX", 0, 1), ("T", 0), ("Tdg", 1), ("CX", 0, 1), ("SWAP", 1, 2)]
    return [op(name, *idx) for name, *idx in seq]


def grover3_oracle_ops(tagged: list[int]) -> list[dict]:
    """The three hard-coded 2-item oracles of ``dv_circuits.oracle`` (``dv_circuits.py:87-109``)."""
    table = {
        (3, 6): [("CZ", 0, 1), ("CZ", 1, 2)],
        (0, 4): [("Z", 1), ("Z", 2), ("CZ", 1, 2)],
        (2, 7): [("Z", 1), ("CZ", 0, 1), ("CZ", 1, 2)],
    }
    key = tuple(sorted(tagged))
    if key not in table:
        raise NotImplementedError("Requested oracle not implemented")
    return [op(name, *idx) for name, *idx in table[key]]


def grover3_ops(tagged: list[int]) -> list[dict]:
    """``dv_circuits.grover(oracle(tagged))`` (``dv_circuits.py:50-79``): one Grover iteration on 3 qubits."""
    h3 = [op("H", q) for q in range(3)]
    x3 = [op("X", q) for q in range(3)]
    inserts = [{"name": "Insert", "indices": [q], "matrix": None, "state": "ZERO", "vector": npq.ZERO}
               for q in range(3)]
    return inserts + h3 + grover3_oracle_ops(tagged) + h3 + x3 + ccz_ops() + x3 + h3
