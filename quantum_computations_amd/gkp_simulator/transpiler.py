"""From a qubit circuit to a layered measurement-based GKP circuit (``simulators/gkp_simulator/transpiler.py:10-208``).

A *layer* holds at most one teleportation gadget per qubit plus the Pauli operators queued behind it (Paulis cost nothing:
they are tracked in the frame).  Gates are packed greedily into the earliest layer after the last one that touches any of
their qubits; a ``T`` / ``Tdg`` is followed by a classically controlled ``P`` / ``Pdg`` (the Clifford correction of gate
teleportation, fired by the x bit of the T gadget's by-product).
"""
from __future__ import annotations

from bisect import insort

import numpy as np

from ..cv_simulator.gate_abc import Gate as CVGate  # noqa: F401
from ..cv_simulator.states import State as CVState
from ..dv_simulator import gates as dv_gates
from ..dv_simulator.gates import Gate as DVGate
from ..dv_simulator.simulator import ClassicalControl
from ..dv_simulator.states import State as DVState
from .gates import *  # noqa: F401,F403
from .gates import MBCZ, MBF, MBI, MBP, MBSWAP, MBT, MeasurementBased
from ..cv_simulator.mps import MPS

IMPLEMENTABLES = (dv_gates.I, dv_gates.H, dv_gates.P, dv_gates.Pdg, dv_gates.T, dv_gates.Tdg, dv_gates.CZ, dv_gates.SWAP)
PAULIS = (dv_gates.I, dv_gates.X, dv_gates.Y, dv_gates.Z)

_CODE_WORD = {DVState.ZERO: CVState.GKP_ZERO, DVState.ONE: CVState.GKP_ONE, DVState.PLUS: CVState.GKP_PLUS,
              DVState.MINUS: CVState.GKP_MINUS, DVState.T: CVState.GKP_T, DVState.TDG: CVState.GKP_TDG,
              DVState.H: CVState.GKP_H}
_GADGET = {dv_gates.I: MBI, dv_gates.H: MBF, dv_gates.P: MBP, dv_gates.Pdg: MBP, dv_gates.T: MBT, dv_gates.Tdg: MBT,
           dv_gates.CZ: MBCZ, dv_gates.SWAP: MBSWAP}
_PAULI_BITS = {dv_gates.X: (1, 0), dv_gates.Y: (1, 1), dv_gates.Z: (0, 1)}


def state_transpile(state: DVState) -> CVState:
    return _CODE_WORD.get(state)


def parse_to_mps(state, epsilon: float, qs: np.ndarray, *, device: int = 0) -> MPS:
    """``None`` -> empty register; an ``MPS`` passes through; a list of qubit ``State`` members becomes the product of
    the corresponding GKP code words with damping ``epsilon`` (a matrix-product register on the GPU)."""
    if state is None:
        return MPS(qs, [], device=device, layout="sites")
    if isinstance(state, MPS):
        return state
    if isinstance(state, list) and all(isinstance(item, DVState) for item in state):
        return MPS(qs, [state_transpile(s).eval(qs, epsilon) for s in state], device=device, layout="sites")
    raise TypeError("Unsupported input type")


def gate_transpile(gate: DVGate, **kwargs) -> MeasurementBased:
    """The gadget of a qubit gate; inverse gates select the adjoint gadget (xor an explicit ``dagger=``)."""
    gadget = _GADGET.get(type(gate))
    if gadget is None:
        raise ValueError(f"Gate {gate} not implementable in MB GKP circuits.")
    dagger = (type(gate) in (dv_gates.Pdg, dv_gates.Tdg)) ^ kwargs.pop("dagger", False)
    return gadget(*gate.indices, dagger=dagger, **kwargs)


class Layer:
    def __init__(self, N: int):
        self._N = N
        self._occupied: list[bool] = [False] * N
        self.gates: list = []
        self.paulis: list[list[int]] = [[0, 0] for _ in range(N)]

    def copy(self) -> "Layer":
        twin = Layer(self._N)
        twin.gates, twin.paulis = self.gates.copy(), self.paulis.copy()
        return twin

    def get_gate(self, index: int):
        return next((g for g in self.gates if index in g.indices), None)

    def occupied(self, indices) -> bool:
        """A qubit is taken once it carries a gadget or a queued Pauli (which must stay behind the gadget)."""
        return any(self._occupied[i] or self.paulis[i] != [0, 0] for i in indices)

    def _insert_gate(self, gate) -> None:
        for i in gate.indices:
            self._occupied[i] = True
        insort(self.gates, gate, key=lambda g: min(g.indices))

    def add_gate(self, gate) -> bool:
        if self.occupied(gate.indices):
            return False
        self._insert_gate(gate)
        return True

    def fill(self) -> None:
        """Identity gadgets (one round of error correction) on every idle qubit."""
        for i in range(self._N):
            if not self.get_gate(i):
                self._insert_gate(dv_gates.I(i))

    def add_pauli(self, index: int, pauli) -> None:
        self.paulis[index][0] = (self.paulis[index][0] + pauli[0]) % 2
        self.paulis[index][1] = (self.paulis[index][1] + pauli[1]) % 2


class MBGKPCircuit:
    def __init__(self, N: int):
        self._N = N
        self._layers: list[Layer] = [Layer(N)]

    @staticmethod
    def transpile(gates: list[DVGate], N: int = None) -> "MBGKPCircuit":
        if N is None:
            N = max(max(gate.indices) for gate in gates) + 1
        circuit = MBGKPCircuit(N)
        for gate in gates:
            circuit.add_gate(gate)
        return circuit

    def depth(self) -> int:
        return len(self._layers)

    def count(self) -> int:
        return sum(len(layer.gates) for layer in self._layers)

    def fill(self) -> None:
        for layer in self._layers:
            layer.fill()

    def to_string(self) -> str:
        rows = []
        for qubit in range(self._N):
            cells = []
            for layer in self._layers:
                gate = layer.get_gate(qubit)
                label = f"'{gate.gate}'" if isinstance(gate, ClassicalControl) else str(gate)
                cells.append(f"{label.ljust(8)} {layer.paulis[qubit]}")
            rows.append(" | ".join(cells))
        return "\n".join(rows)

    def add_gate(self, gate: DVGate) -> None:
        if any(i < 0 or i >= self._N for i in gate.indices):
            raise ValueError(f"Cannot add {gate} to MBGKPCircuit with {self._N} qubits.")
        if len(gate.indices) > 2:
            raise ValueError(f"Only single- and two-mode gates available, but gate {gate} was given.")
        if len(gate.indices) == 2 and abs(gate.indices[0] - gate.indices[1]) != 1:
            raise ValueError(f"Only nearest neighbour interactions available, but gate {gate} was given.")
        kind = type(gate)
        if kind in IMPLEMENTABLES:
            self._add_gate(gate)
            correction = {dv_gates.T: dv_gates.P, dv_gates.Tdg: dv_gates.Pdg}.get(kind)
            if correction is not None:
                self._add_gate(ClassicalControl(correction(gate.indices[0]), [-self._N]))
        elif kind in PAULIS:
            self._add_pauli(gate)
        else:
            raise ValueError(f"Gate {gate} not implementable in MB GKP circuits.")

    def _first_occupied(self, indices):
        """Negative index of the last layer that touches any of ``indices`` (``None`` if none does)."""
        for back in range(1, len(self._layers) + 1):
            if self._layers[-back].occupied(indices):
                return -back
        return None

    def _add_gate(self, gate) -> None:
        last = self._first_occupied(gate.indices)
        if last is None:
            last = -1                              # nothing in the way: ``last + 1`` is the very first layer
        elif last == -1:
            self._layers.append(Layer(self._N))
            last = -2
        self._layers[last + 1].add_gate(gate)

    def _add_pauli(self, gate: DVGate) -> None:
        last = self._first_occupied(gate.indices)
        self._layers[0 if last is None else last].add_pauli(gate.indices[0], _PAULI_BITS[type(gate)])
