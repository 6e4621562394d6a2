"""Runner of measurement-based GKP circuits (``simulators/gkp_simulator/simulator.py:20-165``).

Per layer and gate: (1) resolve a classically controlled correction from the previous layer's by-products, (2) commute
the gate through the current Pauli frame (Clifford gates permute the frame bits, a ``T`` behind an X flips to ``Tdg``),
(3) compile the gadget and run its CV gates on the matrix-product register -- this is where the GPU works -- and
(4) fold the measured by-products and the layer's own Paulis into the frame.  Returns the register and the final frame;
the logical state is ``frame · |register>``.
"""
from __future__ import annotations

import logging
from timeit import default_timer as timer
from typing import Callable

import numpy as np

from ..cv_simulator.gate_abc import MeasurementResult
from ..cv_simulator.gates import F as FourierGate
from ..cv_simulator.mps import MPS, SVD_OPTIONS
from ..cv_simulator.simulator import Simulator as CVSimulator, format_time
from ..dv_simulator import gates as dv_gates
from ..dv_simulator.gates import Gate as DVGate
from .gates import SQPI, Syndrome  # noqa: F401
from .transpiler import ClassicalControl, MBGKPCircuit, MeasurementBased, gate_transpile
from .utils import format_result

logger = logging.getLogger("simulators." + __name__.split(".", 1)[1])


def measurement_formatter(result: MeasurementResult) -> str:
    return format_result(result.result)


def _swap_bits(frame, gate):
    i = gate.indices[0]
    frame[i][0], frame[i][1] = frame[i][1], frame[i][0]
    return gate


def _phase_gate(frame, gate):
    i = gate.indices[0]
    frame[i][1] ^= frame[i][0]
    return gate


def _controlled_z(frame, gate):
    i, j = gate.indices
    frame[i][1] ^= frame[j][0]
    frame[j][1] ^= frame[i][0]
    return gate


def _exchange(frame, gate):
    i, j = gate.indices
    frame[i], frame[j] = frame[j], frame[i]
    return gate


def _t_gate(adjoint):
    def rule(frame, gate):
        return adjoint(*gate.indices) if frame[gate.indices[0]][0] == 1 else gate
    return rule


_COMMUTATION = {dv_gates.I: lambda frame, gate: gate, dv_gates.T: _t_gate(dv_gates.Tdg), dv_gates.Tdg: _t_gate(dv_gates.T),
                dv_gates.H: _swap_bits, dv_gates.P: _phase_gate, dv_gates.Pdg: _phase_gate, dv_gates.CZ: _controlled_z,
                dv_gates.SWAP: _exchange}


def commute(gate: DVGate, paulis: list[Syndrome]) -> tuple[list[Syndrome], DVGate]:
    """``gate · paulis = paulis' · gate'``: the frame after the gate and the gate that must actually be applied."""
    rule = _COMMUTATION.get(type(gate))
    if rule is None:
        raise NotImplementedError(f"Commutator logic for gate: {gate} not implemented.")
    frame = [list(p) for p in paulis]
    gate = rule(frame, gate)
    return [tuple(p) for p in frame], gate


class Simulator(CVSimulator):
    def __init__(self, circuit: MBGKPCircuit, ancilla_epsilon: float, *, rng_seed: int = None, svd_options: dict = {},
                 debug_info: Callable[["Simulator"], None] = None):
        """``ancilla_epsilon`` is the GKP damping of every Bell pair the gadgets insert; ``svd_options`` the truncation
        of every split."""
        self._circuit, self._N = circuit, circuit._N
        self._rng = np.random.default_rng(rng_seed)
        self._epsilon = ancilla_epsilon
        self._state: MPS = None
        self.pauli_syndrome: list[Syndrome] = None
        unknown = [key for key in svd_options if key not in SVD_OPTIONS]
        if unknown:
            logging.warning("%s recieved unexpected keys in svd_options: %s", type(self).__name__, unknown)
        self._svd_options = {key: value for key, value in svd_options.items() if key in SVD_OPTIONS}
        self.debug_info = debug_info or (lambda _: None)

    def apply_gate(self, dv_gate: DVGate) -> tuple[list[Syndrome], list[int]]:
        gadget: MeasurementBased = gate_transpile(dv_gate, epsilon=self._epsilon, **self._svd_options)
        runner = CVSimulator(gadget.compile(), rng_seed=self._rng, measurement_formatter=measurement_formatter)
        self._state = runner.run(self._state)
        return gadget.compute_syndrome([r.result for r in runner.results])

    def apply_paulis(self, paulis: list[Syndrome]) -> None:
        self.pauli_syndrome = [(mine[0] ^ theirs[0], mine[1] ^ theirs[1])
                               for mine, theirs in zip(self.pauli_syndrome, paulis)]

    def run(self, initial_state: MPS) -> tuple[MPS, list[Syndrome]]:
        initial_state.validate()
        self._state = initial_state
        self.pauli_syndrome = [(0, 0)] * self._N
        previous, current = [(0, 0)] * self._N, [(0, 0)] * self._N      # by-products of the last two layers
        started = timer()
        layers = self._circuit._layers
        logger.info("Total number of MB gates: %d in a total of %d layers.", self._circuit.count(), len(layers))
        for number, layer in enumerate(layers, start=1):
            logger.info("Layer %d of %d.", number, len(layers))
            previous, current = current, [(0, 0)] * self._N
            for gate in layer.gates:
                if isinstance(gate, ClassicalControl):
                    # the Clifford correction of a teleported T fires on the x bit of that gadget's by-product
                    gate = gate.gate if previous[gate.indices[0]][0] else dv_gates.I(*gate.indices)
                self.pauli_syndrome, gate = commute(gate, self.pauli_syndrome)
                logger.info("MB gate: %s", gate)
                syndromes, indices = self.apply_gate(gate)
                logger.info("Gate syndrome: %s", syndromes)
                if len(indices) != len(syndromes):
                    raise ValueError("one syndrome per output mode expected")
                for index, syndrome in zip(indices, syndromes):
                    current[index] = syndrome
            logger.info("Applying syndrome correction: %s", current)
            self.apply_paulis(current)
            logger.info("Applying Pauli operators: %s", layer.paulis)
            self.apply_paulis(layer.paulis)
            logger.info("Final Pauli syndrome: %s", self.pauli_syndrome)
            if logger.isEnabledFor(logging.DEBUG):
                self.debug_info(self)
        logger.info("Finished MB GKP simulation!")
        logger.info("Total time: %s", format_time(timer() - started))
        return self._state, [tuple(s) for s in self.pauli_syndrome]


class SimulatorAlt(Simulator):
    """Shortcut variant: identities are skipped and a Hadamard is applied as a bare Fourier gate (no gadget)."""

    def apply_gate(self, dv_gate):
        if type(dv_gate) is dv_gates.I:
            return [(0, 0)], dv_gate.indices
        if type(dv_gate) is dv_gates.H:
            FourierGate(dv_gate.indices[0]).apply(self._state)
            return [(0, 0)], dv_gate.indices
        return super().apply_gate(dv_gate)
