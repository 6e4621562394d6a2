"""Circuit runner of the qubit simulator -- the caller of the hot path.

Mirror of ``simulators/dv_simulator/simulator.py:6-53``: ``Simulator(circuit, rng_seed).run(initial_state)``
applies the gates in order, measurement outputs (tuples) append a bit to ``self.results`` and
``ClassicalControl`` fires its gate iff the recorded bits match.  The register is uploaded to the GPU once,
every gate runs there, and the final ket is downloaded once (pass a ``DeviceState`` to skip both copies).
"""
from __future__ import annotations

import numpy as np

from .gates import Gate, Insert, M, is_device_register
from .states import State
from .numpy_quantum import tensor
from ..device import DeviceState


class ClassicalControl:
    """Apply ``gate`` only if all ``positive_indices`` results are 1 and all ``negative_indices`` are 0."""

    def __init__(self, gate: Gate, positive_indices: list[int] = [], negative_indices: list[int] = []):
        self.gate = gate
        self.indices = gate.indices
        self._pos = positive_indices
        self._neg = negative_indices

    def __repr__(self):
        return f"Classical control: {self.gate}"

    def eval(self, observables: list[bool]) -> bool:
        return all(observables[i] for i in self._pos) and not any(observables[i] for i in self._neg)


def parse_state(state) -> np.ndarray | DeviceState:
    """``None`` -> the empty register ``[1.]``; arrays and device registers pass through; a list of
    :class:`State` becomes their tensor product (``simulator.py:19-28``)."""
    if state is None:
        return np.ones((1,))
    if isinstance(state, np.ndarray) or is_device_register(state):
        return state
    if isinstance(state, list) and all(isinstance(item, State) for item in state):
        return tensor(*(s.get() for s in state))
    raise TypeError("Unsupported input type")


class Simulator:
    def __init__(self, circuit: list[Gate], rng_seed: int = None, *, device: int = 0, fuse: int = 0):
        """``fuse``: 0 applies the circuit gate by gate (the reference's loop); k >= 2 first merges neighbouring
        gates into dense blocks of at most k qubits (``quantum_computations_amd.fusion``) -- same final state to
        rounding, fewer passes over HBM."""
        self.circuit: list[Gate] = circuit
        self.results: list[int] = None
        # kept for signature compatibility; like the reference (simulator.py:34, gates.py:183) measurements
        # draw from the global np.random state, not from this generator
        self._rng = np.random.default_rng(rng_seed)
        self._device = device
        self._fuse = fuse
        self.launch_list: list = None      # what run() actually applied (equals circuit when fuse == 0)

    def run(self, initial_state=None):
        self.results = []
        state = parse_state(initial_state)
        on_host = isinstance(state, np.ndarray)
        if on_host and state.ndim != 1:
            return self._run_on_host(state)   # density matrices: gate by gate through Gate.apply

        operands = [state] if on_host else []
        dev = DeviceState.from_numpy(state, self._device) if on_host else state
        if self._fuse >= 2:
            from ..fusion import fuse_circuit
            remote = dev.remote_qubits() if hasattr(dev, "remote_qubits") else ()
            self.launch_list = fuse_circuit(self.circuit, self._fuse, n_qubits=dev.num_qubits, remote=remote)
        else:
            self.launch_list = self.circuit
        box = [dev]

        def apply(gate) -> None:
            if isinstance(gate, ClassicalControl):
                if not gate.eval(self.results):
                    return
                gate = gate.gate
            if on_host:
                for source in getattr(gate, "sources", [gate]):
                    operands.append(_dtype_witness(source, box[0].num_qubits))
            output = gate.apply(box[0])
            if isinstance(output, tuple):
                box[0] = output[0]
                self.results.append(output[1])
            else:
                box[0] = output

        if hasattr(dev, "run_circuit"):
            # sharded registers plan their qubit exchanges over the whole circuit and take commuting gates local-first
            self.launch_list = dev.run_circuit(self.launch_list, apply)
        else:
            for gate in self.launch_list:
                apply(gate)
        dev = box[0]
        if not on_host:
            return dev
        final = dev.to_numpy()
        dev.close()
        dtype = np.result_type(*operands)
        if np.issubdtype(dtype, np.complexfloating):
            return final.astype(dtype, copy=False)
        return np.ascontiguousarray(final.real).astype(dtype, copy=False)

    def _run_on_host(self, state: np.ndarray) -> np.ndarray:
        for gate in self.circuit:
            if isinstance(gate, ClassicalControl):
                if not gate.eval(self.results):
                    continue
                gate = gate.gate
            output = gate.apply(state)
            if isinstance(output, tuple):
                state = output[0]
                self.results.append(output[1])
            else:
                state = output
        return state


def _dtype_witness(gate, n_qubits: int) -> np.ndarray:
    """A zero-size array carrying the dtype this gate contributes to NumPy's promotion in the reference."""
    if isinstance(gate, M):
        return np.empty(0, dtype=np.complex128)          # RZ(phi) RY(theta) is complex (gates.py:169)
    if gate.matrix is None:
        return np.empty(0, dtype=np.float64)
    if isinstance(gate, Insert) or len(gate.indices) == n_qubits:
        return np.empty(0, dtype=gate.matrix.dtype)
    # expand_gate pads with the float64 identity (numpy_quantum.py:245)
    return np.empty(0, dtype=np.result_type(gate.matrix.dtype, np.float64))
