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
    def __init__(self, circuit: list[Gate], rng_seed: int = None, *, device: int = 0, fuse: int = 0,
                 single_launch: bool = True):
        """``fuse``: 0 applies the circuit gate by gate (the reference's loop); k >= 2 first merges neighbouring
        gates into dense blocks of at most k qubits (``quantum_computations_amd.fusion``) -- same final state to
        rounding, fewer passes over HBM.

        ``single_launch``: a host ket of at most 13 qubits (the reference's own sizes) runs the whole circuit --
        measurements, insertions and classical control included -- in ONE kernel launch with the register in a
        workgroup's LDS (``dv_simulator.program``, ``qsv_run_programs``) whenever every gate can be expressed there;
        otherwise, and for device registers, gate by gate.  Same results, same use of ``np.random``."""
        self.circuit: list[Gate] = circuit
        self.results: list[int] = None
        # kept for signature compatibility; like the reference (simulator.py:34, gates.py:183) measurements
        # draw from the global np.random state, not from this generator
        self._rng = np.random.default_rng(rng_seed)
        self._device = device
        self._fuse = fuse
        self._single_launch = single_launch
        self.single_launch_used = False    # did the last run() go through the one-launch executor?
        self.launch_list: list = None      # what run() actually applied (equals circuit when fuse == 0)

    def run(self, initial_state=None):
        self.results = []
        self.single_launch_used = False
        state = parse_state(initial_state)
        on_host = isinstance(state, np.ndarray)
        if on_host and state.ndim != 1:
            return self._run_on_host(state)   # density matrices: gate by gate through Gate.apply

        if on_host and self._single_launch and self._fuse < 2:
            done = self._run_in_one_launch([self.circuit], [state], self._device)
            if done is not None:
                (final, self.results), = done
                self.launch_list = self.circuit
                self.single_launch_used = True
                return final

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

    @staticmethod
    def _run_in_one_launch(circuits, states, device: int):
        """``[(final ket, results)]`` for host kets through the single-launch executor, or ``None`` if any of the
        circuits needs the gate-by-gate path (nothing has been run, no random number drawn in that case)."""
        from . import program as P

        programs = []
        for circuit, state in zip(circuits, states):
            size = state.shape[0] if state.ndim == 1 else 0
            if size == 0 or size & (size - 1) or size > (1 << P.MAX_QUBITS):
                return None
            if not (np.issubdtype(state.dtype, np.number) or state.dtype == bool):
                return None
            try:
                programs.append(P.compile_circuit(circuit, size.bit_length() - 1))
            except P.Unsupported:
                return None
        out = []
        for prog, state, (final, results, probs) in zip(programs, states, P.run_programs(programs, states, device)):
            operands = [state]
            measured = 0
            for gate, control, n_now in prog.steps:
                if isinstance(gate, M):
                    if gate.result is None and abs(float(probs[measured].sum()) - 1.0) > np.sqrt(np.finfo(np.float64).eps):
                        raise ValueError("probabilities do not sum to 1")      # what np.random.choice says (gates.py:183)
                    measured += 1
                elif control is not None and not control.eval(results):
                    continue
                operands.append(_dtype_witness(gate, n_now))
            dtype = np.result_type(*operands)
            if np.issubdtype(dtype, np.complexfloating):
                final = final.astype(dtype, copy=False)
            else:
                final = np.ascontiguousarray(final.real).astype(dtype, copy=False)
            out.append((final, results))
        return out

    @classmethod
    def run_batch(cls, circuits, initial_states=None, *, device: int = 0):
        """Run many independent circuits: ``[(final state, results), ...]`` in the order given.

        The MI355X replacement for the reference's ``multiprocessing.Pool`` sweeps over small circuits
        (``impact_.../randomised_benchmarking.py:60-76``, ``average_clifford_fidelity.py:212``): all instances go to the
        GPU in ONE launch, one workgroup per instance with its register in LDS, spread over the 256 CUs.  Instances
        the executor cannot express (or registers beyond 13 qubits) run one after the other through ``run``."""
        circuits = list(circuits)
        states = [parse_state(s) for s in (initial_states if initial_states is not None else [None] * len(circuits))]
        if len(states) != len(circuits):
            raise ValueError("one initial state per circuit")
        if all(isinstance(s, np.ndarray) and s.ndim == 1 for s in states):
            done = cls._run_in_one_launch(circuits, states, device)
            if done is not None:
                return done
        out = []
        for circuit, state in zip(circuits, states):
            sim = cls(circuit, device=device)
            out.append((sim.run(state), sim.results))
        return out

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
