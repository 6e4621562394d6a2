"""Circuit runner of the CV simulator (the surface of ``simulators/cv_simulator/simulator.py:11-103``).

``Simulator(gates, rng_seed, *, debug_info, measurement_formatter, svd_options).run(register)`` applies the gates in
order -- in place, on the GPU -- and collects every ``MeasurementResult`` in ``.results``.  As upstream: the
simulation-wide ``svd_options`` fill in whatever a gate does not set itself (and are inert on the dense register),
unknown option keys are logged and dropped, each gate is timed and reported at INFO (name, outcome, register shape,
``mm:ss:ms``), and ``debug_info(simulator)`` is called per gate when the logger is at DEBUG.
"""
from __future__ import annotations

import logging
import time

import numpy as np

from .gate_abc import Gate, MeasurementResult
from .mps import MPS, SVD_OPTIONS

logger = logging.getLogger("simulators." + __name__.split(".", 1)[1])


def format_time(time_in_seconds: float) -> str:
    """Seconds -> ``mm:ss:ms`` with zero padding."""
    whole = int(time_in_seconds)
    millis = round((time_in_seconds - whole) * 1000)
    return f"{whole // 60:02d}:{whole % 60:02d}:{millis:03d}"


class Simulator:
    def __init__(self, gates, rng_seed=None, *, debug_info=None, measurement_formatter=None, svd_options={}):
        self._gates = gates
        self._state = None
        self._rng = np.random.default_rng(rng_seed)
        self.results = None
        self.debug_info = debug_info if debug_info is not None else (lambda simulator: None)
        self.meas_format = measurement_formatter
        known = {key: value for key, value in svd_options.items() if key in SVD_OPTIONS}
        unknown = [key for key in svd_options if key not in SVD_OPTIONS]
        if unknown:
            logging.warning("%s recieved unexpected keys in svd_options: %s", type(self).__name__, unknown)
        self._svd_options = known

    # -- pieces of run(), kept as methods because callers of the reference override / call them ----------------
    def update_gate(self, gate: Gate) -> None:
        for key, value in self._svd_options.items():
            gate.svd_options.setdefault(key, value)

    def apply_gate(self, gate: Gate) -> None:
        started = time.perf_counter()
        outcome = gate.apply(self._state, rng=self._rng)
        self._state.reg.sync()          # launches are asynchronous: wait, so that the time below is the gate's
        elapsed = time.perf_counter() - started
        if isinstance(outcome, MeasurementResult):
            self.results.append(outcome)
            shown = self.meas_format(outcome) if self.meas_format else str(outcome)
            logger.info("   measurement result : %s", shown)
        logger.info("   mps shape: %s", self._state.shape())
        logger.info("   evaluation time : %s", format_time(elapsed))
        if logger.isEnabledFor(logging.DEBUG):
            self.debug_info(self)

    def run(self, initial_state: MPS) -> MPS:
        initial_state.validate()
        self._state, self.results = initial_state, []
        started = time.perf_counter()
        logger.info("Total number of gates: %d", len(self._gates))
        for position, gate in enumerate(self._gates):
            logger.info("Gate %d: %s", position, gate)
            self.update_gate(gate)
            self.apply_gate(gate)
        logger.info("Finished!")
        logger.info("Total time: %s", format_time(time.perf_counter() - started))
        return self._state
