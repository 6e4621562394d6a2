"""Circuit runner of the CV simulator -- mirror of ``simulators/cv_simulator/simulator.py:11-103``.

``Simulator(gates, rng_seed, debug_info=, measurement_formatter=, svd_options=).run(initial_state)`` applies the gates
in order to the register (in place, on the GPU), times each one, logs name / result / shape / time at INFO and
collects ``MeasurementResult``s in ``.results``.  ``svd_options`` are pushed into the gates as in the reference and
have no effect on the dense register.
"""
from __future__ import annotations

import logging
from collections.abc import Callable
from timeit import default_timer as timer

import numpy as np

from .gate_abc import Gate, MeasurementResult
from .mps import MPS, SVD_OPTIONS

logger = logging.getLogger(__name__)


def format_time(time_in_seconds: float) -> str:
    """``mm:ss:ms``."""
    minutes, rest = divmod(time_in_seconds, 60)
    seconds = int(np.floor(rest))
    millis = round((rest - seconds) * 1000)
    return ":".join([str(int(minutes)).rjust(2, "0"), str(seconds).rjust(2, "0"), str(millis).rjust(3, "0")])


class Simulator:
    def __init__(self, gates: list[Gate], rng_seed: int = None, *,
                 debug_info: Callable[["Simulator"], None] = None,
                 measurement_formatter: Callable[[MeasurementResult], str] = None,
                 svd_options: dict = {}):
        self._gates: list[Gate] = gates
        self._state: MPS = None
        self._rng = np.random.default_rng(rng_seed)
        self.results: list[MeasurementResult] = None
        self.debug_info = debug_info or (lambda _: None)
        self.meas_format = measurement_formatter
        options = dict(svd_options)
        self._svd_options = {key: options.pop(key) for key in SVD_OPTIONS if key in options}
        if options:
            logging.warning(f"{type(self).__name__} recieved unexpected keys in svd_options: {options.keys()}")

    def update_gate(self, gate: Gate):
        """Simulation-wide truncation options fill in whatever the gate does not set itself."""
        for key, value in self._svd_options.items():
            gate.svd_options.setdefault(key, value)

    def apply_gate(self, gate: Gate):
        start = timer()
        output = gate.apply(self._state, rng=self._rng)
        self._state.reg.sync()                      # the launch is asynchronous; time the gate, not the enqueue
        elapsed = timer() - start
        if isinstance(output, MeasurementResult):
            self.results.append(output)
            logger.info("   measurement result : " + (self.meas_format(output) if self.meas_format else str(output)))
        logger.info(f"   mps shape: {self._state.shape()}")
        logger.info("   evaluation time : " + format_time(elapsed))
        if logger.isEnabledFor(logging.DEBUG):
            self.debug_info(self)

    def run(self, initial_state: MPS) -> MPS:
        initial_state.validate()
        self._state = initial_state
        self.results = []
        begin = timer()
        logger.info(f"Total number of gates: {len(self._gates)}")
        for i, gate in enumerate(self._gates):
            logger.info(f"Gate {i}: {gate}")
            self.update_gate(gate)
            self.apply_gate(gate)
        logger.info("Finished!")
        logger.info("Total time: " + format_time(timer() - begin))
        return self._state
