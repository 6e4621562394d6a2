"""MI355X-backed drop-in for ``simulators.dv_simulator`` (qubit state-vector simulator)."""
