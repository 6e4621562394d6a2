"""Named single-qubit states for ``Insert`` and initial registers.

Mirror of ``simulators/dv_simulator/states.py:5-31`` (same member names and amplitudes).
"""
from __future__ import annotations

from enum import Enum, auto

import numpy as np

from . import numpy_quantum as npq


class State(Enum):
    ZERO = auto()
    ONE = auto()
    PLUS = auto()
    MINUS = auto()
    T = auto()
    TDG = auto()
    H = auto()

    def __repr__(self):
        return self.name

    def get(self) -> np.ndarray:
        """The 2-vector of amplitudes (magic states: T = (|0> + e^{i pi/4}|1>)/sqrt2, H = cos(pi/8)|0> + sin(pi/8)|1>)."""
        return _AMPLITUDES[self.name]()


_AMPLITUDES = {
    "ZERO": lambda: npq.ZERO,
    "ONE": lambda: npq.ONE,
    "PLUS": lambda: npq.PLUS,
    "MINUS": lambda: npq.MINUS,
    "T": lambda: np.array([1.0, np.exp(0.25j * np.pi)]) * 2 ** -0.5,
    "TDG": lambda: np.array([1.0, np.exp(-0.25j * np.pi)]) * 2 ** -0.5,
    "H": lambda: np.array([np.cos(np.pi / 8.0), np.sin(np.pi / 8.0)]),
}
