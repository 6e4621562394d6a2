"""Named single-qubit states for ``Insert`` and for initial registers given as ``[State.ZERO, State.PLUS, ...]``.

Same members and amplitudes as ``simulators/dv_simulator/states.py:5-31``: the four stabiliser states plus the magic
states ``T`` / ``TDG`` = (|0> + e^{+-i pi/4}|1>)/sqrt(2) and ``H`` = cos(pi/8)|0> + sin(pi/8)|1>.
"""
from __future__ import annotations

import enum

import numpy as np

from . import numpy_quantum as npq


def _phase_state(sign: int) -> np.ndarray:
    return np.array([1.0, np.exp(sign * 1.0j * np.pi / 4.0)]) * 2 ** -0.5


_BUILDERS = {
    "ZERO": lambda: npq.ZERO,
    "ONE": lambda: npq.ONE,
    "PLUS": lambda: npq.PLUS,
    "MINUS": lambda: npq.MINUS,
    "T": lambda: _phase_state(+1),
    "TDG": lambda: _phase_state(-1),
    "H": lambda: np.array([np.cos(np.pi / 8.0), np.sin(np.pi / 8.0)]),
}


class State(enum.Enum):
    """``State.X.get()`` returns the 2-vector of amplitudes; ``repr`` is the bare member name."""

    ZERO, ONE, PLUS, MINUS, T, TDG, H = range(1, 8)

    def __repr__(self):
        return self.name

    def get(self) -> np.ndarray:
        return _BUILDERS[self.name]()
