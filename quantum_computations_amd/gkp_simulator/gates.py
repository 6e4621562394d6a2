"""Measurement-based GKP gates (``simulators/gkp_simulator/gates.py:14-258``).

Every logical Clifford (and T, through a magic Bell pair) is a teleportation gadget: entangle the data mode(s) with fresh
GKP Bell pairs on balanced beam splitters, measure rotated quadratures, and read the Pauli by-product off the outcomes.
A gadget is described by its homodyne angles alone:

* one mode (Walshe et al., PRA 102, 062411): ``InsertBell(i+1) · BS(i, i+1) · Homodyne(i, θa) · Homodyne(i, θb)``;
* two modes (Walshe et al., arXiv:2109.04668): two Bell pairs around the data modes, four beam splitters, four homodyne
  measurements, angles ordered ``[a, c, b, d]``.

``compile()`` returns the CV gate list (run by ``cv_simulator.Simulator`` on the GPU register); ``compute_syndrome`` maps
the measured values to the by-product ``(x, z)`` bits per output mode.
"""
from __future__ import annotations

import logging
from enum import Enum, auto

import numpy as np

from ..cv_simulator.gates import *  # noqa: F401,F403  (the CV gate set is part of this module's namespace upstream)
from ..cv_simulator.gates import BS, Gate, Homodyne
from ..cv_simulator.mps import SVD_OPTIONS
from .insert_bell import GKPBellState, InsertBell
from .utils import PI, SQPI

logger = logging.getLogger("simulators." + __name__.split(".", 1)[1])

Syndrome = tuple[int, int]

_ATAN2 = np.arctan(2)


class MBType(Enum):
    I = auto()
    F = auto()
    P = auto()

    def angles(self) -> list[float]:
        return {"I": [0.0, PI / 2], "F": [PI / 4, -PI / 4], "P": [0.0, _ATAN2]}[self.name]


class MB2Type(Enum):
    II = auto()
    FF = auto()
    PP = auto()
    PPdg = auto()
    CZ = auto()
    SWAP = auto()

    def angles(self) -> list[float]:
        return {"II": [0.0, 0.0, PI / 2, PI / 2], "FF": [PI / 4, PI / 4, -PI / 4, -PI / 4],
                "PP": [0.0, 0.0, _ATAN2, _ATAN2], "PPdg": [0.0, 0.0, _ATAN2, -_ATAN2],
                "CZ": [0.0, 0.0, _ATAN2, -_ATAN2], "SWAP": [-PI / 2, 0.0, 0.0, -PI / 2]}[self.name]


def _byproduct(m_first: float, m_second: float, t_first: float, t_second: float) -> complex:
    """Complex displacement left on the output by a pair of homodyne outcomes at angles (t_first, t_second), measured
    from the q axis (hence the factor i relative to the papers' convention)."""
    return 1j * (m_first * np.exp(1j * t_second) + m_second * np.exp(1j * t_first)) / np.sin(t_first - t_second)


def _parity_bits(mu: complex, scale: float = 1.0) -> Syndrome:
    """``(x, z)``: parities of the nearest multiples of sqrt(pi) in the q and p components of the displacement."""
    quadratures = np.array([mu.real, mu.imag]) * scale
    return tuple(int(v) for v in np.round(quadratures / SQPI) % 2)


class MeasurementBased:
    """A logical gate given by a gadget type: ``indices`` are the data modes it acts on."""

    def __init__(self, indices: list[int], type, epsilon: float = None, *, dagger: bool = False, **kwargs):
        self.indices, self.type, self.epsilon, self.dagger = indices, type, epsilon, dagger
        self.svd_options = {key: kwargs.pop(key) for key in tuple(kwargs) if key in SVD_OPTIONS}
        if kwargs:
            logger.warning("%s recieved unexpected keyword arguments: %s", self.__class__.__name__, kwargs.keys())

    def angles(self) -> np.ndarray:
        return np.array(self.type.angles()) * (-1 if self.dagger else 1)

    def compile(self) -> list[Gate]:
        raise NotImplementedError

    def compute_syndrome(self, results: list[float]) -> tuple[list[Syndrome], list[int]]:
        """By-product bits per output mode and the mode index each belongs to; ``results`` in the order the
        measurements appear in :meth:`compile`."""
        raise NotImplementedError


class MBSingleMode(MeasurementBased):
    def __init__(self, index: int, type: MBType, epsilon: float = None, *, results=None, **kwargs):
        super().__init__([index], type, epsilon, **kwargs)
        self.results = (None, None) if results is None else results
        if len(self.results) != 2:
            raise ValueError("Results list must have exactly 2 elements.")

    def compile(self):
        i = self.indices[0]
        first, second = self.angles()
        return [InsertBell(i + 1, self._bell_state(), gkp_epsilon=self.epsilon, **self.svd_options),
                BS(i, i + 1, **self.svd_options),
                Homodyne(i, first, result=self.results[0]),
                Homodyne(i, second, result=self.results[1])]

    def _bell_state(self) -> GKPBellState:
        return GKPBellState.PLUS

    def compute_syndrome(self, results):
        if len(results) != 2:
            raise ValueError("Exactly two measurement results are needed.")
        ta, tb = self.angles()
        return [_parity_bits(_byproduct(results[0], results[1], ta, tb), 2 ** 0.5)], self.indices


class MBTwoMode(MeasurementBased):
    def __init__(self, index1: int, index2: int, type: MB2Type, epsilon: float = None, *, results=None, **kwargs):
        if abs(index1 - index2) != 1:
            raise ValueError(f"{self.__class__.__name__} can only be applied to neighbours, but indices: "
                             f"{(index1, index2)} were given.")
        results = (None,) * 4 if results is None else results
        if len(results) != 4:
            raise ValueError("Results list must have exactly 4 elements.")
        super().__init__(sorted([index1, index2]), type, epsilon, **kwargs)
        self.results = results

    def compile(self):
        i = min(self.indices)
        ta, tc, tb, td = self.angles()
        ma, mc, mb, md = self.results
        bell = dict(gkp_epsilon=self.epsilon, **self.svd_options)
        # chain after the insertions: [bell, bell, data, data, bell, bell] at i .. i+5
        return [InsertBell(i, **bell), InsertBell(i + 4, **bell),
                BS(i + 2, i + 1, **self.svd_options), BS(i + 3, i + 4, **self.svd_options),
                BS(i + 2, i + 3, **self.svd_options),
                Homodyne(i + 2, ta, result=ma), Homodyne(i + 2, tc, result=mc),
                BS(i + 1, i + 2, **self.svd_options),
                Homodyne(i + 1, tb, result=mb), Homodyne(i + 1, td, result=md)]

    def compute_syndrome(self, results):
        if len(results) != 4:
            raise ValueError("Exactly two measurement results are needed.")
        ta, tc, tb, td = self.angles()
        ma, mc, mb, md = results
        mu_ab, mu_cd = _byproduct(ma, mb, ta, tb), _byproduct(mc, md, tc, td)
        # the 1/sqrt(2) of the last beam splitter and the sqrt(2) of the quadrature convention cancel
        return [_parity_bits(mu_cd + mu_ab), _parity_bits(mu_cd - mu_ab)], self.indices


def _single(name: str, gadget: MBType, doc: str):
    def __init__(self, index, epsilon: float = None, *, results=None, **kwargs):
        MBSingleMode.__init__(self, index, gadget, epsilon=epsilon, results=results, **kwargs)
    return type(name, (MBSingleMode,), {"__init__": __init__, "__doc__": doc, "__module__": __name__})


def _pair(name: str, gadget: MB2Type, doc: str):
    def __init__(self, index1: int, index2: int, epsilon: float = None, *, results=None, **kwargs):
        MBTwoMode.__init__(self, index1, index2, gadget, epsilon=epsilon, results=results, **kwargs)
    return type(name, (MBTwoMode,), {"__init__": __init__, "__doc__": doc, "__module__": __name__})


MBI = _single("MBI", MBType.I, "Error correction by teleportation (Knill).")
GKPEC = MBI
MBF = _single("MBF", MBType.F, "Error-corrected Fourier gate (logical Hadamard).")
MBP = _single("MBP", MBType.P, "Error-corrected quadratic phase gate (logical P).")
MBSWAP = _pair("MBSWAP", MB2Type.SWAP, "Error-corrected SWAP.")
MBCZ = _pair("MBCZ", MB2Type.CZ, "Error-corrected controlled-Z.")


class MBT(MBSingleMode):
    """Logical T (non-Gaussian): the identity gadget fed with a magic Bell pair; ``dagger`` selects T^dagger by the
    pair, not by the measurement angles."""

    def __init__(self, index, epsilon: float = None, *, results=None, **kwargs):
        super().__init__(index, MBType.I, epsilon=epsilon, results=results, **kwargs)

    def _bell_state(self):
        return GKPBellState.Tdg if self.dagger else GKPBellState.T

    def compile(self):
        i = self.indices[0]
        first, second = MBType.I.angles()          # not negated for the adjoint
        return [InsertBell(i + 1, self._bell_state(), gkp_epsilon=self.epsilon, **self.svd_options),
                BS(i, i + 1, **self.svd_options),
                Homodyne(i, first, result=self.results[0]),
                Homodyne(i, second, result=self.results[1])]
