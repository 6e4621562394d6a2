"""Abstract gate API of the CV simulator -- mirror of ``simulators/cv_simulator/gate_abc.py:15-100``.

Same classes, constructor signatures and error behaviour: parametrised gates take ``arg`` / ``dagger`` plus truncation
keywords (accepted for compatibility, unused by the dense register); unknown keywords are logged, not raised;
two-mode gates act on nearest neighbours only.
"""
from __future__ import annotations

import logging
from abc import ABC, abstractmethod
from typing import Any

from .mps import MPS, SVD_OPTIONS

logger = logging.getLogger(__name__)

REPR_DIGITS = 5


class MeasurementResult:
    def __init__(self, result: float, probability: float):
        self.result: float = result
        self.probability: float = probability

    def __repr__(self):
        return str(self.result)


class Gate(ABC):
    def __init__(self, arg: Any = None, dagger: bool = False, **kwargs):
        self.arg = arg
        self.dagger = dagger
        self.svd_options = {key: kwargs.pop(key) for key in SVD_OPTIONS if key in kwargs}
        if kwargs:
            logger.warning(f"{type(self).__name__} recieved unexpected keyword arguments: {kwargs.keys()}")

    def __repr__(self):
        arg = round(self.arg, REPR_DIGITS) if isinstance(self.arg, float) else self.arg
        return type(self).__name__ + (f"({arg})" if arg is not None else "") + ("^†" if self.dagger else "")

    @abstractmethod
    def apply(self, mps: MPS, **kwargs) -> None | MeasurementResult:
        """Apply the gate to ``mps`` in place; measurements return their result.  Keyword arguments a gate does
        not use (e.g. ``rng``) are ignored silently."""


class SingleModeGate(Gate):
    def __init__(self, index: int, **kwargs):
        super().__init__(**kwargs)
        if not isinstance(index, int):
            raise ValueError(f"{type(self).__name__} requires a single integer index.")
        self.index = index

    def __repr__(self):
        return super().__repr__() + f"_{self.index}"


class Measurement(SingleModeGate):
    def __init__(self, index, result: float = None, **kwargs):
        if kwargs.pop("dagger", None):
            logger.info(type(self).__name__ + "gates ignores adjoint/dagger.")
        super().__init__(index, **kwargs)
        self.result: float = result

    def __repr__(self):
        return super().__repr__() + (f" = {round(self.result, REPR_DIGITS)}" if self.result else "")

    @abstractmethod
    def apply(self, mps: MPS, **kwargs) -> MeasurementResult:
        pass


class TwoModeGate(Gate):
    def __init__(self, index1: int, index2: int, **kwargs):
        super().__init__(**kwargs)
        if not isinstance(index1, int) or not isinstance(index2, int):
            raise ValueError(f"{type(self).__name__} requires exactly two indices.")
        if abs(index1 - index2) != 1:
            raise ValueError(f"{type(self).__name__} can only be applied to neighbours, but indices: "
                             f"{(index1, index2)} were given.")
        self.index1, self.index2 = index1, index2
        self.left_index, self.right_index = sorted([index1, index2])

    def __repr__(self):
        return super().__repr__() + f"_{self.index1},{self.index2}"
