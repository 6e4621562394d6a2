"""Base classes of the CV gate API (the surface of ``simulators/cv_simulator/gate_abc.py:15-100``).

What a caller of the reference relies on, and what is kept here:

* ``Gate(arg=None, dagger=False, **svd_options)`` -- truncation keywords (``max_bond_dim``, ``abs_err``, ``rel_err``,
  ``rng_seed``) are collected in ``gate.svd_options`` (the dense register ignores them), anything else is *logged*, never
  raised; ``repr`` is ``Name(arg)^†``.
* ``SingleModeGate(index, ...)`` -- ``index`` must be an ``int``; ``repr`` appends ``_index``.
* ``Measurement(index, result=None, ...)`` -- ``dagger`` is meaningless and dropped with an INFO message.
* ``TwoModeGate(index1, index2, ...)`` -- integer nearest neighbours only; exposes ``left_index < right_index``.
* ``MeasurementResult(result, probability)``.
* ``apply(register, **kwargs)`` mutates the register in place; unknown keyword arguments (``rng=...``) are ignored.
"""
from __future__ import annotations

import logging

from .mps import MPS, SVD_OPTIONS

logger = logging.getLogger("simulators." + __name__.split(".", 1)[1])

#: digits shown for floating-point gate arguments in ``repr``
REPR_DIGITS = 5

__all__ = ["REPR_DIGITS", "MeasurementResult", "Gate", "SingleModeGate", "Measurement", "TwoModeGate"]


def _shown(value) -> str:
    return str(round(value, REPR_DIGITS) if isinstance(value, float) else value)


def _require_int(owner, *values, message: str) -> None:
    if not all(isinstance(v, int) for v in values):
        raise ValueError(f"{type(owner).__name__} {message}")


class MeasurementResult:
    """Outcome of a homodyne measurement: the grid value read off and its probability density."""

    __slots__ = ("result", "probability")

    def __init__(self, result: float, probability: float):
        self.result, self.probability = result, probability

    def __repr__(self):
        return f"{self.result}"


class Gate:
    """Anything with an in-place ``apply(register, **kwargs)``; subclasses implement it."""

    def __init__(self, arg=None, dagger: bool = False, **keywords):
        self.arg, self.dagger = arg, dagger
        self.svd_options = {name: keywords.pop(name) for name in tuple(keywords) if name in SVD_OPTIONS}
        if keywords:
            logger.warning("%s recieved unexpected keyword arguments: %s", type(self).__name__, keywords.keys())

    def _label(self) -> str:
        text = type(self).__name__
        if self.arg is not None:
            text += f"({_shown(self.arg)})"
        return text + "^†" * bool(self.dagger)

    def __repr__(self):
        return self._label()

    def apply(self, mps: MPS, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} does not define apply()")


class SingleModeGate(Gate):
    def __init__(self, index: int, **keywords):
        Gate.__init__(self, **keywords)
        _require_int(self, index, message="requires a single integer index.")
        self.index = index

    def __repr__(self):
        return f"{self._label()}_{self.index}"


class Measurement(SingleModeGate):
    def __init__(self, index, result: float = None, **keywords):
        if keywords.pop("dagger", False):
            logger.info("%s gates ignore adjoint/dagger.", type(self).__name__)
        SingleModeGate.__init__(self, index, **keywords)
        self.result = result

    def __repr__(self):
        tail = f" = {_shown(self.result)}" if self.result else ""
        return SingleModeGate.__repr__(self) + tail


class TwoModeGate(Gate):
    def __init__(self, index1: int, index2: int, **keywords):
        Gate.__init__(self, **keywords)
        _require_int(self, index1, index2, message="requires exactly two indices.")
        if abs(index1 - index2) != 1:
            raise ValueError(f"{type(self).__name__} can only be applied to neighbours, but indices: "
                             f"{(index1, index2)} were given.")
        self.index1, self.index2 = index1, index2
        self.left_index, self.right_index = min(index1, index2), max(index1, index2)

    def __repr__(self):
        return f"{self._label()}_{self.index1},{self.index2}"
