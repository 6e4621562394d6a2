"""Drive ``oracle.mps_oracle.Chain`` with the gate objects of ``quantum_computations_amd.cv_simulator.gates``.

The gate classes only build small host operators (``d x d`` matrices, phase vectors, source coordinates); this module
hands those to the CPU oracle instead of the GPU register, so the same circuit description runs on both.
"""
from __future__ import annotations

import numpy as np

from quantum_computations_amd.cv_simulator import gates as CV
from quantum_computations_amd.cv_simulator.utils import fourier_matrix, rotation_matrix


def apply_to_chain(chain, gate, rng=None):
    """Apply one gate to the oracle chain; returns ``(value, probability)`` for measurements, else ``None``."""
    qs = chain.domain
    options = dict(gate.svd_options, rng_seed=rng)
    if isinstance(gate, CV.D):
        sign = -1 if gate.dagger else 1
        apply_to_chain(chain, CV.X(gate.index, sign * gate.arg[0]), rng)
        apply_to_chain(chain, CV.Z(gate.index, sign * gate.arg[1]), rng)
    elif isinstance(gate, CV.F):
        chain.apply_axis(gate.index, fourier_matrix(qs, inv=gate.dagger))
    elif isinstance(gate, CV._AxisGate):
        op = gate.operator(qs)
        (chain.scale_axis if op.ndim == 1 else chain.apply_axis)(gate.index, op)
    elif isinstance(gate, CV._PlaneResampling):
        x, y = np.meshgrid(qs, qs, indexing="ij")
        chain.plane_resample(gate.left_index, *gate.source_points(x, y), **options)
    elif isinstance(gate, CV.SWAP):
        chain.swap(gate.left_index, **options)
    elif isinstance(gate, CV.CZ):
        strength = -gate.arg if gate.dagger else gate.arg
        chain.plane_phase(gate.left_index, np.exp(1j * strength * np.outer(qs, qs)), **options)
    elif isinstance(gate, CV.Insert):
        chain.insert(gate.index, gate.arg.eval(qs, gate.gkp_epsilon), **options)
    elif isinstance(gate, CV.Mp):
        chain.apply_axis(gate.index, fourier_matrix(qs, inv=True))
        return chain.measure_q(gate.index, gate.result)
    elif isinstance(gate, CV.Homodyne):
        if np.isclose(np.sin(gate.arg), 0):
            value, density = chain.measure_q(gate.index, gate.result)
            return value * np.round(np.cos(gate.arg)), density
        chain.apply_axis(gate.index, rotation_matrix(qs, -gate.arg))
        return chain.measure_q(gate.index, gate.result)
    elif isinstance(gate, CV.Mq):
        return chain.measure_q(gate.index, gate.result)
    else:
        raise TypeError(f"no oracle translation for {gate!r}")
    return None
