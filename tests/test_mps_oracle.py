"""Pin ``oracle/mps_oracle.py`` against what the reference's MPS produced with truncation on (CPU only)."""
from __future__ import annotations

import json

import numpy as np
import pytest

from fixture_io import cv_mps_program
from mps_driver import apply_to_chain
from oracle import mps_oracle as MO
from quantum_computations_amd.cv_simulator import gates as CV
from quantum_computations_amd.cv_simulator.states import State


def maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))


def test_split_matches_reference_tensor_svd(golden):
    g = golden["cv_mps"]
    cases = [c for c in json.loads(str(g["cases"])) if c.get("kind") == "tensor_svd"]
    for c in cases:
        t = g[f"svd_in_{c['index']}"]
        order = c["left"] + c["right"]
        rows = int(np.prod([t.shape[i] for i in c["left"]]))
        m1, m2 = MO.split(np.moveaxis(t, order, range(t.ndim)).reshape(rows, -1), **c["options"])
        assert m1.shape[1] == c["rank"]
        want = g[f"svd_product_{c['index']}"].reshape(rows, -1)
        assert maxdiff(m1 @ m2, want) < 1e-11


@pytest.mark.parametrize("label", ["rel1e-6", "abs1e-3", "cap5"])
def test_chain_matches_reference_mps(golden, label):
    g = golden["cv_mps"]
    case = next(c for c in json.loads(str(g["cases"])) if c.get("label") == label)
    chain, rng, results = MO.Chain(g["qs"]), np.random.default_rng(5), []
    for position, gate in enumerate(cv_mps_program(CV, State, case["options"])):
        out = apply_to_chain(chain, gate, rng)
        if out is not None:
            results.append([position, out[0], out[1]])
        assert chain.shapes() == case["shapes"][position], (position, gate)
        assert abs(chain.norm() - g[f"{label}_norms"][position]) < 1e-10
        key = f"{label}_state_{position}"
        if key in g:
            assert maxdiff(chain.contract(), g[key]) < 1e-10, (position, gate)
    assert np.allclose(np.array(results), g[f"{label}_results"], rtol=0, atol=1e-10)
    assert maxdiff(np.real(np.diag(chain.partial_density(1))), g[f"{label}_marginal"]) < 1e-10
    assert maxdiff(chain.partial_density(0), g[f"{label}_rho0"]) < 1e-10
