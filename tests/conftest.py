"""Shared test plumbing.  ``-m "not gpu"`` runs on CPU only; ``-m gpu`` needs one MI355X."""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
GOLDEN = REPO / "tests" / "golden"
for p in (str(REPO), str(GOLDEN)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


class Golden:
    """Lazy access to the committed fixtures (tests/golden/*.npz)."""

    def __init__(self):
        self._cache = {}

    def __getitem__(self, name: str):
        if name not in self._cache:
            self._cache[name] = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
        return self._cache[name]

    def cases(self, name: str):
        return json.loads(str(self[name]["cases"]))


@pytest.fixture(scope="session")
def golden():
    return Golden()


@pytest.fixture(scope="session")
def state_vectors():
    from quantum_computations_amd.dv_simulator.states import State
    return {s.name: s.get() for s in State}
