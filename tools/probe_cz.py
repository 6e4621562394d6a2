#!/usr/bin/env python3
"""The CZ gates of the benchmark circuit (k_diag on a quarter of the register): total ms by items per thread and tile order."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib
from quantum_computations_amd import workloads as W
from quantum_computations_amd.device import DeviceState

n = 28
ops = W.random_circuit(n, 100, 100)
gates = [g for g, o in zip(W.to_gates(ops), ops) if o["name"] == "CZ"]
dev = DeviceState.random(n, 1)


def timed(fn, reps=6):
    fn(); dev.sync(); dev.timer_start()
    for _ in range(reps): fn()
    return dev.timer_stop() / reps


combos = [(0, -1)] + [(u, r) for u in (1, 2, 4) for r in (0, 8, 32)]
tot = {c: 0.0 for c in combos}
for g in gates:
    bits = sorted(n - 1 - q for q in g.indices)
    cells = []
    for c in combos:
        dev.set_option(_lib.OPT_UNROLL, c[0])
        dev.set_option(_lib.OPT_TILE_REGIONS, c[1])
        ms = timed(lambda: g.apply(dev))
        tot[c] += ms
        cells.append(f"{ms:.3f}")
    print(bits, " ".join(cells), flush=True)
print("total", {c: round(t, 3) for c, t in tot.items()})
