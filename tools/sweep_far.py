#!/usr/bin/env python3
"""Experiment: dense 1q on the 'far' target bits (pair stride 16..512 MiB): unroll x tile regions."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib, workloads as W
from quantum_computations_amd.device import DeviceState

def timed(dev, fn, reps=8):
    fn(); dev.sync(); dev.timer_start()
    for _ in range(reps): fn()
    return dev.timer_stop() / reps

n = 28
dev = DeviceState.random(n, 1)
gb = 2 * 16 * (1 << n) / 1e9
u2 = W.haar_unitary(2, np.random.default_rng(0))
combos = [(u, r) for u in (1, 2, 4, 8) for r in (0, 2, 4, 8)]
print("bit  " + "  ".join(f"U{u}/R{r}" for u, r in combos))
for bit in (19, 20, 21, 22, 23, 24, 25, 26):
    row = []
    for u, r in combos:
        dev.set_option(_lib.OPT_UNROLL, u)
        dev.set_option(_lib.OPT_TILE_REGIONS, r)
        row.append(gb / (timed(dev, lambda: dev.apply_matrix(u2, [n - 1 - bit])) * 1e-3))
    print(f"{bit:3d}  " + "  ".join(f"{v:5.0f}" for v in row))
