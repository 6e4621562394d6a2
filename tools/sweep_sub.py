#!/usr/bin/env python3
"""Experiment: reduced-traffic kernels (CZ = k_diag on a quarter, CX = k_dense_ctrl on a half): unroll x tile regions."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib
from quantum_computations_amd.device import DeviceState

def timed(dev, fn, reps=10):
    fn(); dev.sync(); dev.timer_start()
    for _ in range(reps): fn()
    return dev.timer_stop() / reps

n = 28
dev = DeviceState.random(n, 1)
combos = [(u, r) for u in (1, 2, 4) for r in (0, 8, 32)]
print("gate (bits)      " + "  ".join(f"U{u}/R{r:<2d}" for u, r in combos) + "   [ms]")
cases = [("CZ", 3, 12), ("CZ", 12, 20), ("CZ", 27, 26), ("CZ", 7, 15), ("CZ", 22, 9),
         ("CX", 12, 20), ("CX", 20, 12), ("CX", 27, 26), ("CX", 8, 17), ("CX", 23, 6),
         ("SWAP", 12, 20), ("SWAP", 27, 26), ("SWAP", 9, 16)]
for name, b0, b1 in cases:
    q0, q1 = n - 1 - b0, n - 1 - b1
    fn = {"CZ": lambda: dev.apply_diagonal([1, 1, 1, -1], [q0, q1]), "CX": lambda: dev.apply_cx(q0, q1),
          "SWAP": lambda: dev.apply_swap(q0, q1)}[name]
    row = []
    for u, r in combos:
        dev.set_option(_lib.OPT_UNROLL, u)
        dev.set_option(_lib.OPT_TILE_REGIONS, r)
        row.append(timed(dev, fn))
    print(f"{name:4s} ({b0:2d},{b1:2d})     " + "  ".join(f"{v:6.3f}" for v in row))
