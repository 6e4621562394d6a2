#!/usr/bin/env python3
"""Experiment: XCD-contiguous tile order (option 6) vs default, dense 1q, every target bit."""
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
regions = (-1, 0, 8, 16, 32, 64)
print("bit  " + "  ".join(f"R{r:<5d}" for r in regions) + "   (U default)")
for bit in range(n):
    row = []
    for r in regions:
        dev.set_option(_lib.OPT_TILE_REGIONS, r)
        row.append(gb / (timed(dev, lambda: dev.apply_matrix(u2, [n - 1 - bit])) * 1e-3))
    print(f"{bit:3d}  " + "  ".join(f"{v:6.0f}" for v in row))
