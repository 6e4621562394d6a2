#!/usr/bin/env python3
"""Dense 1q with the shipped defaults: GB/s on every target bit (one column)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import workloads as W
from quantum_computations_amd.device import DeviceState

def timed(dev, fn, reps=8):
    fn(); dev.sync(); dev.timer_start()
    for _ in range(reps): fn()
    return dev.timer_stop() / reps

n = 28
dev = DeviceState.random(n, 1)
gb = 2 * 16 * (1 << n) / 1e9
u2 = W.haar_unitary(2, np.random.default_rng(0))
vals = [gb / (timed(dev, lambda: dev.apply_matrix(u2, [n - 1 - bit])) * 1e-3) for bit in range(n)]
print(" ".join(f"{v:.0f}" for v in vals))
print("mean", sum(vals) / n)
