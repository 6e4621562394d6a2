#!/usr/bin/env python3
"""Experiment: k_dense_big (k = 3..5) by tile regions."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib, workloads as W
from quantum_computations_amd.device import DeviceState

def timed(dev, fn, reps=6):
    fn(); dev.sync(); dev.timer_start()
    for _ in range(reps): fn()
    return dev.timer_stop() / reps

n = 28
dev = DeviceState.random(n, 1)
gb = 2 * 16 * (1 << n) / 1e9
rng = np.random.default_rng(0)
regions = (0, 4, 8, 16, 32)
print("k set      " + "  ".join(f"R{r:<4d}" for r in regions))
for k in (3, 4, 5):
    uk = W.haar_unitary(1 << k, rng)
    sets = {"high": [8 + 3 * j for j in range(k)], "top": [n - 1 - j for j in range(k)], "mid": [12 + j for j in range(k)],
            "mixed": [0, 3][: max(1, k - 3)] + [7 + 4 * j for j in range(k - max(1, k - 3))], "low": list(range(k)),
            "far": [20 + j for j in range(k)]}
    for label, bits in sets.items():
        row = []
        for r in regions:
            dev.set_option(_lib.OPT_TILE_REGIONS, r)
            row.append(gb / (timed(dev, lambda: dev.apply_matrix(uk, [n - 1 - b for b in bits])) * 1e-3))
        print(f"{k} {label:6s}  " + "  ".join(f"{v:5.0f}" for v in row))
