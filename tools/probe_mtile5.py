#!/usr/bin/env python3
"""The tile-fed matrix-core kernel for complex 5-qubit blocks (k_dense_mtile5, QSV_OPT_KQ_VARIANT = 6) by tile order,
against the shipped vector kernels.   python tools/probe_mtile5.py [n]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib
from quantum_computations_amd import workloads as W
from quantum_computations_amd.device import DeviceState


def timed(dev, fn, reps=6):
    fn(); dev.sync(); dev.timer_start()
    for _ in range(reps): fn()
    return dev.timer_stop() / reps


n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
dev = DeviceState.random(n, 1)
rng = np.random.default_rng(0)
u = W.haar_unitary(32, rng)
sets = {"high": [8, 11, 14, 17, 20], "top": [n - 1 - j for j in range(5)], "far": [20, 21, 22, 23, 24], "window 6": [6, 7, 8, 9, 10],
        "window 3": [3, 4, 5, 6, 7], "b5 + high": [5, 7, 11, 15, 19], "b3-5 + high": [3, 4, 5, 7, 11], "window 12": [12, 13, 14, 15, 16],
        "window 17": [17, 18, 19, 20, 21]}
for i in range(8):
    sets[f"scattered {i}"] = sorted(int(b) for b in rng.choice(np.arange(3, n), 5, replace=False))
print(f"# n = {n}: ms per launch; vector kernels (variant 1 / 3 as shipped in round 2) | k_dense_mtile5 with regions -1 (rule) 0 2 4 8 16 32")
for label, bits in sets.items():
    qs = [n - 1 - b for b in bits]
    dev.set_option(_lib.OPT_COMPLEX_PRODUCT, 4)
    dev.set_option(_lib.OPT_KQ_VARIANT, 0)
    dev.set_option(_lib.OPT_TILE_REGIONS, -1)
    base = timed(dev, lambda: dev.apply_matrix(u, qs))
    name = dev.last_kernel()
    dev.set_option(_lib.OPT_COMPLEX_PRODUCT, 0)
    dev.set_option(_lib.OPT_KQ_VARIANT, 6)
    cells = []
    for regions in (-1, 0, 2, 4, 8, 16, 32):
        dev.set_option(_lib.OPT_TILE_REGIONS, regions)
        cells.append(f"{timed(dev, lambda: dev.apply_matrix(u, qs)):.3f}")
    print(f"{label:14s} {str(bits):24s} {base:.3f} {name:34s} | " + " ".join(cells) + f"  {dev.last_kernel()}", flush=True)
