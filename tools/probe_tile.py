#!/usr/bin/env python3
"""k = 4, 5 dense gates: the workgroup-tile kernel (variant 4) against the shipped choice, by tile order (regions)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib
from quantum_computations_amd import workloads as W
from quantum_computations_amd.device import DeviceState


def timed(dev, fn, reps=10):
    fn(); dev.sync(); dev.timer_start()
    for _ in range(reps): fn()
    return dev.timer_stop() / reps


n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
dev = DeviceState.random(n, 1)
gb = 2 * 16 * (1 << n) / 1e9
rng = np.random.default_rng(0)
for k in (4, 5):
    uc = W.haar_unitary(1 << k, rng)
    ur = np.linalg.qr(rng.standard_normal((1 << k, 1 << k)))[0]
    sets = {"high": [8 + 3 * j for j in range(k)], "top": [n - 1 - j for j in range(k)], "far": [20 + j for j in range(k)],
            "b5+": [5] + [7 + 4 * j for j in range(k - 1)], "b3,4,5+": [3, 4, 5] + [7 + 4 * j for j in range(k - 3)],
            "mid": [6 + j for j in range(k)], "mode4": [5 + j for j in range(k)], "mode3": [10 + j for j in range(k)],
            "mode2": [15 + j for j in range(k)], "18+": [18 + j for j in range(k)], "spread": [6, 12, 18, 24, 27][:k]}
    for label, bits in sets.items():
        qs = [n - 1 - b for b in bits]
        cells = []
        for u in (uc, ur):
            for variant, regions in ((0, -1), (4, 0), (4, 8), (4, 32)):
                dev.set_option(_lib.OPT_KQ_VARIANT, variant)
                dev.set_option(_lib.OPT_TILE_REGIONS, regions)
                ms = timed(dev, lambda: dev.apply_matrix(u, qs))
                cells.append(f"{ms:6.3f}")
            cells.append("|")
        print(f"k={k} {label:8s} {str(bits):24s} shipped, tile r0, r8, r32 (complex | real): " + " ".join(cells), flush=True)
dev.set_option(_lib.OPT_KQ_VARIANT, 0)
dev.set_option(_lib.OPT_TILE_REGIONS, -1)
