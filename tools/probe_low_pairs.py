#!/usr/bin/env python3
"""2-qubit gates with both targets inside a wavefront (k_dense<0, 2, U>) and 1-qubit gates on bits 0..2 (k_dense<0, 1, U>):
ms by items per thread (QSV_OPT_UNROLL) and tile order (QSV_OPT_TILE_REGIONS)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib
from quantum_computations_amd import workloads as W
from quantum_computations_amd.device import DeviceState


def timed(dev, fn, reps=8):
    fn(); dev.sync(); dev.timer_start()
    for _ in range(reps): fn()
    return dev.timer_stop() / reps


n = 28
dev = DeviceState.random(n, 1)
rng = np.random.default_rng(0)
u2, u4 = W.haar_unitary(2, rng), W.haar_unitary(4, rng)
combos = [(0, -1)] + [(u, r) for u in (1, 2, 4) for r in (0, 8, 32)]
print("# columns: shipped | (unroll, regions) =", combos[1:])
for bits in ([0], [1], [2], [0, 1], [1, 4], [2, 5], [3, 4], [0, 5], [4, 5], [2, 3]):
    qs = [n - 1 - b for b in bits]
    u = u2 if len(bits) == 1 else u4
    cells = []
    for unroll, regions in combos:
        dev.set_option(_lib.OPT_UNROLL, unroll)
        dev.set_option(_lib.OPT_TILE_REGIONS, regions)
        cells.append(f"{timed(dev, lambda: dev.apply_matrix(u, qs)):.3f}")
    dev.set_option(_lib.OPT_UNROLL, 0)
    dev.set_option(_lib.OPT_TILE_REGIONS, -1)
    print(bits, " ".join(cells), dev.last_kernel(), flush=True)
