#!/usr/bin/env python3
"""Dense 2-qubit gates on every pair of target bits >= 3: shipped k_dense against the workgroup-tile form by tile order."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib
from quantum_computations_amd import workloads as W
from quantum_computations_amd.device import DeviceState


def timed(dev, fn, reps=5):
    fn(); dev.sync(); dev.timer_start()
    for _ in range(reps): fn()
    return dev.timer_stop() / reps


n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
dev = DeviceState.random(n, 1)
u4 = W.haar_unitary(4, np.random.default_rng(0))
print("# lo hi: shipped | tile regions 0 2 4 8 16")
for lo in range(3, n):
    for hi in range(lo + 1, n):
        cells = []
        for variant, regions in ((0, -1), (4, 0), (4, 2), (4, 4), (4, 8), (4, 16)):
            dev.set_option(_lib.OPT_KQ_VARIANT, variant)
            dev.set_option(_lib.OPT_TILE_REGIONS, regions)
            cells.append(f"{timed(dev, lambda: dev.apply_matrix(u4, [n - 1 - lo, n - 1 - hi])):.3f}")
        print(lo, hi, " ".join(cells), flush=True)
