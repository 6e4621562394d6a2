#!/usr/bin/env python3
"""Tile kernel (variant 4) on contiguous target windows [s, s+k): ms by tile order, against the shipped choice."""
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


n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
dev = DeviceState.random(n, 1)
rng = np.random.default_rng(0)
for k, real in ((3, False),) if len(sys.argv) > 2 else ((4, False), (5, True)):
    u = np.linalg.qr(rng.standard_normal((1 << k, 1 << k)))[0] if real else W.haar_unitary(1 << k, rng)
    print(f"# k={k} {'real' if real else 'complex'}: start bit: shipped | tile regions 0, 1 (scrambled), 2, 4, 8", flush=True)
    for s in range(3, n - k + 1):
        qs = [n - 1 - (s + j) for j in range(k)]
        cells = []
        for variant, regions in ((0, -1), (4, 0), (4, 1), (4, 2), (4, 4), (4, 8)):
            dev.set_option(_lib.OPT_KQ_VARIANT, variant)
            dev.set_option(_lib.OPT_TILE_REGIONS, regions)
            cells.append(f"{timed(dev, lambda: dev.apply_matrix(u, qs)):6.3f}")
        print(f"k={k} s={s:2d}  {cells[0]} | " + " ".join(cells[1:]), flush=True)
    print(f"# k={k}: random target sets (bits >= 3)", flush=True)
    for trial in range(16):
        bits = sorted(int(b) for b in rng.choice(np.arange(3, n), size=k, replace=False))
        qs = [n - 1 - b for b in bits]
        cells = []
        for variant, regions in ((0, -1), (4, 0), (4, 1), (4, 2), (4, 4), (4, 8)):
            dev.set_option(_lib.OPT_KQ_VARIANT, variant)
            dev.set_option(_lib.OPT_TILE_REGIONS, regions)
            cells.append(f"{timed(dev, lambda: dev.apply_matrix(u, qs)):6.3f}")
        print(f"k={k} {str(bits):24s}  {cells[0]} | " + " ".join(cells[1:]), flush=True)
dev.set_option(_lib.OPT_KQ_VARIANT, 0)
dev.set_option(_lib.OPT_TILE_REGIONS, -1)
