#!/usr/bin/env python3
"""1- and 2-qubit gates: the register form (k_dense / k_dense_ctrl: QSV_OPT_KQ_VARIANT = 1) against the workgroup-tile
form (k_dense_tile12), shipped tile order and fixed orders.  The data behind tile_regions() for k = 1, 2.

    python tools/probe_tile_12.py [n] [--pairs]     # --pairs: every pair of target bits >= 3 (about a minute)
"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib
from quantum_computations_amd import workloads as W
from quantum_computations_amd.device import DeviceState
from quantum_computations_amd.dv_simulator import gates as G


def timed(dev, fn, reps=6):
    fn(); dev.sync(); dev.timer_start()
    for _ in range(reps): fn()
    return dev.timer_stop() / reps


def row(dev, fn):
    cells = []
    for variant, regions in ((1, -1), (0, -1), (0, 0), (0, 2), (0, 4), (0, 8), (0, 16)):
        dev.set_option(_lib.OPT_KQ_VARIANT, variant)
        dev.set_option(_lib.OPT_TILE_REGIONS, regions)
        cells.append(f"{timed(dev, fn):.3f}")
    dev.set_option(_lib.OPT_KQ_VARIANT, 0)
    dev.set_option(_lib.OPT_TILE_REGIONS, -1)
    return " ".join(cells)


args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 28
dev = DeviceState.random(n, 1)
rng = np.random.default_rng(0)
u2, u4 = W.haar_unitary(2, rng), W.haar_unitary(4, rng)
print(f"# n = {n}; ms per launch: register form | tile form, shipped order | tile form, regions 0 2 4 8 16")
print("# 1-qubit gate, by target bit")
for bit in range(3, n):
    print(bit, row(dev, lambda: dev.apply_matrix(u2, [n - 1 - bit])), flush=True)
print("# CX, by (control, target) bit")
cx_rng = np.random.default_rng(3)
for c, t in [(int(c), int(t)) for c, t in (cx_rng.choice(np.arange(3, n), 2, replace=False) for _ in range(24))]:
    g = G.CX(n - 1 - c, n - 1 - t)
    print(c, t, row(dev, lambda: g.apply(dev)), flush=True)
print("# 2-qubit gate, by (lo, hi) target bits")
pairs = [(lo, hi) for lo in range(3, n) for hi in range(lo + 1, n)]
if "--pairs" not in sys.argv:
    pairs = [pairs[i] for i in np.random.default_rng(5).choice(len(pairs), 40, replace=False)]
for lo, hi in pairs:
    print(lo, hi, row(dev, lambda: dev.apply_matrix(u4, [n - 1 - lo, n - 1 - hi])), flush=True)
