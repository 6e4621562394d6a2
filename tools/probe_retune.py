#!/usr/bin/env python3
"""Re-sweep of the round-1 launch-shape defaults (items per thread, tile order) for the kernels that still serve the
benchmark circuit next to the tile form: k_dense<1, 1> (one target inside a wavefront), k_dense_ctrl with a control on
bits 0..2, k_diag (CZ)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib
from quantum_computations_amd import workloads as W
from quantum_computations_amd.device import DeviceState
from quantum_computations_amd.dv_simulator import gates as G


def timed(dev, fn, reps=8):
    fn(); dev.sync(); dev.timer_start()
    for _ in range(reps): fn()
    return dev.timer_stop() / reps


n = 28
dev = DeviceState.random(n, 1)
rng = np.random.default_rng(0)
u4 = W.haar_unitary(4, rng)
combos = [(0, -1)] + [(u, r) for u in (1, 2, 4) for r in (0, 8, 32)]
print("# columns: shipped | (unroll, regions) =", combos[1:])


def row(label, fn):
    cells = []
    for unroll, regions in combos:
        dev.set_option(_lib.OPT_UNROLL, unroll)
        dev.set_option(_lib.OPT_TILE_REGIONS, regions)
        cells.append(f"{timed(dev, fn):.3f}")
    dev.set_option(_lib.OPT_UNROLL, 0)
    dev.set_option(_lib.OPT_TILE_REGIONS, -1)
    print(label, " ".join(cells), dev.last_kernel(), flush=True)


print("# dense 2q, (lo < 6, hi >= 6)")
for lo in (0, 4):
    for hi in ((8, 13, 16, 20, 22, 25, 27) if len(sys.argv) < 2 else range(17, 28)):
        row(f"2q ({lo},{hi})", lambda: dev.apply_matrix(u4, [n - 1 - lo, n - 1 - hi]))
if len(sys.argv) > 1: sys.exit(0)
print("# CX, control on bits 0..2")
for c in (0, 2):
    for t in (3, 8, 13, 16, 20, 22, 25, 27):
        g = G.CX(n - 1 - c, n - 1 - t)
        row(f"CX c={c} t={t}", lambda: g.apply(dev))
print("# CZ")
for a, b in [(0, 1), (1, 9), (2, 20), (4, 5), (5, 17), (7, 8), (9, 20), (13, 26), (20, 21), (26, 27), (3, 27), (12, 13)]:
    g = G.CZ(n - 1 - a, n - 1 - b)
    row(f"CZ ({a},{b})", lambda: g.apply(dev))
