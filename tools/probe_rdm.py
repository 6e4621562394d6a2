#!/usr/bin/env python3
"""Reduced density matrix of k kept qubits at n = 28: wall time per call (kernel + partial sums + download of rho), the
LDS-staged workgroup tile (k_rdm_tile, shipped) against round 2's per-lane row loads (k_rdm), low / high / scattered
kept-bit sets, and the largest difference between the two results."""
import sys
import time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib
from quantum_computations_amd.device import DeviceState

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
dev = DeviceState.random(n, 1)
sets = ([0], [n - 1], [13], [0, 1, 2], [3, 9, 20], [0, 1, 2, 3], [n - 4, n - 3, n - 2, n - 1], [0, 5, 12, 25], [6, 7, 8, 9], [2, 11, 17, 26],
        [0, 1, 2, 3, 4], [n - 5, n - 4, n - 3, n - 2, n - 1], [0, 5, 12, 19, 25], [4, 9, 14, 20, 26],
        [0, 1, 2, 3, 4, 5], [n - 6, n - 5, n - 4, n - 3, n - 2, n - 1], [0, 5, 10, 15, 20, 25], [3, 8, 13, 18, 22, 27])
print(f"# n = {n}: ms per call (wall: kernel + partial sums + download) and GB/s on the 16 x 2^n bytes read; k_rdm_tile with the shipped tile order | plain order | 8 regions | 64 regions | round-2 form (k_rdm); max |difference| shipped vs round 2")
for bits in sets:
    qs = [n - 1 - b for b in bits]
    cells, rhos = [], []
    for variant, regions in ((0, -1), (0, 0), (0, 8), (0, 64), (2, -1)):
        dev.set_option(_lib.OPT_READOUT_VARIANT, variant)
        dev.set_option(_lib.OPT_TILE_REGIONS, regions)
        rhos.append(dev.reduced_density(qs))
        dev.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            dev.reduced_density(qs)
        ms = (time.perf_counter() - t0) / 10 * 1e3
        cells.append(f"{ms:6.3f} ms {16 * 2**n / ms / 1e6:5.0f}")
    dev.set_option(_lib.OPT_READOUT_VARIANT, 0)
    dev.set_option(_lib.OPT_TILE_REGIONS, -1)
    print(f"kept bits {str(bits):28s} " + " | ".join(cells) + f"   diff {np.max(np.abs(rhos[0] - rhos[-1])):.1e}  trace {np.trace(rhos[0]).real:.12f}", flush=True)
