#!/usr/bin/env python3
"""Complex 5- and 6-qubit blocks: three real multiplications per matrix entry (QSV_OPT_COMPLEX_PRODUCT = 0 / 3) against
four (= 4), per kernel form and target placement.  ms per launch at n qubits, and the largest difference between the
two results on sampled amplitudes.

    python tools/probe_complex_product.py [n]
"""
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
ref = dev.copy()
rng = np.random.default_rng(0)
gb = 2 * 16 * (1 << n) / 1e9
probe = rng.integers(0, 1 << n, 32)
for k in (5, 6):
    u = W.haar_unitary(1 << k, rng)
    sets = {"high": [8 + 3 * j for j in range(k)], "top": [n - 1 - j for j in range(k)], "far": [20 + j for j in range(k)],
            "window 6..": [6 + j for j in range(k)], "window 3..": [3 + j for j in range(k)],
            "1 low (b5)": [5] + [7 + 4 * j for j in range(k - 1)], "1 low (b0)": [0] + [7 + 4 * j for j in range(k - 1)],
            "2 low (b0,b3)": [0, 3] + [7 + 4 * j for j in range(k - 2)], "3 low (b0,b1,b2)": [0, 1, 2] + [7 + 4 * j for j in range(k - 3)],
            "3 low (b3,b4,b5)": [3, 4, 5] + [7 + 4 * j for j in range(k - 3)], "5 low (b0..b4)": [0, 1, 2, 3, 4] + [9] * (k - 5),
            "scattered": sorted(rng.choice(np.arange(3, n), k, replace=False).tolist())}
    variants = (0, 3, 6) if k == 5 else (0,)
    print(f"# k = {k}, n = {n}: ms (TB/s) per kernel form; each cell: 4 multiplications | 3 multiplications; then max |diff| on 32 sampled amplitudes")
    print("# forms: 0 = shipped choice, 3 = k_dense_lds, 6 = k_dense_mtile5 (tile-fed matrix cores; its cells are the same kernel twice)")
    for label, bits in sets.items():
        qs = [n - 1 - b for b in bits]
        cells, worst = [], 0.0
        for variant in variants:
            dev.set_option(_lib.OPT_KQ_VARIANT, variant)
            pair, outs = [], []
            for cp in (4, 3):
                dev.set_option(_lib.OPT_COMPLEX_PRODUCT, cp)
                ref.copy_into(dev)
                dev.apply_matrix(u, qs)
                outs.append(np.array([dev.download(int(i), 1)[0] for i in probe]))
                name = dev.last_kernel()
                ms = timed(dev, lambda: dev.apply_matrix(u, qs))
                pair.append(f"{ms:.3f} ({gb / ms:.2f})")
            worst = max(worst, float(np.max(np.abs(outs[0] - outs[1]))))
            cells.append(f"[{variant}] " + " | ".join(pair) + f" {name}")
        print(f"k={k} {label:18s} {str(bits):26s} " + "   ".join(cells) + f"   diff {worst:.1e}", flush=True)
dev.set_option(_lib.OPT_KQ_VARIANT, 0)
dev.set_option(_lib.OPT_COMPLEX_PRODUCT, 0)
