#!/usr/bin/env python3
"""The fused blocks of the benchmark circuit (Simulator(fuse=k)): ms per block by kernel variant."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib
from quantum_computations_amd import workloads as W
from quantum_computations_amd.device import DeviceState
from quantum_computations_amd.fusion import fuse_circuit

n, depth, seed = 28, 100, 100
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
gates = W.to_gates(W.random_circuit(n, depth, seed))
fused = fuse_circuit(gates, k, n_qubits=n)
dev = DeviceState.random(n, 1)


def timed(fn, reps=5):
    fn(); dev.sync(); dev.timer_start()
    for _ in range(reps): fn()
    return dev.timer_stop() / reps


variants = (0, 1, 3, 4, 5)
tot = {v: 0.0 for v in variants}
for g in fused:
    bits = [n - 1 - q for q in g.indices]
    cells = []
    for v in variants:
        dev.set_option(_lib.OPT_KQ_VARIANT, v)
        ms = timed(lambda: g.apply(dev))
        tot[v] += ms
        cells.append(f"{ms:6.3f}")
        if v == 0: name = dev.last_kernel()
    real = not np.iscomplexobj(g.matrix) or not np.any(np.asarray(g.matrix).imag)
    print(f"{len(bits)}q bits {str(sorted(bits)):28s} {'real' if real else 'cplx'}  variants {variants}: " + " ".join(cells) + "  " + name, flush=True)
print("total ms", {v: round(t, 2) for v, t in tot.items()})
dev.set_option(_lib.OPT_KQ_VARIANT, 0)
