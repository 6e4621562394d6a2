#!/usr/bin/env python3
"""Reduced density matrix of k kept qubits at n = 28: wall time per call (kernel + partial sums + download of rho)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd.device import DeviceState

n = 28
dev = DeviceState.random(n, 1)
for bits in ([0], [27], [0, 1, 2], [0, 5, 12, 25], [0, 1, 2, 3, 4], [0, 1, 2, 3, 4, 5]):
    qs = [n - 1 - b for b in bits]
    dev.reduced_density(qs)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        dev.reduced_density(qs)
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"kept bits {bits}: {ms:6.3f} ms  {16 * 2**n / ms / 1e6:6.0f} GB/s  {dev.last_kernel()}", flush=True)
