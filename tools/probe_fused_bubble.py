#!/usr/bin/env python3
"""Fused benchmark circuit: wall time per pass against the sum of its kernels' durations (run under rocprofv3)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import workloads as W
from quantum_computations_amd.device import DeviceState
from quantum_computations_amd.fusion import fuse_circuit

n = int(sys.argv[2]) if len(sys.argv) > 2 else 28
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
gates = W.to_gates(W.random_circuit(n, 100, 100))
fused = fuse_circuit(gates, k, n_qubits=n)
dev = DeviceState.random(n, 1)
for g in fused: g.apply(dev)
dev.sync()
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    for g in fused: g.apply(dev)
dev.sync()
print(f"n={n} k={k}: {len(fused)} launches, wall {1e3 * (time.perf_counter() - t0) / reps:.3f} ms per pass", flush=True)
