import sys, time
sys.path.insert(0, '.')
import numpy as np
from quantum_computations_amd import workloads as W
from quantum_computations_amd.dv_simulator import gates as G
from quantum_computations_amd.dv_simulator.simulator import Simulator
for n in (4, 10, 12, 16, 20, 24):
    ket = W.random_ket(n, 1)
    g = [G.H(n // 2), G.T(1), G.CX(0, n - 1)]
    for gate in g: gate.apply(ket)
    t0 = time.perf_counter(); reps = 50 if n <= 20 else 5
    for _ in range(reps):
        for gate in g: out = gate.apply(ket)
    dt = (time.perf_counter() - t0) / (3 * reps)
    circ = W.to_gates(W.random_circuit(n, 100, 3))
    Simulator(circ).run(ket)
    t0 = time.perf_counter(); Simulator(circ).run(ket); dt2 = time.perf_counter() - t0
    print(f"n={n:2d}: Gate.apply(ndarray) {dt*1e3:8.3f} ms per gate; Simulator.run(ndarray) of 100 gates {dt2*1e3:8.2f} ms")
