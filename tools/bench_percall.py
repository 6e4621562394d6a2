#!/usr/bin/env python3
"""Per-call cost at the reference's own register sizes: Gate.apply(ndarray), Simulator.run(ndarray) gate by gate and in
one launch (the LDS executor, n <= 13), and a batch of circuits in one launch.

    python tools/bench_percall.py
"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from quantum_computations_amd import workloads as W
from quantum_computations_amd.dv_simulator import gates as G
from quantum_computations_amd.dv_simulator.simulator import Simulator


def best(fn, reps=5):
    fn()
    out = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); out = min(out, time.perf_counter() - t0)
    return out


for n in (4, 8, 10, 12, 13, 16, 20, 24):
    ket = W.random_ket(n, 1)
    g = [G.H(n // 2), G.T(1), G.CX(0, n - 1)]
    for gate in g: gate.apply(ket)
    t0 = time.perf_counter(); reps = 50 if n <= 20 else 5
    for _ in range(reps):
        for gate in g: out = gate.apply(ket)
    dt = (time.perf_counter() - t0) / (3 * reps)
    depth = 1000 if n <= 13 else 100
    circ = W.to_gates(W.random_circuit(n, depth, 3))
    per_gate = best(lambda: Simulator(circ, single_launch=False).run(ket), 3)
    line = f"n={n:2d}: Gate.apply(ndarray) {dt*1e6:8.1f} us per call; Simulator.run, {depth} gates, gate by gate {per_gate/depth*1e6:7.2f} us per gate"
    if n <= 13:
        one = best(lambda: Simulator(circ).run(ket), 5)
        batch = [circ] * 512
        kets = [ket] * 512
        many = best(lambda: Simulator.run_batch(batch, kets), 2)
        from quantum_computations_amd.dv_simulator import program as P
        prog = P.compile_circuit(circ, n)
        run_only = best(lambda: P.run_programs([prog], [ket]), 5)
        many_only = best(lambda: P.run_programs([prog] * 512, kets), 2)
        line += (f"; in one launch {one/depth*1e6:6.3f} us per gate incl. the Python encoding of the circuit, {run_only/depth*1e6:6.3f} "
                 f"us per gate for an encoded program ({run_only*1e3:.2f} ms per circuit incl. copies)"
                 f"; 512 circuits in one launch {many/(512*depth)*1e6:6.3f} / {many_only/(512*depth)*1e6:6.4f} us per gate "
                 f"({many_only*1e3:.1f} ms for {512 * depth} gates)")
    print(line, flush=True)
