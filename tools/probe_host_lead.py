#!/usr/bin/env python3
"""How far the host runs ahead of the GPU on the benchmark circuit: time to issue 100 gates against time until they are done."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import workloads as W
from quantum_computations_amd.device import DeviceState
n = 28
gates = W.to_gates(W.random_circuit(n, 100, 100))
dev = DeviceState.random(n, 1)
for g in gates: g.apply(dev)
dev.sync()
for rec in (False, True):
    t0 = time.perf_counter()
    for i, g in enumerate(gates):
        if rec: dev.event_record(2 * i)
        g.apply(dev)
        if rec: dev.event_record(2 * i + 1)
    t1 = time.perf_counter()
    dev.sync()
    t2 = time.perf_counter()
    print(f"events={rec}: host loop {1e3 * (t1 - t0):.2f} ms, until GPU done {1e3 * (t2 - t0):.2f} ms")
