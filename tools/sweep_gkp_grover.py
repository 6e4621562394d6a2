#!/usr/bin/env python3
"""The reference's Grover-on-GKP experiment across squeezing levels (grover.py:93-96: dB = linspace(5, 15, 13)[2:], its grid and
truncation settings), a few seeds each: logical success probability of one Grover iteration on three GKP qubits.

    python tools/sweep_gkp_grover.py [--seeds 3] [--out FILE]
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--d", type=int, default=1000)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    from quantum_computations_amd import workloads as W
    from quantum_computations_amd.dv_simulator import gates as dv
    from quantum_computations_amd.dv_simulator.states import State
    from quantum_computations_amd.gkp_simulator.simulator import Simulator
    from quantum_computations_amd.gkp_simulator.transpiler import MBGKPCircuit, parse_to_mps
    from quantum_computations_amd.gkp_simulator.utils import db2eps, full_logical_density_mps, syndrome_matrix

    circuit = []
    for gate in W.to_gates(W.grover3_ops([2, 7])[3:]):
        if isinstance(gate, dv.CX):
            circuit += [dv.H(gate.target), dv.CZ(*gate.indices), dv.H(gate.target)]
        else:
            circuit.append(gate)
    layered = MBGKPCircuit.transpile(circuit)
    layered.fill()
    qs = np.linspace(-20.0, 20.0, args.d)
    options = {"rel_err": 1e-2, "max_bond_dim": 100}
    levels = np.linspace(5, 15, 13)[2:]
    rows = []
    started = time.perf_counter()
    for db in levels:
        eps = db2eps(db)
        probs, secs = [], []
        for seed in range(args.seeds):
            sim = Simulator(layered, eps, rng_seed=1000 * seed + int(round(10 * db)), svd_options=options)
            t0 = time.perf_counter()
            mps, frame = sim.run(parse_to_mps([State.ZERO] * 3, eps, qs))
            rho = full_logical_density_mps(mps, normalised=True)
            secs.append(time.perf_counter() - t0)
            fix = syndrome_matrix(frame)
            rho = fix @ rho @ fix.conj().T
            probs.append(float(np.real(rho[2, 2] + rho[7, 7])))
            mps.reg.close()
        rows.append({"dB": float(db), "epsilon": float(eps), "success_probability": probs, "mean": float(np.mean(probs)),
                     "seconds_per_run": float(np.mean(secs))})
        print(f"{db:5.2f} dB: success {np.mean(probs):.4f}  ({np.mean(secs):.2f} s per run)", flush=True)
    result = {"experiment": "one Grover iteration on 3 GKP qubits (oracle |010>, |111>), 95 gadgets, d=1000, "
                            "max_bond_dim=100, rel_err=1e-2", "seeds_per_level": args.seeds, "levels": rows,
              "total_seconds": time.perf_counter() - started}
    line = json.dumps(result)
    print(line)
    if args.out:
        Path(args.out).write_text(line + "\n")


if __name__ == "__main__":
    main()
