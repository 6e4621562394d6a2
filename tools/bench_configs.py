#!/usr/bin/env python3
"""Secondary BASELINE.json configurations that fit one MI355X, plus the host<->device transfer rates.
(Round 2: ``bench.py --config cfg3|cfg4|cfg5`` and the ``secondary`` block of the default bench line cover the same
configurations in the driver-visible line; this script stays as the longer stand-alone form.)

    python tools/bench_configs.py [--out gpurun_out/configs.json] [--grover-n 30] [--skip-cv]

* cfg5 (single-GPU form): n-qubit Grover, k = 8 iterations, success probability vs sin^2(17 asin 2^-n/2).
* cfg4: 6 modes x Fock cutoff d = 32 (2^30 amplitudes, 16 GiB), 60 gates alternating single-mode squeezing
  S(r = 0.1 k mod 0.5) on mode k mod 6 and BS(i, i+1, pi/4); algorithmic bytes 32 GiB per gate.
* PCIe: upload / download of the 4 GiB 28-qubit register (the cost of the reference's ndarray calling convention).
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from quantum_computations_amd import workloads as W  # noqa: E402
from quantum_computations_amd.cv_simulator import fock  # noqa: E402
from quantum_computations_amd.device import DeviceState  # noqa: E402


def grover(n: int, iterations: int) -> dict:
    marked = (0b1011001110001111 << max(0, n - 16)) % (1 << n) | 1
    dev = DeviceState.zeros(n)
    h = W.G.H(0).matrix
    for q in range(n):
        dev.apply_matrix(h, [q])
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(iterations):
        W.grover_iteration(dev, n, marked)
    dev.sync()
    dt = time.perf_counter() - t0
    p = float(dev.probabilities([marked])[0])
    want = W.grover_success_probability(n, iterations)
    gates = iterations * W.grover_gate_count(n, marked)
    return {"config": f"Grover n={n}, {iterations} iterations, 1 GPU", "success_probability": p, "analytic": want,
            "rel_err": abs(p - want) / want, "gate_apps": gates, "seconds": dt, "gate_apps_per_s": gates / dt,
            "norm2": dev.norm2()}


def cv_fock(n_modes: int = 6, d: int = 32, gates: int = 60) -> dict:
    st = fock.FockState(n_modes, d)
    seq = []
    for k in range(gates // 2):
        seq.append(fock.S(k % n_modes, 0.1 * k % 0.5, 0.0))
        i = k % (n_modes - 1)
        seq.append(fock.BS(i, i + 1, np.pi / 4))
    # warm-up (also builds the host matrices once per distinct gate; expm of 1024 x 1024 takes a while)
    mats = {}
    for g in seq:
        key = (type(g).__name__, g.arg)
        if key not in mats:
            mats[key] = (fock.squeeze_matrix(d, g.arg, 0.0) if isinstance(g, fock.S)
                         else fock.beamsplitter_blocks(d, g.arg))
    per_gate = {"S": [], "BS": []}
    st.reg.sync()
    t_all = time.perf_counter()
    for g in seq:
        st.reg.timer_start()
        if isinstance(g, fock.S):
            st.reg.apply_mode(mats[("S", g.arg)], g.index)
        else:
            st.reg.apply_two_mode_blocks(mats[("BS", g.arg)], g.index1, g.index2)
        per_gate[type(g).__name__].append(st.reg.timer_stop())
    st.reg.sync()
    dt = time.perf_counter() - t_all
    gbytes = 2 * 16 * d ** n_modes / 1e9
    return {"config": f"CV Fock path: {n_modes} modes x d={d} ({16 * d ** n_modes / 2**30:.0f} GiB), {len(seq)} gates",
            "seconds": dt, "gate_apps_per_s": len(seq) / dt, "norm2": st.reg.norm2(),
            "algorithmic_GB_per_gate": gbytes,
            "S_avg_ms": float(np.mean(per_gate["S"])), "S_GBps": gbytes / (np.mean(per_gate["S"]) * 1e-3),
            "BS_avg_ms": float(np.mean(per_gate["BS"])), "BS_GBps": gbytes / (np.mean(per_gate["BS"]) * 1e-3),
            "BS_blocks": len(mats[("BS", np.pi / 4)]),
            "BS_ms_by_pair": {f"({i},{i + 1})": float(np.mean(per_gate["BS"][i::n_modes - 1])) for i in range(n_modes - 1)}}


def literal_dense(n: int = 12) -> dict:
    """the reference's literal algorithm (dense 2^N x 2^N operator by kron + permutation, then a
    mat-vec) restated with the package's small-N host helpers, timed on THIS host to calibrate it against the build
    container where the real reference was timed."""
    import statistics

    from quantum_computations_amd.dv_simulator import numpy_quantum as npq
    rng = np.random.default_rng(0)
    ket = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    out = {}
    for name, matrix, targets in [("H", npq.H, [n // 2]), ("T", npq.axis_rotation(np.pi / 4, [0, 0, 1]), [n // 2]),
                                  ("CX", npq.CX, [1, n - 2])]:
        times = []
        for _ in range(4):
            t0 = time.perf_counter()
            full = npq.expand_gate(matrix, n, targets)
            _ = full @ ket
            times.append(time.perf_counter() - t0)
        out[name + "_ms"] = 1e3 * statistics.median(times[1:])
    return {"config": f"literal dense algorithm (expand_gate + mat-vec) at n={n} on this host, NumPy "
                      f"{np.__version__}, {len(__import__('os').sched_getaffinity(0))} cores visible", **out}


def pcie(n: int = 28) -> dict:
    dev = DeviceState.random(n, 1)
    t0 = time.perf_counter()
    host = dev.to_numpy()
    t_down = time.perf_counter() - t0
    t0 = time.perf_counter()
    dev.upload(host)
    t_up = time.perf_counter() - t0
    gib = host.nbytes / 2 ** 30
    return {"config": f"host<->device copy of the {n}-qubit register ({gib:.0f} GiB, pageable NumPy memory)",
            "download_s": t_down, "upload_s": t_up, "download_GBps": host.nbytes / t_down / 1e9,
            "upload_GBps": host.nbytes / t_up / 1e9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/configs.json")
    ap.add_argument("--grover-n", type=int, default=30)
    ap.add_argument("--skip-cv", action="store_true")
    args = ap.parse_args()
    results = []
    for fn, a in [(pcie, ()), (literal_dense, ()), (grover, (args.grover_n, 8))] + ([] if args.skip_cv else [(cv_fock, ())]):
        t0 = time.time()
        r = fn(*a)
        r["wall_s"] = time.time() - t0
        print(json.dumps(r), flush=True)
        results.append(r)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(results, indent=1))


if __name__ == "__main__":
    main()
