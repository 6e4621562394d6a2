#!/usr/bin/env python3
"""Read-out / reshaping kernels on one MI355X at n = 28: measurement (probabilities, collapse), insertion, qubit
permutation, 6-qubit dense gates, k-qubit diagonals, norm, Pauli expectation.  GB/s on each kernel's own algorithmic
bytes (stated per row), timed with HIP events on the register's stream.

    python tools/sweep_readout.py [--n 28] [--reps 5] [--out gpurun_out/sweep_readout.txt] [--csv profiles/...csv]
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from quantum_computations_amd import workloads as W  # noqa: E402
from quantum_computations_amd.device import DeviceState  # noqa: E402
from quantum_computations_amd.dv_simulator import gates as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=28)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default="gpurun_out/sweep_readout.txt")
    ap.add_argument("--csv", default="")
    args = ap.parse_args()
    n = args.n
    out = Path(args.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    log = out.open("w")
    rows = []

    def emit(line=""):
        print(line, flush=True)
        log.write(line + "\n")
        log.flush()

    def report(kernel, case, ms, gbytes, note):
        rate = gbytes / (ms * 1e-3)
        rows.append((kernel, case, ms, gbytes, rate, rate / 8000.0, note))
        emit(f"{kernel:18s} {case:34s} {ms:8.3f} ms  {gbytes:7.3f} GB  {rate:7.0f} GB/s  {rate / 80.0:5.1f} %   {note}")

    rng = np.random.default_rng(0)
    amp_gb = 16 * (1 << n) / 1e9                 # one pass over the register
    emit(f"# n={n}: register {16 * (1 << n) / 2**30:.2f} GiB; GB/s on the algorithmic bytes of each operation, % of 8 TB/s")
    dev = DeviceState.random(n, seed=1)
    eig = G.M(0, 0.9, 2.2).eigenvectors()

    def timed(fn, reps=args.reps, prepare=None):
        if prepare:
            prepare()
        fn()
        dev.sync()
        total = 0.0
        for _ in range(reps):
            if prepare:
                prepare()
                dev.sync()
            dev.timer_start()
            fn()
            total += dev.timer_stop()
        return total / reps

    emit("\n## M.apply, first pass: branch probabilities (read every amplitude once)")
    for bit in (0, 2, 5, 6, 12, 20, n - 1):
        q = n - 1 - bit
        ms = timed(lambda: dev.measure_probs(q, *eig))
        report("measure_probs", f"bit {bit}", ms, amp_gb, "read 16 B x 2^n (includes the host sum of the partials)")
    ms = timed(lambda: dev.norm2())
    report("norm2", "", ms, amp_gb, "read 16 B x 2^n")
    ms = timed(lambda: dev.expect_pauli("XZY", [0, n // 2, n - 1]))
    report("expect_pauli", "X(0) Z(n/2) Y(n-1)", ms, 2 * amp_gb, "reads psi[i] and psi[i ^ xmask]")

    emit("\n## reduced density matrices (read every amplitude once; X X^H on the f64 matrix cores)")
    for label, bits in {"k=1 low": [0], "k=1 high": [n - 1], "k=2 mixed": [1, 20], "k=3 low": [0, 1, 2], "k=3 high": [n - 3, n - 2, n - 1],
                        "k=4 mixed": [0, 5, 12, 25], "k=5 low": [0, 1, 2, 3, 4], "k=5 high": [10, 14, 18, 22, 26],
                        "k=6 low": [0, 1, 2, 3, 4, 5], "k=6 mixed": [2, 6, 11, 17, 23, n - 1]}.items():
        qs = [n - 1 - b for b in bits]
        ms = timed(lambda: dev.reduced_density(qs), reps=3)
        report("reduced_density", label, ms, amp_gb, f"{dev.last_kernel()} (includes the partial sums and the download of rho)")

    emit("\n## M.apply, second pass: collapse n -> n-1 qubits (read all, write half); then Insert back n-1 -> n")
    for bit in (0, 2, 5, 6, 12, 20, n - 1):
        q = n - 1 - bit
        # collapse shrinks the register, insert grows it back: time each on its own
        tc, ti = 0.0, 0.0
        for _ in range(args.reps + 1):
            dev.sync()
            dev.timer_start()
            dev.collapse(q, eig[0], 1.0)
            a = dev.timer_stop()
            dev.timer_start()
            dev.insert(q, [0.6, 0.8j])
            b = dev.timer_stop()
            if _:
                tc += a
                ti += b
        report("collapse", f"bit {bit}", tc / args.reps, 1.5 * amp_gb, "read 2^n, write 2^(n-1) amplitudes")
        report("insert", f"bit {bit}", ti / args.reps, 1.5 * amp_gb, "read 2^(n-1), write 2^n amplitudes")
    dev.fill_random(1)

    emit("\n## permute_tensor_product (read all, write all)")
    ident = list(range(n))
    perms = {"identity": ident,
             "swap qubits n-1, n-2 (bits 0,1)": ident[:-2] + [n - 1, n - 2][::-1] if False else ident[:-2] + [ident[-1], ident[-2]],
             "swap qubits 0, 1 (top bits)": [1, 0] + ident[2:],
             "rotate by one": ident[1:] + ident[:1],
             "rotate by seven": ident[7:] + ident[:7],
             "reverse": ident[::-1],
             "swap low 6 with next 6": ident[:n - 12] + ident[n - 6:] + ident[n - 12:n - 6],
             "random": [int(v) for v in rng.permutation(n)]}
    for label, order in perms.items():
        ms = timed(lambda: dev.permute(order))
        report("permute", label, ms, 2 * amp_gb, dev.last_kernel())

    emit("\n## 6-qubit dense gates (16 flop/B complex: bounded by the fp64 pipe, 78.6 TFLOP/s = 4.9 TB/s-equivalent)")
    u6 = W.haar_unitary(64, rng)
    r6 = np.linalg.qr(rng.standard_normal((64, 64)))[0]
    for label, bits in {"high": [8, 10, 12, 14, 16, 18], "top": [n - 1 - j for j in range(6)], "low": [0, 1, 2, 3, 4, 5],
                        "mixed": [0, 3, 5, 9, 13, 20]}.items():
        qs = [n - 1 - b for b in bits]
        for kind, u in (("complex", u6), ("real", r6)):
            ms = timed(lambda: dev.apply_matrix(u, qs), reps=3)
            flops = (8 if kind == "complex" else 4) * 64 * (1 << n)
            report("apply_kq k=6", f"{label} {kind}", ms, 2 * amp_gb,
                   f"{dev.last_kernel()}  {flops / (ms * 1e-3) / 1e12:.1f} TFLOP/s")

    emit("\n## k-qubit diagonals")
    for k, bits in ((3, [0, 7, 20]), (4, [1, 2, 3, 4]), (6, [0, 5, 9, 13, 20, n - 1])):
        d = np.exp(1j * rng.uniform(0, 6.28, 1 << k))
        qs = [n - 1 - b for b in bits]
        ms = timed(lambda: dev.apply_matrix(np.diag(d), qs))
        report("diag k-qubit", f"k={k} bits {bits}", ms, 2 * amp_gb, dev.last_kernel())
    log.close()
    if args.csv:
        import csv
        with open(args.csv, "w", newline="") as fh:
            wr = csv.writer(fh)
            wr.writerow(["kernel", "case", "avg_ms", "algorithmic_GB", "GB_per_s", "frac_of_8TBps", "note"])
            for r in rows:
                wr.writerow([r[0], r[1], f"{r[2]:.4f}", f"{r[3]:.4f}", f"{r[4]:.1f}", f"{r[5]:.4f}", r[6]])


if __name__ == "__main__":
    main()
