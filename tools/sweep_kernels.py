#!/usr/bin/env python3
"""Per-kernel bandwidth sweep on one MI355X: every target bit, several launch shapes.

    python tools/sweep_kernels.py [--n 28] [--reps 10] [--out gpurun_out/sweep.txt] [--quick]

Algorithmic bytes per gate = 2 * 16 * 2^n.  Timing: HIP events on the register's stream around
`reps` back-to-back launches (after one warm-up launch).
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from quantum_computations_amd import _lib  # noqa: E402
from quantum_computations_amd import workloads as W  # noqa: E402
from quantum_computations_amd.device import DeviceState  # noqa: E402


def timed(dev, fn, reps):
    fn()
    dev.sync()
    dev.timer_start()
    for _ in range(reps):
        fn()
    return dev.timer_stop() / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=28)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--out", default="gpurun_out/sweep.txt")
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    n = args.n
    out = Path(args.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    log = out.open("w")

    def emit(line=""):
        print(line, flush=True)
        log.write(line + "\n")
        log.flush()

    rng = np.random.default_rng(0)
    dev = DeviceState.random(n, seed=1)
    gbytes = 2 * 16 * (1 << n) / 1e9
    emit(f"# n={n} register {16 * (1 << n) / 2**30:.2f} GiB, algorithmic {gbytes:.3f} GB per gate, reps={args.reps}")
    u2, u4 = W.haar_unitary(2, rng), W.haar_unitary(4, rng)

    variants = [(0, 0, 0)] if args.quick else [(0, 0, 0), (1, 0, 0), (2, 0, 0), (4, 0, 0), (8, 0, 0), (0, 1, 0),
                                                (1, 1, 0), (2, 1, 0), (4, 1, 0), (8, 1, 0), (1, 0, 65536)]
    emit("\n## dense 1q: GB/s by target bit (bit = n-1-q); columns = (unroll, nontemporal, grid cap)")
    emit("bit  " + "  ".join(f"u{u}/nt{nt}/g{g:<6d}" for u, nt, g in variants))
    rows = {}
    for u, nt, g in variants:
        dev.set_option(_lib.OPT_UNROLL, u)
        dev.set_option(_lib.OPT_NONTEMPORAL, nt)
        dev.set_option(_lib.OPT_GRID_CAP, g)
        for bit in range(n):
            q = n - 1 - bit
            ms = timed(dev, lambda: dev.apply_matrix(u2, [q]), args.reps)
            rows.setdefault(bit, []).append(gbytes / (ms * 1e-3))
    for bit in range(n):
        emit(f"{bit:3d}  " + "  ".join(f"{v:14.0f}" for v in rows[bit]))
    dev.set_option(_lib.OPT_UNROLL, 0)
    dev.set_option(_lib.OPT_NONTEMPORAL, 1)
    dev.set_option(_lib.OPT_GRID_CAP, 0)

    emit("\n## dense 2q: GB/s for (bit0, bit1) samples, default launch shape")
    pairs = [(0, 1), (1, 0), (0, 5), (2, 9), (9, 2), (5, 6), (6, 7), (7, 6), (6, n - 1), (n - 1, 6), (10, 20),
             (20, 10), (n - 2, n - 1), (n - 1, n - 2), (0, n - 1), (3, n - 1), (13, 14)]
    for b0, b1 in pairs:
        ms = timed(dev, lambda: dev.apply_matrix(u4, [n - 1 - b0, n - 1 - b1]), args.reps)
        emit(f"({b0:2d},{b1:2d})  {gbytes / (ms * 1e-3):8.0f} GB/s   {ms:7.3f} ms")
    if not args.quick:
        emit("\n## dense 2q variants: GB/s; columns = (unroll, nontemporal)")
        combos = [(u, nt) for nt in (0, 1) for u in (1, 2, 4)]
        emit("pair     " + "  ".join(f"u{u}/nt{nt}" for u, nt in combos))
        for b0, b1 in [(0, 1), (2, 9), (5, 6), (6, 7), (10, 20), (6, n - 1), (20, 25), (n - 2, n - 1), (13, 14)]:
            vals = []
            for u, nt in combos:
                dev.set_option(_lib.OPT_UNROLL, u)
                dev.set_option(_lib.OPT_NONTEMPORAL, nt)
                ms = timed(dev, lambda: dev.apply_matrix(u4, [n - 1 - b0, n - 1 - b1]), args.reps)
                vals.append(gbytes / (ms * 1e-3))
            emit(f"({b0:2d},{b1:2d})  " + "  ".join(f"{v:6.0f}" for v in vals))
        emit("\n## general diagonal 1q variants: GB/s; columns = (unroll, nontemporal)")
        combos = [(u, nt) for nt in (0, 1) for u in (1, 2, 4, 8)]
        emit("bit   " + "  ".join(f"u{u}/nt{nt}" for u, nt in combos))
        dd = np.exp(1j * rng.uniform(0, 6.28, 2))
        for bit in (0, 12, n - 1):
            vals = []
            for u, nt in combos:
                dev.set_option(_lib.OPT_UNROLL, u)
                dev.set_option(_lib.OPT_NONTEMPORAL, nt)
                ms = timed(dev, lambda: dev.apply_diagonal(dd, [n - 1 - bit]), args.reps)
                vals.append(gbytes / (ms * 1e-3))
            emit(f"{bit:3d}   " + "  ".join(f"{v:6.0f}" for v in vals))
        dev.set_option(_lib.OPT_UNROLL, 0)
        dev.set_option(_lib.OPT_NONTEMPORAL, 1)

    emit("\n## register-blocked k-qubit dense gates (k = 3..5): GB/s by target-bit set")
    for k in (3, 4, 5):
        uk = W.haar_unitary(1 << k, rng)
        sets = {"high": [8 + 3 * j for j in range(k)], "top": [n - 1 - j for j in range(k)],
                "mixed": [0, 3][: max(1, k - 3)] + [7 + 4 * j for j in range(k - max(1, k - 3))],
                "low": list(range(k)), "far": [20 + j for j in range(k)]}
        for label, bits in sets.items():
            for nt in (1, 0):
                dev.set_option(_lib.OPT_NONTEMPORAL, nt)
                ms = timed(dev, lambda: dev.apply_matrix(uk, [n - 1 - b for b in bits]), max(3, args.reps // 2))
                emit(f"k={k} {label:6s} bits={str(bits):22s} nt={nt}  {ms:7.3f} ms  {gbytes / (ms * 1e-3):7.0f} GB/s"
                     f"  {dev.last_kernel()}")
    dev.set_option(_lib.OPT_NONTEMPORAL, 1)

    emit("\n## specialised kernels (credited with the full algorithmic bytes): ms and equivalent GB/s")
    d1 = np.exp(1j * rng.uniform(0, 6.28, 2))
    d2 = np.exp(1j * rng.uniform(0, 6.28, 4))
    tests = []
    for bit in (0, 3, 6, 12, n - 1):
        q = n - 1 - bit
        tests.append((f"diag1 general bit {bit}", lambda q=q: dev.apply_diagonal(d1, [q])))
        tests.append((f"Z (phase half) bit {bit}", lambda q=q: dev.apply_diagonal([1, -1], [q])))
    for b0, b1 in [(0, 1), (3, 12), (12, 20), (n - 1, n - 2)]:
        q0, q1 = n - 1 - b0, n - 1 - b1
        tests.append((f"diag2 general ({b0},{b1})", lambda q0=q0, q1=q1: dev.apply_diagonal(d2, [q0, q1])))
        tests.append((f"CZ ({b0},{b1})", lambda q0=q0, q1=q1: dev.apply_diagonal([1, 1, 1, -1], [q0, q1])))
        tests.append((f"CX c={b0} t={b1}", lambda q0=q0, q1=q1: dev.apply_cx(q0, q1)))
        tests.append((f"CX c={b1} t={b0}", lambda q0=q0, q1=q1: dev.apply_cx(q1, q0)))
        tests.append((f"SWAP ({b0},{b1})", lambda q0=q0, q1=q1: dev.apply_swap(q0, q1)))
    for name, fn in tests:
        ms = timed(dev, fn, args.reps)
        emit(f"{name:28s} {ms:8.3f} ms  {gbytes / (ms * 1e-3):8.0f} GB/s-equivalent")

    emit("\n## reference points: device-to-device copy and scale (read+write every amplitude)")
    other = dev.copy()
    ms = timed(dev, lambda: _lib.call("qsv_copy", other._h, dev._h), args.reps)
    emit(f"hipMemcpy D2D                {ms:8.3f} ms  {gbytes / (ms * 1e-3):8.0f} GB/s")
    ms = timed(dev, lambda: _lib.call("qsv_scale", dev._h, 1.0, 0.0), args.reps)
    emit(f"scale kernel                 {ms:8.3f} ms  {gbytes / (ms * 1e-3):8.0f} GB/s")
    ms = timed(dev, lambda: dev.norm2(), 3)
    emit(f"norm2 (read only)            {ms:8.3f} ms  {gbytes / 2 / (ms * 1e-3):8.0f} GB/s read")
    log.close()


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"sweep took {time.time() - t0:.1f} s")
