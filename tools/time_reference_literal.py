#!/usr/bin/env python3
"""BUILD CONTAINER ONLY: time the literal reference ``Gate.apply`` (dense 2^N x 2^N operator + mat-vec) at the sizes it
can run, for BASELINE.md section 2 / 4(1).  Imports /root/reference by path; nothing of it is copied.

    OMP_NUM_THREADS=8 python tools/time_reference_literal.py
"""
from __future__ import annotations

import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, "/root/reference")
from simulators.dv_simulator import gates as ref  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    print(f"cores={os.cpu_count()} OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')} numpy={np.__version__}")
    for n in (8, 10, 11, 12):
        ket = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
        ket /= np.linalg.norm(ket)
        row = []
        for gate in (ref.H(n // 2), ref.T(n // 2), ref.CX(1, n - 2)):
            gate.apply(ket)                                   # warm-up
            times = []
            for _ in range(5 if n < 12 else 3):
                t0 = time.perf_counter()
                gate.apply(ket)
                times.append(time.perf_counter() - t0)
            row.append(statistics.median(times))
        print(f"n={n}: H {row[0] * 1e3:8.1f} ms  T {row[1] * 1e3:8.1f} ms  CX {row[2] * 1e3:8.1f} ms  "
              f"-> {1 / max(row):.2f}..{1 / min(row):.2f} gate-apps/s")


if __name__ == "__main__":
    main()
