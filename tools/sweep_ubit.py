#!/usr/bin/env python3
"""Dense 1-qubit kernel: GB/s by target bit for (unroll U, item-stride bit) combinations.  Tuning aid."""
from __future__ import annotations

import argparse
import sys
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
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--out", default="gpurun_out/sweep_ubit.txt")
    args = ap.parse_args()
    n = args.n
    dev = DeviceState.random(n, seed=1)
    gb = 2 * 16 * (1 << n) / 1e9
    u2 = W.haar_unitary(2, np.random.default_rng(0))
    combos = [(1, 8)] + [(u, b) for u in (2, 4) for b in (8, 10, 11, 12, 13, 15)]
    lines = ["bit  " + "  ".join(f"U{u}/s{b:<2d}" for u, b in combos)]
    table = np.zeros((n, len(combos)))
    for j, (u, b) in enumerate(combos):
        dev.set_option(_lib.OPT_UNROLL, u)
        dev.set_option(_lib.OPT_ITEM_STRIDE_BIT, b)
        for bit in range(n):
            ms = timed(dev, lambda: dev.apply_matrix(u2, [n - 1 - bit]), args.reps)
            table[bit, j] = gb / (ms * 1e-3)
    for bit in range(n):
        lines.append(f"{bit:3d}  " + "  ".join(f"{v:6.0f}" for v in table[bit]))
    lines.append("mean " + "  ".join(f"{v:6.0f}" for v in table.mean(axis=0)))
    lines.append("best per bit: " + " ".join(f"{bit}:U{combos[int(np.argmax(table[bit]))][0]}/s"
                                             f"{combos[int(np.argmax(table[bit]))][1]}={table[bit].max():.0f}"
                                             for bit in range(n)))
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
