#!/usr/bin/env python3
"""k-qubit dense gates (k = 3..5, fused blocks / d = 8..32 mode gates) on one MI355X: GB/s by target-bit set, for the
three forms of the kernel (QSV_OPT_KQ_VARIANT) and for complex and real matrices.

    python tools/sweep_kq.py [--n 28] [--reps 5] [--out gpurun_out/sweep_kq.txt]
"""
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
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--k5", action="store_true")
    ap.add_argument("--out", default="gpurun_out/sweep_kq.txt")
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
    emit("# variant 3 = line-granular (k_dense_lds), 1 = wave shuffles (k_dense_big<K,KL>), 2 = no exchange, 0 = shipped per-case choice")
    for k in ((5,) if args.k5 else (5, 4, 3)):
        uc = W.haar_unitary(1 << k, rng)
        ur = np.linalg.qr(rng.standard_normal((1 << k, 1 << k)))[0]
        sets = {"high": [8 + 3 * j for j in range(k)], "top": [n - 1 - j for j in range(k)],
                "far": [20 + j for j in range(k)],
                "1 low (b5)": [5] + [7 + 4 * j for j in range(k - 1)],
                "1 low (b0)": [0] + [7 + 4 * j for j in range(k - 1)],
                "2 low (b0,b3)": [0, 3] + [7 + 4 * j for j in range(k - 2)],
                "2 low (b4,b5)": [4, 5] + [7 + 4 * j for j in range(k - 2)],
                "2 low (b1,b2)": [1, 2] + [7 + 4 * j for j in range(k - 2)],
                "3 low (b0,b1,b2)": [0, 1, 2] + [7 + 4 * j for j in range(k - 3)],
                "3 low (b3,b4,b5)": [3, 4, 5] + [7 + 4 * j for j in range(k - 3)],
                "3 low (b0,b2,b4)": [0, 2, 4] + [7 + 4 * j for j in range(k - 3)]}
        if k >= 4:
            sets["4 low (b0..b3)"] = [0, 1, 2, 3] + [9] * (k - 4)
            sets["4 low (b2..b5)"] = [2, 3, 4, 5] + [9] * (k - 4)
        if k >= 5:
            sets["5 low (b0..b4) = last d=32 mode"] = [0, 1, 2, 3, 4]
            sets["5 low (b1..b5)"] = [1, 2, 3, 4, 5]
            sets["mode 4 of 6 (b5..b9)"] = [5, 6, 7, 8, 9]
        emit(f"\n## k = {k}: ms / GB/s per target-bit set; columns: variant 3 complex | variant 3 real | variant 1 | variant 2 | variant 4 (tile) complex | variant 4 real | variant 5 (MFMA, k = 5) complex | shipped choice complex | shipped choice real")
        for label, bits in sets.items():
            qs = [n - 1 - b for b in bits]
            cells = []
            name0 = ""
            for variant, u in ((3, uc), (3, ur), (1, uc), (2, uc), (4, uc), (4, ur), (5, uc), (0, uc), (0, ur)):
                dev.set_option(_lib.OPT_KQ_VARIANT, variant)
                ms = timed(dev, lambda: dev.apply_matrix(u, qs), args.reps)
                cells.append(f"{ms:7.3f} ms {gbytes / (ms * 1e-3):6.0f}")
                if variant == 0 and u is uc:
                    name0 = dev.last_kernel()
            emit(f"k={k} {label:34s} {str(bits):22s} " + " | ".join(cells) + f"   {name0}")
    dev.set_option(_lib.OPT_KQ_VARIANT, 0)
    log.close()


if __name__ == "__main__":
    main()
