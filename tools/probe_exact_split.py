"""Exact split (tensor_svd without a cap, rel_err = 1e-12) of matrices that are NOT numerically low-rank: the seeded Jacobi
route against the library SVD (QSV_SVD=library), by size.

    python3 tools/probe_exact_split.py            # runs itself once per route in child processes
"""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
SHAPES = [(60, 120), (120, 120), (400, 400), (700, 520), (1200, 1200), (1000, 2000), (2000, 2000)]


def child() -> None:
    from quantum_computations_amd.cv_simulator.mps import tensor_svd

    out = {}
    for rows, cols in SHAPES:
        rng = np.random.default_rng(rows + cols)
        full = min(rows, cols)
        u, _ = np.linalg.qr(rng.standard_normal((rows, full)) + 1j * rng.standard_normal((rows, full)))
        v, _ = np.linalg.qr(rng.standard_normal((cols, full)) + 1j * rng.standard_normal((cols, full)))
        spectrum = np.exp(-12.0 * np.arange(full) / full)          # five decades over the whole width: full rank at 1e-12
        a = (u * spectrum) @ v.conj().T
        tensor_svd(a.reshape(rows, 1, 1, cols), [0, 1], [2, 3])
        t0 = time.perf_counter()
        m1, m2 = tensor_svd(a.reshape(rows, 1, 1, cols), [0, 1], [2, 3])
        seconds = time.perf_counter() - t0
        err = float(np.max(np.abs(np.tensordot(m1, m2, axes=1).reshape(rows, cols) - a)))
        out[f"{rows}x{cols}"] = {"ms": round(1e3 * seconds, 2), "rank": int(m1.shape[-1]), "max_abs_err": err}
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for route in ("jacobi", "library"):
            env = dict(os.environ)
            if route == "library":
                env["QSV_SVD"] = "library"
            done = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            line = [l for l in done.stdout.splitlines() if l.startswith("{")]
            print(route, line[-1] if line else done.stderr[-2000:], flush=True)
