#!/usr/bin/env python3
"""Check and time the MFMA tall-skinny products (qsv_tensor_skinny_gemm) against rocBLAS (through torch)."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib

dev = torch.device("cuda", 0)


def run(conj, A_cm, Q_cm, n, m, l):
    """A_cm / Q_cm: torch tensors holding COLUMN-major data, i.e. row-major transposes."""
    out_rows = m if conj else n
    Y = torch.empty(l, out_rows, dtype=torch.complex128, device=dev)     # column-major (out_rows x l)
    _lib.call("qsv_tensor_skinny_gemm", 0, C.c_void_p(torch.cuda.current_stream().cuda_stream), int(conj), n, m, l,
              C.c_void_p(A_cm.data_ptr()), C.c_void_p(Q_cm.data_ptr()), C.c_void_p(Y.data_ptr()))
    return Y


# correctness on awkward shapes
for n, m, l in [(100, 70, 5), (257, 129, 16), (1000, 333, 26), (4097, 1023, 42), (130, 4099, 64), (64, 32, 1)]:
    At = torch.randn(m, n, dtype=torch.complex128, device=dev)           # = A^T  (A is n x m, column-major)
    A = At.T
    for conj in (0, 1):
        rows_q = n if conj else m
        Qt = torch.randn(l, rows_q, dtype=torch.complex128, device=dev)  # = Q^T
        want = (A.conj().T if conj else A) @ Qt.T
        got = run(conj, At, Qt, n, m, l).T
        torch.cuda.synchronize()
        err = float((got - want).abs().max() / want.abs().max())
        print(f"n={n} m={m} l={l} conj={conj}: rel err {err:.1e}")
        assert err < 1e-12

# speed at the range-finder shapes
for n, m in [(16000, 16000), (32000, 32000), (8000, 8000)]:
    At = torch.randn(m, n, dtype=torch.complex128, device=dev)
    for l in (12, 26, 42):
        for conj in (0, 1):
            rows_q = n if conj else m
            Qt = torch.randn(l, rows_q, dtype=torch.complex128, device=dev)
            run(conj, At, Qt, n, m, l)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                run(conj, At, Qt, n, m, l)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            A = At.T
            ref = lambda: (A.conj().T if conj else A) @ Qt.T
            ref()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                ref()
            torch.cuda.synchronize()
            dt_ref = (time.perf_counter() - t0) / 5
            print(f"n=m={n} l={l} {'A^H Q' if conj else 'A Q  '}: mfma kernel {dt*1e3:6.2f} ms ({n*m*16/dt/1e12:4.2f} TB/s, "
                  f"{8*n*m*l/dt/1e12:5.1f} TFLOP/s) | rocBLAS {dt_ref*1e3:6.2f} ms", flush=True)
    del At
