#!/usr/bin/env python3
"""Minimal driver for PMC passes over the MFMA tall-skinny products: a few launches at n = m = 16000, l = 26."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib

n = m = 16000
l = 26
At = torch.randn(m, n, dtype=torch.complex128, device="cuda")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for op, rows_q, rows_y in ((0, m, n), (1, n, m)):
    Qt = torch.randn(l, rows_q, dtype=torch.complex128, device="cuda")
    Y = torch.empty(l, rows_y, dtype=torch.complex128, device="cuda")
    for _ in range(3):
        _lib.call("qsv_tensor_skinny_gemm", 0, stream, op, n, m, l, C.c_void_p(At.data_ptr()), C.c_void_p(Qt.data_ptr()),
                  C.c_void_p(Y.data_ptr()))
torch.cuda.synchronize()
print("done")
