#!/usr/bin/env python3
"""Probe: device SVD (torch -> hipSOLVER/rocSOLVER) against host LAPACK on two-site MPS shapes."""
import time
import numpy as np
import torch

for n in (500, 1000, 2000, 4000):
    a = torch.randn(n, n, dtype=torch.complex128, device="cuda")
    torch.linalg.svd(a[:64, :64], full_matrices=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    u, s, vh = torch.linalg.svd(a, full_matrices=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    err = float((u * s.to(u.dtype) @ vh - a).abs().max())
    line = f"n={n}: device svd {dt*1e3:9.1f} ms (recon err {err:.1e})"
    if n <= 2000:
        h = a.cpu().numpy()
        t0 = time.perf_counter()
        np.linalg.svd(h, full_matrices=False)
        line += f" | host numpy {1e3*(time.perf_counter()-t0):9.1f} ms"
    print(line, flush=True)
