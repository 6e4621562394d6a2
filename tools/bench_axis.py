#!/usr/bin/env python3
"""Time qsv_tensor_apply_axis on MPS-site shapes of the reference's CV workloads (d = 1000 grid, bond dims <= 100)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from quantum_computations_amd.device import tensor_apply_axis

rng = np.random.default_rng(0)
for L, d, R in [(1, 1000, 1), (10, 1000, 10), (50, 1000, 50), (100, 1000, 100), (1, 1000, 1000), (100, 400, 100)]:
    site = torch.randn(L, d, R, dtype=torch.complex128, device="cuda")
    out = torch.empty_like(site)
    m = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    stream = torch.cuda.current_stream().cuda_stream
    mt = torch.from_numpy(m).cuda()
    tensor_apply_axis(site.data_ptr(), out.data_ptr(), L, d, d, R, mt.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        tensor_apply_axis(site.data_ptr(), out.data_ptr(), L, d, d, R, mt.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    flops = 8.0 * L * d * d * R
    # torch reference (rocBLAS zgemm through tensordot) for comparison
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ref = torch.einsum("ij,ljr->lir", mt, site)
    torch.cuda.synchronize()
    dt_ref = (time.perf_counter() - t0) / reps
    err = float((ref - out).abs().max())
    import os
    tag = "own kernels" if os.environ.get("QSV_NO_ROCBLAS") == "1" else "rocBLAS route"
    print(f"[{tag}] (L={L}, d={d}, R={R}): qsv {dt*1e3:9.2f} ms = {flops/dt/1e12:6.2f} TFLOP/s | torch/rocBLAS {dt_ref*1e3:9.2f} ms = {flops/dt_ref/1e12:6.2f} TFLOP/s | max diff {err:.1e}")
