#!/usr/bin/env python3
"""A handful of launches of the 5-qubit kernels (complex / real, register / tile form) and of a 4-qubit and a 1-qubit
gate for contrast, to be run under `rocprofv3 --pmc <SQ counters> --kernel-trace`.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA \
        SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d out -- python3 tools/probe_k5_counters.py
"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd import _lib
from quantum_computations_amd import workloads as W
from quantum_computations_amd.device import DeviceState

n = 28
dev = DeviceState.random(n, 1)
rng = np.random.default_rng(0)
bits5, bits4 = [8, 11, 14, 17, 20], [8, 11, 14, 17]
uc5, ur5 = W.haar_unitary(32, rng), np.linalg.qr(rng.standard_normal((32, 32)))[0]
uc4 = W.haar_unitary(16, rng)
u1 = W.haar_unitary(2, rng)
for variant, u, bits in ((1, uc5, bits5), (3, uc5, bits5), (4, uc5, bits5), (4, ur5, bits5), (4, uc4, bits4), (0, u1, [12])):
    dev.set_option(_lib.OPT_KQ_VARIANT, variant)
    for _ in range(3):
        dev.apply_matrix(u, [n - 1 - b for b in bits])
    dev.sync()
    print(variant, dev.last_kernel(), flush=True)
