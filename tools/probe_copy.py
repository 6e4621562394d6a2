#!/usr/bin/env python3
"""What a plain copy reaches on this box: the library's copy kernel (qsv_copy; form and tile order from QSV_COPY_MODE /
QSV_COPY_REGIONS, read once per process) from one register into another of the same size.

    QSV_COPY_MODE=2 QSV_COPY_REGIONS=8 python tools/probe_copy.py [n]
"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from quantum_computations_amd.device import DeviceState

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
a = DeviceState.random(n, 1)
b = a.copy()
a.copy_into(b); b.sync()
best = 1e9
for _ in range(3):
    b.timer_start()
    for _ in range(10):
        a.copy_into(b)
    best = min(best, b.timer_stop() / 10)
gb = 2 * 16 * (1 << n) / 1e9
print(f"n={n} mode={os.environ.get('QSV_COPY_MODE', 'default')} regions={os.environ.get('QSV_COPY_REGIONS', '0')}: "
      f"{best:.4f} ms per copy = {gb / best:.3f} TB/s (read + write)")
