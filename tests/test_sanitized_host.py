"""AddressSanitizer + UBSan pass over the HOST side of libqsv (CPU only).

``tests/sanitize/build.py`` compiles every HIP source host-only with ``-fsanitize=address,undefined`` against a
host-memory stand-in for the HIP runtime; ``tests/sanitize/driver.cpp`` then walks the C ABI: every validation branch
(null pointers, out-of-range / duplicate qubits, bad sizes, caller-owned views without room) and the launch
preparation of every kernel family -- enumeration tables, matrix re-indexing for every leg order and bit placement,
controls folded into matrices, permutation lookup tables, block / plane tables of the mode kernels -- on registers
from 0 to 18 qubits and cutoffs 2 to 32.  Kernels do not execute (there is no device code in this build).
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE / "sanitize"))


def test_host_side_is_clean_under_asan_and_ubsan():
    import build as san_build

    if not san_build.CLANG.exists():
        pytest.skip("ROCm clang not installed")
    exe = san_build.build()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    proc = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900, env=env)
    report = proc.stdout[-2000:] + proc.stderr[-6000:]
    assert "ERROR: AddressSanitizer" not in report and "runtime error:" not in report and "LeakSanitizer" not in report, report
    assert proc.returncode == 0, report
    assert "0 failed expectations" in proc.stdout
    launches = int(proc.stdout.split("sanitized host driver: ")[1].split()[0])
    assert launches > 5000          # the driver really went through the launch paths
