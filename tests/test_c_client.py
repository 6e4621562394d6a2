"""The C ABI is usable from plain C: tests/c_abi/ghz.c includes only include/qsv.h and links libqsv.so with gcc."""
from __future__ import annotations

import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
LIBDIR = REPO / "quantum_computations_amd"


def build_client(tmp_path: Path) -> Path:
    exe = tmp_path / "ghz"
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", f"-I{REPO / 'include'}", str(REPO / "tests" / "c_abi" / "ghz.c"),
           f"-L{LIBDIR}", "-lqsv", "-lm", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_client_compiles_links_and_fails_loudly_without_gpu(tmp_path):
    """The header is valid C11 and every symbol the client uses resolves; without a device the client stops at
    qsv_create with the library's error string (exit code 3), it does not compute anything on the CPU."""
    from quantum_computations_amd import _lib
    exe = build_client(tmp_path)
    proc = subprocess.run([str(exe), "8"], capture_output=True, text=True)
    if _lib.device_count() == 0:
        assert proc.returncode == 3, (proc.returncode, proc.stderr)
        assert "no HIP device" in proc.stderr
    else:
        assert proc.returncode == 0, proc.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 12, 22])
def test_c_client_ghz_on_gpu(tmp_path, n):
    exe = build_client(tmp_path)
    proc = subprocess.run([str(exe), str(n)], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    assert f"ghz ok: n={n}" in proc.stdout
