#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: build the AddressSanitizer + UBSan host-side binary of libqsv.

    python tests/sanitize/build.py            # -> tests/sanitize/_build/qsv_host_san

Every HIP source of the library is compiled with ``clang++ -x hip --offload-host-only`` (host code only: the kernels'
bodies are not compiled at all) and ``-fsanitize=address,undefined -fno-sanitize-recover=all``, then linked with
``hip_stub.cpp`` (a host-memory stand-in for the HIP runtime) and ``driver.cpp``.  No GPU and no libamdhip64 involved;
GPU sanitizers are not available on this pool.
"""
from __future__ import annotations

import re
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
CSRC = REPO / "quantum_computations_amd" / "csrc"
OUT = HERE / "_build"
CLANG = Path("/opt/rocm/lib/llvm/bin/clang++")
SOURCES = ["qsv_api.hip", "qsv_kernels.hip", "qsv_qudit.hip", "qsv_gemm.hip", "qsv_decomp.hip", "qsv_circuit.hip"]
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
INC = [f"-I{REPO / 'include'}", f"-I{CSRC}"]


def stale(target: Path, deps: list[Path]) -> bool:
    return not target.exists() or any(d.stat().st_mtime > target.stat().st_mtime for d in deps)


def build(verbose: bool = False) -> Path:
    if not CLANG.exists():
        raise RuntimeError(f"{CLANG} not found: the sanitized host build needs the ROCm clang")
    OUT.mkdir(exist_ok=True)
    headers = [CSRC / "qsv_internal.h", CSRC / "qsv_linalg.h", REPO / "include" / "qsv.h"]
    objs = []

    def run(cmd):
        if verbose:
            print(" ".join(map(str, cmd)), flush=True)
        subprocess.run(cmd, check=True)

    for name in SOURCES:
        obj = OUT / (name + ".o")
        if stale(obj, [CSRC / name] + headers):
            run([CLANG, "-x", "hip", "--offload-host-only", "--rocm-path=/opt/rocm", "-nogpulib", "-std=c++17", *SAN, *INC,
                 "-c", CSRC / name, "-o", obj])
        objs.append(obj)
    for name in ("hip_stub.cpp", "driver.cpp"):
        obj = OUT / (name + ".o")
        if stale(obj, [HERE / name] + headers):
            run([CLANG, "-std=c++17", *SAN, *INC, "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-c", HERE / name, "-o", obj])
        objs.append(obj)
    # the host stubs reference the embedded device image of each translation unit; there is none in a host-only build
    fat = OUT / "fatbins.c"
    names = set()
    for obj in objs:
        names |= set(re.findall(r"U (__hip_fatbin_\w+)", subprocess.run(["nm", str(obj)], capture_output=True, text=True).stdout))
    fat.write_text("".join(f"const char {n}[8] = {{0}};\n" for n in sorted(names)))
    exe = OUT / "qsv_host_san"
    if stale(exe, objs + [fat]):
        run([CLANG, *SAN, "-x", "c", fat, "-x", "none", *objs, "-ldl", "-lpthread", "-o", exe])
    return exe


if __name__ == "__main__":
    print(build(verbose=True))
