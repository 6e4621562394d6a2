"""Build libqsv.so (HIP, gfx950 only) in-tree with hipcc.  ``python -m quantum_computations_amd.build``."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_ROOT = PKG_DIR.parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libqsv.so"
SOURCES = ["qsv_api.hip", "qsv_kernels.hip", "qsv_qudit.hip", "qsv_gemm.hip", "qsv_decomp.hip", "qsv_circuit.hip"]
HEADERS = [CSRC / "qsv_internal.h", CSRC / "qsv_linalg.h", REPO_ROOT / "include" / "qsv.h"]
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: libqsv.so needs the ROCm toolchain (there is no CPU fallback)")


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source for gfx950 and link libqsv.so next to the package."""
    objs = []
    build_dir = PKG_DIR / "build"
    build_dir.mkdir(exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
             f"-I{REPO_ROOT / 'include'}", f"-I{CSRC}"]
    for name in SOURCES:
        src = CSRC / name
        obj = build_dir / (name + ".o")
        if force or _stale(obj, [src] + HEADERS):
            cmd = [hipcc(), *flags, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        objs.append(obj)
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB_PATH), *map(str, objs), "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    path = build_lib(force="--force" in sys.argv, verbose=True)
    print(path)
