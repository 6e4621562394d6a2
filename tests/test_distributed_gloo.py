"""The N > 1 path on CPU: world_size 2 and 4 with the ``gloo`` backend (tests/dist_worker.py).

What is under test is ``quantum_computations_amd.distributed.ShardedState`` -- the logical->physical qubit map,
the half-shard exchange over ``torch.distributed`` and the no-traffic handling of diagonal legs and controls.
The per-shard gate arithmetic is supplied by a test double (tests/oracle_engine.py); on the GPU box the same
worker runs with HIP kernels (tests/test_gpu_parity.py::test_sharded_state_on_one_gpu).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_workers(world: int, *extra: str, timeout: int = 600) -> str:
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(HERE / "dist_worker.py"), *extra]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert proc.returncode == 0, proc.stdout[-3000:] + "\n" + proc.stderr[-3000:]
    return proc.stdout


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_register_matches_oracle(world):
    """world 2: pairwise half-shard swaps; world 4 and 8: all rank bits in one all-to-all.  The staging piece is cut
    to 8 amplitudes so that every exchange runs the multi-slice, double-buffered loop the 1 GiB default only reaches
    on registers of 32+ qubits."""
    out = run_workers(world, "--backend", "gloo", "--qubits", "9" if world < 8 else "10", "--chunk-amps", "8")
    assert f"dist_worker ok: world={world}" in out
    assert "chunk_amps=8" in out


def test_default_chunk_single_slice():
    out = run_workers(2, "--backend", "gloo", "--qubits", "8")
    assert "dist_worker ok: world=2" in out and f"chunk_amps={1 << 26}" in out


@pytest.mark.parametrize("world,qubits,chunk", [(2, 9, 64), (4, 10, 64), (8, 12, 64)])
def test_gates_ride_inside_exchange_steps(world, qubits, chunk):
    """Round 3: local gates whose mixing legs lie inside the slices of an exchange are applied slice by slice while the
    other slices travel.  Parity 1e-12 against the oracle, the same exchange schedule as without the overlap, and a
    counter showing that gates were executed inside exchanges."""
    out = run_workers(world, "--backend", "gloo", "--qubits", str(qubits), "--chunk-amps", "8",
                      "--overlap-chunk-amps", str(chunk))
    assert f"dist_worker ok: world={world}" in out
    line = [l for l in out.splitlines() if l.startswith("overlap ok")][0]
    assert int(line.split("gates_in_exchanges=")[1].split()[0]) > 0
