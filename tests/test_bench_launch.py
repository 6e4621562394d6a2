"""``python bench.py --gpus N`` started as a plain process must bring up its own N ranks (the driver's scaling run).

No GPU here: ``--rehearse`` makes the ranks rendezvous over gloo and compute only the data-free exchange schedule of
the step, which exercises exactly the part that cannot be tested on one GPU -- self-launch before any GPU call,
argument plumbing through ``torch.distributed.run``, rendezvous on 127.0.0.1, one JSON line from rank 0, return code.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def run_bench(*flags: str, timeout: int = 300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    proc = subprocess.run([sys.executable, str(REPO / "bench.py"), *flags], env=env, capture_output=True, text=True,
                          timeout=timeout)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    return proc, [json.loads(ln) for ln in lines]


@pytest.mark.parametrize("flags,n_qubits", [
    (("--gpus", "2"), 29),
    (("--gpus", "4", "--scaling", "strong"), 28),
    (("--gpus", "4", "--config", "cfg3"), 33),
])
def test_plain_python_launch_starts_its_own_ranks(flags, n_qubits):
    proc, lines = run_bench(*flags, "--rehearse", "--steps", "1", "--warmup", "0")
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    assert len(lines) == 1                                   # rank 0 only
    line = lines[0]
    assert line["n_gpus"] == int(flags[1]) and line["config"]["n_qubits"] == n_qubits
    assert line["ranks_agree_on_schedule"] is True
    assert line["exchange_steps_per_circuit"] >= 1           # these circuits all touch rank bits


def test_child_failure_is_the_parents_return_code():
    # without --rehearse the ranks need a GPU: on the CPU box every child stops at the availability check, and the
    # parent must hand that failure on instead of printing a line
    proc, lines = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would really run")
    assert proc.returncode != 0 and not lines
    assert "needs an MI355X" in proc.stderr
