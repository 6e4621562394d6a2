"""A short run of the randomised cross-check (tools/fuzz_gates.py) as part of the GPU suite: registers of 3..19 qubits,
random sequences over every operation of the C ABI's qubit path and every kernel form, against the oracle."""
from __future__ import annotations

import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_random_operation_sequences_match_the_oracle(seed):
    proc = subprocess.run([sys.executable, str(REPO / "tools" / "fuzz_gates.py"), "--rounds", "80", "--seed", str(seed)],
                          capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    assert "fuzz ok: 80 rounds" in proc.stdout
    assert "k_dense_lds<5" in proc.stdout and "k_rdm_tile<" in proc.stdout and "k_permute_s" in proc.stdout


@pytest.mark.gpu
def test_random_mode_operation_sequences_match_the_oracle():
    proc = subprocess.run([sys.executable, str(REPO / "tools" / "fuzz_modes.py"), "--rounds", "80", "--seed", "21"],
                          capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    assert "fuzz ok: 80 rounds" in proc.stdout and "k_mode2_blocks<" in proc.stdout
