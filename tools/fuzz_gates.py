#!/usr/bin/env python3
"""Randomised cross-check of the C ABI against the CPU oracle on one GPU: registers of 3..19 qubits, random sequences of
dense k-qubit gates (k = 1..6; complex, real, diagonal, permutation matrices), controlled gates, multi-controlled
phases, SWAPs, measurements, insertions, qubit permutations and read-out calls -- every kernel family at sizes and
placements the parametrised tests do not enumerate.  Exits non-zero at the first mismatch.

    python tools/fuzz_gates.py [--rounds 200] [--seed 0]
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from oracle import dv_oracle as O  # noqa: E402  (checker)
from quantum_computations_amd import _lib  # noqa: E402
from quantum_computations_amd import workloads as W  # noqa: E402
from quantum_computations_amd.device import DeviceState  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    kernels = {}
    for rnd in range(args.rounds):
        n = int(rng.integers(3, 20))
        ket = W.random_ket(n, int(rng.integers(1 << 30)))
        dev = DeviceState.from_numpy(ket)
        dev.set_option(_lib.OPT_KQ_VARIANT, int(rng.choice([0, 0, 0, 1, 2, 3])))
        dev.set_option(_lib.OPT_READOUT_VARIANT, int(rng.choice([0, 0, 1, 2])))
        dev.set_option(_lib.OPT_COMPLEX_PRODUCT, int(rng.choice([0, 0, 3, 4])))
        dev.set_option(_lib.OPT_SPECIALIZE, int(rng.choice([1, 1, 0])))
        want = ket
        log = []
        for step in range(int(rng.integers(4, 12))):
            n = O.num_qubits(want)
            kind = rng.choice(["dense", "dense", "dense", "ctrl", "mcphase", "swap", "measure", "insert", "permute", "rdm", "pauli"])
            if kind == "dense":
                k = int(rng.integers(1, min(6, n) + 1))
                qs = [int(q) for q in rng.choice(n, size=k, replace=False)]
                flavour = rng.choice(["complex", "real", "diag", "perm"])
                dim = 1 << k
                if flavour == "complex":
                    u = W.haar_unitary(dim, rng)
                elif flavour == "real":
                    u = np.linalg.qr(rng.standard_normal((dim, dim)))[0].astype(complex)
                elif flavour == "diag":
                    u = np.diag(np.exp(1j * rng.uniform(0, 6.28, dim)))
                else:
                    u = np.identity(dim)[rng.permutation(dim)].astype(complex)
                dev.apply_matrix(u, qs)
                want = O.apply_gate(want, u, qs)
                log.append((kind, flavour, qs))
            elif kind == "ctrl" and n >= 2:
                nc = int(rng.integers(1, min(n, 6)))
                qs = [int(q) for q in rng.choice(n, size=nc + 1, replace=False)]
                u = W.haar_unitary(2, rng) if rng.random() < 0.7 else np.diag(np.exp(1j * rng.uniform(0, 6.28, 2)))
                dev.apply_controlled(u, qs[:-1], qs[-1])
                full = np.identity(1 << (nc + 1), dtype=complex)
                full[-2:, -2:] = u
                want = O.apply_gate(want, full, qs)
                log.append((kind, qs))
            elif kind == "mcphase":
                k = int(rng.integers(1, n + 1))
                qs = [int(q) for q in rng.choice(n, size=k, replace=False)]
                phase = np.exp(1j * rng.uniform(0, 6.28))
                dev.apply_mcphase(qs, phase)
                idx = np.arange(1 << n)
                mask = np.ones(1 << n, dtype=bool)
                for q in qs:
                    mask &= ((idx >> (n - 1 - q)) & 1).astype(bool)
                want = np.where(mask, want * phase, want)
                log.append((kind, qs))
            elif kind == "swap" and n >= 2:
                a, b = (int(q) for q in rng.choice(n, size=2, replace=False))
                dev.apply_swap(a, b)
                want = O.apply_gate(want, np.identity(4)[[0, 2, 1, 3]].astype(complex), [a, b])
                log.append((kind, a, b))
            elif kind == "measure" and n >= 4:
                q = int(rng.integers(n))
                theta, phi, result = rng.uniform(0, 3.1), rng.uniform(0, 6.2), int(rng.integers(2))
                from quantum_computations_amd.dv_simulator import gates as G
                out, s = G.M(q, theta, phi, result=result).apply(dev)
                want, _ = O.measure(want, q, theta, phi, result)
                log.append((kind, q, result))
            elif kind == "insert" and n <= 18:
                q = int(rng.integers(n + 1))
                amp = rng.standard_normal(2) + 1j * rng.standard_normal(2)
                dev.insert(q, amp)
                want = O.insert_qubit(want, q, amp)
                log.append((kind, q))
            elif kind == "permute":
                order = [int(v) for v in rng.permutation(n)]
                dev.permute(order)
                want = O.permute_qubits(want, order)
                log.append((kind, order))
            elif kind == "rdm":
                k = int(rng.integers(1, min(6, n) + 1))
                kept = [int(q) for q in rng.choice(n, size=k, replace=False)]
                got = dev.reduced_density(kept)
                ref = O.reduced_density(want, kept)
                if np.max(np.abs(got - ref)) > 1e-12 * max(1.0, np.abs(ref).max()):
                    raise SystemExit(f"round {rnd}: reduced density {kept} on {n} qubits differs by {np.max(np.abs(got - ref)):.3e}\n{log}")
                log.append((kind, kept))
            elif kind == "pauli":
                k = int(rng.integers(1, min(5, n) + 1))
                qs = [int(q) for q in rng.choice(n, size=k, replace=False)]
                letters = "".join(rng.choice(list("IXYZ"), size=k))
                mats = {"I": np.identity(2), "X": np.array([[0, 1], [1, 0]]), "Y": np.array([[0, -1j], [1j, 0]]), "Z": np.diag([1.0, -1.0])}
                tmp = want
                for p, q in zip(letters, qs):
                    tmp = O.apply_gate(tmp, mats[p].astype(complex), [q])
                got, ref = dev.expect_pauli(letters, qs), np.vdot(want, tmp)
                if abs(got - ref) > 1e-11 * max(1.0, abs(ref)):
                    raise SystemExit(f"round {rnd}: <{letters}> on {qs} differs: {got} vs {ref}\n{log}")
                log.append((kind, letters, qs))
            else:
                continue
            name = dev.last_kernel()
            kernels[name] = kernels.get(name, 0) + 1
            scale = max(1.0, float(np.max(np.abs(want))))
            err = float(np.max(np.abs(dev.to_numpy() - want)))
            if not err < 1e-11 * scale:
                raise SystemExit(f"round {rnd}: after {log[-1]} on {n} qubits ({name}): max abs err {err:.3e}\n{log}")
        dev.close()
    print(f"fuzz ok: {args.rounds} rounds; kernels exercised:")
    for name in sorted(kernels):
        print(f"  {kernels[name]:5d}  {name}")


if __name__ == "__main__":
    main()
