"""TEST DOUBLE for the local engine of ``ShardedState``: the same methods as ``DeviceState``, computed by the CPU
oracle on a CPU torch tensor.  It exists so that the sharding / exchange logic (which is plain Python over
``torch.distributed``) can be exercised with the ``gloo`` backend on a machine without a GPU.  Never shipped:
the product's engine is ``DeviceState`` (HIP kernels through the C ABI).
"""
from __future__ import annotations

import numpy as np

from oracle import dv_oracle as O


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def counter_normal(seed: int, start: int, count: int) -> np.ndarray:
    """NumPy restatement of k_fill_random (qsv_kernels.hip): complex normals keyed by (seed, global index)."""
    with np.errstate(over="ignore"):
        key = splitmix64(np.array([seed], dtype=np.uint64))[0]
        g = np.arange(start, start + count, dtype=np.uint64)
        r1 = splitmix64(key ^ (np.uint64(2) * g))
        r2 = splitmix64(key ^ (np.uint64(2) * g + np.uint64(1)))
    u1 = ((r1 >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53
    u2 = ((r2 >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53
    rad = np.sqrt(-2.0 * np.log(u1))
    return rad * np.cos(2 * np.pi * u2) + 1j * rad * np.sin(2 * np.pi * u2)


class OracleEngine:
    def __init__(self, buf, n_local: int):
        self.buf = buf                       # CPU torch tensor (complex128); .numpy() shares its memory
        self.n = n_local
        self.calls = []

    @property
    def arr(self) -> np.ndarray:
        return self.buf.numpy()[: 1 << self.n]

    @property
    def num_qubits(self) -> int:
        return self.n

    def _store(self, values: np.ndarray) -> None:
        self.arr[:] = values

    def sync(self) -> None:
        pass

    def set_basis(self, index: int) -> None:
        self.arr[:] = 0
        self.arr[index] = 1

    def fill_random(self, seed: int, index_offset: int = 0, normalise: bool = True) -> float:
        vals = counter_normal(seed, index_offset, 1 << self.n)
        n2 = float(np.sum(np.abs(vals) ** 2))
        self._store(vals / np.sqrt(n2) if normalise else vals)
        return n2

    def apply_scale(self, factor: complex):
        self.arr[:] *= complex(factor)
        return self

    def apply_matrix(self, matrix, indices):
        self.calls.append(("matrix", tuple(indices)))
        self._store(O.apply_gate(self.arr.copy(), np.asarray(matrix), list(indices)))
        return self

    def apply_swap(self, q0: int, q1: int):
        swap = np.identity(4)[[0, 2, 1, 3]]
        return self.apply_matrix(swap, [q0, q1])

    def apply_controlled(self, matrix, controls, target: int):
        k = len(controls) + 1
        full = np.identity(1 << k, dtype=complex)
        full[-2:, -2:] = np.asarray(matrix)
        return self.apply_matrix(full, list(controls) + [target])

    def apply_mcphase(self, qubits, phase: complex):
        d = np.ones(1 << len(qubits), dtype=complex)
        d[-1] = phase
        return self.apply_matrix(np.diag(d), list(qubits))

    def measure_probs(self, index: int, eig0, eig1):
        psi = np.moveaxis(self.arr.reshape((2,) * self.n), index, 0).reshape(2, -1)
        r0 = eig0[0] * psi[0] + eig0[1] * psi[1]
        r1 = eig1[0] * psi[0] + eig1[1] * psi[1]
        return float(np.vdot(r0, r0).real), float(np.vdot(r1, r1).real)

    def collapse(self, index: int, eig, scale: float):
        psi = np.moveaxis(self.arr.reshape((2,) * self.n), index, 0).reshape(2, -1)
        out = (eig[0] * psi[0] + eig[1] * psi[1]) * scale
        self.n -= 1
        self._store(out)
        return self

    def insert(self, index: int, amplitudes):
        grown = O.insert_qubit(self.arr.copy(), index, np.asarray(amplitudes))
        self.n += 1
        self._store(grown)
        return self

    def norm2(self) -> float:
        return float(np.sum(np.abs(self.arr) ** 2))

    def reduced_density(self, qubits) -> np.ndarray:
        return O.reduced_density(self.arr.copy(), list(qubits))

    def expect_pauli(self, paulis: str, qubits) -> complex:
        letters = {"I": np.identity(2), "X": np.array([[0, 1], [1, 0]]), "Y": np.array([[0, -1j], [1j, 0]]),
                   "Z": np.diag([1.0, -1.0])}
        out = self.arr.copy()
        for p, q in zip(paulis, qubits):
            out = O.apply_gate(out, letters[p].astype(complex), [q])
        return complex(np.vdot(self.arr, out))

    def probabilities(self, indices) -> np.ndarray:
        return np.abs(self.arr[np.asarray(indices, dtype=np.int64)]) ** 2

    def event_record(self, slot: int) -> None:
        pass

    def event_elapsed_ms(self, a: int, b: int) -> float:
        return 0.0

    def last_kernel(self) -> str:
        return "oracle-engine"


def factory(buf, n_local: int) -> OracleEngine:
    return OracleEngine(buf, n_local)
