"""Gate fusion for ``Simulator.run``: merge neighbouring gates into dense blocks of at most ``max_qubits`` qubits.

The gate path is bandwidth-bound -- one pass over the register per kernel, and a dense k-qubit block (k <= 4..5) costs
about the same pass as a single 1-qubit gate (``k_dense_big``: 5.2-5.9 TB/s for k = 3..4 on an MI355X) -- so applying
m gates as one block divides their cost by m.  This sits exactly where the reference loops over the circuit
(``simulators/dv_simulator/simulator.py:40-52``); SURVEY.md 8f ranks it first among the "next" rows.

Scheduler: a list of open blocks with pairwise disjoint qubit sets.  A new gate joins the blocks it touches if the
union stays within ``max_qubits`` (the touched blocks are merged), otherwise those blocks are flushed to the output
and the gate opens a new block.  Disjoint blocks commute, so the order in which they are flushed is free; gates that
share a qubit always keep their order.  Measurements, insertions and classically controlled gates are barriers (they
change the register size or depend on run-time results): everything is flushed before them.  A block that ends up
holding a single gate is emitted as that gate, so the reduced-traffic kernels (CZ, CX, SWAP, diagonals) still apply.
"""
from __future__ import annotations

import numpy as np

from .dv_simulator.gates import Gate, Insert, M


def _apply_to_rows(block: np.ndarray, k: int, matrix: np.ndarray, legs: list[int]) -> np.ndarray:
    """``G_embedded @ block`` where ``matrix`` acts on the row legs ``legs`` of the 2^k x 2^k ``block``."""
    g = np.asarray(matrix, dtype=np.complex128).reshape((2,) * (2 * len(legs)))
    t = block.reshape((2,) * k + (-1,))
    t = np.tensordot(g, t, axes=(list(range(len(legs), 2 * len(legs))), legs))
    t = np.moveaxis(t, list(range(len(legs))), legs)
    return np.ascontiguousarray(t).reshape(1 << k, 1 << k)


class _Block:
    def __init__(self, gate: Gate):
        self.qubits: list[int] = list(gate.indices)
        self.matrix = np.asarray(gate.matrix, dtype=np.complex128)
        self.gates = [gate]

    def absorb(self, gate: Gate) -> None:
        self.matrix = _apply_to_rows(self.matrix, len(self.qubits), gate.matrix,
                                     [self.qubits.index(q) for q in gate.indices])
        self.gates.append(gate)

    def grow(self, qubits: list[int]) -> None:
        """Extend to more qubits (identity on the new ones, appended as less significant legs)."""
        extra = [q for q in qubits if q not in self.qubits]
        if extra:
            self.matrix = np.kron(self.matrix, np.identity(1 << len(extra)))
            self.qubits += extra

    def merge(self, other: "_Block") -> None:
        """Tensor product with a block on disjoint qubits."""
        self.matrix = np.kron(self.matrix, other.matrix)
        self.qubits += other.qubits
        self.gates += other.gates

    def emit(self):
        if len(self.gates) == 1:
            return self.gates[0]
        fused = Gate(list(self.qubits), self.matrix)
        fused.sources = list(self.gates)      # the gates this block replaces (Simulator uses them for dtype parity)
        return fused


MAX_LEGS = 6      # qsv_apply_kq takes matrices on at most six qubits (include/qsv.h)


def _conserves(gate: Gate, qubit: int) -> bool:
    """True if ``gate`` never flips ``qubit`` (a control or a diagonal leg): on a sharded register such a leg costs
    nothing when the qubit is a rank bit -- each rank applies the sub-block its own bit selects."""
    from .distributed import _leg_is_block_diagonal

    if qubit not in gate.indices:
        return True
    return _leg_is_block_diagonal(np.asarray(gate.matrix, dtype=np.complex128), len(gate.indices),
                                  list(gate.indices).index(qubit))


def fuse_circuit(circuit: list, max_qubits: int = 4, n_qubits: int | None = None, remote=()) -> list:
    """Return an equivalent circuit in which runs of plain gates are merged into blocks of <= ``max_qubits`` qubits.

    ``remote``: qubits that sit on rank bits of a sharded register (``ShardedState.remote_qubits()``).  A remote
    qubit that every gate of a block conserves does not count towards the block size (the shard-local matrix does
    not contain it), and a gate that *mixes* a remote qubit is never merged into a block that only conserved it so
    far -- the block can run without an exchange, the gate cannot, and merging would make the whole block wait for
    the qubit to be brought in.

    ``n_qubits`` is accepted for compatibility and unused: round 1 capped 5-qubit unions that contained more than one
    of the six least significant qubits (the shuffle form of ``k_dense_big<5>`` ran them at 2.1-3.8 TB/s); the
    line-granular kernel ``k_dense_lds`` runs 5-qubit blocks at 4.9-5.2 TB/s wherever their qubits sit, so the only
    limit left is the block size."""
    if max_qubits < 2:
        return list(circuit)
    out: list = []
    open_blocks: list[_Block] = []

    remote = frozenset(remote)

    def allowed(union, block_gates, gate) -> bool:
        if len(union) > MAX_LEGS:
            return False
        seen = set().union(*(g.indices for g in block_gates))
        conserved = {q for q in union & remote if all(_conserves(g, q) for g in block_gates)}
        mixed_now = {q for q in gate.indices if q in remote and not _conserves(gate, q)}
        if mixed_now & conserved & seen:
            return False             # the new gate mixes a rank bit the block only conserved so far
        return len(union) - len(conserved - mixed_now) <= max_qubits

    def flush(blocks):
        for b in blocks:
            out.append(b.emit())
            open_blocks.remove(b)

    def effective_size(block) -> int:
        qs = set(block.qubits)
        return len(qs) - len({q for q in qs & remote if all(_conserves(g, q) for g in block.gates)})

    def flush_all():
        """Everything goes out (a barrier or the end of the circuit): open blocks act on disjoint qubits, so they commute
        and may share launches -- pack them, largest first, into tensor products of at most ``max_qubits`` qubits."""
        blocks = sorted(open_blocks, key=effective_size, reverse=True)
        packed: list[_Block] = []
        for b in blocks:
            for host in packed:
                if effective_size(host) + effective_size(b) <= max_qubits and len(host.qubits) + len(b.qubits) <= MAX_LEGS:
                    host.merge(b)
                    break
            else:
                packed.append(b)
        for b in packed:
            out.append(b.emit())
        open_blocks.clear()

    for gate in circuit:
        plain = isinstance(gate, Gate) and not isinstance(gate, (M, Insert)) and gate.matrix is not None \
            and gate.matrix.shape[0] == gate.matrix.shape[1]
        if not plain or len(gate.indices) > min(max_qubits, MAX_LEGS):
            flush_all()
            out.append(gate)
            continue
        touched = [b for b in open_blocks if set(b.qubits) & set(gate.indices)]
        union = set(gate.indices).union(*(b.qubits for b in touched))
        if touched and allowed(union, [g for b in touched for g in b.gates], gate):
            first = touched[0]
            for other in touched[1:]:
                first.merge(other)
                open_blocks.remove(other)
            first.grow(list(gate.indices))
            first.absorb(gate)
        else:
            # too many qubits together: let the largest touched blocks go until the rest and the gate fit one block
            touched.sort(key=lambda b: len(b.qubits), reverse=True)
            while touched:
                flush([touched.pop(0)])
                union = set(gate.indices).union(*(b.qubits for b in touched))
                if touched and allowed(union, [g for b in touched for g in b.gates], gate):
                    break
            if touched:
                first = touched[0]
                for other in touched[1:]:
                    first.merge(other)
                    open_blocks.remove(other)
                first.grow(list(gate.indices))
                first.absorb(gate)
            else:
                open_blocks.append(_Block(gate))
    flush_all()
    return out


def fusion_stats(circuit: list, fused: list) -> dict:
    sizes = [len(g.indices) for g in fused if isinstance(g, Gate)]
    return {"gates": len(circuit), "launches": len(fused), "gates_per_launch": len(circuit) / max(1, len(fused)),
            "block_sizes": {k: sizes.count(k) for k in sorted(set(sizes))}}
