"""Worker of the large sharded-register test: 2 (or 4) ranks on ONE GPU (send/recv staged through the host over gloo), a
register of ``--qubits`` qubits (default 30: two 8 GiB shards), exchanges at the production piece size of 1 GiB, so the
half shard of 4 GiB travels in four double-buffered slices on device tensors.  No CPU oracle can hold this register:
the checks are known answers (basis states pushed around by X / CX / SWAP through rank bits), a circuit followed by its
inverse (must return to the start: |<start|end>|^2 = 1), and norms.

    python -m torch.distributed.run --nproc-per-node 2 tests/dist_big_worker.py [--qubits 30]
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from host_staged import HostStagedShardedState  # noqa: E402
from quantum_computations_amd import workloads as W  # noqa: E402
from quantum_computations_amd.distributed import _default_engine_factory  # noqa: E402
from quantum_computations_amd.dv_simulator import gates as G  # noqa: E402
from quantum_computations_amd.dv_simulator.simulator import Simulator  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--qubits", type=int, default=30)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    world, rank = dist.get_world_size(), dist.get_rank()
    n = args.qubits
    n_local = n - (world - 1).bit_length()
    buf = torch.zeros(1 << n_local, dtype=torch.complex128, device="cuda:0")
    st = HostStagedShardedState(n, buf, _default_engine_factory(0))
    assert st.chunk_amps == 1 << 26                       # the production piece: 1 GiB

    # 1. known answers through the rank bit: |0..0> -> X(0) -> index 2^(n-1); CX(0, n-1); SWAP(0, 5); H on the rank bit
    st.set_basis(0)
    top = 1 << (n - 1)
    G.X(0).apply(st)                                      # mixes the remote qubit: one exchange
    assert st.exchanges == 1
    if world == 2:
        assert st.messages == max(1, (1 << (n_local - 1)) >> 26)          # the half shard in 1 GiB slices
    else:
        piece = (1 << n_local) // world                                    # all rank bits at once: shard / G to every peer
        slice_amps = 1 << (((1 << 26) // (world - 1)).bit_length() - 1)
        assert st.messages == (world - 1) * max(1, piece // slice_amps) and st.qubits_exchanged == (world - 1).bit_length()
    assert abs(st.probabilities([top])[0] - 1.0) < 1e-14
    G.CX(0, n - 1).apply(st)
    assert abs(st.probabilities([top | 1])[0] - 1.0) < 1e-14
    G.SWAP(0, 5).apply(st)
    assert abs(st.probabilities([(1 << (n - 6)) | 1])[0] - 1.0) < 1e-14
    G.H(5).apply(st)                                      # qubit 5 holds the data that came from the rank bit
    p = st.probabilities([(1 << (n - 6)) | 1, 1])
    assert abs(p[0] - 0.5) < 1e-14 and abs(p[1] - 0.5) < 1e-14
    assert abs(st.norm2() - 1.0) < 1e-13

    # 1b. the sharded register against the UNSHARDED one on the same GPU (round 3): the same counter-based initial state,
    # the same 40-gate circuit through Simulator.run on both (the sharded run plans its exchanges, reorders commuting gates
    # and lets low-bit gates ride inside the exchange steps); probabilities of 256 basis states and the reduced density
    # matrix of one remote + one local qubit must agree.  Rank 0 holds the unsharded copy (n <= 31: 32 GiB).
    if n <= 31:
        from quantum_computations_amd.device import DeviceState
        st.fill_random(11)
        circuit = W.to_gates(W.random_circuit(n, 40, 17))
        Simulator(circuit).run(st)
        rng = np.random.default_rng(3)
        probe = [int(v) for v in rng.integers(0, 1 << n, 256)]
        got_p = st.probabilities(probe)
        got_rho = st.reduced_density([0, n - 1])
        if rank == 0:
            ref = DeviceState.random(n, 11)
            for gate in circuit:
                gate.apply(ref)
            want_p = ref.probabilities(probe)
            want_rho = ref.reduced_density([0, n - 1])
            ref.close()
            assert float(np.max(np.abs(got_p - want_p))) < 1e-12 * float(np.max(want_p)) + 1e-24, "sharded vs unsharded probabilities"
            assert float(np.max(np.abs(got_rho - want_rho))) < 1e-12, "sharded vs unsharded reduced density matrix"
        dist.barrier()

    # 2. a random circuit and its inverse on a random register: back to the start
    st.fill_random(7)
    probe = [0, 1, top, top + 12345, (1 << n) - 1]
    before = st.probabilities(probe)
    ops = W.random_circuit(n, 24, 3)
    forward = W.to_gates(ops)
    inverse = [G.Gate(list(g.indices), np.conjugate(np.asarray(g.matrix, dtype=complex)).T) for g in reversed(forward)]
    exchanges_before = st.exchanges
    Simulator(forward + inverse).run(st)
    assert st.exchanges > exchanges_before                # the circuit did use the rank bit
    after = st.probabilities(probe)
    assert float(np.max(np.abs(after - before))) < 1e-20 + 1e-9 * float(np.max(before)), (before, after)
    assert abs(st.norm2() - 1.0) < 1e-12
    dist.barrier()
    if rank == 0:
        print(f"dist_big_worker ok: n={n} shards of {16 * (1 << n_local) / 2**30:.0f} GiB, exchange steps={st.exchanges}, "
              f"messages={st.messages}, GiB sent per rank={st.bytes_sent / 2**30:.1f}, "
              f"gates inside exchanges={st.gates_in_exchanges} ({st.rider_launches} slice launches)")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
