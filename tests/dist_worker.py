"""Worker of the multi-process tests: run under ``python -m torch.distributed.run`` (one process per rank).

    dist_worker.py --backend gloo            # CPU tensors, oracle-backed local engine (tests/oracle_engine.py)
    dist_worker.py --backend nccl            # HBM tensors, HIP kernels (one GPU per rank)

Every rank builds the same circuits, runs them on a ``ShardedState`` and compares the gathered ket with the
CPU oracle's result for the unsharded register.  Exits non-zero on any mismatch.
"""
from __future__ import annotations

import argparse
import os
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import dv_oracle as O  # noqa: E402
from quantum_computations_amd import workloads as W  # noqa: E402
from quantum_computations_amd.distributed import ShardedState  # noqa: E402
from quantum_computations_amd.dv_simulator import gates as G  # noqa: E402
from quantum_computations_amd.dv_simulator.simulator import Simulator  # noqa: E402

TOL = 1e-12


from host_staged import HostStagedShardedState  # noqa: E402


def make_state(n, ket, backend, device, **options):
    world, rank = dist.get_world_size(), dist.get_rank()
    n_local = n - (world - 1).bit_length()
    shard = np.ascontiguousarray(ket[rank << n_local:(rank + 1) << n_local])
    if backend == "gloo":
        import oracle_engine
        buf = torch.from_numpy(shard.copy())
        return ShardedState(n, buf, oracle_engine.factory, **options)
    from quantum_computations_amd.distributed import _default_engine_factory
    buf = torch.from_numpy(shard.copy()).to(torch.device("cuda", device))
    cls = HostStagedShardedState if backend == "gloo-gpu" else ShardedState
    return cls(n, buf, _default_engine_factory(device), **options)


def check(name, got, want, tol=TOL):
    err = float(np.max(np.abs(got - want)))
    if not err < tol:
        raise AssertionError(f"[rank {dist.get_rank()}] {name}: max abs err {err:.3e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--qubits", type=int, default=9)
    ap.add_argument("--chunk-amps", type=int, default=0, help="staging piece of the exchanges (0: the 1 GiB default)")
    ap.add_argument("--overlap-chunk-amps", type=int, default=0,
                    help="also run the gates-inside-exchanges section, with this staging piece")
    args = ap.parse_args()
    if args.chunk_amps:
        os.environ["QSV_EXCHANGE_CHUNK_AMPS"] = str(args.chunk_amps)
    device = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "nccl":
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    else:
        if args.backend == "gloo-gpu":
            device = 0                       # every rank shares the one GPU of the test box
            torch.cuda.set_device(0)
        dist.init_process_group("gloo")
    world, rank = dist.get_world_size(), dist.get_rank()
    g = (world - 1).bit_length()
    n = args.qubits
    rng = np.random.default_rng(5)

    # 1. the cfg2/cfg3 generator: 1- and 2-qubit gates on arbitrary (also remote) qubits
    for seed in (11, 12):
        ops = W.random_circuit(n, 80, seed)
        ket = W.random_ket(n, seed)
        st = make_state(n, ket, args.backend, device)
        for gate in W.to_gates(ops):
            out = gate.apply(st)
            assert out is st
        want, _ = O.run_circuit(ops, ket)
        check(f"random circuit seed {seed}", st.to_numpy(), want)
        assert st.exchanges > 0, "a depth-80 circuit must touch a remote qubit"
        assert abs(st.norm2() - 1.0) < 1e-12

    # 1b. look-ahead eviction (prepare) and the exchange policy: same final state; with the plan no more exchange
    # steps than without; the all-rank-bits exchange ("auto", world >= 4) needs fewer steps and puts fewer bytes on
    # the links than one pairwise half-shard swap per remote qubit (round 1's scheme)
    ops = W.random_circuit(n, 120, 21)
    ket = W.random_ket(n, 21)
    want, _ = O.run_circuit(ops, ket)
    counts, sent, link = {}, {}, {}
    for policy in ("pairwise", "auto"):
        for planned in (False, True):
            st = make_state(n, ket, args.backend, device, policy=policy)
            gates = W.to_gates(ops)
            if planned:
                out = Simulator(gates).run(st)            # Simulator.run announces the circuit to the register
            else:
                for gate in gates:
                    gate.apply(st)
                out = st
            check(f"policy={policy} look-ahead={planned}", out.to_numpy(), want)
            counts[policy, planned], sent[policy, planned] = st.exchanges, st.bytes_sent
            link[policy, planned] = st.link_bytes
            dry = ShardedState.plan_only(n, world, policy)   # the data-free schedule agrees with what really ran
            if planned:
                dry.run_circuit(gates)
            else:
                for gate in gates:
                    gate.apply(dry)
            assert (dry.exchanges, dry.bytes_sent, dry.phys) == (st.exchanges, st.bytes_sent, st.phys)
        assert counts[policy, True] <= counts[policy, False], counts
    # round 1 routed every pairwise swap over all links in two all_to_all phases: 2 (G-1)/G half shards on the wire
    relay = 2 * (world - 1) / world if world >= 4 else 1.0
    if world >= 4:
        assert counts["auto", True] < counts["pairwise", True], counts
        assert link["auto", True] < link["pairwise", True], link            # time on the busiest link
        assert sent["auto", True] < relay * sent["pairwise", True], sent     # bytes on the wire vs round 1's scheme
    counts = {False: counts["auto", False], True: counts["auto", True], "pairwise": counts["pairwise", True]}
    sent = {"auto": sent["auto", True], "pairwise": sent["pairwise", True], "round1": int(relay * sent["pairwise", True])}
    # a gate outside the announced circuit only switches the look-ahead off (announced gates may come in any order)
    st = make_state(n, ket, args.backend, device)
    st.prepare(W.to_gates(ops))
    ccz = np.diag([1, 1, 1, 1, 1, 1, 1, -1]).astype(complex)
    G.Gate([n - 1, 0, 2], ccz).apply(st)                     # no three-qubit gate was announced
    assert st._plan is None
    check("off-plan gate", st.to_numpy(), O.apply_gate(ket, ccz, [n - 1, 0, 2]))
    st = make_state(n, ket, args.backend, device)
    gates = W.to_gates(ops)
    st.prepare(gates)
    for gate in gates[1::2] + gates[0::2]:                    # every announced gate, in another order
        st._advance_plan(gate.indices)
    assert st._plan is not None and st._cursor == len(gates)

    # 1c. gates riding inside the exchange steps (round 3): local gates whose mixing legs lie inside the slices are held
    # back and applied to every slice as it lands, while the next one travels.  Same state, same exchange schedule
    # as with the overlap switched off, and the counter shows that gates really ran in there.
    if args.overlap_chunk_amps:
        for seed, depth in ((31, 150), (32, 150)):
            ops = W.random_circuit(n, depth, seed)
            ket = W.random_ket(n, seed)
            want, _ = O.run_circuit(ops, ket)
            runs = {}
            for overlap in (True, False):
                st = make_state(n, ket, args.backend, device, chunk_amps=args.overlap_chunk_amps)
                st.overlap, st.ride_min_bits = overlap, 1
                out = Simulator(W.to_gates(ops)).run(st)
                check(f"overlap={overlap} seed {seed}", out.to_numpy(), want)
                runs[overlap] = (st.exchanges, st.bytes_sent, list(st.phys), st.gates_in_exchanges, st.rider_launches)
                if overlap:
                    dry = ShardedState.plan_only(n, world)
                    dry.chunk_amps, dry.ride_min_bits = args.overlap_chunk_amps, 1
                    dry.run_circuit(W.to_gates(ops))
                    assert (dry.exchanges, dry.phys, dry.gates_in_exchanges) == (st.exchanges, st.phys, st.gates_in_exchanges)
            assert runs[True][:3] == runs[False][:3], runs            # riding never changes the schedule
            assert runs[True][3] > 0 and runs[False][3] == 0, runs
            ridden = runs[True][3:]
        # fused blocks and Grover's walls ride as well; a forced measurement in the middle is a barrier
        st = make_state(n, ket, args.backend, device, chunk_amps=args.overlap_chunk_amps)
        st.ride_min_bits = 1
        out = Simulator(W.to_gates(ops), fuse=3).run(st)
        check("overlap, fused blocks", out.to_numpy(), want)      # (few blocks have all their mixing legs that low)
        ket = W.random_ket(n, 9)
        st = make_state(n, ket, args.backend, device, chunk_amps=args.overlap_chunk_amps)
        st.ride_min_bits = 1
        circuit = [G.H(n - 1), G.T(n - 2), G.H(0), G.CX(n - 1, 1), G.MZ(n - 1, result=1), G.H(n - 2), G.H(1), G.H(0)]
        sim = Simulator(circuit)
        out = sim.run(st)
        ops_m = [W.op("H", n - 1), W.op("T", n - 2), W.op("H", 0), W.op("CX", n - 1, 1),
                 {"name": "M", "indices": [n - 1], "theta": 0.0, "phi": 0.0, "result": 1, "matrix": None},
                 W.op("H", n - 2), W.op("H", 1), W.op("H", 0)]
        want_m, _ = O.run_circuit(ops_m, ket)
        check("overlap with a measurement barrier", out.to_numpy(), want_m)
        # read-out on one rank only: the others get nothing, rank 1 gets the ket
        got = out.to_numpy(root=world - 1)
        assert (got is None) == (rank != world - 1)
        if got is not None:
            check("to_numpy(root)", got, want_m)
        if rank == 0:
            print(f"overlap ok: gates_in_exchanges={ridden[0]} rider_launches={ridden[1]}")

    # 2. the remote-qubit CX mix of BASELINE config 3: global->local, local->global, global->global
    ket = W.random_ket(n, 3)
    st = make_state(n, ket, args.backend, device)
    pairs = [(0, n - 1), (n - 1, 0), (n - 2, n - 3)]
    if g >= 2:
        pairs += [(0, 1), (1, 0)]
    pairs += [(0, n - 1), (n - 1, 0)]          # again, after the layout has changed
    want = ket
    cx = W.op("CX", 0, 1)["matrix"]
    before = st.exchanges
    for c, t in pairs:
        G.CX(c, t).apply(st)
        want = O.apply_gate(want, cx, [c, t])
    check("CX mix", st.to_numpy(), want)
    exchanges_cx = st.exchanges - before
    # diagonal gates and controls on remote qubits never exchange anything
    before = st.exchanges
    for q in range(n):
        G.T(q).apply(st)
        want = O.apply_gate(want, G.T(q).matrix, [q])
        G.CZ(q, (q + 3) % n).apply(st)
        want = O.apply_gate(want, G.CZ(0, 1).matrix, [q, (q + 3) % n])
    st.apply_mcphase(list(range(0, n, 2)), np.exp(0.7j))
    d = np.ones(1 << len(range(0, n, 2)), dtype=complex)
    d[-1] = np.exp(0.7j)
    want = O.apply_gate(want, np.diag(d), list(range(0, n, 2)))
    assert st.exchanges == before, "diagonal gates must not communicate"
    check("diagonal gates on remote qubits", st.to_numpy(), want)
    # SWAP is a relabelling
    G.SWAP(0, n - 1).apply(st)
    want = O.apply_gate(want, G.SWAP(0, 1).matrix, [0, n - 1])
    assert st.exchanges == before
    check("swap relabel", st.to_numpy(), want)
    # controlled gates with remote controls / remote target
    u = W.haar_unitary(2, rng)
    for controls, target in [([0, n - 1], 3), ([2, 3], 0), ([0], 1), ([n - 1, n - 2, 1], 0)]:
        st.apply_controlled(u, controls, target)
        full = np.identity(1 << (len(controls) + 1), dtype=complex)
        full[-2:, -2:] = u
        want = O.apply_gate(want, full, controls + [target])
    check("controlled gates", st.to_numpy(), want)
    probe = [0, 1, (1 << n) - 1, 37 % (1 << n), 1 << (n - 1)]
    check("probabilities", st.probabilities(probe), np.abs(want[probe]) ** 2, 1e-14)
    # read-out across the shards: reduced density matrices and Pauli strings with legs on rank bits
    for kept in ([n - 1], [0], [1, 0, n - 2], [n - 1, 3, 0, 1]):
        want_now = st.to_numpy()
        check(f"reduced density {kept}", st.reduced_density(kept), O.reduced_density(want_now, kept), 1e-13)
    paulis = {"I": np.identity(2), "X": np.array([[0, 1], [1, 0]]), "Y": np.array([[0, -1j], [1j, 0]]), "Z": np.diag([1.0, -1.0])}
    for letters, qs in (("Z", [0]), ("ZZ", [0, n - 1]), ("XZ", [0, 1]), ("YXZI", [1, n - 1, 0, 2]), ("ZY", [n - 2, 0])):
        want_now = st.to_numpy()
        out = want_now
        for p, q in zip(letters, qs):
            out = O.apply_gate(out, paulis[p].astype(complex), [q])
        got = st.expect_pauli(letters, qs)
        assert abs(got - np.vdot(want_now, out)) < 1e-13, (letters, qs, got)

    # 3. measurement (forced outcomes) of a remote and a local qubit, through the Simulator
    ket = W.random_ket(n, 4)
    st = make_state(n, ket, args.backend, device)
    circuit = [G.H(0), G.CX(0, n - 1), G.M(0, 0.4, 1.1, result=1), G.H(n - 3), G.MZ(n - 2, result=0), G.H(0)]
    sim = Simulator(circuit)
    out = sim.run(st)
    ops = [W.op("H", 0), W.op("CX", 0, n - 1),
           {"name": "M", "indices": [0], "theta": 0.4, "phi": 1.1, "result": 1, "matrix": None},
           W.op("H", n - 3), {"name": "M", "indices": [n - 2], "theta": 0.0, "phi": 0.0, "result": 0, "matrix": None},
           W.op("H", 0)]
    want, results = O.run_circuit(ops, ket)
    assert sim.results == results == [1, 0]
    assert out.num_qubits == n - 2
    check("measurement", out.to_numpy(), want)

    # 3'. unforced measurements: every rank has its OWN np.random state (one process per GPU), the register must
    # still collapse onto one outcome everywhere and ClassicalControl must fire on every rank or on none
    from quantum_computations_amd.dv_simulator.simulator import ClassicalControl
    ket = W.random_ket(n, 8)
    for trial in range(3):
        np.random.seed(1000 * trial + 17 * rank + 1)               # deliberately different on every rank
        st = make_state(n, ket, args.backend, device)
        circuit = [G.H(0), G.H(n - 1), G.CX(0, 1), G.MX(0), G.M(n - 2, 0.9, 0.3),
                   ClassicalControl(G.X(0), [0], []), ClassicalControl(G.H(1), [], [1]), G.H(0)]
        sim = Simulator(circuit)
        out = sim.run(st)
        flags = torch.tensor([float(b) for b in sim.results], dtype=torch.float64)
        gathered = [torch.empty_like(flags) for _ in range(world)]
        dist.all_gather(gathered, flags)
        assert all(torch.equal(gathered[0], t) for t in gathered), f"ranks disagree on outcomes: {gathered}"
        r0, r1 = sim.results
        ops = [W.op("H", 0), W.op("H", n - 1), W.op("CX", 0, 1),
               {"name": "M", "indices": [0], "theta": np.pi / 2, "phi": 0.0, "result": r0, "matrix": None},
               {"name": "M", "indices": [n - 2], "theta": 0.9, "phi": 0.3, "result": r1, "matrix": None}]
        ops += [W.op("X", 0)] if r0 == 1 else []
        ops += [W.op("H", 1)] if r1 == 0 else []
        ops += [W.op("H", 0)]
        want, _ = O.run_circuit(ops, ket)
        check(f"unforced measurement, trial {trial} -> {sim.results}", out.to_numpy(), want)

    # 3a. insertion: the new qubit lands on a local bit, wherever the reference order puts it
    ket = W.random_ket(n, 6)
    st = make_state(n, ket, args.backend, device)
    G.H(0).apply(st)
    want = O.apply_gate(ket, G.H(0).matrix, [0])
    before = st.exchanges
    for position, amplitudes in [(0, [0.6, 0.8j]), (st.num_qubits + 1, [1.0, 0.0]), (3, [2 ** -0.5, -(2 ** -0.5)])]:
        st.insert(position, amplitudes)
        want = O.insert_qubit(want, position, np.array(amplitudes, dtype=complex))
    assert st.exchanges == before, "inserting a product qubit must not communicate"
    assert st.num_qubits == n + 3
    check("insert", st.to_numpy(), want)
    G.CX(0, st.num_qubits - 1).apply(st)             # the grown register keeps working, remote qubits included
    G.H(3).apply(st)
    want = O.apply_gate(want, W.op("CX", 0, 1)["matrix"], [0, n + 2])
    want = O.apply_gate(want, G.H(0).matrix, [3])
    check("gates after insert", st.to_numpy(), want)
    with np.testing.assert_raises(ValueError):
        st.insert(st.num_qubits + 1, [1, 0])

    # 3a'. fused circuits (Simulator(fuse=k)): dense blocks of up to k qubits with local and remote legs
    ops = W.random_circuit(n, 90, 43)
    ket = W.random_ket(n, 7)
    want, _ = O.run_circuit(ops, ket)
    for k in (3, 5):
        st = make_state(n, ket, args.backend, device)
        out = Simulator(W.to_gates(ops), fuse=k).run(st)
        check(f"fused circuit, blocks of <= {k} qubits", out.to_numpy(), want)

    # 3b. BASELINE config 5 in small: Grover search on the sharded register, success probability vs the analytic value
    marked = (0b1011001110 >> max(0, 10 - n)) | 1
    start = np.zeros(1 << n, dtype=complex)
    start[0] = 1
    st = make_state(n, start, args.backend, device)
    for q in range(n):
        G.H(q).apply(st)
    iterations = 3
    for _ in range(iterations):
        W.grover_iteration(st, n, marked)
    p = float(st.probabilities([marked])[0])
    want_p = W.grover_success_probability(n, iterations)
    assert abs(p - want_p) < 1e-12, (p, want_p)
    assert abs(st.norm2() - 1.0) < 1e-12

    # 3c. the same search as an announced gate list: walls of H and X on every qubit are taken local-first, so a wall
    # costs one exchange (if any) instead of one per remote qubit; same probabilities, fewer exchange steps
    direct_steps = st.exchanges
    st = make_state(n, start, args.backend, device)
    circuit = [G.H(q) for q in range(n)] + W.grover_circuit(n, marked, iterations)
    out = Simulator(circuit).run(st)
    assert out is st and abs(float(st.probabilities([marked])[0]) - want_p) < 1e-12
    assert st.exchanges < direct_steps, (st.exchanges, direct_steps)
    grover_steps = (direct_steps, st.exchanges)
    fused = Simulator(circuit, fuse=2).run(make_state(n, start, args.backend, device))     # X H X per qubit in one launch
    assert abs(float(fused.probabilities([marked])[0]) - want_p) < 1e-12
    from quantum_computations_amd.fusion import fuse_circuit
    assert len(circuit) > 2 * len(fuse_circuit(circuit, 2))

    # 4. counter-based fill: the sharded register equals the unsharded one
    if args.backend == "gloo":
        import oracle_engine
        st = make_state(n, np.zeros(1 << n, dtype=complex), args.backend, device)
        st.fill_random(77)
        whole = oracle_engine.counter_normal(77, 0, 1 << n)
        check("fill_random", st.to_numpy(), whole / np.linalg.norm(whole), 1e-13)

    dist.barrier()
    if rank == 0:
        print(f"dist_worker ok: world={world} backend={args.backend} n={n} chunk_amps={st.chunk_amps} "
              f"cx_exchanges={exchanges_cx} exchange_steps_120_gates no-plan={counts[False]} look-ahead={counts[True]} "
              f"pairwise={counts['pairwise']} bytes_sent_per_rank auto={sent['auto']} pairwise_direct={sent['pairwise']} "
              f"pairwise_over_all_links(round 1)={sent['round1']} grover_exchange_steps direct={grover_steps[0]} "
              f"announced={grover_steps[1]}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
