#!/usr/bin/env python3
"""BASELINE.json configs 3 and 5 on N GPUs of one node (one process per GPU, RCCL).  Not run by the driver: for whoever
has the 8-GPU node.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        tools/bench_multi.py --config remote_cx --qubits-per-gpu 31      # 34 qubits = 256 GiB over 8 GPUs
    ... tools/bench_multi.py --config grover --qubits-per-gpu 27           # Grover n = 30, k = 8 iterations

* remote_cx: 32 CX gates whose (control, target) cycle through global->local, local->global and global->global qubit
  pairs on a pseudo-random register generated shard by shard; reports gate-apps/s, the number of
  half-shard exchanges and the norm.
* grover: success probability of one marked item after 8 iterations vs sin^2(17 asin 2^(-n/2)).
QSV_BENCH_ONE_GPU=1 rehearses the code path with all ranks on GPU 0 and host-staged collectives (numbers meaningless).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=("remote_cx", "grover"), required=True)
    ap.add_argument("--qubits-per-gpu", type=int, default=28)
    ap.add_argument("--iterations", type=int, default=8)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from quantum_computations_amd import workloads as W
    from quantum_computations_amd.distributed import ShardedState, _default_engine_factory
    from quantum_computations_amd.dv_simulator import gates as G

    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    g = (world - 1).bit_length()
    n = args.qubits_per_gpu + g
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("QSV_BENCH_ONE_GPU") == "1":
        sys.path.insert(0, str(REPO / "tests"))
        from host_staged import HostStagedShardedState as State
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
        device = 0
    else:
        State = ShardedState
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        device = local_rank
    buf = torch.zeros(1 << args.qubits_per_gpu, dtype=torch.complex128, device=torch.device("cuda", device))
    st = State(n, buf, _default_engine_factory(device))

    if args.config == "remote_cx":
        st.fill_random(34)
        glob, loc = list(range(g)), list(range(g, n))
        pairs = []
        for i in range(32):
            kind = i % 3
            if kind == 0 or g < 2:
                pairs.append((glob[i % g], loc[(7 * i) % len(loc)]) if kind != 1 else (loc[(5 * i) % len(loc)], glob[i % g]))
            elif kind == 1:
                pairs.append((loc[(5 * i) % len(loc)], glob[i % g]))
            else:
                pairs.append((glob[i % g], glob[(i + 1) % g]))
        gates = [G.CX(c, t) for c, t in pairs]
        st.prepare(gates)
        dist.barrier()
        st.sync()
        t0 = time.perf_counter()
        for gate in gates:
            gate.apply(st)
        st.sync()
        dist.barrier()
        dt = time.perf_counter() - t0
        out = {"config": f"remote-qubit CX mix, n={n} over {world} GPUs", "gates": len(gates), "seconds": dt,
               "gate_apps_per_s": len(gates) / dt, "half_shard_exchanges": st.exchanges, "norm2": st.norm2()}
    else:
        if rank == 0:
            st.local.set_basis(0)
        marked = (0b1011001110001111 << max(0, n - 16)) % (1 << n) | 1
        h = G.H(0).matrix
        for q in range(n):
            st.apply_matrix(h, [q])
        dist.barrier()
        st.sync()
        t0 = time.perf_counter()
        for _ in range(args.iterations):
            W.grover_iteration(st, n, marked)
        st.sync()
        dist.barrier()
        dt = time.perf_counter() - t0
        p = float(st.probabilities([marked])[0])
        want = W.grover_success_probability(n, args.iterations)
        gates = args.iterations * W.grover_gate_count(n, marked)
        out = {"config": f"Grover n={n}, {args.iterations} iterations over {world} GPUs", "success_probability": p,
               "analytic": want, "rel_err": abs(p - want) / want, "gate_apps": gates, "seconds": dt,
               "gate_apps_per_s": gates / dt, "half_shard_exchanges": st.exchanges, "norm2": st.norm2()}
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
