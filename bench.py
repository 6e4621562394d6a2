#!/usr/bin/env python3
"""Headline benchmark: gate-apps/sec + achieved HBM GB/s on a 28-qubit complex128 register per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One *step* = one pass of BASELINE.json's config 2 circuit (depth-100 random 1+2-qubit gates, generator of
SURVEY.md 8d, seed 100) over the register, every gate going through the public ``Gate.apply`` -> C ABI -> HIP
path with the register resident in HBM.  N = 1: n = 28 qubits (4 GiB).  N > 1: weak scaling, n = 28 + log2(N)
qubits sharded over the ranks (top qubits = rank id), gates on remote qubits exchange half shards over RCCL;
``value`` is reported in 28-qubit gate-app equivalents (amplitude updates / 2^28 per second) so that it is the
whole-job aggregate, and the raw gate-apps/s on the larger register is given beside it.

Rank 0 prints ONE JSON line.  ``roofline`` is measured live with HIP events around every dense-kernel launch of
the timed region; ``cpu_baseline`` is the C/OpenMP restatement (oracle/, "port") timed on this host on a bounded
prefix of the same circuit (rank 0, N = 1 only), and doubles as a full-size parity check.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

QUBITS_PER_GPU = 28
DEPTH = 100
CIRCUIT_SEED = 100
STATE_SEED = 28
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s


CPU_THREADS = 16        # the GPU box's CPU share for one GPU


def cpu_baseline(ops, dev, n, budget_s=12.0):
    """Time the C/OpenMP oracle on a prefix of the circuit, then use the result as a full-size parity check."""
    from oracle import c_oracle   # checker / baseline only -- never on the product path

    want = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(CPU_THREADS, len(os.sched_getaffinity(0)))
    threads = c_oracle.threads(want)            # omp_set_num_threads + omp_get_max_threads: what really runs
    host = dev.to_numpy()                       # the same initial state the GPU starts from
    t_total, done = 0.0, 0
    for op in ops:
        t0 = time.perf_counter()
        c_oracle.apply_gate_inplace(host, op["matrix"], op["indices"])
        t_total += time.perf_counter() - t0
        done += 1
        if t_total > budget_s and done >= 4:
            break
    # parity at full size: run the same prefix on the GPU and compare every amplitude
    from quantum_computations_amd import workloads as W
    for gate in W.to_gates(ops[:done]):
        gate.apply(dev)
    err = float(np.max(np.abs(dev.to_numpy() - host)))
    return {
        "value": done / t_total, "unit": "gate-apps/s", "cores": threads, "kind": "port",
        "sample": f"first {done} of the {len(ops)} gates of the same n={n} circuit, C/OpenMP restatement "
                  f"(oracle/csrc/qsv_oracle.c), {t_total:.1f} s",
    }, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fusion", action="store_true", help="skip the informational fused-circuit measurement")
    ap.add_argument("--qubits-per-gpu", type=int, default=QUBITS_PER_GPU)
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the gate path has no CPU fallback")
    # Development aid only (never set by the driver): QSV_BENCH_ONE_GPU=1 lets N ranks share GPU 0 with the
    # collectives staged through host memory over gloo (tests/host_staged.py), to rehearse the N > 1 code path on
    # a one-GPU box.  Its numbers are meaningless as a measurement.
    one_gpu_rehearsal = world > 1 and os.environ.get("QSV_BENCH_ONE_GPU") == "1"
    if one_gpu_rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    from quantum_computations_amd import workloads as W

    g_bits = (world - 1).bit_length()
    if 1 << g_bits != world:
        raise SystemExit("the register shards over a power-of-two number of GPUs")
    n_local = args.qubits_per_gpu
    n = n_local + g_bits
    ops = W.random_circuit(n, DEPTH, CIRCUIT_SEED)
    gates = W.to_gates(ops)
    gate_names = [o["name"] for o in ops]

    if world == 1:
        from quantum_computations_amd.device import DeviceState
        dev = DeviceState.random(n, seed=STATE_SEED, device=local_rank)
        barrier = lambda: None
        reduce_max = lambda x: x
        comm_info = {}
    else:
        import torch.distributed as dist
        from quantum_computations_amd.distributed import ShardedState
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_gpu_rehearsal:
            sys.path.insert(0, str(REPO / "tests"))
            from host_staged import HostStagedShardedState
            from quantum_computations_amd.distributed import _default_engine_factory
            dist.init_process_group("gloo")
            buf = torch.empty(1 << n_local, dtype=torch.complex128, device="cuda:0")
            dev = HostStagedShardedState(n, buf, _default_engine_factory(0))
            dev.fill_random(STATE_SEED)
            reduce_device = "cpu"
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dev = ShardedState.random(n, seed=STATE_SEED, device=local_rank)
            reduce_device = "cuda"
        barrier = dist.barrier

        def reduce_max(x):
            t = torch.tensor([x], dtype=torch.float64, device=reduce_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        comm_info = {}

    cpu, parity_err = (None, None)
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu, parity_err = cpu_baseline(ops, dev, n)
        dev.fill_random(STATE_SEED)             # restart from the initial state

    def step(record: bool, slot0: int = 0):
        slot = slot0
        if world > 1:
            dev.prepare(gates)       # what Simulator.run does: lets the shards plan which qubit to give up
        for gate in gates:
            if record:
                dev.event_record(slot)
            gate.apply(dev)
            if record:
                dev.event_record(slot + 1)
            slot += 2
        return slot

    # which kernel each gate of the circuit lands in (asked of the library, not guessed)
    kernels = []
    if world == 1:
        for gate in gates:
            gate.apply(dev)
            kernels.append(dev.last_kernel())
    for _ in range(args.warmup):
        step(False)
    recorded_steps = min(args.steps, 16000 // (2 * DEPTH)) if world == 1 else 0   # the event ring holds 16384 marks
    barrier()
    torch.cuda.synchronize()
    dev.sync()
    t0 = time.perf_counter()
    slot = 0
    for k in range(args.steps):
        slot = step(k < recorded_steps, slot)
    dev.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = reduce_max(time.perf_counter() - t0)

    if rank != 0:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return

    gate_apps = args.steps * DEPTH
    register_rate = gate_apps / max(elapsed, 1e-12)            # gate-apps/s on the n-qubit register
    value = register_rate * (1 << g_bits)                      # 28-qubit equivalents, whole job
    bytes_per_gate_per_gpu = 2 * 16 * (1 << n_local)
    result = {
        "metric": "gate_apps_per_sec", "value": value, "unit": "gate-apps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, args.steps),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"{n}-qubit complex128 state vector ({n_local} qubits = {16 * (1 << n_local) / 2**30:.0f} GiB "
                        f"per GPU), depth-{DEPTH} random 1+2-qubit gates (BASELINE.json configs[1], seed {CIRCUIT_SEED})",
            "n_qubits": n, "depth": DEPTH,
            "gate_mix": {c: gate_names.count(c) for c in sorted(set(gate_names))},
            "sharding": f"top {g_bits} qubits = rank id" if g_bits else "single GPU",
            "unit_note": "value = gate-apps/s on the full register x 2^(n-28): 28-qubit gate-app equivalents",
        },
        "gate_apps_per_sec_on_register": register_rate,
        **({"half_shard_exchanges_per_step": getattr(dev, "exchanges", 0) / max(1, args.steps + args.warmup),
            "exchange": "two-phase all_to_all over all xGMI links" if world >= 4 else "pairwise send/recv"}
           if world > 1 else {}),
        "algorithmic_GBps_per_gpu": bytes_per_gate_per_gpu * register_rate / 1e9,
    }
    if recorded_steps:
        per_kernel = {}
        for s in range(recorded_steps):
            for i, c in enumerate(kernels):
                a = 2 * (s * DEPTH + i)
                per_kernel.setdefault(c, []).append(dev.event_elapsed_ms(a, a + 1))
        # dominant kernel = the full-traffic dense instantiation with the most device time
        full = {k: v for k, v in per_kernel.items() if k.startswith("k_dense<")}
        dominant = max(full, key=lambda k: sum(full[k]))
        avg_ms = float(np.mean(full[dominant]))
        achieved = bytes_per_gate_per_gpu / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        pmc = REPO / "profiles" / "pmc_traffic.json"        # written by tools/summarize_profile.py from rocprofv3 --pmc
        if pmc.exists():
            entry = json.loads(pmc.read_text()).get(dominant)
            if entry:
                traffic, traffic_src = entry["hbm_bytes_per_launch"], entry["source"]
        result["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": dominant, "algorithmic_bytes_per_launch": bytes_per_gate_per_gpu,
            "avg_launch_ms": avg_ms, "launches_timed": len(full[dominant]),
        }
        all_full = [t for v in full.values() for t in v]
        result["dense_full_traffic_all_kernels"] = {
            "launches": len(all_full), "avg_ms": float(np.mean(all_full)),
            "GBps": bytes_per_gate_per_gpu / (np.mean(all_full) * 1e-3) / 1e9,
            "frac_of_peak": bytes_per_gate_per_gpu / (np.mean(all_full) * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        result["per_kernel_ms"] = {c: {"launches": len(v), "avg_ms": float(np.mean(v)),
                                       "equiv_GBps": bytes_per_gate_per_gpu / (np.mean(v) * 1e-3) / 1e9}
                                   for c, v in sorted(per_kernel.items())}
    if cpu is not None:
        result["cpu_baseline"] = cpu
        result["parity_max_abs_err_vs_cpu_at_full_size"] = parity_err
    if world == 1 and not args.no_fusion:
        # Informational, outside the timed region and NOT part of `value`: the same circuit through the gate-fusion
        # scheduler of Simulator(fuse=k) (quantum_computations_amd/fusion.py) -- fewer, denser launches.
        from quantum_computations_amd.fusion import fuse_circuit, fusion_stats
        result["with_gate_fusion"] = {}
        for k in (3, 4, 5):
            fused = fuse_circuit(gates, k, n_qubits=n)
            for g in fused:
                g.apply(dev)
            dev.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                for g in fused:
                    g.apply(dev)
            dev.sync()
            dt = time.perf_counter() - t0
            result["with_gate_fusion"][f"max_{k}_qubits"] = {
                **fusion_stats(gates, fused), "gate_apps_per_sec": args.steps * DEPTH / dt,
                "ms_per_step": 1e3 * dt / args.steps}
    print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
