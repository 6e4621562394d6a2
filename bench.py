#!/usr/bin/env python3
"""Headline benchmark: gate-apps/sec + achieved HBM GB/s on a 28-qubit complex128 register per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--config cfg2|cfg3|cfg4|cfg5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

``python bench.py --gpus N`` with N > 1 and no torchrun environment starts the N ranks itself: before anything
touches a GPU it runs ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a child process, relays
its output and exits with its return code.

One *step* = one pass of the configuration's circuit over the register, every gate going through the public
``Gate.apply`` -> C ABI -> HIP path with the register resident in HBM:

* ``cfg2`` (default; BASELINE.json configs[1], the configuration the metric is quoted on): depth-100 random 1+2-qubit
  gates, generator of SURVEY.md 8d, seed 100.  N = 1: n = 28 qubits (4 GiB).  N > 1, ``--scaling weak``: n = 28 +
  log2(N) qubits sharded over the ranks (top qubits = rank id); ``--scaling strong``: n = 28 whatever N.
* ``cfg3`` (configs[2]): the remote-qubit CX mix on 31 + log2(N) qubits (34 qubits = 256 GiB on 8 GPUs).
* ``cfg4`` (configs[3], one GPU): 6 modes x Fock cutoff 32, squeezing + beam splitters through the cv_simulator API.
* ``cfg5`` (configs[4]): Grover search on 27 + log2(N) qubits (n = 30 on 8 GPUs), 8 iterations per step, success
  probability checked against sin^2(17 asin 2^(-n/2)).

``value`` is the whole-job aggregate in 28-qubit gate-app equivalents (amplitude updates / 2^28 per second); the raw
gate-apps/s on the register is given beside it.  Rank 0 prints ONE JSON line.  ``roofline`` is measured live with HIP
events around every dense-kernel launch of the timed region; ``cpu_baseline`` is the C/OpenMP restatement (oracle/,
"port") timed on this host on a bounded prefix of the same circuit (rank 0, N = 1 only), and doubles as a full-size
parity check of both the per-gate and the fused path.  ``secondary`` (N = 1, cfg2) adds, outside the timed region,
config 4 and the two other CPU baselines of BASELINE.md section 4.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

QUBITS_PER_GPU = {"cfg2": 28, "cfg3": 31, "cfg5": 27}
DEPTH = 100
CIRCUIT_SEED = 100
STATE_SEED = 28
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); the copy rate of THIS box is measured live
                         # (measure_copy_bandwidth) and reported beside it
NCCL_TIMEOUT_S = 180     # a stalled collective ends the rank with a message instead of running into the driver's limit
CPU_THREADS = 16         # the GPU box's CPU share for one GPU
PMC_QUBITS = 28          # register size profiles/pmc_traffic.json was collected on


def self_launch(args) -> int:
    """Start the N ranks as fresh child processes (this process has not touched a GPU and never will)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:                    # rank 0's JSON line (and anything else the ranks print)
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


# ---- CPU legs (test infrastructure used as baselines; never on the product path) -------------------------------------
def cpu_baseline(ops, dev, n, budget_s=12.0):
    """Time the C/OpenMP oracle on a prefix of the circuit; returns the baseline record, the number of gates it ran
    and the resulting host ket (the full-size parity reference for the per-gate and the fused GPU runs)."""
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
    return {
        "value": done / t_total, "unit": "gate-apps/s", "cores": threads, "kind": "port",
        "sample": f"first {done} of the {len(ops)} gates of the same n={n} circuit, C/OpenMP restatement "
                  f"(oracle/csrc/qsv_oracle.c), {t_total:.1f} s",
    }, done, host


def numpy_restatement_baseline(ops, n, seed, budget_s=9.0):
    """the O(2^n) NumPy restatement (oracle/dv_oracle.py) on a bounded prefix of the same circuit."""
    from oracle import dv_oracle
    from quantum_computations_amd import workloads as W

    ket = W.random_ket(n, seed)
    t_total, done = 0.0, 0
    for op in ops:
        t0 = time.perf_counter()
        ket = dv_oracle.apply_gate(ket, op["matrix"], op["indices"])
        t_total += time.perf_counter() - t0
        done += 1
        if t_total > budget_s:
            break
    return {"value": done / t_total, "unit": "gate-apps/s", "kind": "port",
            "cores": f"{len(os.sched_getaffinity(0))} visible, OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS', 'unset')}"
                     " (NumPy's tensordot/einsum path is mostly single-threaded)",
            "sample": f"first {done} gates of the n={n} circuit, oracle/dv_oracle.py (NumPy {np.__version__}), {t_total:.1f} s"}


def literal_dense_baseline(n=12):
    """the reference's literal algorithm (2^n x 2^n operator by kron + permutation, then a mat-vec;
    dv_simulator/gates.py:44-54, numpy_quantum.py:243-247) restated with the package's small-N host helpers, on THIS
    host -- calibrates it against the build container, the only place the reference itself could be timed."""
    from quantum_computations_amd.dv_simulator import numpy_quantum as npq

    rng = np.random.default_rng(0)
    ket = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    out = {}
    for name, matrix, targets in [("H", npq.H, [n // 2]), ("CX", npq.CX, [1, n - 2])]:
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            _ = npq.expand_gate(matrix, n, targets) @ ket
            times.append(time.perf_counter() - t0)
        out[name + "_ms"] = 1e3 * sorted(times)[1]
    mean_s = np.mean(list(out.values())) * 1e-3
    return {"value": 1.0 / mean_s, "unit": "gate-apps/s", "kind": "port", "cores": len(os.sched_getaffinity(0)),
            "sample": f"expand_gate + dense mat-vec at n={n}, H and CX, median of 3", **out}


def measure_copy_bandwidth(dev, reps=10):
    """the fraction is reported against the 8 TB/s spec AND against what a plain copy reaches on this
    box: the library's nontemporal copy kernel (k_copy: 1 KiB per wave-instruction, every byte read once and written
    once -- the traffic shape of a gate) from this register into a second one of the same size, HIP events."""
    other = dev.copy()
    dev.copy_into(other)
    other.sync()
    other.timer_start()
    for _ in range(reps):
        dev.copy_into(other)
    ms = other.timer_stop() / reps
    nbytes = 2 * 16 * dev.num_amps
    other.close()
    return {"GBps": nbytes / (ms * 1e-3) / 1e9, "avg_ms": ms, "bytes_per_copy": nbytes, "reps": reps,
            "kernel": "k_copy (nontemporal dwordx4 loads and stores, 16 KiB per workgroup)"}


def gate_class_and_regime(op, n):
    """SURVEY.md 8d buckets: gate class (dense / diag / perm) x stride regime of the target bits (low: a target inside
    a wavefront's 1 KiB, bits 0-5; high: a target on bits >= 20, strides of 16 MiB and more; mid: the rest), and the
    fraction of the register the specialised kernel really moves (CX, SWAP: half; CZ: a quarter)."""
    name = op["name"]
    bits = [n - 1 - q for q in op["indices"]]
    m = np.asarray(op["matrix"])
    diagonal = not np.any(m - np.diag(np.diag(m)))
    cls = "perm" if name in ("CX", "SWAP") else "diag" if diagonal else "dense"
    regime = "low" if min(bits) < 6 else "high" if max(bits) >= 20 else "mid"
    moved = {"CX": 0.5, "SWAP": 0.5, "CZ": 0.25}.get(name, 1.0)
    if name == "CZ" and min(bits) < 3:
        moved = 0.5 if max(bits) >= 3 else 1.0      # a control inside a 128-byte line cannot be skipped
    if name == "CX" and bits[0] < 3:
        moved = 1.0
    if name == "SWAP" and min(bits) < 3:
        moved = 1.0
    return cls, regime, moved


# ---- config 4: the Fock-truncated CV path ---------------------------------------------------------------------------
def run_cfg4(steps=1, n_modes=6, d=32, gates=60):
    """6 modes x cutoff 32 (2^30 amplitudes, 16 GiB), 60 gates alternating S(r = 0.1 k mod 0.5) on mode k mod 6 and
    BS(i, i + 1, pi / 4).  Returns per-gate-class device times from HIP events."""
    from quantum_computations_amd.cv_simulator import fock

    st = fock.FockState(n_modes, d)
    seq, mats = [], {}
    for k in range(gates // 2):
        seq.append(("S", k % n_modes, round(0.1 * k % 0.5, 12)))
        seq.append(("BS", k % (n_modes - 1), np.pi / 4))
    for kind, _, arg in seq:                                       # host matrices: built once per distinct gate
        if (kind, arg) not in mats:
            mats[kind, arg] = fock.squeeze_matrix(d, arg, 0.0) if kind == "S" else fock.beamsplitter_blocks(d, arg)

    def one_pass(timed):
        for kind, i, arg in seq:
            if timed is not None:
                st.reg.timer_start()
            if kind == "S":
                st.reg.apply_mode(mats[kind, arg], i)
            else:
                st.reg.apply_two_mode_blocks(mats[kind, arg], i, i + 1)
            if timed is not None:
                timed.setdefault((kind, i), []).append(st.reg.timer_stop())

    one_pass(None)                                                  # warm-up
    st.reg.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass(None)
    st.reg.sync()
    dt = time.perf_counter() - t0
    per = {}
    one_pass(per)                                                   # per-gate device times (events serialise: untimed)
    gbytes = 2 * 16 * d ** n_modes / 1e9

    def cls(kind, pick=lambda i: True):
        t = [x for (k, i), v in per.items() if k == kind and pick(i) for x in v]
        ms = float(np.mean(t))
        return {"avg_ms": ms, "GBps": gbytes / (ms * 1e-3), "frac_of_peak": gbytes / (ms * 1e-3) / HBM_PEAK_GBPS}
    out = {"workload": f"{n_modes} modes x Fock cutoff d={d} ({16 * d ** n_modes / 2**30:.0f} GiB), {len(seq)} gates: "
                       f"S(r) on mode k mod {n_modes} alternating with BS(i, i+1, pi/4) (BASELINE.json configs[3])",
           "gate_apps_per_sec": steps * len(seq) / dt, "ms_per_step": 1e3 * dt / steps,
           "algorithmic_bytes_per_gate": gbytes * 1e9, "norm2": st.reg.norm2(),
           "S": cls("S"), "S_on_last_mode": cls("S", lambda i: i == n_modes - 1),
           "BS": cls("BS"), "BS_interior_pairs": cls("BS", lambda i: i < n_modes - 2),
           "BS_last_pair": cls("BS", lambda i: i == n_modes - 2)}
    # full-size parity, outside the timed region: one S on every mode and one BS on every pair of the 16 GiB register,
    # sampled fibres / planes against U @ in computed from their own inputs (quantum_computations_amd.cv_simulator.fock)
    spot = fock.spot_check_register(st.reg, np.random.default_rng(32))
    out["full_size_spot_check"] = spot
    out["max_abs_err"] = max(spot["S_max_abs_err"], spot["BS_max_abs_err"])
    del st
    return out


def run_cfg5_single_gpu(n=30, iterations=2):
    """The one-GPU form of BASELINE.json configs[4]: Grover search on n = 30 qubits (16 GiB), success probability
    against sin^2((2k + 1) asin 2^(-n/2)).  (The configuration itself shards n = 30 over 8 GPUs: --config cfg5.)"""
    from quantum_computations_amd import workloads as W
    from quantum_computations_amd.device import DeviceState
    from quantum_computations_amd.dv_simulator import gates as G

    marked = (0b1011001110001111 << max(0, n - 16)) % (1 << n) | 1
    dev = DeviceState.zeros(n)
    h = G.H(0).matrix
    for q in range(n):
        dev.apply_matrix(h, [q])
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(iterations):
        W.grover_iteration(dev, n, marked)
    dev.sync()
    dt = time.perf_counter() - t0
    p = float(dev.probabilities([marked])[0])
    want = W.grover_success_probability(n, iterations)
    gates = iterations * W.grover_gate_count(n, marked)
    out = {"workload": f"Grover search, n={n} ({16 * (1 << n) / 2**30:.0f} GiB on one GPU), {iterations} iterations, "
                       f"{gates} gates (BASELINE.json configs[4] shards this register over 8 GPUs)",
           "gate_apps_per_sec": gates / dt, "equiv_28_qubit_gate_apps_per_sec": gates / dt * 2.0 ** (n - 28),
           "algorithmic_GBps": 2 * 16 * (1 << n) * gates / dt / 1e9, "success_probability": p, "analytic": want,
           "rel_err": abs(p - want) / want, "norm2": dev.norm2()}
    dev.close()
    return out


# ---- circuits of the sharded configurations -------------------------------------------------------------------------
def remote_cx_pairs(n, g):
    """32 CX gates whose (control, target) cycle through global->local, local->global and global->global pairs."""
    glob, loc = list(range(max(g, 1))), list(range(max(g, 1), n))
    pairs = []
    for i in range(32):
        kind = i % 3
        if kind == 0:
            pairs.append((glob[i % len(glob)], loc[(7 * i) % len(loc)]))
        elif kind == 1 or len(glob) < 2:
            pairs.append((loc[(5 * i) % len(loc)], glob[i % len(glob)]))
        else:
            pairs.append((glob[i % len(glob)], glob[(i + 1) % len(glob)]))
    return pairs


def fail_rank(rank, where, dev, exc):
    """A collective failed or timed out: report which step of which rank, then leave with a non-zero status from this
    fresh child process (torchrun tears the other ranks down).  Nothing is re-executed or retried."""
    stats = ""
    if hasattr(dev, "exchanges"):
        stats = (f"; exchange steps done {dev.exchanges}, messages sent {dev.messages}, "
                 f"last exchange: {getattr(dev, 'last_exchange', None)}")
    sys.stderr.write(f"bench.py: rank {rank} failed in {where}: {type(exc).__name__}: {exc}{stats}\n")
    sys.stderr.flush()
    os._exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--config", choices=("cfg2", "cfg3", "cfg4", "cfg5"), default="cfg2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fusion", action="store_true", help="skip the informational fused-circuit measurement")
    ap.add_argument("--no-secondary", action="store_true", help="skip config 4 and the extra CPU baselines (N = 1)")
    ap.add_argument("--qubits-per-gpu", type=int, default=0, help="weak scaling: register qubits per GPU")
    ap.add_argument("--rehearse", action="store_true",
                    help="development aid: no GPU -- ranks rendezvous over gloo and only compute the exchange schedule")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))      # nothing below runs in the parent: no torch import, no HIP call

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    g_bits = (world - 1).bit_length()
    if 1 << g_bits != world:
        raise SystemExit("the register shards over a power-of-two number of GPUs")
    if args.config == "cfg4":
        if world != 1:
            raise SystemExit("config 4 is a one-GPU configuration")
        return main_cfg4(args)
    per_gpu = args.qubits_per_gpu or QUBITS_PER_GPU[args.config]
    if args.scaling == "strong":
        n, n_local = per_gpu, per_gpu - g_bits
    else:
        n, n_local = per_gpu + g_bits, per_gpu

    from quantum_computations_amd import workloads as W
    from quantum_computations_amd.dv_simulator import gates as G

    # ---- the step of this configuration -----------------------------------------------------------------------------
    ops = None
    if args.config == "cfg2":
        ops = W.random_circuit(n, DEPTH, CIRCUIT_SEED)
        gates = W.to_gates(ops)
        gates_per_step = DEPTH
        workload = (f"{n}-qubit complex128 state vector ({n_local} qubits = {16 * (1 << n_local) / 2**30:.3g} GiB per "
                    f"GPU), depth-{DEPTH} random 1+2-qubit gates (BASELINE.json configs[1], seed {CIRCUIT_SEED})")
    elif args.config == "cfg3":
        gates = [G.CX(c, t) for c, t in remote_cx_pairs(n, g_bits)]
        gates_per_step = len(gates)
        workload = (f"{n}-qubit complex128 state vector sharded over {world} GPUs ({16 * (1 << n_local) / 2**30:.3g} GiB "
                    f"each), 32 CX gates cycling global->local, local->global, global->global (BASELINE.json configs[2])")
    else:
        marked = (0b1011001110001111 << max(0, n - 16)) % (1 << n) | 1
        iterations = 8
        gates = W.grover_circuit(n, marked, iterations)
        gates_per_step = len(gates)
        workload = (f"Grover search on {n} qubits over {world} GPU(s), {iterations} iterations per step, one marked "
                    f"item (BASELINE.json configs[4])")

    if args.rehearse:
        return rehearse(args, world, rank, n, gates, workload)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the gate path has no CPU fallback")
    # Development aid only (never set by the driver): QSV_BENCH_ONE_GPU=1 lets N ranks share GPU 0 with the
    # collectives staged through host memory over gloo (tests/host_staged.py), to rehearse the N > 1 code path on
    # a one-GPU box.  Its numbers are meaningless as a measurement.
    one_gpu_rehearsal = world > 1 and os.environ.get("QSV_BENCH_ONE_GPU") == "1"
    if one_gpu_rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    if world == 1:
        from quantum_computations_amd.device import DeviceState
        dev = DeviceState.random(n, seed=STATE_SEED, device=local_rank)
        barrier = lambda: None
        reduce_max = lambda x: x
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
            from datetime import timedelta
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                    timeout=timedelta(seconds=NCCL_TIMEOUT_S))
            dev = ShardedState.random(n, seed=STATE_SEED, device=local_rank)
            try:
                dev.warm_up_links()              # communicator and per-peer channel set-up stays out of the timing
            except Exception as exc:             # first contact with the other GPUs: say where it broke, exit non-zero
                fail_rank(rank, "link warm-up (one tiny send/recv with every peer + one all-reduce)", dev, exc)
            reduce_device = "cuda"
        barrier = dist.barrier

        def reduce_max(x):
            t = torch.tensor([x], dtype=torch.float64, device=reduce_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

    cpu, cpu_gates, cpu_ket, parity_err = None, 0, None, None
    if args.config == "cfg5":
        h = G.H(0).matrix
        dev.set_basis(0)
        for q in range(n):
            dev.apply_matrix(h, [q])

    trace = []                                   # N > 1, rank 0: (kernel, launch was purely local) per recorded gate
    passes = [0]                                 # passes of the circuit applied so far (Grover: iterations / 8)

    def step(record: bool, slot0: int = 0):
        slot = [slot0]

        def apply(gate):
            if record:
                before = (dev.exchanges, dev.local_swaps) if world > 1 else None
                dev.event_record(slot[0])
            gate.apply(dev)
            if record:
                dev.event_record(slot[0] + 1)
                if world > 1:                    # the shards' kernels depend on the layout: ask after every launch
                    trace.append((dev.last_kernel(), before == (dev.exchanges, dev.local_swaps)))
            slot[0] += 2

        if world > 1:
            # what Simulator.run does on a sharded register: announce the circuit (the shards plan which qubits to
            # give up) and take commuting gates local-first
            dev.run_circuit(gates, apply)
        else:
            for gate in gates:
                apply(gate)
        passes[0] += 1
        return slot[0]

    # which kernel each gate of the circuit lands in (asked of the library, not guessed)
    kernels = []
    if world == 1:
        for gate in gates:
            gate.apply(dev)
            kernels.append(dev.last_kernel())
        passes[0] += 1
    # the event ring holds 16384 marks
    recorded_steps = min(args.steps, 16000 // (2 * len(gates)))
    try:
        for _ in range(args.warmup):
            step(False)
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
    except Exception as exc:
        if world == 1:
            raise
        fail_rank(rank, f"pass {passes[0] + 1} of the {args.config} circuit", dev, exc)

    # The CPU leg and the full-size parity check run AFTER the timed region: 12 s of 16 OpenMP threads, 4 GiB host
    # arrays and their release in front of it left one launch in a few runs 18 ms long (whole-job value 892-908 instead
    # of 920: the default run differed from the same run with --no-cpu-baseline in nothing else).
    if args.config == "cfg2" and world == 1 and not args.no_cpu_baseline:
        dev.fill_random(STATE_SEED)             # back to the initial state: the CPU leg downloads it
        cpu, cpu_gates, cpu_ket = cpu_baseline(ops, dev, n)
        # parity at full size: run the same prefix on the GPU from the same initial state and compare every amplitude
        for gate in gates[:cpu_gates]:
            gate.apply(dev)
        parity_err = float(np.max(np.abs(dev.to_numpy() - cpu_ket)))

    extra = {}
    if args.config == "cfg5":                    # all ranks take part in the read-out
        total_iterations = passes[0] * iterations
        p = float(dev.probabilities([marked])[0])
        want = W.grover_success_probability(n, total_iterations)
        extra["grover"] = {"iterations_applied": total_iterations, "success_probability": p, "analytic": want,
                           "rel_err": abs(p - want) / want, "norm2": dev.norm2()}
    elif world > 1:
        extra["norm2"] = dev.norm2()

    if rank != 0:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return

    gate_apps = args.steps * gates_per_step
    register_rate = gate_apps / max(elapsed, 1e-12)            # gate-apps/s on the n-qubit register
    value = register_rate * 2.0 ** (n - 28)                    # 28-qubit equivalents, whole job
    bytes_per_gate_per_gpu = 2 * 16 * (1 << n_local)
    result = {
        "metric": "gate_apps_per_sec", "value": value, "unit": "gate-apps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, args.steps),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": workload, "name": args.config, "n_qubits": n, "gates_per_step": gates_per_step,
            **({"gate_mix": {c: [o["name"] for o in ops].count(c) for c in sorted({o["name"] for o in ops})}}
               if ops else {}),
            "sharding": f"top {g_bits} qubits = rank id" if g_bits else "single GPU",
            "unit_note": "value = gate-apps/s on the full register x 2^(n-28): 28-qubit gate-app equivalents",
        },
        "gate_apps_per_sec_on_register": register_rate,
        # SURVEY.md 8d credits every gate-app with the full pass B = 2 x 16 x 2^n whatever the kernel moves (CX and
        # SWAP touch half the register, CZ a quarter), so this figure may exceed the HBM peak; the bytes the circuit's
        # kernels really move are given beside it
        "full_pass_equivalent_GBps_per_gpu": bytes_per_gate_per_gpu * register_rate / 1e9,
        **extra,
    }
    if ops and world == 1:
        moved = sum(gate_class_and_regime(o, n)[2] for o in ops) * bytes_per_gate_per_gpu
        result["bytes_moved_per_step"] = moved
        result["moved_GBps_per_gpu"] = moved * args.steps / max(elapsed, 1e-12) / 1e9
    if world > 1:
        n_passes = max(1, args.steps + args.warmup)
        result["exchange"] = {
            "scheme": ("all rank bits swapped with the farthest-next-use local qubits in one all-to-all (grouped "
                       "send/recv over every link)" if dev.policy == "auto" and g_bits >= 2
                       else "pairwise half-shard send/recv"),
            "steps_per_circuit": dev.exchanges / n_passes, "rank_bits_swapped_per_circuit": dev.qubits_exchanged / n_passes,
            "local_line_up_passes_per_circuit": dev.local_swaps / n_passes,
            "GiB_sent_per_rank_per_circuit": dev.bytes_sent / n_passes / 2**30,
            "GiB_on_busiest_link_per_circuit": dev.link_bytes / n_passes / 2**30,
            "messages_per_circuit": dev.messages / n_passes, "piece_GiB": min(dev.chunk_amps, 1 << n_local) * 16 / 2**30,
            # gates applied slice by slice to landed data while the next slice was on the links (distributed.py)
            "gates_inside_exchanges_per_circuit": dev.gates_in_exchanges / n_passes,
            "launches_inside_exchanges_per_circuit": dev.rider_launches / n_passes,
            "hardware_status": "first contact with RCCL across GPUs: rehearsed on gloo / one GPU, unmeasured before this run"}
    if recorded_steps:
        per_kernel = {}
        if world == 1:
            for s in range(recorded_steps):
                for i, c in enumerate(kernels):
                    a = 2 * (s * len(gates) + i)
                    per_kernel.setdefault(c, []).append(dev.event_elapsed_ms(a, a + 1))
        else:       # this rank's launches that moved nothing between GPUs (SWAPs relabel the map and launch nothing)
            last = None
            for i, (c, local) in enumerate(trace):
                if local and c and not (c == last and dev.event_elapsed_ms(2 * i, 2 * i + 1) < 1e-3):
                    per_kernel.setdefault(c, []).append(dev.event_elapsed_ms(2 * i, 2 * i + 1))
                last = c
    if recorded_steps and per_kernel:       # (a rank whose every gate moved data has no purely local launch to report)
        # dominant kernel = the full-traffic dense instantiation with the most device time
        full = {k: v for k, v in per_kernel.items() if k.startswith(("k_dense<", "k_dense_tile12<"))} or per_kernel
        dominant = max(full, key=lambda k: sum(full[k]))
        avg_ms = float(np.mean(full[dominant]))
        achieved = bytes_per_gate_per_gpu / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        pmc = REPO / "profiles" / "pmc_traffic.json"        # written by tools/summarize_profile.py from rocprofv3 --pmc
        if pmc.exists() and n_local == PMC_QUBITS:          # a stored count only describes the size it was taken at
            entry = json.loads(pmc.read_text()).get(dominant)
            if entry:
                traffic, traffic_src = entry["hbm_bytes_per_launch"], entry["source"]
        copy = measure_copy_bandwidth(dev) if world == 1 else None
        result["roofline"] = {
            **({"scope": "rank 0's launches of its dominant kernel, exchanges excluded"} if world > 1 else {}),
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": dominant, "algorithmic_bytes_per_launch": bytes_per_gate_per_gpu,
            "avg_launch_ms": avg_ms, "launches_timed": len(full[dominant]),
            **({"measured_copy_GBps": copy["GBps"], "frac_of_measured_copy": achieved / copy["GBps"],
                "measured_copy": copy} if copy else {}),
        }
        if world == 1 and ops:
            # per-gate HIP-event times bucketed by stride regime x gate class
            buckets = {}
            for st_i in range(recorded_steps):
                for i, o in enumerate(ops):
                    cls, regime, moved = gate_class_and_regime(o, n)
                    a = 2 * (st_i * len(gates) + i)
                    buckets.setdefault((regime, cls), []).append((dev.event_elapsed_ms(a, a + 1), moved))
            result["per_regime_ms"] = {
                "regimes": "low: a target on bits 0-5 (inside a wavefront); high: a target on bits >= 20 (strides >= 16 "
                           "MiB); mid: the rest.  classes: dense (full pass), diag (CZ: a quarter of the register), "
                           "perm (CX, SWAP: half)",
                **{f"{regime}/{cls}": {"launches": len(v), "avg_ms": float(np.mean([t for t, _ in v])),
                                        "moved_GBps": float(np.mean([m for _, m in v])) * bytes_per_gate_per_gpu
                                                      / (np.mean([t for t, _ in v]) * 1e-3) / 1e9}
                   for (regime, cls), v in sorted(buckets.items())}}
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
    if args.config == "cfg2" and world == 1 and not args.no_fusion:
        # Informational, outside the timed region and NOT part of `value`: the same circuit through the gate-fusion
        # scheduler of Simulator(fuse=k) (quantum_computations_amd/fusion.py) -- fewer, denser launches.
        from quantum_computations_amd.fusion import fuse_circuit, fusion_stats
        result["with_gate_fusion"] = {}
        for k in (3, 4, 5, 6):
            fused = fuse_circuit(gates, k, n_qubits=n)
            for gate in fused:
                gate.apply(dev)
            dev.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                for gate in fused:
                    gate.apply(dev)
            dev.sync()
            dt = time.perf_counter() - t0
            result["with_gate_fusion"][f"max_{k}_qubits"] = {
                **fusion_stats(gates, fused), "gate_apps_per_sec": args.steps * DEPTH / dt,
                "ms_per_step": 1e3 * dt / args.steps}
            if cpu_ket is not None:
                # full-size parity of the fused path against the CPU oracle (not against the unfused HIP run): the
                # prefix the CPU leg ran, fused on its own, from the same initial state, all 2^n amplitudes
                dev.fill_random(STATE_SEED)
                for gate in fuse_circuit(gates[:cpu_gates], k, n_qubits=n):
                    gate.apply(dev)
                result["with_gate_fusion"][f"max_{k}_qubits"]["parity_max_abs_err_vs_cpu_at_full_size"] = \
                    float(np.max(np.abs(dev.to_numpy() - cpu_ket)))
    if args.config == "cfg2" and world == 1 and not args.no_secondary:
        dev.close()
        del cpu_ket
        secondary = {"cfg4": run_cfg4(), "cfg5_single_gpu": run_cfg5_single_gpu()}
        if not args.no_cpu_baseline:
            secondary["cpu_numpy_restatement_n28"] = numpy_restatement_baseline(ops, n, STATE_SEED)
            secondary["cpu_literal_dense_algorithm_n12"] = literal_dense_baseline()
        result["secondary"] = secondary
    print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def main_cfg4(args):
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the gate path has no CPU fallback")
    r = run_cfg4(steps=max(1, args.steps))
    gb = r["algorithmic_bytes_per_gate"]
    result = {
        "metric": "gate_apps_per_sec", "value": r["gate_apps_per_sec"] * gb / (2 * 16 * 2.0 ** 28),
        "unit": "gate-apps/s", "n_gpus": 1, "steps": args.steps, "warmup": 1, "ms_per_step": r["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": r["workload"], "name": "cfg4",
                   "unit_note": "value = mode-gate-apps/s x (32 GiB / 8 GiB): 28-qubit gate-app equivalents"},
        "gate_apps_per_sec_on_register": r["gate_apps_per_sec"],
        "roofline": {"bound": "hbm", "achieved": r["BS"]["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": r["BS"]["frac_of_peak"], "traffic": None, "kernel": "k_mode2_blocks (all pairs)",
                     "algorithmic_bytes_per_launch": gb, "avg_launch_ms": r["BS"]["avg_ms"]},
        "detail": r,
    }
    print(json.dumps(result), flush=True)


def rehearse(args, world, rank, n, gates, workload):
    """No GPU: the ranks rendezvous (gloo), every rank computes the data-free exchange schedule of the step, the
    schedules are compared, and rank 0 prints a line with ``value`` null.  Checks the launch plumbing on a CPU box."""
    import torch
    import torch.distributed as dist

    from quantum_computations_amd.distributed import ShardedState

    if world > 1:
        dist.init_process_group("gloo")
    plan = ShardedState.plan_only(n, world)
    plan.run_circuit(gates)
    mine = torch.tensor([plan.exchanges, plan.qubits_exchanged, plan.bytes_sent], dtype=torch.int64)
    agree = True
    if world > 1:
        everyone = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        agree = all(torch.equal(mine, t) for t in everyone)
    if rank == 0:
        print(json.dumps({"metric": "gate_apps_per_sec", "value": None, "unit": "gate-apps/s", "n_gpus": world,
                          "rehearsal": "no GPU: rendezvous and exchange schedule only", "scaling": args.scaling,
                          "config": {"workload": workload, "name": args.config, "n_qubits": n},
                          "ranks_agree_on_schedule": agree, "exchange_steps_per_circuit": plan.exchanges,
                          "rank_bits_swapped_per_circuit": plan.qubits_exchanged,
                          "GiB_sent_per_rank_per_circuit": plan.bytes_sent / 2**30}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
