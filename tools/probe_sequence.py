"""Fused 5-qubit blocks of the cfg2 circuit: dense product vs the sequence of their source gates, by work limit.

    python3 tools/probe_sequence.py [n_qubits] [steps] [max block qubits] [tile regions]

One JSON line per setting of QSV_OPT_SEQUENCE_WORK (0 = every block as its dense product): gate-apps/s of the fused
circuit, how many blocks went as sequences, and the mean time per block of each kernel family (host clock around every
block, synchronised, in a second pass)."""
import json
import sys
import time
from collections import defaultdict
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from quantum_computations_amd import _lib, workloads as W   # noqa: E402
from quantum_computations_amd.device import DeviceState     # noqa: E402
from quantum_computations_amd.fusion import fuse_circuit    # noqa: E402


def main() -> None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    depth = 100
    gates = W.to_gates(W.random_circuit(n, depth, 100))
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    fused = fuse_circuit(gates, k, n_qubits=n)
    work = [sum(256 if len(g.indices) == 1 else 512 for g in getattr(b, "sources", [])) for b in fused]
    print(json.dumps({"n": n, "blocks": len(fused), "block_qubits": [len(b.indices) for b in fused], "work": work,
                      "gates_per_block": [len(getattr(b, "sources", [b])) for b in fused]}), flush=True)
    dev = DeviceState.zeros(n)
    dev.fill_random(28)
    if len(sys.argv) > 4:
        dev.set_option(_lib.OPT_TILE_REGIONS, int(sys.argv[4]))       # tile order of every tile kernel (-1 = built-in rules)
    settings = [("work", w) for w in (0, 2048, 3072, 1 << 20)] if k == 5 else [("work", 0)]
    settings += [("tile_gates", g) for g in (4, 6, 8, 10, 12, 16, 48)]
    for kind, limit in settings:
        dev.set_option(_lib.OPT_SEQUENCE_WORK, limit if kind == "work" else 0)
        dev.set_option(_lib.OPT_TILE_SEQUENCE_GATES, limit if kind == "tile_gates" else 0)
        for b in fused:
            b.apply(dev)
        dev.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            for b in fused:
                b.apply(dev)
        dev.sync()
        dt = time.perf_counter() - t0
        per = defaultdict(list)
        for b in fused:
            dev.sync()
            t1 = time.perf_counter()
            b.apply(dev)
            dev.sync()
            name = dev.last_kernel().split("<")[0] + ("/k%d" % len(b.indices))
            per[name].append(1e3 * (time.perf_counter() - t1))
        print(json.dumps({kind: limit, "gate_apps_per_sec": steps * depth / dt, "ms_per_step": 1e3 * dt / steps,
                          "blocks_by_kernel": {k: [len(v), round(float(np.mean(v)), 3)] for k, v in sorted(per.items())}}),
              flush=True)
    dev.close()


if __name__ == "__main__":
    main()
