"""TEST HELPER: first contact of the shipped collectives with RCCL, as far as one GPU allows.

RCCL refuses two ranks on one device, so the world has ONE rank -- but ``ShardedState._p2p`` (a grouped ncclSend/ncclRecv
of (re, im) views of slices of the complex128 shard), ``_allreduce_sum`` and ``_broadcast`` are the shipped
code on HBM tensors over the real "nccl" backend: a send to self inside a group is legal in RCCL.  What this cannot
show is a second rank; the exchange logic across ranks is covered by the gloo tests and ``tests/host_staged.py``.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main() -> int:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29731")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    from quantum_computations_amd.distributed import ShardedState, _default_engine_factory

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    n = 16
    rng = np.random.default_rng(5)
    host = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    buf = torch.from_numpy(host).to("cuda:0")
    st = ShardedState(n, buf, _default_engine_factory(0))

    # (1) grouped send/recv of strided halves of the shard through RCCL, as _exchange_bits issues them
    half = 1 << (n - 1)
    src = st.buf[half:half + 4096]
    dst = torch.zeros(4096, dtype=torch.complex128, device="cuda:0")
    step = st._p2p([(0, src)], [(0, dst)])
    step.wait()
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), host[half:half + 4096]), "self send/recv through RCCL changed the amplitudes"

    # (2) two sends in one group (what a 4-rank subgroup step looks like from one member)
    d1 = torch.zeros(1024, dtype=torch.complex128, device="cuda:0")
    d2 = torch.zeros(1024, dtype=torch.complex128, device="cuda:0")
    step = st._p2p([(0, st.buf[:1024]), (0, st.buf[2048:3072])], [(0, d1), (0, d2)])
    step.wait()
    torch.cuda.synchronize()
    assert np.array_equal(d1.cpu().numpy(), host[:1024]) and np.array_equal(d2.cpu().numpy(), host[2048:3072])

    # (3) the scalar all-reduce, the piece broadcast and the read-out built on it
    assert st._allreduce_sum([1.5, -2.0]) == [1.5, -2.0]
    probe = st.buf[:4096].clone()
    st._broadcast(probe, 0)
    torch.cuda.synchronize()
    assert np.array_equal(probe.cpu().numpy(), host[:4096])
    assert np.array_equal(st.to_numpy(), host) and np.array_equal(st.to_numpy(root=0), host)

    # (4) a short circuit through the sharded front end on this one-rank world (no exchange, but every collective
    #     call site -- norm, measure_probs, agree_on_outcome -- runs on RCCL)
    from quantum_computations_amd.dv_simulator import gates as G
    for g in (G.H(0), G.CX(0, n - 1), G.H(n - 1), G.CZ(1, 2)):
        g.apply(st)
    p0, p1 = st.measure_probs(0, np.array([1.0, 0.0], complex), np.array([0.0, 1.0], complex))
    assert abs(p0 + p1 - st.norm2()) < 1e-9 * st.norm2()
    assert st.agree_on_outcome(1) == 1
    dist.barrier()
    dist.destroy_process_group()
    print("rccl self ok")
    return 0


if __name__ == "__main__":
    sys.exit(main())
