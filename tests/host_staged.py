"""TEST HELPER: the production sharded register with its three collectives staged through host memory over gloo.

RCCL refuses two ranks on one device, and the development box has one GPU.  This subclass lets several ranks share
that GPU -- HBM shards, HIP kernels, the real ``DeviceState`` engine -- by overriding only the collectives
(``_exchange``, ``_allreduce_sum``, ``_allgather_shards``); everything else is ``ShardedState`` as shipped.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from quantum_computations_amd.distributed import ShardedState


class HostStagedShardedState(ShardedState):
    def _exchange(self, send, recv, peer):
        send_h, recv_h = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        ops = [dist.P2POp(dist.isend, send_h, peer), dist.P2POp(dist.irecv, recv_h, peer)]
        for work in dist.batch_isend_irecv(ops):
            work.wait()
        recv.copy_(recv_h)

    def _allreduce_sum(self, values):
        t = torch.tensor(values, dtype=torch.float64)
        dist.all_reduce(t)
        return [float(v) for v in t]

    def _allgather_shards(self):
        mine = self.buf.cpu().contiguous()
        shards = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(shards, mine)
        return torch.cat(shards).numpy()
