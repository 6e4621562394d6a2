"""TEST HELPER: the production sharded register with its three collectives staged through host memory over gloo.

RCCL refuses two ranks on one device, and the development box has one GPU.  This subclass lets several ranks share
that GPU -- HBM shards, HIP kernels, the real ``DeviceState`` engine -- by overriding only the collectives
(``_p2p``, ``_allreduce_sum``, ``_broadcast``); everything else -- the slicing of the shard into pieces, the two-slice staging buffer, the copies into place, all on
device tensors -- is ``ShardedState`` as shipped.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from quantum_computations_amd.distributed import ShardedState


class HostStagedShardedState(ShardedState):
    def _p2p(self, sends, recvs):
        send_h = [(peer, t.cpu()) for peer, t in sends]
        recv_h = [(peer, torch.empty(t.shape, dtype=t.dtype)) for peer, t in recvs]
        ops = [dist.P2POp(dist.isend, torch.view_as_real(t), peer) for peer, t in send_h]
        ops += [dist.P2POp(dist.irecv, torch.view_as_real(t), peer) for peer, t in recv_h]
        works = dist.batch_isend_irecv(ops)

        class _Step:
            @staticmethod
            def wait():
                for w in works:
                    w.wait()
                for (_, dev_t), (_, host_t) in zip(recvs, recv_h):
                    dev_t.copy_(host_t)
        return _Step

    def _allreduce_sum(self, values):
        t = torch.tensor(values, dtype=torch.float64)
        dist.all_reduce(t)
        return [float(v) for v in t]

    def _broadcast(self, tensor, src):
        host = tensor.cpu().contiguous()
        dist.broadcast(torch.view_as_real(host), src)
        if self.rank != src:
            tensor.copy_(host)
