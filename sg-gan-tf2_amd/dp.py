"""Data-parallel gradient exchange for the train step (new capability; the reference is single-process).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  Each
network keeps ONE flat f32 gradient buffer, so the exchange is one all-reduce(sum) per network bucket,
launched asynchronously as soon as that network's backward has been queued; the 1/world factor is applied
inside the fused Adam kernel (``grad_scale``).  InstanceNorm is per-sample and every loss is a batch mean,
so the result equals a single-device step on the concatenated batch.
"""
from __future__ import annotations

import torch.distributed as dist


class GradExchange:
    def __init__(self, process_group=None):
        self.group = process_group if process_group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.group)
        self.grad_scale = 1.0 / self.world

    def broadcast_(self, flat_params, src_rank_in_group=0):
        """Make replicas identical: copy rank 0's flat parameter buffer to every rank."""
        src = dist.get_global_rank(self.group, src_rank_in_group)
        dist.broadcast(flat_params, src=src, group=self.group)

    def allreduce_async(self, flat_grad):
        """Sum one network's gradient bucket over ranks; returns a handle whose wait() orders the current stream."""
        return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
