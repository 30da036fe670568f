"""Batch data parallelism for the MAC cell: one process per GPU, `torch.distributed` over RCCL/xGMI.

The reference builds one tower per GPU and slices the batch with `initTowerBatch`
(model.py:139-149) but never exchanges gradients -- `averageAcrossTowers` keeps tower 0
(model.py:671-679, "TODO (add back support for multi-gpu..)").  Here every question is independent
in the forward pass, so the only exchange is ONE all-reduce per step of a flat fp32
gradient buffer (cell only: 2.1 M + p * 0.26 M parameters = 21 MB at p = 12).  Each rank computes
the MEAN loss over its shard (model.py:596); shard gradients are combined weighted by shard size so
that the result equals the full-batch gradient.
"""
import torch
import torch.distributed as dist


def tower_slice(batch_size, rank, world):
    """model.py:139-149: floor(B/R) questions per tower, the last tower takes the remainder."""
    per = batch_size // world
    start = rank * per
    end = (rank + 1) * per if rank < world - 1 else batch_size
    return start, end


class GradBucket:
    """Flat gradient buffer reused across steps; one all-reduce (sum) per step."""

    def __init__(self, tensors):
        self.tensors = list(tensors)
        self.sizes = [t.numel() for t in self.tensors]
        n = sum(self.sizes)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.tensors[0].device)

    def allreduce_(self, shard_size, global_size, group=None):
        """grads <- sum_r (shard_r / global) * grads_r  == gradient of the full-batch mean loss."""
        w = float(shard_size) / float(global_size)
        grads = [(t.grad if t.grad is not None else torch.zeros_like(t)).reshape(-1) for t in self.tensors]
        torch.cat(grads, out=self.flat)                       # one gather kernel into the flat buffer
        self.flat.mul_(w)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for t, n in zip(self.tensors, self.sizes):            # gradients become views of the reduced buffer
            t.grad = self.flat[off:off + n].view_as(t)
            off += n
        return self.flat
