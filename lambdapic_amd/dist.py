"""Slab ring communicator: nearest-neighbour exchange between the x-slabs of a 1-D decomposition.

Replaces the reference's ``MPIManager2D`` point-to-point traffic (`core/mpi/mpi_manager.py:96-298`:
Isend/Irecv per (patch, boundary, attribute) on three duplicated communicators) by ONE fused
message per face and exchange, issued as a grouped send/recv pair set through
``torch.distributed.batch_isend_irecv`` -- on ROCm the ``nccl`` backend is RCCL, so each call is an
``ncclGroupStart .. ncclSend/ncclRecv .. ncclGroupEnd`` over the xGMI links to the two neighbours.
The path has no collective: every step is nearest neighbour only (as in the reference, which never
all-reduces inside the step).  With one rank nothing is sent (the engine wraps locally).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class SlabComm:
    def __init__(self, group=None, periodic=True):
        if dist.is_available() and dist.is_initialized():
            self.group = group
            self.rank = dist.get_rank(group)
            self.size = dist.get_world_size(group)
        else:
            self.group, self.rank, self.size = None, 0, 1
        self.periodic = periodic
        self.left = (self.rank - 1) % self.size
        self.right = (self.rank + 1) % self.size

    def exchange(self, send_lo, send_hi, recv_lo, recv_hi, wait=True):
        """send_lo -> left neighbour, send_hi -> right neighbour; recv_lo <- left, recv_hi <- right.

        Posting order matters when left == right (two ranks): messages between one pair of ranks
        match in posting order, so the sends are posted (hi, lo) and the receives (lo, hi): the
        peer's first receive (its low face) takes my high face.
        """
        if self.size == 1:
            recv_lo.copy_(send_hi)   # my own high edge is my low guard's periodic source
            recv_hi.copy_(send_lo)
            return []
        ops = [
            dist.P2POp(dist.isend, send_hi, self.right, self.group),
            dist.P2POp(dist.isend, send_lo, self.left, self.group),
            dist.P2POp(dist.irecv, recv_lo, self.left, self.group),
            dist.P2POp(dist.irecv, recv_hi, self.right, self.group),
        ]
        reqs = dist.batch_isend_irecv(ops)
        if wait:
            for r in reqs:
                r.wait()
        return reqs

    def allreduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        """diagnostics only (never inside the step)"""
        if self.size > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def barrier(self):
        if self.size > 1:
            dist.barrier(group=self.group)
