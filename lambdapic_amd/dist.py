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

import numpy as np
import torch
import torch.distributed as dist


class SlabComm:
    def __init__(self, group=None, periodic=True, single=False, p2p_group=None):
        """``group``: control plane (barrier, diagnostics); ``p2p_group``: the group the face messages
        travel on (default: ``group``) -- e.g. a gloo default group for CPU-side control next to an RCCL
        group for the device-to-device halo traffic"""
        if not single and dist.is_available() and dist.is_initialized():
            self.group = group
            self.rank = dist.get_rank(group)
            self.size = dist.get_world_size(group)
        else:
            self.group, self.rank, self.size = None, 0, 1
        self.p2p_group = p2p_group if p2p_group is not None else self.group
        self.periodic = periodic
        # ring (periodic x) or chain (open / PML x edges: the end ranks have one neighbour)
        self.left = (self.rank - 1) % self.size if (periodic or self.rank > 0) else -1
        self.right = (self.rank + 1) % self.size if (periodic or self.rank < self.size - 1) else -1

    def __getstate__(self):
        """process groups do not pickle (the reference drops and re-Dups its communicators the same way,
        `core/mpi/mpi_manager.py:35-46`): a restored communicator binds to the default group of the process that
        loads it; ``rebind`` attaches other groups (e.g. an RCCL group for the face messages)"""
        st = self.__dict__.copy()
        st["group"] = st["p2p_group"] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        if self.size > 1:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError(f"checkpoint of rank {self.rank} of {self.size}: initialise torch.distributed "
                                   "with the same world size before loading it")
            if dist.get_world_size() != self.size or dist.get_rank() != self.rank:
                raise RuntimeError(f"checkpoint belongs to rank {self.rank} of {self.size}, this process is rank "
                                   f"{dist.get_rank()} of {dist.get_world_size()}")

    def rebind(self, group=None, p2p_group=None):
        self.group = group
        self.p2p_group = p2p_group if p2p_group is not None else group

    @property
    def has_left(self):
        return self.size > 1 and self.left >= 0

    @property
    def has_right(self):
        return self.size > 1 and self.right >= 0

    def exchange(self, send_lo, send_hi, recv_lo, recv_hi, wait=True):
        """send_lo -> left neighbour, send_hi -> right neighbour; recv_lo <- left, recv_hi <- right.

        PAIRING RULE (the only one RCCL offers: ncclSend / ncclRecv carry no tag; messages between one
        pair of ranks match in the order they were posted inside the group): every rank posts its
        sends as (hi, lo) and its receives as (lo, hi).  With three or more ranks the two neighbours
        differ and each pair exchanges one message per direction.  With two ranks left == right: the
        peer's first receive (its low face) takes my first send (my high face), its second receive
        (its high face) my second send (my low face).  No tags are passed, so the gloo rehearsals
        (which WOULD match by tag) exercise exactly this rule
        (the reference tags its messages instead: `core/mpi/sync_fields2d.c:577-578`).

        With the ``gloo`` backend (CPU rehearsal of the multi-rank path, or several ranks sharing
        one GPU in a test) device tensors are staged through host memory.
        """
        if self.size == 1:
            if self.periodic:
                recv_lo.copy_(send_hi)   # my own high edge is my low guard's periodic source
                recv_hi.copy_(send_lo)
            return []
        grp = self.p2p_group
        staged = send_lo.is_cuda and dist.get_backend(grp) == "gloo"
        if staged:
            s_lo, s_hi = send_lo.cpu(), send_hi.cpu()
            r_lo, r_hi = torch.empty_like(recv_lo, device="cpu"), torch.empty_like(recv_hi, device="cpu")
        else:
            s_lo, s_hi, r_lo, r_hi = send_lo, send_hi, recv_lo, recv_hi
        ops = []
        if self.has_right:
            ops.append(dist.P2POp(dist.isend, s_hi, self.right, grp))
        if self.has_left:
            ops.append(dist.P2POp(dist.isend, s_lo, self.left, grp))
            ops.append(dist.P2POp(dist.irecv, r_lo, self.left, grp))
        if self.has_right:
            ops.append(dist.P2POp(dist.irecv, r_hi, self.right, grp))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        if wait or staged:
            for r in reqs:
                r.wait()
            if staged:
                if self.has_left:
                    recv_lo.copy_(r_lo)
                if self.has_right:
                    recv_hi.copy_(r_hi)
        return reqs

    def exchange_many(self, sets):
        """several face exchanges in ONE grouped send / recv round; ``sets`` = [(send_lo, send_hi, recv_lo,
        recv_hi), ...].  Every rank lists its sets in the same order; per set the posting order is the one
        of ``exchange`` (sends hi, lo -- receives lo, hi), set after set, so messages between one pair of ranks
        still match in posting order when both neighbours are the same rank (no tags: see ``exchange``)."""
        if self.size == 1:
            for s_ in sets:
                self.exchange(*s_)
            return []
        grp = self.p2p_group
        staged = sets[0][0].is_cuda and dist.get_backend(grp) == "gloo"
        ops, back = [], []
        for k, (send_lo, send_hi, recv_lo, recv_hi) in enumerate(sets):
            if staged:
                s_lo, s_hi = send_lo.cpu(), send_hi.cpu()
                r_lo, r_hi = torch.empty_like(recv_lo, device="cpu"), torch.empty_like(recv_hi, device="cpu")
                back.append((recv_lo, r_lo, recv_hi, r_hi))
            else:
                s_lo, s_hi, r_lo, r_hi = send_lo, send_hi, recv_lo, recv_hi
            if self.has_right:
                ops.append(dist.P2POp(dist.isend, s_hi, self.right, grp))
            if self.has_left:
                ops.append(dist.P2POp(dist.isend, s_lo, self.left, grp))
                ops.append(dist.P2POp(dist.irecv, r_lo, self.left, grp))
            if self.has_right:
                ops.append(dist.P2POp(dist.irecv, r_hi, self.right, grp))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        for r in reqs:
            r.wait()
        for recv_lo, r_lo, recv_hi, r_hi in back:
            if self.has_left:
                recv_lo.copy_(r_lo)
            if self.has_right:
                recv_hi.copy_(r_hi)
        return reqs

    def allreduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        """diagnostics only (never inside the step): the reference's scalar reductions (energy, charge, live
        count); a device tensor is staged through the host when the control group is gloo"""
        if self.size > 1:
            if t.is_cuda and dist.get_backend(self.group) == "gloo":
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def any(self, flag: bool) -> bool:
        """logical OR over the ranks (one small all-reduce; every rank must call it in the same step)"""
        if self.size == 1:
            return bool(flag)
        on_host = dist.get_backend(self.group) == "gloo"
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64,
                         device="cpu" if on_host else torch.cuda.current_device())
        return float(self.allreduce_sum(t)[0]) > 0.0

    def allmin(self, value: float) -> float:
        """minimum over the ranks (one small all-reduce on the control group; every rank must call it in the same step)"""
        if self.size == 1:
            return float(value)
        on_host = dist.get_backend(self.group) == "gloo"
        t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if on_host else torch.cuda.current_device())
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return float(t[0])

    def reduce_diagnostics(self, d: dict) -> dict:
        """sum a diagnostics() dict (floats, ints and lists of them) over the ranks: one all-reduce"""
        if self.size == 1:
            return d
        keys, flat = [], []
        for k, v in d.items():
            vals = v if isinstance(v, list) else [v]
            keys.append((k, len(vals), isinstance(v, list), all(isinstance(x, (int, np.integer)) for x in vals)))
            flat += [float(x) for x in vals]
        on_host = dist.get_backend(self.group) == "gloo"
        t = torch.tensor(flat, dtype=torch.float64, device="cpu" if on_host else torch.cuda.current_device())
        t = self.allreduce_sum(t)
        out, i = {}, 0
        for k, n, is_list, is_int in keys:
            vals = [int(round(x)) if is_int else float(x) for x in t[i:i + n].tolist()]
            out[k] = vals if is_list else vals[0]
            i += n
        return out

    def barrier(self):
        if self.size > 1:
            dist.barrier(group=self.group)


def exchange_faces(comm: SlabComm, pack, unpack, bufs, pack2=None, unpack2=None):
    """One nearest-neighbour halo step of the slab decomposition, shared by guard copies, current
    folds and particle migration:

        pack(side, buf)    fills ``buf`` with what leaves through face ``side`` (0 = low x, 1 = high x)
        unpack(side, buf)  consumes what arrived through face ``side``

    What leaves my low face arrives at the LEFT neighbour's high face and vice versa (the ring is
    periodic, `core/patch/patch.py:446-507` neighbour tables for periodic x).  ``bufs`` is a dict
    with tensors ``s_lo s_hi r_lo r_hi``.  ``pack2(buf_lo, buf_hi)`` / ``unpack2(buf_lo | None, buf_hi | None)``
    do both faces in one call (one kernel launch instead of two) and replace ``pack`` / ``unpack`` when given.
    """
    single = comm.size == 1 and comm.periodic
    got_lo, got_hi = comm.has_left or single, comm.has_right or single   # nothing arrives through an open face
    # (nothing leaves through an open face either: its guard planes are left alone -- a packed current plane is
    # zeroed, and the open end of a chain must keep what was deposited there exactly like a single slab does)
    if pack2 is not None:
        pack2(bufs["s_lo"] if got_lo else None, bufs["s_hi"] if got_hi else None)
    else:
        if got_lo:
            pack(0, bufs["s_lo"])
        if got_hi:
            pack(1, bufs["s_hi"])
    comm.exchange(bufs["s_lo"], bufs["s_hi"], bufs["r_lo"], bufs["r_hi"])
    if unpack2 is not None:
        unpack2(bufs["r_lo"] if got_lo else None, bufs["r_hi"] if got_hi else None)
        return
    if got_lo:
        unpack(0, bufs["r_lo"])
    if got_hi:
        unpack(1, bufs["r_hi"])
