"""Slab ring communicator: nearest-neighbour exchange between the x-slabs of a 1-D decomposition.

Replaces the reference's ``MPIManager2D`` point-to-point traffic (`core/mpi/mpi_manager.py:96-298`:
Isend/Irecv per (patch, boundary, attribute) on three duplicated communicators) by ONE fused
message per face and exchange, issued as a grouped send/recv pair set through
``torch.distributed.batch_isend_irecv`` -- on ROCm the ``nccl`` backend is RCCL, so each call is an
``ncclGroupStart .. ncclSend/ncclRecv .. ncclGroupEnd`` over the xGMI links to the two neighbours.
The path has no collective: every step is nearest neighbour only (as in the reference, which never
all-reduces inside the step).  With one rank nothing is sent (the engine wraps locally).

NATIVE TRANSPORT (``SlabComm.attach_rccl`` / ``LoopbackComm``): the face messages travel through the library's own
communicator (csrc/lpa_comm.hip: ncclSend / ncclRecv groups issued from C on the step's stream) and the engines enqueue a
whole slab step -- kernels AND exchanges -- with one ``lpa_step`` call; ``torch.distributed`` then only carries the control
plane (id broadcast, barrier, diagnostics).  Without it (gloo rehearsals, ranks sharing a GPU in the tests) the
exchanges below run from Python between sub-ranges of the step.
"""
from __future__ import annotations

import ctypes as C

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
        self.native = None          # lpa_comm* of the library's own transport (attach_rccl / LoopbackComm)
        self.native_kind = None
        self.periodic = periodic
        # ring (periodic x) or chain (open / PML x edges: the end ranks have one neighbour)
        self.left = (self.rank - 1) % self.size if (periodic or self.rank > 0) else -1
        self.right = (self.rank + 1) % self.size if (periodic or self.rank < self.size - 1) else -1

    def __getstate__(self):
        """process groups do not pickle (the reference drops and re-Dups its communicators the same way,
        `core/mpi/mpi_manager.py:35-46`): a restored communicator binds to the default group of the process that
        loads it; ``rebind`` attaches other groups (e.g. an RCCL group for the face messages)"""
        st = self.__dict__.copy()
        st["group"] = st["p2p_group"] = st["native"] = None
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
        if self.native is not None:
            self._native_exchange([(send_lo, send_hi, recv_lo, recv_hi)])
            return []
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
        if self.native is not None:
            self._native_exchange(sets)
            return []
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

    # ---- the library's own transport --------------------------------------------------------------------------------
    def attach_rccl(self, librccl=None):
        """face messages through RCCL calls issued by the library (csrc/lpa_comm.hip): rank 0 makes the 128-byte id, the
        control group broadcasts it, every rank joins -- collective over ``group``.  The calling thread's current device is
        the one the communicator binds (``torch.cuda.set_device`` first).  ``librccl``: path of the RCCL library to dlopen
        (default: the one this process has loaded already -- PyTorch's)."""
        from ._lib import LPA_COMM_RCCL, check, lib
        L = lib()
        path = librccl.encode() if librccl else None
        idbuf = (C.c_char * 128)()
        if self.rank == 0:
            check(L.lpa_comm_unique_id(idbuf, path), "lpa_comm_unique_id")
        if self.size > 1:
            t = torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8).clone()
            if dist.get_backend(self.group) != "gloo":
                t = t.cuda()
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            idbuf = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
        h = C.c_void_p()
        check(L.lpa_comm_create_rccl(C.byref(h), idbuf, self.rank, self.size, int(self.periodic), path),
              "lpa_comm_create_rccl")
        self.native, self.native_kind = h, LPA_COMM_RCCL
        return self

    def native_info(self):
        """(kind, rank, size, left, right, transport library version) of the native communicator"""
        from ._lib import check, lib
        info = (C.c_int32 * 6)()
        check(lib().lpa_comm_info(self.native, info), "lpa_comm_info")
        return tuple(info)

    def close(self):
        if self.native is not None:
            from ._lib import lib
            lib().lpa_comm_destroy(self.native)
            self.native = None

    def _native_exchange(self, sets):
        from ._lib import check, lib, lpa_face_msg
        arr = (lpa_face_msg * len(sets))()
        dev = None
        for k, tensors in enumerate(sets):
            for t in tensors:
                if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
                    raise ValueError("native face messages are contiguous float64 device tensors")
                dev = t.device
            s_lo, s_hi, r_lo, r_hi = tensors
            m = arr[k]
            m.send_lo, m.send_hi, m.recv_lo, m.recv_hi = s_lo.data_ptr(), s_hi.data_ptr(), r_lo.data_ptr(), r_hi.data_ptr()
            m.n_send_lo, m.n_send_hi, m.n_recv_lo, m.n_recv_hi = s_lo.numel(), s_hi.numel(), r_lo.numel(), r_hi.numel()
        check(lib().lpa_comm_exchange(self.native, arr, len(sets), torch.cuda.current_stream(dev).cuda_stream),
              "lpa_comm_exchange")

    def arrival_shift(self, box_length):
        """what is added to x of the particles arriving through (my low face, my high face): the periodic wrap at the two
        ends of the box (`core/patch/sync_particles_2d.c:168-182`)"""
        lo = -box_length if (self.rank == 0 and self.periodic) else 0.0
        hi = box_length if (self.rank == self.size - 1 and self.periodic) else 0.0
        return lo, hi

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

    def allmin(self, value):
        """minimum over the ranks of a number, or element-wise of a list of numbers (one small all-reduce on the control
        group; every rank must call it in the same step)"""
        vec = isinstance(value, (list, tuple))
        vals = [float(v) for v in value] if vec else [float(value)]
        if self.size > 1 and dist.is_initialized():       # (no process group: one process plays every slab -- tests, loopbacks)
            on_host = dist.get_backend(self.group) == "gloo"
            t = torch.tensor(vals, dtype=torch.float64, device="cpu" if on_host else torch.cuda.current_device())
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            vals = t.tolist()
        return vals if vec else vals[0]

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


class LoopbackComm(SlabComm):
    """ONE process as rank 0 of a periodic ring of ``size`` (1 or 2) identical slabs: what a neighbour would send through
    a face is what this slab itself sends through the opposite one (``lpa_comm_create_loopback``; arriving particles are
    translated by one slab width).  Every kernel of the N > 1 path runs, only the wire is a device copy -- the compute-side
    cost of the slab decomposition on one GPU (tools/bench_mirror.py), and a test bed in which a slab must reproduce the
    corresponding half of a single-slab run of the doubled periodic box.  ``rccl=True``: the same ring through RCCL
    (a communicator of one rank sending to itself): the real ncclSend / ncclRecv path, launch costs included."""

    def __init__(self, slab_width, size=2, rccl=False):
        super().__init__(None, periodic=True, single=True)
        from ._lib import LPA_COMM_LOOPBACK, LPA_COMM_RCCL, check, lib
        L = lib()
        h = C.c_void_p()
        if rccl:
            idbuf = (C.c_char * 128)()
            check(L.lpa_comm_unique_id(idbuf, None), "lpa_comm_unique_id")
            check(L.lpa_comm_create_rccl(C.byref(h), idbuf, 0, 1, 1, None), "lpa_comm_create_rccl")
        else:
            check(L.lpa_comm_create_loopback(C.byref(h), int(size), 1), "lpa_comm_create_loopback")
        self.native, self.native_kind = h, (LPA_COMM_RCCL if rccl else LPA_COMM_LOOPBACK)
        self.size, self.rank = int(size), 0
        self.left = self.right = (1 if size == 2 else 0)
        self.slab_width = float(slab_width)

    def arrival_shift(self, box_length):
        # the neighbour is this slab's copy one slab width further: what left through my high face at xhi + d re-enters
        # through my low face at xlo + d
        return -self.slab_width, self.slab_width

    def barrier(self):
        pass

    def reduce_diagnostics(self, d):
        return d

    def any(self, flag):
        return bool(flag)

    def allmin(self, value):
        return [float(v) for v in value] if isinstance(value, (list, tuple)) else float(value)

    def allreduce_sum(self, t):
        return t


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


class MigrateWindowMixin:
    """How much of a particle face message travels.  The message is fixed-size (the receiver posts its receive before it
    can know the count, which rides in band): ``1 + LPA_MIG_NATTR * capacity`` doubles per species and face -- 2.4 MB at
    the default capacity of 32 768, 19 MB at the 262 144 a C5 slab is given, for a few thousand leavers per step.  On a
    wire that is the largest message of the step by far, so the engines send only a WINDOW of it: the first
    ``migrate_window`` slots (the SoA stride of the message is the window, so the used part is a prefix of the buffer).
    The window is retuned where the slab chain agrees on its sort clock anyway (``_tick_chain_clock``: every rank sorted in
    this step, one small all-reduce): MARGIN x the largest count any rank saw in its four headers at its sort (the host
    is synchronised there), a power of two, at least MIN, at most the capacity; it grows at once and shrinks by a factor of
    two per retune at most.  A step whose leavers exceed the window is not an error below the full capacity: a leaver that
    did not fit stays where it is (it deposits through the guard planes like any particle just outside the slab) and is
    offered again in the next step; the surplus counter makes every rank return to the full capacity at the next retune.
    (needs from the engine: ``migrate_capacity``, ``comm``)"""

    adaptive_migrate_window = True
    MIGRATE_WINDOW_MIN = 8192
    MIGRATE_WINDOW_MARGIN = 4
    _mig_window = None
    _mig_seen = 0
    _mig_grow = False
    _mig_forgive = False

    @property
    def migrate_window(self) -> int:
        w = self._mig_window
        return self.migrate_capacity if (w is None or not self.adaptive_migrate_window) else min(int(w), self.migrate_capacity)

    def _mig_views(self, m):
        """the travelling part of a species' four message buffers"""
        from ._lib import LPA_MIG_NATTR
        n = 1 + LPA_MIG_NATTR * self.migrate_window
        return m if n >= m["s_lo"].numel() else {k: v[:n] for k, v in m.items()}

    def _mig_sample(self, m):
        """at a sort (the host is synchronised): the counts of the last step's four messages"""
        if self.adaptive_migrate_window and self.comm.size > 1 and m is not None:
            h = torch.stack([m[k][0] for k in ("s_lo", "s_hi", "r_lo", "r_hi")]).view(torch.int64).tolist()
            self._mig_seen = max(self._mig_seen, *[int(v) for v in h])

    def _mig_surplus(self, surplus) -> bool:
        """leavers did not fit the window: True = handled (back to the full capacity at the next retune), False = the
        caller raises (they did not fit the full capacity).  The device counters run from sort to sort: what they hold
        after a retune has widened the window was counted against the narrower one (``_mig_forgive``: until the retune
        after the sorts that zero them)"""
        if surplus > 0 and self.adaptive_migrate_window:
            if self.migrate_window < self.migrate_capacity:
                if not self._mig_grow:
                    import warnings
                    warnings.warn(f"{surplus} leaver-steps did not fit the message window of {self.migrate_window} slots: they "
                                  f"wait outside the slab; the window returns to migrate_capacity={self.migrate_capacity} at "
                                  "the next sort", RuntimeWarning, stacklevel=3)
                self._mig_grow = True
                return True
            return self._mig_forgive
        return False

    def _mig_request(self):
        """what this rank asks the next window to hold (None: not adaptive)"""
        if not self.adaptive_migrate_window:
            return None
        seen = self.migrate_capacity if self._mig_grow else self._mig_seen
        self._mig_seen, self._mig_grow, self._mig_forgive = 0, False, False
        return float(seen)

    def _mig_apply(self, seen):
        """``seen``: the largest request over the ranks -- every rank computes the same window"""
        want = max(self.MIGRATE_WINDOW_MIN, int(self.MIGRATE_WINDOW_MARGIN * seen))
        want = min(1 << max(want - 1, 1).bit_length(), self.migrate_capacity)
        new = max(want, min(self.migrate_window // 2, self.migrate_capacity))
        if new > self.migrate_window:
            self._mig_forgive = True
        self._mig_window = new
