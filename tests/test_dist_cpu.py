"""The N > 1 path on CPU: world_size 2 and 3 `gloo` processes exercise the slab ring communicator
(posting order when both neighbours are the same rank, periodic ring, face pairing) with numpy
stand-ins for the device pack/unpack kernels (test infrastructure) on a global periodic array."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lambdapic_amd.dist import SlabComm, exchange_faces

NG = 3


def _free_port():
    # below the kernel's ephemeral range (32768-60999): an outgoing gloo connection of an earlier test cannot sit on it
    # (a port taken from bind(0) was, once in a few hundred launches, in use again by the time the store listened)
    import random
    for _ in range(200):
        p = random.randrange(20000, 30000)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", p))
        except OSError:
            continue
        finally:
            s.close()
        return p
    raise RuntimeError("no free port")


def _worker(rank, world, port, nx_loc, ny, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = SlabComm(None)
        assert (comm.left, comm.right) == ((rank - 1) % world, (rank + 1) % world)
        rng = np.random.default_rng(5)
        G = rng.normal(size=(world * nx_loc, ny))          # global periodic field (same on all ranks)
        NX, NY = nx_loc + 2 * NG, ny + 2 * NG
        slab = np.zeros((NX, NY))
        x0 = rank * nx_loc
        slab[NG:NG + nx_loc, NG:NG + ny] = G[x0:x0 + nx_loc]
        # local y wrap first (what lpa_guard_wrap does with axes = y), then the x faces incl. y guards
        slab[NG:NG + nx_loc, :NG] = slab[NG:NG + nx_loc, ny:ny + NG]
        slab[NG:NG + nx_loc, NG + ny:] = slab[NG:NG + nx_loc, NG:2 * NG]
        n = NG * NY
        bufs = {k: torch.zeros(n, dtype=torch.float64) for k in ("s_lo", "s_hi", "r_lo", "r_hi")}

        def pack(side, b):      # interior edge planes (lpa_halo_pack_guard_src)
            rows = slab[NG:2 * NG] if side == 0 else slab[nx_loc:nx_loc + NG]
            b.copy_(torch.from_numpy(rows.reshape(-1).copy()))

        def unpack(side, b):    # my guard planes (lpa_halo_unpack_guard)
            rows = b.numpy().reshape(NG, NY)
            if side == 0:
                slab[:NG] = rows
            else:
                slab[NG + nx_loc:] = rows

        exchange_faces(comm, pack, unpack, bufs)
        # expected: periodic padding of the global array
        Gp = np.pad(G, ((NG, NG), (NG, NG)), mode="wrap")
        exp = Gp[x0:x0 + NX]
        ok_guard = np.array_equal(slab, exp)

        # current fold: every rank's guard planes are added into the neighbour's interior edge
        J = np.arange(NX * NY, dtype=np.float64).reshape(NX, NY) + 1000.0 * rank
        J0 = J.copy()

        def packc(side, b):     # lpa_halo_pack_current: take my guard planes, zero them
            rows = J[:NG] if side == 0 else J[NG + nx_loc:]
            b.copy_(torch.from_numpy(rows.reshape(-1).copy()))
            rows[...] = 0.0

        def unpackc(side, b):   # lpa_halo_unpack_current: add into my interior edge
            rows = b.numpy().reshape(NG, NY)
            if side == 0:
                J[NG:2 * NG] += rows
            else:
                J[nx_loc:nx_loc + NG] += rows

        exchange_faces(comm, packc, unpackc, bufs)
        left, right = (rank - 1) % world, (rank + 1) % world
        base = np.arange(NX * NY, dtype=np.float64).reshape(NX, NY)
        exp_lo = J0[NG:2 * NG] + (base[NG + nx_loc:] + 1000.0 * left)       # left's HIGH guard
        exp_hi = J0[nx_loc:nx_loc + NG] + (base[:NG] + 1000.0 * right)      # right's LOW guard
        if nx_loc >= 2 * NG:
            ok_fold = np.array_equal(J[NG:2 * NG], exp_lo) and np.array_equal(J[nx_loc:nx_loc + NG], exp_hi) \
                and not J[:NG].any() and not J[NG + nx_loc:].any()
        else:
            ok_fold = True
        # several exchanges in one grouped round: set k carries 100 k + rank (lo face) / + 0.5 (hi face)
        ok_many = True
        bufs2 = [(torch.full((4 + k,), 100.0 * k + rank, dtype=torch.float64),
                  torch.full((4 + k,), 100.0 * k + rank + 0.5, dtype=torch.float64),
                  torch.zeros(4 + k, dtype=torch.float64), torch.zeros(4 + k, dtype=torch.float64)) for k in range(3)]
        comm.exchange_many(bufs2)
        lft, rgt = (rank - 1) % world, (rank + 1) % world
        for k, (_, _, r_lo, r_hi) in enumerate(bufs2):
            ok_many = ok_many and bool(torch.all(r_lo == 100.0 * k + lft + 0.5)) and \
                bool(torch.all(r_hi == 100.0 * k + rgt))
        # scalar diagnostics: one all-reduce over the ranks (floats, ints and per-species lists)
        d = comm.reduce_diagnostics({"field_energy": 1.5 + rank, "charge": -2.0, "kinetic": [0.25 * rank, 1.0],
                                     "nalive": [10 + rank, 7]})
        w = world
        ok_diag = d == {"field_energy": 1.5 * w + w * (w - 1) / 2, "charge": -2.0 * w,
                        "kinetic": [0.25 * w * (w - 1) / 2, 1.0 * w], "nalive": [10 * w + w * (w - 1) // 2, 7 * w]} \
            and all(isinstance(v, int) for v in d["nalive"])
        # the control-plane votes of the slab chain (rho.py): OR of a flag, minimum of the sort intervals asked for
        ok_votes = comm.any(rank == world - 1) and not comm.any(False) and comm.allmin(7.0 + rank) == 7.0 \
            and comm.allmin(20 - rank) == 20 - (world - 1)
        q.put((rank, bool(ok_guard), bool(ok_fold) and bool(ok_diag) and ok_many and bool(ok_votes)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_ring_exchange(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 8, 10, q)) for r in range(world)]
    for p in procs:
        p.daemon = True
        p.start()
    try:
        res = [q.get(timeout=120) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:          # a worker stuck in a collective must not outlive the test
            if p.is_alive():
                p.terminate()
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), "guard exchange mismatch"
    assert all(r[2] for r in res), "current fold mismatch"


def test_single_rank_exchange_is_local_wrap():
    comm = SlabComm(None, single=True)
    a, b = torch.arange(4.0), torch.arange(4.0) + 10
    ra, rb = torch.zeros(4), torch.zeros(4)
    comm.exchange(a, b, ra, rb)
    assert torch.equal(ra, b) and torch.equal(rb, a)
