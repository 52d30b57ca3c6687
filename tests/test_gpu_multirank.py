"""The slab-decomposed engine with 2 and 3 ranks against the single-rank engine on the same global
problem.  The ranks share the box's one GPU and talk through `gloo` (device buffers staged through
the host by SlabComm) -- the RCCL path differs only in the transport of the same messages.

Checks per-step field energy / charge / kinetic energy / particle count (summed over ranks) and the
final E, B, rho arrays gathered from the slabs, tolerance 1e-10 (summation order only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

C = 299792458.0
NXG, NY, PPC, NSTEPS = 96, 64, 6, 24


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


def _problem():
    lam = 0.8e-6
    dx = dy = lam / 20
    dt = 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2))
    rng = np.random.default_rng(77)
    n = NXG * NY * PPC
    cell = np.arange(n) // PPC
    x = ((cell // NY) + rng.uniform(-0.5, 0.5, n)) * dx
    y = ((cell % NY) + rng.uniform(-0.5, 0.5, n)) * dy
    # hot plasma (u ~ 0.3): plenty of slab crossings within a few steps
    u = rng.normal(size=(3, n)) * 0.3
    ig = 1 / np.sqrt(1 + (u ** 2).sum(0))
    w = np.full(n, 1.7e27 * dx * dy / PPC)
    return dx, dy, dt, x, y, u, ig, w


def _run(rank, world, port, q, left_only=False, long_ahead=False):
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from lambdapic_amd.dist import SlabComm
    from lambdapic_amd.engine import PicEngine2D
    comm = SlabComm(None, single=(world == 1))
    dx, dy, dt, x, y, u, ig, w = _problem()
    nx = NXG // world
    eng = PicEngine2D(nx, NY, dx, dy, device="cuda:0", comm=comm, sort_interval=20 if long_ahead else 5, block_particles=1024,
                      migrate_capacity=4096)
    eng.overlap = world == 2       # 2 ranks: J / rho guard exchange behind the interior tiles; 3 ranks: in line
    if long_ahead:                 # the sort bins TEN steps ahead (fixed clock of 20): a relativistic particle is filed up to
        eng.overflow_sort_fraction = 0      # 6.7 cells from where it is -- most of a tile column
        u = u * 6.0
        ig = 1 / np.sqrt(1 + (u ** 2).sum(0))
    lo, hi = (rank * nx - 0.5) * dx, ((rank + 1) * nx - 0.5) * dx
    mine = (x >= lo) & (x < hi)
    if left_only:                   # plasma in the left 40 % of the box only: the last rank starts without a particle
        mine &= x < 0.4 * NXG * dx
    n = int(mine.sum())
    eng.add_species(-1.602176634e-19, 9.1093837139e-31, capacity=3 * n + 20000)
    s = eng.species[0].cset
    for name, arr in (("x", x), ("y", y), ("ux", u[0]), ("uy", u[1]), ("uz", u[2]), ("inv_gamma", ig), ("w", w)):
        s.arr(name)[:n] = torch.from_numpy(arr[mine]).cuda()
    s.id[:n] = torch.from_numpy(np.nonzero(mine)[0]).cuda()
    eng.species[0].n = n
    if left_only:                   # ... and a second species that has no particle anywhere
        eng.add_species(1.602176634e-19, 1836 * 9.1093837139e-31, capacity=20000)
    trace = []
    for _ in range(NSTEPS):
        eng.step(dt)
        d = eng.diagnostics()
        trace.append([d["field_energy"], d["charge"], d["kinetic"][0], d["nalive"][0]])
    g = eng.grid
    sl = slice(3, 3 + nx)
    fields = {a: g.view(a)[sl, 3:3 + NY].cpu().numpy() for a in ("ex", "ey", "ez", "bx", "by", "bz", "rho")}
    # the reference's DEFAULT guard sync moves all six components in one call (patch.py:670, mpi_manager.py:111):
    # scribble over the x guard planes, sync E and B together, and hand the padded slabs back for inspection
    eb = ("ex", "ey", "ez", "bx", "by", "bz")
    for a in eb:
        g.view(a)[:3] = 777.0
        g.view(a)[3 + nx:] = -777.0
    eng.sync_guard_fields(eb)
    fields.update({"pad_" + a: g.view(a)[:, 3:3 + NY].cpu().numpy()[None] for a in eb})
    q.put((rank, np.array(trace), fields))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _launch(world, left_only=False, long_ahead=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, q, left_only, long_ahead)) for r in range(world)]
    for p in procs:
        p.daemon = True
        p.start()
    try:
        res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:          # a worker stuck in a collective must not outlive the test
            if p.is_alive():
                p.terminate()
    trace = sum(r[1] for r in res)
    fields = {a: np.concatenate([r[2][a] for r in res], axis=0) for a in res[0][2]}
    return trace, fields


def _check_padded(fields, world):
    """x guard planes after a six-component guard sync == the periodic neighbours' interior edges"""
    nx = NXG // world
    for a in ("ex", "ey", "ez", "bx", "by", "bz"):
        whole = fields[a]                                   # [NXG][NY] interiors, ranks concatenated
        for r in range(world):
            pad = fields["pad_" + a][r]                     # [nx + 6][NY]
            rows = (np.arange(-3, nx + 3) + r * nx) % NXG
            assert np.array_equal(pad, whole[rows]), (a, r)


@pytest.fixture(scope="module")
def single():
    return _launch(1)


@pytest.mark.parametrize("world", [2, 3])
def test_a_rank_without_particles_keeps_pace(world):
    """plasma in the left 40 % of the box: the last rank of 2 / 3 starts EMPTY and fills by migration (a second species is
    empty on every rank).  Every exchange of a step is
    unconditional -- in particular the jx plane of the rho continuity update (rho.py), which a rank that pushed nothing
    used to skip while its neighbour waited for it -- and each rank sorts, and re-anchors rho, when ITS particles ask
    for it.  Against the same problem on one rank."""
    t1, f1 = _launch(1, left_only=True)
    t2, f2 = _launch(world, left_only=True)
    assert np.array_equal(t2[:, 3], t1[:, 3]) and t1[0, 3] > 10000
    np.testing.assert_allclose(t2[:, 0], t1[:, 0], rtol=1e-10)
    np.testing.assert_allclose(t2[:, 1], t1[:, 1], rtol=1e-12)
    np.testing.assert_allclose(t2[:, 2], t1[:, 2], rtol=1e-12)
    for a in f1:
        if not a.startswith("pad_"):
            assert np.abs(f2[a] - f1[a]).max() <= 1e-9 * np.abs(f1[a]).max(), a


@pytest.mark.parametrize("world", [2, 3])
def test_slabs_match_single_rank(single, world):
    t1, f1 = single
    tn, fn = _launch(world)
    assert np.array_equal(tn[:, 3], t1[:, 3])                       # particle count conserved
    np.testing.assert_allclose(tn[:, 0], t1[:, 0], rtol=1e-10)      # field energy
    np.testing.assert_allclose(tn[:, 1], t1[:, 1], rtol=1e-12)      # total charge
    np.testing.assert_allclose(tn[:, 2], t1[:, 2], rtol=1e-12)      # kinetic energy
    for a in f1:
        if a.startswith("pad_"):
            continue
        scale = np.abs(f1[a]).max()
        assert np.abs(fn[a] - f1[a]).max() <= 1e-9 * scale, a
    _check_padded(f1, 1)
    _check_padded(fn, world)


# ---- laser-target like case (BASELINE config C3 in small): PML on all sides, laser from x-min, plasma
# ---- slab with two species, chain of x-slabs (open ends) ------------------------------------------
def _run_laser_target(rank, world, port, q):
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from lambdapic_amd.dist import SlabComm
    from lambdapic_amd.laser import SimpleLaser2D
    from lambdapic_amd.simulation import Simulation, Species
    lam = 0.8e-6
    nx, ny = 192, 64
    dx = dy = lam / 16
    bc = {k: "pml" for k in ("xmin", "xmax", "ymin", "ymax")}
    comm = SlabComm(None, periodic=False, single=(world == 1))
    # npatch_x = 4 / world: the mirrors' patch origins (= loading seeds) do not depend on the world size
    sim = Simulation(nx, ny, dx, dy, npatch_x=4 // world, boundary_conditions=bc, cpml_thickness=6, comm=comm,
                     sort_interval=5)
    nc = 1.742e27
    dens = lambda x, y: np.where((x > 100 * dx) & (x < 130 * dx) & (y > 10 * dy) & (y < 54 * dy), 2 * nc, 0.0)
    sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=4, momentum_sigma=0.02))
    sim.add_species(Species("p", charge=1, mass=1836.0, density=dens, ppc=4, momentum_sigma=0.0))
    sim.random_seed = 1234
    sim.initialize()
    laser = SimpleLaser2D(a0=2.0, w0=2.5e-6, ctau=2.0e-6, l0=lam)
    trace = []
    for it in range(110):
        sim.run(1, callbacks=[laser])
        if it % 10 == 9:
            d = sim.engine.diagnostics()
            trace.append([d["field_energy"], d["charge"], sum(d["kinetic"]), sum(d["nalive"])])
    g = sim.engine.grid
    nxl = nx // world
    fields = {a: g.view(a)[3:3 + nxl, 3:3 + ny].cpu().numpy() for a in ("ex", "ey", "ez", "bx", "by", "bz", "rho")}
    # get_fields (callback/utils.py:26-123): the whole box on rank 0, None elsewhere
    from lambdapic_amd.callbacks import get_fields
    whole = get_fields(sim, ["ey", "rho"])
    if world > 1:
        slabs = [None] * world if rank == 0 else None
        dist.gather_object(fields["ey"], slabs, dst=0)          # an independent (pickled) path
        if rank == 0:
            assert whole[0].shape == (nx, ny) and np.array_equal(whole[0], np.concatenate(slabs, axis=0))
        else:
            assert whole == [None, None]
    else:
        assert np.array_equal(whole[0], fields["ey"]) and np.array_equal(whole[1], fields["rho"])
    # sim.mpi as the reference's callbacks use it (rank / size / comm.gather / Barrier / split sync brackets)
    assert (sim.mpi.rank, sim.mpi.size) == (rank, world) and sim.mpi.comm.Get_rank() == rank
    got = sim.mpi.comm.gather({"rank": rank, "n": int(d["nalive"][0])})
    everyone = sim.mpi.comm.allgather(int(d["nalive"][0]))        # (collectives: every rank calls them)
    if rank == 0:
        assert [g_["rank"] for g_ in got] == list(range(world))
        assert sum(g_["n"] for g_ in got) == sum(everyone)
    else:
        assert got is None
    assert sim.mpi.comm.bcast("hello" if rank == 0 else None) == "hello"
    assert sim.mpi.comm.allreduce(np.array([1.0, rank])).tolist() == [world, sum(range(world))]
    h = sim.mpi.sync_guard_fields_start(["ex", "ey", "ez"])
    assert (h is None) == (world == 1)
    sim.mpi.sync_guard_fields_wait(h)
    sim.mpi.comm.Barrier()
    q.put((rank, np.array(trace), fields))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _launch_lt(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_laser_target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.daemon = True
        p.start()
    try:
        res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:          # a worker stuck in a collective must not outlive the test
            if p.is_alive():
                p.terminate()
    trace = sum(r[1] for r in res)
    fields = {a: np.concatenate([r[2][a] for r in res], axis=0) for a in res[0][2]}
    return trace, fields


def test_laser_target_chain_matches_single_rank():
    t1, f1 = _launch_lt(1)
    t2, f2 = _launch_lt(2)
    assert t1[-1, 0] > 0 and t1[-1, 2] > t1[0, 2]             # the laser heats the target
    assert np.array_equal(t2[:, 3], t1[:, 3])
    np.testing.assert_allclose(t2[:, 0], t1[:, 0], rtol=1e-9)
    np.testing.assert_allclose(t2[:, 2], t1[:, 2], rtol=1e-9)
    for a in f1:
        scale = np.abs(f1[a]).max()
        assert np.abs(f2[a] - f1[a]).max() <= 1e-8 * scale, a


# ---- moving window on a slab chain: the columns that leave a slab's low face become the left
# ---- neighbour's tail (rotation of the neighbour ring by one patch width) ----------------
def _run_window(rank, world, port, q, inject=True, direction=1):
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from lambdapic_amd import constants
    from lambdapic_amd.dist import SlabComm
    from lambdapic_amd.laser import SimpleLaser2D
    from lambdapic_amd.simulation import MovingWindow, Simulation, Species
    lam = 0.8e-6
    nx, ny = 256, 64
    dx = dy = lam / 16
    bc = {k: "pml" for k in ("xmin", "xmax", "ymin", "ymax")}
    comm = SlabComm(None, periodic=False, single=(world == 1))
    sim = Simulation(nx, ny, dx, dy, npatch_x=8 // world, boundary_conditions=bc, cpml_thickness=6, comm=comm,
                     sort_interval=5, random_seed=5)
    nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C / lam) ** 2 / constants.E_CHARGE ** 2
    if direction > 0:
        dens = lambda x, y: np.where((x > 150 * dx) & (abs(y - ny * dy / 2) < 20 * dy), 0.05 * nc, 0.0)
    else:       # a window moving backwards: the plasma lies at and beyond the low-x end (injected at the shifts)
        dens = lambda x, y: np.where((x < 60 * dx) & (abs(y - ny * dy / 2) < 20 * dy), 0.05 * nc, 0.0)
    sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=4, momentum_sigma=0.01))
    laser = SimpleLaser2D(a0=1.5, w0=1.2e-6, ctau=1.5e-6, l0=lam)
    # inject=False: the last rank gets neither arrivals nor fresh particles at a shift -- its tiling must still
    # be rebuilt for the moved origin (edge / leaver columns, tile margins)
    mw = MovingWindow(velocity=direction * C, start_time=(0.6 if direction > 0 else 0.2) * nx * dx / C, inject_particles=inject)
    trace = []
    for it in range(520):
        sim.run(1, callbacks=[laser, mw])
        if it % 40 == 39:
            d = sim.engine.diagnostics()
            trace.append([d["field_energy"], d["charge"], sum(d["kinetic"]), sum(d["nalive"])])
    g = sim.engine.grid
    nxl = nx // world
    fields = {a: g.view(a)[3:3 + nxl, 3:3 + ny].cpu().numpy() for a in ("ex", "ey", "ez", "bx", "by", "bz", "rho")}
    fields["_x0"] = np.full((1, ny), sim.engine.x0_global / dx)
    q.put((rank, np.array(trace), fields))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _launch_window(world, inject=True, direction=1):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_window, args=(r, world, port, q, inject, direction)) for r in range(world)]
    for p in procs:
        p.daemon = True
        p.start()
    try:
        res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    trace = sum(r[1] for r in res)
    fields = {a: np.concatenate([r[2][a] for r in res], axis=0) for a in res[0][2]}
    return trace, fields


@pytest.mark.parametrize("world,direction", [(2, 1), (4, 1), (2, -1)])
def test_moving_window_chain_matches_single_rank(world, direction):
    """(4 ranks: the two left slabs hold no particle until the window has moved the plasma into them; direction -1:
    the backward window -- columns and particles travel to the RIGHT neighbour, rank 0 injects)"""
    t1, f1 = _launch_window(1, direction=direction)
    t2, f2 = _launch_window(world, direction=direction)
    assert direction * f1["_x0"][0, 0] >= 4 * 32 and np.array_equal(f1["_x0"][0], f2["_x0"][0])   # both shifted alike
    assert t1[-1, 3] > 1000                                        # plasma was injected and kept
    assert np.array_equal(t2[:, 3], t1[:, 3])
    np.testing.assert_allclose(t2[:, 0], t1[:, 0], rtol=1e-9)
    np.testing.assert_allclose(t2[:, 1], t1[:, 1], rtol=1e-9, atol=1e-12 * np.abs(t1[:, 1]).max())
    np.testing.assert_allclose(t2[:, 2], t1[:, 2], rtol=1e-9)
    for a in ("ex", "ey", "ez", "bx", "by", "bz", "rho"):
        scale = np.abs(f1[a]).max()
        assert np.abs(f2[a] - f1[a]).max() <= 1e-8 * scale, a


def test_moving_window_chain_without_injection_matches_single_rank():
    t1, f1 = _launch_window(1, inject=False)
    t2, f2 = _launch_window(2, inject=False)
    assert f1["_x0"][0, 0] >= 4 * 32 and np.array_equal(f1["_x0"][0], f2["_x0"][0])
    assert 0 < t1[-1, 3] < t1[0, 3]                                # plasma leaves through the low face, none enters
    assert np.array_equal(t2[:, 3], t1[:, 3])
    np.testing.assert_allclose(t2[:, 0], t1[:, 0], rtol=1e-9)
    np.testing.assert_allclose(t2[:, 1], t1[:, 1], rtol=1e-9, atol=1e-12 * np.abs(t1[:, 1]).max())
    np.testing.assert_allclose(t2[:, 2], t1[:, 2], rtol=1e-9)
    for a in ("ex", "ey", "ez", "bx", "by", "bz", "rho"):
        scale = np.abs(f1[a]).max()
        assert np.abs(f2[a] - f1[a]).max() <= 1e-8 * scale, a


# ---- 3-D slabs: two ranks against one on the same periodic box ----------------------------------------
def _run_3d(rank, world, port, q, left_only=False):
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from lambdapic_amd.dist import SlabComm
    from lambdapic_amd.engine3d import ATTRS3, PicEngine3D
    nxg, ny, nz, ppc = 16, 8, 16, 4
    dx, dy, dz = 4e-8, 5e-8, 6e-8
    dt = 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2 + dz ** -2))
    rng = np.random.default_rng(3)
    n = nxg * ny * nz * ppc
    pos = np.stack([rng.uniform(-0.5, m - 0.5, n) * d for m, d in ((nxg, dx), (ny, dy), (nz, dz))])
    u = rng.normal(size=(3, n)) * 0.3                       # hot: slab crossings and box wraps
    ig = 1 / np.sqrt(1 + (u ** 2).sum(0))
    w = np.full(n, 1e27 * dx * dy * dz / ppc)
    comm = SlabComm(None, periodic=True, single=(world == 1))
    nx = nxg // world
    eng = PicEngine3D(nx, ny, nz, dx, dy, dz, 3, tiled=True, sort_interval=3, block_particles=1024, comm=comm,
                      migrate_capacity=2048)
    lo, hi = (rank * nx - 0.5) * dx, ((rank + 1) * nx - 0.5) * dx
    mine = (pos[0] >= lo) & (pos[0] < hi)
    if left_only:                   # plasma in the left 40 % of the box: rank 1 of 2 starts without a particle
        mine &= pos[0] < 0.4 * nxg * dx
    k = int(mine.sum())
    cap = 3 * k + eng.arrival_area() + 4096
    data = torch.full((len(ATTRS3), cap), float("nan"), dtype=torch.float64, device="cuda:0")
    data[:, :k] = torch.from_numpy(np.concatenate([pos[:, mine], u[:, mine], ig[None, mine], w[None, mine]])).cuda()
    # _id = the particle's index in the global arrays: the same particle has the same id whatever the decomposition
    eng.add_species_device(-1.602176634e-19, 9.1093837139e-31, data, k,
                           ids=torch.from_numpy(np.nonzero(mine)[0].astype(np.int64)))
    trace = []
    for _ in range(14):
        eng.step(dt)
        d = eng.diagnostics()
        trace.append([d["field_energy"], d["charge"], d["kinetic"][0], d["nalive"][0]])
    sl = (slice(3, 3 + nx), slice(3, 3 + ny), slice(3, 3 + nz))
    fields = {a: eng.view(a)[sl].cpu().numpy() for a in ("ex", "ey", "ez", "bx", "by", "bz", "rho")}
    got = eng.download_species(0)
    fields["part_id"] = got["_id"].view(np.int64).astype(np.float64)
    for a in ("x", "y", "z", "ux", "uy", "uz"):
        fields["part_" + a] = got[a]
    # E and B guards in ONE call (which = 3): six components per face message
    eb = ("ex", "ey", "ez", "bx", "by", "bz")
    for a in eb:
        eng.view(a)[:3] = 777.0
        eng.view(a)[3 + nx:] = -777.0
    eng.sync_guard_fields(3)
    fields.update({"pad_" + a: eng.view(a)[:, 3:3 + ny, 3:3 + nz].cpu().numpy()[None] for a in eb})
    q.put((rank, np.array(trace), fields))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _launch_3d(world, left_only=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_3d, args=(r, world, port, q, left_only)) for r in range(world)]
    for p in procs:
        p.daemon = True
        p.start()
    try:
        res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:          # a worker stuck in a collective must not outlive the test
            if p.is_alive():
                p.terminate()
    trace = sum(r[1] for r in res)
    fields = {a: np.concatenate([r[2][a] for r in res], axis=0) for a in res[0][2]}
    return trace, fields


def test_3d_rank_without_particles_keeps_pace():
    """the 3-D twin of test_a_rank_without_particles_keeps_pace: one clock for the chain's sorts and real rho deposits"""
    t1, f1 = _launch_3d(1, left_only=True)
    t2, f2 = _launch_3d(2, left_only=True)
    assert np.array_equal(t2[:, 3], t1[:, 3]) and t1[0, 3] > 2000
    np.testing.assert_allclose(t2[:, 0], t1[:, 0], rtol=1e-10)
    np.testing.assert_allclose(t2[:, 2], t1[:, 2], rtol=1e-12)
    assert np.abs(t2[:, 1] - t1[:, 1]).max() <= 1e-12 * np.abs(t1[:, 1]).max()
    for a in ("ex", "ey", "ez", "bx", "by", "bz", "rho"):
        assert np.abs(f2[a] - f1[a]).max() <= 1e-9 * np.abs(f1[a]).max(), a


def test_3d_slabs_match_single_rank():
    t1, f1 = _launch_3d(1)
    t2, f2 = _launch_3d(2)
    assert np.array_equal(t2[:, 3], t1[:, 3])                       # particle count conserved
    np.testing.assert_allclose(t2[:, 0], t1[:, 0], rtol=1e-10)
    np.testing.assert_allclose(t2[:, 2], t1[:, 2], rtol=1e-12)
    assert np.abs(t2[:, 1] - t1[:, 1]).max() <= 1e-12 * 8192 * 1e27 * 4e-8 * 5e-8 * 6e-8 / 4 * 1.6e-19
    for a in f1:
        if a.startswith("pad_") or a.startswith("part_"):
            continue
        scale = np.abs(f1[a]).max()
        assert np.abs(f2[a] - f1[a]).max() <= 1e-9 * scale, a
    # particle identity through 14 steps of slab crossings, box wraps and 4 re-sorts (the reference carries _id in
    # its migration payload, core/mpi/sync_particles_3d.c, and through its sort, core/sort/cpu3d.c:214-299):
    # every id exactly once over the two ranks, and the same particle under it as in the single-rank run
    n = 16 * 8 * 16 * 4
    for f in (f1, f2):
        assert np.array_equal(np.sort(f["part_id"]), np.arange(n))
    o1, o2 = np.argsort(f1["part_id"]), np.argsort(f2["part_id"])
    for a, scale in (("x", 16 * 4e-8), ("y", 8 * 5e-8), ("z", 16 * 6e-8), ("ux", 1.0), ("uy", 1.0), ("uz", 1.0)):
        assert np.abs(f2["part_" + a][o2] - f1["part_" + a][o1]).max() <= 1e-9 * scale, a
    for f, world in ((f1, 1), (f2, 2)):                 # x guard planes == the periodic neighbours' interior edges
        nxl = 16 // world
        for a in ("ex", "ey", "ez", "bx", "by", "bz"):
            for r in range(world):
                rows = (np.arange(-3, nxl + 3) + r * nxl) % 16
                assert np.array_equal(f["pad_" + a][r], f[a][rows]), (a, world, r)


# ---- 3-D chain with CPML on all faces, laser from x-min, plasma slab: 2 slabs against 1 ----------------
def _run_3d_open(rank, world, port, q, stress=False):
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from lambdapic_amd import constants
    from lambdapic_amd.dist import SlabComm
    from lambdapic_amd.engine3d import ATTRS3, PicEngine3D
    lam = 0.8e-6
    nxg, ny, nz, ppc = 48, 24, 32, 2
    dx, dy, dz = lam / 10, lam / 5, lam / 5
    dt = 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2 + dz ** -2))
    bc = {k: "pml" for k in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")}
    comm = SlabComm(None, periodic=False, single=(world == 1))
    nx = nxg // world
    eng = PicEngine3D(nx, ny, nz, dx, dy, dz, 3, tiled=True, sort_interval=4, block_particles=1024, comm=comm,
                      migrate_capacity=2048, boundary_conditions=bc, cpml_thickness=4)
    rng = np.random.default_rng(8)
    cells = np.array([(i, j, k) for i in (range(6, 16) if stress else range(20, 30)) for j in range(6, 18)
                      for k in range(8, 24)])
    n = len(cells) * ppc
    pos = (np.repeat(cells, ppc, axis=0) + rng.uniform(-0.5, 0.5, (n, 3))).T * np.array([[dx], [dy], [dz]])
    u = rng.normal(size=(3, n)) * (0.5 if stress else 0.05)
    if stress:
        # a hot slab next to the x-min layer, all of it on rank 0 (rank 1 of 2 starts empty), and an absorbed-particle
        # list of EIGHT entries: rank 0 overflows it step after step, rank 1 never does.  What does not fit the list goes to
        # the spill array (lpa_push_params.absorbed_spill) and leaves rho with the listed entries: rho is exact in every
        # step, the ranks need no vote, and the single slab and the chain may sort (and re-deposit) in different steps
        eng.ABSORBED_MIN_CAPACITY = 8
        eng._rho_particle_slots = lambda: 0
    ig = 1 / np.sqrt(1 + (u ** 2).sum(0))
    w = np.full(n, 3e27 * dx * dy * dz / ppc)
    lo, hi = (rank * nx - 0.5) * dx, ((rank + 1) * nx - 0.5) * dx
    mine = (pos[0] >= lo) & (pos[0] < hi)
    k = int(mine.sum())
    cap = 2 * n + eng.arrival_area()
    data = torch.full((len(ATTRS3), cap), float("nan"), dtype=torch.float64, device="cuda:0")
    data[:, :k] = torch.from_numpy(np.concatenate([pos[:, mine], u[:, mine], ig[None, mine], w[None, mine]])).cuda()
    eng.add_species_device(-constants.E_CHARGE, constants.M_E, data, k)
    om = 2 * np.pi * C / lam
    E0 = 3.0 * constants.M_E * C * om / constants.E_CHARGE
    y = (np.arange(ny) * dy - dy / 2 - ny * dy / 2)[:, None]
    z = (np.arange(nz) * dz - dz / 2 - nz * dz / 2)[None, :]
    prof = E0 * np.exp(-(y ** 2 + z ** 2) / (1.2e-6) ** 2)
    state = {"t": 0.0}

    def laser(e, h):
        t = state["t"]
        if C * t < 2 * 1.2e-6:
            e.laser_inject(prof * np.sin(C * t / (2 * 1.2e-6) * np.pi) ** 2 * np.sin(om * t), 0 * prof, h)

    trace = []
    for it in range(60):
        eng.step(dt, laser=laser)
        state["t"] += dt
        if it % 6 == 5:
            d = eng.diagnostics()
            trace.append([d["field_energy"], d["charge"], d["kinetic"][0], d["nalive"][0]])
    sl = (slice(3, 3 + nx), slice(3, 3 + ny), slice(3, 3 + nz))
    fields = {a: eng.view(a)[sl].cpu().numpy() for a in ("ex", "ey", "ez", "bx", "by", "bz", "rho")}
    q.put((rank, np.array(trace), fields))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("stress", [False, True])
def test_3d_laser_target_chain_matches_single_rank(stress):
    def launch(world):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_run_3d_open, args=(r, world, port, q, stress)) for r in range(world)]
        for p in procs:
            p.daemon = True
            p.start()
        try:
            res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
            for p in procs:
                p.join(timeout=60)
                assert p.exitcode == 0
        finally:
            for p in procs:
                if p.is_alive():
                    p.terminate()
        return sum(r[1] for r in res), {a: np.concatenate([r[2][a] for r in res], axis=0) for a in res[0][2]}

    t1, f1 = launch(1)
    t2, f2 = launch(2)
    if stress:
        assert t1[-1, 3] < 0.95 * t1[0, 3]                       # part of the hot slab was absorbed (tens per step: > 8)
        np.testing.assert_allclose(t2[:, 1], t1[:, 1], rtol=1e-9)    # ... and the charge left rho with it, on both
    else:
        assert t1[-1, 2] > 5 * t1[0, 2]                          # the laser heats the slab
    assert np.array_equal(t2[:, 3], t1[:, 3])
    np.testing.assert_allclose(t2[:, 0], t1[:, 0], rtol=1e-9)
    np.testing.assert_allclose(t2[:, 2], t1[:, 2], rtol=1e-9)
    for a in f1:
        scale = np.abs(f1[a]).max()
        assert np.abs(f2[a] - f1[a]).max() <= 1e-8 * scale, a


# ---- 3-D moving window on a slab chain -------------------------------------------------------------------
def _run_window_3d(rank, world, port, q, direction=1):
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from lambdapic_amd import constants
    from lambdapic_amd.dist import SlabComm
    from lambdapic_amd.laser import GaussianLaser3D
    from lambdapic_amd.simulation import MovingWindow
    from lambdapic_amd.simulation3d import Simulation3D, Species
    lam = 0.8e-6
    nx, ny, nz = 96, 16, 32
    dx, dy, dz = lam / 10, lam / 5, lam / 5
    comm = SlabComm(None, periodic=False, single=(world == 1))
    sim = Simulation3D(nx, ny, nz, dx, dy, dz, npatch_x=6 // world, cpml_thickness=4, random_seed=4, sort_interval=4,
                       block_particles=1024, comm=comm)
    nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C / lam) ** 2 / constants.E_CHARGE ** 2
    along_x = (lambda x: x > 70 * dx) if direction > 0 else (lambda x: x < 30 * dx)     # (backwards: plasma at and beyond x-min)
    dens = lambda x, y, z: np.where(along_x(x) & (abs(y - sim.Ly / 2) < 4 * dy) & (abs(z - sim.Lz / 2) < 8 * dz),
                                    0.05 * nc, 0.0)
    sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=2, momentum_sigma=0.01))
    laser = GaussianLaser3D(a0=1.5, l0=lam, w0=0.8e-6, ctau=0.5e-6, x0=1.2e-6)
    mw = MovingWindow(velocity=direction * C, start_time=(0.7 if direction > 0 else 0.1) * nx * dx / C)
    trace = []
    for it in range(170):
        sim.run(1, callbacks=[laser, mw])
        if it % 17 == 16:
            d = sim.engine.diagnostics()
            trace.append([d["field_energy"], d["charge"], sum(d["kinetic"]), sum(d["nalive"])])
    nxl = nx // world
    sl = (slice(3, 3 + nxl), slice(3, 3 + ny), slice(3, 3 + nz))
    fields = {a: sim.engine.view(a)[sl].cpu().numpy() for a in ("ex", "ey", "ez", "bx", "by", "bz", "rho")}
    fields["_x0"] = np.full((1, ny, nz), (sim.engine.x0 - rank * nxl * dx) / dx)
    q.put((rank, np.array(trace), fields))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("direction", [1, -1])
def test_moving_window_3d_chain_matches_single_rank(direction):
    def launch(world):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_run_window_3d, args=(r, world, port, q, direction)) for r in range(world)]
        for p in procs:
            p.daemon = True
            p.start()
        try:
            res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
            for p in procs:
                p.join(timeout=60)
                assert p.exitcode == 0
        finally:
            for p in procs:
                if p.is_alive():
                    p.terminate()
        return sum(r[1] for r in res), {a: np.concatenate([r[2][a] for r in res], axis=0) for a in res[0][2]}

    t1, f1 = launch(1)
    t2, f2 = launch(2)
    assert direction * f1["_x0"][0, 0, 0] >= 3 * 16 and np.array_equal(f1["_x0"][0], f2["_x0"][0])
    assert t1[-1, 3] > 200 and np.array_equal(t2[:, 3], t1[:, 3])
    np.testing.assert_allclose(t2[:, 0], t1[:, 0], rtol=1e-9)
    np.testing.assert_allclose(t2[:, 2], t1[:, 2], rtol=1e-9)
    for a in ("ex", "ey", "ez", "bx", "by", "bz", "rho"):
        scale = np.abs(f1[a]).max()
        assert np.abs(f2[a] - f1[a]).max() <= 1e-8 * scale, a


# ---- particle migration pinned to the reference's fixture (g7: sync_particles on 2 x 2 periodic patches) ----------
def _run_g7_migration(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C_
    from pathlib import Path
    from lambdapic_amd import _lib
    from lambdapic_amd.dist import SlabComm
    from lambdapic_amd.engine import PicEngine2D
    from lambdapic_amd.particles import ParticlesBase
    g = np.load(Path(__file__).resolve().parent / "golden" / "g7_sync_2d.npz")
    nx, ny, dx, dy = int(g["nx"]), int(g["ny"]), float(g["dx"]), float(g["dy"])     # 16 x 16 cells, 2 x 2 patches
    eng = PicEngine2D(nx // world, ny, dx, dy, device="cuda:0", comm=SlabComm(None), sort_interval=4,
                      block_particles=1024, migrate_capacity=512)
    bags = []
    for k in (rank, rank + 2):          # this rank's slab = the patch column ipatch_x = rank (patches k and k + 2)
        p = ParticlesBase(0, 0)
        p.initialize(g[f"pin{k}_x"].size)
        for a in ("x", "y", "ux", "w", "_id"):
            getattr(p, a)[:] = g[f"pin{k}_{a}"]
        p.inv_gamma[:] = 1.0
        p.is_dead[:] = g[f"pin{k}_is_dead"]
        bags.append(p)
    eng.add_species(-1.6e-19, 9.1e-31, capacity=2048)
    sp = eng.species[0]
    sp.upload(bags)
    eng.sort(0)
    eng.sync_particles(0)               # leavers through the x faces travel to the ring neighbour (+- Lx at the box edge)
    pp = eng._push_params(sp, 1.0)      # the y fold the fused kernel applies itself (a patch that is its own y neighbour)
    pc = sp.cset.cstruct(sp.n)
    _lib.check(eng.L.lpa_wrap_positions_2d(C_.byref(pc), C_.byref(pp), eng.stream), "lpa_wrap_positions_2d")
    out = sp.download()
    # the E / B guard copy of the same fixture through the slab path: local wrap along y + one halo message per x face
    from lambdapic_amd.patch import make_patches_2d
    P = make_patches_2d(nx // world, ny, dx, dy, 1, 2)
    E6 = ["ex", "ey", "ez", "bx", "by", "bz"]
    for p, k in zip(P, (rank, rank + 2)):
        for a in E6:
            getattr(p.fields, a)[...] = g[f"in{k}_{a}"]
    eng.grid.upload_patches(list(P), 1, 2, attrs=E6)
    eng.sync_guard_fields(E6)
    eng.grid.download_patches(list(P), attrs=E6)
    guards_ok = all(np.array_equal(getattr(p.fields, a), g[f"out{k}_{a}"]) for p, k in zip(P, (rank, rank + 2)) for a in E6)
    q.put((rank, {a: out[a] for a in ("x", "y", "ux", "w", "_id")}, guards_ok))
    dist.barrier()
    dist.destroy_process_group()


def test_slab_migration_vs_reference_golden(golden):
    """what `get_npart_to_extend_2d` + `fill_particles_from_boundary_2d` do on 2 x 2 periodic patches
    (core/patch/sync_particles_2d.c:204-518, recorded in g7) == what two x-slabs do with one migration message per face
    + the local periodic fold along y: the same particles end up in the same patch column with bit-identical
    coordinates (the +- L shifts included)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_g7_migration, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.daemon = True
        p.start()
    try:
        got_all = [q.get(timeout=120) for _ in range(2)]
        res = {r[0]: r[1] for r in got_all}
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    # sync_guard_fields_2d of the fixture (core/patch/sync_fields2d.c:150-255) == y wrap + x halo of the slabs, bit exact
    assert all(r[2] for r in got_all), "E / B guards of the slabs differ from the reference's patch guards"
    g = golden("g7_sync_2d")
    total = 0
    for rank in range(2):
        exp = {a: np.concatenate([g[f"pout{k}_{a}"][~g[f"pout{k}_is_dead"]] for k in (rank, rank + 2)])
               for a in ("x", "y", "ux", "w", "_id")}
        got = res[rank]
        o, r = np.argsort(got["_id"].view(np.uint64)), np.argsort(exp["_id"].view(np.uint64))
        assert np.array_equal(got["_id"].view(np.uint64)[o], exp["_id"].view(np.uint64)[r]), rank
        for a in ("x", "y", "ux", "w"):
            assert np.array_equal(got[a][o], exp[a][r]), (rank, a)
        total += got["x"].size
    assert total == sum(int((~g[f"pin{k}_is_dead"]).sum()) for k in range(4))       # nobody lost, nobody doubled


# ---- the chain's common sort clock follows the overflow lists ----------------------------------------------------------
def _run_hot_chain(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lambdapic_amd.dist import SlabComm
    from lambdapic_amd.engine import PicEngine2D
    comm = SlabComm(None)
    dx, dy, dt, x, y, u, ig, w = _problem()
    u = u * (2.0 if rank == 0 else 0.05)         # rank 0's half is relativistically hot, rank 1's is cold
    ig = 1 / np.sqrt(1 + (u ** 2).sum(0))
    nx = NXG // world
    eng = PicEngine2D(nx, NY, dx, dy, device="cuda:0", comm=comm, sort_interval=12, block_particles=1024, migrate_capacity=16384)
    lo, hi = (rank * nx - 0.5) * dx, ((rank + 1) * nx - 0.5) * dx
    mine = (x >= lo) & (x < hi)
    n = int(mine.sum())
    eng.add_species(-1.602176634e-19, 9.1093837139e-31, capacity=4 * n + 40000)
    s = eng.species[0].cset
    for name, arr in (("x", x), ("y", y), ("ux", u[0]), ("uy", u[1]), ("uz", u[2]), ("inv_gamma", ig), ("w", w)):
        s.arr(name)[:n] = torch.from_numpy(arr[mine]).cuda()
    s.id[:n] = torch.from_numpy(np.nonzero(mine)[0]).cuda()
    eng.species[0].n = n
    hist, sorts = [], []
    for it in range(48):
        before = eng.rho_steps["anchor"]
        eng.step(dt)
        hist.append(eng._chain_interval())
        sorts.append(eng.species[0].steps_since_sort == 1)      # (a sort ran at the start of this step)
    d = eng.diagnostics(reduce=True)
    q.put((rank, hist, sorts, d["nalive"][0], d["charge"], float(w.sum()) * -1.602176634e-19))
    dist.barrier()
    dist.destroy_process_group()


def test_chain_sort_clock_follows_the_overflow_lists():
    """One clock for all ranks (rho.py) -- but its period is no longer fixed: at every common sort the ranks agree on the
    shortest interval any species' controller asks for.  Rank 0 holds a relativistically hot plasma, rank 1 a cold one:
    both end up sorting every few steps, in the same steps, and nothing is lost on the way."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_hot_chain, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.daemon = True
        p.start()
    try:
        res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    (_, h0, s0, n0, c0, qw), (_, h1, s1, n1, c1, _) = res
    assert h0 == h1 and s0 == s1                      # the same interval and the same sort steps on both ranks
    assert h0[0] == 12 or h0[0] < 12                  # (the first sort's speed estimate may already shorten it)
    assert min(h0) <= 4 and h0[-1] <= 6               # ... and the hot half drives it down for the whole chain
    assert sum(s0) >= 8
    assert n0 == NXG * NY * PPC
    assert c0 == pytest.approx(qw, rel=1e-9)


@pytest.mark.parametrize("world", [2, 3])
def test_look_ahead_binning_on_a_chain(world):
    """the sort files a particle where it will be half an interval on (lpa_sort_tiles_ahead_2d) -- on a chain the edge /
    interior split of the overlapped push and the leaver scan count tile columns from the faces, so they have to allow for
    particles that sit closer to a face than their tile says (engine.edge_columns / leaver_columns).  A long fixed clock
    (20 steps: ten steps of look-ahead) on a relativistic version of the base test's plasma: 2 ranks (overlapped exchange)
    and 3 ranks against the single slab"""
    t1, f1 = _launch(1, long_ahead=True)
    t, f = _launch(world, long_ahead=True)
    assert np.array_equal(t[:, 3], t1[:, 3])                          # live count
    np.testing.assert_allclose(t[:, 0], t1[:, 0], rtol=1e-9)          # field energy
    np.testing.assert_allclose(t[:, 2], t1[:, 2], rtol=1e-10)         # kinetic energy
    assert np.abs(t[:, 1] - t1[:, 1]).max() <= 1e-12 * NXG * NY * PPC * 1.7e27 * (0.8e-6 / 20) ** 2 / PPC * 1.602176634e-19
    for a in ("ex", "ey", "ez", "bx", "by", "bz", "rho"):
        scale = np.abs(f1[a]).max()
        assert np.abs(f[a] - f1[a]).max() <= 1e-8 * scale, a
