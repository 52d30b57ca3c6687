"""RestartDump on the device-resident simulations (reference: `callback/restart.py:13-160`, its own test
`tests/test_restart.py:20-74` = itime / time bookkeeping across a load).

The reference holds no fixture for the pickle's content and its Simulation cannot be imported here, so the FILE FORMAT is
parity unpinned.  Pinned here: resume = no-op --
  (1) the state after ``RestartDump.load`` equals the state at the dump BIT FOR BIT (fields with guards, CPML psi rows,
      every live particle under its id);
  (2) N + M steps == N steps, dump, load into a fresh object, M steps -- to 1e-9 of the field maximum / 1e-9 per particle
      attribute: the resumed run re-sorts at another step than the continuous one and the deposit is an atomic sum, so
      two runs agree to summation order (two identical runs do not agree better);
  (3) itime / time bookkeeping as in the reference's test; ``keep`` trims old checkpoints.
2-D: CPML on all sides + laser + a moving window that shifts (with injection) BEFORE the dump; 3-D: CPML x 6 + laser, two
species; and once with two ranks through gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lambdapic_amd import constants
from lambdapic_amd.restart import RestartDump

pytestmark = pytest.mark.gpu
C = 299792458.0
LAM = 0.8e-6
NC = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C / LAM) ** 2 / constants.E_CHARGE ** 2


# ---- helpers ------------------------------------------------------------------------------------------------
def _particles_2d(eng):
    out = []
    for sp in eng.species:
        d = sp.download()
        o = np.argsort(d["_id"].view(np.uint64))
        out.append({k: v[o] for k, v in d.items()})
    return out


def _particles_3d(eng):
    out = []
    for i in range(len(eng.species)):
        d = eng.download_species(i)
        o = np.argsort(d["_id"].view(np.uint64))
        out.append({k: v[o] for k, v in d.items()})
    return out


def _same_bits(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint64),
                                                 np.ascontiguousarray(b).view(np.uint64))


def _close_particles(pa, pb, tol, scales):
    assert len(pa) == len(pb)
    for a, b in zip(pa, pb):
        assert np.array_equal(a["_id"].view(np.uint64), b["_id"].view(np.uint64))
        for k in a:
            if k == "_id" or k.endswith("_part"):
                continue
            assert np.abs(a[k] - b[k]).max() <= tol * scales.get(k, max(np.abs(b[k]).max(), 1e-300)), k


# ---- 2-D: CPML + laser + moving window with injection ---------------------------------------------------------
def _sim2d(**kw):
    from lambdapic_amd.simulation import Simulation, Species
    nx, ny = 128, 64
    dx = dy = LAM / 16
    sim = Simulation(nx, ny, dx, dy, npatch_x=8, npatch_y=2, cpml_thickness=6, random_seed=11, sort_interval=6, **kw)
    Ly = ny * dy
    dens = lambda x, y: np.where((x > 40 * dx) & (abs(y - Ly / 2) < 20 * dy), 0.5 * NC, 0.0)
    sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=4, momentum_sigma=0.02))
    sim.add_species(Species("p", charge=1, mass=1836.0, density=dens, ppc=2))
    return sim


def _cbs2d(sim_nx_dx):
    from lambdapic_amd.laser import GaussianLaser2D
    from lambdapic_amd.simulation import MovingWindow
    return [GaussianLaser2D(a0=2.0, l0=LAM, w0=1.2e-6, ctau=1.0e-6, x0=1.5e-6),
            MovingWindow(velocity=C, start_time=0.25 * sim_nx_dx / C)]


def test_restart_2d_cpml_laser_window(tmp_path):
    N, M = 70, 40
    a = _sim2d()
    cbs_a = _cbs2d(a.Lx)
    a.run(N + M, callbacks=cbs_a)
    assert a.window_shifts >= 2

    b = _sim2d()
    cbs_b = _cbs2d(b.Lx)
    dump = RestartDump(tmp_path, interval=lambda s: s.itime == N - 1)
    b.run(N, callbacks=cbs_b + [dump])
    assert b.window_shifts >= 1                        # a shift (with injection) lies before the dump
    ckpt = dump._ckpt_dir(N - 1)
    assert (ckpt / "rank_000000.pkl").is_file()
    c = RestartDump.load(ckpt)
    # (3) the reference's own test: itime fast-forwarded by one, time in step with it (tests/test_restart.py:66-74)
    assert c.itime == N == b.itime and c.time == pytest.approx(c.itime * c.dt, rel=1e-15)
    assert c is not b and c.engine is not b.engine
    # (1) the loaded state IS the dumped state
    assert torch.equal(c.engine.grid.buf, b.engine.grid.buf)
    assert c.engine.x0 == b.engine.x0 and c.engine.alo == b.engine.alo and c.window_shifts == b.window_shifts
    assert (c.engine.pml is None) == (b.engine.pml is None)
    if b.engine.pml is not None:
        assert c.engine.pml.sides == b.engine.pml.sides
        for lc, lb in zip(c.engine.pml.layers, b.engine.pml.layers):
            assert torch.equal(lc["psi_a"], lb["psi_a"]) and torch.equal(lc["psi_b"], lb["psi_b"])
    for pc, pb in zip(_particles_2d(c.engine), _particles_2d(b.engine)):
        assert pc["x"].size > 1000
        for k in pb:
            assert _same_bits(pc[k], pb[k]), k
    assert c._id_next == b._id_next
    # (2) the resumed run against the continuous one
    c.run(M, callbacks=cbs_b)
    assert c.itime == a.itime == N + M and c.window_shifts == a.window_shifts
    for name in ("ex", "ey", "ez", "bx", "by", "bz", "jx", "jy", "jz", "rho"):
        va, vc = a.engine.grid.view(name), c.engine.grid.view(name)
        assert (va - vc).abs().max().item() <= 1e-9 * max(va.abs().max().item(), 1e-300), name
    assert a.engine.grid.view("ey").abs().max().item() > 0
    _close_particles(_particles_2d(c.engine), _particles_2d(a.engine), 1e-9,
                     {"x": a.Lx, "y": a.Ly, "ux": 1.0, "uy": 1.0, "uz": 1.0})
    da, dc = a.engine.diagnostics(), c.engine.diagnostics()
    assert da["nalive"] == dc["nalive"]
    assert dc["field_energy"] == pytest.approx(da["field_energy"], rel=1e-9)


def test_restart_keep_and_signal(tmp_path):
    """``keep`` trims all but the newest checkpoints (`callback/restart.py:109-127`); a signal in ``dump_signals``
    requests one dump at the end of the current step and ``run`` returns (`simulation/simulation.py:1124-1127`)"""
    import signal
    sim = _sim2d()
    dump = RestartDump(tmp_path, interval=3, keep=2)
    sim.run(10, callbacks=[dump])
    assert sorted(p.name for p in tmp_path.iterdir()) == ["ckpt_000006", "ckpt_000009"]
    old = signal.getsignal(signal.SIGUSR1)
    try:
        dump2 = RestartDump(tmp_path / "sig", interval=10 ** 9, dump_signals=[signal.SIGUSR1])
        from lambdapic_amd.simulation import callback

        @callback(stage="start", interval=1)
        def kill(s):
            if s.itime == 13:
                os.kill(os.getpid(), signal.SIGUSR1)

        sim.run(20, callbacks=[dump2, kill])
        assert sim.itime == 13                       # returned after the dump, before the increment
        assert (tmp_path / "sig" / "ckpt_000013" / "rank_000000.pkl").is_file()
        c = RestartDump.load(tmp_path / "sig" / "ckpt_000013")
        assert c.itime == 14
        c.run(2)
        assert c.itime == 16
    finally:
        signal.signal(signal.SIGUSR1, old)


# ---- 3-D: CPML x 6 + laser, two species -------------------------------------------------------------------------
def _sim3d():
    from lambdapic_amd.simulation3d import Simulation3D, Species
    nx, ny, nz = 48, 24, 32
    sim = Simulation3D(nx, ny, nz, LAM / 10, LAM / 5, LAM / 5, npatch_x=2, npatch_z=2, cpml_thickness=4,
                       random_seed=3, sort_interval=4, block_particles=1024)
    dens = lambda x, y, z: np.where((x > 16 * sim.dx) & (x < 30 * sim.dx) & (abs(y - sim.Ly / 2) < 6 * sim.dy)
                                    & (abs(z - sim.Lz / 2) < 8 * sim.dz), 2 * NC, 0.0)
    sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=2, momentum_sigma=0.02))
    sim.add_species(Species("p", charge=1, mass=1836.0, density=dens, ppc=2))
    return sim


def test_restart_3d_cpml_laser(tmp_path):
    from lambdapic_amd.laser import GaussianLaser3D
    N, M = 30, 20
    mk = lambda: GaussianLaser3D(a0=3.0, l0=LAM, w0=1.0e-6, ctau=0.8e-6, x0=1.6e-6)
    a = _sim3d()
    a.run(N + M, callbacks=[mk()])
    b = _sim3d()
    laser_b = mk()
    dump = RestartDump(tmp_path, interval=lambda s: s.itime == N - 1)
    b.run(N, callbacks=[laser_b, dump])
    c = RestartDump.load(dump._ckpt_dir(N - 1))
    assert c.itime == N and c.time == pytest.approx(N * c.dt, rel=1e-15)
    assert torch.equal(c.engine.buf, b.engine.buf)
    for lc, lb in zip(c.engine.pml.layers, b.engine.pml.layers):
        assert torch.equal(lc["psi_a"], lb["psi_a"]) and torch.equal(lc["psi_b"], lb["psi_b"])
    for pc, pb in zip(_particles_3d(c.engine), _particles_3d(b.engine)):
        assert pc["x"].size > 1000
        for k in pb:
            assert _same_bits(pc[k], pb[k]), k
    assert c.engine._id_next == b.engine._id_next
    c.run(M, callbacks=[laser_b])
    for name in ("ex", "ey", "ez", "bx", "by", "bz", "jx", "jy", "jz", "rho"):
        va, vc = a.engine.view(name), c.engine.view(name)
        assert (va - vc).abs().max().item() <= 1e-9 * max(va.abs().max().item(), 1e-300), name
    assert a.engine.view("ey").abs().max().item() > 0
    _close_particles(_particles_3d(c.engine), _particles_3d(a.engine), 1e-9,
                     {"x": a.Lx, "y": a.Ly, "z": a.Lz, "ux": 1.0, "uy": 1.0, "uz": 1.0})
    assert a.engine.diagnostics()["nalive"] == c.engine.diagnostics()["nalive"]


# ---- two ranks through gloo: per-rank shards, communicator re-bound on load ---------------------------------------
def _free_port():
    # below the kernel's ephemeral range (32768-60999): an outgoing gloo connection of an earlier test cannot sit on it
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


def _worker(rank, world, port, tmp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lambdapic_amd.dist import SlabComm
    from lambdapic_amd.simulation import Simulation, Species
    N, M = 16, 12
    bc = {k: "periodic" for k in ("xmin", "xmax", "ymin", "ymax")}

    def mk():
        sim = Simulation(64, 32, LAM / 20, LAM / 20, npatch_x=2, npatch_y=2, boundary_conditions=bc, random_seed=5,
                         sort_interval=5, comm=SlabComm(None, periodic=True))
        sim.add_species(Species("e", charge=-1, mass=1, density=1.2 * NC, ppc=6, momentum_sigma=0.25))   # hot: migration
        return sim

    a = mk()
    a.run(N + M)
    b = mk()
    dump = RestartDump(tmp, interval=lambda s: s.itime == N - 1)
    b.run(N, callbacks=[dump])
    c = RestartDump.load(dump._ckpt_dir(N - 1))
    assert c.comm.rank == rank and c.comm.size == world and c.itime == N
    same = torch.equal(c.engine.grid.buf, b.engine.grid.buf)
    pc, pb = _particles_2d(c.engine)[0], _particles_2d(b.engine)[0]
    same = same and all(_same_bits(pc[k], pb[k]) for k in pb)
    c.run(M)
    g, gc = a.engine.grid, c.engine.grid
    err = max((g.view(n) - gc.view(n)).abs().max().item() / max(g.view(n).abs().max().item(), 1e-300)
              for n in ("ex", "ey", "ez", "bx", "by", "bz", "rho"))
    da, dc = a.engine.diagnostics(reduce=True), c.engine.diagnostics(reduce=True)
    ids_a = np.sort(_particles_2d(a.engine)[0]["_id"].view(np.uint64))
    ids_c = np.sort(_particles_2d(c.engine)[0]["_id"].view(np.uint64))
    q.put((rank, same, err, da["nalive"], dc["nalive"], da["field_energy"], dc["field_energy"],
           bool(np.array_equal(ids_a, ids_c)), int(ids_a.size)))
    dist.barrier()
    dist.destroy_process_group()


def test_restart_two_ranks_gloo(tmp_path):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
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
    assert sorted(f.name for f in (tmp_path / "ckpt_000015").iterdir()) == ["rank_000000.pkl", "rank_000001.pkl"]
    for rank, same, err, na, nc, ea, ec, ids_ok, nids in res:
        assert same, rank                              # loaded state == dumped state, per rank
        assert err <= 1e-9, (rank, err)
        assert na == nc and ids_ok and nids > 1000     # the same particles live on this rank after N + M steps
        assert ec == pytest.approx(ea, rel=1e-9)
