"""Simulation3D: the reference's stage / callback protocol on PicEngine3D (laser-target-3d in small)."""
import numpy as np
import pytest
import torch

from lambdapic_amd import constants
from lambdapic_amd.laser import GaussianLaser3D
from lambdapic_amd.simulation3d import Simulation3D, Species, callback

pytestmark = pytest.mark.gpu
C = 299792458.0
LAM = 0.8e-6


def _sim(**kw):
    nx, ny, nz = 48, 24, 32
    sim = Simulation3D(nx, ny, nz, LAM / 10, LAM / 5, LAM / 5, cpml_thickness=4, random_seed=3, sort_interval=4,
                       block_particles=1024, **kw)
    nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C / LAM) ** 2 / constants.E_CHARGE ** 2
    dens = lambda x, y, z: np.where((x > 20 * sim.dx) & (x < 30 * sim.dx) & (abs(y - sim.Ly / 2) < 6 * sim.dy)
                                    & (abs(z - sim.Lz / 2) < 8 * sim.dz), 2 * nc, 0.0)
    sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=2, momentum_sigma=0.02))
    sim.add_species(Species("p", charge=1, mass=1836.0, density=dens, ppc=2))
    return sim


def test_laser_target_3d_through_the_callback_api():
    """GaussianLaser3D at stage '_laser' heats a plasma block; a host callback at stage 'end' reads the
    mirrors (fields with guards, particles binned by patch), another writes an external field in; the
    device-native path and the mirrored path advance the same physics"""
    sim = _sim(npatch_x=2, npatch_y=1, npatch_z=2)
    laser = GaussianLaser3D(a0=3.0, l0=LAM, w0=1.0e-6, ctau=0.8e-6, x0=1.6e-6)
    seen = {}

    @callback(stage="end", interval=10)
    def probe(s):
        f = s.patches[0].fields
        assert f.ey.shape == (24 + 6, 24 + 6, 16 + 6)          # patch interior + guards, reference layout
        seen[s.itime] = (sum(float(np.sum(p.fields.ey[:24, :24, :16] ** 2)) for p in s.patches),
                         sum(p.particles[0].npart for p in s.patches),
                         float(sum(np.sum(1 / p.particles[0].inv_gamma - 1) for p in s.patches)))

    sim.run(60, callbacks=[laser, probe])
    d = sim.engine.diagnostics()
    n0 = seen[0][1]
    # (an a0 = 3 pulse pushes a few electrons sideways into the absorbing layer by the end)
    assert n0 > 2000 and seen[30][1] == n0 and 0.98 * n0 <= d["nalive"][0] <= n0
    assert seen[50][0] > 0 and seen[50][2] > 20 * seen[0][2]               # the pulse arrived and heats
    e_dev = float((sim.engine.view("ey")[3:-3, 3:-3, 3:-3] ** 2).sum().item())
    assert e_dev > 0

    # get_fields (callback/utils.py:125-237): the z = Lz / 2 plane of the whole box, assembled from the mirrors
    # the way the reference does it (patch interiors, plane index int((Lz / 2 + dz / 2) / dz))
    from lambdapic_amd.callbacks import get_fields
    ey_plane, = get_fields(sim, ["ey"])
    sim.download()
    iz = int((sim.Lz / 2 + sim.dz / 2) / sim.dz)
    want = np.zeros((sim.nx, sim.ny))
    for p in sim.patches:
        f = p.fields
        k0 = int(round((f.z0 - 0.0) / sim.dz))
        if k0 <= iz < k0 + f.nz:
            i0, j0 = int(round(f.x0 / sim.dx)), int(round(f.y0 / sim.dy))
            want[i0:i0 + f.nx, j0:j0 + f.ny] = f.ey[:f.nx, :f.ny, iz - k0]
    assert ey_plane.shape == (sim.nx, sim.ny) and np.abs(ey_plane).max() > 0
    assert np.array_equal(ey_plane, want)
    with pytest.raises(ValueError):
        get_fields(sim, ["ey"], slice_at=2 * sim.Lz)

    # the mirrored path: the same run with a do-nothing host callback at every step gives the same state
    sim2 = _sim(npatch_x=2, npatch_y=1, npatch_z=2)
    laser2 = GaussianLaser3D(a0=3.0, l0=LAM, w0=1.0e-6, ctau=0.8e-6, x0=1.6e-6)
    noop = callback(stage="maxwell_1", interval=1)(lambda s: None)
    sim2.run(60, callbacks=[laser2, noop])
    d2 = sim2.engine.diagnostics()
    assert d2["nalive"] == d["nalive"]
    assert d2["field_energy"] == pytest.approx(d["field_energy"], rel=1e-9)
    assert d2["kinetic"][0] == pytest.approx(d["kinetic"][0], rel=1e-9)


def test_mirror_round_trip_is_lossless_and_keeps_ids():
    """download -> upload through the 3-D patch mirrors must not change the device state, and a particle keeps its
    ``_id`` through it (`core/particles.py:63-67`: _id is one of the 15 attributes callbacks see; a tracking callback
    must find the same particle under the same id at every download)"""
    sim = _sim(npatch_x=2, npatch_y=1, npatch_z=2)
    sim.run(6)
    eng = sim.engine

    def by_id(i):
        d = eng.download_species(i)
        o = np.argsort(d["_id"].view(np.uint64))
        return {k: v[o] for k, v in d.items()}

    before = [by_id(i) for i in range(2)]
    diag0, ex0 = eng.diagnostics(), eng.view("ex").clone()
    for b in before:
        ids = b["_id"].view(np.uint64)
        assert ids.size > 1000 and np.unique(ids).size == ids.size
    sim.download()
    seen = np.concatenate([p.particles[0].id for p in sim.patches])           # what a callback sees
    assert np.array_equal(np.sort(seen), before[0]["_id"].view(np.uint64))
    sim.upload()
    diag1 = eng.diagnostics()
    # (the energy is an atomic sum: equal to an ulp; the arrays themselves are compared bit for bit)
    assert diag0["field_energy"] == pytest.approx(diag1["field_energy"], rel=1e-14) and diag0["nalive"] == diag1["nalive"]
    assert (eng.view("ex") - ex0).abs().max().item() == 0.0
    for i in range(2):
        after = by_id(i)
        for k in before[i]:
            assert np.array_equal(after[k].view(np.uint64), before[i][k].view(np.uint64)), (i, k)
    sim.run(5)                   # sorts (interval 4) and pushes: ids stay attached
    sim.download()
    sim.upload()
    sim.run(1)
    for i in range(2):
        assert np.array_equal(by_id(i)["_id"].view(np.uint64), before[i]["_id"].view(np.uint64))
    assert sim.itime == 12


def test_host_callback_writes_reach_the_device():
    sim = _sim()
    sim.initialize()

    @callback(stage="start", interval=1)
    def kick(s):                       # an external-field plugin in the reference's style
        for p in s.patches:
            p.fields.ez[:p.nx, :p.ny, :p.nz] += 1e9
            p.particles[1].uz[:] += 0.01

    sim.run(1, callbacks=[kick])
    assert sim.engine.view("ez")[10, 10, 10].item() != 0.0
    uz = sim.engine.download_species(1)["uz"]
    assert uz.size and np.all(np.abs(uz - 0.01) < 1e-3)

    with pytest.raises(NotImplementedError):
        sim.run(1, callbacks=[callback(stage="_interpolator")(lambda s: None)])


def test_moving_window_3d_matches_long_static_box():
    """MovingWindow on Simulation3D (parity unpinned, as in 2-D): translation invariance -- a 96-cell window
    following a pulse through vacuum and into a plasma block that is injected column by column, against a
    static 224-cell box; compared ahead of the window's open left edge"""
    from lambdapic_amd.simulation import MovingWindow
    ny, nz, pw = 24, 32, 16
    dx, dy, dz = LAM / 10, LAM / 5, LAM / 5
    nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C / LAM) ** 2 / constants.E_CHARGE ** 2

    def case(nx, cbs):
        sim = Simulation3D(nx, ny, nz, dx, dy, dz, npatch_x=nx // pw, cpml_thickness=4, random_seed=9, sort_interval=4,
                           block_particles=1024)
        dens = lambda x, y, z: np.where((x > 110 * dx) & (abs(y - sim.Ly / 2) < 5 * dy) & (abs(z - sim.Lz / 2) < 7 * dz),
                                        0.05 * nc, 0.0)
        sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=2))
        laser = GaussianLaser3D(a0=1.0, l0=LAM, w0=0.8e-6, ctau=0.5e-6, x0=1.2e-6)
        sim.run(200, callbacks=[laser] + cbs)
        return sim

    w = case(96, [MovingWindow(velocity=C, start_time=0.8 * 96 * dx / C)])
    s = case(224, [])
    shifts = w.window_shifts
    assert shifts >= 4
    off = shifts * pw
    assert w.engine.x0 == pytest.approx(off * dx, rel=1e-12)
    ng, margin = 3, 32
    for a in ("ey", "bz", "ex", "jx", "rho"):
        fw = w.engine.view(a)[ng + margin:ng + 96 - 4, ng:-ng, ng:-ng].cpu().numpy()
        fs = s.engine.view(a)[ng + off + margin:ng + off + 96 - 4, ng:-ng, ng:-ng].cpu().numpy()
        scale = np.abs(fs).max()
        if a in ("ey", "bz"):
            assert scale > 0
        if scale > 0:
            assert np.abs(fw - fs).max() <= 1e-8 * scale, a
    assert w.engine.diagnostics()["nalive"][0] > 500


def test_one_call_per_step_and_one_launch_for_all_species_equal_the_staged_loop():
    """3-D: ``lpa_step`` (two calls per step around the laser) with both species in ONE K1-3D launch
    (``lpa_push_deposit_tiled_multi_3d``) against (a) ``lpa_step`` with one launch per species and (b) the stage loop walked
    facade by facade: same fields to summation order, same particles per id"""
    def run(fused_step, fuse_species):
        sim = _sim(npatch_x=2, npatch_y=1, npatch_z=2)
        sim.initialize()
        sim.engine.fused_step, sim.engine.fuse_species = fused_step, fuse_species
        calls = {"n": 0}
        inner = sim.engine.step_stages

        def counted(dt, first, last, *more):
            calls["n"] += 1
            return inner(dt, first, last, *more)

        sim.engine.step_stages = counted
        sim.run(40, callbacks=[GaussianLaser3D(a0=3.0, l0=LAM, w0=1.0e-6, ctau=0.8e-6, x0=1.6e-6)])
        return sim, calls["n"]

    a, na = run(True, True)
    b, nb = run(True, False)
    c, nc = run(False, False)
    assert na == nb == 80 and nc == 0
    for other in (b, c):
        for name in ("ex", "ey", "ez", "bx", "by", "bz", "jx", "jy", "jz", "rho"):
            va, vo = a.engine.view(name), other.engine.view(name)
            assert (va - vo).abs().max().item() <= 1e-9 * max(vo.abs().max().item(), 1e-300), name
        assert a.engine.rho_steps == other.engine.rho_steps
        for i in range(2):
            da, do = a.engine.download_species(i), other.engine.download_species(i)
            oa, oo = np.argsort(da["_id"].view(np.uint64)), np.argsort(do["_id"].view(np.uint64))
            assert np.array_equal(da["_id"].view(np.uint64)[oa], do["_id"].view(np.uint64)[oo])
            for k in ("x", "y", "z", "ux", "uy", "uz"):
                assert np.abs(da[k][oa] - do[k][oo]).max() <= 1e-9 * max(np.abs(do[k]).max(), 1e-300), k
    assert a.engine.view("ey").abs().max().item() > 0 and a.engine.rho_steps["continuity"] > 20
