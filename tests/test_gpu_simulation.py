"""The facade / stage-callback mirror (lambdapic_amd.simulation) driven the way the reference's
integration tests drive ``Simulation``: real simulation, callbacks that read (and write) the patch
mirrors, no mocks (reference tests/test_numerical_heating.py, tests/test_callback.py,
docs/source/write_callbacks.rst 'External fields')."""
import numpy as np
import pytest
import torch

import oracle
from oracle import driver
from lambdapic_amd import constants
from lambdapic_amd.patch import make_patches_2d
from lambdapic_amd.simulation import Simulation, Species, callback

pytestmark = pytest.mark.gpu

LAMBDA = 0.8e-6
C = 299792458.0


def _sim(nx=64, ny=64, npx=2, npy=2, ppc=16, seed=3, **kw):
    dx = dy = LAMBDA / 20
    bc = {k: "periodic" for k in ("xmin", "xmax", "ymin", "ymax")}
    sim = Simulation(nx, ny, dx, dy, npatch_x=npx, npatch_y=npy, random_seed=seed, boundary_conditions=bc, **kw)
    nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C / LAMBDA) ** 2 / constants.E_CHARGE ** 2
    sim.add_species(Species("electron", charge=-1, mass=1, density=nc, ppc=ppc, momentum_sigma=0.0442))
    return sim, nc


def test_numerical_heating_energy_conservation():
    """reference tests/test_numerical_heating.py:103-133: 64x64 periodic thermal plasma, total
    (field + kinetic) energy conserved to < 1 %; energies computed by an 'end' callback from the
    patch mirrors exactly like the reference's get_field_energy / get_kinetic_energy (:19-50)"""
    sim, _ = _sim(ppc=16)
    hist = []

    @callback("end", interval=10)
    def energies(s):
        ef = 0.0
        ek = 0.0
        for p in s.patches:
            f = p.fields
            sl = (slice(0, f.nx), slice(0, f.ny))
            ef += (0.5 * constants.EPSILON_0 * (f.ex[sl] ** 2 + f.ey[sl] ** 2 + f.ez[sl] ** 2).sum()
                   + 0.5 / constants.MU_0 * (f.bx[sl] ** 2 + f.by[sl] ** 2 + f.bz[sl] ** 2).sum()) * f.dx * f.dy
            q = p.particles[0]
            live = ~q.is_dead
            ek += (q.w[live] * (1 / q.inv_gamma[live] - 1)).sum() * constants.M_E * C ** 2
        hist.append((s.itime, ef, ek))

    sim.run(101, callbacks=[energies])
    tot = np.array([h[1] + h[2] for h in hist])
    assert len(hist) == 11
    assert np.all(np.abs(tot / tot[0] - 1) < 0.01)
    d = sim.engine.diagnostics()     # the mirror-based numbers equal the device reductions
    assert hist[-1][1] == pytest.approx(d["field_energy"], rel=1e-12)
    assert hist[-1][2] == pytest.approx(d["kinetic"][0], rel=1e-12)


def test_stage_order_and_intervals():
    """callbacks fire at their stage, in the reference's stage order, honouring int intervals
    (reference tests/test_callback.py; STAGES simulation.py:170-184)"""
    sim, _ = _sim(nx=32, ny=32, npx=1, npy=1, ppc=4)
    seen = []

    def mk(stage, interval=1):
        @callback(stage, interval)
        def cb(s):
            seen.append((s.itime, stage))
        return cb

    def plain(s):                      # undecorated callables default to stage 'end'
        seen.append((s.itime, "plain"))

    cbs = [mk("end"), mk("start"), mk("maxwell_1"), mk("current_deposition"), mk("_laser", 2), mk("init"),
           mk("final"), plain]
    sim.run(3, callbacks=cbs)
    per_step = lambda it: [st for t, st in seen if t == it]
    assert seen[0] == (0, "init") and seen[-1][1] == "final"
    assert per_step(1) == ["start", "maxwell_1", "current_deposition", "end", "plain"]
    assert "_laser" in per_step(0) and "_laser" in per_step(2) and "_laser" not in per_step(1)
    with pytest.raises(ValueError):
        sim.run(1, callbacks=[mk("no_such_stage")])
    with pytest.raises(ValueError):
        sim.run(nsteps=1, sim_time=1e-15)


def test_interpolator_callback_external_field_vs_oracle():
    """a callback in a pusher stage forces the split path (simulation.py:896-911); an external
    field added to ex_part at '_interpolator' (docs write_callbacks.rst 'External fields') must act
    exactly as in the CPU restatement of that path"""
    sim, nc = _sim(nx=32, ny=32, npx=2, npy=1, ppc=8, seed=11, sort_interval=4)
    sim.initialize()
    sim.download()          # particles are loaded on the device; the mirrors fill on download
    # same initial particles for the oracle
    P = make_patches_2d(32, 32, sim.dx, sim.dy, 2, 1)
    for p, m in zip(P, sim.patches):
        q, s = p.particles[0], m.particles[0]
        q.initialize(s.npart)
        for a in ("x", "y", "ux", "uy", "uz", "inv_gamma", "w", "_id"):
            getattr(q, a)[:] = getattr(s, a)
    E0 = 3e11

    @callback("_interpolator")
    def external(s):
        for p in s.patches:
            q = p.particles[s.ispec]
            q.ex_part[~q.is_dead] += E0 * np.sin(2 * np.pi * q.y[~q.is_dead] / s.Ly)

    def hook(patches, ispec):
        for p in patches:
            q = p.particles[ispec]
            q.ex_part[~q.is_dead] += E0 * np.sin(2 * np.pi * q.y[~q.is_dead] / (32 * sim.dy))

    nsteps = 6
    sim.run(nsteps, callbacks=[external])
    qm = [(-constants.E_CHARGE, constants.M_E)]
    for _ in range(nsteps):
        driver.step_split(P, sim.dt, qm, hook)
    d = sim.engine.diagnostics()
    assert d["field_energy"] == pytest.approx(driver.field_energy(P), rel=1e-10)
    assert d["kinetic"][0] == pytest.approx(driver.kinetic_energy(P, 0, constants.M_E), rel=1e-12)
    assert d["charge"] == pytest.approx(driver.total_charge(P), rel=1e-12)
    sim.download()
    for m, p in zip(sim.patches, P):
        for a in ("ex", "ey", "bz", "jx", "rho"):
            A, B = getattr(m.fields, a)[: p.nx, : p.ny], getattr(p.fields, a)[: p.nx, : p.ny]
            assert np.abs(A - B).max() <= 1e-10 * np.abs(getattr(p.fields, a)).max(), a


def test_mirror_round_trip_is_lossless():
    """download -> upload through the patch mirrors must not change the device state"""
    sim, _ = _sim(nx=48, ny=32, npx=3, npy=2, ppc=4, seed=5)
    sim.run(5)
    before = sim.engine.diagnostics()
    ex0 = sim.engine.grid.view("ex").clone()
    sim.download()
    sim.upload()
    after = sim.engine.diagnostics()
    # (the energy diagnostic sums with atomics: the last bit depends on their order)
    assert before["field_energy"] == pytest.approx(after["field_energy"], rel=1e-14)
    assert before["nalive"] == after["nalive"]
    assert before["kinetic"][0] == pytest.approx(after["kinetic"][0], rel=1e-14)
    assert (sim.engine.grid.view("ex") - ex0).abs().max().item() == 0.0
    sim.run(3)                                    # and the run continues from the uploaded state
    assert sim.itime == 8


def test_extract_species_density_on_device():
    """ExtractSpeciesDensity (callback/utils.py:240-293): the density of each species from the rho it adds
    at its 'current_deposition' stage; the currents of every species still reach the interior (the
    callback syncs in the middle of the species loop, simulation.py:991 un-syncs after each deposit)"""
    from lambdapic_amd.callbacks import ExtractSpeciesDensity
    from lambdapic_amd.simulation import Simulation, Species
    nx = ny = 64
    dx = dy = 4e-8
    bc = {k: "periodic" for k in ("xmin", "xmax", "ymin", "ymax")}
    sim = Simulation(nx, ny, dx, dy, boundary_conditions=bc, random_seed=2)
    n1 = lambda x, y: np.where(x < 32 * dx, 1.0e25, 3.0e25)
    n2 = lambda x, y: np.full_like(x, 2.0e25)
    e = Species("e", charge=-1, mass=1, density=n1, ppc=8, momentum_sigma=0.01)
    p = Species("p", charge=2, mass=3672.0, density=n2, ppc=4)
    sim.add_species([e, p])
    sim.initialize()
    de, dp = ExtractSpeciesDensity(sim, e, interval=1), ExtractSpeciesDensity(sim, p, interval=1)
    sim.run(3, callbacks=[de, dp])
    ne, np_ = de.density, dp.density
    assert ne.shape == (nx, ny)
    # TSC-smoothed loading: means per half box, total particle number exact
    assert ne[4:28].mean() == pytest.approx(1.0e25, rel=0.05) and ne[36:60].mean() == pytest.approx(3.0e25, rel=0.05)
    assert np_.mean() == pytest.approx(2.0e25, rel=1e-9)
    assert ne.sum() * dx * dy == pytest.approx(2.0e25 * nx * ny * dx * dy, rel=1e-9)
    # total charge on the grid = sum over species (nothing was lost in the guards)
    d = sim.engine.diagnostics()
    want = (-1 * ne.sum() + 2 * np_.sum()) * dx * dy * 1.602176634e-19
    assert d["charge"] == pytest.approx(want, rel=1e-9)


def test_set_momentum_and_temperature_on_device():
    """SetMomentum / SetTemperature / SetMomentumAndTemperature (callback/utils.py:842-1049) at stage 'init':
    bulk momentum exact, thermal spread with the Maxwell-Juettner mean energy, inv_gamma consistent"""
    from scipy.special import kn
    from lambdapic_amd.callbacks import SetMomentum, SetMomentumAndTemperature
    sim, _ = _sim(nx=64, ny=64, npx=1, npy=1, ppc=16, seed=4)
    e = sim.species[0]
    e.momentum_sigma = 0.0
    sim.run(0, callbacks=[SetMomentum(e, [0.5, 0.0, -0.25])])
    sp = sim.engine.species[0]
    a = sp.cset.arr
    n = sp.n
    assert torch.all(a("ux")[:n] == 0.5) and torch.all(a("uz")[:n] == -0.25) and torch.all(a("uy")[:n] == 0.0)
    assert float((a("inv_gamma")[:n] - 1 / np.sqrt(1 + 0.25 + 0.0625)).abs().max()) < 1e-15
    T_eV = 51099.895                                        # theta = 0.1
    sim2, _ = _sim(nx=64, ny=64, npx=1, npy=1, ppc=16, seed=4)
    e2 = sim2.species[0]
    e2.momentum_sigma = 0.0
    sim2.run(0, callbacks=[SetMomentumAndTemperature(e2, [0.0, 0.0, 0.0], T_eV, seed=9)])
    b = sim2.engine.species[0].cset.arr
    ux, uy, uz, ig = (b(k)[:n] for k in ("ux", "uy", "uz", "inv_gamma"))
    gam = torch.sqrt(1 + ux * ux + uy * uy + uz * uz)
    assert float((ig * gam - 1).abs().max()) < 1e-14
    assert float(gam.mean()) - 1 == pytest.approx(3 * 0.1 + kn(1, 10.0) / kn(2, 10.0) - 1, rel=0.03)
    assert abs(float(ux.mean())) < 0.01


def test_random_seed_reproducible_loading():
    """reference tests/test_random_seed.py: the same seed loads the same particles, another seed different
    ones (the generator of a patch is seeded from (seed, species, patch origin), so the loading does not
    depend on the number of ranks either: tests/test_gpu_multirank.py compares 1 and 2 ranks)"""
    def load(seed, npx):
        sim, _ = _sim(nx=64, ny=32, npx=npx, npy=1, ppc=4, seed=seed)
        sim.initialize()
        sp = sim.engine.species[0]
        a = sp.cset.arr
        return torch.stack([a(k)[: sp.n] for k in ("x", "y", "ux", "uy", "uz")]).cpu().numpy()
    a, b, c = load(7, 2), load(7, 2), load(8, 2)
    assert np.array_equal(a, b)
    assert a.shape == c.shape and not np.array_equal(a, c)


def test_one_call_per_step_equals_the_staged_loop():
    """``lpa_step`` (one host call per step, two with a '_laser' callback in between) against the same run walked stage
    by stage through the facades: laser-target with CPML on all sides, two species, a window that shifts, a device-native
    laser at '_laser', a host callback at 'end' every 7 steps (mirrors refreshed around it), one at 'maxwell_1' every 9
    (forces the staged path for that step).  Same fields to summation order, same particles per id."""
    from lambdapic_amd import _lib
    from lambdapic_amd.laser import GaussianLaser2D
    from lambdapic_amd.simulation import MovingWindow

    def run(fused):
        nx, ny = 128, 64
        dx = dy = LAMBDA / 16
        sim = Simulation(nx, ny, dx, dy, npatch_x=8, npatch_y=2, cpml_thickness=6, random_seed=11, sort_interval=6)
        nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C / LAMBDA) ** 2 / constants.E_CHARGE ** 2
        Ly = ny * dy
        dens = lambda x, y: np.where((x > 40 * dx) & (abs(y - Ly / 2) < 20 * dy), 0.5 * nc, 0.0)
        sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=4, momentum_sigma=0.02))
        sim.add_species(Species("p", charge=1, mass=1836.0, density=dens, ppc=2))
        sim.initialize()
        sim.engine.fused_step = fused
        calls = {"lpa_step": 0, "end": 0, "maxwell_1": 0}
        inner = sim.engine.step_stages

        def counted(dt, first, last, *more):
            calls["lpa_step"] += 1
            calls["deferred"] = calls.get("deferred", 0) + int(bool(more and more[0]))
            return inner(dt, first, last, *more)

        sim.engine.step_stages = counted

        @callback("end", interval=7)
        def probe(s):
            calls["end"] += 1
            assert s.patches[0].fields.ey.shape == (16 + 6, 32 + 6)

        @callback("maxwell_1", interval=9)
        def inner_cb(s):
            calls["maxwell_1"] += 1

        cbs = [GaussianLaser2D(a0=2.0, l0=LAMBDA, w0=1.2e-6, ctau=1.0e-6, x0=1.5e-6),
               MovingWindow(velocity=C, start_time=0.25 * sim.Lx / C), probe, inner_cb]
        sim.run(90, callbacks=cbs)
        return sim, calls

    a, ca = run(True)
    b, cb = run(False)
    assert cb["lpa_step"] == 0 and ca["end"] == cb["end"] == 13 and ca["maxwell_1"] == cb["maxwell_1"] == 10
    # 80 of the 90 steps went through lpa_step: two calls while the laser injects, one once it is disabled (the window
    # removed the x-min layer), none in the 10 steps with the 'maxwell_1' callback
    assert 80 <= ca["lpa_step"] <= 160 and a.window_shifts == b.window_shifts >= 2
    # ... and many of them left their second E half step to the next step's first (Simulation._can_defer_e2): never before a
    # step whose 'end' callback fires, never before a window shift
    assert ca.get("deferred", 0) >= 40
    for name in ("ex", "ey", "ez", "bx", "by", "bz", "jx", "jy", "jz", "rho"):
        va, vb = a.engine.grid.view(name), b.engine.grid.view(name)
        assert (va - vb).abs().max().item() <= 1e-9 * max(vb.abs().max().item(), 1e-300), name
    assert a.engine.grid.view("ey").abs().max().item() > 0
    assert a.engine.rho_steps == b.engine.rho_steps and a.engine.rho_steps["continuity"] > 40
    for sa, sb in zip(a.engine.species, b.engine.species):
        da, db = sa.download(), sb.download()
        oa, ob = np.argsort(da["_id"].view(np.uint64)), np.argsort(db["_id"].view(np.uint64))
        assert np.array_equal(da["_id"].view(np.uint64)[oa], db["_id"].view(np.uint64)[ob])
        for k in ("x", "y", "ux", "uy", "uz"):
            assert np.abs(da[k][oa] - db[k][ob]).max() <= 1e-9 * max(np.abs(db[k]).max(), 1e-300), k


def test_loading_rules_density_min_ppc_function_and_species_classes():
    """`core/patch/cpu.py:7-45`: a cell is loaded when its density exceeds ``density_min``, with ``int(ppc(x, y))``
    particles of weight n dx dy / ppc; `core/species.py:185-221`: Electron / Positron / Proton"""
    from lambdapic_amd.simulation import Electron, Positron, Proton
    nx = ny = 32
    dx = dy = 1e-7
    bc = {k: "periodic" for k in ("xmin", "xmax", "ymin", "ymax")}
    sim = Simulation(nx, ny, dx, dy, npatch_x=2, npatch_y=2, boundary_conditions=bc, random_seed=9)
    ramp = lambda x, y: 1e24 * (x / dx) + 0 * y                     # 0, 1e24, 2e24 ... along x
    e = Electron(density=ramp, ppc=lambda x, y: 1.9 + (y > 16 * dy) * 2, density_min=4.5e24)
    p = Proton(density=2e24, ppc=3)
    pos = Positron(name="e+", density=lambda x, y: 0 * x, ppc=4)   # no cell qualifies
    sim.add_species([e, p, pos])
    sim.initialize()
    assert (e.charge, e.mass, p.charge, pos.charge, pos.mass) == (-1, 1.0, 1, 1, 1.0)
    assert p.mass == pytest.approx(1836.152673, rel=1e-9) and p.name == "proton" and e.name == "electron"
    d = sim.engine.species[0].download()
    ix, iy = np.rint(d["x"] / dx).astype(int), np.rint(d["y"] / dy).astype(int)
    assert ix.min() == 5                                            # 4e24 is not > 4.5e24
    cnt = np.zeros((nx, ny), int)
    np.add.at(cnt, (ix, iy), 1)
    assert (cnt[:5] == 0).all() and (cnt[5:, :17] == 1).all() and (cnt[5:, 17:] == 3).all()
    assert np.allclose(d["w"], 1e24 * ix * dx * dy / cnt[ix, iy], rtol=1e-14)
    assert sim.engine.species[1].n == nx * ny * 3 and sim.engine.species[2].n == 0
    sim.run(2)
    assert sim.engine.diagnostics()["nalive"] == [len(d["x"]), nx * ny * 3, 0]


def test_part_eb_is_written_for_the_steps_a_host_callback_reads():
    """ex_part ... bz_part (the fields the last push saw, `core/particles.py:63-67`) cost the fused kernel six more
    attribute streams and keep inv_gamma in the stream: they are written in the pushes that precede a mirror-reading
    callback, not in every step of a run that registers one -- and what the callback reads is what an always-writing run
    shows it"""
    def run(always):
        sim, _ = _sim(nx=32, ny=32, ppc=8)
        sim.initialize()
        if always:
            sim._host_callback_near = lambda cbs, last: True
        seen, flags = [], []

        @callback("end", interval=7)
        def reader(s):
            parts = [p.particles[0] for p in s.patches]
            ids = np.concatenate([q._id.view(np.uint64)[~q.is_dead] for q in parts])
            o = np.argsort(ids)
            seen.append({a: np.concatenate([getattr(q, a)[~q.is_dead] for q in parts])[o]
                         for a in ("ex_part", "ey_part", "bz_part", "inv_gamma", "ux")})

        @callback("maxwell_2", interval=1)
        def watch(s):
            flags.append((s.itime, bool(s.engine.write_part_eb), bool(s.engine.species[0].ig_stale)))
        watch.device_native = True
        sim.run(16, callbacks=[reader, watch])
        return seen, flags
    a, flags = run(False)
    b, flags_b = run(True)
    assert [f[1] for f in flags] == [it % 7 in (0, 6) for it in range(16)]      # the trigger step and the one before it
    assert all(f[1] for f in flags_b)
    assert any(f[2] for f in flags) and not any(f[2] for f in flags_b)            # lazy inv_gamma ran in between
    assert len(a) == len(b) == 3
    for x, y in zip(a, b):
        for k in x:
            scale = max(np.abs(y[k]).max(), 1e-300)
            assert np.abs(x[k] - y[k]).max() <= 1e-9 * scale, k
    assert np.abs(a[-1]["ex_part"]).max() > 0 and np.abs(a[0]["ex_part"]).max() == 0     # (the first push saw E = 0)
