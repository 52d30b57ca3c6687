"""GPU parity of the resident engine (PicEngine2D): the whole step sequence on the device against
(a) the 40-step trace recorded with the reference's own kernels (golden G8) and (b) the oracle at
config C1 scale, plus size-independent properties at the full C2 size.

Stated tolerances (FP64, BASELINE north_star "field energy and total charge"):
  field energy  <= 1e-10 relative per step,  total charge <= 1e-12,  kinetic energy <= 1e-12.
(the deposit's atomic summation order and FMA contraction are the only differences; measured
agreement is ~1e-14.)
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import driver
from helpers import assert_close
from lambdapic_amd.engine import PicEngine2D
from lambdapic_amd.patch import make_patches_2d

pytestmark = pytest.mark.gpu


def _engine_from_patches(P, nx, ny, dx, dy, q, m, **kw):
    eng = PicEngine2D(nx, ny, dx, dy, n_guard=3, device="cuda:0", **kw)
    n = sum(p.particles[0].npart for p in P)
    eng.add_species(q, m, capacity=int(n * 1.2) + 1024)
    eng.species[0].upload([p.particles[0] for p in P])
    return eng


@pytest.mark.parametrize("tiled,sort_interval,order", [(False, 8, 1), (True, 1, 1), (True, 7, 1), (True, 5, 0)])
def test_g8_trace(golden, tiled, sort_interval, order):
    g = golden("g8_trace_2d")
    nx, ny, dx, dy = int(g["nx"]), int(g["ny"]), float(g["dx"]), float(g["dy"])
    P = make_patches_2d(nx, ny, dx, dy, int(g["npx"]), int(g["npy"]))
    for k, p in enumerate(P):
        q = p.particles[0]
        q.initialize(g[f"in{k}_x"].size)
        for a in ["x", "y", "ux", "uy", "uz", "inv_gamma", "w", "_id"]:
            getattr(q, a)[:] = g[f"in{k}_{a}"]
    eng = _engine_from_patches(P, nx, ny, dx, dy, float(g["q"]), float(g["m"]),
                               sort_interval=sort_interval, block_particles=1024, order=order)
    dt = float(g["dt"])
    fe, ke, ch, na = [], [], [], []
    for _ in range(int(g["nsteps"])):
        eng.step(dt, tiled=tiled)
        d = eng.diagnostics()
        fe.append(d["field_energy"]), ke.append(d["kinetic"][0]), ch.append(d["charge"]), na.append(d["nalive"][0])
    assert np.array_equal(na, g["trace_nalive"])
    np.testing.assert_allclose(fe, g["trace_field_energy"], rtol=1e-10)
    np.testing.assert_allclose(ke, g["trace_kinetic_energy"], rtol=1e-12)
    np.testing.assert_allclose(ch, g["trace_charge"], rtol=1e-12)
    # final fields, patch by patch, against the reference-kernel run
    eng.grid.download_patches(P)
    for k, p in enumerate(P):
        for a in ["ex", "ey", "ez", "bx", "by", "bz", "rho"]:
            fld = getattr(p.fields, a)[: p.nx, : p.ny]
            ref = g[f"final{k}_{a}"][: p.nx, : p.ny]
            assert_close(fld, ref, 1e-9, scale=np.max(np.abs(g[f"final{k}_{a}"])), what=f"{k} {a}")


def test_c1_scale_vs_oracle():
    """config C1 as BASELINE.json states it (256x256, 16 ppc, periodic thermal plasma, 200 steps) on both
    sides (the oracle runs 8x8 patches with OpenMP); field energy, charge, kinetic energy every 10 steps."""
    lam = 0.8e-6
    nx = ny = 256
    dx = dy = lam / 20
    c = 299792458.0
    dt = 0.95 / (c * np.sqrt(dx ** -2 + dy ** -2))
    q, m = -oracle.E_CHARGE, oracle.M_E
    nc = oracle.EPSILON_0 * m * (2 * np.pi * c / lam) ** 2 / q ** 2
    P = make_patches_2d(nx, ny, dx, dy, 8, 8)
    driver.load_uniform_plasma(P, 0, 16, nc, 0.0442, np.random.default_rng(20260722))
    eng = _engine_from_patches(P, nx, ny, dx, dy, q, m, sort_interval=8)
    ks = driver.oracle_kernels()
    nsteps = 200
    for it in range(nsteps):
        driver.step(P, ks, dt, [(q, m)], do_sort=False)
        eng.step(dt)
        if it % 10 == 9 or it == nsteps - 1:
            d = eng.diagnostics()
            assert d["field_energy"] == pytest.approx(driver.field_energy(P), rel=1e-10)
            assert d["charge"] == pytest.approx(driver.total_charge(P), rel=1e-12)
            assert d["kinetic"][0] == pytest.approx(driver.kinetic_energy(P, 0, m), rel=1e-12)
            assert d["nalive"][0] == nx * ny * 16


def test_tiled_equals_global_and_conserves_charge_full_size():
    """size-independent properties at C2 geometry (1024x1024, here 16 ppc to bound memory/time):
    the tiled kernel and the global-atomics kernel agree; total deposited charge equals N q w;
    the discrete continuity equation holds to round-off on every node (charge-conserving
    deposition + guard fold)."""
    lam = 0.8e-6
    nx = ny = 1024
    dx = dy = lam / 20
    c = 299792458.0
    dt = 0.95 / (c * np.sqrt(dx ** -2 + dy ** -2))
    q, m = -oracle.E_CHARGE, oracle.M_E
    nc = oracle.EPSILON_0 * m * (2 * np.pi * c / lam) ** 2 / q ** 2
    ppc = 16
    n = nx * ny * ppc

    def make(tiled):
        eng = PicEngine2D(nx, ny, dx, dy, device="cuda:0", sort_interval=4)
        eng.add_species(q, m, capacity=n + 1024)
        s = eng.species[0].cset
        cell = torch.arange(n, device="cuda:0") // ppc
        g2 = torch.Generator(device="cuda:0").manual_seed(7)
        s.arr("x")[:n] = ((cell // ny).double() + torch.rand(n, device="cuda:0", dtype=torch.float64, generator=g2) - 0.5) * dx
        s.arr("y")[:n] = ((cell % ny).double() + torch.rand(n, device="cuda:0", dtype=torch.float64, generator=g2) - 0.5) * dy
        for a in ("ux", "uy", "uz"):
            s.arr(a)[:n] = torch.randn(n, device="cuda:0", dtype=torch.float64, generator=g2) * 0.0442
        s.arr("inv_gamma")[:n] = 1.0 / torch.sqrt(1 + s.arr("ux")[:n] ** 2 + s.arr("uy")[:n] ** 2 + s.arr("uz")[:n] ** 2)
        s.arr("w")[:n] = nc * dx * dy / ppc
        s.id[:n] = torch.arange(n, device="cuda:0")
        eng.species[0].n = n
        for _ in range(6):
            eng.step(dt, tiled=tiled)
        return eng

    a, b = make(True), make(False)
    da, db = a.diagnostics(), b.diagnostics()
    assert da["nalive"][0] == n
    assert da["charge"] == pytest.approx(n * q * nc * dx * dy / ppc, rel=1e-12)
    assert da["field_energy"] == pytest.approx(db["field_energy"], rel=1e-10)
    assert da["kinetic"][0] == pytest.approx(db["kinetic"][0], rel=1e-12)
    for name in ("ex", "ey", "ez", "bz", "rho", "jx"):
        va, vb = a.grid.view(name), b.grid.view(name)
        scale = vb.abs().max().item()
        assert (va - vb).abs().max().item() <= 1e-9 * scale, name
    # discrete continuity of the Esirkepov deposit (the property the scheme exists for):
    # (rho_n - rho_{n-1})/dt + (jx[i,j]-jx[i-1,j])/dx + (jy[i,j]-jy[i,j-1])/dy = 0 on every node,
    # periodic images folded.  rho_{n-1} is the charge deposited by the previous step.
    g = a.grid
    s = slice(3, 3 + nx)
    rho_prev = g.view("rho")[s, s].clone()
    a.step(dt, tiled=True)
    rho, jx, jy = g.view("rho")[s, s], g.view("jx")[s, s], g.view("jy")[s, s]
    res = (rho - rho_prev) / dt + (jx - torch.roll(jx, 1, 0)) / dx + (jy - torch.roll(jy, 1, 1)) / dy
    assert res.abs().max().item() <= 1e-10 * rho.abs().max().item() / dt


@pytest.mark.k1_variants
def test_in_kernel_reseat_keeps_particles_and_physics():
    """the in-kernel cell-index sort (`lpa_tiling.slot_class`, off by default) permutes particles between slots
    inside a work block every step: nothing may be lost, doubled or detached from its id, and the physics must not
    notice (same traces and fields as without it, to summation order); the slot classes must keep matching the
    particles' rows after many steps (that is what the re-seating is for)"""
    lam = 0.8e-6
    nx, ny, ppc = 64, 96, 24
    dx = dy = lam / 20
    c = 299792458.0
    dt = 0.95 / (c * np.sqrt(dx ** -2 + dy ** -2))
    q, m = -oracle.E_CHARGE, oracle.M_E
    n = nx * ny * ppc

    def run(reseat):
        eng = PicEngine2D(nx, ny, dx, dy, device="cuda:0", sort_interval=12, block_particles=2048)
        eng.reseat = reseat
        eng.add_species(q, m, capacity=n + 64)
        s = eng.species[0].cset
        g2 = torch.Generator(device="cuda:0").manual_seed(11)
        cell = torch.arange(n, device="cuda:0") // ppc
        r = lambda: torch.rand(n, device="cuda:0", dtype=torch.float64, generator=g2)
        s.arr("x")[:n] = ((cell // ny).double() + r() - 0.5) * dx
        s.arr("y")[:n] = ((cell % ny).double() + r() - 0.5) * dy
        for a in ("ux", "uy", "uz"):
            s.arr(a)[:n] = torch.randn(n, device="cuda:0", dtype=torch.float64, generator=g2) * 0.2   # hot: many movers
        s.arr("inv_gamma")[:n] = 1.0 / torch.sqrt(1 + s.arr("ux")[:n] ** 2 + s.arr("uy")[:n] ** 2 + s.arr("uz")[:n] ** 2)
        s.arr("w")[:n] = 1.7e27 * dx * dy / ppc
        s.id[:n] = torch.arange(n, device="cuda:0") + 1000
        eng.species[0].n = n
        tr = []
        for _ in range(30):
            eng.step(dt)
            d = eng.diagnostics()
            tr.append([d["field_energy"], d["charge"], d["kinetic"][0], d["nalive"][0]])
        return eng, np.array(tr)

    a, ta = run(True)
    b, tb = run(False)
    assert a.species[0].tiling.slot_class and not b.species[0].tiling.slot_class
    assert np.array_equal(ta[:, 3], tb[:, 3]) and ta[-1, 3] == n
    np.testing.assert_allclose(ta[:, 0], tb[:, 0], rtol=1e-10)
    np.testing.assert_allclose(ta[:, 1], tb[:, 1], rtol=1e-12)
    np.testing.assert_allclose(ta[:, 2], tb[:, 2], rtol=1e-12)
    for name in ("ex", "ey", "ez", "bz", "rho"):
        va, vb = a.grid.view(name), b.grid.view(name)
        assert (va - vb).abs().max().item() <= 1e-9 * vb.abs().max().item(), name
    # the same particles with the same attributes, whatever slot they sit in now
    da, db = a.species[0].download(), b.species[0].download()
    oa, ob = np.argsort(da["_id"].view(np.uint64)), np.argsort(db["_id"].view(np.uint64))
    assert np.array_equal(da["_id"].view(np.uint64)[oa], np.arange(n, dtype=np.uint64) + 1000)
    for k in ("x", "y", "ux", "uy", "uz", "inv_gamma", "w"):
        np.testing.assert_allclose(da[k][oa], db[k][ob], rtol=1e-9, atol=1e-9 * np.abs(db[k]).max(), err_msg=k)


# ---- rho from the continuity equation (LPA_PUSH_NO_RHO + lpa_rho_continuity) against the deposited rho ------------
@pytest.mark.parametrize("bc", ["periodic", "pml", "x-pml"])
def test_rho_from_continuity_matches_deposited_rho(bc):
    """Two engines, same particles, same steps: one deposits rho in every step (the reference's kernel,
    current/current_deposit.h:180), the other only on sort steps and advances it with the discrete continuity equation in
    between (LPA_PUSH_NO_RHO, lpa_rho_continuity; rho.py).  rho agrees on every node of the padded array to 1e-12 of its
    maximum at every step.  'pml' / 'x-pml' (x open, y periodic: the absorbed particles' charge crosses the y fold):
    hot electrons are absorbed at the open faces and their charge has to leave rho one step after their last deposit"""
    from lambdapic_amd.particles import ParticlesBase
    nx, ny = 48, 64
    dx = dy = 4e-8
    c = 299792458.0
    dt = 0.95 / (c * np.sqrt(dx ** -2 + dy ** -2))
    per = "periodic"
    bcs = {"periodic": dict(xmin=per, xmax=per, ymin=per, ymax=per), "pml": dict(xmin="pml", xmax="pml", ymin="pml", ymax="pml"),
           "x-pml": dict(xmin="pml", xmax="pml", ymin=per, ymax=per)}[bc]
    qe, me = -oracle.E_CHARGE, oracle.M_E
    x_lo, x_hi = (0, nx) if bc == "periodic" else (7, 41)      # 3 cells inside the absorbing bounds
    y_lo, y_hi = (7, 57) if bc == "pml" else (0, ny)

    def block(rng, ppc, uth):
        cells = np.array([(i, j) for i in range(x_lo, x_hi) for j in range(y_lo, y_hi)])
        n = len(cells) * ppc
        p = ParticlesBase(0, 0)
        p.initialize(n)
        pos = (np.repeat(cells, ppc, axis=0) + rng.uniform(-0.5, 0.5, (n, 2))) * dx
        p.x[:], p.y[:] = pos.T
        for a in ("ux", "uy", "uz"):
            getattr(p, a)[:] = rng.normal(size=n) * uth
        p.inv_gamma[:] = 1 / np.sqrt(1 + p.ux ** 2 + p.uy ** 2 + p.uz ** 2)
        p.w[:] = 1e27 * dx * dy / ppc
        return p

    def make(cont):
        eng = PicEngine2D(nx, ny, dx, dy, device="cuda:0", sort_interval=6, block_particles=1024,
                          boundary_conditions=bcs, cpml_thickness=4)
        eng.rho_continuity = cont
        eng.overflow_sort_fraction = 0       # the fixed sort schedule (the step counts asserted below); these electrons are hot
        rng = np.random.default_rng(9)
        for k, (q, m, ppc, uth) in enumerate(((qe, me, 12, 0.4), (-qe, 1836 * me, 6, 0.002))):
            p = block(rng, ppc, uth)
            eng.add_species(q, m, capacity=2 * p.npart)
            eng.species[k].upload([p])
        return eng

    a, b = make(True), make(False)
    n_e = a.species[0].n
    assert a.rho_mode() == "continuity" and b.rho_mode() == "deposited"
    for it in range(20):
        a.step(dt)
        b.step(dt)
        ra, rb = a.grid.view("rho"), b.grid.view("rho")
        err = (ra - rb).abs().max().item() / rb.abs().max().item()
        assert err <= 1e-12, (it, err)
        for name in ("jx", "jy", "jz", "ex", "ey", "bz"):
            va, vb = a.grid.view(name), b.grid.view(name)
            assert (va - vb).abs().max().item() <= 1e-11 * vb.abs().max().item(), (it, name)
    da, db = a.diagnostics(), b.diagnostics()
    assert da["nalive"] == db["nalive"]
    assert a.rho_steps == {"anchor": 4, "continuity": 16} and b.rho_steps["continuity"] == 0
    if bc == "periodic":
        assert da["nalive"][0] == n_e
    else:
        assert da["nalive"][0] < n_e - 100           # electrons were absorbed: the correction path ran


# ---- inv_gamma recomputed instead of streamed (LPA_PUSH_NO_IG + lpa_refresh_inv_gamma) ------------------------------------
@pytest.mark.parametrize("bc", ["periodic", "pml"])
def test_lazy_inv_gamma_matches_the_stored_one(bc):
    """The resident engine's fused kernels neither load nor store inv_gamma (engine.lazy_inv_gamma, the default): 1 / gamma
    is recomputed from the momenta with the function the Boris rotation ends with (unified_pusher_2d.c:50), so a run
    that streams inv_gamma like the reference (:59-60, lazy_inv_gamma = False) and a run that does not are the same run
    up to the order of the deposit's atomics -- every particle attribute incl. the refreshed inv_gamma to 1e-12, and the
    refreshed inv_gamma equals 1 / sqrt(1 + u^2) of the downloaded momenta to 2 ulp.  Sorted (tiled kernel + second pass + overflow list) and unsorted (global kernel) steps, absorption, and a
    mid-run download (the refresh must not disturb the run)."""
    from lambdapic_amd.particles import ParticlesBase
    nx, ny = 48, 64
    dx = dy = 4e-8
    c = 299792458.0
    dt = 0.95 / (c * np.sqrt(dx ** -2 + dy ** -2))
    per = "periodic"
    bcs = dict(xmin=per, xmax=per, ymin=per, ymax=per) if bc == per else dict(xmin="pml", xmax="pml", ymin="pml", ymax="pml")
    lo, hi = (0, nx) if bc == per else (8, 40)
    rng = np.random.default_rng(4)
    cells = np.array([(i, j) for i in range(lo, hi) for j in range(lo, hi + 16)])
    ppc = 10
    n = len(cells) * ppc
    p = ParticlesBase(0, 0)
    p.initialize(n)
    p.x[:], p.y[:] = ((np.repeat(cells, ppc, axis=0) + rng.uniform(-0.5, 0.5, (n, 2))) * dx).T
    for a in ("ux", "uy", "uz"):
        getattr(p, a)[:] = rng.normal(size=n) * 0.5
    p.inv_gamma[:] = 1 / np.sqrt(1 + p.ux ** 2 + p.uy ** 2 + p.uz ** 2)
    p.w[:] = 1e27 * dx * dy / ppc

    def make(lazy):
        eng = PicEngine2D(nx, ny, dx, dy, device="cuda:0", sort_interval=5, block_particles=1024, boundary_conditions=bcs,
                          cpml_thickness=4)
        eng.lazy_inv_gamma = lazy
        eng.add_species(-oracle.E_CHARGE, oracle.M_E, capacity=2 * n)
        eng.species[0].upload([p])
        # both runs start from the kernel's own 1 / gamma (numpy's differs from it in the last bit for some momenta)
        eng.species[0].ig_stale = True
        eng.species[0].refresh_inv_gamma()
        return eng

    a, b = make(True), make(False)
    for it in range(12):
        a.step(dt)
        b.step(dt)
        assert a.species[0].ig_stale and not b.species[0].ig_stale
        if it in (3, 11):
            da, db = a.species[0].download(), b.species[0].download()
            assert da["x"].size == db["x"].size and (bc == per or da["x"].size < n or it == 3)
            oa, ob = np.argsort(da["_id"].view(np.uint64)), np.argsort(db["_id"].view(np.uint64))
            assert np.array_equal(da["_id"][oa].view(np.uint64), db["_id"][ob].view(np.uint64))
            for k, unit in (("x", dx), ("y", dy), ("ux", 1.0), ("uy", 1.0), ("uz", 1.0), ("inv_gamma", 1.0), ("w", 1e27 * dx * dy)):
                # (two runs of the SAME build differ like this too: the order of the deposit's atomics moves the fields
                # in their last bits)
                assert np.abs(da[k][oa] - db[k][ob]).max() <= 1e-12 * unit, (it, k)
            u2 = da["ux"] ** 2 + da["uy"] ** 2 + da["uz"] ** 2
            np.testing.assert_allclose(da["inv_gamma"], 1 / np.sqrt(1 + u2), rtol=4e-16)
        for name in ("jx", "jy", "jz", "rho", "ex", "bz"):
            va, vb = a.grid.view(name), b.grid.view(name)
            assert (va - vb).abs().max().item() <= 1e-12 * max(vb.abs().max().item(), 1e-300), (it, name)
    ka, kb = a.diagnostics(), b.diagnostics()
    assert ka["nalive"] == kb["nalive"] and ka["kinetic"][0] == pytest.approx(kb["kinetic"][0], rel=1e-14)


# ---- the sort interval follows the overflow list ---------------------------------------------------------------------------
def test_sort_interval_follows_the_overflow_list():
    """``sort_interval`` is the LONGEST interval: the first sort of a store starts the species where 2.5-sigma particles
    would outrun the tile margin, and every later sort shortens / lengthens it by what the push before it had to send to
    the overflow list (PicEngine2D.overflow_sort_fraction; nothing is read between sorts).  A hot plasma ends up sorting
    every few steps, a 1 keV one keeps the full interval -- and WHEN a store is sorted changes nothing physical: the hot
    run equals the same run on the fixed interval to the order of the deposit's atomics."""
    from lambdapic_amd.particles import ParticlesBase
    nx = ny = 96
    dx = dy = 4e-8
    c = 299792458.0
    dt = 0.95 / (c * np.sqrt(dx ** -2 + dy ** -2))

    def run(uth, fraction, steps):
        rng = np.random.default_rng(12)
        ppc = 16
        n = nx * ny * ppc
        p = ParticlesBase(0, 0)
        p.initialize(n)
        cells = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij"), -1).reshape(-1, 2)
        p.x[:], p.y[:] = ((np.repeat(cells, ppc, axis=0) + rng.uniform(-0.5, 0.5, (n, 2))) * dx).T
        for a in ("ux", "uy", "uz"):
            getattr(p, a)[:] = rng.normal(size=n) * uth
        p.inv_gamma[:] = 1 / np.sqrt(1 + p.ux ** 2 + p.uy ** 2 + p.uz ** 2)
        p.w[:] = 1e26 * dx * dy / ppc
        eng = PicEngine2D(nx, ny, dx, dy, device="cuda:0", sort_interval=20, block_particles=1024)
        eng.overflow_sort_fraction = fraction
        eng.add_species(-oracle.E_CHARGE, oracle.M_E, capacity=2 * n)
        eng.species[0].upload([p])
        sorts = []
        orig = eng.sort
        eng.sort = lambda i: (sorts.append(eng.species[i].steps_since_sort), orig(i))[1]
        for _ in range(steps):
            eng.step(dt)
        return eng, sorts

    cold, s_cold = run(0.0442, 0.003, 45)
    assert cold.species[0].sort_interval_now == 20 and s_cold[1:] == [20, 20]
    hot, s_hot = run(0.6, 0.003, 45)
    assert hot.species[0].sort_interval_now <= 6 and max(s_hot[1:]) <= 8 and len(s_hot) >= 8
    ws = hot._sort_ws(hot.species[0])
    assert int(ws["counters"][0].item()) <= 0.02 * hot.species[0].n_sorted       # the list stays short
    fixed, s_fixed = run(0.6, 0.0, 45)
    assert s_fixed[1:] == [20, 20]
    for name in ("ex", "ey", "bz", "jx", "jy", "jz", "rho"):
        a, b = hot.grid.view(name), fixed.grid.view(name)
        assert (a - b).abs().max().item() <= 1e-10 * b.abs().max().item(), name
    dh, df = hot.diagnostics(), fixed.diagnostics()
    assert dh["nalive"] == df["nalive"] and dh["kinetic"][0] == pytest.approx(df["kinetic"][0], rel=1e-12)
