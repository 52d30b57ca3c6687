"""Pins the oracle (oracle/picoracle.c, oracle/sync.py, oracle/driver.py) to the golden vectors
recorded from the reference's own kernels (tests/golden/gen_golden.py).  CPU only.

Tolerances (FP64): the reference build contracts a*b+c into FMAs (gcc -O3 -march=native), the
oracle is built -ffp-contract=off, so agreement is to a few ulp of the dominant term:
  per-particle pushed quantities 1e-12 relative to the array's max; gathered fields 1e-11
  (27-term sums of O(1e12) random fields cancel); grid rho/J 1e-12 of the max entry;
  40-step energy / charge traces 1e-9 relative.
"""
import numpy as np
import pytest

import oracle
from oracle import driver, sync
from helpers import (PATTRS, PEB, assert_close, fields2d_from, fields3d_from, particles_from)
from lambdapic_amd.patch import make_patches_2d


def test_g1_fused_2d(golden):
    g = golden("g1_fused_2d")
    for k in range(int(g["npatches"])):
        f = fields2d_from(g, f"in{k}_", g[f"x0_{k}"], g[f"y0_{k}"])
        p = particles_from(g, f"in{k}_")
        oracle.unified_boris_pusher_cpu_2d([p], [f], 1, float(g["dt"]), float(g["q"]), float(g["m"]))
        for a in ["x", "y", "ux", "uy", "uz", "inv_gamma"]:
            assert_close(getattr(p, a), g[f"out{k}_{a}"], 1e-12, what=f"patch{k} {a}")
        alive = ~(p.is_dead | np.isnan(g[f"in{k}_x"]) | np.isnan(g[f"in{k}_y"]))
        for a in PEB:
            assert_close(getattr(p, a)[alive], g[f"out{k}_{a}"][alive], 1e-11, what=f"patch{k} {a}")
        for a in ["rho", "jx", "jy", "jz"]:
            assert_close(getattr(f, a), g[f"out{k}_{a}"], 1e-12, what=f"patch{k} {a}")


def test_g2_fused_3d(golden):
    g = golden("g2_fused_3d")
    f = fields3d_from(g, "in_")
    p = particles_from(g, "in_")
    oracle.unified_boris_pusher_cpu_3d([p], [f], 1, float(g["dt"]), float(g["q"]), float(g["m"]))
    for a in ["x", "y", "z", "ux", "uy", "uz", "inv_gamma"]:
        assert_close(getattr(p, a), g[f"out_{a}"], 1e-12, what=a)
    alive = ~(p.is_dead | np.isnan(g["in_x"]) | np.isnan(g["in_y"]) | np.isnan(g["in_z"]))
    for a in PEB:
        assert_close(getattr(p, a)[alive], g[f"out_{a}"][alive], 1e-11, what=a)
    for a in ["rho", "jx", "jy", "jz"]:
        assert_close(getattr(f, a), g[f"out_{a}"], 1e-12, what=a)


def test_g3_standalone_deposit(golden):
    g = golden("g3_deposit_2d")
    f = fields2d_from(g, "none_", g["x0"], g["y0"])
    p = particles_from(g, "in_")
    oracle.current_deposition_cpu_2d([f], [p], 1, float(g["dt"]), float(g["q"]))
    for a in ["rho", "jx", "jy", "jz"]:
        assert_close(getattr(f, a), g[f"out_{a}"], 1e-12, what=a)
    g = golden("g3_deposit_3d")
    f = fields3d_from(g, "none_")
    p = particles_from(g, "in_")
    oracle.current_deposition_cpu_3d([f], [p], 1, float(g["dt"]), float(g["q"]))
    for a in ["rho", "jx", "jy", "jz"]:
        assert_close(getattr(f, a), g[f"out_{a}"], 1e-12, what=a)


def test_g4_interpolation(golden):
    g = golden("g4_interp_2d")
    f = fields2d_from(g, "in_", g["x0"], g["y0"])
    p = particles_from(g, "in_", ["x", "y"])
    oracle.interpolation_patches_2d([p], [f], 1)
    for a in PEB:
        assert_close(getattr(p, a)[~p.is_dead], g[f"out_{a}"][~p.is_dead], 1e-11, what=a)
    g = golden("g4_interp_3d")
    f = fields3d_from(g, "in_")
    p = particles_from(g, "in_", ["x", "y", "z"])
    oracle.interpolation_patches_3d([p], [f], 1)
    for a in PEB:
        assert_close(getattr(p, a)[~p.is_dead], g[f"out_{a}"][~p.is_dead], 1e-11, what=a)


def test_g5_fdtd(golden):
    g = golden("g5_fdtd_2d")
    f = fields2d_from(g, "in_", 0.0, 0.0)
    oracle.update_efield_2d(f, float(g["dt"]))
    for a in ["ex", "ey", "ez"]:
        assert_close(getattr(f, a), g[f"outE_{a}"], 1e-14, what=a)
    oracle.update_bfield_2d(f, float(g["dt"]))
    for a in ["bx", "by", "bz"]:
        assert_close(getattr(f, a), g[f"outB_{a}"], 1e-14, what=a)
    g = golden("g5_fdtd_3d")
    f = fields3d_from({**{k: g[k] for k in g.files}, "x0": 0.0, "y0": 0.0, "z0": 0.0}, "in_")
    oracle.update_efield_3d(f, float(g["dt"]))
    for a in ["ex", "ey", "ez"]:
        assert_close(getattr(f, a), g[f"outE_{a}"], 1e-14, what=a)
    oracle.update_bfield_3d(f, float(g["dt"]))
    for a in ["bx", "by", "bz"]:
        assert_close(getattr(f, a), g[f"outB_{a}"], 1e-14, what=a)


def test_g6_sort_buckets(golden):
    """bucket counts / bounds are exact; the permutation inside a bucket is implementation
    defined, so compare the per-bucket multiset of particle ids (reference tests/test_sort.py:38-74)"""
    g = golden("g6_sort_2d")
    nx, ny, dx, dy = int(g["nx"]), int(g["ny"]), float(g["dx"]), float(g["dy"])
    Ly = ny * dy
    idx, cnt = oracle.bucket_index_2d(g["in_x"].copy(), g["in_y"].copy(), g["in_is_dead"].copy(),
                                      nx, 1, dx, Ly, float(g["x0"]) - dx / 2, float(g["y0"]) - dy / 2)
    assert np.array_equal(cnt.reshape(nx, 1), g["bucket_count"])
    bmin = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    assert np.array_equal(bmin.reshape(nx, 1), g["bucket_bound_min"])
    assert np.array_equal((bmin + cnt).reshape(nx, 1), g["bucket_bound_max"])
    order = np.argsort(idx, kind="stable")
    ids_mine, ids_ref = g["in__id"].view(np.uint64)[order], g["out__id"].view(np.uint64)
    for b in range(nx):
        lo, hi = bmin[b], bmin[b] + cnt[b]
        assert np.array_equal(np.sort(ids_mine[lo:hi]), np.sort(ids_ref[lo:hi]))
    assert int(g["nbuf_again"]) == 0


def _patches_from_g7(g):
    P = make_patches_2d(int(g["nx"]), int(g["ny"]), float(g["dx"]), float(g["dy"]),
                        int(g["npx"]), int(g["npy"]))
    for k, p in enumerate(P):
        for a in p.fields.attrs:
            getattr(p.fields, a)[...] = g[f"in{k}_{a}"]
    return P


def test_g7_sync_fields(golden):
    g = golden("g7_sync_2d")
    P = _patches_from_g7(g)
    fl = [p.fields for p in P]
    sync.sync_guard_fields_2d(fl, list(P), ["ex", "ey", "ez", "bx", "by", "bz"], 4, P.nx, P.ny, 3)
    sync.sync_currents_2d(fl, list(P), 4, P.nx, P.ny, 3)
    for k, p in enumerate(P):
        for a in ["ex", "ey", "ez", "bx", "by", "bz"]:
            assert np.array_equal(getattr(p.fields, a), g[f"out{k}_{a}"]), (k, a)
        for a in ["jx", "jy", "jz", "rho"]:
            # fold order differs from the reference's (face vs corner first): few-ulp sums
            assert_close(getattr(p.fields, a), g[f"out{k}_{a}"], 1e-15, what=f"{k} {a}")


def test_g7_sync_fields_c_twin(golden):
    """the OpenMP C form used by the CPU baseline gives the same arrays"""
    g = golden("g7_sync_2d")
    P = _patches_from_g7(g)
    fl = [p.fields for p in P]
    oracle.sync_guard_fields_2d_c(fl, list(P), ["ex", "ey", "ez", "bx", "by", "bz"], 4, P.nx, P.ny, 3)
    oracle.sync_currents_2d_c(fl, list(P), 4, P.nx, P.ny, 3)
    for k, p in enumerate(P):
        for a in ["ex", "ey", "ez", "bx", "by", "bz"]:
            assert np.array_equal(getattr(p.fields, a), g[f"out{k}_{a}"]), (k, a)
        for a in ["jx", "jy", "jz", "rho"]:
            assert_close(getattr(p.fields, a), g[f"out{k}_{a}"], 1e-15, what=f"{k} {a}")


def test_periodic_single_patch_twins():
    """the N-D single-patch periodic fill / fold (used by the 3-D engine test) equal the
    reference-pinned 2-D patch functions when the patch is its own neighbour"""
    rng = np.random.default_rng(4)
    P = make_patches_2d(12, 10, 1e-8, 1e-8, 1, 1)
    Q = make_patches_2d(12, 10, 1e-8, 1e-8, 1, 1)
    for a in P[0].fields.attrs:
        v = rng.normal(size=P[0].fields.shape)
        getattr(P[0].fields, a)[...] = v
        getattr(Q[0].fields, a)[...] = v
    E6 = ["ex", "ey", "ez", "bx", "by", "bz"]
    sync.sync_guard_fields_2d([P[0].fields], list(P), E6, 1, 12, 10, 3)
    sync.sync_currents_2d([P[0].fields], list(P), 1, 12, 10, 3)
    sync.periodic_guard_fill(Q[0].fields, E6)
    sync.periodic_current_fold(Q[0].fields)
    for a in E6:
        assert np.array_equal(getattr(P[0].fields, a), getattr(Q[0].fields, a)), a
    for a in ["jx", "jy", "jz", "rho"]:
        assert_close(getattr(Q[0].fields, a), getattr(P[0].fields, a), 1e-15, what=a)


def test_g7_sync_particles(golden):
    g = golden("g7_sync_2d")
    P = _patches_from_g7(g)
    for k, p in enumerate(P):
        q = p.particles[0]
        q.initialize(g[f"pin{k}_x"].size)
        for a in ["x", "y", "ux", "w", "_id"]:
            getattr(q, a)[:] = g[f"pin{k}_{a}"]
        q.is_dead[:] = g[f"pin{k}_is_dead"]
    alive = sync.sync_particles_2d(P, 0, float(g["dx"]), float(g["dy"]))
    assert np.array_equal(alive, g["npart_alive"])
    for k, p in enumerate(P):
        q = p.particles[0]
        assert q.npart == g[f"pout{k}_x"].size
        assert np.array_equal(q.is_dead, g[f"pout{k}_is_dead"])
        live = ~q.is_dead
        for a in ["x", "y", "ux", "w"]:
            assert np.array_equal(getattr(q, a)[live], g[f"pout{k}_{a}"][live]), (k, a)
        assert np.array_equal(q.id[live], g[f"pout{k}__id"].view(np.uint64)[live])


def test_g8_step_trace(golden):
    """the a0 stage order + all kernels in combination: 40-step traces of field energy, kinetic
    energy, total charge and current sums against the reference-kernel run."""
    g = golden("g8_trace_2d")
    P = make_patches_2d(int(g["nx"]), int(g["ny"]), float(g["dx"]), float(g["dy"]),
                        int(g["npx"]), int(g["npy"]))
    for k, p in enumerate(P):
        q = p.particles[0]
        q.initialize(g[f"in{k}_x"].size)
        for a in ["x", "y", "ux", "uy", "uz", "inv_gamma", "w", "_id"]:
            getattr(q, a)[:] = g[f"in{k}_{a}"]
    ks = driver.oracle_kernels()
    q_, m_, dt = float(g["q"]), float(g["m"]), float(g["dt"])
    tr = {k: [] for k in ("field_energy", "kinetic_energy", "charge", "nalive")}
    for _ in range(int(g["nsteps"])):
        driver.step(P, ks, dt, [(q_, m_)])
        tr["field_energy"].append(driver.field_energy(P))
        tr["kinetic_energy"].append(driver.kinetic_energy(P, 0, m_))
        tr["charge"].append(driver.total_charge(P))
        tr["nalive"].append(sum(int((~p.particles[0].is_dead).sum()) for p in P))
    assert np.array_equal(tr["nalive"], g["trace_nalive"])
    np.testing.assert_allclose(tr["field_energy"], g["trace_field_energy"], rtol=1e-9)
    np.testing.assert_allclose(tr["kinetic_energy"], g["trace_kinetic_energy"], rtol=1e-12)
    np.testing.assert_allclose(tr["charge"], g["trace_charge"], rtol=1e-12)
    for k, p in enumerate(P):
        for a in ["ex", "ey", "ez", "bx", "by", "bz", "rho"]:
            assert_close(getattr(p.fields, a), g[f"final{k}_{a}"], 1e-9, what=f"{k} {a}")


def test_known_answers_charge_and_current():
    """reference tests/core/current/test_current_deposition.py:328-369: for one particle,
    sum(rho) = q w/(dx dy) and sum(jx) = q w vx/(dx dy)."""
    from lambdapic_amd.fields import Fields2D
    from lambdapic_amd.particles import ParticlesBase
    dx = dy = 1e-8
    f = Fields2D(16, 16, dx, dy, 0.0, 0.0, 3)
    p = ParticlesBase(0, 0)
    p.initialize(1)
    rng = np.random.default_rng(3)
    p.x[:] = rng.uniform(3, 12) * dx
    p.y[:] = rng.uniform(3, 12) * dy
    p.ux[:], p.uy[:], p.uz[:] = 0.3, -0.2, 0.1
    p.inv_gamma[:] = 1 / np.sqrt(1 + 0.09 + 0.04 + 0.01)
    p.w[:] = 1e27 * dx * dy / 10
    q, dt = -1.602176634e-19, 1e-17
    oracle.current_deposition_cpu_2d([f], [p], 1, dt, q)
    n = p.w[0] / (dx * dy)
    v = np.array([p.ux[0], p.uy[0], p.uz[0]]) * p.inv_gamma[0] * 299792458.0
    assert f.rho.sum() == pytest.approx(q * n, rel=1e-10)
    assert f.jx.sum() == pytest.approx(q * n * v[0], rel=1e-10)
    assert f.jy.sum() == pytest.approx(q * n * v[1], rel=1e-10)
    assert f.jz.sum() == pytest.approx(q * n * v[2], rel=1e-10)


def test_g9_cpml_slab_vs_reference_patches(golden):
    """the slab-level CPML restatement (one field bag, per-axis coefficient arrays) against the
    reference's per-patch PML objects on 3x3 patches, 80 Maxwell stages with PML on all four
    sides; interior fields agree to round-off (FMA-free on both sides: 1e-13 of the max)"""
    from oracle import cpml
    from lambdapic_amd.fields import Fields2D
    g = golden("g9_cpml_2d")
    nx, ny, ng = int(g["nx"]), int(g["ny"]), int(g["ng"])
    dx, dy, dt = float(g["dx"]), float(g["dy"]), float(g["dt"])
    f = Fields2D(nx, ny, dx, dy, 0.0, 0.0, ng)
    for a in ("ex", "ey", "ez", "bx", "by", "bz", "jx", "jy", "jz"):
        getattr(f, a)[:nx, :ny] = g["in_" + a]
    pml = cpml.SlabPML2D(nx, ny, dx, dy, ("xmin", "xmax", "ymin", "ymax"), thickness=int(g["thickness"]))
    I = (slice(0, nx), slice(0, ny))
    en = []
    for it in range(int(g["nsteps"])):
        cpml.update_efield_cpml_2d(f, pml, 0.5 * dt)
        cpml.update_bfield_cpml_2d(f, pml, 0.5 * dt)
        if it == 2:
            f.jx[...] = 0; f.jy[...] = 0; f.jz[...] = 0
        en.append(float(np.sum(0.5 * oracle.EPSILON_0 * (f.ex[I] ** 2 + f.ey[I] ** 2 + f.ez[I] ** 2)
                               + 0.5 / oracle.MU_0 * (f.bx[I] ** 2 + f.by[I] ** 2 + f.bz[I] ** 2))) * dx * dy)
        if it + 1 in (10, 80):
            for a in ("ex", "ey", "ez", "bx", "by", "bz"):
                assert_close(getattr(f, a)[I], g[f"step{it + 1}_{a}"], 1e-12,
                             scale=np.abs(g[f"step10_{a}"]).max(), what=f"step {it + 1} {a}")
    np.testing.assert_allclose(en, g["trace_energy"], rtol=1e-12)
    assert en[-1] < 0.3 * en[0]           # the layer absorbs the outgoing pulse


def test_g10_laser_injection_kernel(golden):
    from oracle import cpml
    g = golden("g10_laser_2d")
    f = fields2d_from(g, "in_", 0.0, 0.0)
    cpml.laser_inject_2d(f, int(g["laserpos"]), float(g["dt"]), int(g["iy_start"]), int(g["iy_end"]),
                         g["ey_source"], g["ez_source"])
    for a in ("bx", "by", "bz"):
        assert_close(getattr(f, a), g["out_" + a], 1e-14, what=a)


def test_cpml_and_laser_3d_vs_reference(golden):
    """oracle/cpml.py 3-D restatement against the reference's six PML objects on one field bag and its
    3-D laser boundary kernel (g12)"""
    from oracle import cpml
    from lambdapic_amd.fields import Fields3D
    g = golden("g12_cpml_laser_3d")
    nx, ny, nz, ng, th = (int(g[k]) for k in ("nx", "ny", "nz", "ng", "thickness"))
    dx, dy, dz, dt = (float(g[k]) for k in ("dx", "dy", "dz", "dt"))
    f = Fields3D(nx, ny, nz, dx, dy, dz, 0.0, 0.0, 0.0, ng)
    for a in ("ex", "ey", "ez", "bx", "by", "bz", "jx", "jy", "jz"):
        getattr(f, a)[...] = g["in_" + a]
    pml = cpml.SlabPML3D(nx, ny, nz, dx, dy, dz, ["xmin", "xmax", "ymin", "ymax", "zmin", "zmax"], thickness=th)
    for it in range(3):
        cpml.update_efield_cpml_3d(f, pml, 0.5 * dt)
        cpml.update_bfield_cpml_3d(f, pml, 0.5 * dt)
        if it in (0, 2):
            for a in ("ex", "ey", "ez", "bx", "by", "bz"):
                assert_close(getattr(f, a), g[f"it{it}_{a}"], 1e-13, what=f"it{it} {a}")
    cpml.laser_inject_3d(f, int(g["laserpos"]), dt, int(g["iy_start"]), int(g["iy_end"]), int(g["iz_start"]),
                         int(g["iz_end"]), g["ey_source"], g["ez_source"])
    for a in ("bx", "by", "bz"):
        assert_close(getattr(f, a), g["lout_" + a], 1e-13, what="laser " + a)


def test_g13_sync_3d_single_periodic_patch(golden):
    """the reference's sync_guard_fields_3d / sync_currents_3d on one patch that is its own neighbour through
    all 26 boundaries (core/patch/sync_fields3d.c) against the oracle's periodic fill / fold"""
    from lambdapic_amd.fields import Fields3D
    from oracle import sync
    g = golden("g13_sync_3d")
    nx, ny, nz, ng = (int(g[k]) for k in ("nx", "ny", "nz", "ng"))
    f = Fields3D(nx, ny, nz, 1e-7, 1e-7, 1e-7, 0.0, 0.0, 0.0, ng)
    for a in f.attrs:
        getattr(f, a)[...] = g["in_" + a]
    sync.periodic_guard_fill(f, ["ex", "ey", "ez", "bx", "by", "bz"])
    sync.periodic_current_fold(f)
    for a in ("ex", "ey", "ez", "bx", "by", "bz"):
        assert np.array_equal(getattr(f, a), g["out_" + a]), a           # copies: bit exact
    for a in ("jx", "jy", "jz", "rho"):
        assert_close(getattr(f, a), g["out_" + a], 1e-14, what=a)        # sums of up to 8 terms


def test_g14_sync_particles_3d_single_periodic_patch(golden):
    """the reference's get_npart_to_extend_3d + fill_particles_from_boundary_3d on one self-periodic patch
    (core/patch/sync_particles_3d.c): the live particles afterwards are the live particles before with their
    coordinates folded into the box -- the rule the 3-D step applies (oracle.sync.periodic_fold_positions)"""
    from lambdapic_amd.particles import ParticlesBase
    g = golden("g14_sync_particles_3d")
    n3 = [int(g[k]) for k in ("nx", "ny", "nz")]
    d3 = [float(g[k]) for k in ("dx", "dy", "dz")]
    p = ParticlesBase(0, 0)
    p.initialize(g["pin_x"].size)
    for a in ("x", "y", "z", "ux", "w", "_id", "is_dead"):
        getattr(p, a)[:] = g["pin_" + a]
    sync.periodic_fold_positions(p, [-d / 2 for d in d3], [n * d - d / 2 for n, d in zip(n3, d3)])
    live, live_ref = ~p.is_dead, ~g["pout_is_dead"]
    assert live.sum() == live_ref.sum()
    o, r = np.argsort(p._id[live].view(np.uint64)), np.argsort(g["pout__id"][live_ref].view(np.uint64))
    assert np.array_equal(p._id[live].view(np.uint64)[o], g["pout__id"][live_ref].view(np.uint64)[r])
    for a in ("x", "y", "z", "ux", "w"):
        assert np.array_equal(getattr(p, a)[live][o], g["pout_" + a][live_ref][r]), a
    moved = sum(int(np.sum(getattr(p, a)[live] != g["pin_" + a][~g["pin_is_dead"]])) for a in ("x", "y", "z"))
    assert moved > 100                               # the case does exercise the fold



def test_g15_sort_variants(golden):
    """the numpy restatement of the bucket rule against the reference's compiled 3-D sort and its mirrored 2-D sort
    (out-of-range particles, dead-slot inheritance, clamping): counts, bounds and the number of misplaced slots"""
    g = golden("g15_sort_variants")
    for tag, axes, rev in (("a", "xyz", False), ("b", "xy", True)):
        pos = [g[f"{tag}_in_{a}"] for a in axes]
        idx, cnt = oracle.bucket_index_nd(pos, g[f"{tag}_in_is_dead"], tuple(g[f"{tag}_nb"]), tuple(g[f"{tag}_d"]),
                                          tuple(g[f"{tag}_o"]), rev)
        assert np.array_equal(cnt.reshape(g[f"{tag}_bucket_count"].shape), g[f"{tag}_bucket_count"])
        lo = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        assert np.array_equal(lo, g[f"{tag}_bucket_bound_min"].ravel())
        assert np.array_equal(lo + cnt, g[f"{tag}_bucket_bound_max"].ravel())
        ref = np.repeat(np.arange(cnt.size), cnt)
        assert int((idx != ref).sum()) == int(g[f"{tag}_nbuf"])
        # the reference's output: slot s holds input slot tag[s]; every slot holds a particle of its bucket
        src = g[f"{tag}_out_tag"].astype(np.int64)
        assert np.array_equal(np.sort(src), np.arange(src.size)) and np.array_equal(idx[src], ref)


# ---- 3-D patch lists: 2 x 2 x 2 periodic patches, every neighbour class another patch (g16 / g17) ----------------
def _g16_patches(g, prefix):
    from lambdapic_amd.patch import make_patches_3d
    npp, npatch, ng = tuple(int(v) for v in g["npp"]), tuple(int(v) for v in g["npatch"]), int(g["ng"])
    P = make_patches_3d(tuple(a * b for a, b in zip(npp, npatch)), (1e-7, 1.5e-7, 0.8e-7), npatch, ng)
    assert np.array_equal(np.stack([p.neighbor_ipatch for p in P]), g["neighbor_ipatch"])     # the table the reference got
    for k, p in enumerate(P):
        for a in ("ex", "by", "bz", "jx", "jy", "jz", "rho"):
            getattr(p.fields, a)[...] = g[f"{prefix}{k}_{a}"] / float(g["scale"])
    return P, npp, ng


def test_g16_sync_fields_3d_patch_list(golden):
    """the oracle's sync_guard_fields_3d / sync_currents_3d against the reference's compiled extension on 2 x 2 x 2
    periodic patches (core/patch/sync_fields3d.c:84-348,350-612): copies and folds exact"""
    g = golden("g16_sync_fields_3d_patches")
    P, npp, ng = _g16_patches(g, "in")
    fl = [p.fields for p in P]
    sync.sync_guard_fields_3d(fl, list(P), ["ex", "by", "bz"], 8, *npp, ng)
    sync.sync_currents_3d(fl, list(P), 8, *npp, ng)
    for k, p in enumerate(P):
        for a in ("ex", "by", "bz", "jx", "jy", "jz", "rho"):
            assert np.array_equal(getattr(p.fields, a) * float(g["scale"]), g[f"out{k}_{a}"].astype(float)), (k, a)


def _g17_patches(g):
    from lambdapic_amd.patch import make_patches_3d
    npp, npatch, d = tuple(int(v) for v in g["npp"]), tuple(int(v) for v in g["npatch"]), tuple(float(v) for v in g["d"])
    P = make_patches_3d(tuple(a * b for a, b in zip(npp, npatch)), d, npatch, 3)
    for k, p in enumerate(P):
        q = p.particles[0]
        q.initialize(g[f"pin{k}_x"].size)
        for a in ("x", "y", "z", "ux", "w", "_id", "is_dead"):
            getattr(q, a)[:] = g[f"pin{k}_{a}"]
    return P, d


def _check_g17_out(g, P):
    for k, p in enumerate(P):
        q = p.particles[0]
        assert q.npart == g[f"pout{k}_x"].size, k
        assert np.array_equal(q.is_dead, g[f"pout{k}_is_dead"]), k                # slot for slot
        live = ~q.is_dead
        for a in ("x", "y", "z", "ux", "w", "_id"):
            assert np.array_equal(getattr(q, a)[live].view(np.uint64), g[f"pout{k}_{a}"][live].view(np.uint64)), (k, a)
        for a in "xyz":                                                           # 3-D: dead slots are blanked
            assert np.isnan(getattr(q, a)[~live]).all() and np.isnan(g[f"pout{k}_{a}"][~live]).all(), (k, a)


def test_g17_sync_particles_3d_patch_list(golden):
    """the oracle's sync_particles_3d against get_npart_to_extend_3d + fill_particles_from_boundary_3d of the
    reference (core/patch/sync_particles_3d.c:365-700) on 2 x 2 x 2 periodic patches: counts, slot placement, +- L,
    is_dead pattern and blanked dead slots bit exact"""
    g = golden("g17_sync_particles_3d_patches")
    P, d = _g17_patches(g)
    alive = sync.sync_particles_3d(P, 0, d)
    assert np.array_equal(alive, g["npart_alive"])
    _check_g17_out(g, P)
    assert int(g["npart_outgoing"].reshape(8, 26).sum(0).min()) > 0              # all 26 classes move particles
