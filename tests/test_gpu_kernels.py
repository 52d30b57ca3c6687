"""GPU parity tests of the HIP kernels, through the C ABI (via the kernel-level drop-ins), against
(a) the committed golden vectors recorded from the reference's own kernels and (b) the oracle on
the same seeded inputs.

Tolerances (FP64, stated per quantity):
  * pushed particle attributes: 1e-12 of the array's max -- few-ulp differences from FMA
    contraction (hipcc contracts, the reference build contracts differently);
  * gathered E/B at the particle: 1e-11 of the max (9/27-term sums of O(1e12) random fields);
  * grid rho/J: 1e-12 of the max entry -- atomics change the summation order only;
  * FDTD: 1e-14.
"""
import numpy as np
import pytest

import oracle
from helpers import PEB, assert_close, fields2d_from, fields3d_from, particles_from
from lambdapic_amd import kernels
from lambdapic_amd.fields import Fields2D
from lambdapic_amd.particles import ParticlesBase

pytestmark = pytest.mark.gpu

C = 299792458.0
QE, ME = -1.602176634e-19, 9.1093837139e-31


STRIPED, CELL_MAJOR = 1, 0   # LPA_ORDER_*
MODES = [(False, STRIPED), (True, STRIPED), (True, CELL_MAJOR)]   # global atomics / tiled x 2 orders


def _cmp_fused_2d(g, k, tiled, order):
    f = fields2d_from(g, f"in{k}_", g[f"x0_{k}"], g[f"y0_{k}"])
    p = particles_from(g, f"in{k}_")
    kernels.unified_boris_pusher_cpu_2d([p], [f], 1, float(g["dt"]), float(g["q"]), float(g["m"]),
                                        tiled=tiled, order=order)
    for a in ["x", "y", "ux", "uy", "uz", "inv_gamma"]:
        assert_close(getattr(p, a), g[f"out{k}_{a}"], 1e-12, what=f"patch{k} {a}")
    alive = ~(p.is_dead | np.isnan(g[f"in{k}_x"]) | np.isnan(g[f"in{k}_y"]))
    for a in PEB:
        assert_close(getattr(p, a)[alive], g[f"out{k}_{a}"][alive], 1e-11, what=f"patch{k} {a}")
    for a in ["rho", "jx", "jy", "jz"]:
        assert_close(getattr(f, a), g[f"out{k}_{a}"], 1e-12, what=f"patch{k} {a}")


@pytest.mark.parametrize("tiled,order", MODES)
def test_fused_2d_vs_golden(golden, tiled, order):
    g = golden("g1_fused_2d")
    for k in range(int(g["npatches"])):
        _cmp_fused_2d(g, k, tiled, order)


def test_fused_3d_vs_golden(golden):
    g = golden("g2_fused_3d")
    f = fields3d_from(g, "in_")
    p = particles_from(g, "in_")
    kernels.unified_boris_pusher_cpu_3d([p], [f], 1, float(g["dt"]), float(g["q"]), float(g["m"]))
    for a in ["x", "y", "z", "ux", "uy", "uz", "inv_gamma"]:
        assert_close(getattr(p, a), g[f"out_{a}"], 1e-12, what=a)
    alive = ~(p.is_dead | np.isnan(g["in_x"]) | np.isnan(g["in_y"]) | np.isnan(g["in_z"]))
    for a in PEB:
        assert_close(getattr(p, a)[alive], g[f"out_{a}"][alive], 1e-11, what=a)
    for a in ["rho", "jx", "jy", "jz"]:
        assert_close(getattr(f, a), g[f"out_{a}"], 1e-12, what=a)


def test_standalone_deposit_and_interpolation_vs_golden(golden):
    g = golden("g3_deposit_2d")
    f = fields2d_from(g, "none_", g["x0"], g["y0"])
    p = particles_from(g, "in_")
    kernels.current_deposition_cpu_2d([f], [p], 1, float(g["dt"]), float(g["q"]))
    for a in ["rho", "jx", "jy", "jz"]:
        assert_close(getattr(f, a), g[f"out_{a}"], 1e-12, what=a)
    g = golden("g4_interp_2d")
    f = fields2d_from(g, "in_", g["x0"], g["y0"])
    p = particles_from(g, "in_", ["x", "y"])
    kernels.interpolation_patches_2d([p], [f], 1)
    for a in PEB:
        assert_close(getattr(p, a)[~p.is_dead], g[f"out_{a}"][~p.is_dead], 1e-11, what=a)
    # the 3-D standalone twins (core/current/cpu3d.c:118-183, core/interpolation/cpu3d.c:99-169)
    g = golden("g3_deposit_3d")
    f = fields3d_from(g, "none_")
    p = particles_from(g, "in_")
    kernels.current_deposition_cpu_3d([f], [p], 1, float(g["dt"]), float(g["q"]))
    for a in ["rho", "jx", "jy", "jz"]:
        assert_close(getattr(f, a), g[f"out_{a}"], 1e-12, what=a)
    g = golden("g4_interp_3d")
    f = fields3d_from(g, "in_")
    p = particles_from(g, "in_", ["x", "y", "z"])
    kernels.interpolation_patches_3d([p], [f], 1)
    for a in PEB:
        assert_close(getattr(p, a)[~p.is_dead], g[f"out_{a}"][~p.is_dead], 1e-11, what=a)


def test_fdtd_vs_golden(golden):
    g = golden("g5_fdtd_2d")
    f = fields2d_from(g, "in_", 0.0, 0.0)
    kernels.update_efield_2d(f, float(g["dt"]))
    for a in ["ex", "ey", "ez"]:
        assert_close(getattr(f, a), g[f"outE_{a}"], 1e-14, what=a)
    kernels.update_bfield_2d(f, float(g["dt"]))
    for a in ["bx", "by", "bz"]:
        assert_close(getattr(f, a), g[f"outB_{a}"], 1e-14, what=a)
    g = golden("g5_fdtd_3d")
    f = fields3d_from({**{k: g[k] for k in g.files}, "x0": 0.0, "y0": 0.0, "z0": 0.0}, "in_")
    kernels.update_efield_3d(f, float(g["dt"]))
    for a in ["ex", "ey", "ez"]:
        assert_close(getattr(f, a), g[f"outE_{a}"], 1e-14, what=a)
    kernels.update_bfield_3d(f, float(g["dt"]))
    for a in ["bx", "by", "bz"]:
        assert_close(getattr(f, a), g[f"outB_{a}"], 1e-14, what=a)


def _random_case(nx, ny, n, seed, u_scale):
    rng = np.random.default_rng(seed)
    dx, dy = 4e-8, 5e-8
    f = Fields2D(nx, ny, dx, dy, 3 * dx, -2 * dy, 3)
    for a in ("ex", "ey", "ez"):
        getattr(f, a)[...] = rng.normal(size=f.shape) * 1e12
    for a in ("bx", "by", "bz"):
        getattr(f, a)[...] = rng.normal(size=f.shape) * 1e4
    p = ParticlesBase(0, 0)
    p.initialize(n)
    p.x[:] = f.x0 + rng.uniform(-0.5, nx - 0.5, n) * dx
    p.y[:] = f.y0 + rng.uniform(-0.5, ny - 0.5, n) * dy
    for a in ("ux", "uy", "uz"):
        getattr(p, a)[:] = rng.normal(size=n) * u_scale
    p.inv_gamma[:] = 1 / np.sqrt(1 + p.ux ** 2 + p.uy ** 2 + p.uz ** 2)
    p.w[:] = rng.uniform(0.5, 1.5, n) * 1e27 * dx * dy / 10
    p.is_dead[rng.random(n) < 0.03] = True
    dt = 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2))
    return f, p, dt


def _copy_case(f, p):
    import copy
    return copy.deepcopy(f), copy.deepcopy(p)


@pytest.mark.parametrize("tiled,order,u_scale", [(False, STRIPED, 1.0), (True, STRIPED, 1.0), (True, STRIPED, 0.05),
                                                 (True, CELL_MAJOR, 1.0), (True, CELL_MAJOR, 0.05)])
def test_fused_2d_vs_oracle_multi_tile(tiled, order, u_scale):
    """80x48 cells (10x2 tiles, ragged in y), 200k particles incl. relativistic ones: every tile edge,
    the torus wrap at the patch edge, dead slots."""
    f, p, dt = _random_case(80, 48, 200_000, 11, u_scale)
    fo, po = _copy_case(f, p)
    oracle.unified_boris_pusher_cpu_2d([po], [fo], 1, dt, QE, ME)
    kernels.unified_boris_pusher_cpu_2d([p], [f], 1, dt, QE, ME, tiled=tiled, order=order)
    for a in ["x", "y", "ux", "uy", "uz", "inv_gamma"]:
        assert_close(getattr(p, a), getattr(po, a), 1e-12, what=a)
    live = ~p.is_dead
    for a in PEB:
        assert_close(getattr(p, a)[live], getattr(po, a)[live], 1e-11, what=a)
    for a in ["rho", "jx", "jy", "jz"]:
        assert_close(getattr(f, a), getattr(fo, a), 1e-12, what=a)


def test_dead_particle_invariance():
    """reference tests/core/pusher/test_unified_pusher_2d.py:159-208: a dead particle changes
    nothing (2-particle run == 1-particle run at rtol 1e-10)"""
    dx = dy = 1e-8
    def run(n, dead):
        f = Fields2D(16, 16, dx, dy, 0.0, 0.0, 3)
        f.ez[...] = 1e10
        f.bz[...] = 3.0
        p = ParticlesBase(0, 0)
        p.initialize(n)
        p.x[:] = 5.3 * dx
        p.y[:] = 7.1 * dy
        p.ux[:] = 0.1
        p.inv_gamma[:] = 1 / np.sqrt(1.01)
        p.w[:] = 1e27 * dx * dy / 10
        if dead:
            p.is_dead[1] = True
        kernels.unified_boris_pusher_cpu_2d([p], [f], 1, 1e-17, QE, ME)
        return f, p
    f1, p1 = run(1, False)
    f2, p2 = run(2, True)
    for a in ["rho", "jx", "jy", "jz"]:
        np.testing.assert_allclose(getattr(f2, a), getattr(f1, a), rtol=1e-10, atol=0)
    assert p2.x[1] == 5.3 * dx and p2.ux[1] == 0.1          # untouched
    np.testing.assert_allclose(p2.x[0], p1.x[0], rtol=1e-15)


def test_known_answer_charge_and_current():
    """reference tests/core/current/test_current_deposition.py:328-369: sum rho = q n,
    sum J = q n v for a single particle (1e-10 rel)"""
    dx = dy = 1e-8
    f = Fields2D(16, 16, dx, dy, 0.0, 0.0, 3)
    p = ParticlesBase(0, 0)
    p.initialize(1)
    p.x[:], p.y[:] = 6.3 * dx, 9.8 * dy
    p.ux[:], p.uy[:], p.uz[:] = 0.3, -0.2, 0.1
    p.inv_gamma[:] = 1 / np.sqrt(1 + 0.09 + 0.04 + 0.01)
    p.w[:] = 1e27 * dx * dy / 10
    dt = 1e-17
    kernels.current_deposition_cpu_2d([f], [p], 1, dt, QE)
    n = p.w[0] / (dx * dy)
    v = np.array([0.3, -0.2, 0.1]) * p.inv_gamma[0] * C
    assert f.rho.sum() == pytest.approx(QE * n, rel=1e-10)
    assert f.jx.sum() == pytest.approx(QE * n * v[0], rel=1e-10)
    assert f.jy.sum() == pytest.approx(QE * n * v[1], rel=1e-10)
    assert f.jz.sum() == pytest.approx(QE * n * v[2], rel=1e-10)


def test_boundary_wrap_deposits_through_the_torus():
    """reference tests/core/pusher/test_unified_pusher_2d.py:218-252: particle at the patch corner
    moving out with u = (-1,-1) deposits through the wrap; compare with the oracle cell by cell"""
    dx = dy = 1e-8
    def mk():
        f = Fields2D(8, 8, dx, dy, 0.0, 0.0, 3)
        f.ez[...] = 1e10
        f.bx[...], f.by[...], f.bz[...] = 1.0, 2.0, 3.0
        p = ParticlesBase(0, 0)
        p.initialize(1)
        p.x[:], p.y[:] = 0.1 * dx, 0.1 * dy
        p.ux[:], p.uy[:] = -1.0, -1.0
        p.inv_gamma[:] = 1 / np.sqrt(3.0)
        p.w[:] = 1e27 * dx * dy / 10
        return f, p
    f, p = mk()
    fo, po = mk()
    kernels.unified_boris_pusher_cpu_2d([p], [f], 1, 1e-17, QE, ME)
    oracle.unified_boris_pusher_cpu_2d([po], [fo], 1, 1e-17, QE, ME)
    assert f.rho.sum() != 0.0
    for a in ["rho", "jx", "jy", "jz"]:
        assert_close(getattr(f, a), getattr(fo, a), 1e-12, what=a)
    assert_close(p.x, po.x, 1e-13)


@pytest.mark.k1_variants
def test_empty_and_all_dead_inputs():
    dx = dy = 1e-8
    f = Fields2D(8, 8, dx, dy, 0.0, 0.0, 3)
    p = ParticlesBase(0, 0)
    p.initialize(0)
    kernels.unified_boris_pusher_cpu_2d([p], [f], 1, 1e-17, QE, ME)           # no particles
    kernels.unified_boris_pusher_cpu_2d([p], [f], 0, 1e-17, QE, ME)           # npatches <= 0
    p.initialize(5)
    p.is_dead[:] = True
    for tiled, order in MODES:
        kernels.unified_boris_pusher_cpu_2d([p], [f], 1, 1e-17, QE, ME, tiled=tiled, order=order)
    assert not f.rho.any() and not f.jx.any()


def test_split_kernels_vs_oracle():
    """the non-fused path: push_position, interpolate, boris, push_position, deposit"""
    f, p, dt = _random_case(32, 32, 20_000, 5, 0.5)
    fo, po = _copy_case(f, p)
    kernels.push_position_2d(p, 0.5 * dt)
    oracle.push_position_2d(po, 0.5 * dt)
    assert_close(p.x, po.x, 1e-15)
    kernels.interpolation_patches_2d([p], [f], 1)
    oracle.interpolation_patches_2d([po], [fo], 1)
    live = ~p.is_dead
    for a in PEB:
        assert_close(getattr(p, a)[live], getattr(po, a)[live], 1e-11, what=a)
    kernels.boris_push(p, QE, ME, dt)
    oracle.boris_push(po, QE, ME, dt)
    for a in ["ux", "uy", "uz", "inv_gamma"]:
        assert_close(getattr(p, a), getattr(po, a), 1e-12, what=a)
    kernels.current_deposition_cpu_2d([f], [p], 1, dt, QE)
    oracle.current_deposition_cpu_2d([fo], [po], 1, dt, QE)
    for a in ["rho", "jx", "jy", "jz"]:
        assert_close(getattr(f, a), getattr(fo, a), 1e-12, what=a)
    kernels.reset_current_cpu_2d([f], 1)
    assert not f.jx.any() and not f.rho.any()


@pytest.mark.k1_variants
def test_wave_reduce_scatter_selftest():
    """the DPP / permlane reduce-scatter used by the tiled deposit: lane L must end with the sum
    over lanes of value L (all 64 x 64 inputs distinct)"""
    import ctypes as C
    import torch
    from lambdapic_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(2)
    a = rng.normal(size=(64, 64))
    d_in = torch.from_numpy(a).cuda()
    d_out = torch.zeros(64, dtype=torch.float64, device="cuda")
    _lib.check(L.lpa_selftest_wave_reduce(d_in.data_ptr(), d_out.data_ptr(), None), "selftest")
    torch.cuda.synchronize()
    np.testing.assert_allclose(d_out.cpu().numpy(), a.sum(axis=1), rtol=1e-13, atol=1e-13)


def _expected_tile_sequence(cnt, order, rmax=128):
    """the cell sequence of one tile's particles for a striped order, from its 256 cell counts (``rmax``: the ranks the
    workspace keeps in stripes, lpa_sort_stripe_ranks)"""
    if order == 3:     # LPA_ORDER_COLUMN: rank by rank inside each 32-cell row of the 2-D tile, row after row
        exp = []
        for c0 in range(0, 256, 32):
            col = cnt[c0:c0 + 32]
            exp += [c0 + np.nonzero(col > r)[0] for r in range(col.max())]
        return np.concatenate(exp) if exp else np.zeros(0, int)
    # for r < rmax the cells with count > r ascending, then the deep tails
    exp = [np.nonzero(cnt > r)[0] for r in range(min(cnt.max(), rmax))]
    exp += [np.repeat(c, cnt[c] - rmax) for c in np.nonzero(cnt > rmax)[0]]
    return np.concatenate(exp)


def keys_of(x, y, nx, ny, dx, dy):
    """tile-major cell key of the 2-D tile sort (tiles of 8 x 32 cells)"""
    tiles_y = (ny + 31) // 32
    i = np.clip(np.floor(x / dx + 0.5).astype(int), 0, nx - 1)
    j = np.clip(np.floor(y / dy + 0.5).astype(int), 0, ny - 1)
    return ((i // 8) * tiles_y + j // 32) * 256 + (i % 8) * 32 + (j % 32)


@pytest.mark.parametrize("order,adapt", [(STRIPED, False), (CELL_MAJOR, False), (2, False), (3, False), (STRIPED, True)])
def test_cell_sort_properties(order, adapt):
    """reference tests/test_sort.py:38-117,201-251 restated for the device sort: per-cell counts
    equal a numpy histogram, tiles are contiguous and in order, inside a tile the particles are cell
    by cell (CELL_MAJOR), rank by rank with cells ascending inside a rank (STRIPED) or the same inside each
    column of the tile (COLUMN), the multiset of live particles is preserved, dead / NaN particles are dropped,
    re-sorting keeps the keys.  ``adapt``: the engine re-sizes the striped ranks for the deep cells (its default; the
    other cases switch that off and pin the un-striped tail of the default rule)."""
    import torch
    from lambdapic_amd.engine import PicEngine2D
    rng = np.random.default_rng(9)
    nx, ny, dx, dy = 20, 72, 4e-8, 5e-8      # 3 x 3 tiles of 8 x 32 cells, ragged edges
    n = 60_000
    p = ParticlesBase(0, 0)
    p.initialize(n)
    p.x[:] = rng.uniform(-0.5, nx - 0.5, n) * dx
    p.y[:] = rng.uniform(-0.5, ny - 0.5, n) * dy
    # a few very deep cells (> 128 particles) exercise the un-striped tail
    p.x[:2000] = 3.2 * dx
    p.y[:2000] = rng.choice([5.1, 6.0, 40.2], 2000) * dy
    p.ux[:] = rng.normal(size=n)
    p.w[:] = rng.uniform(1, 2, n)
    p.is_dead[rng.random(n) < 0.1] = True
    p.x[7] = np.nan
    live = ~p.is_dead & ~np.isnan(p.x)
    eng = PicEngine2D(nx, ny, dx, dy, device="cuda:0", block_particles=1024, order=order)
    eng.add_species(QE, ME, capacity=2 * n if order == 2 else n + 64)     # LPA_ORDER_PADDED stores holes
    eng.species[0].upload([p])
    assert eng.species[0].n == live.sum()
    if not adapt:
        eng.deep_tail_fraction = 2.0
    eng.sort(0)
    rmax = int(eng.L.lpa_sort_stripe_ranks(eng._g(), eng.species[0].capacity))
    assert 32 <= rmax < 2000 // 3           # the deep cells below (~600 live particles each) do leave the default stripes
    used = eng._ws[id(eng.species[0])]["tiling"].stripe_ranks
    if adapt:                               # 3 % of the store lay beyond them: re-sized for the deepest cell + 25 %
        deep = np.bincount(keys_of(p.x[live], p.y[live], nx, ny, dx, dy)).max()
        assert used == 1024 and deep * 1.25 > 512 and eng.species[0].stripe_ranks == deep + deep // 4
        rmax = used
    else:
        assert used == rmax
    out = eng.species[0].download()
    assert out["x"].size == live.sum()

    def keys(x, y):
        return keys_of(x, y, nx, ny, dx, dy)

    k_out = keys(out["x"], out["y"])
    assert np.array_equal(np.bincount(k_out), np.bincount(keys(p.x[live], p.y[live])))
    tile = k_out >> 8
    assert np.all(np.diff(tile) >= 0)
    if order == CELL_MAJOR:
        assert np.all(np.diff(k_out) >= 0)
    else:
        for t in np.unique(tile):
            kt = k_out[tile == t] & 255
            cnt = np.bincount(kt, minlength=256)
            assert np.array_equal(kt, _expected_tile_sequence(cnt, order, rmax)), t
    order_in = np.argsort(p.id[live])
    order_out = np.argsort(out["_id"].view(np.uint64))
    assert np.array_equal(p.id[live][order_in], out["_id"].view(np.uint64)[order_out])
    for a in ("x", "y", "ux", "w"):
        assert np.array_equal(getattr(p, a)[live][order_in], out[a][order_out])
    eng.sort(0)
    out2 = eng.species[0].download()
    assert np.array_equal(keys(out2["x"], out2["y"]), k_out)

    # a RE-sort after the particles moved (tile-local count + tile-staged scatter): drift of up to 1.5
    # cells (cell and tile changes), some particles killed since the last sort
    sp = eng.species[0]
    m = sp.n
    gx = torch.Generator(device="cuda").manual_seed(3)
    for a, d in (("x", dx), ("y", dy)):
        sp.cset.arr(a)[:m] += (torch.rand(m, device="cuda", dtype=torch.float64, generator=gx) - 0.5) * 3.0 * d
    sp.cset.arr("x")[:m][torch.rand(m, device="cuda", generator=gx) < 0.05] = float("nan")
    before = sp.download()
    eng.sort(0)
    out3 = sp.download()
    assert out3["x"].size == before["x"].size
    k3 = keys(out3["x"], out3["y"])
    assert np.array_equal(np.bincount(k3, minlength=k_out.max() + 1),
                          np.bincount(keys(before["x"], before["y"]), minlength=k_out.max() + 1))
    t3 = k3 >> 8
    assert np.all(np.diff(t3) >= 0)
    if order == CELL_MAJOR:
        assert np.all(np.diff(k3) >= 0)
    else:
        for t in np.unique(t3):
            kt = k3[t3 == t] & 255
            cnt = np.bincount(kt, minlength=256)
            assert np.array_equal(kt, _expected_tile_sequence(cnt, order, rmax)), t
    ib, ia = np.argsort(before["_id"].view(np.uint64)), np.argsort(out3["_id"].view(np.uint64))
    assert np.array_equal(before["_id"].view(np.uint64)[ib], out3["_id"].view(np.uint64)[ia])
    for a in ("x", "y", "ux", "w"):
        assert np.array_equal(before[a][ib], out3[a][ia])


@pytest.mark.k1_variants
def test_staggered_constant_fields_known_answer():
    """reference tests/core/interpolation/test_field_interpolation_2d.py:322-359: constant fields are
    gathered exactly wherever the particle sits, despite the Yee staggering (TSC weights sum to one on the
    node-centred and on the half-cell-centred stencil) -- standalone gather and the gather inside the fused
    kernels (global-memory and LDS-tiled), 2-D and 3-D"""
    from lambdapic_amd.fields import Fields3D
    nx, ny, nz, dx, dy, dz = 16, 12, 10, 1.0e-6, 1.5e-6, 0.7e-6
    vals = dict(ex=1.0, ey=2.0, ez=3.0, bx=4.0, by=5.0, bz=6.0)
    pos = [(0.5, 0.5, 0.5), (1.0, 1.0, 1.0), (0.25, 0.75, 0.4), (nx - 0.5, ny - 0.5, nz - 0.5), (0.0, 0.0, 0.0),
           (-0.49, ny - 0.51, 3.3)]

    def bag(dim):
        p = ParticlesBase(0, 0)
        p.initialize(len(pos))
        p.x[:] = [a * dx for a, _, _ in pos]
        p.y[:] = [b * dy for _, b, _ in pos]
        if dim == 3:
            p.z[:] = [min(c, nz - 0.5) * dz for _, _, c in pos]
        p.w[:] = 1.0
        p.inv_gamma[:] = 1.0
        return p

    f2 = Fields2D(nx, ny, dx, dy, 0.0, 0.0, 3)
    f3 = Fields3D(nx, ny, nz, dx, dy, dz, 0.0, 0.0, 0.0, 3)
    for f in (f2, f3):
        for a, v in vals.items():
            getattr(f, a)[...] = v
    p = bag(2)
    kernels.interpolation_patches_2d([p], [f2], 1)
    for a, v in vals.items():
        assert np.allclose(getattr(p, a + "_part"), v, rtol=1e-14, atol=0), a
    p = bag(3)
    kernels.interpolation_patches_3d([p], [f3], 1)
    for a, v in vals.items():
        assert np.allclose(getattr(p, a + "_part"), v, rtol=1e-14, atol=0), a
    # the gather inside the fused kernels writes the same *_part (dt tiny: the mid-step position is the input)
    for tiled, order in MODES:
        p = bag(2)
        kernels.unified_boris_pusher_cpu_2d([p], [f2], 1, 1e-22, QE, ME, tiled=tiled, order=order)
        for a, v in vals.items():
            assert np.allclose(getattr(p, a + "_part"), v, rtol=1e-14, atol=0), (a, tiled, order)
    p = bag(3)
    kernels.unified_boris_pusher_cpu_3d([p], [f3], 1, 1e-22, QE, ME)
    for a, v in vals.items():
        assert np.allclose(getattr(p, a + "_part"), v, rtol=1e-14, atol=0), a


def test_known_answer_charge_and_current_3d():
    """reference tests/core/current/test_current_deposition.py:517-557 (test_precision_3d): for one particle
    sum rho = q n and sum J = q n v to 1e-10 -- standalone 3-D deposit and the deposit inside the fused kernel"""
    from lambdapic_amd.fields import Fields3D
    dx, dy, dz = 1e-8, 1.2e-8, 0.9e-8
    n = None
    for fused in (False, True):
        f = Fields3D(12, 10, 14, dx, dy, dz, 0.0, 0.0, 0.0, 3)
        p = ParticlesBase(0, 0)
        p.initialize(1)
        p.x[:], p.y[:], p.z[:] = 6.3 * dx, 4.8 * dy, 7.45 * dz
        p.ux[:], p.uy[:], p.uz[:] = 0.3, -0.2, 0.15
        p.inv_gamma[:] = 1 / np.sqrt(1 + 0.09 + 0.04 + 0.0225)
        p.w[:] = 1e27 * dx * dy * dz / 10
        dt = 1e-17
        if fused:      # E = B = 0: the push keeps u, the deposit sees the same velocity
            kernels.unified_boris_pusher_cpu_3d([p], [f], 1, dt, QE, ME)
        else:
            kernels.current_deposition_cpu_3d([f], [p], 1, dt, QE)
        n = p.w[0] / (dx * dy * dz)
        v = np.array([0.3, -0.2, 0.15]) * p.inv_gamma[0] * C
        assert f.rho.sum() == pytest.approx(QE * n, rel=1e-10)
        assert f.jx.sum() == pytest.approx(QE * n * v[0], rel=1e-10)
        assert f.jy.sum() == pytest.approx(QE * n * v[1], rel=1e-10)
        assert f.jz.sum() == pytest.approx(QE * n * v[2], rel=1e-10)


@pytest.mark.k1_variants
def test_padded_order_layout():
    """LPA_ORDER_PADDED: in every tile the leading ranks that at least 192 of the 256 cells have are FULL stripes --
    slot = tile start + rank * 256 + cell, holes (NaN) where a cell has no such particle --, tile starts are
    multiples of 64, the compact rest follows, and a re-sort of the padded store (holes in the source) keeps all of
    that"""
    import torch
    from lambdapic_amd.engine import PicEngine2D
    rng = np.random.default_rng(19)
    nx, ny, dx, dy = 24, 64, 4e-8, 5e-8          # 3 x 2 tiles
    ppc = 40
    n = nx * ny * ppc
    p = ParticlesBase(0, 0)
    p.initialize(n)
    cell = np.arange(n) // ppc
    p.x[:] = ((cell // ny) + rng.uniform(-0.5, 0.5, n)) * dx
    p.y[:] = ((cell % ny) + rng.uniform(-0.5, 0.5, n)) * dy
    p.is_dead[rng.random(n) < 0.3] = True          # Poisson-like counts: mean 28, partial ranks
    p.w[:] = 1.0
    p.inv_gamma[:] = 1.0
    live = ~p.is_dead
    eng = PicEngine2D(nx, ny, dx, dy, device="cuda:0", block_particles=1024, order=2)
    eng.add_species(QE, ME, capacity=2 * n)
    sp = eng.species[0]
    sp.upload([p])

    def check():
        x = sp.cset.arr("x")[: sp.n_sorted].cpu().numpy()
        y = sp.cset.arr("y")[: sp.n_sorted].cpu().numpy()
        ok = ~np.isnan(x)
        assert ok.sum() == live.sum() and np.array_equal(np.isnan(x), np.isnan(y))
        i = np.clip(np.floor(x[ok] / dx + 0.5).astype(int), 0, nx - 1)
        j = np.clip(np.floor(y[ok] / dy + 0.5).astype(int), 0, ny - 1)
        tile = (i // 8) * 2 + j // 32
        cellk = (i % 8) * 32 + (j % 32)
        slot = np.nonzero(ok)[0]
        assert np.all(np.diff(tile) >= 0)
        start = 0
        for t in range(6):
            st = slot[tile == t]
            ct = cellk[tile == t]
            cnt = np.bincount(ct, minlength=256)
            ra = int(sum(1 for r in range(cnt.max()) if (cnt > r).sum() >= 192))
            assert ra >= 10                                            # most of the tile is in full stripes
            assert start % 64 == 0 and st.min() >= start
            rel = st - start
            full = rel < ra * 256
            assert np.array_equal(rel[full] % 256, ct[full])           # the slot tells the cell
            assert np.array_equal(np.bincount(rel[full] // 256, minlength=ra),
                                  np.array([(cnt > r).sum() for r in range(ra)]))   # holes exactly where a cell is short
            total = ra * 256 + int(np.maximum(cnt - ra, 0).sum())
            assert rel.max() < total
            start += (total + 63) // 64 * 64
        assert sp.n_sorted == start

    eng.sort(0)
    check()
    eng.sort(0)       # re-sort: tile-staged path, source with holes
    check()


@pytest.mark.k1_variants
def test_wave_shift_selftest():
    """DPP wave_shr:1 / wave_shl:1 as the cooperative deposit uses them: lane l reads lane l - 1 / l + 1 across the
    four 16-lane rows of the wave, the ends read zero"""
    import torch
    from lambdapic_amd import _lib
    L = _lib.lib()
    a = np.arange(64, dtype=np.float64) * 1.5 + 0.25
    d_in = torch.from_numpy(a).cuda()
    d_out = torch.zeros(128, dtype=torch.float64, device="cuda")
    _lib.check(L.lpa_selftest_wave_shift(d_in.data_ptr(), d_out.data_ptr(), None), "selftest")
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    assert np.array_equal(out[:64], np.concatenate([[0.0], a[:-1]]))
    assert np.array_equal(out[64:], np.concatenate([a[1:], [0.0]]))


@pytest.mark.k1_variants
def test_cooperative_deposit_vs_oracle():
    """the cooperative deposit (LPA_ORDER_PADDED store, neighbour exchange, row-end lanes, parked misfits) against the
    oracle's fused kernel on a multi-tile patch: hot enough for cell changes, ragged cell counts, dead slots"""
    import copy
    rng = np.random.default_rng(23)
    nx, ny, dx, dy = 24, 70, 4e-8, 5e-8
    dt = 0.95 / (299792458.0 * np.sqrt(dx ** -2 + dy ** -2))
    f = Fields2D(nx, ny, dx, dy, 0.0, 0.0, 3)
    for a in ("ex", "ey", "ez"):
        getattr(f, a)[...] = rng.normal(size=f.shape) * 1e12
    for a in ("bx", "by", "bz"):
        getattr(f, a)[...] = rng.normal(size=f.shape) * 1e4
    n = nx * ny * 30
    p = ParticlesBase(0, 0)
    p.initialize(n)
    p.x[:] = rng.uniform(-0.5, nx - 0.5, n) * dx
    p.y[:] = rng.uniform(-0.5, ny - 0.5, n) * dy
    for a in ("ux", "uy", "uz"):
        getattr(p, a)[:] = rng.normal(size=n) * 0.3
    p.inv_gamma[:] = 1 / np.sqrt(1 + p.ux ** 2 + p.uy ** 2 + p.uz ** 2)
    p.w[:] = rng.uniform(0.5, 1.5, n) * 1e27 * dx * dy / 30
    p.is_dead[rng.random(n) < 0.05] = True
    fo, po = copy.deepcopy(f), copy.deepcopy(p)
    kernels.unified_boris_pusher_cpu_2d([p], [f], 1, dt, QE, ME, tiled=True, order=2)
    oracle.unified_boris_pusher_cpu_2d([po], [fo], 1, dt, QE, ME)
    live = ~p.is_dead
    for a in ("x", "y", "ux", "uy", "uz", "inv_gamma"):
        assert_close(getattr(p, a)[live], getattr(po, a)[live], 1e-12, what=a)
    for a in ("rho", "jx", "jy", "jz"):
        assert_close(getattr(f, a), getattr(fo, a), 1e-12, what=a)


@pytest.mark.k1_variants
def test_padded_sort_refuses_a_store_that_is_too_small():
    """LPA_ORDER_PADDED stores full stripes with holes and rounds every tile to 64 slots: up to 4/3 n + 63 per tile.  A
    destination sized for the live count cannot hold that; the device compares the slot total with dst->n after its
    scan, moves NOTHING and flags the sort (lpa_sort_overflow) instead of scattering past the arrays"""
    from lambdapic_amd import _lib
    from lambdapic_amd.engine import PicEngine2D
    nx, ny, ppc = 32, 64, 12
    dx = dy = 4e-8
    rng = np.random.default_rng(5)
    n = nx * ny * ppc
    p = ParticlesBase(0, 0)
    p.initialize(n)
    p.x[:] = rng.uniform(-0.5, nx - 0.5, n) * dx          # Poisson occupancies: plenty of holes in the full stripes
    p.y[:] = rng.uniform(-0.5, ny - 0.5, n) * dy
    p.inv_gamma[:] = 1.0
    p.w[:] = 1.0
    eng = PicEngine2D(nx, ny, dx, dy, device="cuda:0", block_particles=1024, order=2)
    eng.add_species(QE, ME, capacity=n + 64)            # fits the live particles, not their padded order
    eng.species[0].upload([p])
    before = eng.species[0].download()
    with pytest.raises(_lib.LpaError, match="more slots"):
        eng.sort(0)
    after = eng.species[0].download()                   # the source store is untouched and still current
    for k in before:
        assert np.array_equal(before[k].view(np.uint64), after[k].view(np.uint64)), k
    roomy = PicEngine2D(nx, ny, dx, dy, device="cuda:0", block_particles=1024, order=2)
    roomy.add_species(QE, ME, capacity=2 * n)
    roomy.species[0].upload([p])
    roomy.sort(0)
    assert roomy.species[0].n_sorted >= n and roomy.diagnostics()["nalive"][0] == n


def test_sort_refuses_a_prefix_hint_its_workspace_does_not_confirm():
    """lpa_tiling.prefix_hint lets a re-sort skip the per-particle launches over the tile-ordered prefix; a hint the
    workspace header does not vouch for (here: the header is wiped, as if the store had been replaced behind the
    engine's back) would leave those particles unsorted -- the device refuses the whole sort instead"""
    import torch
    from lambdapic_amd import _lib
    from lambdapic_amd.engine import PicEngine2D
    rng = np.random.default_rng(2)
    nx, ny, dx, dy, n = 16, 64, 4e-8, 4e-8, 20_000
    p = ParticlesBase(0, 0)
    p.initialize(n)
    p.x[:] = rng.uniform(-0.5, nx - 0.5, n) * dx
    p.y[:] = rng.uniform(-0.5, ny - 0.5, n) * dy
    p.w[:] = 1.0
    eng = PicEngine2D(nx, ny, dx, dy, device="cuda:0", block_particles=1024)
    eng.add_species(QE, ME, capacity=n + 64)
    eng.species[0].upload([p])
    eng.sort(0)
    eng.sort(0)                                    # a confirmed hint: fine
    assert eng.species[0].n_sorted == n
    ws = eng._sort_ws(eng.species[0])
    ws["sort"][:64].zero_()                        # the header forgets the previous sort ...
    with pytest.raises(_lib.LpaError, match="prefix_hint"):
        eng.sort(0)                                # ... the engine still hints n_sorted
    eng.species[0].tiling = None                   # what an upload does: no hint, a first sort
    eng.sort(0)
    assert eng.species[0].n_sorted == n
