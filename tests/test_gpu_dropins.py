"""Kernel-level drop-ins with the reference's own argument lists against the vectors recorded
from the reference's compiled extensions: g6 (`sort_particles_patches_2d`, core/sort/cpu2d.c:220-303) and g7
(`sync_guard_fields_2d` / `sync_currents_2d` on 2 x 2 periodic patches, core/patch/sync_fields2d.c:47,154);
plus the resident engine's TILE sort pinned to g6's bucket histogram and particle set."""
import numpy as np
import pytest

from helpers import assert_close, particles_from
from lambdapic_amd import kernels
from lambdapic_amd.patch import make_patches_2d

pytestmark = pytest.mark.gpu
ATTRS = ["x", "y", "ux", "uy", "uz", "w", "_id"]


def _g6_lists(g):
    p = particles_from(g, "in_", ["x", "y", "ux", "uy", "uz", "w"])
    nx, ny, dx, dy = int(g["nx"]), int(g["ny"]), float(g["dx"]), float(g["dy"])
    z = lambda: [np.zeros((nx, 1), dtype=np.int64)]
    n = p.npart
    scratch = lambda: [np.full(n, -1, dtype=np.int64)]
    # the sorter's bucket grid: x cells only, one bucket across y (core/sort/particle_sort.py:64-89,
    # simulation.py:696-698); origin half a cell below node 0 (particle_sort.py:193)
    args = dict(x_list=[p.x], y_list=[p.y], is_dead_list=[p.is_dead],
                attrs_list=[getattr(p, a) for a in ATTRS], x0s=[float(g["x0"]) - dx / 2],
                y0s=[float(g["y0"]) - dy / 2], nx=nx, ny=1, dx=dx, dy=ny * dy, npatches=1,
                bucket_count_list=z(), bucket_bound_min_list=z(), bucket_bound_max_list=z(),
                bucket_count_not_list=z(), bucket_start_counter_list=z(), particle_index_list=scratch(),
                particle_index_ref_list=scratch(), particle_index_target_list=scratch(),
                buf_list=[np.zeros(n)], reverse_x=0)
    return p, args


def test_sort_particles_patches_2d_vs_reference_golden(golden):
    g = golden("g6_sort_2d")
    p, args = _g6_lists(g)
    nbuf = kernels.sort_particles_patches_2d(**args)
    # bookkeeping the collision module reads: exact (dead slots inherit the bucket of the slot before them)
    assert np.array_equal(args["bucket_count_list"][0], g["bucket_count"])
    assert np.array_equal(args["bucket_bound_min_list"][0], g["bucket_bound_min"])
    assert np.array_equal(args["bucket_bound_max_list"][0], g["bucket_bound_max"])
    # the number of misplaced slots is a property of the keys, not of the permutation chosen
    assert nbuf == int(g["nbuf"])
    # inside a bucket the order is implementation defined (reference tests/test_sort.py:38-74): compare per
    # bucket the multiset of slots (id for live ones, deadness for all) and that attributes travelled with ids
    bmin, bmax = g["bucket_bound_min"].ravel(), g["bucket_bound_max"].ravel()
    ids, ids_ref = p._id.view(np.uint64), g["out__id"].view(np.uint64)
    for lo, hi in zip(bmin, bmax):
        assert (p.is_dead[lo:hi]).sum() == g["out_is_dead"][lo:hi].sum()
        live, live_ref = ~p.is_dead[lo:hi], ~g["out_is_dead"][lo:hi]
        assert np.array_equal(np.sort(ids[lo:hi][live]), np.sort(ids_ref[lo:hi][live_ref]))
    live = ~p.is_dead
    by_id = {a: dict(zip(g["in__id"].view(np.uint64)[~g["in_is_dead"]], g["in_" + a][~g["in_is_dead"]]))
             for a in ("x", "y", "ux", "uy", "uz", "w")}
    for a in by_id:
        assert np.array_equal(getattr(p, a)[live], np.array([by_id[a][i] for i in ids[live]])), a
    # live particles sit in the bucket their position says
    ix = np.floor((p.x[live] - args["x0s"][0]) / args["dx"]).astype(int)
    assert np.array_equal(ix, np.repeat(np.arange(bmin.size), [int((~p.is_dead[a:b]).sum()) for a, b in zip(bmin, bmax)]))
    # already sorted: nothing moves (reference tests/test_sort.py:140-148, g6's nbuf_again)
    before = {a: getattr(p, a).copy() for a in ATTRS + ["is_dead"]}
    assert kernels.sort_particles_patches_2d(**args) == int(g["nbuf_again"]) == 0
    for a, v in before.items():
        assert np.array_equal(getattr(p, a), v, equal_nan=True), a


def test_sort_particles_patches_reverse_and_out_of_range():
    """the two branches g6 does not take (core/sort/cpu2d.c:25-42): mirrored x order clamps, the plain order
    sends out-of-range particles to the last bucket -- against a line-by-line numpy restatement"""
    rng = np.random.default_rng(12)
    n, nx, ny, dx, dy = 5000, 12, 3, 1.0, 2.0
    x, y = rng.uniform(-2, nx + 2, n), rng.uniform(-1, ny * dy + 1, n)
    dead = rng.random(n) < 0.15
    for reverse in (0, 1):
        idx, icell = np.zeros(n, dtype=np.int64), 0
        for ip in range(n):
            if not dead[ip]:
                ix, iy = int(np.floor(x[ip] / dx)), int(np.floor(y[ip] / dy))
                if reverse:
                    ix, iy = min(max(ix, 0), nx - 1), min(max(iy, 0), ny - 1)
                    icell = iy + (nx - 1 - ix) * ny
                elif 0 <= ix < nx and 0 <= iy < ny:
                    icell = iy + ix * ny
                else:
                    icell = nx * ny - 1
            idx[ip] = icell
        cnt = np.bincount(idx, minlength=nx * ny)
        xs, ys, ds, tag = x.copy(), y.copy(), dead.copy(), np.arange(n, dtype=np.float64)
        z = lambda: [np.zeros((nx, ny), dtype=np.int64)]
        bc, bmin, bmax = z(), z(), z()
        s = lambda: [np.full(n, -1, dtype=np.int64)]
        moved = kernels.sort_particles_patches_2d([xs], [ys], [ds], [tag], [0.0], [0.0], nx, ny, dx, dy, 1, bc, bmin,
                                                  bmax, z(), z(), s(), s(), s(), [np.zeros(n)], reverse)
        assert np.array_equal(bc[0].ravel(), cnt)
        lo = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        assert np.array_equal(bmin[0].ravel(), lo) and np.array_equal(bmax[0].ravel(), lo + cnt)
        ref = np.repeat(np.arange(nx * ny), cnt)             # bucket every slot must hold afterwards
        assert moved == int((idx != ref).sum())
        src = tag.astype(np.int64)                           # where each slot's content came from
        assert np.array_equal(np.sort(src), np.arange(n))    # a permutation
        assert np.array_equal(idx[src], ref)
        assert np.array_equal(xs, x[src]) and np.array_equal(ys, y[src]) and np.array_equal(ds, dead[src])


def test_sort_particles_patches_3d_properties():
    rng = np.random.default_rng(13)
    n, nb, d = 20000, (5, 4, 3), (1.0, 1.5, 2.0)
    pos = [rng.uniform(0, nb[a] * d[a], n) for a in range(3)]
    dead = rng.random(n) < 0.1
    dead[0] = True
    key = np.zeros(n, dtype=np.int64)
    run = 0
    live_key = (np.floor(pos[2] / d[2]) + np.floor(pos[1] / d[1]) * nb[2] + np.floor(pos[0] / d[0]) * nb[1] * nb[2]).astype(int)
    for ip in range(n):                       # core/sort/cpu3d.c:22-58
        if not dead[ip]:
            run = live_key[ip]
        key[ip] = run
    cnt = np.bincount(key, minlength=int(np.prod(nb)))
    x, y, z, ds, tag = pos[0].copy(), pos[1].copy(), pos[2].copy(), dead.copy(), np.arange(n, dtype=np.float64)
    zz = lambda: [np.zeros(nb, dtype=np.int64)]
    bc, bmin, bmax = zz(), zz(), zz()
    s = lambda: [np.full(n, -1, dtype=np.int64)]
    moved = kernels.sort_particles_patches_3d([x], [y], [z], [ds], [tag], [0.0], [0.0], [0.0], *nb, *d, 1, bc, bmin, bmax,
                                              zz(), zz(), s(), s(), s(), [np.zeros(n)], 0)
    assert np.array_equal(bc[0].ravel(), cnt)
    ref = np.repeat(np.arange(cnt.size), cnt)
    src = tag.astype(np.int64)
    assert moved == int((key != ref).sum())
    assert np.array_equal(np.sort(src), np.arange(n)) and np.array_equal(key[src], ref)
    assert np.array_equal(z, pos[2][src]) and np.array_equal(ds, dead[src])


def _patches_from_g7(g):
    P = make_patches_2d(int(g["nx"]), int(g["ny"]), float(g["dx"]), float(g["dy"]), int(g["npx"]), int(g["npy"]))
    for k, p in enumerate(P):
        for a in p.fields.attrs:
            getattr(p.fields, a)[...] = g[f"in{k}_{a}"]
    return P


def test_sync_fields_dropins_vs_reference_golden(golden):
    g = golden("g7_sync_2d")
    P = _patches_from_g7(g)
    fl = [p.fields for p in P]
    kernels.sync_guard_fields_2d(fl, list(P), ["ex", "ey", "ez", "bx", "by", "bz"], 4, P.nx, P.ny, 3)
    kernels.sync_currents_2d(fl, list(P), 4, P.nx, P.ny, 3)
    for k, p in enumerate(P):
        for a in ["ex", "ey", "ez", "bx", "by", "bz"]:
            assert np.array_equal(getattr(p.fields, a), g[f"out{k}_{a}"]), (k, a)     # copies: bit exact
        for a in ["jx", "jy", "jz", "rho"]:
            # sums of up to four terms, added in the reference's boundary order
            assert_close(getattr(p.fields, a), g[f"out{k}_{a}"], 1e-15, what=f"{k} {a}")


def test_sync_fields_dropins_open_boundaries():
    """patches without a neighbour keep their guards (fields) / are not folded nor zeroed (currents)"""
    import oracle.sync as osync
    rng = np.random.default_rng(3)
    bc = {"xmin": "pml", "xmax": "pml", "ymin": "periodic", "ymax": "periodic"}
    P = make_patches_2d(24, 16, 1e-8, 1e-8, 3, 2, boundary_conditions=bc)
    Q = make_patches_2d(24, 16, 1e-8, 1e-8, 3, 2, boundary_conditions=bc)
    for p, q in zip(P, Q):
        for a in p.fields.attrs:
            v = rng.normal(size=p.fields.shape)
            getattr(p.fields, a)[...] = v
            getattr(q.fields, a)[...] = v
    E6 = ["ex", "ey", "ez", "bx", "by", "bz"]
    kernels.sync_guard_fields_2d([p.fields for p in P], list(P), E6, 6, P.nx, P.ny, 3)
    kernels.sync_currents_2d([p.fields for p in P], list(P), 6, P.nx, P.ny, 3)
    osync.sync_guard_fields_2d([q.fields for q in Q], list(Q), E6, 6, Q.nx, Q.ny, 3)
    osync.sync_currents_2d([q.fields for q in Q], list(Q), 6, Q.nx, Q.ny, 3)
    for p, q in zip(P, Q):
        for a in E6:
            assert np.array_equal(getattr(p.fields, a), getattr(q.fields, a)), a
        for a in ["jx", "jy", "jz", "rho"]:
            assert_close(getattr(p.fields, a), getattr(q.fields, a), 1e-15, what=a)


def test_device_tile_sort_pinned_to_g6(golden):
    """the resident engine's sort (lpa_sort_tiles_2d: tile-major cell key, dead slots dropped) on the reference's
    fixture: per x-bucket (= x-cell, core/sort/cpu2d.c:23) histogram of the live particles and their id set"""
    from lambdapic_amd.engine import PicEngine2D
    g = golden("g6_sort_2d")
    p = particles_from(g, "in_", ["x", "y", "ux", "uy", "uz", "w"])
    p.inv_gamma[:] = 1 / np.sqrt(1 + p.ux ** 2 + p.uy ** 2 + p.uz ** 2)
    nx, ny, dx, dy = int(g["nx"]), int(g["ny"]), float(g["dx"]), float(g["dy"])
    bmin, bmax = g["bucket_bound_min"].ravel(), g["bucket_bound_max"].ravel()
    live_ref = ~g["out_is_dead"]
    hist_ref = np.array([int(live_ref[a:b].sum()) for a, b in zip(bmin, bmax)])
    for order in (0, 1):
        eng = PicEngine2D(nx, ny, dx, dy, device="cuda:0", x0=float(g["x0"]), y0=float(g["y0"]),
                          block_particles=1024, order=order)
        eng.add_species(-1.6e-19, 9.1e-31, capacity=p.npart + 64)
        eng.species[0].upload([p])
        eng.sort(0)
        out = eng.species[0].download()
        assert out["x"].size == live_ref.sum()
        # device cell counts summed over y == the reference's live count per bucket
        ix = np.floor((out["x"] - (float(g["x0"]) - dx / 2)) / dx).astype(int)
        assert np.array_equal(np.bincount(ix, minlength=nx), hist_ref)
        if order == 0:       # cell-major inside the (single) tile column: x-buckets appear in order within a tile
            tile = ix // 8
            assert np.all(np.diff(tile) >= 0)
        # the same particles, attributes attached to their ids
        o, r = np.argsort(out["_id"].view(np.uint64)), np.argsort(g["out__id"].view(np.uint64)[live_ref])
        assert np.array_equal(out["_id"].view(np.uint64)[o], g["out__id"].view(np.uint64)[live_ref][r])
        for a in ("x", "y", "ux", "w"):
            assert np.array_equal(out[a][o], g["out_" + a][live_ref][r]), a


def test_sort_dropins_vs_reference_golden_variants(golden):
    """g15: the reference's 3-D sort (6 x 5 x 4 buckets, out-of-range -> last bucket) and its mirrored 2-D sort with
    2-D buckets: bookkeeping exact, number of moved slots exact, every slot holds a particle of its bucket, attributes
    travel with their particle (per-bucket multisets equal the reference's)"""
    g = golden("g15_sort_variants")
    for tag, axes, rev in (("a", "xyz", 0), ("b", "xy", 1)):
        nb, d, o = tuple(int(v) for v in g[f"{tag}_nb"]), tuple(g[f"{tag}_d"]), tuple(g[f"{tag}_o"])
        arrs = {k: g[f"{tag}_in_{k}"].copy() for k in list(axes) + ["w", "tag"]}
        dead = g[f"{tag}_in_is_dead"].copy()
        n = dead.size
        z = lambda: [np.zeros(nb, dtype=np.int64)]
        cnt, bmin, bmax = z(), z(), z()
        s = lambda: [np.full(n, -1, dtype=np.int64)]
        attrs = [arrs[k] for k in list(axes) + ["w", "tag"]]
        if len(nb) == 3:
            moved = kernels.sort_particles_patches_3d([arrs["x"]], [arrs["y"]], [arrs["z"]], [dead], attrs, [o[0]], [o[1]],
                                                      [o[2]], *nb, *d, 1, cnt, bmin, bmax, z(), z(), s(), s(), s(),
                                                      [np.zeros(n)], rev)
        else:
            moved = kernels.sort_particles_patches_2d([arrs["x"]], [arrs["y"]], [dead], attrs, [o[0]], [o[1]], *nb, *d, 1,
                                                      cnt, bmin, bmax, z(), z(), s(), s(), s(), [np.zeros(n)], rev)
        assert np.array_equal(cnt[0], g[f"{tag}_bucket_count"])
        assert np.array_equal(bmin[0], g[f"{tag}_bucket_bound_min"])
        assert np.array_equal(bmax[0], g[f"{tag}_bucket_bound_max"])
        assert moved == int(g[f"{tag}_nbuf"])
        src, src_ref = arrs["tag"].astype(np.int64), g[f"{tag}_out_tag"].astype(np.int64)
        assert np.array_equal(np.sort(src), np.arange(n))
        for k in list(axes) + ["w"]:
            assert np.array_equal(arrs[k], g[f"{tag}_in_{k}"][src]), k
        assert np.array_equal(dead, g[f"{tag}_in_is_dead"][src])
        for lo, hi in zip(g[f"{tag}_bucket_bound_min"].ravel(), g[f"{tag}_bucket_bound_max"].ravel()):
            assert np.array_equal(np.sort(src[lo:hi]), np.sort(src_ref[lo:hi]))      # same slots' contents per bucket


def test_sync_particles_dropins_vs_reference_golden(golden):
    """`get_npart_to_extend_2d` + `fill_particles_from_boundary_2d` (core/patch/sync_particles_2d.c:204,322) on the
    reference's fixture (g7: 2 x 2 periodic patches, everything shifted across two boundaries): counts, the slot every
    incoming particle lands in, the +- L shifts and the dead pattern are the reference's, bit for bit"""
    g = golden("g7_sync_2d")
    dx, dy = float(g["dx"]), float(g["dy"])
    P = make_patches_2d(int(g["nx"]), int(g["ny"]), dx, dy, int(g["npx"]), int(g["npy"]))
    for k, p in enumerate(P):
        q = p.particles[0]
        q.initialize(g[f"pin{k}_x"].size)
        for a in ["x", "y", "ux", "w", "_id"]:
            getattr(q, a)[:] = g[f"pin{k}_{a}"]
        q.is_dead[:] = g[f"pin{k}_is_dead"]
    parts = [p.particles[0] for p in P]
    ext, inc, outg, alive = kernels.get_npart_to_extend_2d(parts, list(P), P.npatches, dx, dy)
    assert np.array_equal(alive, g["npart_alive"])
    for q, n in zip(parts, ext):                       # Patches.sync_particles, core/patch/patch.py:705-742
        if n > 0:
            q.extend(int(n))
    kernels.fill_particles_from_boundary_2d(parts, list(P), inc, outg, P.npatches, dx, dy, P.xmin_global, P.xmax_global,
                                            P.ymin_global, P.ymax_global, parts[0].attrs)
    for k, q in enumerate(parts):
        assert q.npart == g[f"pout{k}_x"].size
        assert np.array_equal(q.is_dead, g[f"pout{k}_is_dead"]), k
        live = ~q.is_dead
        for a in ["x", "y", "ux", "w"]:
            assert np.array_equal(getattr(q, a)[live], g[f"pout{k}_{a}"][live]), (k, a)
        assert np.array_equal(q.id[live], g[f"pout{k}__id"].view(np.uint64)[live])
        assert np.all(np.isnan(q.x[q.is_dead & np.isnan(g[f"pout{k}_x"])]))


def test_maxwell_patch_drivers_vs_reference_golden(golden):
    """`update_efield_patches_2d/3d`, `update_bfield_patches_2d/3d` with the reference's argument lists
    (core/maxwell/cpu.py:38-79,115-158) on g5 (recorded from the reference's own FDTD), two patches per call"""
    for dim, name in ((2, "g5_fdtd_2d"), (3, "g5_fdtd_3d")):
        g = golden(name)
        n = tuple(int(g[k]) for k in ("nx", "ny", "nz")[:dim])
        d = tuple(float(g[k]) for k in ("dx", "dy", "dz")[:dim])
        ng, dt = int(g["ng"]), float(g["dt"])
        names = ["ex", "ey", "ez", "bx", "by", "bz", "jx", "jy", "jz"]
        lists = {a: [g["in_" + a].copy(), g["in_" + a].copy()] for a in names}      # the same patch twice
        args_e = [lists[a] for a in names] + [2, *d, dt, *n, ng]
        (kernels.update_efield_patches_2d if dim == 2 else kernels.update_efield_patches_3d)(*args_e)
        for k in range(2):
            for a in ["ex", "ey", "ez"]:
                assert_close(lists[a][k], g[f"outE_{a}"], 1e-14, what=f"{name} {a}")
        args_b = [lists[a] for a in names[:6]] + [2, *d, dt, *n, ng]
        (kernels.update_bfield_patches_2d if dim == 2 else kernels.update_bfield_patches_3d)(*args_b)
        for k in range(2):
            for a in ["bx", "by", "bz"]:
                assert_close(lists[a][k], g[f"outB_{a}"], 1e-14, what=f"{name} {a}")


# ---- 3-D patch lists on 2 x 2 x 2 periodic patches (g16 / g17: every neighbour class is ANOTHER patch) ---------------
def _patches_3d(g, d):
    from lambdapic_amd.patch import make_patches_3d
    npp, npatch = tuple(int(v) for v in g["npp"]), tuple(int(v) for v in g["npatch"])
    return make_patches_3d(tuple(a * b for a, b in zip(npp, npatch)), d, npatch, 3), npp


def test_sync_fields_3d_dropins_vs_reference_golden(golden):
    """`sync_guard_fields_3d(fields_list, patches_list, attrs, npatches, nx, ny, nz, ng)` and `sync_currents_3d(fields_list,
    patches_list, npatches, nx, ny, nz, ng)` (core/patch/sync_fields3d.c:350,84) with the reference's argument lists on its
    own fixture: guard copies and current folds (26 neighbours, consumed guards zeroed) exact; and against the oracle on an
    open (non-periodic) patch grid with random doubles: copies bit exact, folds 1e-14"""
    import copy
    from oracle import sync
    g = golden("g16_sync_fields_3d_patches")
    P, npp = _patches_3d(g, (1e-7, 1.5e-7, 0.8e-7))
    assert np.array_equal(np.stack([p.neighbor_ipatch for p in P]), g["neighbor_ipatch"])
    names = ("ex", "by", "bz", "jx", "jy", "jz", "rho")
    for k, p in enumerate(P):
        for a in names:
            getattr(p.fields, a)[...] = g[f"in{k}_{a}"] / float(g["scale"])
    fl = [p.fields for p in P]
    kernels.sync_guard_fields_3d(fl, list(P), ["ex", "by", "bz"], 8, *npp, 3)
    kernels.sync_currents_3d(fl, list(P), 8, *npp, 3)
    for k, p in enumerate(P):
        for a in names:
            assert np.array_equal(getattr(p.fields, a) * float(g["scale"]), g[f"out{k}_{a}"].astype(float)), (k, a)
    # open box, 3 x 2 x 2 patches, doubles: missing neighbours (-1) on the outer faces
    from lambdapic_amd.patch import make_patches_3d
    bc = {k: "pml" for k in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")}
    Q = make_patches_3d((18, 12, 14), (1.0, 1.0, 1.0), (3, 2, 2), 3, boundary_conditions=bc)
    rng = np.random.default_rng(3)
    for p in Q:
        for a in p.fields.attrs:
            getattr(p.fields, a)[...] = rng.normal(size=p.fields.shape)
    R = copy.deepcopy(Q)
    allE = ["ex", "ey", "ez", "bx", "by", "bz"]
    kernels.sync_guard_fields_3d([p.fields for p in Q], list(Q), allE, 12, 6, 6, 7, 3)
    kernels.sync_currents_3d([p.fields for p in Q], list(Q), 12, 6, 6, 7, 3)
    sync.sync_guard_fields_3d([p.fields for p in R], list(R), allE, 12, 6, 6, 7, 3)
    sync.sync_currents_3d([p.fields for p in R], list(R), 12, 6, 6, 7, 3)
    assert (np.stack([p.neighbor_ipatch for p in Q]) < 0).sum() > 100
    for p, r in zip(Q, R):
        for a in allE:
            assert np.array_equal(getattr(p.fields, a), getattr(r.fields, a)), a
        for a in ("jx", "jy", "jz", "rho"):
            assert_close(getattr(p.fields, a), getattr(r.fields, a), 1e-14, what=a)


def test_sync_particles_3d_dropins_vs_reference_golden(golden):
    """`get_npart_to_extend_3d(particles_list, patch_list, npatches, dx, dy, dz)` + `fill_particles_from_boundary_3d(…)`
    (core/patch/sync_particles_3d.c:365,484) on the reference's fixture (2 x 2 x 2 periodic patches, leavers through all
    26 boundary classes into seven different neighbours): the four count arrays, the slot every incoming particle lands
    in, the +- L shifts, the is_dead pattern and the blanked dead slots are the reference's, bit for bit"""
    g = golden("g17_sync_particles_3d_patches")
    d = tuple(float(v) for v in g["d"])
    P, _ = _patches_3d(g, d)
    for k, p in enumerate(P):
        q = p.particles[0]
        q.initialize(g[f"pin{k}_x"].size)
        for a in ("x", "y", "z", "ux", "w", "_id", "is_dead"):
            getattr(q, a)[:] = g[f"pin{k}_{a}"]
    parts = [p.particles[0] for p in P]
    ext, inc, outg, alive = kernels.get_npart_to_extend_3d(parts, list(P), 8, *d)
    for got, key in ((ext, "npart_to_extend"), (inc, "npart_incoming"), (outg, "npart_outgoing"), (alive, "npart_alive")):
        assert np.array_equal(np.asarray(got).ravel(), g[key].ravel()), key
    assert int(np.asarray(outg).reshape(8, 26).sum(0).min()) > 0
    for q, n in zip(parts, ext):                       # Patches.sync_particles, core/patch/patch.py:739-763
        if n > 0:
            q.extend(int(n))
    kernels.fill_particles_from_boundary_3d(parts, list(P), inc, outg, 8, *d, P.xmin_global, P.xmax_global,
                                            P.ymin_global, P.ymax_global, P.zmin_global, P.zmax_global, parts[0].attrs)
    for k, q in enumerate(parts):
        assert q.npart == g[f"pout{k}_x"].size
        assert np.array_equal(q.is_dead, g[f"pout{k}_is_dead"]), k
        live = ~q.is_dead
        for a in ("x", "y", "z", "ux", "w", "_id"):
            assert np.array_equal(getattr(q, a)[live].view(np.uint64), g[f"pout{k}_{a}"][live].view(np.uint64)), (k, a)
        for a in "xyz":
            assert np.isnan(getattr(q, a)[~live]).all(), (k, a)
