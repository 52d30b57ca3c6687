#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own kernels.  Runs only in the build container
(needs /root/reference and `make -C oracle ref`); the fixtures it writes are plain data (inputs and
expected outputs) and are what travels -- no reference source or binary is stored under tests/.

What runs:
  * the reference's compiled C extensions (oracle/_ref, built from the read-only tree by
    oracle/Makefile with the reference's own flags) on duck-typed field / particle bags;
  * the reference's Python FDTD (core/maxwell/cpu.py), executed as plain Python: the file is loaded
    by path with `numba.njit` bound to the identity decorator (what NUMBA_DISABLE_JIT does) because
    numba is not installed here.

Usage:  python tests/golden/gen_golden.py
"""
from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference/src/lambdapic")
OUT = Path(__file__).resolve().parent

import oracle  # noqa: E402
from oracle import driver  # noqa: E402
from lambdapic_amd.fields import Fields2D, Fields3D  # noqa: E402
from lambdapic_amd.particles import ParticlesBase  # noqa: E402
from lambdapic_amd.patch import make_patches_2d  # noqa: E402

C = 299792458.0
QE = -1.602176634e-19
ME = 9.1093837139e-31
SEED = 20260722

PATTRS = ["x", "y", "z", "w", "ux", "uy", "uz", "inv_gamma"]
PEB = ["ex_part", "ey_part", "ez_part", "bx_part", "by_part", "bz_part"]


def load_ref_maxwell():
    """core/maxwell/cpu.py as plain Python (decorators -> identity, prange -> range)."""
    nb = types.ModuleType("numba")
    nb.njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    nb.prange = range
    sys.modules.setdefault("numba", nb)
    for name in ("lambdapic", "lambdapic.core", "lambdapic.core.utils", "lambdapic.core.maxwell"):
        sys.modules.setdefault(name, types.ModuleType(name))
    js = types.ModuleType("lambdapic.core.utils.jit_spinner")
    js.jit_spinner = lambda f: f
    sys.modules["lambdapic.core.utils.jit_spinner"] = js
    spec = importlib.util.spec_from_file_location("lambdapic.core.maxwell.cpu",
                                                  REF / "core/maxwell/cpu.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def rand_fields(f, rng, e_amp=1e12, b_amp=1e4):
    for a in ("ex", "ey", "ez"):
        getattr(f, a)[...] = rng.normal(size=f.shape) * e_amp
    for a in ("bx", "by", "bz"):
        getattr(f, a)[...] = rng.normal(size=f.shape) * b_amp


def rand_particles_2d(f, n, rng, u_scale=1.0, ipatch=0):
    p = ParticlesBase(ipatch, 0)
    p.initialize(n)
    p.x[:] = f.x0 + rng.uniform(-0.5, f.nx - 0.5, n) * f.dx
    p.y[:] = f.y0 + rng.uniform(-0.5, f.ny - 0.5, n) * f.dy
    for a in ("ux", "uy", "uz"):
        getattr(p, a)[:] = rng.normal(size=n) * u_scale
    p.inv_gamma[:] = 1 / np.sqrt(1 + p.ux ** 2 + p.uy ** 2 + p.uz ** 2)
    p.w[:] = rng.uniform(0.5, 1.5, n) * 1e27 * f.dx * f.dy / 10
    # edge cases: dead slots, NaN position, particles sitting on the patch edge (deposit through
    # the wrap), fast movers crossing a cell boundary
    p.is_dead[::17] = True
    p.x[5] = np.nan
    p.y[11] = np.nan
    p.x[20:24] = f.x0 + np.array([-0.49, -0.2, f.nx - 0.51, f.nx - 0.8]) * f.dx
    p.y[24:28] = f.y0 + np.array([-0.49, -0.2, f.ny - 0.51, f.ny - 0.8]) * f.dy
    return p


def rand_particles_3d(f, n, rng, u_scale=1.0):
    p = rand_particles_2d(f, n, rng, u_scale)
    p.z[:] = f.z0 + rng.uniform(-0.5, f.nz - 0.5, n) * f.dz
    p.w[:] = rng.uniform(0.5, 1.5, n) * 1e27 * f.dx * f.dy * f.dz / 10
    p.z[13] = np.nan
    p.z[28:32] = f.z0 + np.array([-0.49, -0.2, f.nz - 0.51, f.nz - 0.8]) * f.dz
    return p


def snap(obj, names, prefix):
    return {prefix + n: np.array(getattr(obj, n), copy=True) for n in names}


def g1_fused_2d(rng):
    mod = oracle.ref_module("pusher", "unified_pusher_2d")
    nx, ny, ng, dx, dy = 16, 12, 3, 4e-8, 5e-8
    dt = 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2))
    fl = [Fields2D(nx, ny, dx, dy, 3 * dx, -2 * dy, ng), Fields2D(nx, ny, dx, dy, 19 * dx, -2 * dy, ng)]
    pl = []
    out = dict(nx=nx, ny=ny, ng=ng, dx=dx, dy=dy, dt=dt, q=QE, m=ME, npatches=2)
    for k, f in enumerate(fl):
        rand_fields(f, rng)
        p = rand_particles_2d(f, 1500, rng, u_scale=1.0 if k == 0 else 0.05, ipatch=k)
        pl.append(p)
        out.update({f"x0_{k}": f.x0, f"y0_{k}": f.y0})
        out.update(snap(f, f.attrs[:6], f"in{k}_"))
        out.update(snap(p, PATTRS + ["is_dead"], f"in{k}_"))
    mod.unified_boris_pusher_cpu_2d(pl, fl, 2, dt, QE, ME)
    for k, (f, p) in enumerate(zip(fl, pl)):
        out.update(snap(f, ["rho", "jx", "jy", "jz"], f"out{k}_"))
        out.update(snap(p, PATTRS + PEB, f"out{k}_"))
    np.savez_compressed(OUT / "g1_fused_2d.npz", **out)


def g2_fused_3d(rng):
    mod = oracle.ref_module("pusher", "unified_pusher_3d")
    nx, ny, nz, ng, dx, dy, dz = 8, 6, 7, 3, 4e-8, 5e-8, 6e-8
    dt = 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2 + dz ** -2))
    f = Fields3D(nx, ny, nz, dx, dy, dz, 3 * dx, -2 * dy, 5 * dz, ng)
    rand_fields(f, rng)
    p = rand_particles_3d(f, 1200, rng)
    out = dict(nx=nx, ny=ny, nz=nz, ng=ng, dx=dx, dy=dy, dz=dz, dt=dt, q=QE, m=ME,
               x0=f.x0, y0=f.y0, z0=f.z0)
    out.update(snap(f, f.attrs[:6], "in_"))
    out.update(snap(p, PATTRS + ["is_dead"], "in_"))
    mod.unified_boris_pusher_cpu_3d([p], [f], 1, dt, QE, ME)
    out.update(snap(f, ["rho", "jx", "jy", "jz"], "out_"))
    out.update(snap(p, PATTRS + PEB, "out_"))
    np.savez_compressed(OUT / "g2_fused_3d.npz", **out)


def g3_deposit(rng):
    nx, ny, ng, dx, dy = 16, 12, 3, 4e-8, 5e-8
    dt = 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2))
    f = Fields2D(nx, ny, dx, dy, 3 * dx, -2 * dy, ng)
    p = rand_particles_2d(f, 1500, rng)
    out = dict(nx=nx, ny=ny, ng=ng, dx=dx, dy=dy, dt=dt, q=QE, x0=f.x0, y0=f.y0)
    out.update(snap(p, PATTRS + ["is_dead"], "in_"))
    oracle.ref_module("current", "cpu2d").current_deposition_cpu_2d([f], [p], 1, dt, QE)
    out.update(snap(f, ["rho", "jx", "jy", "jz"], "out_"))
    np.savez_compressed(OUT / "g3_deposit_2d.npz", **out)

    nx, ny, nz, dz = 8, 6, 7, 6e-8
    dt = 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2 + dz ** -2))
    f = Fields3D(nx, ny, nz, dx, dy, dz, 3 * dx, -2 * dy, 5 * dz, ng)
    p = rand_particles_3d(f, 1200, rng, u_scale=0.5)
    # the standalone 3-D depositor only wraps negative indices (current/cpu3d.c:56-116 with
    # INDEX3): keep particles one cell inside the upper edges so no index reaches NX
    for a, n_, d_, o_ in (("x", nx, dx, f.x0), ("y", ny, dy, f.y0), ("z", nz, dz, f.z0)):
        v = getattr(p, a)
        np.clip(v, o_ - 0.49 * d_, o_ + (n_ - 0.5) * d_, out=v, where=~np.isnan(v))
    out = dict(nx=nx, ny=ny, nz=nz, ng=ng, dx=dx, dy=dy, dz=dz, dt=dt, q=QE,
               x0=f.x0, y0=f.y0, z0=f.z0)
    out.update(snap(p, PATTRS + ["is_dead"], "in_"))
    oracle.ref_module("current", "cpu3d").current_deposition_cpu_3d([f], [p], 1, dt, QE)
    out.update(snap(f, ["rho", "jx", "jy", "jz"], "out_"))
    np.savez_compressed(OUT / "g3_deposit_3d.npz", **out)


def g4_interp(rng):
    nx, ny, ng, dx, dy = 16, 12, 3, 4e-8, 5e-8
    f = Fields2D(nx, ny, dx, dy, 3 * dx, -2 * dy, ng)
    rand_fields(f, rng)
    p = rand_particles_2d(f, 1000, rng)
    p.x[5] = f.x0
    p.y[11] = f.y0   # standalone interpolator does not skip NaN positions: keep them finite
    out = dict(nx=nx, ny=ny, ng=ng, dx=dx, dy=dy, x0=f.x0, y0=f.y0)
    out.update(snap(f, f.attrs[:6], "in_"))
    out.update(snap(p, ["x", "y", "is_dead"], "in_"))
    oracle.ref_module("interpolation", "cpu2d").interpolation_patches_2d([p], [f], 1)
    out.update(snap(p, PEB, "out_"))
    np.savez_compressed(OUT / "g4_interp_2d.npz", **out)

    nx, ny, nz, dz = 8, 6, 7, 6e-8
    f = Fields3D(nx, ny, nz, dx, dy, dz, 3 * dx, -2 * dy, 5 * dz, ng)
    rand_fields(f, rng)
    p = rand_particles_3d(f, 800, rng)
    p.x[5] = f.x0
    p.y[11] = f.y0
    p.z[13] = f.z0
    out = dict(nx=nx, ny=ny, nz=nz, ng=ng, dx=dx, dy=dy, dz=dz, x0=f.x0, y0=f.y0, z0=f.z0)
    out.update(snap(f, f.attrs[:6], "in_"))
    out.update(snap(p, ["x", "y", "z", "is_dead"], "in_"))
    oracle.ref_module("interpolation", "cpu3d").interpolation_patches_3d([p], [f], 1)
    out.update(snap(p, PEB, "out_"))
    np.savez_compressed(OUT / "g4_interp_3d.npz", **out)


def g5_fdtd(rng, mx):
    nx, ny, ng, dx, dy = 24, 20, 3, 4e-8, 5e-8
    dt = 0.5 * 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2))
    f = Fields2D(nx, ny, dx, dy, 0.0, 0.0, ng)
    rand_fields(f, rng)
    for a in ("jx", "jy", "jz"):
        getattr(f, a)[...] = rng.normal(size=f.shape) * 1e15
    out = dict(nx=nx, ny=ny, ng=ng, dx=dx, dy=dy, dt=dt)
    out.update(snap(f, f.attrs[:9], "in_"))
    mx.update_efield_2d(f.ex, f.ey, f.ez, f.bx, f.by, f.bz, f.jx, f.jy, f.jz, dx, dy, dt, nx, ny, ng)
    out.update(snap(f, ["ex", "ey", "ez"], "outE_"))
    mx.update_bfield_2d(f.ex, f.ey, f.ez, f.bx, f.by, f.bz, dx, dy, dt, nx, ny, ng)
    out.update(snap(f, ["bx", "by", "bz"], "outB_"))
    np.savez_compressed(OUT / "g5_fdtd_2d.npz", **out)

    nx, ny, nz, dz = 10, 8, 6, 6e-8
    dt = 0.5 * 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2 + dz ** -2))
    f = Fields3D(nx, ny, nz, dx, dy, dz, 0.0, 0.0, 0.0, ng)
    rand_fields(f, rng)
    for a in ("jx", "jy", "jz"):
        getattr(f, a)[...] = rng.normal(size=f.shape) * 1e15
    out = dict(nx=nx, ny=ny, nz=nz, ng=ng, dx=dx, dy=dy, dz=dz, dt=dt)
    out.update(snap(f, f.attrs[:9], "in_"))
    mx.update_efield_3d(f.ex, f.ey, f.ez, f.bx, f.by, f.bz, f.jx, f.jy, f.jz, dx, dy, dz, dt,
                        nx, ny, nz, ng)
    out.update(snap(f, ["ex", "ey", "ez"], "outE_"))
    mx.update_bfield_3d(f.ex, f.ey, f.ez, f.bx, f.by, f.bz, dx, dy, dz, dt, nx, ny, nz, ng)
    out.update(snap(f, ["bx", "by", "bz"], "outB_"))
    np.savez_compressed(OUT / "g5_fdtd_3d.npz", **out)


class RefSorter:
    """drives the reference's sort_particles_patches_2d like ParticleSort2D.__call__
    (core/sort/particle_sort.py:196-211) with the default 1-D x buckets."""

    def __init__(self, patches, ispec):
        self.mod = oracle.ref_module("sort", "cpu2d")
        self.patches, self.ispec = patches, ispec
        self.nxb, self.nyb = patches.nx, 1
        n = patches.npatches
        mk = lambda: [np.zeros((self.nxb, self.nyb), dtype=np.int64) for _ in range(n)]
        self.count, self.bmin, self.bmax, self.cnot, self.start = mk(), mk(), mk(), mk(), mk()
        self.nbuf_last = 0

    def __call__(self):
        P, s = self.patches, self.ispec
        parts = [p.particles[s] for p in P]
        attrs = parts[0].attrs
        attrs_list = [getattr(q, a) for q in parts for a in attrs]
        scratch = lambda dt: [np.zeros(q.npart, dtype=dt) for q in parts]
        self.nbuf_last = self.mod.sort_particles_patches_2d(
            [q.x for q in parts], [q.y for q in parts], [q.is_dead for q in parts], attrs_list,
            [p.x0 - p.dx / 2 for p in P], [p.y0 - p.dy / 2 for p in P],
            self.nxb, self.nyb, P.dx, P.ny * P.dy, P.npatches,
            self.count, self.bmin, self.bmax, self.cnot, self.start,
            scratch(np.int64), scratch(np.int64), scratch(np.int64), scratch(np.float64), 0)
        return self.nbuf_last


def ref_sync_particles(patches, ispec, dx, dy):
    """Patches.sync_particles for one species (core/patch/patch.py:705-742) on the ref extension"""
    mod = oracle.ref_module("patch", "sync_particles_2d")
    parts = [p.particles[ispec] for p in patches]
    ext, inc, outg, alive = mod.get_npart_to_extend_2d(parts, list(patches), patches.npatches, dx, dy)
    for q, n in zip(parts, ext):
        if n > 0:
            q.extend(int(n))
    mod.fill_particles_from_boundary_2d(parts, list(patches), inc, outg, patches.npatches, dx, dy,
                                        patches.xmin_global, patches.xmax_global,
                                        patches.ymin_global, patches.ymax_global, parts[0].attrs)
    return alive


def ref_kernels(mx, sorters):
    sf = oracle.ref_module("patch", "sync_fields2d")
    up = oracle.ref_module("pusher", "unified_pusher_2d")

    def upd_e(f, dt):
        mx.update_efield_2d(f.ex, f.ey, f.ez, f.bx, f.by, f.bz, f.jx, f.jy, f.jz, f.dx, f.dy, dt,
                            f.nx, f.ny, f.n_guard)

    def upd_b(f, dt):
        mx.update_bfield_2d(f.ex, f.ey, f.ez, f.bx, f.by, f.bz, f.dx, f.dy, dt, f.nx, f.ny, f.n_guard)

    return driver.KernelSet(
        unified=up.unified_boris_pusher_cpu_2d, update_e=upd_e, update_b=upd_b,
        sync_guard=sf.sync_guard_fields_2d, sync_currents=sf.sync_currents_2d,
        sync_particles=ref_sync_particles, sort=lambda P, s: sorters[s](),
        reset=oracle.ref_module("current", "cpu2d").reset_current_cpu_2d)


def g6_sort(rng):
    P = make_patches_2d(16, 8, 4e-8, 4e-8, 1, 1)
    p = P[0]
    q = p.particles[0]
    n = 3000
    q.initialize(n)
    q.x[:] = rng.uniform(-0.5, p.nx - 0.5, n) * p.dx
    q.y[:] = rng.uniform(-0.5, p.ny - 0.5, n) * p.dy
    for a in ("ux", "uy", "uz", "w"):
        getattr(q, a)[:] = rng.normal(size=n)
    q.is_dead[rng.random(n) < 0.1] = True
    q.is_dead[0] = True   # leading dead particle inherits bucket 0 (sort/cpu2d.c:18,44-52)
    out = dict(nx=p.nx, ny=p.ny, dx=p.dx, dy=p.dy, x0=p.x0, y0=p.y0)
    out.update(snap(q, ["x", "y", "ux", "uy", "uz", "w", "_id", "is_dead"], "in_"))
    s = RefSorter(P, 0)
    out["nbuf"] = s()
    out["bucket_count"] = s.count[0].copy()
    out["bucket_bound_min"] = s.bmin[0].copy()
    out["bucket_bound_max"] = s.bmax[0].copy()
    out.update(snap(q, ["x", "y", "ux", "uy", "uz", "w", "_id", "is_dead"], "out_"))
    out["nbuf_again"] = s()   # already sorted -> nothing moves (tests/test_sort.py:140-148)
    np.savez_compressed(OUT / "g6_sort_2d.npz", **out)


def g7_sync(rng):
    nx = ny = 16
    dx = dy = 4e-8
    P = make_patches_2d(nx, ny, dx, dy, 2, 2)
    sf = oracle.ref_module("patch", "sync_fields2d")
    out = dict(nx=nx, ny=ny, dx=dx, dy=dy, npx=2, npy=2, ng=3)
    for k, p in enumerate(P):
        for a in p.fields.attrs:
            getattr(p.fields, a)[...] = rng.normal(size=p.fields.shape)
        out.update(snap(p.fields, p.fields.attrs, f"in{k}_"))
    fl = [p.fields for p in P]
    sf.sync_guard_fields_2d(fl, list(P), ["ex", "ey", "ez", "bx", "by", "bz"], 4, P.nx, P.ny, 3)
    sf.sync_currents_2d(fl, list(P), 4, P.nx, P.ny, 3)
    for k, p in enumerate(P):
        out.update(snap(p.fields, p.fields.attrs, f"out{k}_"))
    # particle migration: everything shifted by +0.8 cell in x and -0.6 cell in y
    for k, p in enumerate(P):
        q = p.particles[0]
        n = 400
        q.initialize(n)
        q.x[:] = p.x0 + rng.uniform(-0.5, p.nx - 0.5, n) * dx + 0.8 * dx
        q.y[:] = p.y0 + rng.uniform(-0.5, p.ny - 0.5, n) * dy - 0.6 * dy
        q.ux[:] = rng.normal(size=n)
        q.w[:] = rng.uniform(1, 2, n)
        q.is_dead[::9] = True
        out.update(snap(q, ["x", "y", "ux", "w", "_id", "is_dead"], f"pin{k}_"))
    alive = ref_sync_particles(P, 0, dx, dy)
    out["npart_alive"] = np.asarray(alive)
    for k, p in enumerate(P):
        out.update(snap(p.particles[0], ["x", "y", "ux", "w", "_id", "is_dead"], f"pout{k}_"))
    np.savez_compressed(OUT / "g7_sync_2d.npz", **out)


def g8_trace(mx):
    """C1-like multi-step trace with the reference kernels: 32x32 periodic,
    2x2 patches, e- 8 ppc, thermal u ~ N(0, 0.05), n = n_c(0.8 um), dx = lambda/20."""
    lam = 0.8e-6
    nx = ny = 32
    dx = dy = lam / 20
    dt = 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2))
    omega = 2 * np.pi * C / lam
    nc = driver.EPSILON_0 * ME * omega ** 2 / QE ** 2
    nsteps, ppc, uth = 40, 8, 0.05
    P = make_patches_2d(nx, ny, dx, dy, 2, 2)
    rng = np.random.default_rng(SEED + 8)
    driver.load_uniform_plasma(P, 0, ppc, nc, uth, rng)
    out = dict(nx=nx, ny=ny, dx=dx, dy=dy, dt=dt, npx=2, npy=2, ng=3, q=QE, m=ME, nsteps=nsteps,
               ppc=ppc, density=nc, uth=uth, seed=SEED + 8)
    for k, p in enumerate(P):
        out.update(snap(p.particles[0], ["x", "y", "ux", "uy", "uz", "inv_gamma", "w", "_id"],
                        f"in{k}_"))
    ks = ref_kernels(mx, [RefSorter(P, 0)])
    tr = {k: [] for k in ("field_energy", "kinetic_energy", "charge", "jx", "jy", "jz", "nalive")}
    for _ in range(nsteps):
        driver.step(P, ks, dt, [(QE, ME)])
        tr["field_energy"].append(driver.field_energy(P))
        tr["kinetic_energy"].append(driver.kinetic_energy(P, 0, ME))
        tr["charge"].append(driver.total_charge(P))
        jx, jy, jz = driver.current_sums(P)
        tr["jx"].append(jx), tr["jy"].append(jy), tr["jz"].append(jz)
        tr["nalive"].append(sum(int((~p.particles[0].is_dead).sum()) for p in P))
    for k, v in tr.items():
        out["trace_" + k] = np.array(v)
    for k, p in enumerate(P):
        out.update(snap(p.fields, ["ex", "ey", "ez", "bx", "by", "bz", "rho"], f"final{k}_"))
    np.savez_compressed(OUT / "g8_trace_2d.npz", **out)


def load_ref_cpml():
    """core/fields.py + core/boundary/cpml.py as plain Python (same numba binding as the FDTD)"""
    load_ref_maxwell()                      # installs the numba / jit_spinner stand-ins
    for name in ("lambdapic.core.boundary",):
        sys.modules.setdefault(name, types.ModuleType(name))

    def load(mod, rel):
        spec = importlib.util.spec_from_file_location(mod, REF / rel)
        m = importlib.util.module_from_spec(spec)
        sys.modules[mod] = m
        spec.loader.exec_module(m)
        return m

    fields = load("lambdapic.core.fields", "core/fields.py")
    cpml = load("lambdapic.core.boundary.cpml", "core/boundary/cpml.py")
    return fields, cpml


def load_ref_laser_kernel():
    """only the boundary kernel _update_laser_bfields_2d of callback/laser.py (the module itself
    imports the whole package); its source segment is executed here, never stored"""
    import ast
    src = (REF / "callback/laser.py").read_text()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "_update_laser_bfields_2d")
    fn.decorator_list = []
    mod = ast.Module(body=[fn], type_ignores=[])
    from scipy.constants import c, epsilon_0
    ns = {"np": np, "NDArray": object, "prange": range, "c": c, "epsilon_0": epsilon_0}
    exec(compile(ast.fix_missing_locations(mod), "laser_kernel", "exec"), ns)
    return ns["_update_laser_bfields_2d"]


def g9_cpml(mx):
    """vacuum EM pulse in a 48x48 box, 3x3 patches of 16x16, PML (thickness 6) on all four sides:
    the reference's PML classes + kappa-scaled updates on the edge patches, plain updates on the
    centre patch, guard sync by the reference's C extension; 80 full Maxwell stages."""
    RF, RC = load_ref_cpml()
    sf = oracle.ref_module("patch", "sync_fields2d")
    npx = npy = 3
    nxp = nyp = 16
    nx, ny, ng, th = npx * nxp, npy * nyp, 3, 6
    dx, dy = 4e-8, 5e-8
    dt = 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2))
    bc = {"xmin": "pml", "xmax": "pml", "ymin": "pml", "ymax": "pml"}
    P = make_patches_2d(nx, ny, dx, dy, npx, npy, ng, boundary_conditions=bc)
    for p in P:   # reference field bags (the PML classes isinstance-check them)
        p.set_fields(RF.Fields2D(nxp, nyp, dx, dy, p.x0, p.y0, ng))
        if p.ipatch_x == 0: p.pml_boundary.append(RC.PMLXmin(p.fields, thickness=th))
        if p.ipatch_x == npx - 1: p.pml_boundary.append(RC.PMLXmax(p.fields, thickness=th))
        if p.ipatch_y == 0: p.pml_boundary.append(RC.PMLYmin(p.fields, thickness=th))
        if p.ipatch_y == npy - 1: p.pml_boundary.append(RC.PMLYmax(p.fields, thickness=th))
    # initial condition: Gaussian blobs in ez, bz (both polarisations) + a weak random current
    rng = np.random.default_rng(SEED + 9)
    jglob = {a: rng.normal(size=(nx, ny)) * 1e13 for a in ("jx", "jy", "jz")}
    for p in P:
        f = p.fields
        X, Y = f.xaxis[:nxp, :], f.yaxis[:, :nyp]
        r2 = ((X - 0.45 * nx * dx) ** 2 + (Y - 0.55 * ny * dy) ** 2) / (4 * dx) ** 2
        f.ez[:nxp, :nyp] = 1e12 * np.exp(-r2)
        f.bz[:nxp, :nyp] = 2e3 * np.exp(-r2 * 1.3)
        i0, j0 = p.ipatch_x * nxp, p.ipatch_y * nyp
        for a in jglob:
            getattr(f, a)[:nxp, :nyp] = jglob[a][i0:i0 + nxp, j0:j0 + nyp]
    fl = [p.fields for p in P]
    out = dict(nx=nx, ny=ny, ng=ng, dx=dx, dy=dy, dt=dt, thickness=th, npx=npx, npy=npy, nsteps=80)
    E, B = ["ex", "ey", "ez"], ["bx", "by", "bz"]
    sf.sync_guard_fields_2d(fl, list(P), E + B, len(P), nxp, nyp, ng)

    def glob(a):
        G = np.zeros((nx, ny))
        for p in P:
            G[p.ipatch_x * nxp:(p.ipatch_x + 1) * nxp, p.ipatch_y * nyp:(p.ipatch_y + 1) * nyp] = \
                getattr(p.fields, a)[:nxp, :nyp]
        return G

    for a in E + B + ["jx", "jy", "jz"]:
        out["in_" + a] = glob(a)

    def kappas(p):
        ke_x = ke_y = kb_x = kb_y = None
        first = p.pml_boundary[0]
        ke_x, ke_y, kb_x, kb_y = first.kappa_ex, first.kappa_ey, first.kappa_bx, first.kappa_by
        for pml in p.pml_boundary:        # core/maxwell/solver/solver.py:88-105
            if isinstance(pml, RC.PMLX): ke_x, kb_x = pml.kappa_ex, pml.kappa_bx
            elif isinstance(pml, RC.PMLY): ke_y, kb_y = pml.kappa_ey, pml.kappa_by
        return ke_x, ke_y, kb_x, kb_y

    def upd_e(h):
        for p in P:
            f = p.fields
            if p.pml_boundary:
                ke_x, ke_y, _, _ = kappas(p)
                RC.update_efield_cpml_2d(f.ex, f.ey, f.ez, f.bx, f.by, f.bz, f.jx, f.jy, f.jz, ke_x, ke_y,
                                         dx, dy, h, nxp, nyp, ng)
                for pml in p.pml_boundary:
                    pml.advance_e_currents(h)
            else:
                mx.update_efield_2d(f.ex, f.ey, f.ez, f.bx, f.by, f.bz, f.jx, f.jy, f.jz, dx, dy, h, nxp, nyp, ng)

    def upd_b(h):
        for p in P:
            f = p.fields
            if p.pml_boundary:
                _, _, kb_x, kb_y = kappas(p)
                RC.update_bfield_cpml_2d(f.ex, f.ey, f.ez, f.bx, f.by, f.bz, kb_x, kb_y, dx, dy, h, nxp, nyp, ng)
                for pml in p.pml_boundary:
                    pml.advance_b_currents(h)
            else:
                mx.update_bfield_2d(f.ex, f.ey, f.ez, f.bx, f.by, f.bz, dx, dy, h, nxp, nyp, ng)

    energy = []
    for it in range(80):
        upd_e(0.5 * dt); sf.sync_guard_fields_2d(fl, list(P), E, len(P), nxp, nyp, ng)
        upd_b(0.5 * dt); sf.sync_guard_fields_2d(fl, list(P), B, len(P), nxp, nyp, ng)
        if it == 2:   # the currents act for three steps, then vacuum
            for f in fl:
                f.jx[...] = 0; f.jy[...] = 0; f.jz[...] = 0
        energy.append(sum(float(np.sum(0.5 * driver.EPSILON_0 * (f.ex[:nxp, :nyp] ** 2 + f.ey[:nxp, :nyp] ** 2 + f.ez[:nxp, :nyp] ** 2)
                                        + 0.5 / driver.MU_0 * (f.bx[:nxp, :nyp] ** 2 + f.by[:nxp, :nyp] ** 2 + f.bz[:nxp, :nyp] ** 2))) for f in fl) * dx * dy)
        if it in (9, 79):
            for a in E + B:
                out[f"step{it + 1}_{a}"] = glob(a)
    out["trace_energy"] = np.array(energy)
    np.savez_compressed(OUT / "g9_cpml_2d.npz", **out)


def g10_laser(rng):
    kern = load_ref_laser_kernel()
    nx, ny, ng, dx, dy = 24, 20, 3, 4e-8, 5e-8
    dt = 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2))
    f = Fields2D(nx, ny, dx, dy, 0.0, 0.0, ng)
    for a in f.attrs[:9]:
        getattr(f, a)[...] = rng.normal(size=f.shape) * (1e12 if a[0] == "e" else (1e4 if a[0] == "b" else 1e15))
    ey_s, ez_s = rng.normal(size=ny + 2 * ng) * 1e12, rng.normal(size=ny + 2 * ng) * 1e12
    out = dict(nx=nx, ny=ny, ng=ng, dx=dx, dy=dy, dt=dt, laserpos=8, iy_start=6, iy_end=ny - 6,
               ey_source=ey_s, ez_source=ez_s)
    out.update(snap(f, f.attrs[:9], "in_"))
    kern(8, f.ex, f.ey, f.ez, f.bx, f.by, f.bz, f.jx, f.jy, f.jz, dx, dy, dt, 6, ny - 6, ey_s, ez_s)
    out.update(snap(f, ["bx", "by", "bz"], "out_"))
    np.savez_compressed(OUT / "g10_laser_2d.npz", **out)


def load_ref_laser_classes():
    """the laser profile classes of callback/laser.py (class statements only; the module itself
    imports the whole package and mpi4py); executed here with duck-typed sim / patch objects"""
    import ast
    from numpy.typing import NDArray
    from scipy.constants import c, e, epsilon_0, m_e, pi
    from scipy.special import factorial, genlaguerre
    tree = ast.parse((REF / "callback/laser.py").read_text())
    body = [n for n in tree.body if isinstance(n, ast.ClassDef)]
    ns = {"np": np, "NDArray": NDArray, "c": c, "e": e, "m_e": m_e, "pi": pi, "epsilon_0": epsilon_0,
          "factorial": factorial, "genlaguerre": genlaguerre, "Optional": object}
    for name in ("Simulation", "Simulation2D", "Simulation3D", "Patch", "Patch2D", "Patch3D", "Fields"):
        ns[name] = object
    exec(compile(ast.fix_missing_locations(ast.Module(body=body, type_ignores=[])), "laser_classes", "exec"), ns)
    return ns


def g11_laser_profiles():
    """source rows (ey_source, ez_source) of SimpleLaser2D, GaussianLaser2D (plain, LG, defocused,
    elliptical) and of a sum of two lasers at several times"""
    ns = load_ref_laser_classes()
    ny, ng, dx, dy, t = 96, 3, 5e-8, 6e-8, 6
    Ly = ny * dy
    yaxis = ((np.arange(ny + 2 * ng) - ng) * dy)            # fields.yaxis incl. guards, origin 0
    yaxis = np.roll(yaxis, -ng)[None, :]                     # wrapped layout: interior first
    patch = types.SimpleNamespace(fields=types.SimpleNamespace(yaxis=yaxis))
    cases = {
        "simple": ("SimpleLaser2D", dict(a0=2.0, w0=1.2e-6, ctau=1.5e-6, pol_angle=0.3, ellipticity=0.4, cep=0.7)),
        "simple_oblique": ("SimpleLaser2D", dict(a0=1.0, w0=1.0e-6, ctau=2.0e-6, angle_y=0.35, y0=2.0e-6)),
        "gauss": ("GaussianLaser2D", dict(a0=1.5, l0=0.8e-6, w0=1.5e-6, ctau=1.0e-6, focus_position=8e-6)),
        "gauss_ellip": ("GaussianLaser2D", dict(a0=1.0, l0=1.0e-6, w0=2.0e-6, ctau=1.2e-6, pol_angle=1.1,
                                                ellipticity=-1.0, cep=0.5, y0=3.1e-6)),
        "gauss_lg": ("GaussianLaser2D", dict(a0=1.0, l0=0.8e-6, w0=1.0e-6, ctau=1.0e-6, focus_position=-3e-6,
                                             l=1, p=2)),
    }
    out = dict(ny=ny, ng=ng, dx=dx, dy=dy, thickness=t, Ly=Ly, yaxis=yaxis[0])
    times = np.array([0.2e-15, 3.3e-15, 7.7e-15, 12.1e-15])
    out["times"] = times
    import json
    out["cases"] = np.array(json.dumps(cases))
    lasers = {k: ns[cls](**kw) for k, (cls, kw) in cases.items()}
    lasers["sum"] = lasers["simple"] + lasers["gauss"]
    for name, las in lasers.items():
        for it, tm in enumerate(times):
            sim = types.SimpleNamespace(time=float(tm), Ly=Ly, dy=dy, dx=dx, cpml_thickness=t)
            ey, ez = las._calculate_bound_fields(sim, patch)
            if ey is None:                      # pulse over: stored as empty rows
                ey = ez = np.zeros(0)
            out[f"{name}_t{it}_ey"], out[f"{name}_t{it}_ez"] = ey, ez
    # 3-D classes: r, phi from the y-z plane of the boundary (callback/laser.py:194-238)
    nz3, dz3 = 20, 7e-8
    Lz = nz3 * dz3
    ya3 = np.roll((np.arange(24 + 2 * ng) - ng) * dy, -ng)[None, :, None]
    za3 = np.roll((np.arange(nz3 + 2 * ng) - ng) * dz3, -ng)[None, None, :]
    patch3 = types.SimpleNamespace(fields=types.SimpleNamespace(yaxis=ya3, zaxis=za3))
    cases3 = {
        "simple3": ("SimpleLaser3D", dict(a0=1.2, w0=0.6e-6, ctau=1.5e-6, pol_angle=0.4, ellipticity=0.3, angle_y=0.2)),
        "gauss3_lg": ("GaussianLaser3D", dict(a0=0.9, l0=0.8e-6, w0=0.5e-6, ctau=1.0e-6, focus_position=2e-6,
                                               l=-1, p=1, z0=0.6e-6, ellipticity=0.5)),
    }
    out.update(ny3=24, nz3=nz3, dz3=dz3, Ly3=24 * dy, Lz3=Lz, cases3=np.array(json.dumps(cases3)))
    for name, (cls, kw) in cases3.items():
        las = ns[cls](**kw)
        for it, tm in enumerate(times):
            sim = types.SimpleNamespace(time=float(tm), Ly=24 * dy, Lz=Lz, dy=dy, dz=dz3, dx=dx, cpml_thickness=t)
            ey, ez = las._calculate_bound_fields(sim, patch3)
            if ey is None:
                ey = ez = np.zeros(0)
            out[f"{name}_t{it}_ey"], out[f"{name}_t{it}_ez"] = ey, ez
    np.savez_compressed(OUT / "g11_laser_profiles.npz", **out)


def g12_cpml_laser_3d():
    """3-D: the reference's six PML objects on ONE field bag (function level: kappa-scaled update with the
    min / max kappa arrays merged, then every object's psi recursion), three half steps of E and B on
    random fields with static random guards; and the 3-D laser boundary kernel.  The slab = union of
    edge patches argument is the 2-D one (g9)."""
    import ast
    RF, RC = load_ref_cpml()
    rng = np.random.default_rng(SEED + 12)
    nx, ny, nz, ng, th = 10, 9, 11, 3, 3
    dx, dy, dz = 4e-8, 5e-8, 6e-8
    dt = 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2 + dz ** -2))
    f = RF.Fields3D(nx, ny, nz, dx, dy, dz, 0.0, 0.0, 0.0, ng)
    scale = dict(ex=1e12, ey=1e12, ez=1e12, bx=3e3, by=3e3, bz=3e3, jx=1e15, jy=1e15, jz=1e15)
    for a, sc in scale.items():
        getattr(f, a)[...] = rng.normal(size=getattr(f, a).shape) * sc
    out = dict(nx=nx, ny=ny, nz=nz, ng=ng, dx=dx, dy=dy, dz=dz, dt=dt, thickness=th)
    for a in scale:
        out["in_" + a] = getattr(f, a).copy()
    pmls = [cls(f, thickness=th) for cls in (RC.PMLXmin, RC.PMLXmax, RC.PMLYmin, RC.PMLYmax, RC.PMLZmin, RC.PMLZmax)]

    def merged(name):
        lo, hi = {"x": pmls[0:2], "y": pmls[2:4], "z": pmls[4:6]}[name[-1]]
        a, b = getattr(lo, name), getattr(hi, name)
        return np.where(a != 1.0, a, b)

    for it in range(3):
        RC.update_efield_cpml_3d(f.ex, f.ey, f.ez, f.bx, f.by, f.bz, f.jx, f.jy, f.jz, merged("kappa_ex"),
                                 merged("kappa_ey"), merged("kappa_ez"), dx, dy, dz, 0.5 * dt, nx, ny, nz, ng)
        for p in pmls:
            p.advance_e_currents(0.5 * dt)
        RC.update_bfield_cpml_3d(f.ex, f.ey, f.ez, f.bx, f.by, f.bz, merged("kappa_bx"), merged("kappa_by"),
                                 merged("kappa_bz"), dx, dy, dz, 0.5 * dt, nx, ny, nz, ng)
        for p in pmls:
            p.advance_b_currents(0.5 * dt)
        if it in (0, 2):
            for a in ("ex", "ey", "ez", "bx", "by", "bz"):
                out[f"it{it}_{a}"] = getattr(f, a).copy()
    # laser kernel (the module defines _update_laser_bfields_3d twice; the later definition is the live one)
    tree = ast.parse((REF / "callback/laser.py").read_text())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "_update_laser_bfields_3d"][-1]
    fn.decorator_list = []
    from scipy.constants import c, epsilon_0
    ns = {"np": np, "NDArray": object, "prange": range, "c": c, "epsilon_0": epsilon_0}
    exec(compile(ast.fix_missing_locations(ast.Module(body=[fn], type_ignores=[])), "laser_kernel_3d", "exec"), ns)
    g = f                                      # input of the laser kernel = the state after it2 (+ in_j*)
    eys, ezs = rng.normal(size=g.ex.shape[1:]) * 1e12, rng.normal(size=g.ex.shape[1:]) * 1e12
    out.update(laserpos=th + 2, iy_start=th, iy_end=ny - th, iz_start=th, iz_end=nz - th, ey_source=eys, ez_source=ezs)
    ns["_update_laser_bfields_3d"](th + 2, g.ex, g.ey, g.ez, g.bx, g.by, g.bz, g.jx, g.jy, g.jz, dx, dy, dz, dt,
                                   th, ny - th, th, nz - th, eys, ezs)
    for a in ("bx", "by", "bz"):
        out["lout_" + a] = getattr(g, a).copy()
    np.savez_compressed(OUT / "g12_cpml_laser_3d.npz", **out)


def g13_sync_3d():
    """3-D guard copy and current fold of ONE patch that is its own periodic neighbour through all 26
    boundaries (`core/patch/sync_fields3d.c:84-348,350-612`): what a one-slab-per-GPU 3-D run does locally.
    Pins the oracle's periodic_guard_fill / periodic_current_fold and the device's lpa_guard_wrap /
    lpa_current_fold in 3-D."""
    import types
    from lambdapic_amd.fields import Fields3D
    sf = oracle.ref_module("patch", "sync_fields3d")
    rng = np.random.default_rng(SEED + 13)
    nx, ny, nz, ng = 8, 6, 10, 3
    f = Fields3D(nx, ny, nz, 1e-7, 1e-7, 1e-7, 0.0, 0.0, 0.0, ng)
    out = dict(nx=nx, ny=ny, nz=nz, ng=ng)
    for a in f.attrs:
        getattr(f, a)[...] = rng.normal(size=f.shape)
    out.update(snap(f, f.attrs, "in_"))
    patch = types.SimpleNamespace(neighbor_ipatch=np.zeros(26, dtype=np.intp))
    sf.sync_guard_fields_3d([f], [patch], ["ex", "ey", "ez", "bx", "by", "bz"], 1, nx, ny, nz, ng)
    sf.sync_currents_3d([f], [patch], 1, nx, ny, nz, ng)
    out.update(snap(f, f.attrs, "out_"))
    np.savez_compressed(OUT / "g13_sync_3d.npz", **out)


def g14_sync_particles_3d():
    """3-D particle ownership on ONE patch that is its own periodic neighbour (`core/patch/sync_particles_3d.c`
    get_npart_to_extend_3d :362-488, fill_particles_from_boundary_3d :490-...; orchestration
    `core/patch/patch.py:739-763`): particles pushed out through faces, edges and corners come back shifted by
    the box length; the leaver's old slot dies.  Pins the periodic fold the 3-D step applies."""
    import types
    from lambdapic_amd.particles import ParticlesBase
    mod = oracle.ref_module("patch", "sync_particles_3d")
    rng = np.random.default_rng(SEED + 14)
    nx, ny, nz = 8, 6, 10
    dx, dy, dz = 1.0e-7, 1.5e-7, 0.8e-7
    n = 900
    q = ParticlesBase(0, 0)
    q.initialize(n)
    shift = rng.uniform(-0.95, 0.95, (3, n))
    q.x[:] = (rng.uniform(-0.5, nx - 0.5, n) + shift[0]) * dx
    q.y[:] = (rng.uniform(-0.5, ny - 0.5, n) + shift[1]) * dy
    q.z[:] = (rng.uniform(-0.5, nz - 0.5, n) + shift[2]) * dz
    q.ux[:] = rng.normal(size=n)
    q.w[:] = rng.uniform(1, 2, n)
    q.is_dead[::11] = True
    out = dict(nx=nx, ny=ny, nz=nz, dx=dx, dy=dy, dz=dz)
    out.update(snap(q, ["x", "y", "z", "ux", "w", "_id", "is_dead"], "pin_"))
    patch = types.SimpleNamespace(neighbor_ipatch=np.zeros(26, dtype=np.intp), xmin=0.0, xmax=(nx - 1) * dx,
                                  ymin=0.0, ymax=(ny - 1) * dy, zmin=0.0, zmax=(nz - 1) * dz)
    ext, inc, outg, alive = mod.get_npart_to_extend_3d([q], [patch], 1, dx, dy, dz)
    if ext[0] > 0:
        q.extend(int(ext[0]))
    mod.fill_particles_from_boundary_3d([q], [patch], inc, outg, 1, dx, dy, dz,
                                        -dx / 2, nx * dx - dx / 2, -dy / 2, ny * dy - dy / 2,
                                        -dz / 2, nz * dz - dz / 2, q.attrs)
    out["npart_alive"] = np.asarray(alive)
    out.update(snap(q, ["x", "y", "z", "ux", "w", "_id", "is_dead"], "pout_"))
    np.savez_compressed(OUT / "g14_sync_particles_3d.npz", **out)


def g16_sync_fields_3d_patches():
    """3-D guard copy and current fold between DIFFERENT patches: 2 x 2 x 2 periodic patches of 6 x 6 x 7 cells, every
    one of the 26 neighbour classes is another patch (g13's single self-periodic patch cannot tell a swapped neighbour
    from the right one).  `core/patch/sync_fields3d.c:84-348,350-612` driven like `Patches.sync_guard_fields /
    sync_currents` (`core/patch/patch.py:670-703`).  Values are small multiples of 1/64 so that the fixture stores them
    as int8 / int16 (copies and sums of up to 8 of them are exact in FP64 whatever the order)."""
    from lambdapic_amd.patch import make_patches_3d
    sf = oracle.ref_module("patch", "sync_fields3d")
    rng = np.random.default_rng(SEED + 16)
    npp, npatch, ng = (6, 6, 7), (2, 2, 2), 3
    P = make_patches_3d(tuple(a * b for a, b in zip(npp, npatch)), (1e-7, 1.5e-7, 0.8e-7), npatch, ng)
    guard_attrs, cur_attrs = ["ex", "by", "bz"], ["jx", "jy", "jz", "rho"]
    out = dict(npp=np.array(npp), npatch=np.array(npatch), ng=ng, scale=64.0,
               neighbor_ipatch=np.stack([p.neighbor_ipatch for p in P]))
    for k, p in enumerate(P):
        for a in guard_attrs + cur_attrs:
            v = rng.integers(-100, 101, size=p.fields.shape).astype(np.int8)
            getattr(p.fields, a)[...] = v / 64.0
            out[f"in{k}_{a}"] = v
    fl = [p.fields for p in P]
    sf.sync_guard_fields_3d(fl, list(P), guard_attrs, 8, *npp, ng)
    sf.sync_currents_3d(fl, list(P), 8, *npp, ng)
    for k, p in enumerate(P):
        for a in guard_attrs + cur_attrs:
            v = getattr(p.fields, a) * 64.0
            r = np.rint(v)
            assert np.array_equal(v, r) and np.abs(r).max() < 32000
            out[f"out{k}_{a}"] = r.astype(np.int16)
    np.savez_compressed(OUT / "g16_sync_fields_3d_patches.npz", **out)


def g17_sync_particles_3d_patches():
    """3-D particle ownership between DIFFERENT patches: the same 2 x 2 x 2 periodic patches, every particle displaced by
    up to 0.95 cell per axis so that leavers go through faces, edges and corners into seven different neighbours (and
    around the box: +- L); dead slots every 11th.  `get_npart_to_extend_3d` + `fill_particles_from_boundary_3d`
    (`core/patch/sync_particles_3d.c:365-482,484-700`) driven like `Patches.sync_particles` (`core/patch/patch.py:739-763`)."""
    from lambdapic_amd.patch import make_patches_3d
    mod = oracle.ref_module("patch", "sync_particles_3d")
    rng = np.random.default_rng(SEED + 17)
    npp, npatch = (6, 6, 7), (2, 2, 2)
    d = (1.0e-7, 1.5e-7, 0.8e-7)
    P = make_patches_3d(tuple(a * b for a, b in zip(npp, npatch)), d, npatch, 3)
    out = dict(npp=np.array(npp), npatch=np.array(npatch), d=np.array(d))
    names = ["x", "y", "z", "ux", "w", "_id", "is_dead"]
    for k, p in enumerate(P):
        q = p.particles[0]
        n = 240 + 7 * k
        q.initialize(n)
        for ax, n_ax, dd in zip("xyz", npp, d):
            getattr(q, ax)[:] = getattr(p, ax + "0") + (rng.uniform(-0.5, n_ax - 0.5, n) + rng.uniform(-0.95, 0.95, n)) * dd
        q.ux[:] = rng.normal(size=n)
        q.w[:] = rng.uniform(1, 2, n)
        q.is_dead[k % 5::11] = True
        # one sure leaver per neighbour class (corner leavers are rare among the random ones), scattered over the bag
        from lambdapic_amd.patch import OFFSET_3D, Boundary3D
        slots = rng.permutation(n)[:26]
        for b, slot in zip(Boundary3D, slots):
            for ax, off, n_ax, dd in zip("xyz", OFFSET_3D[b], npp, d):
                lo = getattr(p, ax + "0")
                getattr(q, ax)[slot] = lo + {-1: -0.8, 0: 0.5 * (n_ax - 1), 1: n_ax - 0.2}[off] * dd
            q.is_dead[slot] = False
        out.update(snap(q, names, f"pin{k}_"))
    parts = [p.particles[0] for p in P]
    ext, inc, outg, alive = mod.get_npart_to_extend_3d(parts, list(P), 8, *d)
    out.update(npart_to_extend=np.asarray(ext).copy(), npart_incoming=np.asarray(inc).copy(),
               npart_outgoing=np.asarray(outg).copy(), npart_alive=np.asarray(alive).copy())
    for q, n in zip(parts, ext):
        if n > 0:
            q.extend(int(n))
    mod.fill_particles_from_boundary_3d(parts, list(P), inc, outg, 8, *d, P.xmin_global, P.xmax_global, P.ymin_global,
                                        P.ymax_global, P.zmin_global, P.zmax_global, parts[0].attrs)
    for k, q in enumerate(parts):
        out.update(snap(q, names, f"pout{k}_"))
    assert int(np.asarray(outg).reshape(8, 26).sum(0).min()) > 0        # every neighbour class is exercised
    np.savez_compressed(OUT / "g17_sync_particles_3d_patches.npz", **out)


def g15_sort_variants():
    """the branches of the reference's sort that g6 does not take, recorded from its compiled extensions:
    (a) sort_particles_patches_3d (core/sort/cpu3d.c) on a 6 x 5 x 4 bucket grid with dead slots and out-of-range
    particles (-> last bucket); (b) sort_particles_patches_2d with 2-D buckets and reverse_x = 1 (mirrored x order,
    clamped; core/sort/cpu2d.c:25-32)"""
    rng = np.random.default_rng(SEED + 15)
    out = {}
    # ---- (a) 3-D
    n, nb, d, o = 4000, (6, 5, 4), (1.0e-7, 1.5e-7, 2.0e-7), (0.5e-7, -1.0e-7, 0.0)
    pos = [o[a] + rng.uniform(-0.4, nb[a] + 0.4, n) * d[a] for a in range(3)]     # some out of range
    dead = rng.random(n) < 0.12
    dead[0] = True
    attrs = {"x": pos[0], "y": pos[1], "z": pos[2], "w": rng.uniform(1, 2, n), "tag": np.arange(n, dtype=np.float64)}
    out.update({"a_in_" + k: v.copy() for k, v in attrs.items()})
    out["a_in_is_dead"] = dead.copy()
    out.update(a_nb=np.array(nb), a_d=np.array(d), a_o=np.array(o))
    mod = oracle.ref_module("sort", "cpu3d")
    z = lambda: [np.zeros(nb, dtype=np.int64)]
    cnt, bmin, bmax = z(), z(), z()
    sc = lambda dt: [np.zeros(n, dtype=dt)]
    arrs = {k: v.copy() for k, v in attrs.items()}
    dd = dead.copy()
    out["a_nbuf"] = mod.sort_particles_patches_3d(
        [arrs["x"]], [arrs["y"]], [arrs["z"]], [dd], [arrs[k] for k in ("x", "y", "z", "w", "tag")],
        [o[0]], [o[1]], [o[2]], *nb, *d, 1, cnt, bmin, bmax, z(), z(), sc(np.int64), sc(np.int64), sc(np.int64),
        sc(np.float64), 0)
    out.update(a_bucket_count=cnt[0].copy(), a_bucket_bound_min=bmin[0].copy(), a_bucket_bound_max=bmax[0].copy())
    out.update({"a_out_" + k: v.copy() for k, v in arrs.items()})
    out["a_out_is_dead"] = dd.copy()
    # ---- (b) 2-D, 2-D buckets, reverse_x
    n, nb2, d2, o2 = 3000, (7, 3), (1.0e-7, 2.5e-7), (-2.0e-7, 1.0e-7)
    pos = [o2[a] + rng.uniform(-0.4, nb2[a] + 0.4, n) * d2[a] for a in range(2)]
    dead = rng.random(n) < 0.12
    attrs = {"x": pos[0], "y": pos[1], "w": rng.uniform(1, 2, n), "tag": np.arange(n, dtype=np.float64)}
    out.update({"b_in_" + k: v.copy() for k, v in attrs.items()})
    out["b_in_is_dead"] = dead.copy()
    out.update(b_nb=np.array(nb2), b_d=np.array(d2), b_o=np.array(o2))
    mod2 = oracle.ref_module("sort", "cpu2d")
    z2 = lambda: [np.zeros(nb2, dtype=np.int64)]
    cnt, bmin, bmax = z2(), z2(), z2()
    arrs = {k: v.copy() for k, v in attrs.items()}
    dd = dead.copy()
    out["b_nbuf"] = mod2.sort_particles_patches_2d(
        [arrs["x"]], [arrs["y"]], [dd], [arrs[k] for k in ("x", "y", "w", "tag")], [o2[0]], [o2[1]], *nb2, *d2, 1,
        cnt, bmin, bmax, z2(), z2(), sc(np.int64), sc(np.int64), sc(np.int64), sc(np.float64), 1)
    out.update(b_bucket_count=cnt[0].copy(), b_bucket_bound_min=bmin[0].copy(), b_bucket_bound_max=bmax[0].copy())
    out.update({"b_out_" + k: v.copy() for k, v in arrs.items()})
    out["b_out_is_dead"] = dd.copy()
    np.savez_compressed(OUT / "g15_sort_variants.npz", **out)


def main():
    if "--only-g15" in sys.argv:
        g15_sort_variants()
        return
    if "--only-g16" in sys.argv:
        g16_sync_fields_3d_patches()
        g17_sync_particles_3d_patches()
        sys.exit(0)
    if "--only-g14" in sys.argv:
        g14_sync_particles_3d()
        return
    if "--only-g13" in sys.argv:
        g13_sync_3d()
        return
    if "--only-g11" in sys.argv:
        g11_laser_profiles()
        return
    if "--only-g12" in sys.argv:
        g12_cpml_laser_3d()
        return
    assert oracle.ref_available(), "run `make -C oracle ref` first"
    mx = load_ref_maxwell()
    rng = np.random.default_rng(SEED)
    g1_fused_2d(rng)
    g2_fused_3d(rng)
    g3_deposit(rng)
    g4_interp(rng)
    g5_fdtd(rng, mx)
    g6_sort(rng)
    g7_sync(rng)
    g8_trace(mx)
    g9_cpml(mx)
    g10_laser(np.random.default_rng(SEED + 10))
    g11_laser_profiles()
    g12_cpml_laser_3d()
    g13_sync_3d()
    g14_sync_particles_3d()
    g16_sync_fields_3d_patches()
    g17_sync_particles_3d_patches()
    g15_sort_variants()
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
