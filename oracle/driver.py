"""oracle/driver.py -- TEST INFRASTRUCTURE ONLY.

CPU step loop restating the stage order of the reference's ``Simulation.run`` loop body
(simulation/simulation.py:937-1130) for the no-callback, Boris-only, periodic case (the unified
pusher path, :896-911,988-990), on the host patch mirrors, with a pluggable kernel set:

* ``oracle_kernels()`` -- this repo's CPU restatement (picoracle.c + sync.py);
* ``tests/golden/gen_golden.py`` plugs in the reference's own compiled kernels to record traces.

Also the diagnostics the parity tests compare (tests/test_numerical_heating.py:19-50 of the
reference): field energy, kinetic energy, total charge.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable

import numpy as np

from . import C_LIGHT, EPSILON_0, MU_0, M_E, E_CHARGE  # noqa: F401


@dataclass
class KernelSet:
    unified: Callable        # (particles_list, fields_list, npatches, dt, q, m)
    update_e: Callable       # (fields, dt) one patch
    update_b: Callable
    sync_guard: Callable     # (fields_list, patches_list, attrs, npatches, nx, ny, ng)
    sync_currents: Callable  # (fields_list, patches_list, npatches, nx, ny, ng)
    sync_particles: Callable  # (patches, ispec, dx, dy)
    sort: Callable           # (patches, ispec) -> None
    reset: Callable          # (fields_list, npatches)


def sort_by_x_bucket(patches, ispec):
    """cell sort with the reference's default policy (simulation.py:696-698: one bucket per
    x-cell, single y bucket), bucket rule of sort/cpu2d.c:9-54.  The permutation inside a bucket
    is implementation defined in the reference; a stable argsort is used here."""
    import oracle

    for p in patches:
        q = p.particles[ispec]
        if q.npart == 0:
            continue
        Ly = p.ny * p.dy
        idx, _ = oracle.bucket_index_2d(q.x, q.y, q.is_dead, p.nx, 1, p.dx, Ly,
                                        p.x0 - p.dx / 2, p.y0 - p.dy / 2)
        order = np.argsort(idx, kind="stable")
        for a in q.attrs:
            arr = getattr(q, a)
            arr[:] = arr[order]
        q.is_dead[:] = q.is_dead[order]


def oracle_kernels() -> KernelSet:
    import oracle
    from oracle import sync

    return KernelSet(
        unified=oracle.unified_boris_pusher_cpu_2d,
        update_e=oracle.update_efield_2d,
        update_b=oracle.update_bfield_2d,
        sync_guard=sync.sync_guard_fields_2d,
        sync_currents=sync.sync_currents_2d,
        sync_particles=sync.sync_particles_2d,
        sort=sort_by_x_bucket,
        reset=oracle.reset_current,
    )


def step(patches, ks: KernelSet, dt, species, do_sort=True):
    """one time step; ``species`` = list of (q, m).  Order: simulation.py:946-1118."""
    fl = [p.fields for p in patches]
    pl = list(patches)
    n, nx, ny, ng = patches.npatches, patches.nx, patches.ny, patches.n_guard
    E, B = ["ex", "ey", "ez"], ["bx", "by", "bz"]
    for f in fl:
        ks.update_e(f, 0.5 * dt)
    ks.sync_guard(fl, pl, E, n, nx, ny, ng)
    for f in fl:
        ks.update_b(f, 0.5 * dt)
    ks.sync_guard(fl, pl, B, n, nx, ny, ng)
    if do_sort:
        for ispec in range(len(species)):
            ks.sort(patches, ispec)
    ks.reset(fl, n)
    for ispec, (q, m) in enumerate(species):
        ks.unified([p.particles[ispec] for p in patches], fl, n, dt, q, m)
    ks.sync_currents(fl, pl, n, nx, ny, ng)
    for ispec in range(len(species)):
        ks.sync_particles(patches, ispec, patches.dx, patches.dy)
    for f in fl:
        ks.update_b(f, 0.5 * dt)
    ks.sync_guard(fl, pl, B, n, nx, ny, ng)
    for f in fl:
        ks.update_e(f, 0.5 * dt)
    ks.sync_guard(fl, pl, E, n, nx, ny, ng)


def step_split(patches, dt, species, hook=None):
    """one step on the NON-unified path the reference takes when a callback sits in a pusher stage
    (simulation.py:993-1038): push_position(dt/2), interpolate, [hook = the '_interpolator'
    callback], Boris on the stored *_part, push_position(dt/2), standalone deposit."""
    import oracle
    from oracle import sync

    fl, pl = [p.fields for p in patches], list(patches)
    n, nx, ny, ng = patches.npatches, patches.nx, patches.ny, patches.n_guard
    E, B = ["ex", "ey", "ez"], ["bx", "by", "bz"]
    for f in fl:
        oracle.update_efield_2d(f, 0.5 * dt)
    sync.sync_guard_fields_2d(fl, pl, E, n, nx, ny, ng)
    for f in fl:
        oracle.update_bfield_2d(f, 0.5 * dt)
    sync.sync_guard_fields_2d(fl, pl, B, n, nx, ny, ng)
    oracle.reset_current(fl, n)
    for ispec, (q, m) in enumerate(species):
        parts = [p.particles[ispec] for p in patches]
        for pr in parts:
            oracle.push_position_2d(pr, 0.5 * dt)
        oracle.interpolation_patches_2d(parts, fl, n)
        if hook is not None:
            hook(patches, ispec)
        for pr in parts:
            oracle.boris_push(pr, q, m, dt)
            oracle.push_position_2d(pr, 0.5 * dt)
        oracle.current_deposition_cpu_2d(fl, parts, n, dt, q)
    sync.sync_currents_2d(fl, pl, n, nx, ny, ng)
    for ispec in range(len(species)):
        sync.sync_particles_2d(patches, ispec, patches.dx, patches.dy)
    for f in fl:
        oracle.update_bfield_2d(f, 0.5 * dt)
    sync.sync_guard_fields_2d(fl, pl, B, n, nx, ny, ng)
    for f in fl:
        oracle.update_efield_2d(f, 0.5 * dt)
    sync.sync_guard_fields_2d(fl, pl, E, n, nx, ny, ng)


def step_3d_periodic(f, parts, dt, species, box_lo, box_hi):
    """one 3-D step on ONE patch that is its own periodic neighbour (same stage order as `step`);
    ``parts[ispec]`` are particle bags, ``box_lo/hi`` the particle box (-d/2, L - d/2) per axis."""
    import oracle
    from oracle import sync

    E, B = ["ex", "ey", "ez"], ["bx", "by", "bz"]
    oracle.update_efield_3d(f, 0.5 * dt); sync.periodic_guard_fill(f, E)
    oracle.update_bfield_3d(f, 0.5 * dt); sync.periodic_guard_fill(f, B)
    oracle.reset_current([f], 1)
    for p, (q, m) in zip(parts, species):
        oracle.unified_boris_pusher_cpu_3d([p], [f], 1, dt, q, m)
    sync.periodic_current_fold(f)
    for p in parts:   # Patches.sync_particles with a self neighbour: periodic shift of leavers
        sync.periodic_fold_positions(p, box_lo, box_hi)
    oracle.update_bfield_3d(f, 0.5 * dt); sync.periodic_guard_fill(f, B)
    oracle.update_efield_3d(f, 0.5 * dt); sync.periodic_guard_fill(f, E)


def field_energy_3d(f) -> float:
    s = (slice(0, f.nx), slice(0, f.ny), slice(0, f.nz))
    e2 = f.ex[s] ** 2 + f.ey[s] ** 2 + f.ez[s] ** 2
    b2 = f.bx[s] ** 2 + f.by[s] ** 2 + f.bz[s] ** 2
    return float(np.sum(0.5 * EPSILON_0 * e2 + 0.5 / MU_0 * b2)) * f.dx * f.dy * f.dz


def field_energy(patches) -> float:
    """sum over patch interiors of (eps0 E^2 + B^2/mu0)/2 * dx*dy
    (reference tests/test_numerical_heating.py:19-37)"""
    tot = 0.0
    for p in patches:
        f = p.fields
        s = (slice(0, f.nx), slice(0, f.ny))
        e2 = f.ex[s] ** 2 + f.ey[s] ** 2 + f.ez[s] ** 2
        b2 = f.bx[s] ** 2 + f.by[s] ** 2 + f.bz[s] ** 2
        tot += float(np.sum(0.5 * EPSILON_0 * e2 + 0.5 / MU_0 * b2)) * f.dx * f.dy
    return tot


def kinetic_energy(patches, ispec, m) -> float:
    """sum over live particles of w (gamma-1) m c^2 (tests/test_numerical_heating.py:40-50)"""
    tot = 0.0
    for p in patches:
        q = p.particles[ispec]
        a = ~q.is_dead
        tot += float(np.sum(q.w[a] * (1.0 / q.inv_gamma[a] - 1.0))) * m * C_LIGHT ** 2
    return tot


def total_charge(patches) -> float:
    """sum over patch interiors of rho * dx*dy (valid after sync_currents)"""
    return sum(float(np.sum(p.fields.rho[: p.fields.nx, : p.fields.ny])) * p.fields.dx * p.fields.dy
               for p in patches)


def current_sums(patches):
    return tuple(sum(float(np.sum(getattr(p.fields, a)[: p.fields.nx, : p.fields.ny]))
                     for p in patches) for a in ("jx", "jy", "jz"))


def load_uniform_plasma(patches, ispec, ppc, density, u_th, rng, q_sign=-1):
    """synthetic loader of SURVEY 8(d): cell by cell, ``ppc`` macro-particles uniform in
    [x_i - dx/2, x_i + dx/2), w = n dx dy / ppc (core/patch/cpu.py:36-44), u ~ N(0, u_th)."""
    for p in patches:
        q = p.particles[ispec]
        n = p.nx * p.ny * ppc
        q.initialize(n)
        ix = np.repeat(np.arange(p.nx), p.ny * ppc)
        iy = np.tile(np.repeat(np.arange(p.ny), ppc), p.nx)
        q.x[:] = p.x0 + (ix + rng.uniform(-0.5, 0.5, n)) * p.dx
        q.y[:] = p.y0 + (iy + rng.uniform(-0.5, 0.5, n)) * p.dy
        q.w[:] = density * p.dx * p.dy / ppc
        q.ux[:] = rng.normal(0.0, u_th, n)
        q.uy[:] = rng.normal(0.0, u_th, n)
        q.uz[:] = rng.normal(0.0, u_th, n)
        q.inv_gamma[:] = 1.0 / np.sqrt(1.0 + q.ux ** 2 + q.uy ** 2 + q.uz ** 2)
