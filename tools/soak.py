#!/usr/bin/env python3
"""Long runs of the laser-target configs at full per-GPU size with the checks of tests/test_gpu_configs.py (exact live
count, charge of the padded rho array to 1e-12, per-node continuity to 1e-10) every so many steps -- far past the point
the tests stop at (C3: 720 steps, C5 slab: 24): the pulse has hit the target, the plasma is relativistically hot, the
sort-interval controller is at work, in 2-D the window keeps shifting.  No LpaError may be raised on the way.
Usage: python tools/soak.py [--c3 3000] [--c5 400]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_configs as T                                                    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--c3", type=int, default=3000)
ap.add_argument("--c5", type=int, default=400)
ap.add_argument("--c4", type=int, default=0, help="steps of the LWFA config (4096 x 512 cells, window at c with injection)")
ap.add_argument("--fraction", type=float, default=None, help="engine.overflow_sort_fraction (0: fixed sort interval)")
ap.add_argument("--min-interval", type=int, default=None)
ap.add_argument("--relaxed-anchor", action="store_true", help="experiment: engine.anchor_every_sort = False")
ap.add_argument("--no-lookahead", action="store_true", help="plain sorts (engine.sort_lookahead = False)")
ap.add_argument("--deep-tail", type=float, default=None, help="engine.deep_tail_fraction (2: never re-size the stripes)")
a = ap.parse_args()
LAM, NC, C = T.LAM, T.NC, T.C


def tune(eng):
    if a.fraction is not None:
        eng.overflow_sort_fraction = a.fraction
    if a.min_interval is not None:
        eng.min_sort_interval = a.min_interval
    if a.deep_tail is not None:
        eng.deep_tail_fraction = a.deep_tail
    if a.no_lookahead:
        eng.sort_lookahead = False
    if a.relaxed_anchor:
        eng.anchor_every_sort = False


def c3(nsteps):
    from lambdapic_amd.laser import GaussianLaser2D
    from lambdapic_amd.simulation import MovingWindow, Simulation, Species
    nx, ny, ppc = 2048, 1024, 32
    dx = dy = LAM / 50
    sim = Simulation(nx, ny, dx, dy, npatch_x=nx // 64, npatch_y=ny // 64, random_seed=1, sort_interval=20)
    Lx = nx * dx
    dens = lambda x, y: np.where((x > Lx / 2) & (x < Lx / 2 + 1e-6), 10 * NC, 0.0)
    sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=ppc, momentum_sigma=0.01))
    sim.add_species(Species("p", charge=1, mass=1836.0, density=dens, ppc=ppc))
    sim.initialize()
    eng = sim.engine
    tune(eng)
    n_init = T._live_qw_2d(eng)[2]
    cbs = [GaussianLaser2D(a0=10.0, l0=LAM, w0=2e-6, ctau=2e-6, x0=4e-6), MovingWindow(velocity=C, start_time=0.6 * Lx / C)]
    ledger = T._Ledger(sim)
    done = 0
    while done < nsteps:
        seg = min(500, nsteps - done)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        checks, absorbed = T._run_checked(sim, cbs, nsteps=seg, every=50, ledger=ledger)
        torch.cuda.synchronize()
        done += seg
        d = eng.diagnostics()
        umax = max(float(sp.cset.arr("ux")[: sp.n].nan_to_num().abs().max()) for sp in eng.species)
        print(f"C3 step {done:5d}: {1e3 * (time.perf_counter() - t0) / seg:.3f} ms/step (checks included), alive "
              f"{d['nalive']}, shifts {ledger.shifts}, dropped {ledger.dropped}, checks {checks}, max|ux| {umax:.1f}, "
              f"sort_interval_now {[getattr(sp, 'sort_interval_now', None) for sp in eng.species]}, rho steps "
              f"{dict(eng.rho_steps)}", flush=True)
    assert T._unique_ids(eng) and 0 < sum(eng.diagnostics()["nalive"]) <= n_init
    print("C3 soak ok", flush=True)


def c4(nsteps):
    from lambdapic_amd.laser import SimpleLaser2D
    from lambdapic_amd.simulation import MovingWindow, Simulation, Species
    nx, ny, ppc = 4096, 512, 16
    dx = dy = LAM / 20
    sim = Simulation(nx, ny, dx, dy, npatch_x=nx // 64, npatch_y=ny // 64, random_seed=1, sort_interval=20)
    Ly = ny * dy
    dens = lambda x, y: np.where((x > 1e-6) & (y > 1e-6) & (y < Ly - 1e-6), 0.01 * NC, 0.0)
    sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=ppc))
    sim.initialize()
    eng = sim.engine
    tune(eng)
    cbs = [SimpleLaser2D(a0=2.0, w0=5e-6, ctau=5e-6, l0=LAM), MovingWindow(velocity=C, start_time=0.03 * sim.Lx / C)]
    sim.run(5, callbacks=cbs)
    n_init = T._live_qw_2d(eng)[2]
    ledger = T._Ledger(sim)
    done = 0
    while done < nsteps:
        seg = min(500, nsteps - done)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        checks, absorbed = T._run_checked(sim, cbs, nsteps=seg, every=50, ledger=ledger)
        torch.cuda.synchronize()
        done += seg
        d = eng.diagnostics()
        umax = max(float(sp.cset.arr("ux")[: sp.n].nan_to_num().abs().max()) for sp in eng.species)
        print(f"C4 step {done:5d}: {1e3 * (time.perf_counter() - t0) / seg:.3f} ms/step (checks included), alive "
              f"{d['nalive']}, shifts {ledger.shifts}, dropped {ledger.dropped}, injected {ledger.injected}, checks {checks}, "
              f"max|ux| {umax:.1f}, sort_interval_now {[getattr(sp, 'sort_interval_now', None) for sp in eng.species]}, "
              f"stripe ranks {[getattr(sp, 'stripe_ranks', 0) for sp in eng.species]}", flush=True)
    n_end = sum(eng.diagnostics()["nalive"])
    assert T._unique_ids(eng) and 0 < n_end <= n_init - ledger.dropped + ledger.injected
    assert n_end >= n_init - ledger.dropped + ledger.injected - 0.02 * n_init
    print("C4 soak ok", flush=True)


def c5(nsteps):
    from lambdapic_amd._lib import LPA_ABSORB_X
    from lambdapic_amd.laser import GaussianLaser3D
    from lambdapic_amd.simulation3d import Simulation3D, Species
    nx, ny, nz, ppc = 64, 256, 256, 8
    dx, dy, dz = LAM / 20, LAM / 10, LAM / 10
    sim = Simulation3D(nx, ny, nz, dx, dy, dz, npatch_x=nx // 32, npatch_y=ny // 64, npatch_z=nz // 64, random_seed=1,
                       sort_interval=10)
    dens = lambda x, y, z: np.where(x > 1e-6, NC, 0.0)
    sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=ppc, momentum_sigma=0.01))
    sim.add_species(Species("p", charge=1, mass=1836.0, density=dens, ppc=ppc))
    sim.initialize()
    eng = sim.engine
    tune(eng)
    laser = GaussianLaser3D(a0=10.0, l0=LAM, w0=2e-6, ctau=3e-6, x0=6e-6)

    def live():
        tot, tot_abs, n = 0.0, 0.0, 0
        for sp in eng.species:
            d = sp["data"][:, : sp["n"]]
            ok = ~torch.isnan(d[0])
            w = d[7][ok].sum().item()
            tot, tot_abs, n = tot + sp["q"] * w, tot_abs + abs(sp["q"]) * w, n + int(ok.sum().item())
        return tot, tot_abs, n

    def near_bounds():
        cnt, dd = 0, (dx, dy, dz)
        for sp in eng.species:
            d = sp["data"][:, : sp["n"]]
            m = torch.zeros_like(d[0], dtype=torch.bool)
            for ax in range(3):
                if eng.absorb & (LPA_ABSORB_X << ax):
                    m |= (d[ax] < eng.alo[ax] + 1.05 * dd[ax]) | (d[ax] > eng.ahi[ax] - 1.05 * dd[ax])
            cnt += int((m & ~torch.isnan(d[0])).sum().item())
        return cnt

    n_init = live()[2]
    s = (slice(3, 3 + nx), slice(3, 3 + ny), slice(3, 3 + nz))
    inner = eng.cpml_thickness + 6
    t0, checks = time.perf_counter(), 0
    for it in range(nsteps):
        check = it % 40 == 39
        if check:
            qw, qw_abs, n0 = live()
            near = near_bounds()
            rho_prev = eng.view("rho")[s].clone()
        sim.run(1, callbacks=[laser])
        if not check:
            continue
        d = eng.diagnostics()
        charge = eng.view("rho").sum().item() * dx * dy * dz
        assert abs(charge - qw) <= 1e-12 * qw_abs, (it, charge, qw)
        n1 = live()[2]
        assert sum(d["nalive"]) == n1 and 0 <= n0 - n1 <= near, (it, n0, n1, near)
        rho, jx, jy, jz = (eng.view(c)[s] for c in ("rho", "jx", "jy", "jz"))
        res = ((rho - rho_prev) / sim.dt + (jx - torch.roll(jx, 1, 0)) / dx + (jy - torch.roll(jy, 1, 1)) / dy
               + (jz - torch.roll(jz, 1, 2)) / dz)[inner:-inner, inner:-inner, inner:-inner]
        assert res.abs().max().item() <= 1e-10 * rho.abs().max().item() / sim.dt, it
        checks += 1
        torch.cuda.synchronize()
        umax = max(float(sp["data"][3, : sp["n"]].nan_to_num().abs().max()) for sp in eng.species)
        print(f"C5 slab step {it + 1:4d}: {1e3 * (time.perf_counter() - t0) / 40:.2f} ms/step (check included), alive "
              f"{d['nalive']}, max|ux| {umax:.1f}, sort_interval_now {[sp.get('sort_interval_now') for sp in eng.species]}, "
              f"rho steps {dict(eng.rho_steps)}, stripe ranks {[sp.get('stripe_ranks', 0) for sp in eng.species]}", flush=True)
        t0 = time.perf_counter()
    assert T._unique_ids(eng) and 0 < live()[2] <= n_init and checks == nsteps // 40
    print("C5 soak ok", flush=True)


if a.c3:
    c3(a.c3)
    torch.cuda.empty_cache()
if a.c4:
    c4(a.c4)
    torch.cuda.empty_cache()
if a.c5:
    c5(a.c5)
