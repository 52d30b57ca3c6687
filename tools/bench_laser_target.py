#!/usr/bin/env python3
"""Config C3 on one GPU (BASELINE.json: 2-D laser-target, example/laser-target.py scale): 2048 x 1024 cells,
dx = lambda / 50, slab target x in (Lx/2, Lx/2 + 1 um) of e- and ions at 32 ppc each, CPML on all sides,
GaussianLaser2D(a0 = 10, w0 = 2 um, ctau = 5 um, x0 = 10 um), cell sort + moving window -- through the
Simulation facade.  Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lambdapic_amd import constants
from lambdapic_amd.laser import GaussianLaser2D
from lambdapic_amd.simulation import MovingWindow, Simulation, Species

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=2048); ap.add_argument("--ny", type=int, default=1024)
ap.add_argument("--ppc", type=int, default=32); ap.add_argument("--steps", type=int, default=300)
a = ap.parse_args()
C = constants.C_LIGHT
lam = 0.8e-6
dx = dy = lam / 50                                           # example/laser-target.py:30-31
nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C / lam) ** 2 / constants.E_CHARGE ** 2
sim = Simulation(a.nx, a.ny, dx, dy, npatch_x=a.nx // 64, npatch_y=a.ny // 64, random_seed=1, sort_interval=20)
Lx = a.nx * dx
dens = lambda x, y: np.where((x > Lx / 2) & (x < Lx / 2 + 1e-6), 10 * nc, 0.0)      # :37-43
sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=a.ppc, momentum_sigma=0.01))
sim.add_species(Species("p", charge=1, mass=1836.0, density=dens, ppc=a.ppc))
t0 = time.perf_counter()
sim.initialize()
t_init = time.perf_counter() - t0
laser = GaussianLaser2D(a0=10.0, l0=lam, w0=2e-6, ctau=2e-6, x0=4e-6)               # :45-53 (shortened)
win = MovingWindow(velocity=C, start_time=0.6 * Lx / C)
cbs = [laser, win]
sim.run(20, callbacks=cbs)
torch.cuda.synchronize()
n0 = sum(sim.engine.diagnostics()["nalive"])
t0 = time.perf_counter()
sim.run(a.steps, callbacks=cbs)
torch.cuda.synchronize()
el = time.perf_counter() - t0
d = sim.engine.diagnostics()
n1 = sum(d["nalive"])
print(json.dumps({"metric": "particle-updates/sec (C3 laser-target on one GPU, 2 species, moving window)",
                  "value": 0.5 * (n0 + n1) * a.steps / el, "ms_per_step": 1e3 * el / a.steps, "steps": a.steps,
                  "cells": [a.nx, a.ny], "alive_start": n0, "alive_end": n1,
                  "window_shifts": getattr(sim, "window_shifts", 0), "init_s": round(t_init, 2),
                  "field_energy_J_per_m": d["field_energy"], "kinetic_J_per_m": d["kinetic"]}))
