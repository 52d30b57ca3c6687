#!/usr/bin/env python3
"""Config C5's geometry on one GPU (BASELINE.json: 3-D laser-target, example/laser-target-3d.py): a
128 x 256 x 256 box (two of the eight x-slabs), dx = lambda/20, dy = dz = lambda/10, e- and p at n = nc for
x > 1 um, 4 + 4 ppc, CPML on all six faces, GaussianLaser3D -- through the Simulation3D stage loop.
Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lambdapic_amd import constants
from lambdapic_amd.laser import GaussianLaser3D
from lambdapic_amd.simulation3d import Simulation3D, Species

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=128); ap.add_argument("--ny", type=int, default=256)
ap.add_argument("--nz", type=int, default=256); ap.add_argument("--ppc", type=int, default=4)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--foil", type=float, default=0.0, help="target thickness in um (0: everything beyond x = 1 um)")
ap.add_argument("--deep-tail", type=float, default=None, help="engine.deep_tail_fraction (2: never re-size the stripes)")
a = ap.parse_args()
C = constants.C_LIGHT
lam = 0.8e-6
dx, dy, dz = lam / 20, lam / 10, lam / 10                    # example/laser-target-3d.py:26-31
nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C / lam) ** 2 / constants.E_CHARGE ** 2
sim = Simulation3D(a.nx, a.ny, a.nz, dx, dy, dz, npatch_x=a.nx // 32, npatch_y=a.ny // 64, npatch_z=a.nz // 64,
                   random_seed=1, sort_interval=10)
dens = lambda x, y, z: np.where((x > 1e-6) & ((a.foil == 0) | (x < 1e-6 + a.foil * 1e-6)), nc, 0.0)       # :37-42
sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=a.ppc, momentum_sigma=0.01))
sim.add_species(Species("p", charge=1, mass=1836.0, density=dens, ppc=a.ppc))
t0 = time.perf_counter()
sim.initialize()
if a.deep_tail is not None:
    sim.engine.deep_tail_fraction = a.deep_tail
t_init = time.perf_counter() - t0
laser = GaussianLaser3D(a0=10.0, l0=lam, w0=2e-6, ctau=3e-6, x0=6e-6)   # :44-51 (shortened pulse)
sim.run(12, callbacks=[laser])
torch.cuda.synchronize()
n0 = sum(sim.engine.diagnostics()["nalive"])
t0 = time.perf_counter()
sim.run(a.steps, callbacks=[laser])
torch.cuda.synchronize()
el = time.perf_counter() - t0
d = sim.engine.diagnostics()
n1 = sum(d["nalive"])
print(json.dumps({"metric": "particle-updates/sec (C5 geometry, 2 species, one GPU, Simulation3D)",
                  "value": 0.5 * (n0 + n1) * a.steps / el, "ms_per_step": 1e3 * el / a.steps, "steps": a.steps,
                  "cells": [a.nx, a.ny, a.nz], "alive_start": n0, "alive_end": n1, "init_s": round(t_init, 2),
                  "field_energy_J": d["field_energy"], "kinetic_J": d["kinetic"]}))
