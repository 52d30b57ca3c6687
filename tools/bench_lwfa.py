#!/usr/bin/env python3
"""Config C4 in one piece (BASELINE.json: 2-D LWFA, example/lwfa.py scale): 4096 x 512 cells, 16 ppc electrons
for x > 1 um with 1 um vacuum margins in y, ne = 0.01 nc, SimpleLaser2D(a0 = 2, w0 = 5 um, ctau = 5 um), CPML
on all sides, moving window at c -- through the Simulation facade (stage loop, device-native laser and
window callbacks) on ONE GPU.  Not the headline bench: a realistic non-uniform run that exercises layers,
laser, window shifts with particle injection and re-sorts.  Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lambdapic_amd import constants
from lambdapic_amd.laser import SimpleLaser2D
from lambdapic_amd.simulation import MovingWindow, Simulation, Species

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=4096); ap.add_argument("--ny", type=int, default=512)
ap.add_argument("--ppc", type=int, default=16); ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--start-frac", type=float, default=0.05, help="window start time as a fraction of Lx / c")
a = ap.parse_args()
C = constants.C_LIGHT
lam = 0.8e-6
dx = dy = lam / 20                                           # example/lwfa.py:30-36 (scaled)
nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C / lam) ** 2 / constants.E_CHARGE ** 2
sim = Simulation(a.nx, a.ny, dx, dy, npatch_x=a.nx // 64, npatch_y=a.ny // 64, random_seed=1, sort_interval=20)
Ly = a.ny * dy
dens = lambda x, y: np.where((x > 1e-6) & (y > 1e-6) & (y < Ly - 1e-6), 0.01 * nc, 0.0)
sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=a.ppc))
t0 = time.perf_counter()
sim.initialize()
t_init = time.perf_counter() - t0
laser = SimpleLaser2D(a0=2.0, w0=5e-6, ctau=5e-6, l0=lam)
win = MovingWindow(velocity=C, start_time=a.start_frac * sim.Lx / C)
cbs = [laser, win]
sim.run(20, callbacks=cbs)                                    # warm-up: first sort ...
warm = 20
while getattr(sim, "window_shifts", 0) < 1 and warm < 600:    # ... and the first window shift (one-time
    sim.run(10, callbacks=cbs); warm += 10                    # module loads / layer rebuild: ~150 ms)
torch.cuda.synchronize()
n0 = sim.engine.diagnostics()["nalive"][0]
t0 = time.perf_counter()
sim.run(a.steps, callbacks=cbs)
torch.cuda.synchronize()
el = time.perf_counter() - t0
d = sim.engine.diagnostics()
n1 = d["nalive"][0]
print(json.dumps({"metric": "particle-updates/sec (C4 LWFA on one GPU, moving window)", "value": 0.5 * (n0 + n1) * a.steps / el,
                  "ms_per_step": 1e3 * el / a.steps, "steps": a.steps, "cells": [a.nx, a.ny], "alive_start": n0,
                  "alive_end": n1, "window_shifts": getattr(sim, "window_shifts", 0), "warmup_steps": warm, "init_s": round(t_init, 2),
                  "field_energy_J_per_m": d["field_energy"], "kinetic_J_per_m": d["kinetic"][0]}))
