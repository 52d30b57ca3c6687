#!/usr/bin/env python3
"""host-side profile of the C4 stage loop: cProfile over 200 steps of tools/bench_lwfa.py's setup, top
functions by own time.  Tells whether the step is bound by the Python / launch path or by the GPU."""
import cProfile, io, os, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lambdapic_amd import constants
from lambdapic_amd.laser import SimpleLaser2D
from lambdapic_amd.simulation import MovingWindow, Simulation, Species

C = constants.C_LIGHT
lam = 0.8e-6
dx = dy = lam / 20
nx, ny = 4096, 512
nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C / lam) ** 2 / constants.E_CHARGE ** 2
sim = Simulation(nx, ny, dx, dy, npatch_x=nx // 64, npatch_y=ny // 64, random_seed=1, sort_interval=20)
Ly = ny * dy
dens = lambda x, y: np.where((x > 1e-6) & (y > 1e-6) & (y < Ly - 1e-6), 0.01 * nc, 0.0)
sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=16))
sim.initialize()
cbs = [SimpleLaser2D(a0=2.0, w0=5e-6, ctau=5e-6, l0=lam), MovingWindow(velocity=C, start_time=0.05 * sim.Lx / C)]
sim.run(20, callbacks=cbs)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
sim.run(200, callbacks=cbs)
t_cpu = time.perf_counter() - t0          # host time to ISSUE 200 steps
torch.cuda.synchronize()
pr.disable()
t_all = time.perf_counter() - t0
print(f"issue {1e3 * t_cpu / 200:.3f} ms/step, complete {1e3 * t_all / 200:.3f} ms/step")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])

# ---- one window shift (64 cells, with injection) in isolation
for rep in range(3):
    torch.cuda.synchronize()
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
    sim.shift_window_right(True)
    sim.run(1, callbacks=cbs[:1])
    torch.cuda.synchronize(); pr.disable()
    print(f"shift {rep}: {1e3 * (time.perf_counter() - t0):.1f} ms (shift + the step after it)")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(28)
print(s.getvalue()[:7000])
