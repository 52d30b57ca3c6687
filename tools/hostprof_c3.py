#!/usr/bin/env python3
"""host-side profile of the C3 stage loop (bench.py's extra_c3 setup, window already moving): how long the host
takes to ISSUE a step against how long the GPU takes to complete it, and where the host time goes (cProfile)."""
import cProfile, io, os, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lambdapic_amd import constants
from lambdapic_amd.laser import GaussianLaser2D
from lambdapic_amd.simulation import MovingWindow, Simulation, Species

C = constants.C_LIGHT
lam = 0.8e-6
nx, ny, ppc = 2048, 1024, 32
dx = dy = lam / 50
nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C / lam) ** 2 / constants.E_CHARGE ** 2
sim = Simulation(nx, ny, dx, dy, npatch_x=nx // 64, npatch_y=ny // 64, random_seed=1, sort_interval=20)
Lx = nx * dx
dens = lambda x, y: np.where((x > Lx / 2) & (x < Lx / 2 + 1e-6), 10 * nc, 0.0)
sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=ppc, momentum_sigma=0.01))
sim.add_species(Species("p", charge=1, mass=1836.0, density=dens, ppc=ppc))
sim.initialize()
cbs = [GaussianLaser2D(a0=10.0, l0=lam, w0=2e-6, ctau=2e-6, x0=4e-6), MovingWindow(velocity=C, start_time=40 * sim.dt)]
sim.run(160, callbacks=cbs)
torch.cuda.synchronize()
for profile in (False, True):
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if profile:
        pr.enable()
    sim.run(200, callbacks=cbs)
    t_cpu = time.perf_counter() - t0          # host time to ISSUE 200 steps
    torch.cuda.synchronize()
    if profile:
        pr.disable()
    t_all = time.perf_counter() - t0
    print(f"profile={profile}: issue {1e3 * t_cpu / 200:.3f} ms/step, complete {1e3 * t_all / 200:.3f} ms/step")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue()[:7000])
