#!/usr/bin/env python3
"""what one call of a laser callback costs (host time and GPU time), 2-D (C3's 1024 boundary nodes) and 3-D (C5's 256 x 256)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lambdapic_amd.laser import GaussianLaser2D, GaussianLaser3D
from lambdapic_amd.simulation import Simulation
from lambdapic_amd.simulation3d import Simulation3D
lam = 0.8e-6
PML2 = {k: "pml" for k in ("xmin", "xmax", "ymin", "ymax")}
def run(sim, laser, tag):
    sim.initialize()
    sim.time = 2e-15
    for _ in range(5):
        laser(sim)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for k in range(200):
        sim.time = 2e-15 + k * sim.dt
        laser(sim)
    e1.record(); host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{tag}: host {host / 200 * 1e3:.3f} ms per call, GPU {e0.elapsed_time(e1) / 200:.3f} ms per call")
run(Simulation(256, 1024, lam / 50, lam / 50, boundary_conditions=PML2), GaussianLaser2D(a0=10.0, l0=lam, w0=2e-6, ctau=2e-6, x0=4e-6), "2-D GaussianLaser2D, 1024 nodes")
run(Simulation3D(64, 256, 256, lam / 20, lam / 10, lam / 10), GaussianLaser3D(a0=10.0, l0=lam, w0=2e-6, ctau=3e-6, x0=6e-6), "3-D GaussianLaser3D, 256 x 256 nodes")
