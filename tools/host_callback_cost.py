#!/usr/bin/env python3
"""What a registered host (mirror-reading) callback costs the steps in which it does NOT run: C2's plasma at 512 x 512 cells
through Simulation.run with a callback every 100 steps -- part_eb / inv_gamma streamed in every step (the old rule: decided by
registration) against only in the pushes that precede a trigger."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lambdapic_amd import constants
from lambdapic_amd.simulation import Simulation, Species, callback

LAM = 0.8e-6
nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * constants.C_LIGHT / LAM) ** 2 / constants.E_CHARGE ** 2
bc = {k: "periodic" for k in ("xmin", "xmax", "ymin", "ymax")}
for rule in ("none", "per step", "always"):
    sim = Simulation(1024, 1024, LAM / 20, LAM / 20, npatch_x=8, npatch_y=8, boundary_conditions=bc, random_seed=1, sort_interval=20)
    sim.add_species(Species("e", charge=-1, mass=1, density=nc, ppc=64, momentum_sigma=0.0442))
    sim.initialize()
    hits = []

    @callback("end", interval=1000)
    def diag(s):
        hits.append(s.itime)
    cbs = [] if rule == "none" else [diag]
    if rule == "always":
        sim._host_callback_near = lambda cbs_, last: True
    sim.itime = 1                      # (keep the trigger out of the timed steps: the mirror refresh itself is not the point)
    sim.run(25, callbacks=cbs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sim.run(100, callbacks=cbs)
    torch.cuda.synchronize()
    print(f"host callback registered: {rule:9s} {1e3 * (time.perf_counter() - t0) / 100:.3f} ms/step", flush=True)
    del sim
    torch.cuda.empty_cache()
