#!/usr/bin/env python3
"""C2's box with a drifting plasma (every particle crosses a cell every 1.5 steps): the worst case for a tile-ordered store.
    python tools/drift_c2.py <ux_drift> [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lambdapic_amd.dist import SlabComm
class A: pass
a = A(); a.nx = 1024; a.ny = 1024; a.ppc = 64; a.sort_interval = 20; a.block_particles = 8192
eng, dt, n = bench.build_engine(a, SlabComm(None, single=True), torch.device("cuda:0"))
drift = float(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sp = eng.species[0]
s = sp.cset
s.arr("ux")[: sp.n] += drift
s.arr("inv_gamma")[: sp.n] = 1.0 / torch.sqrt(1 + s.arr("ux")[: sp.n] ** 2 + s.arr("uy")[: sp.n] ** 2 + s.arr("uz")[: sp.n] ** 2)
for _ in range(10):
    eng.step(dt)
eng.kernel_events = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    eng.step(dt)
torch.cuda.synchronize(); el = time.perf_counter() - t0
k = sum(x.elapsed_time(y) for x, y in eng.kernel_events) / steps
d = eng.diagnostics()
print("ux drift %.2f: %.3f ms per step, k1(tiled) %.3f ms, %.3e particle-updates/s, sort interval now %s, rho steps %s, alive %d" %
      (drift, 1e3 * el / steps, k, n * steps / el, getattr(sp, "sort_interval_now", None), eng.rho_steps, d["nalive"][0]))
