#!/usr/bin/env python3
"""tools/bench_mirror.py for the 3-D engine: rank 0 of a mirrored periodic 2-slab ring on C5's per-GPU slab
(64 x 256 x 256 cells, 8 ppc), no wire.  --overlap / default in-line J exchange; --single = the N = 1 path."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, argv = sys.argv[:1], sys.argv[1:]
from bench_mirror_comm import MirrorComm
from lambdapic_amd import constants
from lambdapic_amd.engine3d import PicEngine3D

ap = argparse.ArgumentParser()
ap.add_argument("--overlap", action="store_true"); ap.add_argument("--single", action="store_true")
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=4)
ap.add_argument("--transport", default="loopback", choices=["loopback", "rccl", "python"])   # see bench_mirror.py
ap.add_argument("--run-steps", action="store_true")
ap.add_argument("--b-messages", action="store_true", help="the B guard planes travel (local_b_guards off): four message rounds per step")
a = ap.parse_args(argv)
nx, ny, nz, ppc = 64, 256, 256, 8
lam = 0.8e-6
dx, dy, dz = lam / 20, lam / 10, lam / 10
dt = 0.95 / (299792458.0 * np.sqrt(dx ** -2 + dy ** -2 + dz ** -2))
from lambdapic_amd.dist import LoopbackComm
comm = None if a.single else (MirrorComm(nx * dx, 262144) if a.transport == "python" else
                              LoopbackComm(nx * dx, 2, rccl=a.transport == "rccl"))
eng = PicEngine3D(nx, ny, nz, dx, dy, dz, 3, sort_interval=10, comm=comm, migrate_capacity=262144)
eng.overlap = a.overlap
eng.local_b_guards = not a.b_messages
n = nx * ny * nz * ppc
dev = eng.device
g = torch.Generator(device=dev).manual_seed(1)
cell = torch.arange(n, device=dev) // ppc
r = lambda: torch.rand(n, device=dev, dtype=torch.float64, generator=g)
cap = n + eng.arrival_area() + 4096
data = torch.full((8, cap), float("nan"), dtype=torch.float64, device=dev)
data[0, :n] = ((cell // (ny * nz)).double() + r() - 0.5) * dx
data[1, :n] = (((cell // nz) % ny).double() + r() - 0.5) * dy
data[2, :n] = ((cell % nz).double() + r() - 0.5) * dz
for k in (3, 4, 5):
    data[k, :n] = torch.randn(n, device=dev, dtype=torch.float64, generator=g) * 0.0442
data[6, :n] = 1.0 / torch.sqrt(1 + data[3, :n] ** 2 + data[4, :n] ** 2 + data[5, :n] ** 2)
data[7, :n] = 1.742e27 * dx * dy * dz / ppc
eng.add_species_device(-constants.E_CHARGE, constants.M_E, data, n)
for _ in range(a.warmup):
    eng.step(dt)
torch.cuda.synchronize(); t0 = time.perf_counter()
if a.run_steps:
    eng.run_steps(a.steps, dt)
else:
    for _ in range(a.steps):
        eng.step(dt)
torch.cuda.synchronize(); el = time.perf_counter() - t0
d = eng.diagnostics()
print(json.dumps({"what": "3-D, " + ("single slab" if a.single else f"rank 0 of a mirrored 2-slab ring, transport {a.transport}" + (", run_steps" if a.run_steps else "") +
                                      (", B messages" if a.b_messages else "") + (", overlapped" if a.overlap else ", in line")),
                  "ms_per_step": 1e3 * el / a.steps, "alive": d["nalive"][0], "particles": n, "message_window": eng.migrate_window,
                  "charge_rel_err": abs(d["charge"] / (d["nalive"][0] * 1.742e27 * dx * dy * dz / ppc * -constants.E_CHARGE) - 1)}))
