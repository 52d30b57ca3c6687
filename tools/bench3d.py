#!/usr/bin/env python3
"""3-D measurement (not the headline bench): one GPU's share of BASELINE config C5 -- a 64x256x256-cell
periodic slab (512x256x256 / 8 GPUs), 8 ppc thermal plasma -- through PicEngine3D (tile sort + LDS-tiled
kernel, or --global for the global-memory kernel).  Prints one JSON line with particle-updates/s and the algorithmic HBM rate
(121 B per 3-D particle-update)."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lambdapic_amd import _lib, constants
from lambdapic_amd.engine3d import PicEngine3D, ATTRS3

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=64); ap.add_argument("--ny", type=int, default=256)
ap.add_argument("--nz", type=int, default=256); ap.add_argument("--ppc", type=int, default=8)
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=4)
ap.add_argument("--global", dest="glob", action="store_true"); ap.add_argument("--sort-interval", type=int, default=10)
ap.add_argument("--block-particles", type=int, default=4096)
ap.add_argument("--order", default="striped", choices=["striped", "padded", "column"])
ap.add_argument("--uth", type=float, default=0.0442, help="thermal momentum spread per axis (gamma beta)")
ap.add_argument("--rho", default="continuity", choices=["continuity", "deposited"],
                help="rho between two sorts: from the continuity equation (default) or deposited in every step")
ap.add_argument("--no-fuse", action="store_true", help="per-species launches (lpa_push_deposit_tiled_3d) instead of the "
                                                       "one-launch form (lpa_push_deposit_tiled_multi_3d)")
ap.add_argument("--fixed-sort", action="store_true", help="no early sorts on overflow growth (overflow_sort_fraction = 0)")
ap.add_argument("--species", type=int, default=1, help="split the particles over this many species (same q / m)")
a = ap.parse_args()
lam = 0.8e-6
dx, dy, dz = lam / 20, lam / 10, lam / 10                 # example/laser-target-3d.py:26-31
dt = 0.95 / (299792458.0 * np.sqrt(dx ** -2 + dy ** -2 + dz ** -2))
eng = PicEngine3D(a.nx, a.ny, a.nz, dx, dy, dz, 3, tiled=not a.glob, sort_interval=a.sort_interval,
                  block_particles=a.block_particles)
eng.rho_continuity = a.rho == "continuity"
eng.fuse_species = not a.no_fuse
if a.fixed_sort:
    eng.overflow_sort_fraction = 0
n = a.nx * a.ny * a.nz * a.ppc
if a.order == "padded":
    eng.order = _lib.LPA_ORDER_PADDED
if a.order == "column":
    eng.order = _lib.LPA_ORDER_COLUMN
cap = int(1.6 * n) + 65536 if a.order == "padded" else n
dev = eng.device
g = torch.Generator(device=dev).manual_seed(1)
cell = torch.arange(n, device=dev) // a.ppc
r = lambda: torch.rand(n, device=dev, dtype=torch.float64, generator=g)
data = torch.full((8, cap), float('nan'), dtype=torch.float64, device=dev)
data[0, :n] = ((cell // (a.ny * a.nz)).double() + r() - 0.5) * dx
data[1, :n] = (((cell // a.nz) % a.ny).double() + r() - 0.5) * dy
data[2, :n] = ((cell % a.nz).double() + r() - 0.5) * dz
for k in (3, 4, 5):
    data[k, :n] = torch.randn(n, device=dev, dtype=torch.float64, generator=g) * a.uth
data[6, :n] = 1.0 / torch.sqrt(1 + data[3, :n] ** 2 + data[4, :n] ** 2 + data[5, :n] ** 2)
omega = 2 * np.pi * 299792458.0 / lam
data[7, :n] = constants.EPSILON_0 * constants.M_E * omega ** 2 / constants.E_CHARGE ** 2 * dx * dy * dz / a.ppc
if a.species == 1:
    eng.add_species_device(-constants.E_CHARGE, constants.M_E, data, n)
else:          # the same particles dealt to several species (cell by cell), each in its own store
    assert a.order != "padded"
    for k in range(a.species):
        part = data[:, k:n:a.species].contiguous()
        eng.add_species_device(-constants.E_CHARGE, constants.M_E, part, part.shape[1])
for _ in range(a.warmup):
    eng.step(dt)
eng.kernel_events = []
eng.reserve_kernel_events(2 * a.steps + 8)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    eng.step(dt)
torch.cuda.synchronize(); el = time.perf_counter() - t0
d = eng.diagnostics()
ov = int(eng.species[0]["ws"]["count"].item()) if eng.species[0]["ws"] else None
k_ms = sum(x.elapsed_time(y) for x, y in eng.kernel_events) / a.steps if eng.kernel_events else None
print(json.dumps({"metric": "particle-updates/sec (3-D, %s kernel)" % ("global-memory" if a.glob else "LDS-tiled"),
                  "k1_3d_ms": k_ms, "k1_3d_frac_of_hbm": (121.0 * n / (k_ms * 1e-3) / 8e12) if k_ms else None,
                  "overflow_last_step": ov, "sort_interval": a.sort_interval, "rho": eng.rho_mode(),
                  "rho_steps": eng.rho_steps, "species": a.species, "fused_species_launch": bool(eng.fuse_species),
                  "value": n * a.steps / el,
                  "ms_per_step": 1e3 * el / a.steps, "particles": n, "cells": [a.nx, a.ny, a.nz],
                  "algorithmic_GBps": (121.0 * n) * a.steps / el / 1e9, "alive": sum(d["nalive"]),
                  "charge_rel_err": abs(d["charge"] / (n * float(data[7][0]) * -constants.E_CHARGE) - 1)}))
