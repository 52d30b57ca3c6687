#!/usr/bin/env python3
"""K1-3D on a grid that is no multiple of the 4 x 4 x 16 tile (64 x 254 x 250 cells, 8 ppc): the tile-sorted path with
partial tiles against the global-atomics kernels the engine used to fall back to for such grids."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lambdapic_amd import constants
from lambdapic_amd.engine3d import PicEngine3D

def run(n3, tiled, steps=24, ppc=8):
    lam = 0.8e-6
    d3 = (lam / 20, lam / 10, lam / 10)
    dt = 0.95 / (constants.C_LIGHT * np.sqrt(sum(d ** -2 for d in d3)))
    eng = PicEngine3D(*n3, *d3, 3, tiled=tiled, sort_interval=10)
    dev = eng.device
    n = int(np.prod(n3)) * ppc
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    cell = torch.arange(n, device=dev) // ppc
    data = torch.full((9, int(1.2 * n) + 300000), float("nan"), dtype=torch.float64, device=dev)
    idx = (cell // (n3[1] * n3[2]), (cell // n3[2]) % n3[1], cell % n3[2])
    for a in range(3):
        data[a, :n] = (idx[a] + torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5) * d3[a]
        data[3 + a, :n] = torch.randn(n, dtype=torch.float64, device=dev, generator=gen) * 0.0442
    data[6, :n] = torch.rsqrt(1 + data[3, :n] ** 2 + data[4, :n] ** 2 + data[5, :n] ** 2)
    nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * constants.C_LIGHT / lam) ** 2 / constants.E_CHARGE ** 2
    data[7, :n] = nc * np.prod(d3) / ppc
    data[8, :n] = torch.arange(n, device=dev, dtype=torch.int64).view(torch.float64)
    eng.add_species_device(-constants.E_CHARGE, constants.M_E, data, n)
    for _ in range(4):
        eng.step(dt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(dt)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    d = eng.diagnostics()
    return ms, n, d["field_energy"], d["kinetic"][0]

cases = [((64, 256, 256), 8, (True, False)), ((64, 254, 250), 8, (True, False))]
if len(sys.argv) > 1:          # nx ny nz ppc: one shape on the tiled path (e.g. deep tiles: 16 16 64 1024)
    cases = [(tuple(int(v) for v in sys.argv[1:4]), int(sys.argv[4]), (True,))]
for n3, ppc, modes in cases:
    for tiled in modes:
        ms, n, fe, ke = run(n3, tiled, ppc=ppc)
        print(f"{n3} tiled={tiled}: {ms:.2f} ms/step, {n / ms / 1e6:.2f} G particle-updates/s, field energy {fe:.6e}, kinetic {ke:.9e}", flush=True)
        torch.cuda.empty_cache()
