#!/usr/bin/env python3
"""A laser-target script the way a user of the reference would write it -- every argument left at its default that can be,
device-native writers next to a plain host callback -- run for 1200 steps with the invariants checked at the end."""
import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lambdapic_amd import constants, h5lite
from lambdapic_amd.callbacks import ExtractSpeciesDensity
from lambdapic_amd.hdf5 import SaveFieldsToHDF5, SaveParticlesToHDF5, SaveSpeciesDensityToHDF5
from lambdapic_amd.laser import GaussianLaser2D
from lambdapic_amd.restart import RestartDump
from lambdapic_amd.simulation import Electron, MovingWindow, Proton, Simulation, callback

lam = 0.8e-6
nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * constants.C_LIGHT / lam) ** 2 / constants.E_CHARGE ** 2
out = tempfile.mkdtemp()
sim = Simulation(nx=768, ny=256, dx=lam / 32, dy=lam / 32, npatch_x=12, npatch_y=4, random_seed=3)
Lx = sim.Lx
ele = Electron(density=lambda x, y: np.where((x > 0.5 * Lx) & (x < 0.5 * Lx + 1e-6), 5 * nc, 0.0), ppc=16, momentum_sigma=0.02)
ion = Proton(density=lambda x, y: np.where((x > 0.5 * Lx) & (x < 0.5 * Lx + 1e-6), 5 * nc, 0.0), ppc=16)
sim.add_species([ele, ion])
energies = []


@callback("end", interval=300)
def host_diag(s):                      # a reference-style callback: reads the patch mirrors
    ke = sum(float(np.sum(p.particles[0].w[~p.particles[0].is_dead] * (1 / p.particles[0].inv_gamma[~p.particles[0].is_dead] - 1)))
             for p in s.patches)
    energies.append((s.itime, ke * constants.M_E * constants.C_LIGHT ** 2))


dens = ExtractSpeciesDensity(sim, ele, interval=400)
cbs = [GaussianLaser2D(a0=5.0, l0=lam, w0=2e-6, ctau=2e-6, x0=3e-6), MovingWindow(velocity=constants.C_LIGHT, start_time=0.55 * Lx / constants.C_LIGHT),
       SaveFieldsToHDF5(prefix=out + "/f", interval=400, components=["ey", "rho"], slice=np.s_[::2, ::2]),
       SaveSpeciesDensityToHDF5(ele, prefix=out + "/d", interval=300), SaveParticlesToHDF5(ion, prefix=out + "/p", interval=300, attrs=["x", "y", "w"]),
       RestartDump(out + "/ckpt", interval=800), host_diag, dens]
sim.run(1199, callbacks=cbs)
eng = sim.engine
live = lambda f: sum(f(sp.q) * sp.cset.arr("w")[: sp.n][~torch.isnan(sp.cset.arr("x")[: sp.n])].sum().item() for sp in eng.species)
# rho of a step holds every particle that was alive when the step began (what it absorbs leaves rho a step later): the
# charge of the live particles is taken before the last step, the charge of rho after it
qw, gross, shifts_before = live(lambda q: q), live(abs), getattr(sim, "window_shifts", 0)
sim.run(1, callbacks=cbs)
exact = getattr(sim, "window_shifts", 0) == shifts_before        # (a shift in that very step drops / injects at stage 'start')
d = eng.diagnostics()
print("alive", d["nalive"], "window shifts", getattr(sim, "window_shifts", 0), "rho steps", dict(eng.rho_steps))
print("host callback saw", energies)
files = sorted(os.path.relpath(os.path.join(r, f), out) for r, _, fs in os.walk(out) for f in fs)
print(len(files), "files:", files[:6], "...")
with h5lite.File(out + "/f/000800.h5", "r") as f:
    assert f["ey"].shape == (384, 128) and f.attrs["slice"] == "[::2, ::2]" and np.abs(f["ey"][:]).max() > 0
with h5lite.File(out + "/p/proton_particles_000300.h5", "r") as f:
    assert len(f["id"]) == len(np.unique(f["id"][:])) > 100000
assert dens.density.shape == (768, 256) and dens.density.max() > 0
assert len(energies) == 4 and energies[-1][1] > 10 * max(energies[0][1], 1e-300)     # the pulse heats the target
padded = eng.grid.view("rho").sum().item() * sim.dx * sim.dy
print("charge of the padded rho array %.6e, of the particles alive before the last step %.6e, gross %.3e" % (padded, qw, gross))
assert abs(padded - qw) <= (1e-9 if exact else 1e-3) * gross, (padded, qw, gross, exact)
assert os.path.exists(out + "/ckpt/ckpt_000800/rank_000000.pkl")
print("user script ok")
