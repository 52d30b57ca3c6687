"""Device-native diagnostics callbacks (mirrors of the reference's; the HDF5 writers are in hdf5.py, no plotting).

``get_fields`` -- `callback/utils.py:26-237`: whole-box field arrays (a z-plane in 3-D) on rank 0.
``SetMomentum`` / ``SetTemperature`` / ``SetMomentumAndTemperature`` -- `callback/utils.py:842-1049` on the device.

``ExtractSpeciesDensity`` -- `callback/utils.py:240-293` on top of `callback/hdf5.py:402-482`: at stage
``current_deposition`` (which runs once per species, right after that species' deposit) the currents are
synchronised and the species' number density is the rho it just added, divided by its charge:
``(rho - rho_before_this_species) / q``.  Here rho lives on the device, so the callback never triggers the
host-mirror refresh; ``density`` is this rank's slab (numpy, interior cells), ``gather()`` assembles the
box on rank 0 like the reference's writer does.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import constants
from .hdf5 import SaveSpeciesDensityToHDF5
from .slices import normalize_slice


class ExtractSpeciesDensity(SaveSpeciesDensityToHDF5):
    """the density writer's computation (hdf5.py) with the result kept instead of written, as in the reference, where
    this class derives from ``SaveSpeciesDensityToHDF5`` too (`callback/utils.py:240-293`).  ``slice``: an ``np.s_``
    selection of the box (`callback/hdf5.py:14-99`)."""

    def __init__(self, sim, species, interval=100, slice=None):
        self.stage, self.interval = self.DEFAULT_STAGE, interval
        self.species, self.prev_rho = species, None
        self.slice = slice
        dims = (sim.nx, sim.ny) + ((sim.nz,) if getattr(sim, "dimension", 2) == 3 else ())
        self._normalized_slice = normalize_slice(len(dims), slice, dims)       # (ValueError now, not at the first call)
        self._dev = self._out_idx = self._shape = None

    def _deliver(self, sim, density, out_idx, shape, norm):
        self._dev, self._out_idx, self._shape = density, out_idx, shape

    @property
    def density_device(self) -> torch.Tensor:
        """this rank's share of the selection (``None`` when the slab holds nothing of it)"""
        return self._dev

    @property
    def density(self) -> np.ndarray:
        """this rank's share (zeros before the first trigger)"""
        return self._dev.cpu().numpy() if self._dev is not None else np.zeros(0)

    def gather(self, sim):
        """the whole selection on rank 0 (None elsewhere)"""
        if self._shape is None:
            raise RuntimeError("ExtractSpeciesDensity.gather() before the callback was triggered")
        if self._normalized_slice is None:
            return _gather_slabs(sim, self._dev)
        parts = sim.mpi.comm.gather((self._out_idx, self.density))
        if parts is None:
            return None
        out = np.zeros(self._shape)
        for idx, block in parts:
            if idx is not None:
                out[idx] = block
        return out


def _gather_slabs(sim, mine: torch.Tensor):
    """rank 0: the slabs of all ranks concatenated along x (numpy); other ranks: None"""
    comm = sim.comm
    if comm.size == 1:
        return mine.cpu().numpy()
    mine = mine.contiguous()
    if dist.get_backend(comm.group) == "gloo":
        mine = mine.cpu()
    parts = [torch.empty_like(mine) for _ in range(comm.size)] if comm.rank == 0 else None
    dist.gather(mine, parts, dst=0, group=comm.group)
    return torch.cat(parts, dim=0).cpu().numpy() if comm.rank == 0 else None


def get_fields(sim, fields, slice_at=None):
    """`callback/utils.py:26-237` (``get_fields`` / ``get_fields_2d`` / ``get_fields_3d``): the interior of the
    named field arrays over the whole box, assembled on rank 0 (``None`` on the other ranks).  In 3-D the
    plane ``z = slice_at`` (default ``Lz / 2``), index ``int((slice_at + dz / 2) / dz)`` as in the reference
    (`:176-177`), so the result is always 2-dimensional ``(nx, ny)``.  Reads the device arrays directly:
    no host-mirror refresh."""
    out = []
    if not fields:
        return out
    eng = sim.engine
    g = eng.ng
    if getattr(sim, "dimension", 2) == 3:
        if slice_at is None:
            slice_at = sim.Lz / 2
        if slice_at < 0 or slice_at > sim.Lz:
            raise ValueError(f"Slice position {slice_at} is outside the simulation domain [0, {sim.Lz}]")
        iz = min(int((slice_at + sim.dz / 2) / sim.dz), sim.nz - 1)
        for name in fields:
            out.append(_gather_slabs(sim, eng.view(name)[g:-g, g:-g, g + iz]))
    else:
        for name in fields:
            out.append(_gather_slabs(sim, eng.grid.view(name)[g:-g, g:-g]))
    return out


# ---- momentum / temperature initialisers (`callback/utils.py:842-1049`), on the device ----------------------
def _species_arrays(sim, species):
    """(ux, uy, uz, inv_gamma, alive) device views of one species' store, 2-D or 3-D engine"""
    eng = sim.engine
    if getattr(sim, "dimension", 2) == 3:
        sp = eng.species[species.ispec]
        d = sp["data"][:, : sp["n"]]
        return d[3], d[4], d[5], d[6], ~torch.isnan(d[0])
    sp = eng.species[species.ispec]
    sp.refresh_inv_gamma()
    a = sp.cset.arr
    n = sp.n
    return a("ux")[:n], a("uy")[:n], a("uz")[:n], a("inv_gamma")[:n], ~torch.isnan(a("x")[:n])


class SetMomentum:
    """`callback/utils.py:842-888`: set (or add to) the momenta ``u = gamma beta`` of every live particle of a
    species; ``inv_gamma`` follows.  Stage 'init', by default once at the first step.  Device native."""
    stage = "init"
    device_native = True

    def __init__(self, species, momentum, interval=None, add=False):
        self.species, self.momentum, self.add = species, [float(v) for v in momentum], add
        self.interval = (lambda sim: sim.itime == 0) if interval is None else interval

    def __call__(self, sim):
        ux, uy, uz, ig, alive = _species_arrays(sim, self.species)
        for u, t in zip((ux, uy, uz), self.momentum):
            u[alive] = (u[alive] + t) if self.add else t
        ig[alive] = torch.rsqrt(1 + ux[alive] ** 2 + uy[alive] ** 2 + uz[alive] ** 2)


def sample_maxwell_juttner(size, theta, generator, device):
    """gamma ~ Maxwell-Juettner(theta = kT / mc^2), isotropic directions -> (ux, uy, uz) device tensors.  The
    three regimes of the reference (`callback/utils.py:988-1049`): gamma - 1 ~ Gamma(3/2, theta) for theta <= 0.01,
    uniform proposal + rejection on the exact pdf up to theta = 0.5, Gamma(3, theta) proposal accepted with
    probability beta above.  Other random streams than numpy's: statistically equivalent, not bit-equal."""
    f64 = dict(dtype=torch.float64, device=device)
    rand = lambda n: torch.rand(n, generator=generator, **f64)

    def gamma_rv(shape_k, n):      # Gamma(k, scale = theta) from the sum / Box-Muller forms of k = 3/2 and 3
        e = lambda: -torch.log1p(-rand(n))
        if shape_k == 3:
            return theta * (e() + e() + e())
        z = torch.randn(n, generator=generator, **f64)
        return theta * (e() + 0.5 * z * z)          # Gamma(1) + Gamma(1/2)

    if theta <= 0.01:
        g = 1.0 + gamma_rv(1.5, size)
    elif theta <= 0.5:
        from scipy.optimize import minimize_scalar
        from scipy.special import kn
        k2 = float(kn(2, 1.0 / theta))
        pdf_np = lambda gg: gg * np.sqrt(gg * gg - 1.0) / (theta * k2) * np.exp(-gg / theta)
        gmax = 1.0 + 10.0 * theta
        fmax = -minimize_scalar(lambda gg: -pdf_np(gg), bounds=(1.0, gmax), method="bounded").fun
        M = 1.1 * fmax + 1e-10
        g = torch.empty(size, **f64)
        have = 0
        while have < size:
            n = max(int(1.3 * (size - have) * M * (gmax - 1.0)) + 1024, size - have)   # expected acceptance
            prop = 1.0 + (gmax - 1.0) * rand(n)
            f = prop * torch.sqrt(prop * prop - 1.0) / (theta * k2) * torch.exp(-prop / theta)
            ok = prop[M * rand(n) < f][: size - have]
            g[have:have + ok.numel()] = ok
            have += ok.numel()
    else:
        g = torch.empty(size, **f64)
        have = 0
        while have < size:
            n = 2 * (size - have) + 1024
            prop = gamma_rv(3, n)
            beta = torch.sqrt(torch.clamp(1.0 - 1.0 / (prop * prop), min=0.0))
            ok = prop[(prop >= 1.0) & (rand(n) < beta)][: size - have]
            g[have:have + ok.numel()] = ok
            have += ok.numel()
    u = torch.sqrt(g * g - 1.0)
    phi = 2 * np.pi * rand(size)
    ct = 2.0 * rand(size) - 1.0
    st = torch.sqrt(1.0 - ct * ct)
    return u * st * torch.cos(phi), u * st * torch.sin(phi), u * ct


class SetTemperature:
    """`callback/utils.py:922-972`: momenta of a species from a Maxwell-Juettner distribution of the given
    temperature [eV] (a list = anisotropic: uy, uz stretched by T_y / T_x, T_z / T_x like the reference);
    ``add=True`` puts the thermal spread on top of the existing momenta.  Stage 'init', once by default."""
    stage = "init"
    device_native = True

    def __init__(self, species, temperature, interval=None, add=False, seed=None):
        self.species, self.add, self.seed = species, add, seed
        self.temperature = [float(temperature)] * 3 if isinstance(temperature, (int, float)) else \
            [float(t) for t in temperature]
        self.interval = (lambda sim: sim.itime == 0) if interval is None else interval

    def __call__(self, sim):
        ux, uy, uz, ig, alive = _species_arrays(sim, self.species)
        n = int(alive.sum().item())
        if n == 0:
            return
        gen = torch.Generator(device=ux.device)
        base = self.seed if self.seed is not None else (getattr(sim, "random_seed", None) or 0)
        gen.manual_seed((int(base) * 1000003 + 7919 * self.species.ispec + 104729 * sim.comm.rank + sim.itime) % (2 ** 63))
        theta = self.temperature[0] * constants.E_CHARGE / (self.species.m * constants.C_LIGHT ** 2)
        tx, ty, tz = sample_maxwell_juttner(n, theta, gen, ux.device)
        ty, tz = ty * (self.temperature[1] / self.temperature[0]), tz * (self.temperature[2] / self.temperature[0])
        for u, t in zip((ux, uy, uz), (tx, ty, tz)):
            u[alive] = (u[alive] + t) if self.add else t
        ig[alive] = torch.rsqrt(1 + ux[alive] ** 2 + uy[alive] ** 2 + uz[alive] ** 2)


class SetMomentumAndTemperature:
    """`callback/utils.py:891-920`: bulk momentum first, thermal spread on top"""
    stage = "init"
    device_native = True

    def __init__(self, species, momentum, temperature, interval=None, add=False, seed=None):
        self._m = SetMomentum(species, momentum, interval, add=add)
        self._t = SetTemperature(species, temperature, interval, add=True, seed=seed)
        self.interval = self._m.interval

    def __call__(self, sim):
        self._m(sim)
        self._t(sim)
