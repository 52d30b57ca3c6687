"""Device-native diagnostics callbacks (mirrors of the reference's, without the HDF5 / plotting parts).

``get_fields`` -- `callback/utils.py:26-237`: whole-box field arrays (a z-plane in 3-D) on rank 0.

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


class ExtractSpeciesDensity:
    stage = "current_deposition"
    device_native = True

    def __init__(self, sim, species, interval=100, slice=None):
        if slice is not None:
            raise NotImplementedError("slices of the density are not supported")
        self.species, self.interval = species, interval
        self.prev_rho = None
        self._dev = None

    @property
    def ispec_target(self):
        return self.species.ispec

    @staticmethod
    def _rho(sim):
        eng = sim.engine
        g = eng.ng
        if getattr(sim, "dimension", 2) == 3:
            return eng.view("rho")[g:-g, g:-g, g:-g]
        return eng.grid.view("rho")[g:-g, g:-g]

    def __call__(self, sim):
        t = self.ispec_target
        if t > 0 and sim.ispec == t - 1:
            sim.sync_currents()
            self.prev_rho = self._rho(sim).clone()
        elif sim.ispec == t:
            sim.sync_currents()
            rho = self._rho(sim)
            d = rho.clone() if t == 0 else rho - self.prev_rho
            self._dev = d / self.species.q
            self.prev_rho = None

    @property
    def density_device(self) -> torch.Tensor:
        return self._dev

    @property
    def density(self) -> np.ndarray:
        """this rank's slab (zeros before the first trigger)"""
        return self._dev.cpu().numpy() if self._dev is not None else np.zeros(0)

    def gather(self, sim):
        """the whole box on rank 0 (None elsewhere): slabs concatenated along x"""
        if self._dev is None:
            raise RuntimeError("ExtractSpeciesDensity.gather() before the callback was triggered")
        return _gather_slabs(sim, self._dev)


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
