"""Simulation3D -- the reference's facade / stage-callback protocol (`simulation/simulation.py:858-1141`,
3-D class `:1143-1292`) on top of ``PicEngine3D``: same stage list, same facade names, callbacks
``cb(sim)`` with ``.stage`` / ``.interval``; host callbacks see ``sim.patches[i].fields`` /
``.particles[ispec]`` mirrors in the reference's layout (refreshed before the stage, written back after
it), device-native ones (lasers) touch the engine directly.

Only the fused pusher path exists in 3-D (`unified_boris_pusher_cpu_3d`); a callback in one of the five
pusher stages -- which makes the reference fall back to its split kernels -- is refused.
"""
from __future__ import annotations

import numpy as np
import torch

from . import constants
from .dist import SlabComm
from .engine3d import ATTRS3, NROWS3, SIDES3, PicEngine3D
from .fields import FIELD_ATTRS, Fields3D, from_device_layout, to_device_layout
from .particles import ParticlesBase
from .simulation import (Callback, MPIFacade, Simulation, Species, _Facade, callback, interval_triggered,  # noqa: F401
                         load_block_device, validate_interval)          # (shared with 2-D)


class Patch3D:
    """host mirror of one patch (`core/patch/patch.py:198-386` for the attributes callbacks use)"""

    def __init__(self, index, ip, origin, n, d, n_guard, nspecies):
        self.index = index
        self.ipatch_x, self.ipatch_y, self.ipatch_z = ip
        self.x0, self.y0, self.z0 = origin
        self.nx, self.ny, self.nz = n
        self.dx, self.dy, self.dz = d
        self.fields = Fields3D(*n, *d, *origin, n_guard)
        self.particles = [ParticlesBase(ipatch=index) for _ in range(nspecies)]
        self.pml_boundary = []


class MaxwellSolver3D(_Facade):
    def update_efield(self, dt):
        if self._enabled:
            self.sim.engine.update_efield(dt)

    def update_bfield(self, dt):
        if self._enabled:
            self.sim.engine.update_bfield(dt)


class BorisPusher3D(_Facade):
    def __init__(self, sim, ispec):
        super().__init__(sim)
        self.ispec = ispec

    def __call__(self, dt, unified=True):
        if not unified:
            raise NotImplementedError("3-D: only the fused pusher path exists")
        if self._enabled:
            self.sim.engine.push_deposit(self.ispec, dt)


class ParticleSort3D(_Facade):
    def __init__(self, sim, ispec):
        super().__init__(sim)
        self.ispec = ispec

    def __call__(self, force=False):
        eng = self.sim.engine
        sp = eng.species[self.ispec]
        if self._enabled and eng.tiled and (force or eng.sort_due(sp)):
            eng.sort(self.ispec)


class DevicePatches3D:
    def __init__(self, sim, mirrors):
        self.sim, self._m = sim, mirrors

    def __getitem__(self, i):
        return self._m[i]

    def __len__(self):
        return len(self._m)

    def __iter__(self):
        return iter(self._m)

    def sync_guard_fields(self, attrs=("ex", "ey", "ez", "bx", "by", "bz")):
        which = (1 if any(a in attrs for a in ("ex", "ey", "ez")) else 0) | \
                (2 if any(a in attrs for a in ("bx", "by", "bz")) else 0)
        self.sim.engine.sync_guard_fields(which)

    def sync_currents(self):
        self.sim.engine.sync_currents()

    def sync_particles(self):
        for i in range(len(self.sim.species)):
            self.sim.engine.sync_particles(i)


class Simulation3D:
    STAGES = ["init", "start", "maxwell_1", "_push_position_1", "_interpolator", "_qed", "_push_momentum",
              "_push_position_2", "current_deposition", "qed_create_particles", "_laser", "maxwell_2", "end",
              "final"]
    DEFAULT_STAGE = "end"
    _PUSHER_STAGES = {"_push_position_1", "_interpolator", "_qed", "_push_momentum", "_push_position_2"}

    def __init__(self, nx, ny, nz, dx, dy, dz, npatch_x=1, npatch_y=1, npatch_z=1, nsteps=None, sim_time=None,
                 dt_cfl=0.95, n_guard=3, boundary_conditions=None, cpml_thickness=6, random_seed=None,
                 device="cuda:0", comm=None, sort_interval=10, capacity_factor=1.5, block_particles=4096):
        bc = dict(boundary_conditions or {k: "pml" for k in SIDES3})     # simulation.py:1180-1187
        if dt_cfl > 1.0:
            raise ValueError("dt_cfl must be <= 1")
        self.boundary_conditions, self.cpml_thickness = bc, int(cpml_thickness)
        self.comm = comm or SlabComm(None, periodic=bc["xmin"] == "periodic")
        self.mpi = MPIFacade(self)          # what callbacks know as sim.mpi (rank, size, comm, sync_*_start/_wait)
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self.dx, self.dy, self.dz = float(dx), float(dy), float(dz)
        if self.nx % self.comm.size or (self.nx // self.comm.size) % npatch_x or self.ny % npatch_y or self.nz % npatch_z:
            raise ValueError("the grid must split evenly over ranks and patches")
        self.npatch = (npatch_x, npatch_y, npatch_z)
        self.n_guard = n_guard
        self.dt = dt_cfl * (self.dx ** -2 + self.dy ** -2 + self.dz ** -2) ** -0.5 / constants.C_LIGHT  # :1288
        self.Lx, self.Ly, self.Lz = self.nx * self.dx, self.ny * self.dy, self.nz * self.dz
        self.nsteps, self.sim_time, self.random_seed = nsteps, sim_time, random_seed
        self.device, self.sort_interval, self.capacity_factor = device, sort_interval, capacity_factor
        self.block_particles = block_particles
        self.species: list[Species] = []
        self.itime, self.time, self.ispec = 0, 0.0, None
        self.initialized, self.current_synced = False, False
        self.dimension = 3

    def add_species(self, species):
        for s in species if isinstance(species, (list, tuple)) else [species]:
            s.ispec = len(self.species)
            self.species.append(s)

    # ---- initialisation -----------------------------------------------------------------------------------
    def initialize(self):
        nxl = self.nx // self.comm.size
        self.engine = PicEngine3D(nxl, self.ny, self.nz, self.dx, self.dy, self.dz, self.n_guard, self.device,
                                  sort_interval=self.sort_interval, block_particles=self.block_particles,
                                  comm=self.comm, boundary_conditions=self.boundary_conditions,
                                  cpml_thickness=self.cpml_thickness)
        px, py, pz = self.npatch
        self.n_per_patch = (nxl // px, self.ny // py, self.nz // pz)
        npp, d = self.n_per_patch, (self.dx, self.dy, self.dz)
        mirrors, idx = [], 0
        for k in range(pz):
            for j in range(py):
                for i in range(px):
                    org = (self.engine.x0 + i * npp[0] * self.dx, j * npp[1] * self.dy, k * npp[2] * self.dz)
                    mirrors.append(Patch3D(idx, (i, j, k), org, npp, d, self.n_guard, len(self.species)))
                    idx += 1
        self.patches = DevicePatches3D(self, mirrors)
        for s in self.species:
            blocks = []
            for p in mirrors:
                org = (p.x0, p.y0, p.z0)
                b = load_block_device(s, org, npp, d, self._seed(s, org), self.device)
                if b is not None:
                    blocks.append(b)
                p.particles[s.ispec].initialize(0)     # the mirrors fill at the first download()
            n_tot = sum(b["x"].numel() for b in blocks)
            cap = int(n_tot * self.capacity_factor) + 65536 + self.engine.arrival_area()
            data = torch.full((NROWS3, cap), float("nan"), dtype=torch.float64, device=self.device)
            o = 0
            for b in blocks:
                k = b["x"].numel()
                for i, a in enumerate(ATTRS3):
                    data[i, o:o + k] = b[a]
                o += k
            # ids: rank << 50 | one running count per rank and species (the 2-D rule, simulation.py:_next_ids;
            # reference layout `core/particles.py:91-116`); loading, window injection and appends draw from it
            ids = self.engine.new_ids(len(self.engine.species), n_tot)
            self.engine.add_species_device(s.q, s.m, data, n_tot, ids=ids)
        self.maxwell = MaxwellSolver3D(self)
        self.pusher = [BorisPusher3D(self, i) for i in range(len(self.species))]
        self.sorter = [ParticleSort3D(self, i) for i in range(len(self.species))]
        self.initialized = True

    def _seed(self, s, org):
        return None if self.random_seed is None else \
            [self.random_seed, s.ispec] + [int(round(o / dd)) for o, dd in zip(org, (self.dx, self.dy, self.dz))]

    @property
    def nx_per_patch(self):
        return self.n_per_patch[0]

    @property
    def ny_per_patch(self):
        return self.n_per_patch[1]

    @property
    def nz_per_patch(self):
        return self.n_per_patch[2]

    def shift_window_left(self, inject):
        """recycle the rightmost patch column (`callback/utils.py:622-648`): the window moves one patch width to -x"""
        self.shift_window_right(inject, direction=-1)

    def shift_window_right(self, inject, direction=1):
        """recycle the leftmost patch column (`callback/utils.py:594-620`, 3-D relabelling `:705-730`)"""
        eng, n = self.engine, self.n_per_patch[0]
        eng.shift_window(direction * n)
        self.window_shifts = getattr(self, "window_shifts", 0) + 1
        for p in self.patches:
            p.x0 += direction * n * self.dx
            p.fields.x0 = p.x0
            p.fields.xaxis += direction * n * self.dx
        if not inject or self.comm.rank != (self.comm.size - 1 if direction > 0 else 0):
            return
        px, py, pz = self.npatch
        npp, d = self.n_per_patch, (self.dx, self.dy, self.dz)
        x_new = eng.x0 + (eng.n[0] - n) * self.dx if direction > 0 else eng.x0
        for s in self.species:
            for k in range(pz):
                for j in range(py):
                    org = (x_new, j * npp[1] * self.dy, k * npp[2] * self.dz)
                    b = load_block_device(s, org, npp, d, self._seed(s, org), self.device)
                    if b is not None:     # fresh ids from the engine's per-rank counter
                        eng.append_device(s.ispec, torch.stack([b[a] for a in ATTRS3]))

    # ---- host mirrors <-> device ----------------------------------------------------------------------------
    def _upload_particles(self, ispec):
        """mirrors -> device: live slots only (``is_dead`` or a NaN position = dead, the reference's rule
        `unified_pusher_3d.c` strip test); ``_id`` travels, so a particle keeps its identity through any number
        of mirror round trips"""
        sp = self.engine.species[ispec]
        live = [~q.is_dead & ~np.isnan(q.x) & ~np.isnan(q.y) & ~np.isnan(q.z)
                for q in (p.particles[ispec] for p in self.patches)]
        cols = [np.concatenate([getattr(p.particles[ispec], a)[m] for p, m in zip(self.patches, live)])
                for a in ATTRS3 + ("_id",)]
        n = cols[0].size
        if n > sp["data"].shape[1] - self.engine.arrival_area():
            raise RuntimeError("particle capacity exceeded by the host mirrors")
        sp["data"][:, :n] = torch.from_numpy(np.stack(cols)).to(self.device)
        sp["data"][0, n:] = float("nan")
        sp["n"], sp["n_sorted"], sp["tiling"] = n, 0, None
        sp["c"] = self.engine._cstruct(sp["data"], n)

    def download(self):
        eng, g = self.engine, self.n_guard
        npp = self.n_per_patch
        for name in FIELD_ATTRS:
            slab = eng.view(name).cpu().numpy()
            for p in self.patches:
                o = [ip * n for ip, n in zip((p.ipatch_x, p.ipatch_y, p.ipatch_z), npp)]
                blk = slab[tuple(slice(a, a + n + 2 * g) for a, n in zip(o, npp))]
                getattr(p.fields, name)[...] = from_device_layout(blk, g)
        for s in self.species:
            d = eng.download_species(s.ispec)
            cell = [np.floor((d[a] - o) / dd + 0.5).astype(int) // n for a, o, dd, n in
                    zip("xyz", (eng.x0, 0.0, 0.0), (self.dx, self.dy, self.dz), npp)]
            cell = [np.clip(c, 0, m - 1) for c, m in zip(cell, self.npatch)]
            owner = cell[0] + self.npatch[0] * (cell[1] + self.npatch[1] * cell[2])
            for k, p in enumerate(self.patches):
                sel = owner == k
                q = p.particles[s.ispec]
                q.initialize(int(sel.sum()))
                for a in ATTRS3 + ("_id",):
                    getattr(q, a)[:] = d[a][sel]

    def upload(self):
        eng, g = self.engine, self.n_guard
        npp = self.n_per_patch
        for name in FIELD_ATTRS:
            slab = eng.view(name).cpu().numpy()        # guards outside the mirrors' reach keep their values
            for p in self.patches:
                o = [ip * n + g for ip, n in zip((p.ipatch_x, p.ipatch_y, p.ipatch_z), npp)]
                a = to_device_layout(getattr(p.fields, name), g)
                slab[tuple(slice(b, b + n) for b, n in zip(o, npp))] = a[tuple(slice(g, g + n) for n in npp)]
            eng.view(name).copy_(torch.from_numpy(slab))
        for s in self.species:
            self._upload_particles(s.ispec)

    # ---- the stage loop --------------------------------------------------------------------------------------
    _triggered = Simulation._triggered

    def _run_stage(self, table, stage):
        cbs = self._triggered(table.get(stage, []))
        if not cbs:
            return
        host = [cb for cb in cbs if not getattr(cb, "device_native", False)]
        if host:
            self.download()
        for cb in cbs:
            cb(self)
        if host:
            self.upload()

    def update_lists(self):
        """the reference re-points its facades at the (possibly re-allocated) per-patch arrays
        (`simulation/simulation.py:781-824`; called by RestartDump.load).  The facades here hold no array
        pointers -- every call reads the engine's current stores -- so there is nothing to re-point."""

    _INNER_STAGES = Simulation._INNER_STAGES
    _fused_step = Simulation._fused_step

    def sync_currents(self):
        if not self.current_synced:
            self.patches.sync_currents()
            self.current_synced = True

    _can_defer_e2 = Simulation._can_defer_e2
    defer_e2 = True

    def run(self, nsteps=None, sim_time=None, callbacks=None, stop_callback=None):
        if nsteps is not None and sim_time is not None:
            raise ValueError("Cannot specify both nsteps and sim_time in run() method")
        from .simulation import _NO_STOP
        stop_callback = _NO_STOP if stop_callback is None else stop_callback
        if not self.initialized:
            self.initialize()
        table = {}
        for cb in callbacks or []:
            validate_interval(getattr(cb, "interval", 1))
            table.setdefault(getattr(cb, "stage", self.DEFAULT_STAGE), []).append(cb)
        for st in table:
            if st not in self.STAGES:
                raise ValueError(f"unknown stage {st!r}")
            if st in self._PUSHER_STAGES:
                # the reference cannot take this path either: PusherBase.push_position
                # (core/pusher/pusher.py:103-110) moves particles in 2-D only, so a 3-D run with a
                # callback in a pusher stage would silently stop advancing positions
                raise NotImplementedError(f"3-D: the reference has no split 3-D position push; "
                                          f"callback stage {st!r} is not available")
        if nsteps is None:
            nsteps = int(sim_time / self.dt) if sim_time is not None else \
                (self.nsteps if self.nsteps is not None else int(self.sim_time / self.dt))
        self.engine.rho_continuity_blocked = False      # (decided step by step below)
        self._run_stage(table, "init")
        # a RestartDump among the callbacks may ask for a last dump (signal): simulation.py:889-894
        restart_cb = next((cb for cb in callbacks or [] if cb.__class__.__name__ == "RestartDump"), None)
        E, B = ["ex", "ey", "ez"], ["bx", "by", "bz"]
        eng = self.engine
        last_istep = self.itime + nsteps - 1
        for self.istep in range(self.itime, self.itime + nsteps):
            self.engine._dt_hint = self.dt      # (rho.py, the engines' first sort: the step's dt before any push)
            # a callback between the species' deposits reads per-species rho: that step deposits rho for real
            self.engine.rho_continuity_blocked = bool(self._triggered(table.get("current_deposition", [])))
            self._run_stage(table, "start")
            defer = self._can_defer_e2(table, self.istep == last_istep, stop_callback is _NO_STOP) and not (restart_cb is not None and restart_cb._dump_requested)
            if self._fused_step(table, True, defer):
                self._run_stage(table, "maxwell_2")
                self._run_stage(table, "end")
                if restart_cb is not None and restart_cb._dump_requested:
                    restart_cb._call(self)
                    return
                self.time += self.dt
                self.itime += 1
                if stop_callback():
                    return "stop by callback"
                continue
            self.maxwell.update_efield(0.5 * self.dt)
            self.patches.sync_guard_fields(E)
            self.maxwell.update_bfield(0.5 * self.dt)
            self.patches.sync_guard_fields(B)
            self._run_stage(table, "maxwell_1")
            for ispec in range(len(self.species)):
                self.sorter[ispec]()
            eng.reset_current()
            self.current_synced = False
            fused_all = not table.get("current_deposition") and eng.overlap and \
                all(p._enabled for p in self.pusher) and eng.push_deposit_overlapped(self.dt)
            if fused_all:
                self.current_synced = True
            for ispec in range(len(self.species) if not fused_all else 0):
                self.ispec = ispec
                self.pusher[ispec](self.dt, unified=True)
                self.current_synced = False          # simulation.py:991: every deposit un-syncs the currents
                self._run_stage(table, "current_deposition")
            self.sync_currents()
            self.ispec = None
            self.patches.sync_particles()
            self._run_stage(table, "qed_create_particles")
            self.maxwell.update_bfield(0.5 * self.dt)
            self._run_stage(table, "_laser")
            self.patches.sync_guard_fields(B)
            self.maxwell.update_efield(0.5 * self.dt)
            self.patches.sync_guard_fields(E)
            self._run_stage(table, "maxwell_2")
            self._run_stage(table, "end")
            if restart_cb is not None and restart_cb._dump_requested:      # simulation.py:1124-1127
                restart_cb._call(self)
                return
            self.time += self.dt
            self.itime += 1
            if stop_callback():
                return "stop by callback"
        self.engine._flush_e2()
        self._run_stage(table, "final")
