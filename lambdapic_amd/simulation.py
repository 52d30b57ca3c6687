"""Host-side mirror of the reference's facade / stage-callback interface for the hot path.

The reference's ``Simulation.run`` (`simulation/simulation.py:858-1141`) walks fixed stages per step
and calls one facade method per stage; callbacks ``cb(sim)`` with ``.stage`` / ``.interval`` run
between them and may read or write any field / particle array of ``sim.patches``.  This module
reproduces exactly that protocol on top of ``PicEngine2D``:

    sim.maxwell.update_efield(dt) / update_bfield(dt)       MaxwellSolver2D   (core/maxwell/solver/solver.py:143-190)
    sim.patches.sync_guard_fields(attrs) / sync_currents() / sync_particles()   (core/patch/patch.py:670-764)
    sim.sorter[ispec]()                                     ParticleSort2D    (core/sort/particle_sort.py:196-211)
    sim.current_depositor.reset() / (ispec, dt)             CurrentDeposition2D (core/current/deposition.py:138-208)
    sim.pusher[ispec](dt, unified=True) / .push_position(dt)  BorisPusher     (core/pusher/pusher.py:102-141)
    sim.interpolator(ispec)                                 FieldInterpolation2D (core/interpolation/field_interpolation.py:149-180)

The device owns the state; ``sim.patches[i].fields.*`` / ``.particles[ispec].*`` are host MIRRORS in
λPIC's layout.  They are refreshed (device -> host) before a stage that has a triggered callback and
written back (host -> device) after it, so reference-style callbacks run unchanged.  CPML layers,
laser injection (``lambdapic_amd.laser``) and the moving window are device native.  Out of scope here
(SURVEY.md section 2): QED, collisions, load balancing, I/O.  The 3-D twin is ``simulation3d.py``.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch

from . import constants
from .dist import SlabComm
from .engine import PicEngine2D
from .patch import make_patches_2d


@dataclass
class Species:
    """subset of the reference's ``Species`` (`core/species.py:50-182`) the hot path needs"""
    name: str
    charge: float            # in units of e
    mass: float              # in units of m_e
    density: object = None   # callable (x, y[, z]) -> n [m^-3] or a float
    ppc: object = 0          # particles per cell: an int or a callable (x, y[, z]) -> int
    momentum_sigma: float = 0.0   # thermal u = gamma*beta spread per axis (SetTemperature stand-in)
    density_min: float = 0.0      # only cells with density > density_min are loaded (core/patch/cpu.py:16,41)
    ispec: int = field(default=-1, init=False)

    @property
    def q(self):
        return self.charge * constants.E_CHARGE

    @property
    def m(self):
        return self.mass * constants.M_E


@dataclass
class Electron(Species):
    """`core/species.py:185-208` without the radiation / photon links (QED is out of scope)"""
    name: str = "electron"
    charge: float = field(default=-1, init=False)
    mass: float = field(default=1.0, init=False)


@dataclass
class Positron(Species):
    name: str = "positron"
    charge: float = field(default=1, init=False)
    mass: float = field(default=1.0, init=False)


@dataclass
class Proton(Species):
    name: str = "proton"
    charge: float = field(default=1, init=False)
    mass: float = field(default=constants.M_P / constants.M_E, init=False)      # core/species.py:219


def load_block_device(species, origin, n, d, seed, device, id_prefix=0):
    """Uniform loading of one block of cells ON THE DEVICE: ``ppc`` particles (a number, or a callable evaluated per
    cell like the density) in every cell whose density is > ``density_min``, positions uniform inside the cell, weight
    ``n d^dim / ppc``, thermal momenta N(0, sigma) (`core/patch/cpu.py:7-18,21-45`).  The density callable is user code written for numpy: it is evaluated on
    the host on the CELL grid (small), everything per particle happens on the device (the reference loads on
    the host; at 16 ppc a recycled window column is 5 x 10^5 particles, ~0.1 s of numpy per shift).
    Returns a dict of device tensors x, y(, z), ux, uy, uz, inv_gamma, w, id (may be empty: None).
    The generator is seeded from ``seed`` (a list of ints): the loading of a block is a function of its
    origin, not of the decomposition or of the moment it enters a moving window."""
    dim = len(n)
    if not species.ppc or species.density is None:
        return None
    axes = [o + np.arange(m) * dd for o, m, dd in zip(origin, n, d)]
    grids = np.meshgrid(*axes, indexing="ij")
    dens = species.density(*grids) if callable(species.density) else np.full(grids[0].shape, float(species.density))
    dens = np.broadcast_to(np.asarray(dens, dtype=np.float64), grids[0].shape)
    if callable(species.ppc):          # per cell, truncated like the reference's int(ppc_func(...))
        ppc_cells = np.broadcast_to(np.asarray(species.ppc(*grids)), grids[0].shape).astype(np.int64).ravel()
    else:
        ppc_cells = np.full(dens.size, int(species.ppc), dtype=np.int64)
    sel = np.nonzero((np.ravel(dens) > float(species.density_min)) & (ppc_cells > 0))[0]
    if sel.size == 0:
        return None
    ppc = torch.from_numpy(ppc_cells[sel]).to(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(np.random.SeedSequence([int(v) & 0xFFFFFFFF for v in seed]).generate_state(1, np.uint64)[0] >> 1)
                    if seed is not None else torch.seed())
    out = {}
    k = int(ppc_cells[sel].sum())
    for a, g_, dd in zip("xyz", grids, d):
        c = torch.from_numpy(np.ascontiguousarray(np.ravel(g_)[sel])).to(device).repeat_interleave(ppc)
        out[a] = c + (torch.rand(k, dtype=torch.float64, device=device, generator=gen) - 0.5) * dd
    vol = float(np.prod(d))
    out["w"] = torch.from_numpy(np.ascontiguousarray(np.ravel(dens)[sel] * vol / ppc_cells[sel])).to(device) \
        .repeat_interleave(ppc)
    if species.momentum_sigma:
        for a in ("ux", "uy", "uz"):
            out[a] = torch.randn(k, dtype=torch.float64, device=device, generator=gen) * float(species.momentum_sigma)
        out["inv_gamma"] = 1.0 / torch.sqrt(1 + out["ux"] ** 2 + out["uy"] ** 2 + out["uz"] ** 2)
    else:
        for a in ("ux", "uy", "uz"):
            out[a] = torch.zeros(k, dtype=torch.float64, device=device)
        out["inv_gamma"] = torch.ones(k, dtype=torch.float64, device=device)
    out["id"] = torch.arange(k, dtype=torch.int64, device=device) + int(id_prefix)
    return out


def validate_interval(interval):
    """`callback/callback.py:11-19`: an int >= 1 (steps), a float in (0, 1) (seconds) or a callable(sim) -> bool"""
    if isinstance(interval, bool) or not (isinstance(interval, (int, float, np.integer)) or callable(interval)):
        raise TypeError(f"Invalid interval: {interval}. Must be int, float, or Callable")
    if isinstance(interval, float) and not 0 < interval < 1:
        raise ValueError(f"Invalid interval: {interval}. Must be between 0 and 1s if it is a float")
    if isinstance(interval, (int, np.integer)) and interval < 1:
        raise ValueError(f"Invalid interval: {interval}. Must be greater than 0 if it is an integer")


def interval_triggered(sim, interval) -> bool:
    """`callback/callback.py:22-45`"""
    if callable(interval):
        return bool(interval(sim))
    if isinstance(interval, float):
        return (sim.time % interval) < sim.dt
    return sim.itime % int(interval) == 0


def callback(stage="end", interval=1):
    """decorator mirroring `callback/callback.py:48-109`: attaches ``stage`` and ``interval`` (the stage loop asks
    ``interval`` before it calls -- and before it refreshes the host mirrors for the call)"""
    validate_interval(interval)

    def wrap(fn):
        fn.stage, fn.interval = stage, interval
        return fn
    return wrap


class Callback:
    """base class of the reference's class-style callbacks (`callback/callback.py:111-145`): subclasses set ``stage`` /
    ``interval`` and define ``_call(sim)``; calling the object runs ``_call`` when the interval says so"""
    stage = "end"
    interval = 1

    def __call__(self, sim):
        validate_interval(self.interval)
        if not interval_triggered(sim, self.interval):
            return None
        ret = self._call(sim)
        if sim.mpi.size > 1:
            sim.mpi.comm.Barrier()
        return ret

    def _call(self, sim):
        raise NotImplementedError


class _Facade:
    def __init__(self, sim):
        self.sim = sim
        self._enabled = True

    # EnableMixin (`core/utils/enable_mixin.py:4-38`)
    def enable(self):
        self._enabled = True

    def disable(self):
        self._enabled = False

    def is_enabled(self):
        return self._enabled


class MaxwellSolver2D(_Facade):
    def update_efield(self, dt):
        if self._enabled:
            self.sim.engine.update_efield(dt)

    def update_bfield(self, dt):
        if self._enabled:
            self.sim.engine.update_bfield(dt)


class CurrentDeposition2D(_Facade):
    def reset(self):
        if self._enabled:
            self.sim.engine.reset_current()

    def __call__(self, ispec, dt):
        if self._enabled:
            self.sim.engine.deposit(ispec, dt)


class FieldInterpolation2D(_Facade):
    def __call__(self, ispec):
        if self._enabled:
            self.sim.engine.interpolate(ispec)


class BorisPusher(_Facade):
    def __init__(self, sim, ispec):
        super().__init__(sim)
        self.ispec = ispec

    def __call__(self, dt, unified=False):
        if not self._enabled:
            return
        if unified:
            self.sim.engine.push_deposit(self.ispec, dt)
        else:
            self.sim.engine.boris(self.ispec, dt)

    def push_position(self, dt):
        if self._enabled:
            self.sim.engine.push_position(self.ispec, dt)


class ParticleSort2D(_Facade):
    """the reference re-sorts every step (cheap there: in-place, nothing moves when already
    sorted); the device sort is out of place, so it runs when the engine's cadence says so"""

    def __init__(self, sim, ispec):
        super().__init__(sim)
        self.ispec = ispec
        self.nbuf_last = 0

    def __call__(self, force=False):
        if not self._enabled:
            return 0
        sp = self.sim.engine.species[self.ispec]
        if force or self.sim.engine.sort_due(sp):
            self.sim.engine.sort(self.ispec)
            self.nbuf_last = sp.n_sorted
        else:
            self.nbuf_last = 0
        return self.nbuf_last


class DevicePatches:
    """``sim.patches``: list-like over the host mirrors + the sync entry points of ``Patches``"""

    def __init__(self, sim, mirrors):
        self.sim, self._m = sim, mirrors
        self.species = sim.species

    def __getitem__(self, i):
        return self._m[i]

    def __len__(self):
        return len(self._m)

    def __iter__(self):
        return iter(self._m)

    def __getattr__(self, name):       # nx, ny, dx, dy, n_guard, npatches, xmin_global, ...
        if name.startswith("_"):       # (un)pickling probes dunder / private names before _m exists
            raise AttributeError(name)
        return getattr(self._m, name)

    def sync_guard_fields(self, attrs=("ex", "ey", "ez", "bx", "by", "bz")):
        self.sim.engine.sync_guard_fields(attrs)

    def sync_currents(self):
        self.sim.engine.sync_currents()

    def sync_particles(self):
        for i in range(len(self.sim.species)):
            self.sim.engine.sync_particles(i)


class _CommFacade:
    """the handful of ``mpi4py`` communicator calls the reference's callbacks use (`sim.mpi.comm.gather`,
    ``.Barrier``, ``.bcast``, ``.allreduce``, ``.Get_rank`` ...; `callback/utils.py:80,121`, `callback/hdf5.py`)
    on top of ``torch.distributed`` object collectives -- control plane only, never inside the step"""

    def __init__(self, comm):
        self._c = comm

    def Get_rank(self):
        return self._c.rank

    def Get_size(self):
        return self._c.size

    def Barrier(self):
        self._c.barrier()

    barrier = Barrier

    def gather(self, obj, root=0):
        if self._c.size == 1:
            return [obj]
        import torch.distributed as dist
        out = [None] * self._c.size if self._c.rank == root else None
        dist.gather_object(obj, out, dst=root, group=self._c.group)
        return out

    def bcast(self, obj, root=0):
        if self._c.size == 1:
            return obj
        import torch.distributed as dist
        box = [obj]
        dist.broadcast_object_list(box, src=root, group=self._c.group)
        return box[0]

    def allgather(self, obj):
        if self._c.size == 1:
            return [obj]
        import torch.distributed as dist
        out = [None] * self._c.size
        dist.all_gather_object(out, obj, group=self._c.group)
        return out

    def allreduce(self, value):
        """sum of a python / numpy scalar or array over the ranks"""
        parts = self.allgather(value)
        total = parts[0]
        for v in parts[1:]:
            total = total + v
        return total

    def reduce(self, value, root=0):
        parts = self.gather(value, root)
        if parts is None:
            return None
        total = parts[0]
        for v in parts[1:]:
            total = total + v
        return total


class MPIFacade:
    """``sim.mpi`` as callbacks see it (`core/mpi/mpi_manager.py:23-298`): ``rank``, ``size``, ``comm`` and the
    split ``sync_*_start`` / ``_wait`` brackets.  On the device slab the exchange is issued by ``_start`` (it is
    asynchronous on the stream anyway) and ``_wait`` has nothing left to do; the handle is ``None`` with one
    rank, like the reference's (`mpi_manager.py:111-195`)."""

    def __init__(self, sim):
        self._sim = sim
        self.comm = _CommFacade(sim.comm)

    @property
    def rank(self):
        return self._sim.comm.rank

    @property
    def size(self):
        return self._sim.comm.size

    def sync_guard_fields(self, attrs=("ex", "ey", "ez", "bx", "by", "bz")):
        self._sim.patches.sync_guard_fields(list(attrs))

    def sync_guard_fields_start(self, attrs=("ex", "ey", "ez", "bx", "by", "bz")):
        if self.size == 1:
            return None
        self.sync_guard_fields(attrs)
        return ("guard_fields", tuple(attrs))

    def sync_guard_fields_wait(self, handle):
        return None

    def sync_currents(self):
        self._sim.patches.sync_currents()

    def sync_currents_start(self):
        if self.size == 1:
            return None
        self.sync_currents()
        return ("currents",)

    def sync_currents_wait(self, handle):
        return None

    def sync_particles(self, ispec=None):
        if ispec is None:
            self._sim.patches.sync_particles()
        else:
            self._sim.engine.sync_particles(ispec)

    def sync_particles_start(self, ispec=None):
        if self.size == 1:
            return None
        self.sync_particles(ispec)
        return ("particles", ispec)

    def sync_particles_wait(self, handle):
        return None


def _NO_STOP():
    """the default ``stop_callback`` of ``run`` (`simulation/simulation.py:858`: ``lambda: False``)"""
    return False


class MovingWindow:
    """Mirror of the reference's ``MovingWindow`` callback (`callback/utils.py:471-648`): stage
    ``start``; once ``sim.time >= start_time`` (default Lx / c) the x layers are removed and the
    window advances by ``velocity * dt`` per step; every time a whole patch width has accumulated
    the leftmost patch column is recycled to the right end (fields and psi zeroed, its particles
    replaced by a fresh loading of the species' density profiles); a negative velocity moves the window
    backwards -- the rightmost column is recycled to the left end (`callback/utils.py:570-573,622-648`)."""
    stage = "start"
    interval = 1
    device_native = True

    def __init__(self, velocity, start_time=None, inject_particles=True, stop_inject_time=None):
        self.velocity, self.start_time = velocity, start_time
        self.inject_particles, self.stop_inject_time = inject_particles, stop_inject_time
        self.total_shift = self.patch_this_shift = None
        self.num_shifts = 0

    def fields_untouched(self, sim):
        """will the NEXT call (stage 'start' of the step after this one) leave the field arrays and the layers alone?
        (Simulation.run: the second E half step of this step may then be done together with the next step's first)"""
        t = sim.time + sim.dt
        start = sim.Lx / constants.C_LIGHT if self.start_time is None else self.start_time
        if t < start:
            return True
        if self.num_shifts == 0:
            return False               # its first active call removes the x layers
        patch_Lx = sim.nx_per_patch * sim.dx
        v = self.velocity(t) if callable(self.velocity) else self.velocity
        return abs(self.patch_this_shift + v * sim.dt) < patch_Lx

    def __call__(self, sim):
        patch_Lx = sim.nx_per_patch * sim.dx
        if self.start_time is None:
            self.start_time = sim.Lx / constants.C_LIGHT
        if self.total_shift is None:
            self.total_shift = self.patch_this_shift = patch_Lx
        if sim.time < self.start_time:
            return
        if self.num_shifts == 0:
            sim.engine.remove_x_pml()
        v = self.velocity(sim.time) if callable(self.velocity) else self.velocity
        self.total_shift += v * sim.dt
        self.patch_this_shift += v * sim.dt
        self.num_shifts += 1
        inject = self.inject_particles and (self.stop_inject_time is None or sim.time < self.stop_inject_time)
        if self.patch_this_shift >= patch_Lx:            # callback/utils.py:567-572
            self.patch_this_shift -= patch_Lx
            sim.shift_window_right(inject)
        elif self.patch_this_shift <= -patch_Lx:         # a window moving backwards: the rightmost column is recycled
            self.patch_this_shift += patch_Lx
            sim.shift_window_left(inject)


class Simulation:
    """2-D periodic simulation driver with the reference's constructor vocabulary
    (`simulation/simulation.py:118-168`) and stage list (`:170-184`)."""

    STAGES = ["init", "start", "maxwell_1", "_push_position_1", "_interpolator", "_qed", "_push_momentum",
              "_push_position_2", "current_deposition", "qed_create_particles", "_laser", "maxwell_2", "end",
              "final"]
    DEFAULT_STAGE = "end"
    _PUSHER_STAGES = {"_push_position_1", "_interpolator", "_qed", "_push_momentum", "_push_position_2"}

    def __init__(self, nx, ny, dx, dy, npatch_x=1, npatch_y=1, nsteps=None, sim_time=None, dt_cfl=0.95,
                 n_guard=3, boundary_conditions=None, cpml_thickness=6, random_seed=None, device="cuda:0",
                 comm=None, sort_interval=32, capacity_factor=1.5):
        # reference default: PML on all four sides (simulation.py:157-162)
        bc = dict(boundary_conditions or {k: "pml" for k in ("xmin", "xmax", "ymin", "ymax")})
        if dt_cfl > 1.0:
            raise ValueError("dt_cfl must be <= 1")
        self.cpml_thickness = int(cpml_thickness)
        self.comm = comm or SlabComm(None, periodic=bc["xmin"] == "periodic")
        self.mpi = MPIFacade(self)          # what callbacks know as sim.mpi (rank, size, comm, sync_*_start/_wait)
        self.nx, self.ny, self.dx, self.dy = int(nx), int(ny), float(dx), float(dy)
        if self.nx % self.comm.size or (self.nx // self.comm.size) % npatch_x or self.ny % npatch_y:
            raise ValueError("nx must split evenly over ranks and patches")
        self.npatch_x, self.npatch_y, self.n_guard = npatch_x, npatch_y, n_guard
        self.dt = dt_cfl * (self.dx ** -2 + self.dy ** -2) ** -0.5 / constants.C_LIGHT   # simulation.py:219
        self.Lx, self.Ly = self.nx * self.dx, self.ny * self.dy
        self.nsteps, self.sim_time = nsteps, sim_time
        self.boundary_conditions = bc
        self.random_seed = random_seed
        self.device, self.sort_interval, self.capacity_factor = device, sort_interval, capacity_factor
        self.species: list[Species] = []
        self.itime, self.time, self.ispec = 0, 0.0, None
        self.initialized = False
        self.current_synced = False
        self.stages = list(self.STAGES)
        self.dimension = 2

    def add_species(self, species):
        for s in species if isinstance(species, (list, tuple)) else [species]:
            s.ispec = len(self.species)
            self.species.append(s)

    # ---- initialisation (`simulation.py:284-423` for the in-scope parts) ----------------------------
    def initialize(self):
        nx_loc = self.nx // self.comm.size
        self.nx_per_patch, self.ny_per_patch = nx_loc // self.npatch_x, self.ny // self.npatch_y
        self.engine = PicEngine2D(nx_loc, self.ny, self.dx, self.dy, self.n_guard, self.device, self.comm,
                                  sort_interval=self.sort_interval, boundary_conditions=self.boundary_conditions,
                                  cpml_thickness=self.cpml_thickness)
        mirrors = make_patches_2d(nx_loc, self.ny, self.dx, self.dy, self.npatch_x, self.npatch_y, self.n_guard,
                                  boundary_conditions=self.boundary_conditions, nspecies=len(self.species))
        for p in mirrors:                       # patch origins in global coordinates
            p.x0 += self.engine.x0
            p.fields.x0 = p.x0
            p.fields.xaxis += self.engine.x0
        mirrors.xmin_global, mirrors.xmax_global = -self.dx / 2, self.Lx - self.dx / 2
        self.patches = DevicePatches(self, mirrors)
        for s in self.species:
            blocks = []
            for p in mirrors:
                # one generator per (species, patch origin): the same particles whatever the number
                # of ranks (the reference spawns one generator per rank, simulation.py:700-716, so ITS
                # loading depends on the decomposition; identical physics, different noise)
                b = load_block_device(s, (p.x0, p.y0), (p.nx, p.ny), (self.dx, self.dy), self._seed(s, p.x0, p.y0),
                                      self.device, id_prefix=self._next_ids(s.ispec))
                if b is not None:
                    self._id_next[s.ispec] += b["x"].numel()
                    blocks.append(b)
                p.particles[s.ispec].initialize(0)     # the mirrors fill at the first download()
            n_tot = sum(b["x"].numel() for b in blocks)
            self.engine.add_species(s.q, s.m, capacity=int(n_tot * self.capacity_factor) + 65536, with_eb=True)
            sp = self.engine.species[s.ispec]
            o = 0
            for b in blocks:
                k = b["x"].numel()
                for a in sp.cset.names:
                    if a in b:
                        sp.cset.arr(a)[o:o + k] = b[a]
                sp.cset.id[o:o + k] = b["id"]
                o += k
            sp.n, sp.n_sorted, sp.tiling = n_tot, 0, None
        self.maxwell = MaxwellSolver2D(self)
        self.interpolator = FieldInterpolation2D(self)
        self.current_depositor = CurrentDeposition2D(self)
        self.pusher = [BorisPusher(self, i) for i in range(len(self.species))]
        self.sorter = [ParticleSort2D(self, i) for i in range(len(self.species))]
        self.initialized = True

    def _next_ids(self, ispec):
        """first id of the next block this rank creates for species ``ispec``: rank in the bits above 50 (the
        reference's layout, `core/particles.py:91-116`: rank << 50 | ipatch << 32 | serial), below it ONE
        running count per rank and species -- initial loading and every window injection draw from it, so
        ids cannot collide however large a block or however many shifts"""
        if not hasattr(self, "_id_next"):
            self._id_next = {}
        n = self._id_next.setdefault(ispec, 0)
        if n >= 1 << 50:
            raise OverflowError("particle id counter exceeds 50 bits")
        return (self.comm.rank << 50) | n

    def _seed(self, s, x0, y0):
        return None if self.random_seed is None else \
            [self.random_seed, s.ispec, int(round(x0 / self.dx)), int(round(y0 / self.dy))]

    def shift_window_left(self, inject):
        """recycle the rightmost patch column (`callback/utils.py:622-648`): the window moves one patch width to -x"""
        self.shift_window_right(inject, direction=-1)

    def shift_window_right(self, inject, direction=1):
        """recycle the leftmost patch column (`callback/utils.py:594-620`) on the device slab"""
        eng, n = self.engine, self.nx_per_patch
        eng.shift_window(direction * n)
        self.window_shifts = getattr(self, "window_shifts", 0) + 1
        for p in self.patches:              # the mirrors follow the window
            p.x0 += direction * n * self.dx
            p.fields.x0 = p.x0
            p.fields.xaxis += direction * n * self.dx
            p.xaxis = p.xaxis + direction * n * self.dx
        self.patches._m.xmin_global += direction * n * self.dx
        self.patches._m.xmax_global += direction * n * self.dx
        if not inject or self.comm.rank != (self.comm.size - 1 if direction > 0 else 0):
            return                      # only the leading slab's new columns are new ground
        x_new = eng.x0 + (eng.nx - n) * self.dx if direction > 0 else eng.x0
        for s in self.species:
            for j in range(self.npatch_y):
                # same seed rule as initialize(): the loading of a column is a function of its origin,
                # so a moving window reproduces what a long static box would have held there
                y0 = j * self.ny_per_patch * self.dy
                b = load_block_device(s, (x_new, y0), (n, self.ny_per_patch), (self.dx, self.dy),
                                      self._seed(s, x_new, y0), self.device, id_prefix=self._next_ids(s.ispec))
                if b is not None:
                    self._id_next[s.ispec] += b["x"].numel()
                    eng.append_particles_device(s.ispec, b)

    # ---- host mirrors <-> device --------------------------------------------------------------------
    def download(self):
        """device -> ``sim.patches`` mirrors (fields with guards, live particles binned by patch)"""
        self.engine.grid.download_patches(list(self.patches))
        nxp, nyp = self.nx_per_patch, self.ny_per_patch
        for s in self.species:
            d = self.engine.species[s.ispec].download()
            i = np.clip(np.floor((d["x"] - self.engine.x0) / self.dx + 0.5).astype(int) // nxp, 0, self.npatch_x - 1)
            j = np.clip(np.floor(d["y"] / self.dy + 0.5).astype(int) // nyp, 0, self.npatch_y - 1)
            owner = i + j * self.npatch_x
            for k, p in enumerate(self.patches):
                sel = owner == k
                q = p.particles[s.ispec]
                q.initialize(int(sel.sum()))
                for a in d:
                    if hasattr(q, a):
                        getattr(q, a)[:] = d[a][sel]

    def upload(self):
        """``sim.patches`` mirrors -> device (after a callback modified them)"""
        self.engine.grid.upload_patches(list(self.patches), self.npatch_x, self.npatch_y)
        for s in self.species:
            self.engine.species[s.ispec].upload([p.particles[s.ispec] for p in self.patches])

    # ---- the stage loop (`simulation.py:937-1130`) ----------------------------------------------------
    def _triggered(self, cbs):
        """the callbacks of ``cbs`` that fire in this step.  An interval FUNCTION is asked once per step and callback (the
        reference evaluates it once, at the callback's stage): the stage loop asks several times, a stateful predicate
        must not see that"""
        cache = self.__dict__.setdefault("_trig_cache", {})
        if cache.get("step") != (self.itime, id(self)):
            cache.clear()
            cache["step"] = (self.itime, id(self))
        out = []
        for cb in cbs:
            iv = getattr(cb, "interval", 1)
            if callable(iv):
                if id(cb) not in cache:
                    cache[id(cb)] = bool(interval_triggered(self, iv))
                hit = cache[id(cb)]
            else:
                hit = interval_triggered(self, iv)
            if hit:
                out.append(cb)
        return out

    def _run_stage(self, table, stage):
        cbs = self._triggered(table.get(stage, []))
        if not cbs:
            return
        host_cbs = [cb for cb in cbs if not getattr(cb, "device_native", False)]
        if host_cbs:
            self.download()
        for cb in cbs:
            cb(self)
        if host_cbs:
            self.upload()

    def _host_callback_near(self, host_cbs, last_step):
        """will a mirror-reading callback run before the NEXT push overwrites ex_part ... bz_part -- in this step, or in
        the next one's stages before its push?  (An interval FUNCTION cannot be asked about the next step: taken as yes.)"""
        for cb in host_cbs:
            iv = getattr(cb, "interval", 1)
            if callable(iv) or interval_triggered(self, iv) or (last_step and getattr(cb, "stage", "") == "final"):
                return True
            if isinstance(iv, float):
                if ((self.time + self.dt) % iv) < self.dt:
                    return True
            elif (self.itime + 1) % int(iv) == 0:
                return True
        return False

    def update_lists(self):
        """the reference re-points its facades at the (possibly re-allocated) per-patch arrays
        (`simulation/simulation.py:781-824`; called by RestartDump.load).  The facades here hold no array
        pointers -- every call reads the engine's current stores -- so there is nothing to re-point."""

    _INNER_STAGES = ("maxwell_1", "current_deposition", "qed_create_particles")

    def _can_defer_e2(self, table, last_step, stop_default):
        """may this step leave its second E half step to the next step's first (engines: ``run_steps``)?  Only when nothing
        can read or move E in between: no callback at 'maxwell_2' / 'end' of this step, the default ``stop_callback``, and at
        'start' of the next step only callbacks that say they will leave the fields alone (``fields_untouched``);
        interval FUNCTIONS cannot be asked about the next step: they block.  The next step need not be a fused one: the
        per-stage facades complete a pending half step before they touch the fields (``engine._flush_e2``)."""
        if last_step or not stop_default or not self.defer_e2:
            return False
        if any(self._triggered(table.get(st, [])) for st in ("maxwell_2", "end")):
            return False
        for cb in table.get("start", []):
            iv = getattr(cb, "interval", 1)
            if callable(iv):
                return False
            nxt = ((self.time + self.dt) % iv) < self.dt if isinstance(iv, float) else (self.itime + 1) % int(iv) == 0
            if nxt and not (hasattr(cb, "fields_untouched") and cb.fields_untouched(self)):
                return False
        return True

    defer_e2 = True        # Simulation.run may merge E half steps across the step boundary (see _can_defer_e2)

    def _fused_step(self, table, unified=True, defer_e2=False):
        """When no callback is triggered between 'start' and '_laser' / 'maxwell_2' (the predicate the reference uses to
        skip work around callbacks, ``has_triggered_callbacks``, `simulation/simulation.py:1493-1508`) and every facade is
        enabled, the whole stage sequence of this step is enqueued by ONE engine call (``lpa_step``, step.py) -- two when a
        '_laser' callback injects in between.  False: the caller walks the stages one facade call at a time."""
        eng = self.engine
        segments = eng.can_fuse() and eng.comm.size > 1 and not eng.one_call_step() and not eng.overlap
        if not (unified and (eng.one_call_step() or segments)):
            return False
        facades = [self.maxwell, *self.pusher, *self.sorter] + \
            ([self.current_depositor] if hasattr(self, "current_depositor") else [])
        if not all(f._enabled for f in facades):
            return False
        if any(self._triggered(table.get(st, [])) for st in self._INNER_STAGES):
            return False
        from . import _lib
        lasers = bool(self._triggered(table.get("_laser", [])))
        if segments:      # slab ranks whose faces travel through torch.distributed: sub-ranges between the exchanges
            eng._step_segments(self.dt, laser=(lambda e, dt: self._run_stage(table, "_laser")) if lasers else None,
                               defer_e2=defer_e2)
            self.current_synced, self.ispec = True, None
            return True
        eng.step_stages(self.dt, _lib.LPA_STAGE_E1, _lib.LPA_STAGE_B2 if lasers else _lib.LPA_STAGE_E2, defer_e2 and not lasers)
        self.current_synced, self.ispec = True, None
        if lasers:
            self._run_stage(table, "_laser")
            eng.step_stages(self.dt, _lib.LPA_STAGE_B2_GUARD, _lib.LPA_STAGE_E2, defer_e2)
        return True

    def sync_currents(self):
        if not self.current_synced:
            self.patches.sync_currents()
            self.current_synced = True

    def maxwell_stage(self):
        """`simulation.py:743-761`"""
        self.maxwell.update_efield(0.5 * self.dt)
        self.patches.sync_guard_fields(["ex", "ey", "ez"])
        self.maxwell.update_bfield(0.5 * self.dt)
        self.patches.sync_guard_fields(["bx", "by", "bz"])

    def run(self, nsteps=None, sim_time=None, callbacks=None, stop_callback=None):
        if nsteps is not None and sim_time is not None:
            raise ValueError("Cannot specify both nsteps and sim_time in run() method")
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
        if nsteps is None:
            nsteps = int(sim_time / self.dt) if sim_time is not None else \
                (self.nsteps if self.nsteps is not None else int(self.sim_time / self.dt))
        # host callbacks may read ex_part..bz_part: the pushes that precede one write them (decided step by step below;
        # registering a diagnostic that runs every 100 steps must not cost six more attribute streams in every step)
        # (+ the device-native writers that store the per-particle fields: SaveParticlesToHDF5 with every attribute,
        # RestartDump)
        host_cbs = [cb for cb in callbacks or [] if not getattr(cb, "device_native", False) or
                    getattr(cb, "reads_part_eb", False)]
        self.engine.write_part_eb = bool(host_cbs)
        unified = not (self._PUSHER_STAGES & {s for s, c in table.items() if c})   # :896-911
        # rho between two sorts comes from the continuity equation (rho.py) unless the split path deposits with the
        # standalone kernel, or -- decided step by step below -- a callback reads per-species rho between the deposits
        self.engine.rho_continuity_blocked = not unified
        self.engine.sort_part_eb = not unified      # (pusher-stage callbacks may look at ex_part between sort and interpolation)
        self._run_stage(table, "init")
        # a RestartDump among the callbacks may ask for a last dump (signal): simulation.py:889-894
        restart_cb = next((cb for cb in callbacks or [] if cb.__class__.__name__ == "RestartDump"), None)
        E, B = ["ex", "ey", "ez"], ["bx", "by", "bz"]
        self.itime_end = self.itime + nsteps
        for self.istep in range(self.itime, self.itime_end):
            self.engine._dt_hint = self.dt      # (rho.py, the engines' first sort: the step's dt before any push)
            # a step in which a 'current_deposition' callback runs deposits rho for real (it reads per-species rho); the
            # steps in between carry rho on from there (a density diagnostic every 100 steps costs one real deposit)
            self.engine.rho_continuity_blocked = not unified or bool(self._triggered(table.get("current_deposition", [])))
            self.engine.write_part_eb = self._host_callback_near(host_cbs, self.istep == self.itime_end - 1)
            self._run_stage(table, "start")
            defer = self._can_defer_e2(table, self.istep == self.itime_end - 1, stop_callback is _NO_STOP) and \
                not (restart_cb is not None and restart_cb._dump_requested)
            if self._fused_step(table, unified, defer):
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
                self.ispec = ispec
                self.sorter[ispec]()
            self.current_depositor.reset()
            self.current_synced = False
            # no callback between the species' deposits: push the edge tiles first and hide the J / rho
            # exchange behind the interior (the reference's sync_currents_start ... _wait bracket)
            fused_all = unified and not table.get("current_deposition") and self.engine.overlap and \
                all(p._enabled for p in self.pusher) and self.engine.push_deposit_overlapped(self.dt)
            if fused_all:
                self.current_synced = True
            for ispec in range(len(self.species) if not fused_all else 0):
                self.ispec = ispec
                if unified:
                    self.pusher[ispec](self.dt, unified=True)
                    self.current_synced = False      # simulation.py:991: every deposit un-syncs the currents
                else:
                    self.pusher[ispec].push_position(0.5 * self.dt)
                    self._run_stage(table, "_push_position_1")
                    self.interpolator(ispec)
                    self._run_stage(table, "_interpolator")
                    self._run_stage(table, "_qed")
                    self.pusher[ispec](self.dt)
                    self._run_stage(table, "_push_momentum")
                    self.pusher[ispec].push_position(0.5 * self.dt)
                    self._run_stage(table, "_push_position_2")
                    self.current_depositor(ispec, self.dt)
                    self.current_synced = False      # simulation.py:1038
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
