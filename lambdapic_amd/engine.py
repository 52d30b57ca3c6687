"""PicEngine2D -- the device-resident PIC inner loop of one rank (one MI355X).

Holds the rank's field slab and particle stores in HBM and exposes the operations the reference's
facades perform on its patch lists, one call per reference call:

    reference (per step, simulation/simulation.py:937-1122)      engine
    ---------------------------------------------------------    ---------------------------
    maxwell.update_efield(dt) / update_bfield(dt)                update_efield / update_bfield
    patches.sync_guard_fields + mpi.sync_guard_fields_*          sync_guard_fields
    sorter[ispec]()                                              sort
    current_depositor.reset()                                    reset_current
    pusher[ispec](dt, unified=True)                              push_deposit
    patches.sync_currents + mpi.sync_currents_*                  sync_currents
    mpi.sync_particles_* + patches.sync_particles                sync_particles

Multi-GPU: the domain is split into 1-D slabs along x, one process per GPU (torch.distributed,
backend nccl == RCCL); x faces are exchanged with the two ring (periodic x) or chain (open x) neighbours
(``SlabComm``); the J / rho exchange runs on a second stream behind the interior tiles
(``push_deposit_overlapped``).  Open faces carry CPML layers (``DevicePML2D``), absorb particles and
take the laser (``laser_inject``); ``shift_window`` moves the slab chain (MovingWindow).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib, constants
from ._lib import LPA_MIG_NATTR, check, lib
from .device import DeviceGrid2D, DeviceParticles, current_stream_ptr, restore_device, to_host
from .dist import MigrateWindowMixin, SlabComm, exchange_faces
from .rho import RhoContinuityMixin
from .step import FusedStepMixin


def psi_ptr(layer, key):
    """device address of a CPML layer's psi array as the kernels index it: row 0 = the slab's x node 0.  The arrays of
    the y / z layers carry ``xpad`` extra x rows in front (and behind): a slab with neighbours advances B -- and with it the
    B psi of those layers -- on its x guard planes too (lpa_step, LPA_STEP_B_EXT_*)"""
    t = layer[key]
    return t.data_ptr() + 8 * layer.get("xpad", 0) * layer.get("row", 0)


def psi_rows(layer, key, guards=False):
    """psi array of a y / z layer as [x rows][...]: the nx rows of the slab's nodes, or (``guards``) with the xpad guard rows"""
    pad = layer.get("xpad", 0)
    v = layer[key].view(-1, layer["row"])
    return v if guards or not pad else v[pad:v.shape[0] - pad]


class DevicePML2D:
    """CPML coefficients and psi arrays of one rank's slab (reference: per-patch ``PML`` objects,
    `core/boundary/cpml.py:23-340`; slab mapping as in oracle/cpml.py).  Host builds the per-axis
    kappa / sigma / a profiles; bcoeff / ccoeff_d (`cpml.py:537-538`) are cached per dt.  psi arrays are compact: x layers
    [layer][ny], y layers [xpad + nx + xpad][layer] (``psi_ptr`` / ``psi_rows``)."""

    def __init__(self, nx, ny, dx, dy, sides, thickness, device, kappa_max=20.0, a_max=0.15, sigma_max=0.7, xpad=0):
        self.nx, self.ny, self.dx, self.dy, self.t = nx, ny, dx, dy, int(thickness)
        self.sides, self.device, self.xpad = set(sides), device, int(xpad)
        m, ma = 3, 1
        smax = sigma_max * constants.C_LIGHT * 0.8 * (m + 1.0) / dx      # cpml.py:60 (dx for every axis)
        self.host = {}
        for ax, n in (("x", nx), ("y", ny)):
            for fld in ("e", "b"):
                self.host[fld + ax] = dict(kappa=np.ones(n), sigma=np.zeros(n), a=np.zeros(n))

            def fill(fld, pos, sl, ax=ax):
                c = self.host[fld + ax]
                c["kappa"][sl] = 1 + (kappa_max - 1) * pos ** m           # cpml.py:119-125
                c["sigma"][sl] = smax * pos ** m
                c["a"][sl] = a_max * (1 - pos) ** ma

            ar = np.arange(self.t, dtype=float)
            if ax + "min" in self.sides:                                   # cpml.py:233-250,271-287
                fill("e", 1.0 - ar / self.t, np.s_[:self.t])
                fill("b", 1.0 - (ar + 0.5) / self.t, np.s_[:self.t])
            if ax + "max" in self.sides:                                   # cpml.py:253-269,289-305
                fill("e", 1.0 - ar[::-1] / self.t, np.s_[n - self.t:n])
                fill("b", 1.0 - (ar + 0.5)[::-1] / self.t, np.s_[n - self.t - 1:n - 1])
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.kappa = {k: dev(v["kappa"]) for k, v in self.host.items()}
        # layers: (efield?, axis, start, stop, psi_a, psi_b); x layers first, then y (cpml.py order of
        # pml_boundary: simulation.py:455-463)
        self.layers = []
        for fld in ("e", "b"):
            for axis, ax, n, nt in ((0, "x", nx, ny), (1, "y", ny, nx + 2 * self.xpad)):
                rng = []
                if ax + "min" in self.sides:
                    rng.append((0, self.t))
                if ax + "max" in self.sides:
                    rng.append((n - self.t, n) if fld == "e" else (n - self.t - 1, n - 1))
                for s0, s1 in rng:
                    z = lambda: torch.zeros((s1 - s0) * nt, dtype=torch.float64, device=device)
                    self.layers.append(dict(e=fld == "e", axis=axis, key=fld + ax, start=s0, stop=s1,
                                            psi_a=z(), psi_b=z(), xpad=self.xpad if axis else 0,
                                            row=(s1 - s0) if axis else ny))
        self._coef = {}

    def __getstate__(self):
        st = {k: v for k, v in self.__dict__.items() if k not in ("kappa", "_coef", "layers")}
        st["device"] = str(self.device)
        st["layers_host"] = [{**{k: v for k, v in l.items() if k not in ("psi_a", "psi_b")},
                              "psi_a": to_host(l["psi_a"]), "psi_b": to_host(l["psi_b"])} for l in self.layers]
        return st

    def __setstate__(self, st):
        layers = st.pop("layers_host")
        self.__dict__.update(st)
        self.device = restore_device(st["device"])
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.kappa = {k: dev(v["kappa"]) for k, v in self.host.items()}
        self.layers = [{**l, "psi_a": dev(l["psi_a"]), "psi_b": dev(l["psi_b"])} for l in layers]
        self._coef = {}

    def coef(self, key, dt, d):
        k = (key, dt)
        if k not in self._coef:
            c = self.host[key]
            kap, sig, a = c["kappa"], c["sigma"], c["a"]
            b = np.exp(-(sig / kap + a) * dt)
            with np.errstate(invalid="ignore", divide="ignore"):
                cc = (b - 1) * sig / kap / (sig + kap * a) / d
            cc = np.where(np.isfinite(cc), cc, 0.0)
            mk = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(self.device)
            self._coef[k] = (mk(b), mk(cc))
        return self._coef[k]


class PicEngine2D(RhoContinuityMixin, FusedStepMixin, MigrateWindowMixin):
    dim = 2

    def __init__(self, nx, ny, dx, dy, n_guard=3, device="cuda:0", comm: SlabComm | None = None,
                 x0=0.0, y0=0.0, sort_interval=8, block_particles=8192, migrate_capacity=32768,
                 boundary_conditions=None, cpml_thickness=6, order=_lib.LPA_ORDER_STRIPED):
        """``nx`` is the LOCAL number of cells along x (this rank's slab); the global box has
        ``nx * comm.size`` cells and this slab starts at ``x0 + rank*nx*dx``."""
        self.L = lib()
        bc = dict(boundary_conditions or {k: "periodic" for k in ("xmin", "xmax", "ymin", "ymax")})
        for axis in "xy":
            pair = (bc[axis + "min"], bc[axis + "max"])
            if any(v not in ("periodic", "pml") for v in pair) or (("periodic" in pair) and pair[0] != pair[1]):
                raise ValueError(f"boundary conditions of {axis}: both 'periodic' or each 'pml', got {pair}")
        self.bc = bc
        self.periodic_x, self.periodic_y = bc["xmin"] == "periodic", bc["ymin"] == "periodic"
        self.comm = comm or SlabComm(None, periodic=self.periodic_x)
        if self.comm.periodic != self.periodic_x:
            raise ValueError("SlabComm(periodic=...) must match the x boundary condition")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LpaError("PicEngine2D needs a GPU device; there is no CPU path")
        self.nx, self.ny, self.dx, self.dy, self.ng = int(nx), int(ny), float(dx), float(dy), int(n_guard)
        self.cpml_thickness = int(cpml_thickness)
        self.x0_global, self.y0 = float(x0), float(y0)
        self.x0 = self.x0_global + self.comm.rank * self.nx * self.dx
        self.Lx, self.Ly = self.nx * self.comm.size * self.dx, self.ny * self.dy
        self.grid = DeviceGrid2D(self.nx, self.ny, dx, dy, self.x0, self.y0, n_guard, self.device)
        self.species: list[DeviceParticles] = []
        self.sort_interval = int(sort_interval)
        self.block_particles = int(block_particles)
        self.migrate_capacity = int(migrate_capacity)
        self.order = int(order)   # LPA_ORDER_STRIPED (conflict-free LDS atomics) or LPA_ORDER_CELL_MAJOR
        # axes handled by a local periodic wrap: y if periodic; x only when periodic AND this rank owns
        # the whole box
        self.local_axes = (2 if self.periodic_y else 0) | (1 if (self.periodic_x and self.comm.size == 1) else 0)
        # CPML layers owned by this rank: x faces only on the end ranks
        sides = [s for s in ("ymin", "ymax") if bc[s] == "pml"]
        if bc["xmin"] == "pml" and self.comm.rank == 0:
            sides.append("xmin")
        if bc["xmax"] == "pml" and self.comm.rank == self.comm.size - 1:
            sides.append("xmax")
        self.pml = DevicePML2D(self.nx, self.ny, self.dx, self.dy, sides, self.cpml_thickness,
                               self.device, xpad=self.ng) if sides else None
        # particle absorption at open (PML) faces: the owner's bounds are pulled in by the layer
        # thickness (core/patch/patch.py:105-148) and a particle beyond them has no neighbour to go to
        self.absorb = 0
        self.alo, self.ahi = [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]
        t = self.cpml_thickness
        if not self.periodic_x:
            self.absorb |= _lib.LPA_ABSORB_X
            self.alo[0] = self.x0_global + t * self.dx - self.dx / 2
            self.ahi[0] = self.x0_global + (self.nx * self.comm.size - 1 - t) * self.dx + self.dx / 2
        if not self.periodic_y:
            self.absorb |= _lib.LPA_ABSORB_X << 1
            self.alo[1] = self.y0 + t * self.dy - self.dy / 2
            self.ahi[1] = self.y0 + (self.ny - 1 - t) * self.dy + self.dy / 2
        self.eps0, self.mu0 = constants.EPSILON_0, constants.MU_0
        self._ws = {}
        self._halo = None
        self._side = None       # second stream: J / rho guard planes travel while the interior is pushed
        # hide both message rounds behind the interior tiles?  None = decide per step (the ``overlap`` property); the 3-D
        # engine keeps it on
        self._overlap = None
        self.reuse_slots = True   # arrivals take the slots freed by leavers of their tile (lpa_free_slots)
        self.defer_crossers = True
        # in-kernel cell-index sort (lpa_tiling.slot_class): re-seat the particles whose y-class changed, every step.
        # Correct and tested, but OFF: on C2 it removes 43 % of the LDS bank conflicts and 70 % of the LDS wait cycles
        # (profiles/r02_pmc_reseat.txt) and still leaves K1 at 1.95-1.99 ms against 1.91-1.94 ms without it -- the
        # class test costs 6 % more VALU instructions and the extra parked particles a longer second pass
        self.reseat = False
        # inv_gamma is a function of the momenta: the fused kernels recompute it instead of streaming it (two of the
        # thirteen attribute streams of a particle-update; LPA_PUSH_NO_IG) and the array is rebuilt on demand
        # (DeviceParticles.refresh_inv_gamma: download, diagnostics, the split kernels, a checkpoint)
        self.lazy_inv_gamma = True
        # The tiled kernels hand the particles that have left their tile's staged region since the sort to the overflow
        # list: pushed one by one on global memory at ~30 x the cost of a tiled particle, their number growing two- to
        # three-fold per step once it starts.  ``sort_interval`` (tuned for a 1 keV plasma) is therefore only the LONGEST
        # interval: every sort looks at the overflow count of the push before it (already on the host with the sort's
        # own read-back: no extra sync, nothing per step) and shortens the species' interval when more than
        # ``overflow_sort_fraction`` of it was on the list, lengthens it again when (almost) nobody was.  A hot plasma
        # (u_th >= 0.2) runs at 2.8-4 ms per step instead of 7.5-26 (profiles/r03_sweep_hot2d.txt).  Single slab only: on a
        # slab chain neighbouring slabs must sort -- and re-deposit rho -- in the same steps (sort_due: one clock for all
        # ranks).  0 = fixed interval
        self.overflow_sort_fraction = 0.003
        self.min_sort_interval = 2
        self.reseat_stats = False  # diagnostics: count parked particles / movers / unmatched movers (ws["reloc_stats"])
        self.fused_cpml = True
        self._axes = {}
        self._diag = torch.zeros(8, dtype=torch.float64, device=self.device)
        # bench instrumentation: when a list, (start, end) HIP events are recorded around every
        # launch of the tiled push+deposit kernel on the stream it runs on
        self.kernel_events = None
        # the fused kernel stores the gathered E/B per particle (the reference's ex_part..bz_part side
        # effect, +48 B/particle) only when asked: callbacks that read them set this
        self.write_part_eb = False
        self.sort_part_eb = False  # the sort moves ex_part ... bz_part too (see sort)
        self._rho_init()     # rho from the continuity equation between two real deposits: see rho.py

    def _rho_available(self):
        # the variant deposit paths (CELL_MAJOR / PADDED stores, re-seating) always carry rho
        return self.order == _lib.LPA_ORDER_STRIPED and not self.reseat

    def _rho_sort_due(self):
        return any(self.sort_due(sp) for sp in self.species if sp.n or self.comm.size > 1)

    def _rho_forced_sort_due(self):
        return any(sp.n and (sp.tiling is None or sp.steps_since_sort >= (1 << 29)) for sp in self.species)

    def sort_due(self, sp):
        """the sorter's rule: whenever the store is not tile ordered, and every ``sort_interval`` steps -- or sooner for
        a species whose last intervals ended with a long overflow list (``_adapt_sort_interval``)"""
        if self.comm.size > 1:
            # slab chain: ONE clock for all ranks and species (rho.py: neighbouring slabs must re-deposit rho in the same
            # steps, and a sort step is such a step), whatever a rank's own stores look like -- a slab that was empty
            # pushes its first arrivals with the global kernel until the common sort -- plus the forced re-sorts, which
            # every rank is told at once (window shifts, uploads)
            return self._chain_clock >= self._chain_interval() or sp.steps_since_sort >= (1 << 29)
        return sp.tiling is None or sp.steps_since_sort >= min(self.sort_interval, getattr(sp, "sort_interval_now", 1 << 30))

    def _first_sort_interval(self, sp, cset, comps, d, margin):
        """a store is sorted for the first time: start its interval where 2.5-sigma particles would outrun the tile margin
        (one reduction over the momenta; a 1 keV plasma gets the full ``sort_interval``), the controller takes it from
        there"""
        dt = getattr(self, "_dt_hint", None)
        if self.overflow_sort_fraction <= 0 or not dt or sp.n == 0:
            return
        u = [cset.arr(a)[: sp.n] for a in ("ux", "uy", "uz")]
        live = ~torch.isnan(cset.arr("x")[: sp.n])
        v2 = [torch.where(live, c * c / (1 + u[0] ** 2 + u[1] ** 2 + u[2] ** 2), torch.zeros_like(c)) for c in u[: len(comps)]]
        nl = max(int(live.sum().item()), 1)
        cells = max(float(torch.sqrt(w.sum() / nl).item()) * constants.C_LIGHT * dt / dd for w, dd in zip(v2, d))
        est = int(margin / max(2.5 * cells, 1e-12)) * (2 if self.sort_lookahead else 1)   # (binned for mid-interval: +- T / 2)
        sp.sort_interval_now = max(self.min_sort_interval, min(self.sort_interval, est))

    def _adapt_sort_interval(self, sp, overflow, n_sorted_before):
        """called by sort() with the overflow count of the last push of the interval that just ended"""
        if self.overflow_sort_fraction <= 0 or n_sorted_before <= 0 \
                or sp.steps_since_sort > self.sort_interval:          # (forced sorts of stale stores tell nothing)
            return
        now = min(getattr(sp, "sort_interval_now", self.sort_interval), self.sort_interval)
        f = overflow / n_sorted_before
        if f > self.overflow_sort_fraction:
            # (measured on a u_th = 0.2 plasma: 0.27 % on the list after 9 steps, 8.5 % after 20: ~1.4-fold per step that far out)
            cut = 1 + int(np.log(f / self.overflow_sort_fraction) / np.log(1.4))
            now = max(self.min_sort_interval, min(now, sp.steps_since_sort) - cut)
        elif f < 0.1 * self.overflow_sort_fraction:
            now = min(self.sort_interval, now + 1)
        sp.sort_interval_now = now

    def _rho_particle_slots(self):
        return sum(sp.capacity for sp in self.species)

    def _species_sort_interval(self, sp):
        return getattr(sp, "sort_interval_now", None) if sp.n else None

    def _rho_last_jx_plane(self):
        return self.grid.view("jx")[self.ng + self.nx - 1]

    def _rho_array(self):
        return self.grid.view("rho")

    # ---- restart (RestartDump, `callback/restart.py:88-107`: the reference pickles the whole Simulation) ----
    _TRANSIENT = ("L", "_ws", "_halo", "_side", "_axes", "_diag", "_keep", "kernel_events", "_absorbed", "_jx_plane",
                  "_one", "_step_keep", "_event_pool")

    def __getstate__(self):
        """everything but handles and scratch: the library handle, sort workspaces (and with them the tilings),
        halo buffers, the side stream and cached ctypes descriptors are rebuilt on load; the grid, the particle
        stores and the CPML layers pickle as host arrays (device.py)"""
        self._flush_e2()
        torch.cuda.synchronize(self.device)
        st = {k: v for k, v in self.__dict__.items() if k not in self._TRANSIENT}
        st["device"] = str(self.device)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self.device = restore_device(st["device"])
        self.L = lib()
        self._ws, self._halo, self._side, self._axes, self.kernel_events = {}, None, None, {}, None
        self._diag = torch.zeros(8, dtype=torch.float64, device=self.device)
        self._rho_restore()

    # ---------------------------------------------------------------------------------------------
    @property
    def stream(self):
        return current_stream_ptr(self.device)

    def arrival_area(self) -> int:
        """slots reserved behind the sorted particles for arrivals from the neighbour slabs between
        two sorts (0 on a single rank)"""
        return self.migrate_capacity * 2 * max(self.sort_interval, 1) if self.comm.size > 1 else 0

    def add_species(self, q, m, capacity, with_eb=False) -> int:
        """``capacity`` = live particles this rank must be able to hold; the arrival area is added"""
        self.species.append(DeviceParticles(int(capacity) + self.arrival_area(), self.device, q, m, with_eb))
        return len(self.species) - 1

    def _g(self):
        return C.byref(self.grid.c)

    # ---- Maxwell (MaxwellSolver2D.update_efield/bfield, core/maxwell/solver/solver.py:143-190) ----
    def update_efield(self, dt):
        self._flush_e2()      # (a deferred E half step of run_steps / Simulation.run is completed first)
        if self.pml is None:
            check(self.L.lpa_fdtd_e_2d(self._g(), dt, self.eps0, self.stream), "lpa_fdtd_e_2d")
            return
        p, st = self.pml, self.stream
        if self.fused_cpml:     # kappa sweep + every layer's psi recursion in one launch
            ax, ay = self._cpml_axes(True, dt)
            check(self.L.lpa_fdtd_e_cpml_fused_2d(self._g(), dt, self.eps0, C.byref(ax), C.byref(ay), st),
                  "lpa_fdtd_e_cpml_fused_2d")
            return
        check(self.L.lpa_fdtd_e_cpml_2d(self._g(), dt, self.eps0, p.kappa["ex"].data_ptr(),
                                        p.kappa["ey"].data_ptr(), st), "lpa_fdtd_e_cpml_2d")
        self._psi(True, dt)

    def update_bfield(self, dt):
        self._flush_e2()      # (a deferred E half step of run_steps / Simulation.run is completed first)
        if self.pml is None:
            check(self.L.lpa_fdtd_b_2d(self._g(), dt, self.stream), "lpa_fdtd_b_2d")
            return
        p, st = self.pml, self.stream
        if self.fused_cpml:
            ax, ay = self._cpml_axes(False, dt)
            check(self.L.lpa_fdtd_b_cpml_fused_2d(self._g(), dt, C.byref(ax), C.byref(ay), st),
                  "lpa_fdtd_b_cpml_fused_2d")
            return
        check(self.L.lpa_fdtd_b_cpml_2d(self._g(), dt, p.kappa["bx"].data_ptr(), p.kappa["by"].data_ptr(), st),
              "lpa_fdtd_b_cpml_2d")
        self._psi(False, dt)

    def _cpml_axes(self, efield, dt):
        """lpa_cpml_axis descriptors (x, y) of the E or B update for this dt, cached per PML object"""
        key = (id(self.pml), bool(efield), dt)
        if key not in self._axes:
            out = []
            for axis, ax in enumerate("xy"):
                k = ("e" if efield else "b") + ax
                d = _lib.lpa_cpml_axis()
                d.kappa = self.pml.kappa[k].data_ptr()
                b, cc = self.pml.coef(k, dt, self.dx if axis == 0 else self.dy)
                d.bcoeff, d.ccoeff_d = b.data_ptr(), cc.data_ptr()
                d.lo0 = d.lo1 = d.hi0 = d.hi1 = 0
                for ly in self.pml.layers:
                    if ly["e"] != bool(efield) or ly["axis"] != axis:
                        continue
                    if ly["start"] == 0:
                        d.lo0, d.lo1 = ly["start"], ly["stop"]
                        d.psi_a_lo, d.psi_b_lo = psi_ptr(ly, "psi_a"), psi_ptr(ly, "psi_b")
                    else:
                        d.hi0, d.hi1 = ly["start"], ly["stop"]
                        d.psi_a_hi, d.psi_b_hi = psi_ptr(ly, "psi_a"), psi_ptr(ly, "psi_b")
                out.append(d)
            if len(self._axes) > 16:
                self._axes.clear()
            self._axes[key] = (tuple(out), self.pml)      # keeps the PML object (and its arrays) alive
        return self._axes[key][0]

    def _psi(self, efield, dt):
        for ly in self.pml.layers:
            if ly["e"] != efield:
                continue
            b, cc = self.pml.coef(ly["key"], dt, self.dx if ly["axis"] == 0 else self.dy)
            check(self.L.lpa_cpml_psi_2d(self._g(), int(efield), ly["axis"], ly["start"], ly["stop"], dt,
                                         b.data_ptr(), cc.data_ptr(), psi_ptr(ly, "psi_a"),
                                         psi_ptr(ly, "psi_b"), self.stream), "lpa_cpml_psi_2d")

    # ---- laser injection (Laser.__call__ at stage '_laser', callback/laser.py:109-137) -------------
    def laser_inject(self, ey_source, ez_source, dt):
        """``ey_source, ez_source``: host or device arrays [ny] of the source fields on the x-min
        boundary at the current time.  Only the rank that owns the x-min layer injects."""
        if self.pml is None or "xmin" not in self.pml.sides:
            return
        t = self.cpml_thickness
        iy0 = t if self.bc["ymin"] == "pml" else 0
        iy1 = self.ny - t if self.bc["ymax"] == "pml" else self.ny
        to = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(
            self.device, torch.float64)
        ey, ez = to(ey_source), to(ez_source)
        check(self.L.lpa_laser_inject_2d(self._g(), t + 2, dt, self.eps0, iy0, iy1, ey.data_ptr(), ez.data_ptr(),
                                         self.stream), "lpa_laser_inject_2d")
        self._keep = (ey, ez)   # keep the buffers alive until the stream has consumed them

    def laser_inject_sep(self, pc, ps, k4, dt):
        """factorised sources (device tensors ``pc``, ``ps`` [ny], four floats ``k4``): ey = k4[0] pc + k4[1] ps,
        ez = k4[2] pc + k4[3] ps, evaluated inside the injection kernel (lpa_laser_inject_sep_2d)"""
        if self.pml is None or "xmin" not in self.pml.sides:
            return
        t = self.cpml_thickness
        iy0 = t if self.bc["ymin"] == "pml" else 0
        iy1 = self.ny - t if self.bc["ymax"] == "pml" else self.ny
        kk = (C.c_double * 4)(*[float(v) for v in k4])
        check(self.L.lpa_laser_inject_sep_2d(self._g(), t + 2, dt, self.eps0, iy0, iy1, pc.data_ptr(), ps.data_ptr(), kk,
                                             self.stream), "lpa_laser_inject_sep_2d")

    # ---- guard cells (Patches.sync_guard_fields + MPIManager.sync_guard_fields_start/_wait) --------
    def _halo_bufs(self):
        if self._halo is None:
            n = self.ng * self.grid.NY
            mk = lambda c: torch.empty(c * n, dtype=torch.float64, device=self.device)
            # 6 components: a guard sync of E and B together (the reference's default attrs) is the largest message
            self._halo = {"s_lo": mk(6), "s_hi": mk(6), "r_lo": mk(6), "r_hi": mk(6)}
        return self._halo

    def _halo_views(self, n):
        """the first ``n`` doubles of the four face buffers; the C ABI takes no length, so check here"""
        bufs = self._halo_bufs()
        if any(n > v.numel() for v in bufs.values()):
            raise _lib.LpaError(f"halo message of {n} doubles exceeds the face buffers")
        return {k: v[:n] for k, v in bufs.items()}

    def _faces(self, op, which=0):
        """both x faces in one launch (lpa_halo_faces); a missing buffer (open chain end) is skipped"""
        def run(b_lo, b_hi):
            check(self.L.lpa_halo_faces(self._g(), op, which, b_lo.data_ptr() if b_lo is not None else None,
                                        b_hi.data_ptr() if b_hi is not None else None, self.stream), "lpa_halo_faces")
        return run

    def sync_guard_fields(self, attrs):
        self._flush_e2()
        which = (1 if any(a in attrs for a in ("ex", "ey", "ez")) else 0) | \
                (2 if any(a in attrs for a in ("bx", "by", "bz")) else 0)
        st = self.stream
        check(self.L.lpa_guard_wrap(self._g(), which, self.local_axes, st), "lpa_guard_wrap")
        if self.comm.size > 1:
            h = self._halo_views(3 * bin(which).count("1") * self.ng * self.grid.NY)
            # my low interior edge becomes the LEFT neighbour's high guard and vice versa
            rho_msg = self._rho_message()      # a deferred rho update: its jx plane rides with these planes (rho.py)
            if rho_msg is not None:
                lo, hi = (h["s_lo"] if self.comm.has_left else None), (h["s_hi"] if self.comm.has_right else None)
                self._faces(_lib.LPA_HALO_PACK_GUARD_SRC, which)(lo, hi)
                self.comm.exchange_many([(h["s_lo"], h["s_hi"], h["r_lo"], h["r_hi"]), rho_msg])
                self._faces(_lib.LPA_HALO_UNPACK_GUARD, which)(h["r_lo"] if self.comm.has_left else None,
                                                               h["r_hi"] if self.comm.has_right else None)
                self._complete_rho()
                return
            exchange_faces(
                self.comm,
                lambda side, b: check(self.L.lpa_halo_pack_guard_src(self._g(), which, side, b.data_ptr(), st),
                                      "lpa_halo_pack_guard_src"),
                lambda side, b: check(self.L.lpa_halo_unpack_guard(self._g(), which, side, b.data_ptr(), st),
                                      "lpa_halo_unpack_guard"),
                h, pack2=self._faces(_lib.LPA_HALO_PACK_GUARD_SRC, which),
                unpack2=self._faces(_lib.LPA_HALO_UNPACK_GUARD, which))

    # ---- currents (CurrentDeposition2D.reset, Patches.sync_currents + MPIManager.sync_currents_*) --
    def reset_current(self):
        """`CurrentDeposition2D.reset` (core/current/cpu2d.c:19-72); also decides this step's rho mode (rho.py): a real
        deposit zeroes jx jy jz rho, a continuity step zeroes the currents only"""
        self._begin_deposit_step()

    def sync_currents(self):
        st = self.stream
        if self.comm.size > 1:
            h = self._halo_views(4 * self.ng * self.grid.NY)
            # my low GUARD planes are added to the LEFT neighbour's high interior edge
            exchange_faces(
                self.comm,
                lambda side, b: check(self.L.lpa_halo_pack_current(self._g(), side, b.data_ptr(), st),
                                      "lpa_halo_pack_current"),
                lambda side, b: check(self.L.lpa_halo_unpack_current(self._g(), side, b.data_ptr(), st),
                                      "lpa_halo_unpack_current"),
                h, pack2=self._faces(_lib.LPA_HALO_PACK_CURRENT), unpack2=self._faces(_lib.LPA_HALO_UNPACK_CURRENT))
        check(self.L.lpa_current_fold(self._g(), self.local_axes, st), "lpa_current_fold")
        self._finish_rho()

    # ---- sort (ParticleSort2D.__call__, core/sort/particle_sort.py:196-211) ------------------------
    def _sort_ws(self, sp: DeviceParticles):
        key = id(sp)
        if key not in self._ws:
            nbytes = self.L.lpa_sort_workspace_bytes_ranks(self._g(), sp.capacity, getattr(sp, "stripe_ranks", 0))
            area = self.arrival_area()
            self._ws[key] = {
                "sort": torch.zeros(nbytes, dtype=torch.uint8, device=self.device),
                "overflow": torch.empty(sp.capacity, dtype=torch.int32, device=self.device),
                # 0: overflow list, 1: arrival cursor, 2: overflow list of the edge part, 3: leavers that did not fit
                # (4: this step's leaver list, lpa_push_params.leavers)
                "counters": torch.zeros(8, dtype=torch.int32, device=self.device),
                "area": area,
                "tiling": _lib.lpa_tiling(),
                "mig": None,
            }
        return self._ws[key]

    deep_tail_fraction = 0.005   # re-size the stripes when this share of a store lies beyond them (see sort)
    sort_lookahead = True        # a store the controller re-sorts early is binned for the middle of its interval
    # ... and one on the full interval too: the particles' mean distance from the cells they were binned for halves, the
    # order ages half as fast (C2: K1 1.658 -> 1.601 ms, step -2 %; LPA_SORT_LOOKAHEAD_COLD=0 for A/B runs)
    sort_lookahead_cold = os.environ.get("LPA_SORT_LOOKAHEAD_COLD", "1") != "0"

    def _sort_ahead(self, sp):
        """look-ahead time of the sort (lpa_sort_tiles_ahead_*): half the interval the controller runs the store on, once
        that is shorter than ``sort_interval`` -- i.e. its particles outrun the tile margin.  Cold stores: 0 (the plain
        sort, two attribute streams less in its count pass)."""
        now = sp.get("sort_interval_now") if isinstance(sp, dict) else getattr(sp, "sort_interval_now", None)
        if self.comm.size > 1:          # a slab chain sorts on its common clock
            now = self._chain_interval()
        if self.sort_lookahead_cold and self.sort_lookahead:      # (experiment: also for stores on the full interval)
            return 0.5 * min(now or self.sort_interval, self.sort_interval) * getattr(self, "_dt_hint", 0.0)
        if not self.sort_lookahead or now is None or now >= self.sort_interval:
            return 0.0
        return 0.5 * now * getattr(self, "_dt_hint", 0.0)

    def sort(self, ispec, _again=False):
        """tile-bin species ``ispec`` (drops dead slots).  One host sync (live count read-back).

        The striped order keeps ``stripe_ranks`` particles per cell in stripes (default: twice the store's mean over ALL
        tiles); what a cell holds beyond that follows cell by cell, where the lanes of a wave share a cell and the LDS
        atomics of the tiled kernel serialise.  A target that fills a fraction of the box is much deeper than that mean
        (a laser-target slab at 256 per cell in a box that is 97 % empty: 4 x slower per particle).  The sort reports
        its deepest cell and the particles beyond the stripes with the live count; when they matter the workspace is
        re-sized for the deepest cell + 25 % and the store sorted once more, here and now."""
        sp = self.species[ispec]
        ws = self._ws_checked(sp)
        forced = _again or sp.tiling is None or sp.steps_since_sort >= (1 << 29)     # (no valid order: rho.py _rho_sorted)
        src, dst = sp.cset, sp.other()
        # ex_part ... bz_part are what the LAST push saw: the push that follows this sort rewrites them (or nobody reads
        # them before one that does, Simulation._host_callback_near), so the sort leaves them where they are -- six of
        # thirteen attribute arrays.  The split pusher path has stages between the sort and its interpolation: it asks
        # for them to travel (sort_part_eb).
        ps, pd = src.cstruct(sp.n, eb=self.sort_part_eb), dst.cstruct(dst.capacity, eb=self.sort_part_eb)
        if self._noig():          # nobody reads the store's inv_gamma before a refresh: the sort need not move it
            ps.inv_gamma = pd.inv_gamma = None
            sp.ig_stale = True
        # small stores: smaller work blocks, or a 2 M-particle species is 256 workgroups on 256 CUs (config C3: K1
        # 0.19 -> 0.15 ms per step with 4096, tools/exp_c3_blocks.py)
        bp = self.block_particles if sp.n >= (1 << 23) else min(self.block_particles, 4096)
        # a re-sort: the first n_sorted slots are the previous sort's result (lpa_tiling.prefix_hint)
        ws["tiling"].prefix_hint = int(sp.n_sorted) if sp.tiling is not None else 0
        ws["tiling"].stripe_ranks = getattr(sp, "stripe_ranks", 0)
        ahead = self._sort_ahead(sp)
        check(self.L.lpa_sort_tiles_ahead_2d(self._g(), C.byref(ps), C.byref(pd), ws["sort"].data_ptr(),
                                             ws["sort"].numel(), bp, self.order,
                                             C.byref(ws["tiling"]), ahead, self.stream), "lpa_sort_tiles_2d")
        sp.sort_ahead_used = ahead
        n_live, deepest, tail, _, _ = _lib.sort_result(self.L, ws["sort"], True)    # sync point (once per sort_interval steps)
        cnts = ws["counters"].tolist()
        arrivals, surplus = cnts[1], cnts[3]
        self._mig_sample(ws["mig"])      # (the counts of the last step's face messages: dist.MigrateWindowMixin)
        if _again:
            pass                                                     # (the controller saw the first pass)
        elif sp.tiling is not None:
            self._adapt_sort_interval(sp, cnts[0] + cnts[2], sp.n_sorted)      # the overflow lists of the last push (interior + edge part)
        else:
            self._first_sort_interval(sp, src, ("ux", "uy"), (self.dx, self.dy), _lib.LPA_TILE_MARGIN)
        if arrivals > ws["area"]:
            raise _lib.LpaError(f"arrival area overflow: {arrivals} > {ws['area']} (raise migrate_capacity)")
        if surplus > 0 and not self._mig_surplus(surplus):
            raise _lib.LpaError(self._surplus_message(surplus))
        sp.cur = 1 - sp.cur
        sp.n_sorted = n_live
        area = ws["area"]
        if n_live + area > sp.capacity:
            raise _lib.LpaError(f"particle capacity {sp.capacity} < live {n_live} + arrival area {area}")
        if area:
            dst.arr("x")[n_live:n_live + area].fill_(float("nan"))
        sp.n = n_live + area
        ws["counters"].zero_()
        ws["tiling"].n_sorted = n_live
        # the set that was just sorted FROM is idle until the next sort: scratch for the dense second
        # pass of the tiled kernel (particles that changed cell)
        idle = sp.other()
        for c, a in enumerate(("x", "y", "ux", "uy", "uz", "inv_gamma", "w")):
            ws["tiling"].scratch[c] = idle.arr(a).data_ptr() if self.defer_crossers else None
        # in-kernel cell-index sort (lpa_tiling.slot_class): every step the tiled kernel re-seats the particles whose
        # y-class changed, so the order stays as conflict-free as right after this sort; the first push writes the
        # classes
        if self.reseat and self.defer_crossers and self.order == _lib.LPA_ORDER_STRIPED:
            if ws.get("cls") is None or ws["cls"].numel() < sp.capacity:
                ws["cls"] = torch.zeros(sp.capacity, dtype=torch.int16, device=self.device)
            ws["tiling"].scratch[7] = idle.id.data_ptr()
            ws["tiling"].slot_class = ws["cls"].data_ptr()
            ws["tiling"].class_init = 1
            if self.reseat_stats:
                ws.setdefault("reloc_stats", torch.zeros(4, dtype=torch.int32, device=self.device))
                ws["tiling"].reloc_stats = ws["reloc_stats"].data_ptr()
        else:
            ws["tiling"].scratch[7] = None
            ws["tiling"].slot_class = None
            ws["tiling"].class_init = 0
        sp.tiling = ws["tiling"]
        sp.steps_since_sort = 0
        self._rho_sorted(forced)
        self._reset_free_slots(ws, ws["tiling"].tiles_x, ws["tiling"].tiles_y, _lib.LPA_TILE_X)
        used, want = ws["tiling"].stripe_ranks, min(_lib.LPA_MAX_STRIPE_RANKS, deepest + deepest // 4)
        if not _again and want > used and used < getattr(sp, "stripe_ranks_limit", 1 << 30) and \
                self.order == _lib.LPA_ORDER_STRIPED and tail > self.deep_tail_fraction * max(n_live, 1):
            sp.stripe_ranks = want
            self._ws.pop(id(sp), None)              # a workspace with room for the deeper stripes; its header knows
            sp.tiling = None                        # nothing of this order: a full sort, no prefix hint
            self.sort(ispec, _again=True)
            if self._ws[id(sp)]["tiling"].stripe_ranks < want:     # the table would not fit (1 GiB): do not ask again
                sp.stripe_ranks_limit = self._ws[id(sp)]["tiling"].stripe_ranks

    FREE_SLOT_DEPTH = 64

    def _reset_free_slots(self, ws, tiles_x, tiles_per_col, tile_x):
        """free-slot stacks of the edge tile columns (lpa_free_slots): emptied after every sort; the columns
        cover what a particle can leave the slab from until the next sort (< 1 cell per step)"""
        ws["fs"] = None
        if self.comm.size == 1:
            return
        cols = int(np.ceil((1.5 * self.sort_interval + 2) / tile_x))      # (age + the sort's look-ahead of up to half an interval)
        if 2 * cols > tiles_x:
            return
        n = 2 * cols * tiles_per_col
        if ws.get("fs_count") is None or ws["fs_count"].numel() != n:
            ws["fs_count"] = torch.zeros(n, dtype=torch.int32, device=self.device)
            ws["fs_slot"] = torch.empty(n * self.FREE_SLOT_DEPTH, dtype=torch.int32, device=self.device)
        else:
            ws["fs_count"].zero_()
        fs = _lib.lpa_free_slots()
        fs.count, fs.slot = ws["fs_count"].data_ptr(), ws["fs_slot"].data_ptr()
        fs.edge_cols, fs.depth = cols, self.FREE_SLOT_DEPTH
        ws["fs"] = fs

    def _ws_checked(self, sp):
        ws = self._sort_ws(sp)
        if sp.n > sp.capacity:
            raise _lib.LpaError("particle store over capacity")
        return ws

    # ---- fused pusher (BorisPusher.__call__(dt, unified=True), core/pusher/pusher.py:116-127) ------
    def _push_params(self, sp, dt):
        pp = _lib.lpa_push_params()
        pp.dt, pp.q, pp.m = dt, sp.q, sp.m
        pp.wrap = self.local_axes | self.absorb
        for a in range(3):
            pp.alo[a], pp.ahi[a] = self.alo[a], self.ahi[a]
        pp.lo[0], pp.hi[0] = self.x0_global - self.dx / 2, self.x0_global + self.Lx - self.dx / 2
        pp.lo[1], pp.hi[1] = self.y0 - self.dy / 2, self.y0 + self.Ly - self.dy / 2
        pp.lo[2], pp.hi[2] = 0.0, 0.0
        self._push_flags(pp, dt, self.absorb)
        if self._noig():
            pp.flags |= _lib.LPA_PUSH_NO_IG
            sp.ig_stale = True
        else:
            sp.refresh_inv_gamma()      # this launch streams inv_gamma: it must be there (a no-op unless it is stale)
        return pp

    def _noig(self):
        """may the fused kernels leave inv_gamma alone?  (what the tiled kernel's LPA_PUSH_NO_IG instantiation needs)"""
        return (self.lazy_inv_gamma and not self.write_part_eb and self.order == _lib.LPA_ORDER_STRIPED
                and not self.reseat and self.defer_crossers)

    def push_deposit(self, ispec, dt, tiled=True, part=_lib.LPA_PART_ALL, edge_cols=0):
        """``part``: LPA_PART_EDGE pushes the edge tile columns, the overflow list and the loose
        (arrival-area) particles -- everything that can deposit into the x guard planes --,
        LPA_PART_INTERIOR the remaining tiles (+ their overflow list)"""
        sp = self.species[ispec]
        if sp.n == 0:
            return
        st = self.stream
        pp = self._push_params(sp, dt)
        pc = sp.cset.cstruct(sp.n, eb=self.write_part_eb)
        if tiled and sp.tiling is not None and sp.n_sorted > 0:
            ws = self._sort_ws(sp)
            # the edge part may run on a second stream beside the interior part: own overflow list + counter
            if part == _lib.LPA_PART_EDGE:
                if "overflow_edge" not in ws:
                    ws["overflow_edge"] = torch.empty(sp.capacity, dtype=torch.int32, device=self.device)
                ovf, cslot = ws["overflow_edge"], 2
            else:
                ovf, cslot = ws["overflow"], 0
            ws["counters"][cslot:cslot + 1].zero_()
            cnt = ws["counters"][cslot:cslot + 1].data_ptr()
            timed = self.kernel_events is not None
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream(self.device))
            check(self.L.lpa_push_deposit_tiled_part_2d(self._g(), C.byref(pc), C.byref(pp), C.byref(sp.tiling),
                                                        ovf.data_ptr(), cnt, part, edge_cols, st),
                  "tiled")
            if timed:
                e1.record(torch.cuda.current_stream(self.device))
                self.kernel_events.append((e0, e1))
            check(self.L.lpa_push_deposit_list_2d(self._g(), C.byref(pc), C.byref(pp),
                                                  ovf.data_ptr(), cnt, sp.n_sorted, st), "list")
            loose = sp.n - sp.n_sorted
            if loose > 0 and part != _lib.LPA_PART_INTERIOR:
                check(self.L.lpa_push_deposit_2d(self._g(), C.byref(pc), C.byref(pp), sp.n_sorted, loose, st),
                      "loose")
        elif part != _lib.LPA_PART_INTERIOR:
            check(self.L.lpa_push_deposit_2d(self._g(), C.byref(pc), C.byref(pp), 0, sp.n, st), "global")
        if part != _lib.LPA_PART_EDGE:
            sp.steps_since_sort += 1
            if sp.tiling is not None:
                sp.tiling.class_init = 0      # the slot classes are written by the first push after a sort

    def edge_columns(self, dt):
        """tile columns at each x face whose particles (or anything that drifted out of them since the
        last sort, at < c) can reach the x guard planes: the rest is safe to push while those planes
        travel.  Uses the actual age of the order (a disabled sorter makes it grow).  0 = no overlap
        possible (slab too thin for that much drift)."""
        age = max([sp.steps_since_sort for sp in self.species if sp.n] + [0]) + 1
        # (+ how far ahead of its position the sort may have binned a particle: one that sits at the face and moves inward
        # is filed that much further in)
        ahead = max([getattr(sp, "sort_ahead_used", 0.0) for sp in self.species] + [0.0]) * constants.C_LIGHT / self.dx
        drift = constants.C_LIGHT * dt / self.dx * age + 5.0 + ahead     # + the 3 nodes a deposit window reaches, + 1, + 1 (the
        # last node plane's own jx travels with the guard planes: rho_exchange 2)
        cols = int(np.ceil(drift / _lib.LPA_TILE_X))
        return cols if 2 * cols < self.nx // _lib.LPA_TILE_X else 0

    def leaver_columns(self, age):
        """tile columns at each x face that can hold particles which left the slab during the ``age`` steps
        since the sort (each step moves a particle less than one cell: the tiled push requires
        c dt <= dx); 0 = scan everything"""
        # (+ the sort's look-ahead: a particle that turned round after the sort was filed that much further in)
        ahead = max([getattr(sp, "sort_ahead_used", 0.0) for sp in self.species] + [0.0]) * constants.C_LIGHT / self.dx
        cols = int(np.ceil((age + 1 + ahead) / _lib.LPA_TILE_X))
        return cols if 2 * cols <= self.nx // _lib.LPA_TILE_X else 0

    def push_deposit_overlapped(self, dt):
        """push + deposit of all species with the J / rho guard-plane exchange hidden behind the interior
        tiles: edge tile columns (+ overflow + arrivals) first, then the exchange runs on a second
        stream while the interior is pushed; the received planes are folded in afterwards (the
        reference overlaps the same way: sync_currents_start ... intra-rank work ... _wait,
        simulation.py:1155-1188).  Returns False when the slab is too thin to split."""
        cols = self.edge_columns(dt)
        if self.comm.size == 1 or cols == 0 or any(sp.tiling is None for sp in self.species if sp.n):
            return False
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device, priority=-1)   # high priority: the edge goes first
        ready, done = torch.cuda.Event(), torch.cuda.Event()
        ready.record(main)
        h = self._halo_views(4 * self.ng * self.grid.NY)
        # edge tiles + pack + exchange on the high-priority side stream, the interior tiles on the main stream AT
        # THE SAME TIME (they touch disjoint particles and never the x guard planes; both add into J with
        # atomics): a separate edge launch in front of the interior one cost a whole extra round of workgroups
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            for i in range(len(self.species)):
                self.push_deposit(i, dt, part=_lib.LPA_PART_EDGE, edge_cols=cols)
            self._faces(_lib.LPA_HALO_PACK_CURRENT)(h["s_lo"] if self.comm.has_left else None,
                                                    h["s_hi"] if self.comm.has_right else None)
            self.comm.exchange(h["s_lo"], h["s_hi"], h["r_lo"], h["r_hi"])
            done.record(self._side)
        for i in range(len(self.species)):
            self.push_deposit(i, dt, part=_lib.LPA_PART_INTERIOR, edge_cols=cols)
        main.wait_event(done)
        self._faces(_lib.LPA_HALO_UNPACK_CURRENT)(h["r_lo"] if self.comm.has_left else None,
                                                  h["r_hi"] if self.comm.has_right else None)
        check(self.L.lpa_current_fold(self._g(), self.local_axes, self.stream), "lpa_current_fold")
        self._finish_rho()
        return True

    # ---- split kernels: the path the reference takes when a callback sits in a pusher stage --------
    # (`simulation/simulation.py:993-1038`): push_position, interpolate, boris, push_position, deposit
    def _pc_eb(self, ispec):
        sp = self.species[ispec]
        if not sp.with_eb:
            raise _lib.LpaError("the split pusher path needs per-particle E/B arrays: add_species(with_eb=True)")
        return sp, sp.cset.cstruct(sp.n)

    def push_position(self, ispec, dt):
        sp = self.species[ispec]
        sp.refresh_inv_gamma()
        pc = sp.cset.cstruct(sp.n)
        check(self.L.lpa_push_position_2d(C.byref(pc), dt, self.stream), "lpa_push_position_2d")

    def interpolate(self, ispec):
        sp, pc = self._pc_eb(ispec)
        check(self.L.lpa_interpolate_2d(self._g(), C.byref(pc), self.stream), "lpa_interpolate_2d")

    def boris(self, ispec, dt):
        sp, pc = self._pc_eb(ispec)
        check(self.L.lpa_boris(C.byref(pc), dt, sp.q, sp.m, self.stream), "lpa_boris")

    def deposit(self, ispec, dt):
        """standalone Esirkepov deposit + the periodic position fold the fused kernel applies itself"""
        sp = self.species[ispec]
        if self._no_rho:
            raise _lib.LpaError("the standalone deposit carries rho: set rho_continuity_blocked (or rho_continuity = "
                                "False) before reset_current() when the split pusher path is used")
        sp.refresh_inv_gamma()
        pc = sp.cset.cstruct(sp.n)
        check(self.L.lpa_deposit_2d(self._g(), C.byref(pc), dt, sp.q, self.stream), "lpa_deposit_2d")
        pp = self._push_params(sp, dt)
        check(self.L.lpa_wrap_positions_2d(C.byref(pc), C.byref(pp), self.stream), "lpa_wrap_positions_2d")
        sp.steps_since_sort += 1

    # ---- particle ownership (mpi.sync_particles_* + Patches.sync_particles) ------------------------
    def sync_particles(self, ispec):
        """periodic wrap is fused into the push kernel; between slabs leavers travel to the ring
        neighbours in one fixed-size message per face (count in band, no host sync)."""
        if self.comm.size == 1:
            return
        m, fs = self._mig_pack(ispec)
        v = self._mig_views(m)
        self.comm.exchange(v["s_lo"], v["s_hi"], v["r_lo"], v["r_hi"])
        self._mig_unpack(ispec, m, fs)

    def _owner_bounds_x(self):
        """a particle left of / right of these belongs to the left / right neighbour"""
        return self.x0 - self.dx / 2, self.x0 + (self.nx - 1) * self.dx + self.dx / 2

    def _slab_species(self, sp, pushed):
        """migration bookkeeping of one species on a slab rank: face message buffers, arrival cursor, surplus counter,
        free-slot stacks (or None), arrival-area capacity, tile columns the leaver scan covers (0 = every slot).
        ``pushed``: this step's push has run already (its age counts)"""
        ws = self._sort_ws(sp)
        cap = self.migrate_capacity
        if ws["mig"] is None:
            mk = lambda: torch.zeros(1 + LPA_MIG_NATTR * cap, dtype=torch.float64, device=self.device)
            ws["mig"] = {"s_lo": mk(), "s_hi": mk(), "r_lo": mk(), "r_hi": mk()}
        if sp.tiling is None:
            raise _lib.LpaError("sync_particles on a slab decomposition needs a sorted store (arrival area)")
        # only the tile columns within drift range of an x face (and the loose particles) can hold leavers
        cols = self.leaver_columns(sp.steps_since_sort + (0 if pushed else 1))
        fs = ws.get("fs") if self.reuse_slots else None
        if fs is not None and (cols == 0 or cols > fs.edge_cols):
            fs = None        # the order is older than the stacks were sized for
        out = {"bufs": ws["mig"], "cursor": ws["counters"][1:2], "surplus": ws["counters"][3:4], "fs": fs,
               "area": ws["area"], "cols": cols}
        if self.overlap and self.native_slab():       # the edge part of an overlapped push has a list and counter of its own
            if "overflow_edge" not in ws:
                ws["overflow_edge"] = torch.empty(sp.capacity, dtype=torch.int32, device=self.device)
            out.update(overflow_edge=ws["overflow_edge"], edge_count=ws["counters"][2:3])
        if self.leaver_lists and self.native_slab():
            # the push kernels list the slots that left the slab (exact: no scan of the edge tile columns, and the free-slot
            # stacks stay usable however old the order is)
            if ws.get("leavers") is None:
                ws["leavers"] = torch.empty(2 * cap, dtype=torch.int64, device=self.device)
            out.update(leavers=ws["leavers"], leaver_count=ws["counters"][4:5], fs=ws.get("fs") if self.reuse_slots else None)
        return out

    leaver_lists = True      # native slab steps: the push kernels report the leavers (lpa_push_params.leavers)

    def n_x_local(self):
        return self.nx

    OVERLAP_MAX_EDGE_FRACTION = 0.10

    @property
    def overlap(self):
        """None (the default) = behind a real RCCL communicator, when the edge tile columns are a small part of the slab:
        one-rank RCCL communicator, ``run_steps`` -- 512 x 512 cells (edge 2 x 2 of 64 tile columns) 0.263 ms overlapped
        against 0.278 in line, 1024 x 1024 1.937 against 1.980, 128 x 1024 (4 of 16 columns) 0.400 against 0.402; without a
        wire the split only costs (0.221 against 0.215)"""
        if self._overlap is not None:
            return self._overlap
        if not (self.comm.size > 1 and self.comm.native is not None and self.comm.native_kind == _lib.LPA_COMM_RCCL):
            return False
        dt = getattr(self, "_dt_hint", 0.0)
        cols = self.edge_columns(dt) if dt > 0 else 0
        return cols > 0 and 2 * cols <= self.OVERLAP_MAX_EDGE_FRACTION * (self.nx // _lib.LPA_TILE_X)

    @overlap.setter
    def overlap(self, v):
        self._overlap = None if v is None else bool(v)

    def _slab_fill(self, slab):
        """the slab section of an lpa_step descriptor (step.py); returns what must stay alive until the launches ran"""
        slab.xlo, slab.xhi = self._owner_bounds_x()
        slab.shift_lo, slab.shift_hi = self.comm.arrival_shift(self.Lx)
        slab.migrate_capacity = self.migrate_window       # (the message and its SoA stride: what travels of the buffers)
        h = self._halo_views(4 * self.ng * self.grid.NY)
        slab.cur_r_lo, slab.cur_r_hi = h["r_lo"].data_ptr(), h["r_hi"].data_ptr()
        # (2: the jx plane of the continuity update is formed from what travels with J -- steps without B messages)
        slab.rho_exchange = int(self.rho_continuity and self._rho_available()) * (2 if self.local_b() else 1)
        if slab.rho_exchange:
            self._jx_plane_bufs()
            slab.jx_left_plane = self._jx_plane.data_ptr()
        # overlapped: edge tile columns, leaver pack and the exchange on a second stream beside the interior tiles
        dt = getattr(self, "_dt_hint", 0.0)
        slab.overlap_cols = self.edge_columns(dt) if (self.overlap and dt > 0) else 0
        return h

    def _mig_pack(self, ispec):
        """leavers of species ``ispec`` into its two face messages; returns (buffers, free-slot stacks)"""
        sp = self.species[ispec]
        mig = self._slab_species(sp, pushed=True)
        ws, cap, st = self._sort_ws(sp), self.migrate_window, self.stream
        m, cols, fs = mig["bufs"], mig["cols"], mig["fs"]
        pc = sp.cset.cstruct(sp.n)
        xlo, xhi = self._owner_bounds_x()
        surplus = mig["surplus"].data_ptr()          # leavers beyond migrate_capacity (checked at the next sort)
        if cols:
            check(self.L.lpa_migrate_pack_edges_x(C.byref(pc), C.byref(sp.tiling), cols, xlo, xhi,
                                                  m["s_lo"].data_ptr(), m["s_hi"].data_ptr(), cap,
                                                  C.byref(fs) if fs is not None else None, surplus, st),
                  "lpa_migrate_pack_edges_x")
        else:
            check(self.L.lpa_migrate_pack_x(C.byref(pc), xlo, xhi, m["s_lo"].data_ptr(), m["s_hi"].data_ptr(),
                                            cap, surplus, st), "lpa_migrate_pack_x")
        if not self.comm.has_left:
            m["r_lo"][:1].zero_()    # open face: nothing arrives (count = 0)
        if not self.comm.has_right:
            m["r_hi"][:1].zero_()
        return m, fs

    def _mig_unpack(self, ispec, m, fs):
        sp = self.species[ispec]
        ws, cap, st = self._sort_ws(sp), self.migrate_window, self.stream
        pc = sp.cset.cstruct(sp.n)
        cur = ws["counters"][1:2].data_ptr()
        # arrivals through my low face come from the left neighbour; at the global low edge they
        # crossed the periodic boundary: x > xmax_global -> x - Lx (sync_particles_2d.c:168-182)
        shift_lo, shift_hi = self.comm.arrival_shift(self.Lx)
        for buf, shift in ((m["r_lo"], shift_lo), (m["r_hi"], shift_hi)):
            if fs is not None:   # arrivals take the slots the leavers of their tile freed, when there are any
                check(self.L.lpa_migrate_unpack_tiled(C.byref(pc), self._g(), C.byref(sp.tiling), C.byref(fs),
                                                      sp.n_sorted, ws["area"], cur, buf.data_ptr(), cap, shift, st),
                      "lpa_migrate_unpack_tiled")
            else:
                check(self.L.lpa_migrate_unpack(C.byref(pc), sp.n_sorted, ws["area"], cur, buf.data_ptr(), cap,
                                                shift, st), "lpa_migrate_unpack")

    def sync_currents_and_particles(self):
        """the J / rho guard fold and the migration of every species in ONE message round (the reference
        issues them back to back: sync_currents, then sync_particles, simulation.py:1155-1200): one grouped
        send / recv instead of 1 + nspecies -- every round costs a launch and a handshake with both
        neighbours"""
        if self.comm.size == 1:
            self.sync_currents()
            return
        h = self._halo_views(4 * self.ng * self.grid.NY)
        self._faces(_lib.LPA_HALO_PACK_CURRENT)(h["s_lo"] if self.comm.has_left else None,
                                                    h["s_hi"] if self.comm.has_right else None)
        packed = [self._mig_pack(i) for i in range(len(self.species)) if self.species[i].n]
        idx = [i for i in range(len(self.species)) if self.species[i].n]
        views = [self._mig_views(m) for m, _ in packed]
        self.comm.exchange_many([(h["s_lo"], h["s_hi"], h["r_lo"], h["r_hi"])] +
                                [(v["s_lo"], v["s_hi"], v["r_lo"], v["r_hi"]) for v in views])
        self._faces(_lib.LPA_HALO_UNPACK_CURRENT)(h["r_lo"] if self.comm.has_left else None,
                                                  h["r_hi"] if self.comm.has_right else None)
        check(self.L.lpa_current_fold(self._g(), self.local_axes, self.stream), "lpa_current_fold")
        for i, (m, fs) in zip(idx, packed):     # (before the fold closes: its clock tick may retune the message window)
            self._mig_unpack(i, m, fs)
        self._finish_rho()

    # ---- moving window (MovingWindow callback, callback/utils.py:471-648) ---------------------------
    def remove_x_pml(self):
        """the reference drops the x layers when the window starts moving (callback/utils.py:547-553)"""
        if self.pml is None:
            return
        old = self.pml
        sides = [s_ for s_ in old.sides if s_[0] != "x"]
        if len(sides) == len(old.sides):
            return
        new = DevicePML2D(self.nx, self.ny, self.dx, self.dy, sides, self.cpml_thickness, self.device,
                          xpad=old.xpad) if sides else None
        if new is not None:      # the y layers keep their psi history
            keep = {(l["e"], l["axis"], l["start"]): l for l in old.layers}
            for l in new.layers:
                o = keep.get((l["e"], l["axis"], l["start"]))
                if o is not None:
                    l["psi_a"], l["psi_b"] = o["psi_a"], o["psi_b"]
        self.pml = new
        # open x edges without a layer: the owner bounds are the slab's own (patch.py:105-148)
        self.alo[0] = self.x0_global - self.dx / 2
        self.ahi[0] = self.x0_global + (self.nx * self.comm.size - 1) * self.dx + self.dx / 2

    def shift_window(self, ncells):
        """move every slab ``ncells`` to the right (``ncells`` < 0: to the left): what relabelling the leftmost
        (rightmost) patch column to the other end does (callback/utils.py:594-620 / :622-648, :576-585).  Surviving
        cells keep their values (the guard behind the window holds the cells that just left, like the reference's stale
        guard); on a slab chain the columns (and y-layer psi rows) that leave through a slab's trailing face travel to
        the neighbour behind it, whose leading columns and guard they become (a rotation of the neighbour ring
        by one patch width); the leading rank's new columns start from zero fields and zero psi.  Particles behind the new
        bound follow their columns; on the trailing rank they are dropped."""
        self._flush_e2()      # (a deferred E half step of run_steps / Simulation.run is completed first)
        fwd = ncells > 0
        n, g = abs(int(ncells)), self.grid
        if not 0 < n <= g.nx - g.ng:
            raise _lib.LpaError("window shift must be between 1 and nx - n_guard cells")
        ng, nx, NX = g.ng, g.nx, g.nx + 2 * g.ng
        keep = ng + nx - n
        ylayers = [l for l in self.pml.layers if l["axis"] == 1] if self.pml is not None else []
        one = lambda: torch.zeros(1, dtype=torch.float64, device=self.device)

        def to_trailing(send, recv):
            """``send`` to the slab behind me (left when the window moves right), ``recv`` from the one ahead"""
            if self.comm.size > 1:
                if fwd:
                    self.comm.exchange(send, one(), one(), recv)
                else:
                    self.comm.exchange(one(), send, recv, one())

        # ---- what leaves through the trailing face: n + ng interior columns next to it, and the same rows of the y layers'
        # psi arrays (they carry x guard rows like the fields: psi_rows)
        cols = slice(ng, ng + n + ng) if fwd else slice(nx - n, nx + ng)
        assert all(l["xpad"] == ng for l in ylayers)
        parts = [g.buf[:, cols].reshape(-1)]
        for l in ylayers:
            for k in ("psi_a", "psi_b"):
                parts.append(psi_rows(l, k, guards=True)[cols].reshape(-1))
        send = torch.cat(parts)
        recv = torch.zeros_like(send)          # stays zero on the leading rank: fresh columns
        to_trailing(send, recv)
        nf = 10 * (n + ng) * g.NY
        if fwd:
            g.buf[:, :keep] = g.buf[:, n:n + keep].clone()
            g.buf[:, keep:] = recv[:nf].view(10, n + ng, g.NY)
        else:
            g.buf[:, NX - keep:] = g.buf[:, NX - keep - n:NX - n].clone()
            g.buf[:, :n + ng] = recv[:nf].view(10, n + ng, g.NY)
        off = nf
        for l in ylayers:
            nl = l["row"]
            for k in ("psi_a", "psi_b"):
                v = psi_rows(l, k, guards=True)          # [NX][nl]: shifted exactly like the field columns
                if fwd:
                    v[:keep] = v[n:n + keep].clone()
                    v[keep:] = recv[off:off + (n + ng) * nl].view(n + ng, nl)
                else:
                    v[NX - keep:] = v[NX - keep - n:NX - n].clone()
                    v[:n + ng] = recv[off:off + (n + ng) * nl].view(n + ng, nl)
                off += (n + ng) * nl
        shift = (n if fwd else -n) * self.dx
        self.x0_global += shift
        self.x0 += shift
        g.x0 += shift
        g.c.x0 = g.x0
        self.alo[0] += shift
        self.ahi[0] += shift
        # ---- particles that are now behind the slab
        xlo, xhi = self._owner_bounds_x()
        from_ahead = self.comm.has_right if fwd else self.comm.has_left
        for sp in self.species:
            st = sp.cset
            x = st.arr("x")[: sp.n]
            gone = (x < xlo) if fwd else (x > xhi)
            if self.comm.size == 1:
                x[gone] = float("nan")
                continue
            idx = gone.nonzero().squeeze(1)          # host sync: a window shift is a rare event
            cnt = torch.tensor([float(idx.numel())], dtype=torch.float64, device=self.device)
            got = torch.zeros_like(cnt)
            to_trailing(cnt, got)
            rows_ = len(st.core) + 1                      # core attributes + id (bit pattern); the next push rewrites ex_part ...
            out = torch.empty((rows_, idx.numel()), dtype=torch.float64, device=self.device)
            out[:-1] = st.data[:, idx]
            out[-1] = st.id[idx].view(torch.float64)
            x[idx] = float("nan")
            k = int(got.item()) if from_ahead else 0
            inc = torch.empty((rows_, k), dtype=torch.float64, device=self.device)
            # zero-length messages are legal but pointless: pad to one column
            to_trailing(out.reshape(-1) if out.numel() else one(), inc.reshape(-1) if inc.numel() else one())
            if k:
                self._append_device(sp, inc[:-1], inc[-1].view(torch.int64))
        # the tiling (and the age every edge / leaver-column estimate is derived from) refers to the grid origin of
        # before the shift: re-sort before the next push, whether or not anything arrived or was injected
        for sp in self.species:
            sp.steps_since_sort = 1 << 30

    def _append_device(self, sp, data_rows, ids):
        k = int(data_rows.shape[1])
        if sp.n + k + self.arrival_area() > sp.capacity:
            # the tiling of the old set dies with it: the forced re-sort rebuilds it
            sp.reserve(int(1.5 * (sp.n + k)) + self.arrival_area())
            self._ws.pop(id(sp), None)
            sp.n_sorted, sp.tiling = 0, None
        st = sp.cset
        st.data[:, sp.n:sp.n + k] = data_rows
        st.id[sp.n:sp.n + k] = ids
        sp.n += k
        sp.steps_since_sort = 1 << 30

    def append_particles_device(self, ispec, dev):
        """``dev``: dict of device tensors (x y ux uy uz inv_gamma w id) -> loose particles + forced re-sort.
        On a slab chain appends are collective in effect: the forced re-sort re-anchors rho, and neighbouring slabs must
        do that in the same step (rho.py) -- every rank calls in the same step, with no particles if it has none."""
        sp = self.species[ispec]
        k = int(dev["x"].numel())
        if k == 0:
            if self.comm.size > 1:
                sp.steps_since_sort = 1 << 30
            return
        rows = torch.zeros((len(sp.cset.core), k), dtype=torch.float64, device=self.device)
        for i, a in enumerate(sp.cset.core):
            if a in dev:
                rows[i] = dev[a]
        self._append_device(sp, rows, dev["id"])

    def append_particles(self, ispec, host):
        """append host particles (dict of arrays: x y ux uy uz inv_gamma w [_id]) behind the stored
        ones as loose particles and force a re-sort"""
        sp = self.species[ispec]
        k = int(host["x"].size)
        if k == 0:
            if self.comm.size > 1:          # (see append_particles_device)
                sp.steps_since_sort = 1 << 30
            return
        st = sp.cset
        rows = torch.zeros((len(st.core), k), dtype=torch.float64)
        for i, a in enumerate(st.core):
            if a in host:
                rows[i] = torch.from_numpy(np.ascontiguousarray(host[a], dtype=np.float64))
        ids = torch.from_numpy(np.ascontiguousarray(host["_id"]).view(np.int64)) if "_id" in host \
            else torch.zeros(k, dtype=torch.int64)
        self._append_device(sp, rows.to(self.device), ids.to(self.device))

    # ---- hooks of FusedStepMixin (step.py: the whole stage sequence in one lpa_step call) -------------
    def _grid_struct(self):
        return self.grid.c

    def sort_due_species(self):
        return [i for i, sp in enumerate(self.species) if self.sort_due(sp)]

    def _species_entries(self, dt, with_mig=False, pushed=False):
        for sp in self.species:
            if sp.n == 0 and not with_mig:
                continue

            def after(sp=sp):
                sp.steps_since_sort += 1
                if sp.tiling is not None:
                    sp.tiling.class_init = 0

            pp = self._push_params(sp, dt)
            pc = sp.cset.cstruct(sp.n, eb=self.write_part_eb)
            ent = {"pc": pc, "tiling": None, "n_sorted": 0, "pp": pp, "overflow": None, "count": None, "after": after}
            if sp.tiling is not None and (sp.n_sorted > 0 or with_mig):
                ws = self._sort_ws(sp)
                ent.update(tiling=sp.tiling, n_sorted=sp.n_sorted, overflow=ws["overflow"], count=ws["counters"][0:1])
            if with_mig:
                ent["mig"] = mig = self._slab_species(sp, pushed)
                if "leavers" in mig and not pushed:
                    pp.leavers, pp.leaver_count = mig["leavers"].data_ptr(), mig["leaver_count"].data_ptr()
                    pp.leaver_capacity = mig["leavers"].numel()
                    pp.leave_lo, pp.leave_hi = self._owner_bounds_x()
            yield ent

    # ---- one full step in the reference's stage order (simulation/simulation.py:946-1118) ----------
    def step(self, dt, tiled=True, defer_e2=False):
        """``defer_e2``: leave the E guards to the next step (``run_steps``: nothing reads the fields in between)"""
        self._dt_hint = dt
        if tiled and self.one_call_step():
            self.step_fused(dt, defer_e2=defer_e2)
            return
        if tiled and self.can_fuse() and not self.overlap:
            self._step_segments(dt, defer_e2)
            return
        E, B = ("ex", "ey", "ez"), ("bx", "by", "bz")
        self._flush_e2()            # (the per-stage path neither defers nor doubles: complete what a fused step left)
        defer_e2 = False
        self.update_efield(0.5 * dt)
        self.sync_guard_fields(E)
        self.update_bfield(0.5 * dt)
        self.sync_guard_fields(B)
        if tiled:
            for i, sp in enumerate(self.species):
                if self.sort_due(sp):
                    self.sort(i)
        self.reset_current()
        if tiled and self.overlap and self.push_deposit_overlapped(dt):
            for i in range(len(self.species)):
                self.sync_particles(i)
        else:
            for i in range(len(self.species)):
                self.push_deposit(i, dt, tiled=tiled)
            if tiled:
                self.sync_currents_and_particles()
            else:
                self.sync_currents()
                for i in range(len(self.species)):
                    self.sync_particles(i)
        self.update_bfield(0.5 * dt)
        self.sync_guard_fields(B)
        self.update_efield(0.5 * dt)
        if not defer_e2:
            self.sync_guard_fields(E)

    def _step_segments(self, dt, defer_e2=False, laser=None):
        """a slab rank whose faces travel through torch.distributed: the kernels between two exchanges are enqueued by one
        ``lpa_step`` sub-range each, Python moves the faces in between (E1 | xchg | B1 | xchg | reset + push | J, rho and
        leavers | B2 | xchg + the jx plane of rho | E2 | xchg) -- four or five message rounds per step; with ``local_b``
        (E1 | xchg | B1 + reset + push | J, rho and leavers | the jx plane of rho | B2 + E2 | xchg): three or four"""
        S = _lib
        local_b = self.local_b()        # the B sweeps advance the x guard planes themselves: no B message (step.py)
        self.step_stages(dt, S.LPA_STAGE_E1, S.LPA_STAGE_E1)       # (E half step + the local guard wrap)
        self._exchange_guards(1)
        if local_b:
            self.step_stages(dt, S.LPA_STAGE_B1, S.LPA_STAGE_PUSH)
        else:
            self.step_stages(dt, S.LPA_STAGE_B1, S.LPA_STAGE_B1)
            self._exchange_guards(2)
            self.step_stages(dt, S.LPA_STAGE_RESET, S.LPA_STAGE_PUSH)  # (sorts when due, decides the rho mode)
        self.defer_rho = not local_b    # (the jx plane rides with the B planes that follow -- or travels alone)
        try:
            self.sync_currents_and_particles()
        finally:
            self.defer_rho = False
        if laser is None:
            self.step_stages(dt, S.LPA_STAGE_B2, S.LPA_STAGE_E2 if local_b else S.LPA_STAGE_B2_GUARD, defer_e2 and local_b)
        else:
            self.step_stages(dt, S.LPA_STAGE_B2, S.LPA_STAGE_B2)
            laser(self, dt)
            self.step_stages(dt, S.LPA_STAGE_B2_GUARD, S.LPA_STAGE_E2 if local_b else S.LPA_STAGE_B2_GUARD, defer_e2 and local_b)
        if not local_b:
            self._exchange_guards(2)                                    # (+ the jx plane; completes rho)
            self.step_stages(dt, S.LPA_STAGE_E2, S.LPA_STAGE_E2, defer_e2)  # (deferred: nothing is launched, the next E1 doubles)
        if not defer_e2:
            self._exchange_guards(1)

    def _exchange_guards(self, which):
        """the slab-to-slab half of sync_guard_fields (the local wrap has run): pack, exchange, unpack"""
        h = self._halo_views(3 * self.ng * self.grid.NY)
        left, right = self.comm.has_left, self.comm.has_right
        self._faces(_lib.LPA_HALO_PACK_GUARD_SRC, which)(h["s_lo"] if left else None, h["s_hi"] if right else None)
        rho_msg = self._rho_message()
        if rho_msg is not None:
            self.comm.exchange_many([(h["s_lo"], h["s_hi"], h["r_lo"], h["r_hi"]), rho_msg])
        else:
            self.comm.exchange(h["s_lo"], h["s_hi"], h["r_lo"], h["r_hi"])
        self._faces(_lib.LPA_HALO_UNPACK_GUARD, which)(h["r_lo"] if left else None, h["r_hi"] if right else None)
        self._complete_rho()

    def _surplus_message(self, surplus):
        return (f"migration message overflow: {surplus} leaver-steps beyond migrate_capacity={self.migrate_capacity} "
                "(a leaver that did not fit stays outside the slab, deposits through the torus wrap and is counted "
                "again every step until it leaves; raise migrate_capacity)")

    def check_migration(self):
        """leavers that did not fit a face message since the last sort (device counter, one host read): checked at
        every sort and by ``diagnostics()`` -- also when the sorter is disabled or its interval is long"""
        for sp in self.species:
            ws = self._ws.get(id(sp))
            if ws is not None and self.comm.size > 1:
                surplus = int(ws["counters"][3].item())
                if surplus > 0 and not self._mig_surplus(surplus):
                    raise _lib.LpaError(self._surplus_message(surplus))

    # ---- diagnostics ------------------------------------------------------------------------------
    def diagnostics(self, reduce=False):
        """dict of field energy (E, B parts), total charge, current sums, kinetic energy and live
        count per species -- local to this rank; ``reduce=True`` sums over the ranks (one all-reduce)."""
        self._flush_e2()
        st = self.stream
        self.check_migration()
        self._diag.zero_()
        check(self.L.lpa_diag_fields(self._g(), self.eps0, self.mu0, self._diag.data_ptr(), st), "diag")
        out = {}
        f = self._diag.cpu().numpy().copy()
        out.update(energy_e=f[0], energy_b=f[1], field_energy=f[0] + f[1], charge=f[2],
                   jx=f[3], jy=f[4], jz=f[5], kinetic=[], nalive=[])
        for sp in self.species:
            d = torch.zeros(2, dtype=torch.float64, device=self.device)
            sp.refresh_inv_gamma()
            pc = sp.cset.cstruct(sp.n)
            check(self.L.lpa_diag_particles(C.byref(pc), sp.m, d.data_ptr(), st), "diag particles")
            d = d.cpu().numpy()
            out["kinetic"].append(float(d[0]))
            out["nalive"].append(int(round(d[1])))
        return self.comm.reduce_diagnostics(out) if reduce else out
