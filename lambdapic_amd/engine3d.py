"""PicEngine3D -- device-resident 3-D step of one x-slab of a periodic box, same stage order as
``PicEngine2D``.  Particles are binned into 4 x 4 x 16-cell tiles (``lpa_sort_tiles_3d``) and pushed by the
LDS-tiled kernel (``lpa_push_deposit_tiled_3d``: E / B and J / rho of the tile staged in LDS) + the
overflow list; grids whose extents are not multiples of the tile use the global-memory form
(``lpa_push_deposit_3d``).  With more than one rank the box is cut into x-slabs (x is the slowest
index, so a face is ``ng`` contiguous y-z planes): guard planes, current folds and leaving particles
travel to the two ring neighbours exactly as in 2-D (``dist.exchange_faces``); y and z wrap locally.

Replaces, per step: ``update_efield/bfield_patches_3d`` (`core/maxwell/cpu.py:115-158`),
``sync_guard_fields_3d`` / ``sync_currents_3d`` (`core/patch/sync_fields3d.c`, MPI twins
`core/mpi/sync_fields3d.c`), ``reset_current_cpu_3d`` (`core/current/cpu3d.c:185-240`),
``unified_boris_pusher_cpu_3d`` (`core/pusher/unified/unified_pusher_3d.c:219-436`),
``sort_particles_patches_3d`` (`core/sort/cpu3d.c`) and ``sync_particles_3d``
(`core/patch/sync_particles_3d.c`, `core/mpi/sync_particles_3d.c`) for periodic boxes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, constants
from ._lib import LPA_MIG_NATTR, check, lib
from .device import restore_device, to_host
from .dist import MigrateWindowMixin, SlabComm, exchange_faces
from .engine import PicEngine2D, psi_ptr, psi_rows
from .fields import FIELD_ATTRS, from_device_layout, to_device_layout
from .rho import RhoContinuityMixin
from .step import FusedStepMixin

ATTRS3 = ("x", "y", "z", "ux", "uy", "uz", "inv_gamma", "w")
# the resident store is float64[NROWS3][capacity]: the eight attributes + the bit pattern of ParticlesBase._id
# (`core/particles.py:50-51,91-116`: a uint64 viewed as float64), carried through the tile sort
# (`core/sort/cpu3d.c:214-299` permutes every attribute), the migration message and the window shift
ID_ROW = len(ATTRS3)
NROWS3 = ID_ROW + 1
SIDES3 = ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")


class DevicePML3D:
    """CPML coefficients and psi arrays of one rank's 3-D slab (reference: per-patch ``PML`` objects,
    `core/boundary/cpml.py:23-340`; slab mapping as in oracle/cpml.py:SlabPML3D).  psi arrays are compact
    (axis 0: [layer][ny][nz], axis 1: [nx][layer][nz], axis 2: [nx][ny][layer]; the y / z layers' arrays carry ``xpad``
    extra x rows in front and behind, see ``engine.psi_ptr``)."""

    def __init__(self, n, d, sides, thickness, device, kappa_max=20.0, a_max=0.15, sigma_max=0.7, xpad=0):
        self.n, self.d, self.t = tuple(n), tuple(d), int(thickness)
        self.sides, self.device, self.xpad = set(sides), device, int(xpad)
        m, ma = 3, 1
        smax = sigma_max * constants.C_LIGHT * 0.8 * (m + 1.0) / d[0]      # cpml.py:60 (dx for every axis)
        self.host = {}
        ar = np.arange(self.t, dtype=float)
        for ax, nn in zip("xyz", self.n):
            for fld in ("e", "b"):
                self.host[fld + ax] = dict(kappa=np.ones(nn), sigma=np.zeros(nn), a=np.zeros(nn))

            def fill(fld, pos, sl, ax=ax):
                c = self.host[fld + ax]
                c["kappa"][sl] = 1 + (kappa_max - 1) * pos ** m           # cpml.py:119-125
                c["sigma"][sl] = smax * pos ** m
                c["a"][sl] = a_max * (1 - pos) ** ma

            if ax + "min" in self.sides:
                fill("e", 1.0 - ar / self.t, np.s_[:self.t])
                fill("b", 1.0 - (ar + 0.5) / self.t, np.s_[:self.t])
            if ax + "max" in self.sides:
                fill("e", 1.0 - ar[::-1] / self.t, np.s_[nn - self.t:nn])
                fill("b", 1.0 - (ar + 0.5)[::-1] / self.t, np.s_[nn - self.t - 1:nn - 1])
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.kappa = {k: dev(v["kappa"]) for k, v in self.host.items()}
        self.layers = []
        for fld in ("e", "b"):
            for axis, ax in enumerate("xyz"):
                nn = self.n[axis]
                rng = []
                if ax + "min" in self.sides:
                    rng.append((0, self.t))
                if ax + "max" in self.sides:
                    rng.append((nn - self.t, nn) if fld == "e" else (nn - self.t - 1, nn - 1))
                for s0, s1 in rng:
                    row = (s1 - s0) * int(np.prod([v for k, v in enumerate(self.n) if k not in (axis, 0)]))   # per x row
                    cells = row * (self.n[0] + 2 * self.xpad) if axis else (s1 - s0) * self.n[1] * self.n[2]
                    z = lambda: torch.zeros(cells, dtype=torch.float64, device=device)
                    self.layers.append(dict(e=fld == "e", axis=axis, key=fld + ax, start=s0, stop=s1,
                                            psi_a=z(), psi_b=z(), xpad=self.xpad if axis else 0,
                                            row=row if axis else self.n[1] * self.n[2]))
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


class PicEngine3D(RhoContinuityMixin, FusedStepMixin, MigrateWindowMixin):
    dim = 3
    DEFAULT_ORDER = _lib.LPA_ORDER_STRIPED   # the in-tile order new engines sort to (LPA_ORDER_*)

    def __init__(self, nx, ny, nz, dx, dy, dz, n_guard=3, device="cuda:0", tiled=None, sort_interval=10,
                 block_particles=4096, comm=None, migrate_capacity=32768, boundary_conditions=None,
                 cpml_thickness=6):
        """``nx`` = cells of THIS rank's slab along x; the box is ``nx * comm.size`` cells long.
        ``boundary_conditions``: {'xmin': 'pml' | 'periodic', ...} for the six faces (default: periodic)"""
        bc = dict(boundary_conditions or {k: "periodic" for k in SIDES3})
        for ax in "xyz":
            if (bc[ax + "min"] == "periodic") != (bc[ax + "max"] == "periodic"):
                raise ValueError(f"{ax}: periodic must be set on both faces")
        self.bc = bc
        self.periodic = tuple(bc[ax + "min"] == "periodic" for ax in "xyz")
        self.cpml_thickness = int(cpml_thickness)
        self.L = lib()
        # Tiles of 4 x 4 x 16 cells; a grid that is no multiple of that ends in partial tiles (the sort counts tiles
        # with ceil, the kernel's images address the padded array as a torus, so nodes beyond the grid are legal
        # addresses that no particle reads or deposits to).  Only a slab CHAIN needs whole tile columns along x: its
        # edge / interior split and arrival bookkeeping count cells in tile columns from both faces.
        self.tiled = True if tiled is None else bool(tiled)
        self.sort_interval, self.block_particles = int(sort_interval), int(block_particles)
        self.comm = comm or SlabComm(None, periodic=self.periodic[0], single=True)
        if self.comm.periodic != self.periodic[0]:
            raise ValueError("SlabComm(periodic=...) must match the x boundary condition")
        if self.comm.size > 1 and not self.tiled:
            raise ValueError("a slab decomposition needs the tile-sorted store (arrival area)")
        if self.comm.size > 1 and nx % _lib.LPA_TILE3_X:
            raise ValueError(f"a 3-D slab needs nx (cells per rank) to be a multiple of {_lib.LPA_TILE3_X}")
        self.migrate_capacity = int(migrate_capacity)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LpaError("PicEngine3D needs a GPU device; there is no CPU path")
        self.n = (int(nx), int(ny), int(nz))
        self.d = (float(dx), float(dy), float(dz))
        self.ng = int(n_guard)
        N = tuple(v + 2 * self.ng for v in self.n)
        self.N = N
        self.buf = torch.zeros((10,) + N, dtype=torch.float64, device=self.device)
        self.x0 = self.comm.rank * self.n[0] * self.d[0]
        self.Lbox = (self.n[0] * self.comm.size * self.d[0], self.n[1] * self.d[1], self.n[2] * self.d[2])
        g = _lib.lpa_grid()
        g.nx, g.ny, g.nz, g.ng = *self.n, self.ng
        g.dx, g.dy, g.dz = self.d
        g.x0, g.y0, g.z0 = self.x0, 0.0, 0.0
        for k, name in enumerate(FIELD_ATTRS):
            setattr(g, name, self.buf[k].data_ptr())
        self.c = g
        # axes wrapped locally: periodic y and z; x only when periodic AND this rank owns the whole box
        self.local_axes = (2 if self.periodic[1] else 0) | (4 if self.periodic[2] else 0) | \
                          (1 if (self.periodic[0] and self.comm.size == 1) else 0)
        # CPML layers owned by this rank: x faces only on the end ranks
        sides = [s_ for s_ in SIDES3[2:] if bc[s_] == "pml"]
        if bc["xmin"] == "pml" and self.comm.rank == 0:
            sides.append("xmin")
        if bc["xmax"] == "pml" and self.comm.rank == self.comm.size - 1:
            sides.append("xmax")
        self.pml = DevicePML3D(self.n, self.d, sides, self.cpml_thickness, self.device, xpad=self.ng) if sides else None
        # particle absorption at open faces: bounds pulled in by the layer thickness (patch.py:105-148)
        self.absorb = 0
        self.alo, self.ahi = [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]
        ntot = (self.n[0] * self.comm.size, self.n[1], self.n[2])
        for a in range(3):
            if not self.periodic[a]:
                self.absorb |= _lib.LPA_ABSORB_X << a
                t = self.cpml_thickness
                self.alo[a] = t * self.d[a] - self.d[a] / 2
                self.ahi[a] = (ntot[a] - 1 - t) * self.d[a] + self.d[a] / 2
        self.species = []
        self._id_next = {}       # per species: ids issued by this rank so far (new_ids)
        self.eps0, self.mu0 = constants.EPSILON_0, constants.MU_0
        self._diag = torch.zeros(8, dtype=torch.float64, device=self.device)
        self._halo = None
        self._side = None       # second stream: J / rho guard planes travel while the interior is pushed
        self.overlap = True
        self.overflow_sort_fraction = 0.003    # shorten a species' sort interval when its intervals end with more than
        self.min_sort_interval = 2             # this fraction on the overflow list (see PicEngine2D; single slab only)
        self.defer_crossers = True   # cell-crossers deposit in a dense second pass of the tiled kernel
        self.reuse_slots = True   # arrivals take the slots freed by leavers of their tile (lpa_free_slots)
        self.fused_cpml = True
        self._axes = {}
        # LPA_ORDER_STRIPED, or LPA_ORDER_PADDED: the leading ranks of every tile as full stripes with holes (the store
        # needs a few per cent more slots) -- every 16-lane group is then one z column of 16 different cells
        self.order = self.DEFAULT_ORDER
        # bench instrumentation: when a list, (start, end) HIP events are recorded around every launch of the
        # tiled push+deposit kernel on the stream it runs on
        self.kernel_events = None
        self._rho_init()     # rho from the continuity equation between two real deposits: see rho.py

    def _rho_available(self):
        return self.tiled

    def _rho_sort_due(self):
        return any(self.sort_due(sp) for sp in self.species)     # (an engine without species: nothing to re-deposit)

    def _rho_forced_sort_due(self):
        return any(sp["n"] and (sp["tiling"] is None or sp["since"] >= (1 << 29)) for sp in self.species)

    def sort_due(self, sp):
        """the sorter's rule: whenever the store is not tile ordered, and every ``sort_interval`` steps -- or sooner for
        a species whose last intervals ended with a long overflow list (see PicEngine2D.overflow_sort_fraction)"""
        if self.comm.size > 1:      # slab chain: one clock for all ranks and species (PicEngine2D.sort_due)
            return self._chain_clock >= self._chain_interval() or sp["since"] >= (1 << 29)
        return sp["tiling"] is None or sp["since"] >= min(self.sort_interval, sp.get("sort_interval_now", 1 << 30))

    def _first_sort_interval(self, sp):
        """see PicEngine2D._first_sort_interval (3-D tiles keep a margin of ONE cell)"""
        dt = getattr(self, "_dt_hint", None)
        n = sp["n"]
        if self.overflow_sort_fraction <= 0 or not dt or n == 0:
            return
        dta = sp["data"]
        u = [dta[k, :n] for k in (3, 4, 5)]
        live = ~torch.isnan(dta[0, :n])
        g2 = 1 + u[0] ** 2 + u[1] ** 2 + u[2] ** 2
        nl = max(int(live.sum().item()), 1)
        cells = max(float(torch.sqrt(torch.where(live, c * c / g2, torch.zeros_like(c)).sum() / nl).item())
                    * constants.C_LIGHT * dt / dd for c, dd in zip(u, self.d))
        est = int(_lib.LPA_TILE3_MARGIN / max(2.5 * cells, 1e-12)) * (2 if self.sort_lookahead else 1)
        sp["sort_interval_now"] = max(self.min_sort_interval, min(self.sort_interval, est))

    def _adapt_sort_interval(self, sp, overflow, n_sorted_before):
        """called by sort() with the overflow count of the last push of the interval that just ended"""
        if self.overflow_sort_fraction <= 0 or n_sorted_before <= 0 or sp["since"] > self.sort_interval:
            return
        now = min(sp.get("sort_interval_now", self.sort_interval), self.sort_interval)
        f = overflow / n_sorted_before
        if f > self.overflow_sort_fraction:
            cut = 1 + int(np.log(f / self.overflow_sort_fraction) / np.log(1.4))
            now = max(self.min_sort_interval, min(now, sp["since"]) - cut)
        elif f < 0.1 * self.overflow_sort_fraction:
            now = min(self.sort_interval, now + 1)
        sp["sort_interval_now"] = now


    def _rho_particle_slots(self):
        return sum(int(sp["data"].shape[1]) for sp in self.species)

    def _species_sort_interval(self, sp):
        return sp.get("sort_interval_now") if sp["n"] else None

    def _rho_last_jx_plane(self):
        return self.view("jx")[self.ng + self.n[0] - 1]

    def _rho_array(self):
        return self.view("rho")

    # ---- restart (RestartDump, `callback/restart.py:88-107`) ------------------------------------------------
    _TRANSIENT = ("L", "c", "buf", "species", "_halo", "_side", "_axes", "_diag", "_keep", "kernel_events", "_absorbed",
                  "_jx_plane", "_one", "_step_keep", "_event_pool")

    def __getstate__(self):
        """see PicEngine2D.__getstate__: fields and the slots in use of every store as host arrays, handles,
        workspaces and tilings dropped (the first push after a load re-sorts)"""
        self._flush_e2()
        torch.cuda.synchronize(self.device)
        st = {k: v for k, v in self.__dict__.items() if k not in self._TRANSIENT}
        st["device"], st["buf_host"] = str(self.device), to_host(self.buf)
        st["species_host"] = [{"q": sp["q"], "m": sp["m"], "n": sp["n"], "capacity": int(sp["data"].shape[1]),
                               "data": to_host(sp["data"][:, : sp["n"]])} for sp in self.species]
        return st

    def __setstate__(self, st):
        buf, species = st.pop("buf_host"), st.pop("species_host")
        self.__dict__.update(st)
        self.device = restore_device(st["device"])
        self.L = lib()
        self.buf = torch.from_numpy(buf).to(self.device)
        g = _lib.lpa_grid()
        g.nx, g.ny, g.nz, g.ng = *self.n, self.ng
        g.dx, g.dy, g.dz = self.d
        g.x0, g.y0, g.z0 = self.x0, 0.0, 0.0
        for k, name in enumerate(FIELD_ATTRS):
            setattr(g, name, self.buf[k].data_ptr())
        self.c = g
        self._halo, self._side, self._axes, self.kernel_events = None, None, {}, None
        self._diag = torch.zeros(8, dtype=torch.float64, device=self.device)
        self._rho_restore()
        self.species = []
        for h in species:
            data = torch.full((NROWS3, h["capacity"]), float("nan"), dtype=torch.float64, device=self.device)
            data[:, : h["n"]] = torch.from_numpy(h["data"])
            self.species.append({"q": h["q"], "m": h["m"], "data": data, "c": self._cstruct(data, h["n"]),
                                 "n": h["n"], "n_sorted": 0, "alt": None, "tiling": None, "since": 1 << 30,
                                 "ws": None})

    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def view(self, name):
        return self.buf[FIELD_ATTRS.index(name)]

    def upload_field(self, name, wrapped):
        self.view(name).copy_(torch.from_numpy(np.ascontiguousarray(to_device_layout(wrapped, self.ng))))

    def download_field(self, name):
        return np.ascontiguousarray(from_device_layout(self.view(name).cpu().numpy(), self.ng))

    # ---- particle store -------------------------------------------------------------------------------
    def arrival_area(self):
        return self.migrate_capacity * 2 * max(self.sort_interval, 1) if self.comm.size > 1 else 0

    def add_species(self, q, m, host_particles, capacity=None):
        """upload the live particles of one host bag (ParticlesBase-like with z, ``is_dead`` honoured: dead or
        NaN-position slots are not stored -- dead = x NaN in the resident store); ``capacity`` = live particles
        this rank must be able to hold (the arrival area is added).  ``_id`` travels when the bag has one."""
        live = ~np.asarray(host_particles.is_dead, dtype=bool) & ~np.isnan(host_particles.x)
        n = int(live.sum())
        cap = max(int(capacity or n), n) + self.arrival_area()
        data = torch.full((NROWS3, max(cap, 1)), float("nan"), dtype=torch.float64, device=self.device)
        if n:
            data[:ID_ROW, :n] = torch.from_numpy(np.stack([getattr(host_particles, a)[live] for a in ATTRS3])).to(self.device)
        ids = None
        if n and getattr(host_particles, "_id", None) is not None:
            ids = torch.from_numpy(np.ascontiguousarray(host_particles._id[live]).view(np.int64)).to(self.device)
        return self.add_species_device(q, m, data, n, ids=ids)

    def new_ids(self, ispec, k):
        """``k`` fresh ids of species ``ispec`` (device int64): rank in the bits above 50 like the reference
        (`core/particles.py:91-116`), below it ONE running count per rank and species (``PicEngine2D`` /
        ``Simulation._next_ids`` use the same layout), so loading, injection and appends never collide"""
        start = self._id_next.get(ispec, 0)
        if start + k >= 1 << 50:
            raise OverflowError("particle id counter exceeds 50 bits")
        self._id_next[ispec] = start + int(k)
        return torch.arange(start, start + int(k), dtype=torch.int64, device=self.device) + (self.comm.rank << 50)

    def add_species_device(self, q, m, data, n, ids=None):
        """``data``: device tensor [NROWS3][capacity] (ATTRS3 order + the id row), the first ``n`` columns in
        use; an [8][capacity] tensor is accepted and gets its id row here.  ``ids`` (device int64[n]) overrides
        the id row; without either the engine issues fresh ids (``new_ids``)."""
        ispec = len(self.species)
        if data.shape[0] == ID_ROW:
            full = torch.empty((NROWS3, data.shape[1]), dtype=torch.float64, device=self.device)
            full[:ID_ROW] = data
            data, have_ids = full, False
        elif data.shape[0] == NROWS3:
            have_ids = True
        else:
            raise ValueError(f"particle data must have {ID_ROW} or {NROWS3} rows")
        if ids is not None:
            data[ID_ROW, :n] = ids.to(self.device).view(torch.float64)
        elif not have_ids:
            data[ID_ROW, :n] = self.new_ids(ispec, n).view(torch.float64)
        self.species.append({"q": float(q), "m": float(m), "data": data, "c": self._cstruct(data, n), "n": int(n),
                             "n_sorted": 0, "alt": None, "tiling": None, "since": 0, "ws": None})
        return ispec

    @staticmethod
    def _cstruct(data, n):
        p = _lib.lpa_particles()
        p.n = int(n)
        for k, a in enumerate(ATTRS3):
            setattr(p, a, data[k].data_ptr())
        for k in range(6):
            p.part_eb[k] = None
        p.id, p.is_dead = data[ID_ROW].data_ptr(), None
        return p

    def ids(self, i):
        """device int64 view of the id row of species ``i`` (slots [0, n))"""
        sp = self.species[i]
        return sp["data"][ID_ROW, : sp["n"]].view(torch.int64)

    def download_species(self, i):
        """host arrays of the LIVE particles (device order): ATTRS3 + ``_id`` (float64 bit pattern, as in
        ParticlesBase)"""
        sp = self.species[i]
        d = sp["data"][:, : sp["n"]]
        d = d[:, ~torch.isnan(d[0])].cpu().numpy()
        out = {a: d[k] for k, a in enumerate(ATTRS3)}
        out["_id"] = np.ascontiguousarray(d[ID_ROW])
        return out

    def _g(self):
        return C.byref(self.c)

    def _ws(self, sp):
        if sp["ws"] is None:
            cap = sp["data"].shape[1]
            if sp.get("alt") is None or sp["alt"].shape != sp["data"].shape:
                sp["alt"] = torch.empty_like(sp["data"])
            nbytes = self.L.lpa_sort_workspace_bytes_ranks(self._g(), cap, sp.get("stripe_ranks", 0))
            cnt = torch.zeros(8, dtype=torch.int32, device=self.device)   # 0: overflow, 1: arrivals, 2: edge overflow, 3: surplus, 4: leavers
            sp["ws"] = {"sort": torch.zeros(nbytes, dtype=torch.uint8, device=self.device),
                        "overflow": torch.empty(max(cap, 1), dtype=torch.int32, device=self.device),
                        "counters": cnt, "count": cnt[0:1], "tiling": _lib.lpa_tiling(), "mig": None}
        return sp["ws"]

    deep_tail_fraction = 0.005   # see PicEngine2D.sort
    sort_lookahead = True
    from .engine import PicEngine2D as _E2
    sort_lookahead_cold = _E2.sort_lookahead_cold
    _sort_ahead = _E2._sort_ahead
    del _E2

    def fuse_worthwhile(self):
        """One launch for all species (a workgroup per TILE: one E / B staging for every species; also the faster form for
        a single species: 64^3 cells at 64 per cell 1.22 against 1.36 ms) pays when the tiles are many or shallow.  Few
        deep tiles -- 64 tiles of 260 000 particles -- are 64 workgroups on 256 CUs that way; the per-species launches split
        a tile into work blocks (5.1 -> 3.0 ms per step)."""
        live = [sp for sp in self.species if sp["tiling"] is not None and sp["n_sorted"] > 0]
        return bool(live) and all(sp.get("tiles_in_use", 0) >= 1024 or sp.get("n_blocks", 0) <= 2 * sp.get("tiles_in_use", 0) for sp in live)

    def sort(self, i, _again=False):
        """tile-bin species ``i`` (replaces sort_particles_patches_3d, core/sort/cpu3d.c); drops dead
        slots; one host sync for the live count.  Re-sizes the striped ranks for a store that is deeper than its mean
        over all tiles says (PicEngine2D.sort)"""
        sp = self.species[i]
        ws = self._ws(sp)
        forced = _again or sp["tiling"] is None or sp["since"] >= (1 << 29)
        cap = sp["data"].shape[1]
        src, dst = self._cstruct(sp["data"], sp["n"]), self._cstruct(sp["alt"], cap)
        # a re-sort: the first n_sorted slots are the previous sort's result (lpa_tiling.prefix_hint)
        ws["tiling"].prefix_hint = int(sp["n_sorted"]) if sp["tiling"] is not None else 0
        ws["tiling"].stripe_ranks = sp.get("stripe_ranks", 0)
        ahead = self._sort_ahead(sp)
        check(self.L.lpa_sort_tiles_ahead_3d(self._g(), C.byref(src), C.byref(dst), ws["sort"].data_ptr(),
                                             ws["sort"].numel(), self.block_particles, self.order,
                                             C.byref(ws["tiling"]), ahead, self.stream), "lpa_sort_tiles_3d")
        sp["sort_ahead_used"] = ahead
        n_live, deepest, tail, sp["tiles_in_use"], sp["n_blocks"] = _lib.sort_result(self.L, ws["sort"], True)
        area = self.arrival_area()
        cnts = ws["counters"].tolist()
        self._mig_sample(ws["mig"])      # (the counts of the last step's face messages: dist.MigrateWindowMixin)
        if _again:
            pass
        elif sp["tiling"] is not None:
            self._adapt_sort_interval(sp, cnts[0] + cnts[2], sp["n_sorted"])   # the overflow lists of the last push (interior + edge part)
        else:
            self._first_sort_interval(sp)
        if cnts[1] > area:
            raise _lib.LpaError("arrival area overflow (raise migrate_capacity)")
        if cnts[3] > 0 and not self._mig_surplus(cnts[3]):
            raise _lib.LpaError(self._surplus_message(cnts[3]))
        if n_live + area > cap:
            raise _lib.LpaError(f"particle capacity {cap} < live {n_live} + arrival area {area}")
        sp["data"], sp["alt"] = sp["alt"], sp["data"]
        if area:
            sp["data"][0, n_live:n_live + area] = float("nan")
        sp["n_sorted"], sp["n"] = n_live, n_live + area
        sp["c"] = self._cstruct(sp["data"], sp["n"])
        ws["counters"].zero_()
        ws["tiling"].n_sorted = n_live
        # the store that was just sorted FROM is idle until the next sort: scratch for the dense second pass of
        # the tiled kernel (particles that changed cell park their 8 attributes there)
        for c in range(8):
            ws["tiling"].scratch[c] = sp["alt"][c].data_ptr() if self.defer_crossers else None
        sp["tiling"] = ws["tiling"]
        sp["since"] = 0
        self._rho_sorted(forced)
        self._reset_free_slots(ws)
        used, want = ws["tiling"].stripe_ranks, min(_lib.LPA_MAX_STRIPE_RANKS, deepest + deepest // 4)
        if not _again and want > used and used < sp.get("stripe_ranks_limit", 1 << 30) and \
                self.order == _lib.LPA_ORDER_STRIPED and tail > self.deep_tail_fraction * max(n_live, 1):
            sp["stripe_ranks"] = want
            sp["ws"], sp["tiling"] = None, None
            self.sort(i, _again=True)
            if sp["ws"]["tiling"].stripe_ranks < want:             # the table would not fit (1 GiB): do not ask again
                sp["stripe_ranks_limit"] = sp["ws"]["tiling"].stripe_ranks

    FREE_SLOT_DEPTH = 64

    def _reset_free_slots(self, ws):
        """see PicEngine2D._reset_free_slots"""
        ws["fs"] = None
        t = ws["tiling"]
        if self.comm.size == 1:
            return
        cols = int(np.ceil((1.5 * self.sort_interval + 2) / _lib.LPA_TILE3_X))     # (age + the sort's look-ahead)
        if 2 * cols > t.tiles_x:
            return
        n = 2 * cols * t.tiles_y * t.tiles_z
        if ws.get("fs_count") is None or ws["fs_count"].numel() != n:
            ws["fs_count"] = torch.zeros(n, dtype=torch.int32, device=self.device)
            ws["fs_slot"] = torch.empty(n * self.FREE_SLOT_DEPTH, dtype=torch.int32, device=self.device)
        else:
            ws["fs_count"].zero_()
        fs = _lib.lpa_free_slots()
        fs.count, fs.slot = ws["fs_count"].data_ptr(), ws["fs_slot"].data_ptr()
        fs.edge_cols, fs.depth = cols, self.FREE_SLOT_DEPTH
        ws["fs"] = fs

    # ---- guards / currents between slabs (sync_guard_fields_3d, sync_currents_3d) -------------------
    def _halo_bufs(self):
        if self._halo is None:
            # 6 components: a guard sync of E and B together is the largest message
            n = 6 * self.ng * self.N[1] * self.N[2]
            mk = lambda: torch.empty(n, dtype=torch.float64, device=self.device)
            self._halo = {"s_lo": mk(), "s_hi": mk(), "r_lo": mk(), "r_hi": mk()}
        return self._halo

    def _halo_views(self, ncomp):
        """the first ``ncomp`` components' worth of the four face buffers (the C ABI takes no length: check here)"""
        n = ncomp * self.ng * self.N[1] * self.N[2]
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

    def sync_guard_fields(self, which):
        """``which``: 1 = E, 2 = B"""
        self._flush_e2()
        st = self.stream
        check(self.L.lpa_guard_wrap(self._g(), which, self.local_axes, st), "lpa_guard_wrap")
        if self.comm.size > 1:
            h = self._halo_views(3 * bin(which).count("1"))
            if self._rho_message() is not None:      # a deferred rho update: its jx plane rides with these planes (rho.py)
                self._exchange_guards(which, h)
                return
            exchange_faces(
                self.comm,
                lambda side, b: check(self.L.lpa_halo_pack_guard_src(self._g(), which, side, b.data_ptr(), st),
                                      "lpa_halo_pack_guard_src"),
                lambda side, b: check(self.L.lpa_halo_unpack_guard(self._g(), which, side, b.data_ptr(), st),
                                      "lpa_halo_unpack_guard"),
                h, pack2=self._faces(_lib.LPA_HALO_PACK_GUARD_SRC, which),
                unpack2=self._faces(_lib.LPA_HALO_UNPACK_GUARD, which))      # whole planes travel: the sender wrapped their y / z guard strips already

    def sync_currents(self):
        st = self.stream
        if self.comm.size > 1:
            h = self._halo_views(4)
            exchange_faces(
                self.comm,
                lambda side, b: check(self.L.lpa_halo_pack_current(self._g(), side, b.data_ptr(), st),
                                      "lpa_halo_pack_current"),
                lambda side, b: check(self.L.lpa_halo_unpack_current(self._g(), side, b.data_ptr(), st),
                                      "lpa_halo_unpack_current"),
                h, pack2=self._faces(_lib.LPA_HALO_PACK_CURRENT), unpack2=self._faces(_lib.LPA_HALO_UNPACK_CURRENT))
        check(self.L.lpa_current_fold(self._g(), self.local_axes, st), "lpa_current_fold")
        self._finish_rho()

    def _owner_bounds_x(self):
        return self.x0 - self.d[0] / 2, self.x0 + (self.n[0] - 1) * self.d[0] + self.d[0] / 2

    def _slab_species(self, sp, pushed):
        """see PicEngine2D._slab_species"""
        ws, cap = self._ws(sp), self.migrate_capacity
        if ws["mig"] is None:
            mk = lambda: torch.zeros(1 + LPA_MIG_NATTR * cap, dtype=torch.float64, device=self.device)
            ws["mig"] = {"s_lo": mk(), "s_hi": mk(), "r_lo": mk(), "r_hi": mk()}
        ahead = sp.get("sort_ahead_used", 0.0) * constants.C_LIGHT / self.d[0]      # (see PicEngine2D.leaver_columns)
        age = sp["since"] + (0 if pushed else 1)
        cols = int(np.ceil((age + 1 + ahead) / _lib.LPA_TILE3_X)) if sp["tiling"] is not None else 0
        if not (cols and 2 * cols <= self.n[0] // _lib.LPA_TILE3_X):
            cols = 0
        fs = ws.get("fs") if (self.reuse_slots and cols) else None
        if fs is not None and cols > fs.edge_cols:
            fs = None
        out = {"bufs": ws["mig"], "cursor": ws["counters"][1:2], "surplus": ws["counters"][3:4], "fs": fs,
               "area": self.arrival_area(), "cols": cols}
        if self.overlap and self.native_slab():
            if "overflow_edge" not in ws:
                ws["overflow_edge"] = torch.empty_like(ws["overflow"])
            out.update(overflow_edge=ws["overflow_edge"], edge_count=ws["counters"][2:3])
        if self.leaver_lists and self.native_slab() and sp["tiling"] is not None:      # (see PicEngine2D._slab_species)
            if ws.get("leavers") is None:
                ws["leavers"] = torch.empty(2 * cap, dtype=torch.int64, device=self.device)
            out.update(leavers=ws["leavers"], leaver_count=ws["counters"][4:5], fs=ws.get("fs") if self.reuse_slots else None)
        return out

    leaver_lists = True

    def n_x_local(self):
        return self.n[0]

    def _slab_fill(self, slab):
        """the slab section of an lpa_step descriptor (step.py)"""
        slab.xlo, slab.xhi = self._owner_bounds_x()
        slab.shift_lo, slab.shift_hi = self.comm.arrival_shift(self.Lbox[0])
        slab.migrate_capacity = self.migrate_window       # (the message and its SoA stride: what travels of the buffers)
        h = self._halo_views(4)
        slab.cur_r_lo, slab.cur_r_hi = h["r_lo"].data_ptr(), h["r_hi"].data_ptr()
        slab.rho_exchange = int(self.rho_continuity and self._rho_available()) * (2 if self.local_b() else 1)
        if slab.rho_exchange:
            self._jx_plane_bufs()
            slab.jx_left_plane = self._jx_plane.data_ptr()
        dt = getattr(self, "_dt_hint", 0.0)
        slab.overlap_cols = self.edge_columns(dt) if (self.overlap and dt > 0) else 0
        return h

    def _mig_pack(self, i):
        """leavers of species ``i`` into its two face messages (count in band); returns the bookkeeping dict"""
        sp = self.species[i]
        mig = self._slab_species(sp, pushed=True)
        m, cap, st, cols, fs = mig["bufs"], self.migrate_window, self.stream, mig["cols"], mig["fs"]
        xlo, xhi = self._owner_bounds_x()
        surplus = mig["surplus"].data_ptr()     # leavers beyond migrate_capacity (checked at the next sort)
        if cols:   # only the edge tile columns + loose particles
            check(self.L.lpa_migrate_pack_edges_x(C.byref(sp["c"]), C.byref(sp["tiling"]), cols, xlo, xhi,
                                                  m["s_lo"].data_ptr(), m["s_hi"].data_ptr(), cap,
                                                  C.byref(fs) if fs is not None else None, surplus, st),
                  "lpa_migrate_pack_edges_x")
        else:
            check(self.L.lpa_migrate_pack_x(C.byref(sp["c"]), xlo, xhi, m["s_lo"].data_ptr(), m["s_hi"].data_ptr(),
                                            cap, surplus, st), "lpa_migrate_pack_x")
        return mig

    def _mig_unpack(self, i, mig):
        sp = self.species[i]
        m, cap, st, fs = mig["bufs"], self.migrate_window, self.stream, mig["fs"]
        cur, area = mig["cursor"].data_ptr(), mig["area"]
        if not self.comm.has_left:
            m["r_lo"][:1].zero_()    # open face: nothing arrives
        if not self.comm.has_right:
            m["r_hi"][:1].zero_()
        shift_lo, shift_hi = self.comm.arrival_shift(self.Lbox[0])
        for buf, shift in ((m["r_lo"], shift_lo), (m["r_hi"], shift_hi)):
            if fs is not None:
                check(self.L.lpa_migrate_unpack_tiled(C.byref(sp["c"]), self._g(), C.byref(sp["tiling"]), C.byref(fs),
                                                      sp["n_sorted"], area, cur, buf.data_ptr(), cap, shift, st),
                      "lpa_migrate_unpack_tiled")
            else:
                check(self.L.lpa_migrate_unpack(C.byref(sp["c"]), sp["n_sorted"], area, cur, buf.data_ptr(), cap,
                                                shift, st), "lpa_migrate_unpack")

    def sync_particles(self, i):
        """leavers travel to the ring neighbours in one fixed-size message per face (count in band)"""
        if self.comm.size == 1:
            return
        mig = self._mig_pack(i)
        m = self._mig_views(mig["bufs"])
        self.comm.exchange(m["s_lo"], m["s_hi"], m["r_lo"], m["r_hi"])
        self._mig_unpack(i, mig)

    def sync_currents_and_particles(self):
        """the J / rho guard fold and the migration of every species in ONE message round (see PicEngine2D)"""
        if self.comm.size == 1:
            self.sync_currents()
            return
        h = self._halo_views(4)
        left, right = self.comm.has_left, self.comm.has_right
        self._faces(_lib.LPA_HALO_PACK_CURRENT)(h["s_lo"] if left else None, h["s_hi"] if right else None)
        packed = [self._mig_pack(i) for i in range(len(self.species))]
        self.comm.exchange_many([(h["s_lo"], h["s_hi"], h["r_lo"], h["r_hi"])] +
                                [tuple(self._mig_views(mg["bufs"])[k] for k in ("s_lo", "s_hi", "r_lo", "r_hi")) for mg in packed])
        self._faces(_lib.LPA_HALO_UNPACK_CURRENT)(h["r_lo"] if left else None, h["r_hi"] if right else None)
        check(self.L.lpa_current_fold(self._g(), self.local_axes, self.stream), "lpa_current_fold")
        for i, mg in enumerate(packed):         # (before the fold closes: its clock tick may retune the message window)
            self._mig_unpack(i, mg)
        self._finish_rho()

    def _exchange_guards(self, which, h=None):
        """the slab-to-slab half of sync_guard_fields (the local wrap has run): pack, exchange, unpack"""
        h = h or self._halo_views(3 * bin(which).count("1"))
        left, right = self.comm.has_left, self.comm.has_right
        self._faces(_lib.LPA_HALO_PACK_GUARD_SRC, which)(h["s_lo"] if left else None, h["s_hi"] if right else None)
        rho_msg = self._rho_message()
        if rho_msg is not None:
            self.comm.exchange_many([(h["s_lo"], h["s_hi"], h["r_lo"], h["r_hi"]), rho_msg])
        else:
            self.comm.exchange(h["s_lo"], h["s_hi"], h["r_lo"], h["r_hi"])
        self._faces(_lib.LPA_HALO_UNPACK_GUARD, which)(h["r_lo"] if left else None, h["r_hi"] if right else None)
        self._complete_rho()

    # ---- Maxwell with CPML layers (update_e/bfield_cpml_patches_3d, cpml.py:477-530) --------------------
    def update_efield(self, dt):
        self._flush_e2()      # (a deferred E half step of run_steps / Simulation.run is completed first)
        g, st = self._g(), self.stream
        if self.pml is None:
            check(self.L.lpa_fdtd_e_3d(g, dt, self.eps0, st), "lpa_fdtd_e_3d")
            return
        if self.fused_cpml:     # kappa sweep + every layer's psi recursion in one launch
            a = self._cpml_axes(True, dt)
            check(self.L.lpa_fdtd_e_cpml_fused_3d(g, dt, self.eps0, C.byref(a[0]), C.byref(a[1]), C.byref(a[2]), st),
                  "lpa_fdtd_e_cpml_fused_3d")
            return
        k = self.pml.kappa
        check(self.L.lpa_fdtd_e_cpml_3d(g, dt, self.eps0, k["ex"].data_ptr(), k["ey"].data_ptr(),
                                        k["ez"].data_ptr(), st), "lpa_fdtd_e_cpml_3d")
        self._psi(True, dt)

    def update_bfield(self, dt):
        self._flush_e2()      # (a deferred E half step of run_steps / Simulation.run is completed first)
        g, st = self._g(), self.stream
        if self.pml is None:
            check(self.L.lpa_fdtd_b_3d(g, dt, st), "lpa_fdtd_b_3d")
            return
        if self.fused_cpml:
            a = self._cpml_axes(False, dt)
            check(self.L.lpa_fdtd_b_cpml_fused_3d(g, dt, C.byref(a[0]), C.byref(a[1]), C.byref(a[2]), st),
                  "lpa_fdtd_b_cpml_fused_3d")
            return
        k = self.pml.kappa
        check(self.L.lpa_fdtd_b_cpml_3d(g, dt, k["bx"].data_ptr(), k["by"].data_ptr(), k["bz"].data_ptr(), st),
              "lpa_fdtd_b_cpml_3d")
        self._psi(False, dt)

    def _cpml_axes(self, efield, dt):
        """lpa_cpml_axis descriptors (x, y, z) of the E or B update for this dt, cached per PML object"""
        key = (id(self.pml), bool(efield), dt)
        if key not in self._axes:
            out = []
            for axis, ax in enumerate("xyz"):
                k = ("e" if efield else "b") + ax
                d = _lib.lpa_cpml_axis()
                d.kappa = self.pml.kappa[k].data_ptr()
                b, cc = self.pml.coef(k, dt, self.d[axis])
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
            self._axes[key] = (tuple(out), self.pml)
        return self._axes[key][0]

    def _psi(self, efield, dt):
        for ly in self.pml.layers:
            if ly["e"] != efield:
                continue
            b, cc = self.pml.coef(ly["key"], dt, self.d[ly["axis"]])
            check(self.L.lpa_cpml_psi_3d(self._g(), int(efield), ly["axis"], ly["start"], ly["stop"], dt,
                                         b.data_ptr(), cc.data_ptr(), psi_ptr(ly, "psi_a"),
                                         psi_ptr(ly, "psi_b"), self.stream), "lpa_cpml_psi_3d")

    def laser_inject(self, ey_source, ez_source, dt):
        """``ey_source, ez_source``: [ny][nz] source fields on the x-min boundary at the current time
        (Laser.__call__ at stage '_laser', callback/laser.py:109-137,218-238); only the rank that owns
        the x-min layer injects"""
        if self.pml is None or "xmin" not in self.pml.sides:
            return
        t = self.cpml_thickness
        iy0, iy1 = (t, self.n[1] - t) if self.bc["ymin"] == "pml" else (0, self.n[1])
        iz0, iz1 = (t, self.n[2] - t) if self.bc["zmin"] == "pml" else (0, self.n[2])
        to = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(
            self.device, torch.float64).contiguous()
        ey, ez = to(ey_source), to(ez_source)
        check(self.L.lpa_laser_inject_3d(self._g(), t + 2, dt, self.eps0, iy0, iy1, iz0, iz1, ey.data_ptr(),
                                         ez.data_ptr(), self.stream), "lpa_laser_inject_3d")
        self._keep = (ey, ez)

    def laser_inject_sep(self, pc, ps, k4, dt):
        """factorised sources (device tensors ``pc``, ``ps`` [ny][nz], four floats ``k4``), see PicEngine2D"""
        if self.pml is None or "xmin" not in self.pml.sides:
            return
        t = self.cpml_thickness
        iy0, iy1 = (t, self.n[1] - t) if self.bc["ymin"] == "pml" else (0, self.n[1])
        iz0, iz1 = (t, self.n[2] - t) if self.bc["zmin"] == "pml" else (0, self.n[2])
        kk = (C.c_double * 4)(*[float(v) for v in k4])
        check(self.L.lpa_laser_inject_sep_3d(self._g(), t + 2, dt, self.eps0, iy0, iy1, iz0, iz1, pc.data_ptr(),
                                             ps.data_ptr(), kk, self.stream), "lpa_laser_inject_sep_3d")

    # ---- moving window (MovingWindow callback, callback/utils.py:471-648; 3-D patch relabelling :705-730) ----
    def remove_x_pml(self):
        """the reference drops the x layers when the window starts moving (callback/utils.py:547-553)"""
        if self.pml is None:
            return
        old = self.pml
        sides = [s_ for s_ in old.sides if s_[0] != "x"]
        if len(sides) == len(old.sides):
            return
        new = DevicePML3D(self.n, self.d, sides, self.cpml_thickness, self.device, xpad=old.xpad) if sides else None
        if new is not None:      # the y / z layers keep their psi history
            keep = {(l["e"], l["axis"], l["start"]): l for l in old.layers}
            for l in new.layers:
                o = keep.get((l["e"], l["axis"], l["start"]))
                if o is not None:
                    l["psi_a"], l["psi_b"] = o["psi_a"], o["psi_b"]
        self.pml = new
        ntot = self.n[0] * self.comm.size
        xg = self.x0 - self.comm.rank * self.n[0] * self.d[0]          # current global origin
        self.alo[0] = xg - self.d[0] / 2
        self.ahi[0] = xg + (ntot - 1) * self.d[0] + self.d[0] / 2

    def shift_window(self, ncells):
        """move every slab ``ncells`` to the right, or to the left when negative (see PicEngine2D.shift_window): field
        columns and the psi rows of the y / z layers that leave through a slab's trailing face travel to the neighbour
        behind it; the leading rank's new columns start from zero; particles behind the new bound follow or are dropped"""
        self._flush_e2()      # (a deferred E half step of run_steps / Simulation.run is completed first)
        fwd = ncells > 0
        n, ng, nx = abs(int(ncells)), self.ng, self.n[0]
        if not 0 < n <= nx - ng:
            raise _lib.LpaError("window shift must be between 1 and nx - n_guard cells")
        NX = nx + 2 * ng
        keep = ng + nx - n
        lay = [l for l in self.pml.layers if l["axis"] != 0] if self.pml is not None else []
        one = lambda: torch.zeros(1, dtype=torch.float64, device=self.device)

        def rows(l, k):          # psi array of a y / z layer as [NX][...]: x guard rows included, shifted like the fields
            return psi_rows(l, k, guards=True)

        def to_trailing(send, recv):
            if self.comm.size > 1:
                if fwd:
                    self.comm.exchange(send, one(), one(), recv)
                else:
                    self.comm.exchange(one(), send, recv, one())

        cols = slice(ng, ng + n + ng) if fwd else slice(nx - n, nx + ng)
        assert all(l["xpad"] == ng for l in lay)
        parts = [self.buf[:, cols].reshape(-1)]
        for l in lay:
            for k in ("psi_a", "psi_b"):
                parts.append(rows(l, k)[cols].reshape(-1))
        send = torch.cat(parts)
        recv = torch.zeros_like(send)
        to_trailing(send, recv)
        plane = self.N[1] * self.N[2]
        nf = 10 * (n + ng) * plane
        if fwd:
            self.buf[:, :keep] = self.buf[:, n:n + keep].clone()
            self.buf[:, keep:] = recv[:nf].view(10, n + ng, self.N[1], self.N[2])
        else:
            self.buf[:, NX - keep:] = self.buf[:, NX - keep - n:NX - n].clone()
            self.buf[:, :n + ng] = recv[:nf].view(10, n + ng, self.N[1], self.N[2])
        off = nf
        for l in lay:
            for k in ("psi_a", "psi_b"):
                v = rows(l, k)
                w = v.shape[1]
                if fwd:
                    v[:keep] = v[n:n + keep].clone()
                    v[keep:] = recv[off:off + (n + ng) * w].view(n + ng, w)
                else:
                    v[NX - keep:] = v[NX - keep - n:NX - n].clone()
                    v[:n + ng] = recv[off:off + (n + ng) * w].view(n + ng, w)
                off += (n + ng) * w
        shift = (n if fwd else -n) * self.d[0]
        self.x0 += shift
        self.c.x0 = self.x0
        self.alo[0] += shift
        self.ahi[0] += shift
        xlo, xhi = self._owner_bounds_x()
        from_ahead = self.comm.has_right if fwd else self.comm.has_left
        for sp in self.species:
            x = sp["data"][0, : sp["n"]]
            gone = (x < xlo) if fwd else (x > xhi)
            if self.comm.size == 1:
                x[gone] = float("nan")
                continue
            idx = gone.nonzero().squeeze(1)          # host sync: a window shift is a rare event
            cnt = torch.tensor([float(idx.numel())], dtype=torch.float64, device=self.device)
            got = torch.zeros_like(cnt)
            to_trailing(cnt, got)
            out = sp["data"][:, idx].contiguous()
            x[idx] = float("nan")
            k = int(got.item()) if from_ahead else 0
            inc = torch.empty((NROWS3, k), dtype=torch.float64, device=self.device)
            to_trailing(out.reshape(-1) if out.numel() else one(), inc.reshape(-1) if inc.numel() else one())
            if k:
                self.append_device(self.species.index(sp), inc)
        # the tiling (and the age every edge / leaver-column estimate is derived from) refers to the grid origin of
        # before the shift: re-sort before the next push, whether or not anything arrived or was injected
        for sp in self.species:
            sp["since"] = 1 << 30
        self._anchor_pending = True

    def append_device(self, i, rows, ids=None):
        """append particles (device tensor [NROWS3][k]: ATTRS3 order + id row; or [8][k] with ``ids`` int64[k],
        fresh ids when omitted) behind the stored ones as loose particles and force a re-sort; grows the store
        when needed.  On a slab chain every rank calls in the same step (with no columns if it has none): the forced
        re-sort re-anchors rho, which neighbouring slabs must do together (rho.py)"""
        sp = self.species[i]
        k = int(rows.shape[1])
        if k == 0:
            if self.comm.size > 1:
                sp["since"] = 1 << 30
            return
        if rows.shape[0] == ID_ROW:
            full = torch.empty((NROWS3, k), dtype=torch.float64, device=self.device)
            full[:ID_ROW] = rows
            full[ID_ROW] = (ids if ids is not None else self.new_ids(i, k)).to(self.device).view(torch.float64)
            rows = full
        elif ids is not None:
            rows = rows.clone()
            rows[ID_ROW] = ids.to(self.device).view(torch.float64)
        cap = sp["data"].shape[1]
        if sp["n"] + k + self.arrival_area() > cap:
            new = torch.full((NROWS3, int(1.5 * (sp["n"] + k)) + self.arrival_area()), float("nan"),
                             dtype=torch.float64, device=self.device)
            new[:, : sp["n"]] = sp["data"][:, : sp["n"]]
            sp["data"], sp["alt"], sp["ws"], sp["tiling"], sp["n_sorted"] = new, None, None, None, 0
        sp["data"][:, sp["n"]:sp["n"] + k] = rows
        sp["n"] += k
        sp["c"] = self._cstruct(sp["data"], sp["n"])
        sp["since"] = 1 << 30
        self._anchor_pending = True

    def reset_current(self):
        """`CurrentDeposition3D.reset` (core/current/cpu3d.c:185-240); also decides this step's rho mode (rho.py): a
        real deposit zeroes jx jy jz rho, a continuity step zeroes the currents only"""
        self._begin_deposit_step()

    # ---- one step --------------------------------------------------------------------------------------
    def push_deposit(self, i, dt, part=_lib.LPA_PART_ALL, edge_cols=0):
        """``part``: LPA_PART_EDGE = edge tile columns + overflow list + arrival area (everything that can
        deposit into the x guard planes), LPA_PART_INTERIOR = the remaining tiles (+ their overflow)"""
        L, st, g, sp = self.L, self.stream, self._g(), self.species[i]
        pp = self._push_params(sp, dt)
        if self._no_rho and self.sort_due(sp):
            raise _lib.LpaError("a store needs sorting inside a continuity step: call reset_current() first")
        self._push_flags(pp, dt, self.absorb)
        if not self.tiled:
            check(L.lpa_push_deposit_3d(g, C.byref(sp["c"]), C.byref(pp), 0, sp["n"], st), "lpa_push_deposit_3d")
            return
        if part != _lib.LPA_PART_INTERIOR and self.sort_due(sp):
            self.sort(i)
        ws = sp["ws"]
        # the edge part may run on a second stream beside the interior part: own overflow list + counter
        if part == _lib.LPA_PART_EDGE:
            if "overflow_edge" not in ws:
                ws["overflow_edge"] = torch.empty_like(ws["overflow"])
            ovf, cnt = ws["overflow_edge"], ws["counters"][2:3]
        else:
            ovf, cnt = ws["overflow"], ws["count"]
        cnt.zero_()
        timed = self.kernel_events is not None
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        check(L.lpa_push_deposit_tiled_part_3d(g, C.byref(sp["c"]), C.byref(pp), C.byref(sp["tiling"]),
                                               ovf.data_ptr(), cnt.data_ptr(), part, edge_cols,
                                               st), "lpa_push_deposit_tiled_3d")
        if timed:
            e1.record(torch.cuda.current_stream(self.device))
            self.kernel_events.append((e0, e1))
        check(L.lpa_push_deposit_list_3d(g, C.byref(sp["c"]), C.byref(pp), ovf.data_ptr(),
                                         cnt.data_ptr(), sp["n_sorted"], st), "lpa_push_deposit_list_3d")
        loose = sp["n"] - sp["n_sorted"]        # arrival area: pushed by the global kernel until the next sort
        if loose > 0 and part != _lib.LPA_PART_INTERIOR:
            check(L.lpa_push_deposit_3d(g, C.byref(sp["c"]), C.byref(pp), sp["n_sorted"], loose, st),
                  "lpa_push_deposit_3d")
        if part != _lib.LPA_PART_EDGE:
            sp["since"] += 1

    def edge_columns(self, dt):
        """see PicEngine2D.edge_columns (the sort, when due, runs inside the edge pass: age 0 then)"""
        age = max([0 if sp["since"] >= self.sort_interval else sp["since"] for sp in self.species] + [0]) + 1
        ahead = max([sp.get("sort_ahead_used", 0.0) for sp in self.species] + [0.0]) * constants.C_LIGHT / self.d[0]
        drift = constants.C_LIGHT * dt / self.d[0] * age + 5.0 + ahead   # + the 3 nodes a deposit window reaches, + 1, + 1 (2-D twin); + the sort's look-ahead
        cols = int(np.ceil(drift / _lib.LPA_TILE3_X))
        return cols if 2 * cols < self.n[0] // _lib.LPA_TILE3_X else 0

    def push_deposit_overlapped(self, dt):
        """as PicEngine2D.push_deposit_overlapped: edge tile columns first, the J / rho guard planes travel on a
        second stream while the interior tiles are pushed.  False when the slab is too thin to split."""
        cols = self.edge_columns(dt)
        if self.comm.size == 1 or not self.tiled or cols == 0:
            return False
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device, priority=-1)
        for i in range(len(self.species)):      # a sort that is due runs here, before the two parts split
            sp = self.species[i]
            if self.sort_due(sp):
                self.sort(i)
        ready, done = torch.cuda.Event(), torch.cuda.Event()
        ready.record(main)
        h = self._halo_views(4)
        # edge tiles, pack and exchange on the high-priority side stream, the interior tiles on the main stream
        # at the same time (disjoint particles; both add into J with atomics; see PicEngine2D)
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

    # ---- hooks of FusedStepMixin (step.py) -------------------------------------------------------------------
    def _grid_struct(self):
        return self.c

    def sort_due_species(self):
        if not self.tiled:
            return []
        return [i for i, sp in enumerate(self.species) if self.sort_due(sp)]

    def _push_params(self, sp, dt):
        pp = _lib.lpa_push_params()
        pp.dt, pp.q, pp.m, pp.wrap = dt, sp["q"], sp["m"], self.local_axes | self.absorb
        for a in range(3):
            pp.lo[a], pp.hi[a] = -self.d[a] / 2, self.Lbox[a] - self.d[a] / 2
            pp.alo[a], pp.ahi[a] = self.alo[a], self.ahi[a]
        return pp

    def _species_entries(self, dt, with_mig=False, pushed=False):
        for sp in self.species:
            def after(sp=sp):
                sp["since"] += 1

            pp = self._push_params(sp, dt)
            ent = {"pc": sp["c"], "tiling": None, "n_sorted": 0, "pp": pp, "overflow": None, "count": None, "after": after}
            if self.tiled and sp["tiling"] is not None:
                ws = sp["ws"]
                ent.update(tiling=sp["tiling"], n_sorted=sp["n_sorted"], overflow=ws["overflow"], count=ws["count"])
            if with_mig:
                ent["mig"] = mig = self._slab_species(sp, pushed)
                if "leavers" in mig and not pushed:
                    pp.leavers, pp.leaver_count = mig["leavers"].data_ptr(), mig["leaver_count"].data_ptr()
                    pp.leaver_capacity = mig["leavers"].numel()
                    pp.leave_lo, pp.leave_hi = self._owner_bounds_x()
            yield ent

    def step(self, dt, laser=None, defer_e2=False):
        """``laser``: optional callable ``laser(engine, dt)`` run at the reference's '_laser' stage
        (between the second B half step and its guard sync, simulation.py:1098-1112); ``defer_e2``: see
        PicEngine2D.step"""
        self._dt_hint = dt
        if self.one_call_step():
            self.step_fused(dt, laser, defer_e2)
            return
        if self.can_fuse() and self.tiled and not self.overlap:
            self._step_segments(dt, laser, defer_e2)
            return
        L, st, g = self.L, self.stream, self._g()
        self._flush_e2()            # (the per-stage path neither defers nor doubles: complete what a fused step left)
        defer_e2 = False
        self.update_efield(0.5 * dt)
        self.sync_guard_fields(1)
        self.update_bfield(0.5 * dt)
        self.sync_guard_fields(2)
        self.reset_current()
        if not (self.overlap and self.push_deposit_overlapped(dt)):
            for i in range(len(self.species)):
                self.push_deposit(i, dt)
            self.sync_currents()
        for i in range(len(self.species)):
            self.sync_particles(i)
        self.update_bfield(0.5 * dt)
        if laser is not None:
            laser(self, dt)
        self.sync_guard_fields(2)
        self.update_efield(0.5 * dt)
        if not defer_e2:
            self.sync_guard_fields(1)

    def _step_segments(self, dt, laser=None, defer_e2=False):
        """see PicEngine2D._step_segments: sub-ranges of lpa_step between the exchanges torch.distributed carries"""
        S = _lib
        local_b = self.local_b()
        self.step_stages(dt, S.LPA_STAGE_E1, S.LPA_STAGE_E1)
        self._exchange_guards(1)
        if local_b:
            self.step_stages(dt, S.LPA_STAGE_B1, S.LPA_STAGE_PUSH)
        else:
            self.step_stages(dt, S.LPA_STAGE_B1, S.LPA_STAGE_B1)
            self._exchange_guards(2)
            self.step_stages(dt, S.LPA_STAGE_RESET, S.LPA_STAGE_PUSH)
        self.defer_rho = not local_b
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
            self._exchange_guards(2)
            self.step_stages(dt, S.LPA_STAGE_E2, S.LPA_STAGE_E2, defer_e2)  # (deferred: nothing is launched, the next E1 doubles)
        if not defer_e2:
            self._exchange_guards(1)

    _surplus_message = PicEngine2D._surplus_message

    def check_migration(self):
        """see PicEngine2D.check_migration"""
        for sp in self.species:
            if sp["ws"] is not None and self.comm.size > 1:
                surplus = int(sp["ws"]["counters"][3].item())
                if surplus > 0 and not self._mig_surplus(surplus):
                    raise _lib.LpaError(self._surplus_message(surplus))

    def diagnostics(self, reduce=False):
        """this rank's share; ``reduce=True`` sums over the ranks (one all-reduce)"""
        self._flush_e2()
        self.check_migration()
        self._diag.zero_()
        check(self.L.lpa_diag_fields(self._g(), self.eps0, self.mu0, self._diag.data_ptr(), self.stream), "diag")
        f = self._diag.cpu().numpy().copy()
        out = dict(field_energy=f[0] + f[1], charge=f[2], kinetic=[], nalive=[])
        for sp in self.species:
            d = torch.zeros(2, dtype=torch.float64, device=self.device)
            check(self.L.lpa_diag_particles(C.byref(sp["c"]), sp["m"], d.data_ptr(), self.stream), "diag p")
            d = d.cpu().numpy()
            out["kinetic"].append(float(d[0]))
            out["nalive"].append(int(round(d[1])))
        return self.comm.reduce_diagnostics(out) if reduce else out
