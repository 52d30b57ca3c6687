"""HDF5 writers fed from device state -- the file-producing callbacks of `callback/hdf5.py` in the reference:

``SaveFieldsToHDF5``          `:282-399`  ``<prefix>/<itime:06d>.h5``, one dataset per field component
``SaveSpeciesDensityToHDF5``  `:402-614`  ``<prefix>/<species>_<itime:06d>.h5``, dataset ``density``
``SaveParticlesToHDF5``       `:616-700`  ``<prefix>/<species>_particles_<itime:06d>.h5``, one dataset per attribute + ``id``
``LoadParticles``             `callback/utils.py:1051-1178`  the reverse: a particle file into the resident stores

Same constructor arguments, file names, dataset names / shapes / types, chunking (one patch) and attributes.  What is
different is where the data comes from: the reference walks its host patches; here every rank selects its share of the
(optionally sliced) box as a strided view of the RESIDENT arrays, copies only that to the host and writes it into the
file as one block.  The callbacks are ``device_native``: they never trigger the host-mirror refresh of ``Simulation``.

Several ranks: rank 0 creates the file, then the ranks add their blocks ONE AFTER THE OTHER (a barrier between two
writers).  The reference's non-MPI mode lets all ranks append to the same file at once without file locking
(`:253-273`); serial HDF5 does not promise that works, and a slab per rank is at most eight turns.  The ``mpi`` argument
is accepted for compatibility; a parallel (MPI-IO) HDF5 build is never assumed.

File access goes through ``h5lite`` (h5py, or libhdf5 through ctypes when h5py is not installed).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from . import h5lite
from .slices import normalize_slice, slab_selection, slice_text

FIELD_COMPONENTS = ("ex", "ey", "ez", "bx", "by", "bz", "jx", "jy", "jz", "rho")


def _dims(sim):
    return (sim.nx, sim.ny) + ((sim.nz,) if sim.dimension == 3 else ())


def _per_patch(sim):
    return (sim.nx_per_patch, sim.ny_per_patch) + ((sim.nz_per_patch,) if sim.dimension == 3 else ())


def _interior(sim, name):
    """device view of the interior cells of one field array of this rank's slab"""
    eng, g = sim.engine, sim.engine.ng
    if sim.dimension == 3:
        return eng.view(name)[g:-g, g:-g, g:-g]
    return eng.grid.view(name)[g:-g, g:-g]


def _box_attrs(sim, norm):
    at = {}
    for k, ax in enumerate("xyz"[: sim.dimension]):
        at["n" + ax], at["d" + ax], at["L" + ax] = _dims(sim)[k], getattr(sim, "d" + ax), getattr(sim, "L" + ax)
    at["time"], at["itime"] = sim.time, sim.itime
    if norm is not None:
        at["slice"] = slice_text(norm, _dims(sim))
    return at


def _in_turn(sim, filename, create, add):
    """rank 0: ``create(f)`` on a new file; then every rank in rank order: ``add(f)`` on the open file"""
    comm = sim.mpi.comm
    for r in range(comm.Get_size()):
        if r == comm.Get_rank():
            with h5lite.File(filename, "w" if r == 0 else "a") as f:
                if r == 0:
                    create(f)
                add(f)
        if comm.Get_size() > 1:
            comm.Barrier()


def write_box(sim, filename, parts, out_idx, shape, norm, extra_attrs=None):
    """``parts``: {dataset name: device tensor = this rank's block, or None}; ``out_idx``: where the block goes"""
    chunks = tuple(min(c, s) for c, s in zip(_per_patch(sim), shape))
    attrs = dict(_box_attrs(sim, norm), **(extra_attrs or {}))

    def create(f):
        for name in parts:
            f.create_dataset(name, shape=shape, dtype="f8", chunks=chunks)      # (fill value 0, like np.zeros)
        for k, v in attrs.items():
            f.attrs[k] = v

    def add(f):
        if out_idx is None:
            return
        for name, t in parts.items():
            f[name][out_idx] = t.to(torch.float64).cpu().numpy()

    _in_turn(sim, filename, create, add)


class _Writer:
    device_native = True

    def __call__(self, sim):
        return self._call(sim)

    def _normalize(self, sim):
        self._normalized_slice = normalize_slice(sim.dimension, self.slice, _dims(sim))
        return self._normalized_slice


class SaveFieldsToHDF5(_Writer):
    DEFAULT_STAGE = "end"

    def __init__(self, prefix="", interval=100, components=None, mpi=False, slice=None):
        h5lite.require()
        self.stage, self.interval, self.mpi = self.DEFAULT_STAGE, interval, mpi
        self.prefix = Path(prefix)
        self.prefix.mkdir(parents=True, exist_ok=True)
        self.all_components = set(FIELD_COMPONENTS)
        if components is None:
            self.components = list(FIELD_COMPONENTS)
        else:
            bad = set(components) - self.all_components
            if bad:
                raise ValueError(f"Invalid field components: {bad}")
            self.components = list(components)
        self.slice, self._normalized_slice = slice, None

    def _call(self, sim):
        norm = self._normalize(sim)
        local, out_idx, shape = slab_selection(sim, norm)
        parts = {c: (_interior(sim, c)[local] if local is not None else None) for c in self.components}
        write_box(sim, self.prefix / f"{sim.itime:06d}.h5", parts, out_idx, shape, norm)


class SaveSpeciesDensityToHDF5(_Writer):
    """stage ``current_deposition`` runs once per species, right after that species' deposit: the species' number
    density is the rho it just added, over its charge (`callback/hdf5.py:451-481,517-561`).  The snapshot of rho taken
    before the species' deposit and the difference stay on the device, already reduced to the selection."""
    DEFAULT_STAGE = "current_deposition"

    def __init__(self, species, prefix="", interval=100, mpi=False, slice=None):
        h5lite.require()
        self.stage, self.interval, self.mpi = self.DEFAULT_STAGE, interval, mpi
        self.species = species
        self.prefix = Path(prefix)
        self.prefix.mkdir(parents=True, exist_ok=True)
        self.prev_rho = None
        self.slice, self._normalized_slice = slice, None

    @property
    def ispec_target(self):
        return self.species.ispec

    def _selected_rho(self, sim, local):
        return _interior(sim, "rho")[local] if local is not None else None

    def _call(self, sim):
        t = self.ispec_target
        if sim.ispec is None or sim.ispec not in (t - 1, t):
            return
        norm = self._normalize(sim)
        local, out_idx, shape = slab_selection(sim, norm)
        sim.sync_currents()
        rho = self._selected_rho(sim, local)
        if sim.ispec == t - 1:
            self.prev_rho = rho.clone() if rho is not None else None
            return
        density = None
        if rho is not None:
            density = (rho if t == 0 else rho - self.prev_rho) / self.species.q
        self.prev_rho = None
        self._deliver(sim, density, out_idx, shape, norm)

    def _deliver(self, sim, density, out_idx, shape, norm):
        write_box(sim, self.prefix / f"{self.species.name}_{sim.itime:06d}.h5", {"density": density}, out_idx, shape,
                  norm, {"species": self.species.name})


def live_attributes(sim, species, names):
    """{name: host array} of the LIVE particles of this rank (device order) for the asked attributes, + ``id`` (uint64).
    ``_id`` is the id's bit pattern as float64, as the reference's attribute list carries it (`core/particles.py:63-67`)."""
    eng = sim.engine
    if sim.dimension == 3:
        from .engine3d import ATTRS3, ID_ROW
        sp = eng.species[species.ispec]
        d = sp["data"][:, : sp["n"]]
        live = ~torch.isnan(d[0])
        rows = {a: d[k] for k, a in enumerate(ATTRS3)}
        rows["_id"] = d[ID_ROW]
        ids = d[ID_ROW].view(torch.int64)
    else:
        sp = eng.species[species.ispec]
        if "inv_gamma" in names:
            sp.refresh_inv_gamma()
        s, n = sp.cset, sp.n
        live = ~torch.isnan(s.arr("x")[:n])
        ids = s.id[:n]
        have = list(s.names) + ["_id"]
        rows = {a: (ids.view(torch.float64) if a == "_id" else s.arr(a)[:n]) for a in names if a in have}
    missing = [a for a in names if a not in rows]
    if missing:
        have = sorted(rows) if sim.dimension == 3 else sorted(have)
        raise ValueError(f"the resident store of {species.name!r} has no attribute {missing}; it holds {have}")
    out = {a: rows[a][live].cpu().numpy() for a in names}
    out["id"] = ids[live].cpu().numpy().view(np.uint64)
    return out


def store_attributes(sim, species):
    if sim.dimension == 3:
        from .engine3d import ATTRS3
        return list(ATTRS3) + ["_id"]
    return list(sim.engine.species[species.ispec].cset.names) + ["_id"]


class SaveParticlesToHDF5(_Writer):
    DEFAULT_STAGE = "end"

    def __init__(self, species, prefix="", interval=100, attrs=None):
        h5lite.require()
        self.stage, self.interval = self.DEFAULT_STAGE, interval
        self.species = species
        self.prefix = Path(prefix)
        self.prefix.mkdir(parents=True, exist_ok=True)
        self.attrs = None if attrs is None else [a for a in attrs if a != "id"]      # ('id' is always written)

    @property
    def reads_part_eb(self):
        """does this writer store ex_part ... bz_part (every attribute, or one of them by name)?  The pushes that precede
        it must then write the gathered fields back (Simulation.run: ``engine.write_part_eb``), as the reference's do"""
        return self.attrs is None or any(a.endswith("_part") for a in self.attrs)

    def _call(self, sim):
        if self.attrs is None:
            self.attrs = store_attributes(sim, self.species)
        comm = sim.mpi.comm
        data = live_attributes(sim, self.species, self.attrs)
        counts = comm.allgather(len(data["id"]))
        start = sum(counts[: comm.Get_rank()])

        def create(f):
            for a in self.attrs:
                f.create_dataset(a, shape=(sum(counts),), dtype="f8")
            f.create_dataset("id", shape=(sum(counts),), dtype="u8")
            f.attrs["time"], f.attrs["itime"] = sim.time, sim.itime

        def add(f):
            if len(data["id"]):
                for a, v in data.items():
                    f[a][start: start + len(v)] = v

        _in_turn(sim, self.prefix / f"{self.species.name}_particles_{sim.itime:06d}.h5", create, add)


class LoadParticles(_Writer):
    """stage ``init``, by default at the first step only: read ``/x /y (/z) /w`` and whatever else of the store's
    attributes the file holds (``_id`` and ``inv_gamma`` are never taken from a file; other datasets are ignored) in
    batches, keep what lies in this rank's slab -- [first node - d/2, last node + d/2) per axis, the patches' own
    bounds in the reference (`callback/utils.py:1122-1129`) -- and append it to the resident store as loose particles
    with fresh ids; the next step's sort files them.  ``inv_gamma`` is set to 1 / sqrt(1 + u^2) (the reference line
    `:1146` stores sqrt(1 + u^2) there, which its first half position push would take for 1 / gamma; the resident 2-D
    engine recomputes it from the momenta anyway)."""
    DEFAULT_STAGE = "init"
    REQUIRED = {2: ("x", "y", "w"), 3: ("x", "y", "z", "w")}

    def __init__(self, species, file, interval=None):
        h5lite.require()
        self.stage, self.species, self.file = self.DEFAULT_STAGE, species, file
        self._batch_size = 1 << 20
        self.interval = (lambda sim: sim.itime == 0) if interval is None else interval

    def _filter_attributes(self, sim):
        with h5lite.File(self.file, "r") as f:
            in_file = set(f.keys())
        for a in self.REQUIRED[sim.dimension]:
            if a not in in_file:
                raise ValueError(f"Attribute '{a}' not found in {self.file}")
        return (in_file & set(store_attributes(sim, self.species))) - {"_id", "inv_gamma"}

    def _call(self, sim):
        names = sorted(self._filter_attributes(sim))
        eng, dim = sim.engine, sim.dimension
        d = (sim.dx, sim.dy) + ((sim.dz,) if dim == 3 else ())
        n_loc = (sim.nx // sim.comm.size, sim.ny) + ((sim.nz,) if dim == 3 else ())
        origin = (eng.x0, getattr(eng, "y0", 0.0), getattr(eng, "z0", 0.0))[:dim]
        with h5lite.File(self.file, "r") as f:
            total = f["x"].shape[0]
            for lo in range(0, total, self._batch_size):
                hi = min(lo + self._batch_size, total)
                data = {a: np.asarray(f[a][lo:hi], dtype=np.float64) for a in names}
                keep = np.ones(hi - lo, dtype=bool)
                for a, o, n, dd in zip("xyz", origin, n_loc, d):
                    keep &= (data[a] >= o - dd / 2) & (data[a] < o + (n - 0.5) * dd)
                k = int(keep.sum())
                if k == 0:
                    continue
                dev = {a: torch.from_numpy(np.ascontiguousarray(v[keep])).to(eng.device) for a, v in data.items()}
                for a in ("ux", "uy", "uz"):
                    dev.setdefault(a, torch.zeros(k, dtype=torch.float64, device=eng.device))
                dev["inv_gamma"] = torch.rsqrt(1.0 + dev["ux"] ** 2 + dev["uy"] ** 2 + dev["uz"] ** 2)
                if dim == 3:
                    from .engine3d import ATTRS3
                    eng.append_device(self.species.ispec, torch.stack([dev[a] for a in ATTRS3]))
                else:
                    i = self.species.ispec
                    dev["id"] = torch.arange(k, dtype=torch.int64, device=eng.device) + sim._next_ids(i)
                    sim._id_next[i] += k
                    eng.append_particles_device(i, dev)
        if sim.comm.size > 1 and total:     # appends are collective on a chain (engine.append_particles_device)
            if dim == 3:
                eng.append_device(self.species.ispec, torch.empty((9, 0), dtype=torch.float64, device=eng.device))
            else:
                eng.append_particles_device(self.species.ispec, {"x": torch.empty(0, device=eng.device)})
