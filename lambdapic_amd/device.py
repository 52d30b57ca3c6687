"""Device-resident state: one field slab and per-species SoA particle stores.

PyTorch is plumbing here (HBM allocation, streams, host<->device copies); every computation
goes through the C ABI of ``liblambdapic_amd.so``.

HBM layout
  * fields: ONE allocation ``float64[10][NX][NY]`` (order ex ey ez bx by bz jx jy jz rho,
    conventional guard layout ``[ng | interior | ng]``, y fastest).  x is the slowest axis, so an
    x-face halo of a component is ``ng*NY`` contiguous doubles, and jx jy jz rho are contiguous
    (one memset zeroes all four).
  * particles: per species TWO sets (ping/pong for the out-of-place tile sort) of SoA arrays
    ``x y ux uy uz inv_gamma w id`` (+ the six ``*_part`` arrays only when a callback needs them),
    each ``float64[capacity]``.  Dead slots are marked by ``x = NaN``.
"""
from __future__ import annotations


import ctypes as C

import numpy as np
import torch

from . import _lib
from .fields import FIELD_ATTRS, from_device_layout, to_device_layout

PART_CORE = ("x", "y", "ux", "uy", "uz", "inv_gamma", "w")
PART_EB = ("ex_part", "ey_part", "ez_part", "bx_part", "by_part", "bz_part")


def current_stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# ---- restart support (RestartDump, `callback/restart.py:13-160`) ---------------------------------------------
# Device state pickles as HOST arrays of exactly the slots in use (never as torch views: a view would drag its whole
# storage along), ctypes structs / library handles / streams / process groups are dropped and rebuilt on load.
# ``RESTORE_DEVICE`` (set by ``RestartDump.load(..., device=...)``) overrides the device a checkpoint was written
# from -- a restarted rank need not sit on the same GPU index.
RESTORE_DEVICE = None


def restore_device(saved) -> torch.device:
    return torch.device(RESTORE_DEVICE if RESTORE_DEVICE is not None else saved)


def to_host(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy().copy()


class DeviceGrid2D:
    """One rank's slab of the Yee grid in HBM."""

    def __init__(self, nx, ny, dx, dy, x0, y0, n_guard, device):
        self.nx, self.ny, self.ng = int(nx), int(ny), int(n_guard)
        self.dx, self.dy, self.x0, self.y0 = float(dx), float(dy), float(x0), float(y0)
        self.device = torch.device(device)
        self.NX, self.NY = self.nx + 2 * self.ng, self.ny + 2 * self.ng
        self.buf = torch.zeros((10, self.NX, self.NY), dtype=torch.float64, device=self.device)
        g = _lib.lpa_grid()
        g.nx, g.ny, g.nz, g.ng = self.nx, self.ny, 1, self.ng
        g.dx, g.dy, g.dz = self.dx, self.dy, 0.0
        g.x0, g.y0, g.z0 = self.x0, self.y0, 0.0
        for k, name in enumerate(FIELD_ATTRS):
            setattr(g, name, self.buf[k].data_ptr())
        self.c = g

    def view(self, name) -> torch.Tensor:
        return self.buf[FIELD_ATTRS.index(name)]

    def __getstate__(self):
        st = {k: v for k, v in self.__dict__.items() if k not in ("buf", "c")}
        st["device"], st["buf_host"] = str(self.device), to_host(self.buf)
        return st

    def __setstate__(self, st):
        buf = st.pop("buf_host")
        self.__dict__.update(st)
        self.device = restore_device(st["device"])
        self.buf = torch.from_numpy(buf).to(self.device)
        g = _lib.lpa_grid()
        g.nx, g.ny, g.nz, g.ng = self.nx, self.ny, 1, self.ng
        g.dx, g.dy, g.dz = self.dx, self.dy, 0.0
        g.x0, g.y0, g.z0 = self.x0, self.y0, 0.0
        for k, name in enumerate(FIELD_ATTRS):
            setattr(g, name, self.buf[k].data_ptr())
        self.c = g

    # ---- host mirrors ---------------------------------------------------------------------------
    def upload(self, name, wrapped: np.ndarray):
        """host array in λPIC's wrapped guard layout -> device (conventional layout)"""
        a = np.ascontiguousarray(to_device_layout(wrapped, self.ng))
        self.view(name).copy_(torch.from_numpy(a))

    def download(self, name) -> np.ndarray:
        """device -> host array in λPIC's wrapped guard layout"""
        return np.ascontiguousarray(from_device_layout(self.view(name).cpu().numpy(), self.ng))

    def upload_patches(self, patches, npatch_x, npatch_y, attrs=FIELD_ATTRS):
        """assemble the patch mirrors (interiors + outer guards) into the slab"""
        for name in attrs:
            slab = np.zeros((self.NX, self.NY))
            for p in patches:
                f = p.fields
                a = to_device_layout(getattr(f, name), f.n_guard)
                i0, j0 = p.ipatch_x * f.nx, p.ipatch_y * f.ny
                slab[i0:i0 + f.nx + 2 * f.n_guard, j0:j0 + f.ny + 2 * f.n_guard] = a
            # interiors win over neighbours' guards
            for p in patches:
                f = p.fields
                g = f.n_guard
                a = to_device_layout(getattr(f, name), g)
                i0, j0 = p.ipatch_x * f.nx + g, p.ipatch_y * f.ny + g
                slab[i0:i0 + f.nx, j0:j0 + f.ny] = a[g:g + f.nx, g:g + f.ny]
            self.view(name).copy_(torch.from_numpy(slab))

    def download_patches(self, patches, attrs=FIELD_ATTRS):
        """scatter the slab back into the patch mirrors (each patch gets its interior + guards)"""
        for name in attrs:
            slab = self.view(name).cpu().numpy()
            for p in patches:
                f = p.fields
                g = f.n_guard
                i0, j0 = p.ipatch_x * f.nx, p.ipatch_y * f.ny
                blk = slab[i0:i0 + f.nx + 2 * g, j0:j0 + f.ny + 2 * g]
                getattr(f, name)[...] = from_device_layout(blk, g)


class ParticleSet:
    """one SoA set; ``c`` is the lpa_particles view of the first ``n`` slots"""

    def __init__(self, capacity, device, with_eb=False):
        self.capacity = int(capacity)
        self.core = list(PART_CORE)
        self.names = self.core + (list(PART_EB) if with_eb else [])
        # ``data``: the core attributes.  ex_part ... bz_part (what the last push saw, for host callbacks) get their six
        # rows when somebody first asks for them -- a run without mirror-reading callbacks never does, and a set holds
        # 64 instead of 112 bytes per slot
        self.data = torch.empty((len(self.core), self.capacity), dtype=torch.float64, device=device)
        self.eb = None
        self.id = torch.zeros(self.capacity, dtype=torch.int64, device=device)
        self.with_eb = with_eb

    def arr(self, name) -> torch.Tensor:
        if name in PART_EB:
            if not self.with_eb:
                raise KeyError(name)
            if self.eb is None:
                self.eb = torch.zeros((len(PART_EB), self.capacity), dtype=torch.float64, device=self.data.device)
            return self.eb[PART_EB.index(name)]
        return self.data[self.core.index(name)]

    def cstruct(self, n, eb=True) -> _lib.lpa_particles:
        p = _lib.lpa_particles()
        p.n = int(n)
        for name in PART_CORE:
            setattr(p, name, self.arr(name).data_ptr())
        p.z = None
        for k, name in enumerate(PART_EB):
            p.part_eb[k] = self.arr(name).data_ptr() if (self.with_eb and eb) else None
        p.id = self.id.data_ptr()
        p.is_dead = None
        return p


class DeviceParticles:
    """One species on one rank: ping/pong SoA sets, tile binning state, arrival area."""

    def __init__(self, capacity, device, q, m, with_eb=False):
        self.device = torch.device(device)
        self.q, self.m = float(q), float(m)
        self.capacity = int(capacity)
        self.sets = [ParticleSet(capacity, self.device, with_eb), None]
        self.with_eb = with_eb
        self.cur = 0
        self.n = 0            # slots in use in the current set (live + dead + arrival area)
        self.n_sorted = 0     # [0, n_sorted) is tile ordered
        self.tiling = None
        self.steps_since_sort = 0
        # the fused kernels of the resident engine do not stream inv_gamma (LPA_PUSH_NO_IG): the array is rebuilt from
        # the momenta for whoever reads it (refresh_inv_gamma)
        self.ig_stale = False

    def refresh_inv_gamma(self):
        """inv_gamma[0:n) = 1 / sqrt(1 + u^2) if a fused push left it stale (same function, same bits as the kernel
        would have stored); call before reading ``cset.arr('inv_gamma')`` directly"""
        if getattr(self, "ig_stale", False) and self.n > 0:
            pc = self.cset.cstruct(self.n, eb=False)
            _lib.check(_lib.lib().lpa_refresh_inv_gamma(C.byref(pc), 0, self.n,
                                                      torch.cuda.current_stream(self.device).cuda_stream),
                       "lpa_refresh_inv_gamma")
        self.ig_stale = False

    @property
    def cset(self) -> ParticleSet:
        return self.sets[self.cur]

    def other(self) -> ParticleSet:
        if self.sets[1 - self.cur] is None:
            self.sets[1 - self.cur] = ParticleSet(self.capacity, self.device, self.with_eb)
        return self.sets[1 - self.cur]

    def reserve(self, capacity):
        """grow the store to at least ``capacity`` slots (the reference's ParticlesBase.extend,
        core/particles.py:141-168); existing slots keep their order"""
        if capacity <= self.capacity:
            return False
        new = ParticleSet(int(capacity), self.device, self.with_eb)
        old = self.cset
        new.data[:, : self.n].copy_(old.data[:, : self.n])
        if old.eb is not None:
            new.arr(PART_EB[0])
            new.eb[:, : self.n].copy_(old.eb[:, : self.n])
        new.id[: self.n].copy_(old.id[: self.n])
        self.sets = [new, None]
        self.cur = 0
        self.capacity = int(capacity)
        return True

    def upload(self, host_particles_list):
        """concatenate the live particles of the host mirrors into the device store"""
        names = self.cset.names           # core attributes (+ ex_part..bz_part when carried)
        cols = {a: [] for a in names}
        ids = []
        for hp in host_particles_list:
            live = ~hp.is_dead & ~np.isnan(hp.x) & ~np.isnan(hp.y)
            for a in names:
                cols[a].append(getattr(hp, a)[live])
            ids.append(hp._id.view(np.int64)[live])
        n = int(sum(c.size for c in ids))
        if n > self.capacity:
            raise _lib.LpaError(f"particle capacity {self.capacity} < {n}")
        s = self.cset
        for a in names:
            s.arr(a)[:n].copy_(torch.from_numpy(np.concatenate(cols[a])))
        s.id[:n].copy_(torch.from_numpy(np.concatenate(ids)))
        self.n, self.n_sorted, self.tiling = n, 0, None
        self.ig_stale = False

    def __getstate__(self):
        """slots [0, n) of the current set as host arrays (dead slots included: x = NaN marks them); the tile
        order is not kept -- the first push after a load re-sorts, like the first push of a run"""
        self.refresh_inv_gamma()
        st = {k: v for k, v in self.__dict__.items() if k not in ("sets", "tiling", "device")}
        s = self.cset
        kept = list(s.core) + (list(PART_EB) if s.eb is not None else [])
        rows = s.data[:, : self.n] if s.eb is None else torch.cat([s.data[:, : self.n], s.eb[:, : self.n]])
        st.update(device=str(self.device), names=kept, data_host=to_host(rows), id_host=to_host(s.id[: self.n]))
        return st

    def __setstate__(self, st):
        data, ids, names = st.pop("data_host"), st.pop("id_host"), st.pop("names")
        self.__dict__.update(st)
        self.device = restore_device(st["device"])
        s = ParticleSet(self.capacity, self.device, self.with_eb)
        assert names[: len(s.core)] == s.core and len(names) in (len(s.core), len(s.names))
        rows = torch.from_numpy(data)
        for k, a in enumerate(names):          # (the particle-field rows only when the dumped set carried them)
            s.arr(a)[: self.n].copy_(rows[k])
        s.id[: self.n].copy_(torch.from_numpy(ids))
        self.sets, self.cur = [s, None], 0
        self.n_sorted, self.tiling = 0, None
        self.steps_since_sort = 1 << 30          # forces the sort before the next tiled push

    def download(self):
        """dict of host arrays of the LIVE particles (order = device order)"""
        self.refresh_inv_gamma()
        s = self.cset
        x = s.arr("x")[: self.n]
        live = ~torch.isnan(x)
        out = {a: s.arr(a)[: self.n][live].cpu().numpy() for a in s.names}
        out["_id"] = s.id[: self.n][live].cpu().numpy().view(np.float64)
        return out
