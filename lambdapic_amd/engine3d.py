"""PicEngine3D -- device-resident 3-D step on one GPU (periodic box), same stage order as
``PicEngine2D``.  Particles are binned into 4 x 4 x 16-cell tiles (``lpa_sort_tiles_3d``) and pushed by the
LDS-tiled kernel (``lpa_push_deposit_tiled_3d``: J / rho of the tile accumulated in LDS, E / B
gathered from global memory) + the overflow list; grids whose extents are not multiples of the tile
use the global-memory form (``lpa_push_deposit_3d``).  The 3-D slab exchange is the next row
(DESIGN.md section 7).

Replaces, per step: ``update_efield/bfield_patches_3d`` (`core/maxwell/cpu.py:115-158`),
``sync_guard_fields_3d`` / ``sync_currents_3d`` with a self neighbour (`core/patch/sync_fields3d.c`),
``reset_current_cpu_3d`` (`core/current/cpu3d.c:185-240`), ``unified_boris_pusher_cpu_3d``
(`core/pusher/unified/unified_pusher_3d.c:219-436`) and the periodic part of
``sync_particles_3d`` (`core/patch/sync_particles_3d.c`).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, constants
from ._lib import check, lib
from .fields import FIELD_ATTRS, from_device_layout, to_device_layout

ATTRS3 = ("x", "y", "z", "ux", "uy", "uz", "inv_gamma", "w")


class PicEngine3D:
    def __init__(self, nx, ny, nz, dx, dy, dz, n_guard=3, device="cuda:0", tiled=None, sort_interval=10,
                 block_particles=4096):
        self.L = lib()
        fits = nx % _lib.LPA_TILE3_X == 0 and ny % _lib.LPA_TILE3_Y == 0 and nz % _lib.LPA_TILE3_Z == 0
        if tiled and not fits:
            raise ValueError("tiled 3-D path needs nx, ny multiples of 4 and nz a multiple of 16")
        self.tiled = fits if tiled is None else bool(tiled)
        self.sort_interval, self.block_particles = int(sort_interval), int(block_particles)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LpaError("PicEngine3D needs a GPU device; there is no CPU path")
        self.n = (int(nx), int(ny), int(nz))
        self.d = (float(dx), float(dy), float(dz))
        self.ng = int(n_guard)
        N = tuple(v + 2 * self.ng for v in self.n)
        self.buf = torch.zeros((10,) + N, dtype=torch.float64, device=self.device)
        g = _lib.lpa_grid()
        g.nx, g.ny, g.nz, g.ng = *self.n, self.ng
        g.dx, g.dy, g.dz = self.d
        g.x0 = g.y0 = g.z0 = 0.0
        for k, name in enumerate(FIELD_ATTRS):
            setattr(g, name, self.buf[k].data_ptr())
        self.c = g
        self.species = []
        self.eps0, self.mu0 = constants.EPSILON_0, constants.MU_0
        self._diag = torch.zeros(8, dtype=torch.float64, device=self.device)

    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def view(self, name):
        return self.buf[FIELD_ATTRS.index(name)]

    def upload_field(self, name, wrapped):
        self.view(name).copy_(torch.from_numpy(np.ascontiguousarray(to_device_layout(wrapped, self.ng))))

    def download_field(self, name):
        return np.ascontiguousarray(from_device_layout(self.view(name).cpu().numpy(), self.ng))

    def add_species(self, q, m, host_particles):
        """upload the live particles of one host bag (ParticlesBase-like with z)"""
        live = ~host_particles.is_dead
        n = int(live.sum())
        data = torch.from_numpy(np.stack([getattr(host_particles, a)[live] for a in ATTRS3])).to(self.device)
        p = _lib.lpa_particles()
        p.n = n
        for k, a in enumerate(ATTRS3):
            setattr(p, a, data[k].data_ptr())
        for k in range(6):
            p.part_eb[k] = None
        p.id, p.is_dead = None, None
        self.species.append({"q": float(q), "m": float(m), "data": data, "c": p, "n": n, "alt": None,
                             "tiling": None, "since": 0, "ws": None})
        return len(self.species) - 1

    @staticmethod
    def _cstruct(data, n):
        p = _lib.lpa_particles()
        p.n = int(n)
        for k, a in enumerate(ATTRS3):
            setattr(p, a, data[k].data_ptr())
        for k in range(6):
            p.part_eb[k] = None
        p.id, p.is_dead = None, None
        return p

    def sort(self, i):
        """tile-bin species ``i`` (replaces sort_particles_patches_3d, core/sort/cpu3d.c); one host
        sync for the live count"""
        sp = self.species[i]
        cap = sp["data"].shape[1]
        if sp["alt"] is None:
            sp["alt"] = torch.empty_like(sp["data"])
            nbytes = self.L.lpa_sort_workspace_bytes(self._g(), cap)
            sp["ws"] = {"sort": torch.zeros(nbytes, dtype=torch.uint8, device=self.device),
                        "overflow": torch.empty(max(cap, 1), dtype=torch.int32, device=self.device),
                        "count": torch.zeros(1, dtype=torch.int32, device=self.device),
                        "tiling": _lib.lpa_tiling()}
        ws = sp["ws"]
        src, dst = self._cstruct(sp["data"], sp["n"]), self._cstruct(sp["alt"], cap)
        check(self.L.lpa_sort_tiles_3d(self._g(), C.byref(src), C.byref(dst), ws["sort"].data_ptr(),
                                       ws["sort"].numel(), self.block_particles, _lib.LPA_ORDER_STRIPED,
                                       C.byref(ws["tiling"]), self.stream), "lpa_sort_tiles_3d")
        n_live = int(ws["sort"][:4].view(torch.int32)[0].item())
        sp["data"], sp["alt"] = sp["alt"], sp["data"]
        sp["n"] = n_live
        sp["c"] = self._cstruct(sp["data"], n_live)
        ws["tiling"].n_sorted = n_live
        sp["tiling"] = ws["tiling"]
        sp["since"] = 0

    def download_species(self, i):
        d = self.species[i]["data"][:, : self.species[i]["n"]].cpu().numpy()
        return {a: d[k] for k, a in enumerate(ATTRS3)}

    def _g(self):
        return C.byref(self.c)

    def step(self, dt):
        L, st, g = self.L, self.stream, self._g()
        check(L.lpa_fdtd_e_3d(g, 0.5 * dt, self.eps0, st), "lpa_fdtd_e_3d")
        check(L.lpa_guard_wrap(g, 1, 7, st), "lpa_guard_wrap")
        check(L.lpa_fdtd_b_3d(g, 0.5 * dt, st), "lpa_fdtd_b_3d")
        check(L.lpa_guard_wrap(g, 2, 7, st), "lpa_guard_wrap")
        check(L.lpa_reset_current(g, st), "lpa_reset_current")
        for sp in self.species:
            pp = _lib.lpa_push_params()
            pp.dt, pp.q, pp.m, pp.wrap = dt, sp["q"], sp["m"], 7
            for a in range(3):
                pp.lo[a], pp.hi[a] = -self.d[a] / 2, self.n[a] * self.d[a] - self.d[a] / 2
            if not self.tiled:
                check(L.lpa_push_deposit_3d(g, C.byref(sp["c"]), C.byref(pp), 0, sp["n"], st), "lpa_push_deposit_3d")
                continue
            if sp["tiling"] is None or sp["since"] >= self.sort_interval:
                self.sort(self.species.index(sp))
            ws = sp["ws"]
            ws["count"].zero_()
            check(L.lpa_push_deposit_tiled_3d(g, C.byref(sp["c"]), C.byref(pp), C.byref(sp["tiling"]),
                                              ws["overflow"].data_ptr(), ws["count"].data_ptr(), st),
                  "lpa_push_deposit_tiled_3d")
            check(L.lpa_push_deposit_list_3d(g, C.byref(sp["c"]), C.byref(pp), ws["overflow"].data_ptr(),
                                             ws["count"].data_ptr(), sp["n"], st), "lpa_push_deposit_list_3d")
            sp["since"] += 1
        check(L.lpa_current_fold(g, 7, st), "lpa_current_fold")
        check(L.lpa_fdtd_b_3d(g, 0.5 * dt, st), "lpa_fdtd_b_3d")
        check(L.lpa_guard_wrap(g, 2, 7, st), "lpa_guard_wrap")
        check(L.lpa_fdtd_e_3d(g, 0.5 * dt, self.eps0, st), "lpa_fdtd_e_3d")
        check(L.lpa_guard_wrap(g, 1, 7, st), "lpa_guard_wrap")

    def diagnostics(self):
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
        return out
