"""Kernel-level drop-ins: the call signatures of the reference's compiled extension functions,
executed on the GPU.

Each function takes the same duck-typed host objects the reference's C extensions take (lists of
``ParticlesBase`` / ``Fields2D``-like bags with numpy arrays in λPIC's wrapped guard layout), moves
them to HBM, runs the HIP kernel through the C ABI and writes the results back into the host arrays
in place -- so a reference test that calls e.g. ``unified_boris_pusher_cpu_2d([p], [f], 1, dt, q, m)``
reads the same here.  These are host-buffer (PCIe-inclusive) entry points for parity tests and for
callbacks; the resident path is ``engine.PicEngine2D``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, constants
from ._lib import check, lib
from .device import PART_CORE, PART_EB
from .fields import from_device_layout, to_device_layout

FIELD_ORDER = ("ex", "ey", "ez", "bx", "by", "bz", "jx", "jy", "jz", "rho")


def _device():
    if not torch.cuda.is_available():
        raise _lib.LpaError("no GPU visible: lambdapic_amd has no CPU fallback")
    return torch.device("cuda:0")


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


class _GridOnDevice:
    """one patch's field arrays in HBM (conventional layout); dim 2 or 3"""

    def __init__(self, f, dev):
        self.f, self.dev = f, dev
        self.dim = 3 if hasattr(f, "nz") and getattr(f, "nz", 0) and f.ex.ndim == 3 else 2
        ng = f.n_guard
        host = np.stack([to_device_layout(getattr(f, a), ng) for a in FIELD_ORDER])
        self.buf = torch.from_numpy(np.ascontiguousarray(host)).to(dev)
        g = _lib.lpa_grid()
        g.nx, g.ny, g.nz, g.ng = f.nx, f.ny, (f.nz if self.dim == 3 else 1), ng
        g.dx, g.dy, g.dz = f.dx, f.dy, (f.dz if self.dim == 3 else 0.0)
        g.x0, g.y0, g.z0 = f.x0, f.y0, (f.z0 if self.dim == 3 else 0.0)
        for k, a in enumerate(FIELD_ORDER):
            setattr(g, a, self.buf[k].data_ptr())
        self.c = g

    def ref(self):
        return C.byref(self.c)

    def download(self, names):
        host = self.buf.cpu().numpy()
        for a in names:
            getattr(self.f, a)[...] = from_device_layout(host[FIELD_ORDER.index(a)], self.f.n_guard)


class _PartsOnDevice:
    """one particle bag in HBM; keeps the host order (is_dead passed through)"""

    def __init__(self, p, dev, dim=2, with_eb=True):
        self.p, self.dev, self.dim = p, dev, dim
        names = list(PART_CORE) + (["z"] if dim == 3 else []) + (list(PART_EB) if with_eb else [])
        self.names = names
        n = p.npart
        host = np.stack([np.asarray(getattr(p, a)[:n], dtype=np.float64) for a in names]) if n else \
            np.zeros((len(names), 0))
        self.data = torch.from_numpy(np.ascontiguousarray(host)).to(dev)
        self.dead = torch.from_numpy(np.ascontiguousarray(p.is_dead[:n].view(np.uint8))).to(dev)
        c = _lib.lpa_particles()
        c.n = n
        for a in PART_CORE:
            setattr(c, a, self.data[names.index(a)].data_ptr() if n else None)
        c.z = self.data[names.index("z")].data_ptr() if (dim == 3 and n) else None
        for k, a in enumerate(PART_EB):
            c.part_eb[k] = self.data[names.index(a)].data_ptr() if (with_eb and n) else None
        c.id = None
        c.is_dead = self.dead.data_ptr() if n else None
        self.c = c

    def ref(self):
        return C.byref(self.c)

    def download(self, names):
        host = self.data.cpu().numpy()
        n = self.p.npart
        for a in names:
            getattr(self.p, a)[:n] = host[self.names.index(a)]


def _push_params(dt, q, m):
    pp = _lib.lpa_push_params()
    pp.dt, pp.q, pp.m, pp.wrap = dt, q, m, 0
    return pp


def unified_boris_pusher_cpu_2d(particles_list, fields_list, npatches, dt, q, m, tiled=False,
                                order=_lib.LPA_ORDER_STRIPED):
    """GPU drop-in for `core/pusher/unified/unified_pusher_2d.c:157-365`
    (``unified_boris_pusher_cpu_2d(particles_list, fields_list, npatches, dt, q, m) -> None``).
    ``tiled=True`` runs the LDS-tiled kernel (cell sort into ``order`` + tiled + overflow list)
    instead of the global-atomics kernel; results agree to summation order."""
    L, dev = lib(), _device()
    if npatches <= 0:
        return None
    for p, f in zip(particles_list[:npatches], fields_list[:npatches]):
        g = _GridOnDevice(f, dev)
        pp = _push_params(dt, q, m)
        if not tiled:
            d = _PartsOnDevice(p, dev)
            check(L.lpa_push_deposit_2d(g.ref(), d.ref(), C.byref(pp), 0, p.npart, _stream(dev)),
                  "lpa_push_deposit_2d")
            d.download(list(PART_CORE[:6]) + list(PART_EB))
        else:
            _unified_tiled_patch(L, dev, g, p, pp, order=order)
        g.download(["rho", "jx", "jy", "jz"])
    torch.cuda.synchronize(dev)
    return None


def _unified_tiled_patch(L, dev, g, p, pp, block_particles=1024, order=_lib.LPA_ORDER_STRIPED):
    """tile-bin the live particles of one host bag, run the tiled kernel + overflow list, and
    scatter the results back to the host slots (the original slot travels in the id field)."""
    n = p.npart
    live = np.nonzero(~p.is_dead[:n] & ~np.isnan(p.x[:n]) & ~np.isnan(p.y[:n]))[0]
    nl = live.size
    if nl == 0:
        return
    names = list(PART_CORE) + list(PART_EB)
    host = np.stack([np.asarray(getattr(p, a)[:n], dtype=np.float64)[live] for a in names])
    src = torch.from_numpy(np.ascontiguousarray(host)).to(dev)
    dst = torch.empty_like(src)
    sid = torch.from_numpy(live.astype(np.int64)).to(dev)
    did = torch.empty_like(sid)

    def cs(t, tid):
        c = _lib.lpa_particles()
        c.n = nl
        for a in PART_CORE:
            setattr(c, a, t[names.index(a)].data_ptr())
        c.z = None
        for k, a in enumerate(PART_EB):
            c.part_eb[k] = t[names.index(a)].data_ptr()
        c.id, c.is_dead = tid.data_ptr(), None
        return c

    ps, pd = cs(src, sid), cs(dst, did)
    nbytes = L.lpa_sort_workspace_bytes(g.ref(), nl)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    tiling = _lib.lpa_tiling()
    st = _stream(dev)
    check(L.lpa_sort_tiles_2d(g.ref(), C.byref(ps), C.byref(pd), ws.data_ptr(), nbytes, block_particles,
                              order, C.byref(tiling), st), "lpa_sort_tiles_2d")
    n_live = int(ws[:4].view(torch.int32)[0].item())
    assert n_live == nl
    tiling.n_sorted = nl
    overflow = torch.empty(nl, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    check(L.lpa_push_deposit_tiled_2d(g.ref(), C.byref(pd), C.byref(pp), C.byref(tiling),
                                      overflow.data_ptr(), cnt.data_ptr(), st), "lpa_push_deposit_tiled_2d")
    check(L.lpa_push_deposit_list_2d(g.ref(), C.byref(pd), C.byref(pp), overflow.data_ptr(), cnt.data_ptr(),
                                     nl, st), "lpa_push_deposit_list_2d")
    out = dst.cpu().numpy()
    slot = did.cpu().numpy()
    for a in list(PART_CORE[:6]) + list(PART_EB):
        getattr(p, a)[slot] = out[names.index(a)]
    return int(cnt.item())


def unified_boris_pusher_cpu_3d(particles_list, fields_list, npatches, dt, q, m):
    """GPU drop-in for `core/pusher/unified/unified_pusher_3d.c:219-436`"""
    L, dev = lib(), _device()
    for p, f in zip(particles_list[:npatches], fields_list[:npatches]):
        g = _GridOnDevice(f, dev)
        d = _PartsOnDevice(p, dev, dim=3)
        pp = _push_params(dt, q, m)
        check(L.lpa_push_deposit_3d(g.ref(), d.ref(), C.byref(pp), 0, p.npart, _stream(dev)),
              "lpa_push_deposit_3d")
        d.download(["x", "y", "z", "ux", "uy", "uz", "inv_gamma"] + list(PART_EB))
        g.download(["rho", "jx", "jy", "jz"])
    return None


def interpolation_patches_2d(particles_list, fields_list, npatches):
    """GPU drop-in for `core/interpolation/cpu2d.c:71-136`"""
    L, dev = lib(), _device()
    for p, f in zip(particles_list[:npatches], fields_list[:npatches]):
        g, d = _GridOnDevice(f, dev), _PartsOnDevice(p, dev)
        check(L.lpa_interpolate_2d(g.ref(), d.ref(), _stream(dev)), "lpa_interpolate_2d")
        d.download(list(PART_EB))


def current_deposition_cpu_2d(fields_list, particles_list, npatches, dt, q):
    """GPU drop-in for `core/current/cpu2d.c:74-184` (standalone deposit, accumulates into J/rho)"""
    L, dev = lib(), _device()
    for p, f in zip(particles_list[:npatches], fields_list[:npatches]):
        g, d = _GridOnDevice(f, dev), _PartsOnDevice(p, dev, with_eb=False)
        check(L.lpa_deposit_2d(g.ref(), d.ref(), dt, q, _stream(dev)), "lpa_deposit_2d")
        g.download(["rho", "jx", "jy", "jz"])


def interpolation_patches_3d(particles_list, fields_list, npatches):
    """GPU drop-in for `core/interpolation/cpu3d.c:99-169`"""
    L, dev = lib(), _device()
    for p, f in zip(particles_list[:npatches], fields_list[:npatches]):
        g, d = _GridOnDevice(f, dev), _PartsOnDevice(p, dev, dim=3)
        check(L.lpa_interpolate_3d(g.ref(), d.ref(), _stream(dev)), "lpa_interpolate_3d")
        d.download(list(PART_EB))


def current_deposition_cpu_3d(fields_list, particles_list, npatches, dt, q):
    """GPU drop-in for `core/current/cpu3d.c:118-183` (standalone deposit, accumulates into J/rho)"""
    L, dev = lib(), _device()
    for p, f in zip(particles_list[:npatches], fields_list[:npatches]):
        g, d = _GridOnDevice(f, dev), _PartsOnDevice(p, dev, dim=3, with_eb=False)
        check(L.lpa_deposit_3d(g.ref(), d.ref(), dt, q, _stream(dev)), "lpa_deposit_3d")
        g.download(["rho", "jx", "jy", "jz"])


def reset_current_cpu_2d(fields_list, npatches):
    """GPU drop-in for `core/current/cpu2d.c:19-72`"""
    L, dev = lib(), _device()
    for f in fields_list[:npatches]:
        g = _GridOnDevice(f, dev)
        check(L.lpa_reset_current(g.ref(), _stream(dev)), "lpa_reset_current")
        g.download(["rho", "jx", "jy", "jz"])


def boris_push(p, q, m, dt):
    """GPU drop-in for one bag of `boris_push_patches` (`core/pusher/cpu.py:11-35`)"""
    L, dev = lib(), _device()
    d = _PartsOnDevice(p, dev)
    check(L.lpa_boris(d.ref(), dt, q, m, _stream(dev)), "lpa_boris")
    d.download(["ux", "uy", "uz", "inv_gamma"])


def push_position_2d(p, dt):
    """GPU drop-in for one bag of `push_position_patches_2d` (`core/pusher/cpu.py:73-91`)"""
    L, dev = lib(), _device()
    d = _PartsOnDevice(p, dev, with_eb=False)
    check(L.lpa_push_position_2d(d.ref(), dt, _stream(dev)), "lpa_push_position_2d")
    d.download(["x", "y"])


def update_efield_2d(f, dt):
    """GPU drop-in for `update_efield_2d` on one field bag (`core/maxwell/cpu.py:9-22`)"""
    L, dev = lib(), _device()
    g = _GridOnDevice(f, dev)
    check(L.lpa_fdtd_e_2d(g.ref(), dt, constants.EPSILON_0, _stream(dev)), "lpa_fdtd_e_2d")
    g.download(["ex", "ey", "ez"])


def update_bfield_2d(f, dt):
    """GPU drop-in for `update_bfield_2d` (`core/maxwell/cpu.py:25-35`)"""
    L, dev = lib(), _device()
    g = _GridOnDevice(f, dev)
    check(L.lpa_fdtd_b_2d(g.ref(), dt, _stream(dev)), "lpa_fdtd_b_2d")
    g.download(["bx", "by", "bz"])


def update_efield_3d(f, dt):
    """GPU drop-in for `update_efield_3d` (`core/maxwell/cpu.py:83-98`)"""
    L, dev = lib(), _device()
    g = _GridOnDevice(f, dev)
    check(L.lpa_fdtd_e_3d(g.ref(), dt, constants.EPSILON_0, _stream(dev)), "lpa_fdtd_e_3d")
    g.download(["ex", "ey", "ez"])


def update_bfield_3d(f, dt):
    """GPU drop-in for `update_bfield_3d` (`core/maxwell/cpu.py:101-112`)"""
    L, dev = lib(), _device()
    g = _GridOnDevice(f, dev)
    check(L.lpa_fdtd_b_3d(g.ref(), dt, _stream(dev)), "lpa_fdtd_b_3d")
    g.download(["bx", "by", "bz"])
