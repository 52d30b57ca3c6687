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
    padded = order == _lib.LPA_ORDER_PADDED
    # LPA_ORDER_PADDED stores holes: every tile may grow by its padded stripes and a 64-slot rounding
    ntiles = -(-g.c.nx // _lib.LPA_TILE_X) * -(-g.c.ny // _lib.LPA_TILE_Y)
    cap = 2 * nl + 320 * ntiles if padded else nl
    dst = torch.empty((src.shape[0], cap), dtype=src.dtype, device=dev)
    sid = torch.from_numpy(live.astype(np.int64)).to(dev)
    did = torch.empty(cap, dtype=torch.int64, device=dev)

    def cs(t, tid, n):
        c = _lib.lpa_particles()
        c.n = n
        for a in PART_CORE:
            setattr(c, a, t[names.index(a)].data_ptr())
        c.z = None
        for k, a in enumerate(PART_EB):
            c.part_eb[k] = t[names.index(a)].data_ptr()
        c.id, c.is_dead = tid.data_ptr(), None
        return c

    ps, pd = cs(src, sid, nl), cs(dst, did, cap)
    nbytes = L.lpa_sort_workspace_bytes(g.ref(), cap)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    tiling = _lib.lpa_tiling()
    st = _stream(dev)
    check(L.lpa_sort_tiles_2d(g.ref(), C.byref(ps), C.byref(pd), ws.data_ptr(), nbytes, block_particles,
                              order, C.byref(tiling), st), "lpa_sort_tiles_2d")
    n_slots = _lib.sort_result(L, ws)                          # live particles (+ holes of the padded order)
    assert n_slots == nl or (padded and nl <= n_slots <= cap)
    tiling.n_sorted = n_slots
    pd.n = n_slots
    if padded:
        # the cooperative deposit parks what it cannot deposit in the main loop: give it the scratch arrays
        # (and no per-particle E / B write-back: that variant has no cooperative form)
        scratch = torch.empty((7, n_slots), dtype=torch.float64, device=dev)
        for c in range(7):
            tiling.scratch[c] = scratch[c].data_ptr()
        for k in range(6):
            pd.part_eb[k] = None
    overflow = torch.empty(max(n_slots, 1), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    check(L.lpa_push_deposit_tiled_2d(g.ref(), C.byref(pd), C.byref(pp), C.byref(tiling),
                                      overflow.data_ptr(), cnt.data_ptr(), st), "lpa_push_deposit_tiled_2d")
    check(L.lpa_push_deposit_list_2d(g.ref(), C.byref(pd), C.byref(pp), overflow.data_ptr(), cnt.data_ptr(),
                                     n_slots, st), "lpa_push_deposit_list_2d")
    out = dst[:, :n_slots].cpu().numpy()
    slot = did[:n_slots].cpu().numpy()
    keep = ~np.isnan(out[names.index("x")])                     # (holes of the padded order)
    assert int(keep.sum()) == nl
    for a in list(PART_CORE[:6]) + ([] if padded else list(PART_EB)):
        getattr(p, a)[slot[keep]] = out[names.index(a)][keep]
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


# ---- patch-list drop-ins (lpa_patches.hip): the reference's data model kept on the device ----------------------
def _patch_arrays_on_device(fields_list, attrs, npatches, dev):
    """stack the (wrapped-layout) arrays of ``attrs`` of every patch into one device tensor
    [npatches][len(attrs)][NX][NY] + the device pointer table the kernels index"""
    host = np.stack([np.stack([np.ascontiguousarray(getattr(f, a), dtype=np.float64) for a in attrs])
                     for f in fields_list[:npatches]])
    buf = torch.from_numpy(host).to(dev)
    ptrs = torch.tensor([buf[p, c].data_ptr() for p in range(npatches) for c in range(len(attrs))],
                        dtype=torch.int64, device=dev)
    return buf, ptrs


def _neighbor_table(patches_list, npatches, dev, nbound=8):
    nb = np.stack([np.asarray(p.neighbor_ipatch, dtype=np.int64) for p in patches_list[:npatches]])
    if nb.shape != (npatches, nbound):
        raise ValueError(f"neighbor_ipatch must have the {nbound} entries of Boundary{2 if nbound == 8 else 3}D")
    if (nb >= npatches).any():
        raise ValueError("neighbor_ipatch points outside the patch list")
    return torch.from_numpy(nb).to(dev)


def sync_guard_fields_2d(fields_list, patches_list, attrs, npatches, nx, ny, ng):
    """GPU drop-in for `core/patch/sync_fields2d.c:150-255`
    (``sync_guard_fields_2d(fields_list, patches_list, attrs, npatches, nx, ny, ng) -> None``)"""
    L, dev = lib(), _device()
    attrs = list(attrs)
    if npatches <= 0 or not attrs:
        return None
    buf, ptrs = _patch_arrays_on_device(fields_list, attrs, npatches, dev)
    nb = _neighbor_table(patches_list, npatches, dev)
    check(L.lpa_sync_guard_fields_2d(ptrs.data_ptr(), len(attrs), nb.data_ptr(), npatches, nx, ny, ng, _stream(dev)),
          "lpa_sync_guard_fields_2d")
    out = buf.cpu().numpy()
    for p, f in enumerate(fields_list[:npatches]):
        for c, a in enumerate(attrs):
            getattr(f, a)[...] = out[p, c]
    return None


def sync_currents_2d(fields_list, patches_list, npatches, nx, ny, ng):
    """GPU drop-in for `core/patch/sync_fields2d.c:43-148`
    (``sync_currents_2d(fields_list, patches_list, npatches, nx, ny, ng) -> None``)"""
    L, dev = lib(), _device()
    if npatches <= 0:
        return None
    attrs = ["jx", "jy", "jz", "rho"]
    buf, ptrs = _patch_arrays_on_device(fields_list, attrs, npatches, dev)
    nb = _neighbor_table(patches_list, npatches, dev)
    check(L.lpa_sync_currents_2d(ptrs.data_ptr(), nb.data_ptr(), npatches, nx, ny, ng, _stream(dev)),
          "lpa_sync_currents_2d")
    out = buf.cpu().numpy()
    for p, f in enumerate(fields_list[:npatches]):
        for c, a in enumerate(attrs):
            getattr(f, a)[...] = out[p, c]
    return None


def sync_guard_fields_3d(fields_list, patches_list, attrs, npatches, nx, ny, nz, ng):
    """GPU drop-in for `core/patch/sync_fields3d.c:350-612`
    (``sync_guard_fields_3d(fields_list, patches_list, attrs, npatches, nx, ny, nz, ng) -> None``)"""
    L, dev = lib(), _device()
    attrs = list(attrs)
    if npatches <= 0 or not attrs:
        return None
    buf, ptrs = _patch_arrays_on_device(fields_list, attrs, npatches, dev)
    nb = _neighbor_table(patches_list, npatches, dev, 26)
    check(L.lpa_sync_guard_fields_3d(ptrs.data_ptr(), len(attrs), nb.data_ptr(), npatches, nx, ny, nz, ng,
                                     _stream(dev)), "lpa_sync_guard_fields_3d")
    out = buf.cpu().numpy()
    for p, f in enumerate(fields_list[:npatches]):
        for c, a in enumerate(attrs):
            getattr(f, a)[...] = out[p, c]
    return None


def sync_currents_3d(fields_list, patches_list, npatches, nx, ny, nz, ng):
    """GPU drop-in for `core/patch/sync_fields3d.c:84-348`
    (``sync_currents_3d(fields_list, patches_list, npatches, nx, ny, nz, ng) -> None``)"""
    L, dev = lib(), _device()
    if npatches <= 0:
        return None
    attrs = ["jx", "jy", "jz", "rho"]
    buf, ptrs = _patch_arrays_on_device(fields_list, attrs, npatches, dev)
    nb = _neighbor_table(patches_list, npatches, dev, 26)
    check(L.lpa_sync_currents_3d(ptrs.data_ptr(), nb.data_ptr(), npatches, nx, ny, nz, ng, _stream(dev)),
          "lpa_sync_currents_3d")
    out = buf.cpu().numpy()
    for p, f in enumerate(fields_list[:npatches]):
        for c, a in enumerate(attrs):
            getattr(f, a)[...] = out[p, c]
    return None


def _bucket_sort_patch(L, dev, x, y, z, is_dead, attrs, x0, y0, z0, n, d, bucket_count, bmin, bmax, reverse_x):
    npart = int(x.shape[0])
    nbin = int(np.prod(n))
    dim3 = z is not None
    host = [x, y] + ([z] if dim3 else []) + [a for a in attrs if a is not x and a is not y and a is not z]
    data = torch.from_numpy(np.stack([np.asarray(a, dtype=np.float64) for a in host]) if npart else
                            np.zeros((len(host), 0))).to(dev)
    dead = torch.from_numpy(np.ascontiguousarray(is_dead[:npart]).view(np.uint8).copy()).to(dev)
    cnt = torch.zeros(3 * nbin, dtype=torch.int64, device=dev)
    nbytes = L.lpa_bucket_sort_workspace_bytes(npart, nbin)
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=dev)
    nbuf = torch.zeros(1, dtype=torch.int64, device=dev)
    ptr = lambda k: data[k].data_ptr() if npart else None
    table = (C.c_void_p * len(host))(*[ptr(k) for k in range(len(host))])
    check(L.lpa_bucket_sort(ptr(0), ptr(1), ptr(2) if dim3 else None, dead.data_ptr() if npart else None, table,
                            len(host), npart, n[0], n[1], n[2], d[0], d[1], d[2], x0, y0, z0, int(bool(reverse_x)),
                            cnt[:nbin].data_ptr(), cnt[nbin:2 * nbin].data_ptr(), cnt[2 * nbin:].data_ptr(),
                            ws.data_ptr(), ws.numel(), nbuf.data_ptr(), _stream(dev)), "lpa_bucket_sort")
    out = data.cpu().numpy()
    for k, a in enumerate(host):
        a[:npart] = out[k]
    is_dead[:npart] = dead.cpu().numpy().view(np.bool_)
    c = cnt.cpu().numpy()
    bucket_count.reshape(-1)[:] = c[:nbin]
    bmin.reshape(-1)[:] = c[nbin:2 * nbin]
    bmax.reshape(-1)[:] = c[2 * nbin:]
    return int(nbuf.item())


def sort_particles_patches_2d(x_list, y_list, is_dead_list, attrs_list, x0s, y0s, nx, ny, dx, dy, npatches,
                              bucket_count_list, bucket_bound_min_list, bucket_bound_max_list,
                              bucket_count_not_list, bucket_start_counter_list, particle_index_list,
                              particle_index_ref_list, particle_index_target_list, buf_list, reverse_x):
    """GPU drop-in for `core/sort/cpu2d.c:220-303` with the reference's argument list (the last six lists are
    the CPU algorithm's scratch and are left untouched).  Returns the number of slots moved; arrays are
    permuted in place, ``bucket_count`` / ``bucket_bound_min`` / ``bucket_bound_max`` are filled."""
    L, dev = lib(), _device()
    if npatches <= 0:
        return 0
    nattrs = len(attrs_list) // npatches
    moved = 0
    for ip in range(npatches):
        moved += _bucket_sort_patch(L, dev, x_list[ip], y_list[ip], None, is_dead_list[ip],
                                    attrs_list[ip * nattrs:(ip + 1) * nattrs], float(x0s[ip]), float(y0s[ip]), 0.0,
                                    (int(nx), int(ny), 1), (float(dx), float(dy), 1.0), bucket_count_list[ip],
                                    bucket_bound_min_list[ip], bucket_bound_max_list[ip], reverse_x)
    return moved


def sort_particles_patches_3d(x_list, y_list, z_list, is_dead_list, attrs_list, x0s, y0s, z0s, nx, ny, nz, dx, dy,
                              dz, npatches, bucket_count_list, bucket_bound_min_list, bucket_bound_max_list,
                              bucket_count_not_list, bucket_start_counter_list, particle_index_list,
                              particle_index_ref_list, particle_index_target_list, buf_list, reverse_x):
    """GPU drop-in for ``sort_particles_patches_3d`` (`core/sort/cpu3d.c`), see the 2-D twin"""
    L, dev = lib(), _device()
    if npatches <= 0:
        return 0
    nattrs = len(attrs_list) // npatches
    moved = 0
    for ip in range(npatches):
        moved += _bucket_sort_patch(L, dev, x_list[ip], y_list[ip], z_list[ip], is_dead_list[ip],
                                    attrs_list[ip * nattrs:(ip + 1) * nattrs], float(x0s[ip]), float(y0s[ip]),
                                    float(z0s[ip]), (int(nx), int(ny), int(nz)), (float(dx), float(dy), float(dz)),
                                    bucket_count_list[ip], bucket_bound_min_list[ip], bucket_bound_max_list[ip],
                                    reverse_x)
    return moved


# ---- particle ownership between the patches of a list (core/patch/sync_particles_2d.c) -------------------------
def _patch_bounds(patches_list, npatches, dx, dy):
    return np.array([[p.xmin - 0.5 * dx, p.xmax + 0.5 * dx, p.ymin - 0.5 * dy, p.ymax + 0.5 * dy]
                     for p in patches_list[:npatches]], dtype=np.float64)


class _PatchParticlesOnDevice:
    """the attribute arrays of every patch's particle bag, padded to the longest bag, + the pointer tables"""

    def __init__(self, particles_list, npatches, attrs, dev):
        self.parts, self.attrs, self.dev = particles_list[:npatches], list(attrs), dev
        self.npart = np.array([q.npart for q in self.parts], dtype=np.int64)
        self.nmax = int(self.npart.max()) if npatches else 0
        host = np.full((npatches, len(self.attrs), max(self.nmax, 1)), np.nan)
        dead = np.ones((npatches, max(self.nmax, 1)), dtype=np.uint8)
        for k, q in enumerate(self.parts):
            for a, name in enumerate(self.attrs):
                host[k, a, :q.npart] = getattr(q, name)[:q.npart]
            dead[k, :q.npart] = q.is_dead[:q.npart]
        self.data = torch.from_numpy(host).to(dev)
        self.dead = torch.from_numpy(dead).to(dev)
        self.ptrs = torch.tensor([self.data[k, a].data_ptr() for k in range(npatches) for a in range(len(self.attrs))],
                                 dtype=torch.int64, device=dev)
        self.dead_ptrs = torch.tensor([self.dead[k].data_ptr() for k in range(npatches)], dtype=torch.int64, device=dev)
        self.npart_dev = torch.from_numpy(self.npart).to(dev)

    def download(self):
        data, dead = self.data.cpu().numpy(), self.dead.cpu().numpy()
        for k, q in enumerate(self.parts):
            for a, name in enumerate(self.attrs):
                getattr(q, name)[:q.npart] = data[k, a, :q.npart]
            q.is_dead[:q.npart] = dead[k, :q.npart].view(np.bool_)


def get_npart_to_extend_2d(particles_list, patches_list, npatches, dx, dy):
    """GPU drop-in for `core/patch/sync_particles_2d.c:204-320`: returns ``(npart_to_extend, npart_incoming,
    npart_outgoing, npart_alive)`` (int64 arrays; npart_outgoing is [npatches * 8] in Boundary2D order)"""
    L, dev = lib(), _device()
    z = lambda n: np.zeros(n, dtype=np.int64)
    if npatches <= 0:
        return z(0), z(0), z(0), z(0)
    d = _PatchParticlesOnDevice(particles_list, npatches, ["x", "y"], dev)
    bounds = torch.from_numpy(_patch_bounds(patches_list, npatches, dx, dy)).to(dev)
    nout = torch.zeros(npatches * 8, dtype=torch.int64, device=dev)
    ndead = torch.zeros(npatches, dtype=torch.int64, device=dev)
    check(L.lpa_sync_particles_count_2d(d.ptrs.data_ptr(), d.dead_ptrs.data_ptr(), d.npart_dev.data_ptr(),
                                        bounds.data_ptr(), npatches, d.nmax, nout.data_ptr(), ndead.data_ptr(),
                                        _stream(dev)), "lpa_sync_particles_count_2d")
    nout_h, ndead_h = nout.cpu().numpy(), ndead.cpu().numpy()
    opp = [1, 0, 3, 2, 7, 6, 5, 4]                        # OPPOSITE_BOUNDARY, sync_particles_2d.c:26-35
    ext, inc, alive = z(npatches), z(npatches), z(npatches)
    for ip, p in enumerate(patches_list[:npatches]):      # sync_particles_2d.c:285-318
        new = sum(int(nout_h[int(nb) * 8 + opp[b]]) for b, nb in enumerate(p.neighbor_ipatch) if nb >= 0)
        npart = int(d.npart[ip])
        alive[ip] = npart - int(ndead_h[ip]) + new
        if new - int(ndead_h[ip]) > 0:
            ext[ip] = new - int(ndead_h[ip]) + int(npart * 0.25)
        inc[ip] = new
    return ext, inc, nout_h, alive


def fill_particles_from_boundary_2d(particles_list, patches_list, npart_incoming, npart_outgoing, npatches, dx, dy,
                                    xmin_global, xmax_global, ymin_global, ymax_global, attrs):
    """GPU drop-in for `core/patch/sync_particles_2d.c:322-518` (arrays are filled / killed in place)"""
    L, dev = lib(), _device()
    attrs = list(attrs)
    if "x" not in attrs or "y" not in attrs:
        raise ValueError("attrs must contain 'x' and 'y'")
    if npatches <= 0:
        return None
    d = _PatchParticlesOnDevice(particles_list, npatches, attrs, dev)
    bounds = torch.from_numpy(_patch_bounds(patches_list, npatches, dx, dy)).to(dev)
    nb = _neighbor_table(patches_list, npatches, dev)
    nin = torch.from_numpy(np.ascontiguousarray(npart_incoming, dtype=np.int64)).to(dev)
    nout = torch.from_numpy(np.ascontiguousarray(npart_outgoing, dtype=np.int64)).to(dev)
    nbytes = L.lpa_sync_particles_workspace_bytes(npatches, d.nmax)
    ws = torch.zeros(max(nbytes, 8), dtype=torch.uint8, device=dev)
    check(L.lpa_sync_particles_fill_2d(d.ptrs.data_ptr(), len(attrs), attrs.index("x"), attrs.index("y"),
                                       d.dead_ptrs.data_ptr(), d.npart_dev.data_ptr(), bounds.data_ptr(),
                                       nb.data_ptr(), nin.data_ptr(), nout.data_ptr(), npatches, d.nmax,
                                       xmin_global, xmax_global, ymin_global, ymax_global, dx, dy, ws.data_ptr(),
                                       ws.numel(), _stream(dev)), "lpa_sync_particles_fill_2d")
    d.download()
    return None


# ---- 3-D twins (core/patch/sync_particles_3d.c) ---------------------------------------------------------------
def _patch_bounds_3d(patches_list, npatches, d):
    return np.array([[getattr(p, ax + side) + sg * 0.5 * dd for ax, dd in zip("xyz", d) for side, sg in (("min", -1), ("max", 1))]
                     for p in patches_list[:npatches]], dtype=np.float64)


def _opposite_3d():
    from .patch import OPPOSITE_3D, Boundary3D
    return [int(OPPOSITE_3D[b]) for b in Boundary3D]


def get_npart_to_extend_3d(particles_list, patch_list, npatches, dx, dy, dz):
    """GPU drop-in for `core/patch/sync_particles_3d.c:365-482`: returns ``(npart_to_extend, npart_incoming,
    npart_outgoing, npart_alive)`` (int64 arrays; npart_outgoing is [npatches * 26] in Boundary3D order)"""
    L, dev = lib(), _device()
    z = lambda n: np.zeros(n, dtype=np.int64)
    if npatches <= 0:
        return z(0), z(0), z(0), z(0)
    d = _PatchParticlesOnDevice(particles_list, npatches, ["x", "y", "z"], dev)
    bounds = torch.from_numpy(_patch_bounds_3d(patch_list, npatches, (dx, dy, dz))).to(dev)
    nout = torch.zeros(npatches * 26, dtype=torch.int64, device=dev)
    ndead = torch.zeros(npatches, dtype=torch.int64, device=dev)
    check(L.lpa_sync_particles_count_3d(d.ptrs.data_ptr(), d.dead_ptrs.data_ptr(), d.npart_dev.data_ptr(),
                                        bounds.data_ptr(), npatches, d.nmax, nout.data_ptr(), ndead.data_ptr(),
                                        _stream(dev)), "lpa_sync_particles_count_3d")
    nout_h, ndead_h = nout.cpu().numpy(), ndead.cpu().numpy()
    opp = _opposite_3d()
    ext, inc, alive = z(npatches), z(npatches), z(npatches)
    for ip, p in enumerate(patch_list[:npatches]):        # sync_particles_3d.c:439-476
        new = sum(int(nout_h[int(nb) * 26 + opp[b]]) for b, nb in enumerate(p.neighbor_ipatch) if nb >= 0)
        npart = int(d.npart[ip])
        alive[ip] = npart - int(ndead_h[ip]) + new
        if new - int(ndead_h[ip]) > 0:
            ext[ip] = new - int(ndead_h[ip]) + int(npart * 0.25)
        inc[ip] = new
    return ext, inc, nout_h, alive


def fill_particles_from_boundary_3d(particles_list, patch_list, npart_incoming, npart_outgoing, npatches, dx, dy, dz,
                                    xmin_global, xmax_global, ymin_global, ymax_global, zmin_global, zmax_global,
                                    attrs):
    """GPU drop-in for `core/patch/sync_particles_3d.c:484-700` (arrays are filled / killed in place)"""
    L, dev = lib(), _device()
    attrs = list(attrs)
    if "x" not in attrs or "y" not in attrs or "z" not in attrs:
        raise ValueError("attrs must contain 'x', 'y', and 'z'")
    if npatches <= 0:
        return None
    d = _PatchParticlesOnDevice(particles_list, npatches, attrs, dev)
    bounds = torch.from_numpy(_patch_bounds_3d(patch_list, npatches, (dx, dy, dz))).to(dev)
    nb = _neighbor_table(patch_list, npatches, dev, 26)
    nin = torch.from_numpy(np.ascontiguousarray(npart_incoming, dtype=np.int64)).to(dev)
    nout = torch.from_numpy(np.ascontiguousarray(npart_outgoing, dtype=np.int64)).to(dev)
    nbytes = L.lpa_sync_particles_workspace_bytes(npatches, d.nmax)
    ws = torch.zeros(max(nbytes, 8), dtype=torch.uint8, device=dev)
    v3 = lambda *v: (C.c_double * 3)(*[float(x) for x in v])
    check(L.lpa_sync_particles_fill_3d(d.ptrs.data_ptr(), len(attrs), attrs.index("x"), attrs.index("y"),
                                       attrs.index("z"), d.dead_ptrs.data_ptr(), d.npart_dev.data_ptr(),
                                       bounds.data_ptr(), nb.data_ptr(), nin.data_ptr(), nout.data_ptr(), npatches,
                                       d.nmax, v3(xmin_global, ymin_global, zmin_global),
                                       v3(xmax_global, ymax_global, zmax_global), v3(dx, dy, dz), ws.data_ptr(),
                                       ws.numel(), _stream(dev)), "lpa_sync_particles_fill_3d")
    d.download()
    return None


# ---- numba-level Maxwell drivers on array lists (core/maxwell/cpu.py:38-79,115-158) ---------------------------------
class _ArrayBag:
    """the minimum of a Fields2D / Fields3D the grid upload reads, around caller-owned arrays (wrapped layout)"""

    def __init__(self, arrays, n, d, ng):
        self.nx, self.ny = n[0], n[1]
        self.dx, self.dy = d[0], d[1]
        self.x0 = self.y0 = self.z0 = 0.0
        if len(n) == 3:
            self.nz, self.dz = n[2], d[2]
        self.n_guard = ng
        zero = None
        for name in FIELD_ORDER:
            a = arrays.get(name)
            if a is None:
                zero = np.zeros_like(arrays["ex"]) if zero is None else zero
                a = zero
            setattr(self, name, a)


def _maxwell_patches(which, lists, npatches, n, d, dt, ng):
    L, dev = lib(), _device()
    dim = len(n)
    names = ("ex", "ey", "ez", "bx", "by", "bz", "jx", "jy", "jz")[:len(lists)]
    for ip in range(npatches):
        bag = _ArrayBag({a: lst[ip] for a, lst in zip(names, lists)}, n, d, ng)
        g = _GridOnDevice(bag, dev)
        st = _stream(dev)
        if which == "e":
            fn = L.lpa_fdtd_e_2d if dim == 2 else L.lpa_fdtd_e_3d
            check(fn(g.ref(), dt, constants.EPSILON_0, st), "lpa_fdtd_e")
            g.download(["ex", "ey", "ez"])
        else:
            fn = L.lpa_fdtd_b_2d if dim == 2 else L.lpa_fdtd_b_3d
            check(fn(g.ref(), dt, st), "lpa_fdtd_b")
            g.download(["bx", "by", "bz"])


def update_efield_patches_2d(ex_list, ey_list, ez_list, bx_list, by_list, bz_list, jx_list, jy_list, jz_list, npatches,
                             dx, dy, dt, nx, ny, n_guard):
    """GPU drop-in for `core/maxwell/cpu.py:38-59` (arrays in λPIC's wrapped layout, updated in place)"""
    _maxwell_patches("e", [ex_list, ey_list, ez_list, bx_list, by_list, bz_list, jx_list, jy_list, jz_list], npatches,
                     (nx, ny), (dx, dy), dt, n_guard)


def update_bfield_patches_2d(ex_list, ey_list, ez_list, bx_list, by_list, bz_list, npatches, dx, dy, dt, nx, ny, n_guard):
    """GPU drop-in for `core/maxwell/cpu.py:61-79`"""
    _maxwell_patches("b", [ex_list, ey_list, ez_list, bx_list, by_list, bz_list], npatches, (nx, ny), (dx, dy), dt, n_guard)


def update_efield_patches_3d(ex_list, ey_list, ez_list, bx_list, by_list, bz_list, jx_list, jy_list, jz_list, npatches,
                             dx, dy, dz, dt, nx, ny, nz, n_guard):
    """GPU drop-in for `core/maxwell/cpu.py:115-137`"""
    _maxwell_patches("e", [ex_list, ey_list, ez_list, bx_list, by_list, bz_list, jx_list, jy_list, jz_list], npatches,
                     (nx, ny, nz), (dx, dy, dz), dt, n_guard)


def update_bfield_patches_3d(ex_list, ey_list, ez_list, bx_list, by_list, bz_list, npatches, dx, dy, dz, dt, nx, ny, nz,
                             n_guard):
    """GPU drop-in for `core/maxwell/cpu.py:139-158`"""
    _maxwell_patches("b", [ex_list, ey_list, ez_list, bx_list, by_list, bz_list], npatches, (nx, ny, nz), (dx, dy, dz),
                     dt, n_guard)
