"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot-path algorithms (``picoracle.c`` + the numpy patch
synchronisation in ``sync.py``) and the loader for the reference's own compiled kernels
(``oracle/_ref``, built from the read-only tree by ``oracle/Makefile``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package -- as the checker / the reported CPU baseline, never as the thing measured or shipped.
The product package ``lambdapic_amd`` never imports it.

Parity status: pinned (see the header of ``picoracle.c``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_pp = C.POINTER(C.c_void_p)


def build(native: bool = False, force: bool = False) -> Path:
    """Compile picoracle.c with gcc.  ``native`` adds -O3 -march=native (CPU-baseline timing on the
    machine it runs on); the default is a portable build so the prebuilt .so can travel."""
    out = HERE / ("liboracle_native.so" if native else "liboracle.so")
    src = HERE / "picoracle.c"
    if out.exists() and not force and out.stat().st_mtime >= src.stat().st_mtime:
        return out
    # checker build: strict IEEE evaluation (no FMA contraction), portable ISA.
    # baseline build: the reference's own optimisation flags (setup.py:13), FMA contraction allowed.
    flags = ["-O3", "-march=native", "-ftree-vectorize", "-fno-trapping-math"] if native \
        else ["-O2", "-ffp-contract=off"]
    cmd = ["gcc", "-shared", "-fPIC", "-fopenmp", "-fno-math-errno", *flags, "-o", str(out), str(src), "-lm"]
    subprocess.run(cmd, check=True)
    return out


def lib(native: bool = False):
    global _LIB
    if _LIB is not None and not native:
        return _LIB
    path = build(native=native)
    L = C.CDLL(str(path))
    L.orc_num_threads.restype = C.c_int
    if not native:
        _LIB = L
    return L


def _p(a):
    """numpy array -> void* (arrays must be C-contiguous and stay alive during the call)."""
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def _tab(arrs):
    t = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    return t


# scipy.constants of the scipy the golden vectors were generated with (1.15.3, CODATA 2022)
EPSILON_0 = 8.8541878188e-12
MU_0 = 1.25663706127e-06
M_E = 9.1093837139e-31
E_CHARGE = 1.602176634e-19
C_LIGHT = 299792458.0

FIELD_ORDER = ("ex", "ey", "ez", "bx", "by", "bz", "jx", "jy", "jz", "rho")
PART_EB = ("ex_part", "ey_part", "ez_part", "bx_part", "by_part", "bz_part")


def _dead(p):
    return p.is_dead.view(np.uint8)


# -------------------------------------------------------------------------------------------------
# kernel-level entry points with the reference's call signatures (duck-typed particle / field bags)
# -------------------------------------------------------------------------------------------------
def unified_boris_pusher_cpu_2d(particles_list, fields_list, npatches, dt, q, m, native=False):
    """restates core/pusher/unified/unified_pusher_2d.c:157-365 (OpenMP over patches, static
    schedule, like the reference's `#pragma omp parallel for`, :213-214)"""
    L = lib(native)
    if npatches <= 0:
        return
    P, F = particles_list[:npatches], fields_list[:npatches]
    f0 = F[0]
    npart = (C.c_long * npatches)(*[p.npart for p in P])
    tabs = [_tab([getattr(p, a) for p in P]) for a in ("x", "y", "ux", "uy", "uz", "inv_gamma", "w")]
    dead = _tab([_dead(p) for p in P])
    part_eb = _tab([getattr(p, n) for p in P for n in PART_EB])
    fields = _tab([getattr(f, n) for f in F for n in FIELD_ORDER])
    x0 = (C.c_double * npatches)(*[f.x0 for f in F])
    y0 = (C.c_double * npatches)(*[f.y0 for f in F])
    L.orc_unified_2d_patches(C.c_long(npatches), npart, *tabs, dead, part_eb, fields, x0, y0,
                             C.c_long(f0.nx), C.c_long(f0.ny), C.c_long(f0.n_guard),
                             C.c_double(f0.dx), C.c_double(f0.dy), C.c_double(dt), C.c_double(q),
                             C.c_double(m))


def unified_boris_pusher_cpu_3d(particles_list, fields_list, npatches, dt, q, m):
    """restates core/pusher/unified/unified_pusher_3d.c:219-436"""
    L = lib()
    for p, f in zip(particles_list[:npatches], fields_list[:npatches]):
        eb = _tab([getattr(f, n) for n in FIELD_ORDER[:6]])
        pe = _tab([getattr(p, n) for n in PART_EB])
        L.orc_unified_3d(C.c_long(p.npart), _p(p.x), _p(p.y), _p(p.z), _p(p.ux), _p(p.uy),
                         _p(p.uz), _p(p.inv_gamma), _p(p.w), _p(_dead(p)), pe, eb,
                         _p(f.rho), _p(f.jx), _p(f.jy), _p(f.jz),
                         C.c_long(f.nx), C.c_long(f.ny), C.c_long(f.nz), C.c_long(f.n_guard),
                         C.c_double(f.dx), C.c_double(f.dy), C.c_double(f.dz),
                         C.c_double(f.x0), C.c_double(f.y0), C.c_double(f.z0),
                         C.c_double(dt), C.c_double(q), C.c_double(m))


def interpolation_patches_2d(particles_list, fields_list, npatches):
    """restates core/interpolation/cpu2d.c:71-136"""
    L = lib()
    for p, f in zip(particles_list[:npatches], fields_list[:npatches]):
        eb = _tab([getattr(f, n) for n in FIELD_ORDER[:6]])
        pe = _tab([getattr(p, n) for n in PART_EB])
        L.orc_interpolate_2d(C.c_long(p.npart), _p(p.x), _p(p.y), _p(_dead(p)), pe, eb,
                             C.c_long(f.nx), C.c_long(f.ny), C.c_long(f.n_guard),
                             C.c_double(f.dx), C.c_double(f.dy), C.c_double(f.x0), C.c_double(f.y0))


def interpolation_patches_3d(particles_list, fields_list, npatches):
    """restates core/interpolation/cpu3d.c:99-169"""
    L = lib()
    for p, f in zip(particles_list[:npatches], fields_list[:npatches]):
        eb = _tab([getattr(f, n) for n in FIELD_ORDER[:6]])
        pe = _tab([getattr(p, n) for n in PART_EB])
        L.orc_interpolate_3d(C.c_long(p.npart), _p(p.x), _p(p.y), _p(p.z), _p(_dead(p)), pe, eb,
                             C.c_long(f.nx), C.c_long(f.ny), C.c_long(f.nz), C.c_long(f.n_guard),
                             C.c_double(f.dx), C.c_double(f.dy), C.c_double(f.dz),
                             C.c_double(f.x0), C.c_double(f.y0), C.c_double(f.z0))


def current_deposition_cpu_2d(fields_list, particles_list, npatches, dt, q):
    """restates core/current/cpu2d.c:74-184 (standalone, non-fast factor grouping)"""
    L = lib()
    for p, f in zip(particles_list[:npatches], fields_list[:npatches]):
        L.orc_deposit_2d(C.c_long(p.npart), _p(p.x), _p(p.y), _p(p.ux), _p(p.uy), _p(p.uz),
                         _p(p.inv_gamma), _p(p.w), _p(_dead(p)),
                         _p(f.rho), _p(f.jx), _p(f.jy), _p(f.jz),
                         C.c_long(f.nx), C.c_long(f.ny), C.c_long(f.n_guard),
                         C.c_double(f.dx), C.c_double(f.dy), C.c_double(f.x0), C.c_double(f.y0),
                         C.c_double(dt), C.c_double(q))


def current_deposition_cpu_3d(fields_list, particles_list, npatches, dt, q):
    """restates core/current/cpu3d.c:118-183"""
    L = lib()
    for p, f in zip(particles_list[:npatches], fields_list[:npatches]):
        L.orc_deposit_3d(C.c_long(p.npart), _p(p.x), _p(p.y), _p(p.z), _p(p.ux), _p(p.uy),
                         _p(p.uz), _p(p.inv_gamma), _p(p.w), _p(_dead(p)),
                         _p(f.rho), _p(f.jx), _p(f.jy), _p(f.jz),
                         C.c_long(f.nx), C.c_long(f.ny), C.c_long(f.nz), C.c_long(f.n_guard),
                         C.c_double(f.dx), C.c_double(f.dy), C.c_double(f.dz),
                         C.c_double(f.x0), C.c_double(f.y0), C.c_double(f.z0),
                         C.c_double(dt), C.c_double(q))


def reset_current(fields_list, npatches):
    """restates core/current/cpu2d.c:19-72 / cpu3d.c:185-240 (zero jx,jy,jz,rho incl. guards)"""
    for f in fields_list[:npatches]:
        for n in ("jx", "jy", "jz", "rho"):
            getattr(f, n).fill(0.0)


def boris_push(p, q, m, dt):
    """restates core/pusher/cpu.py:11-35 + boris.py:6-48 on one particle bag"""
    pe = _tab([getattr(p, n) for n in PART_EB])
    lib().orc_boris(C.c_long(p.npart), _p(p.ux), _p(p.uy), _p(p.uz), _p(p.inv_gamma), pe,
                    _p(_dead(p)), C.c_double(q), C.c_double(m), C.c_double(dt))


def push_position_2d(p, dt):
    """restates core/pusher/cpu.py:58-91"""
    lib().orc_push_position_2d(C.c_long(p.npart), _p(p.x), _p(p.y), _p(p.ux), _p(p.uy),
                               _p(p.inv_gamma), _p(_dead(p)), C.c_double(dt))


def update_efield_2d(f, dt):
    """restates core/maxwell/cpu.py:9-22 on one field bag"""
    lib().orc_fdtd_e_2d(_p(f.ex), _p(f.ey), _p(f.ez), _p(f.bx), _p(f.by), _p(f.bz),
                        _p(f.jx), _p(f.jy), _p(f.jz), C.c_long(f.nx), C.c_long(f.ny),
                        C.c_long(f.n_guard), C.c_double(f.dx), C.c_double(f.dy), C.c_double(dt),
                        C.c_double(EPSILON_0))


def update_bfield_2d(f, dt):
    """restates core/maxwell/cpu.py:25-35"""
    lib().orc_fdtd_b_2d(_p(f.ex), _p(f.ey), _p(f.ez), _p(f.bx), _p(f.by), _p(f.bz),
                        C.c_long(f.nx), C.c_long(f.ny), C.c_long(f.n_guard),
                        C.c_double(f.dx), C.c_double(f.dy), C.c_double(dt))


def update_efield_3d(f, dt):
    """restates core/maxwell/cpu.py:83-98"""
    lib().orc_fdtd_e_3d(_p(f.ex), _p(f.ey), _p(f.ez), _p(f.bx), _p(f.by), _p(f.bz),
                        _p(f.jx), _p(f.jy), _p(f.jz), C.c_long(f.nx), C.c_long(f.ny),
                        C.c_long(f.nz), C.c_long(f.n_guard), C.c_double(f.dx), C.c_double(f.dy),
                        C.c_double(f.dz), C.c_double(dt), C.c_double(EPSILON_0))


def update_bfield_3d(f, dt):
    """restates core/maxwell/cpu.py:101-112"""
    lib().orc_fdtd_b_3d(_p(f.ex), _p(f.ey), _p(f.ez), _p(f.bx), _p(f.by), _p(f.bz),
                        C.c_long(f.nx), C.c_long(f.ny), C.c_long(f.nz), C.c_long(f.n_guard),
                        C.c_double(f.dx), C.c_double(f.dy), C.c_double(f.dz), C.c_double(dt))


def sync_guard_fields_2d_c(fields_list, patches_list, attrs, npatches, nx, ny, ng, native=False):
    """C/OpenMP twin of oracle.sync.sync_guard_fields_2d (core/patch/sync_fields2d.c:150-255)"""
    L = lib(native)
    nb = np.ascontiguousarray(np.stack([p.neighbor_ipatch for p in patches_list[:npatches]]).astype(np.int64))
    for a in attrs:
        L.orc_sync_guard_2d_patches(C.c_long(npatches), _tab([getattr(f, a) for f in fields_list[:npatches]]),
                                    _p(nb), C.c_long(nx), C.c_long(ny), C.c_long(ng))


def sync_currents_2d_c(fields_list, patches_list, npatches, nx, ny, ng, native=False):
    """C/OpenMP twin of oracle.sync.sync_currents_2d (core/patch/sync_fields2d.c:43-148)"""
    L = lib(native)
    nb = np.ascontiguousarray(np.stack([p.neighbor_ipatch for p in patches_list[:npatches]]).astype(np.int64))
    for a in ("jx", "jy", "jz", "rho"):
        L.orc_sync_currents_2d_patches(C.c_long(npatches), _tab([getattr(f, a) for f in fields_list[:npatches]]),
                                       _p(nb), C.c_long(nx), C.c_long(ny), C.c_long(ng))


def bucket_index_2d(x, y, is_dead, nx, ny, dx, dy, x0, y0, reverse_x=False):
    """restates core/sort/cpu2d.c:9-54; returns (particle_index, bucket_count)"""
    n = x.size
    index = np.empty(n, dtype=np.int64)
    count = np.zeros(nx * ny, dtype=np.int64)
    lib().orc_bucket_index_2d(C.c_long(n), _p(x), _p(y), _p(is_dead.view(np.uint8)),
                              C.c_long(nx), C.c_long(ny), C.c_double(dx), C.c_double(dy),
                              C.c_double(x0), C.c_double(y0), _p(index), _p(count),
                              C.c_int(int(reverse_x)))
    return index, count


def bucket_index_nd(pos, is_dead, nb, d, origin, reverse_x=False):
    """restates core/sort/cpu2d.c:9-54 / cpu3d.c:8-58 for 2 or 3 axes in numpy (pure-Python loop only for the
    dead-slot inheritance): bucket = floor((r - r0) / d) per axis, z fastest; out of range -> last bucket (or clamped
    when the x order is mirrored); a dead slot inherits the bucket of the slot before it (0 at the start).
    Returns (particle_index, bucket_count)."""
    dim = len(nb)
    idx = [np.floor((pos[a] - origin[a]) / d[a]).astype(np.int64) for a in range(dim)]
    nbin = int(np.prod(nb))
    if reverse_x:
        idx = [np.clip(idx[a], 0, nb[a] - 1) for a in range(dim)]
        idx[0] = nb[0] - 1 - idx[0]
        inside = np.ones(pos[0].size, dtype=bool)
    else:
        inside = np.logical_and.reduce([(idx[a] >= 0) & (idx[a] < nb[a]) for a in range(dim)])
    lin = idx[0]
    for a in range(1, dim):
        lin = lin * nb[a] + idx[a]
    lin = np.where(inside, lin, nbin - 1)
    out = np.empty(pos[0].size, dtype=np.int64)
    run = 0
    for ip in range(out.size):
        if not is_dead[ip]:
            run = lin[ip]
        out[ip] = run
    return out, np.bincount(out, minlength=nbin)


# -------------------------------------------------------------------------------------------------
# reference's own compiled kernels (oracle/_ref); only present where `make -C oracle ref` ran
# -------------------------------------------------------------------------------------------------
def ref_available() -> bool:
    return any((HERE / "_ref").glob("pusher/unified_pusher_2d*.so"))


def ref_module(group: str, name: str, portable: bool = False):
    """import oracle/_ref/<group>/<name>.<EXT_SUFFIX> as a CPython module (no reference source
    is read at run time -- these are binaries this repo's Makefile compiled).  ``portable``: the -march=x86-64-v3 build
    (oracle/_ref/portable/, `make ref_portable`) for hosts whose CPU is not the build container's."""
    import importlib.machinery
    import importlib.util
    import sysconfig

    path = HERE / "_ref" / ("portable/" if portable else "") / group / (name + sysconfig.get_config_var("EXT_SUFFIX"))
    if not path.exists():
        raise FileNotFoundError(f"{path} missing: run `make -C oracle ref` in the build container")
    loader = importlib.machinery.ExtensionFileLoader(name, str(path))
    spec = importlib.util.spec_from_file_location(name, str(path), loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod
