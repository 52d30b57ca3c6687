"""oracle/cpml.py -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's CPML absorbing boundary (core/boundary/cpml.py) and of the
laser-injection boundary kernel (callback/laser.py:17-46), written at SLAB level: one Fields2D bag
covering a whole (rank-local) domain with per-axis coefficient arrays, instead of the reference's
per-patch objects.  The mapping is exact because
  * the reference runs the kappa-scaled update on every patch that carries a PML with kappa == 1
    outside the layer (cpml.py:95-100,343-377), and bfactor / 1.0 == bfactor, so one kappa-scaled
    update over the slab equals its mix of plain and kappa-scaled patches;
  * the psi recursions (cpml.py:531-606) run over the layer's [start, stop) along the normal axis and
    over the patch interior along the other; the union over the edge patches is the slab interior.
Pinned by tests/golden/g9_cpml_2d.npz (reference PML classes on 2x2 patches) and g10_laser_2d.npz.
"""
from __future__ import annotations

import numpy as np

from . import C_LIGHT, EPSILON_0


class SlabPML2D:
    """coefficients of cpml.py:23-125,233-340 laid out over a slab of nx x ny interior cells;
    ``sides`` subset of {'xmin','xmax','ymin','ymax'}"""

    def __init__(self, nx, ny, dx, dy, sides, thickness=6, kappa_max=20.0, a_max=0.15, sigma_max=0.7):
        self.nx, self.ny, self.dx, self.dy, self.t = nx, ny, dx, dy, thickness
        self.sides = set(sides)
        m, ma = 3, 1
        self.k = {}
        for ax, n, d in (("x", nx, dx), ("y", ny, dy)):
            # cpml.py:60 uses self.dx for EVERY axis (also for the y layers); reproduced as is
            smax = sigma_max * C_LIGHT * 0.8 * (m + 1.0) / dx
            for fld in ("e", "b"):
                self.k[fld + ax] = dict(kappa=np.ones(n), sigma=np.zeros(n), a=np.zeros(n))

            def fill(fld, pos, sl, smax=smax, ax=ax):
                c = self.k[fld + ax]
                c["kappa"][sl] = 1 + (kappa_max - 1) * pos ** m          # cpml.py:119-125
                c["sigma"][sl] = smax * pos ** m
                c["a"][sl] = a_max * (1 - pos) ** ma

            ar = np.arange(thickness, dtype=float)
            if ax + "min" in self.sides:                                  # cpml.py:233-250, 271-287
                fill("e", 1.0 - ar / thickness, np.s_[:thickness])
                fill("b", 1.0 - (ar + 0.5) / thickness, np.s_[:thickness])
            if ax + "max" in self.sides:                                  # cpml.py:253-269, 289-305
                fill("e", 1.0 - ar[::-1] / thickness, np.s_[n - thickness:n])
                fill("b", 1.0 - (ar + 0.5)[::-1] / thickness, np.s_[n - thickness - 1:n - 1])
        # psi arrays, full interior size like the reference's (only the layer rows are touched)
        self.psi = {k: np.zeros((nx, ny)) for k in
                    ("ey_x", "ez_x", "by_x", "bz_x", "ex_y", "ez_y", "bx_y", "bz_y")}

    def ranges(self, fld, ax):
        """[start, stop) of each layer along `ax` for field kind `fld` ('e' or 'b')"""
        n = self.nx if ax == "x" else self.ny
        out = []
        if ax + "min" in self.sides:
            out.append((0, self.t))
        if ax + "max" in self.sides:
            out.append((n - self.t, n) if fld == "e" else (n - self.t - 1, n - 1))
        return out

    def bc(self, fld, ax, dt, d):
        c = self.k[fld + ax]
        kap, sig, a = c["kappa"], c["sigma"], c["a"]
        b = np.exp(-(sig / kap + a) * dt)                                # cpml.py:537
        with np.errstate(invalid="ignore", divide="ignore"):
            cc = (b - 1) * sig / kap / (sig + kap * a) / d               # cpml.py:538
        return b, cc


def _sh(a, di, dj, nx, ny):
    """interior-shaped view of the wrapped-layout array `a` shifted by (di, dj) nodes"""
    i = (np.arange(nx) + di)[:, None]
    j = (np.arange(ny) + dj)[None, :]
    return a[i, j]     # negative indices reach the lower guards, n reaches the upper ones


def update_efield_cpml_2d(f, pml: SlabPML2D, dt):
    """cpml.py:343-360 on the slab, then the psi recursions of every layer (x before y)"""
    nx, ny = f.nx, f.ny
    bfac = dt * C_LIGHT ** 2
    jfac = dt / EPSILON_0
    bx_ = (bfac / pml.k["ex"]["kappa"])[:, None]
    by_ = (bfac / pml.k["ey"]["kappa"])[None, :]
    I = (slice(0, nx), slice(0, ny))
    bz, by, bx = f.bz, f.by, f.bx
    f.ex[I] += by_ * ((bz[I] - _sh(bz, 0, -1, nx, ny)) / f.dy) - jfac * f.jx[I]
    f.ey[I] += bx_ * (-(bz[I] - _sh(bz, -1, 0, nx, ny)) / f.dx) - jfac * f.jy[I]
    f.ez[I] += bx_ * ((by[I] - _sh(by, -1, 0, nx, ny)) / f.dx) \
        - by_ * ((bx[I] - _sh(bx, 0, -1, nx, ny)) / f.dy) - jfac * f.jz[I]
    fac = dt * C_LIGHT ** 2
    b, cc = pml.bc("e", "x", dt, f.dx)
    for s0, s1 in pml.ranges("e", "x"):                                   # cpml.py:531-548
        r = np.arange(s0, s1)
        pe, pz = pml.psi["ey_x"], pml.psi["ez_x"]
        pe[r] = b[r, None] * pe[r] + cc[r, None] * (bz[r, :ny] - bz[r - 1, :ny])
        pz[r] = b[r, None] * pz[r] + cc[r, None] * (by[r, :ny] - by[r - 1, :ny])
        f.ey[r, :ny] -= fac * pe[r]
        f.ez[r, :ny] += fac * pz[r]
    b, cc = pml.bc("e", "y", dt, f.dy)
    for s0, s1 in pml.ranges("e", "y"):                                   # cpml.py:569-586
        r = np.arange(s0, s1)
        px, pz = pml.psi["ex_y"], pml.psi["ez_y"]
        px[:, r] = b[None, r] * px[:, r] + cc[None, r] * (bz[:nx, r] - bz[:nx, r - 1])
        pz[:, r] = b[None, r] * pz[:, r] + cc[None, r] * (bx[:nx, r] - bx[:nx, r - 1])
        f.ex[:nx, r] += fac * px[:, r]
        f.ez[:nx, r] -= fac * pz[:, r]


def update_bfield_cpml_2d(f, pml: SlabPML2D, dt):
    """cpml.py:362-377 on the slab, then the psi recursions (cpml.py:550-567, 588-606)"""
    nx, ny = f.nx, f.ny
    ex_ = (dt / pml.k["bx"]["kappa"])[:, None]
    ey_ = (dt / pml.k["by"]["kappa"])[None, :]
    I = (slice(0, nx), slice(0, ny))
    ex, ey, ez = f.ex, f.ey, f.ez
    f.bx[I] -= ey_ * ((_sh(ez, 0, 1, nx, ny) - ez[I]) / f.dy)
    f.by[I] -= ex_ * (-(_sh(ez, 1, 0, nx, ny) - ez[I]) / f.dx)
    f.bz[I] -= ex_ * ((_sh(ey, 1, 0, nx, ny) - ey[I]) / f.dx) \
        - ey_ * ((_sh(ex, 0, 1, nx, ny) - ex[I]) / f.dy)
    b, cc = pml.bc("b", "x", dt, f.dx)
    for s0, s1 in pml.ranges("b", "x"):
        r = np.arange(s0, s1)
        py, pz = pml.psi["by_x"], pml.psi["bz_x"]
        py[r] = b[r, None] * py[r] + cc[r, None] * (ez[r + 1, :ny] - ez[r, :ny])
        pz[r] = b[r, None] * pz[r] + cc[r, None] * (ey[r + 1, :ny] - ey[r, :ny])
        f.by[r, :ny] += dt * py[r]
        f.bz[r, :ny] -= dt * pz[r]
    b, cc = pml.bc("b", "y", dt, f.dy)
    for s0, s1 in pml.ranges("b", "y"):
        r = np.arange(s0, s1)
        px, pz = pml.psi["bx_y"], pml.psi["bz_y"]
        px[:, r] = b[None, r] * px[:, r] + cc[None, r] * (ez[:nx, r + 1] - ez[:nx, r])
        pz[:, r] = b[None, r] * pz[:, r] + cc[None, r] * (ex[:nx, r + 1] - ex[:nx, r])
        f.bx[:nx, r] -= dt * px[:, r]
        f.bz[:nx, r] += dt * pz[:, r]


def laser_inject_2d(f, laserpos, dt, iy_start, iy_end, ey_source, ez_source):
    """callback/laser.py:17-46 (Mur-type injecting boundary on the B components one node inside
    the x-min layer); ey_source / ez_source are indexed like the field rows (length ny + 2 ng)"""
    c = C_LIGHT
    iy = np.arange(iy_start, iy_end)
    lp = laserpos
    f.bx[lp - 1, iy] = f.bx[0, iy]
    k = 1 / ((c * dt / f.dx + 1) * c)
    f.bz[lp - 1, iy] = k * (
        + 4 * ey_source[iy]
        + 2 * (f.ey[0, iy] + c * 0.5 * (f.bz[0, iy] + f.bz[-1, iy]))
        - 2 * f.ey[lp, iy]
        + dt / EPSILON_0 * f.jy[lp, iy]
        + (c * dt / f.dx - 1) * c * f.bz[lp, iy])
    f.by[lp - 1, iy] = k * (
        - 4 * ez_source[iy]
        - 2 * (f.ez[0, iy] - c * 0.5 * (f.by[0, iy] + f.by[-1, iy]))
        + 2 * f.ez[lp, iy]
        - (dt * c ** 2) * (f.bx[lp, iy] - f.bx[lp, iy - 1]) / f.dy
        - dt / EPSILON_0 * f.jz[lp, iy]
        + (c * dt / f.dx - 1) * c * f.by[lp, iy])


# ---- 3-D (cpml.py:431-475, 609-729; callback/laser.py:63-92) -------------------------------------------
class SlabPML3D:
    """coefficients over a slab of nx x ny x nz interior cells; ``sides`` subset of
    {'xmin','xmax','ymin','ymax','zmin','zmax'}.  Pinned by tests/golden/g12_cpml_3d.npz."""

    def __init__(self, nx, ny, nz, dx, dy, dz, sides, thickness=6, kappa_max=20.0, a_max=0.15, sigma_max=0.7):
        self.n, self.d, self.t = (nx, ny, nz), (dx, dy, dz), thickness
        self.sides = set(sides)
        m, ma = 3, 1
        smax = sigma_max * C_LIGHT * 0.8 * (m + 1.0) / dx          # cpml.py:60: dx for every axis
        self.k = {}
        ar = np.arange(thickness, dtype=float)
        for ax, n in zip("xyz", self.n):
            for fld in ("e", "b"):
                self.k[fld + ax] = dict(kappa=np.ones(n), sigma=np.zeros(n), a=np.zeros(n))

            def fill(fld, pos, sl, ax=ax):
                c = self.k[fld + ax]
                c["kappa"][sl] = 1 + (kappa_max - 1) * pos ** m
                c["sigma"][sl] = smax * pos ** m
                c["a"][sl] = a_max * (1 - pos) ** ma

            if ax + "min" in self.sides:
                fill("e", 1.0 - ar / thickness, np.s_[:thickness])
                fill("b", 1.0 - (ar + 0.5) / thickness, np.s_[:thickness])
            if ax + "max" in self.sides:
                fill("e", 1.0 - ar[::-1] / thickness, np.s_[n - thickness:n])
                fill("b", 1.0 - (ar + 0.5)[::-1] / thickness, np.s_[n - thickness - 1:n - 1])
        names = {"e": {"x": ("ey_x", "ez_x"), "y": ("ex_y", "ez_y"), "z": ("ex_z", "ey_z")},
                 "b": {"x": ("by_x", "bz_x"), "y": ("bx_y", "bz_y"), "z": ("bx_z", "by_z")}}
        self.names = names
        self.psi = {k: np.zeros(self.n) for f in names.values() for pair in f.values() for k in pair}

    def ranges(self, fld, ax):
        n = self.n["xyz".index(ax)]
        out = []
        if ax + "min" in self.sides:
            out.append((0, self.t))
        if ax + "max" in self.sides:
            out.append((n - self.t, n) if fld == "e" else (n - self.t - 1, n - 1))
        return out

    def bc(self, fld, ax, dt):
        c = self.k[fld + ax]
        kap, sig, a = c["kappa"], c["sigma"], c["a"]
        d = self.d["xyz".index(ax)]
        b = np.exp(-(sig / kap + a) * dt)
        with np.errstate(invalid="ignore", divide="ignore"):
            cc = (b - 1) * sig / kap / (sig + kap * a) / d
        return b, cc


def _sh3(a, s, n):
    """interior-shaped view of the wrapped-layout array `a` shifted by s = (di, dj, dk) nodes"""
    idx = np.ix_(*[np.arange(m) + o for m, o in zip(n, s)])
    return a[idx]


# (field read by psi_a, by psi_b, target of psi_a, its sign, target of psi_b, its sign) per normal axis
_E3 = {"x": ("bz", "by", "ey", -1, "ez", +1), "y": ("bz", "bx", "ex", +1, "ez", -1), "z": ("by", "bx", "ex", -1, "ey", +1)}
_B3 = {"x": ("ez", "ey", "by", +1, "bz", -1), "y": ("ez", "ex", "bx", -1, "bz", +1), "z": ("ey", "ex", "bx", +1, "by", -1)}


def _psi3(f, pml, fld, dt):
    n = pml.n
    fac = dt * C_LIGHT ** 2 if fld == "e" else dt
    tab = _E3 if fld == "e" else _B3
    for ai, ax in enumerate("xyz"):                      # x layers, then y, then z (PML list order)
        b, cc = pml.bc(fld, ax, dt)
        f1n, f2n, tan, sa, tbn, sb = tab[ax]
        pa, pb = (pml.psi[k] for k in pml.names[fld][ax])
        for s0, s1 in pml.ranges(fld, ax):
            r = np.arange(s0, s1)
            sl = [np.arange(m) for m in n]
            sl[ai] = r
            I = np.ix_(*sl)
            lo = [np.arange(m) for m in n]
            lo[ai] = r - 1 if fld == "e" else r + 1
            J = np.ix_(*lo)
            shape = [1, 1, 1]
            shape[ai] = r.size
            bb, cb = b[r].reshape(shape), cc[r].reshape(shape)
            f1, f2 = getattr(f, f1n), getattr(f, f2n)
            if fld == "e":
                pa[I] = bb * pa[I] + cb * (f1[I] - f1[J])
                pb[I] = bb * pb[I] + cb * (f2[I] - f2[J])
            else:
                pa[I] = bb * pa[I] + cb * (f1[J] - f1[I])
                pb[I] = bb * pb[I] + cb * (f2[J] - f2[I])
            getattr(f, tan)[I] += sa * (fac * pa[I])
            getattr(f, tbn)[I] += sb * (fac * pb[I])


def update_efield_cpml_3d(f, pml: SlabPML3D, dt):
    n = pml.n
    bfac = dt * C_LIGHT ** 2
    jfac = dt / EPSILON_0
    bx_ = (bfac / pml.k["ex"]["kappa"])[:, None, None]
    by_ = (bfac / pml.k["ey"]["kappa"])[None, :, None]
    bz_ = (bfac / pml.k["ez"]["kappa"])[None, None, :]
    I = tuple(slice(0, m) for m in n)
    bx, by, bz = f.bx, f.by, f.bz
    f.ex[I] += (by_ * (bz[I] - _sh3(bz, (0, -1, 0), n)) / f.dy - bz_ * (by[I] - _sh3(by, (0, 0, -1), n)) / f.dz) \
        - jfac * f.jx[I]
    f.ey[I] += (bz_ * (bx[I] - _sh3(bx, (0, 0, -1), n)) / f.dz - bx_ * (bz[I] - _sh3(bz, (-1, 0, 0), n)) / f.dx) \
        - jfac * f.jy[I]
    f.ez[I] += (bx_ * (by[I] - _sh3(by, (-1, 0, 0), n)) / f.dx - by_ * (bx[I] - _sh3(bx, (0, -1, 0), n)) / f.dy) \
        - jfac * f.jz[I]
    _psi3(f, pml, "e", dt)


def update_bfield_cpml_3d(f, pml: SlabPML3D, dt):
    n = pml.n
    ex_ = (dt / pml.k["bx"]["kappa"])[:, None, None]
    ey_ = (dt / pml.k["by"]["kappa"])[None, :, None]
    ez_ = (dt / pml.k["bz"]["kappa"])[None, None, :]
    I = tuple(slice(0, m) for m in n)
    ex, ey, ez = f.ex, f.ey, f.ez
    f.bx[I] -= (ey_ * (_sh3(ez, (0, 1, 0), n) - ez[I]) / f.dy - ez_ * (_sh3(ey, (0, 0, 1), n) - ey[I]) / f.dz)
    f.by[I] -= (ez_ * (_sh3(ex, (0, 0, 1), n) - ex[I]) / f.dz - ex_ * (_sh3(ez, (1, 0, 0), n) - ez[I]) / f.dx)
    f.bz[I] -= (ex_ * (_sh3(ey, (1, 0, 0), n) - ey[I]) / f.dx - ey_ * (_sh3(ex, (0, 1, 0), n) - ex[I]) / f.dy)
    _psi3(f, pml, "b", dt)


def laser_inject_3d(f, laserpos, dt, iy_start, iy_end, iz_start, iz_end, ey_source, ez_source):
    """callback/laser.py:63-92; ey_source / ez_source indexed like the field planes [ny+2ng][nz+2ng]"""
    c = C_LIGHT
    lp = laserpos
    iy = np.arange(iy_start, iy_end)[:, None]
    iz = np.arange(iz_start, iz_end)[None, :]
    f.bx[lp - 1, iy_start:iy_end, :] = f.bx[0, iy_start:iy_end, :]
    k = 1 / ((c * dt / f.dx + 1) * c)
    f.bz[lp - 1, iy, iz] = k * (
        + 4 * ey_source[iy, iz]
        + 2 * (f.ey[0, iy, iz] + c * 0.5 * (f.bz[0, iy, iz] + f.bz[-1, iy, iz]))
        - 2 * f.ey[lp, iy, iz]
        - (dt * c ** 2) * (f.bx[lp, iy, iz] - f.bx[lp, iy, iz - 1]) / f.dz
        + dt / EPSILON_0 * f.jy[lp, iy, iz]
        + (c * dt / f.dx - 1) * c * f.bz[lp, iy, iz])
    f.by[lp - 1, iy, iz] = k * (
        - 4 * ez_source[iy, iz]
        - 2 * (f.ez[0, iy, iz] - c * 0.5 * (f.by[0, iy, iz] + f.by[-1, iy, iz]))
        + 2 * f.ez[lp, iy, iz]
        - (dt * c ** 2) * (f.bx[lp, iy, iz] - f.bx[lp, iy - 1, iz]) / f.dy
        - dt / EPSILON_0 * f.jz[lp, iy, iz]
        + (c * dt / f.dx - 1) * c * f.by[lp, iy, iz])
