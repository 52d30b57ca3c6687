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
