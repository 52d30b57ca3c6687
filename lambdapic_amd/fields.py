"""Host-side mirror of the λPIC field containers (layout contract only).

Mirrors the attribute contract of the reference's ``Fields2D`` / ``Fields3D``
(`core/fields.py:6-170`): ten ``float64`` arrays ``ex ey ez bx by bz jx jy jz rho`` of
shape ``(nx+2ng, ny+2ng[, nz+2ng])`` in C order with the *wrapped guard layout*
(`core/fields.py:24-27`, `core/utils/cutils.h:19-26`): along every axis, interior cells
are ``[0, n)``, the upper guard is ``[n, n+ng)`` and the lower guard is ``[n+ng, n+2ng)``,
i.e. it is reached with negative indices.  ``x0, y0(, z0)`` is the position of node 0.

The device slab uses the conventional ``[ng | interior | ng]`` layout; the two are
related by a cyclic roll of ``ng`` along each axis (see ``to_device_layout``).
"""
from __future__ import annotations

import numpy as np

FIELD_ATTRS = ("ex", "ey", "ez", "bx", "by", "bz", "jx", "jy", "jz", "rho")


def _axis(n: int, ng: int, d: float, origin: float) -> np.ndarray:
    k = np.arange(n + 2 * ng, dtype=np.float64)
    k[n + ng:] -= n + 2 * ng          # lower guard carries negative node numbers
    return k * d + origin


class Fields:
    attrs = list(FIELD_ATTRS)

    def _alloc(self, attrs=None):
        if attrs is not None:
            self.attrs = list(attrs)
        for name in self.attrs:
            setattr(self, name, np.zeros(self.shape, dtype=np.float64))


class Fields2D(Fields):
    """2-D field bag; same constructor signature as the reference (`core/fields.py:78-112`)."""

    def __init__(self, nx, ny, dx, dy, x0, y0, n_guard, attrs=None):
        self.nx, self.ny = int(nx), int(ny)
        self.dx, self.dy = float(dx), float(dy)
        self.x0, self.y0 = float(x0), float(y0)
        self.n_guard = int(n_guard)
        self.shape = (self.nx + 2 * self.n_guard, self.ny + 2 * self.n_guard)
        self._alloc(attrs)
        self.xaxis = _axis(self.nx, self.n_guard, self.dx, self.x0)[:, None]
        self.yaxis = _axis(self.ny, self.n_guard, self.dy, self.y0)[None, :]


class Fields3D(Fields):
    """3-D field bag (`core/fields.py:115-170`)."""

    def __init__(self, nx, ny, nz, dx, dy, dz, x0, y0, z0, n_guard, attrs=None):
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self.dx, self.dy, self.dz = float(dx), float(dy), float(dz)
        self.x0, self.y0, self.z0 = float(x0), float(y0), float(z0)
        self.n_guard = int(n_guard)
        g = self.n_guard
        self.shape = (self.nx + 2 * g, self.ny + 2 * g, self.nz + 2 * g)
        self._alloc(attrs)
        self.xaxis = _axis(self.nx, g, self.dx, self.x0)[:, None, None]
        self.yaxis = _axis(self.ny, g, self.dy, self.y0)[None, :, None]
        self.zaxis = _axis(self.nz, g, self.dz, self.z0)[None, None, :]


def to_device_layout(a: np.ndarray, ng: int) -> np.ndarray:
    """wrapped-guard array -> conventional [ng|interior|ng] array (cyclic roll by +ng)."""
    return np.roll(a, ng, axis=tuple(range(a.ndim)))


def from_device_layout(a: np.ndarray, ng: int) -> np.ndarray:
    """conventional [ng|interior|ng] array -> wrapped-guard array (cyclic roll by -ng)."""
    return np.roll(a, -ng, axis=tuple(range(a.ndim)))
