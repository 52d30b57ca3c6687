"""Shared builders for the parity tests: rebuild field / particle bags from golden fixtures."""
from __future__ import annotations

import numpy as np

from lambdapic_amd.fields import Fields2D, Fields3D
from lambdapic_amd.particles import ParticlesBase

PATTRS = ["x", "y", "z", "w", "ux", "uy", "uz", "inv_gamma"]
PEB = ["ex_part", "ey_part", "ez_part", "bx_part", "by_part", "bz_part"]


def fields2d_from(g, prefix, x0, y0):
    f = Fields2D(int(g["nx"]), int(g["ny"]), float(g["dx"]), float(g["dy"]), float(x0), float(y0),
                 int(g["ng"]))
    for a in f.attrs:
        if prefix + a in g:
            getattr(f, a)[...] = g[prefix + a]
    return f


def fields3d_from(g, prefix):
    f = Fields3D(int(g["nx"]), int(g["ny"]), int(g["nz"]), float(g["dx"]), float(g["dy"]),
                 float(g["dz"]), float(g["x0"]), float(g["y0"]), float(g["z0"]), int(g["ng"]))
    for a in f.attrs:
        if prefix + a in g:
            getattr(f, a)[...] = g[prefix + a]
    return f


def particles_from(g, prefix, names=None):
    names = names or [a for a in PATTRS if prefix + a in g]
    n = g[prefix + names[0]].size
    p = ParticlesBase(0, 0)
    p.initialize(n)
    for a in names:
        getattr(p, a)[:] = g[prefix + a]
    if prefix + "is_dead" in g:
        p.is_dead[:] = g[prefix + "is_dead"]
    if prefix + "_id" in g:
        p._id[:] = g[prefix + "_id"]
    return p


def assert_close(actual, expected, rtol, scale=None, what=""):
    """|a-e| <= rtol * scale, scale defaulting to max|e|; NaN patterns must coincide."""
    a, e = np.asarray(actual, dtype=np.float64), np.asarray(expected, dtype=np.float64)
    assert a.shape == e.shape, what
    na, ne = np.isnan(a), np.isnan(e)
    assert np.array_equal(na, ne), f"{what}: NaN pattern differs"
    if scale is None:
        scale = np.max(np.abs(e[~ne])) if (~ne).any() else 1.0
    if scale == 0:
        scale = 1.0
    err = np.max(np.abs(a[~ne] - e[~ne])) / scale if (~ne).any() else 0.0
    assert err <= rtol, f"{what}: err {err:.3e} > {rtol:.1e} (scale {scale:.3e})"
    return err
