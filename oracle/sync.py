"""oracle/sync.py -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's intra-rank patch synchronisation
(core/patch/sync_fields2d.c, core/patch/sync_particles_2d.c), operating on the host patch mirrors.
Checked against the reference's compiled extensions through tests/golden (G7).
"""
from __future__ import annotations

import numpy as np

# Boundary2D order (core/patch/sync_fields2d.c:19-29): the (x, y) side of each of the 8 neighbours
_SIDE = [(-1, 0), (1, 0), (0, -1), (0, 1), (-1, -1), (1, -1), (-1, 1), (1, 1)]
_OPP = [1, 0, 3, 2, 7, 6, 5, 4]  # sync_fields2d.c:31-40


def _rng_guard(side, n, ng):
    """index arrays (dst, src) along one axis for a guard *copy* (sync_fields2d.c:191-247):
    my guard on `side` <- neighbour's interior edge."""
    if side == 0:
        k = np.arange(n)
        return k, k
    if side < 0:   # my lower guard [-ng,0) <- neighbour [n-ng, n)
        return np.arange(-ng, 0), np.arange(n - ng, n)
    return np.arange(n, n + ng), np.arange(0, ng)   # my upper guard <- neighbour [0, ng)


def _rng_fold(side, n, ng):
    """(dst, src) along one axis for the current *fold* (sync_fields2d.c:84-144):
    my interior edge on `side` += neighbour's guard beyond its opposite side."""
    if side == 0:
        k = np.arange(n)
        return k, k
    if side < 0:   # dst [0,ng) += src upper guard [n, n+ng)
        return np.arange(0, ng), np.arange(n, n + ng)
    return np.arange(n - ng, n), np.arange(-ng, 0)  # dst [n-ng,n) += src lower guard


def sync_guard_fields_2d(fields_list, patches_list, attrs, npatches, nx, ny, ng):
    """restates core/patch/sync_fields2d.c:150-255"""
    for a in attrs:
        # the reference copies in place patch by patch; the copied regions (guards) are never a
        # source (sources are interior cells), so a sequential sweep is order independent
        for ip in range(npatches):
            nb = patches_list[ip].neighbor_ipatch
            dst = getattr(fields_list[ip], a)
            for b, (sx, sy) in enumerate(_SIDE):
                if nb[b] < 0:
                    continue
                src = getattr(fields_list[nb[b]], a)
                dx_, sx_ = _rng_guard(sx, nx, ng)
                dy_, sy_ = _rng_guard(sy, ny, ng)
                dst[np.ix_(dx_, dy_)] = src[np.ix_(sx_, sy_)]


def sync_currents_2d(fields_list, patches_list, npatches, nx, ny, ng):
    """restates core/patch/sync_fields2d.c:43-148 (fold neighbour guards into my interior edge and
    zero them).  The destination regions are interior cells and the sources guard cells, so the
    sweep order does not matter; sources are zeroed as they are consumed."""
    for a in ("jx", "jy", "jz", "rho"):
        for ip in range(npatches):
            nb = patches_list[ip].neighbor_ipatch
            dst = getattr(fields_list[ip], a)
            for b, (sx, sy) in enumerate(_SIDE):
                if nb[b] < 0:
                    continue
                src = getattr(fields_list[nb[b]], a)
                dx_, sx_ = _rng_fold(sx, nx, ng)
                dy_, sy_ = _rng_fold(sy, ny, ng)
                dst[np.ix_(dx_, dy_)] += src[np.ix_(sx_, sy_)]
                src[np.ix_(sx_, sy_)] = 0.0


def _classify(x, y, dead, xmin, xmax, ymin, ymax):
    """boundary id (0..7) of every live particle outside [xmin,xmax]x[ymin,ymax], -1 otherwise
    (core/patch/sync_particles_2d.c:37-84)."""
    out = np.full(x.size, -1, dtype=np.int64)
    lo_y, hi_y = y < ymin, y > ymax
    lo_x, hi_x = x < xmin, x > xmax
    mid_y = ~(lo_y | hi_y)
    out[lo_y & lo_x] = 4
    out[lo_y & hi_x] = 5
    out[lo_y & ~(lo_x | hi_x)] = 2
    out[hi_y & lo_x] = 6
    out[hi_y & hi_x] = 7
    out[hi_y & ~(lo_x | hi_x)] = 3
    out[mid_y & lo_x] = 0
    out[mid_y & hi_x] = 1
    out[dead] = -1
    return out


def sync_particles_2d(patches, ispec, dx, dy, attrs=None):
    """restates Patches.sync_particles for one species (core/patch/patch.py:705-742 driving
    core/patch/sync_particles_2d.c:204-518): count leavers per direction, grow the receiving
    arrays (+25 % slack), copy leavers into the receivers' dead slots in (boundary, index) order,
    periodic coordinate shift on wrap, then kill everything outside the owner's bounds."""
    npatches = patches.npatches
    parts = [p.particles[ispec] for p in patches]
    attrs = attrs or parts[0].attrs
    bounds = [(p.xmin - 0.5 * dx, p.xmax + 0.5 * dx, p.ymin - 0.5 * dy, p.ymax + 0.5 * dy)
              for p in patches]
    cls = [_classify(q.x, q.y, q.is_dead, *b) for q, b in zip(parts, bounds)]
    nout = np.array([[np.count_nonzero(c == b) for b in range(8)] for c in cls], dtype=np.int64)
    Lx = patches.xmax_global - patches.xmin_global
    Ly = patches.ymax_global - patches.ymin_global
    incoming = []
    for ip, p in enumerate(patches):
        rows = []
        for b in range(8):
            nb = p.neighbor_ipatch[b]
            if nb < 0:
                continue
            idx = np.nonzero(cls[nb] == _OPP[b])[0]
            if idx.size:
                rows.append(np.stack([getattr(parts[nb], a)[idx] for a in attrs], axis=1))
        incoming.append(np.concatenate(rows, axis=0) if rows else np.zeros((0, len(attrs))))
    npart_alive = np.zeros(npatches, dtype=np.int64)
    for ip, (q, buf) in enumerate(zip(parts, incoming)):
        nnew = buf.shape[0]
        ndead = int(q.is_dead.sum())
        npart_alive[ip] = q.npart - ndead + nnew
        if nnew - ndead > 0:
            q.extend(nnew - ndead + int(q.npart * 0.25))
    ix, iy = attrs.index("x"), attrs.index("y")
    for ip, (q, buf) in enumerate(zip(parts, incoming)):
        if buf.shape[0] == 0:
            continue
        xmin, xmax, ymin, ymax = bounds[ip]
        buf = buf.copy()
        cx, cy = buf[:, ix], buf[:, iy]
        if abs(xmin - patches.xmin_global) < dx:
            cx[buf[:, ix] > patches.xmax_global] -= Lx
        if abs(xmax - patches.xmax_global) < dx:
            cx[buf[:, ix] < patches.xmin_global] += Lx
        if abs(ymin - patches.ymin_global) < dy:
            cy[buf[:, iy] > patches.ymax_global] -= Ly
        if abs(ymax - patches.ymax_global) < dy:
            cy[buf[:, iy] < patches.ymin_global] += Ly
        slots = np.nonzero(q.is_dead)[0][: buf.shape[0]]
        for k, a in enumerate(attrs):
            getattr(q, a)[slots] = buf[:, k]
        q.is_dead[slots] = False
    for q, c, b in zip(parts, cls, bounds):
        # leavers were classified before the fill; freshly filled slots are inside by construction
        # except for particles the periodic shift could not bring back (none in valid runs), so
        # re-test everything like the reference does (sync_particles_2d.c:185-202)
        xmin, xmax, ymin, ymax = b
        with np.errstate(invalid="ignore"):
            outside = (~q.is_dead) & ((q.x < xmin) | (q.x > xmax) | (q.y < ymin) | (q.y > ymax))
        q.is_dead[outside] = True
        q.x[outside] = np.nan
        q.y[outside] = np.nan
    return npart_alive


# -------------------------------------------------------------------------------------------------
# 3-D patch lists (core/patch/sync_fields3d.c, core/patch/sync_particles_3d.c): 26 neighbours in Boundary3D order
# (core/patch/patch.py:37-69).  Checked against the reference's compiled extensions through tests/golden (G16, G17).
# -------------------------------------------------------------------------------------------------
def _boundary3(sx, sy, sz):
    """index in Boundary3D of the neighbour on side (sx, sy, sz); -1 for (0, 0, 0)"""
    nz = (sx != 0) + (sy != 0) + (sz != 0)
    if nz == 0:
        return -1
    if nz == 1:
        return int(sx > 0) if sx else (2 + int(sy > 0) if sy else 4 + int(sz > 0))
    if nz == 3:
        return 18 + 4 * int(sx > 0) + 2 * int(sy > 0) + int(sz > 0)
    if sz == 0:
        return 6 + 4 * int(sx > 0) + int(sy > 0)
    if sy == 0:
        return 8 + 4 * int(sx > 0) + int(sz > 0)
    return 14 + 2 * int(sy > 0) + int(sz > 0)


_SIDE3 = [None] * 26
for _sx in (-1, 0, 1):
    for _sy in (-1, 0, 1):
        for _sz in (-1, 0, 1):
            if _boundary3(_sx, _sy, _sz) >= 0:
                _SIDE3[_boundary3(_sx, _sy, _sz)] = (_sx, _sy, _sz)
_OPP3 = [_boundary3(-a, -b, -c) for a, b, c in _SIDE3]      # sync_fields3d.c:52-82


def sync_guard_fields_3d(fields_list, patches_list, attrs, npatches, nx, ny, nz, ng):
    """restates core/patch/sync_fields3d.c:350-612: my guard on side s <- the neighbour's interior edge"""
    n3 = (nx, ny, nz)
    for a in attrs:
        for ip in range(npatches):
            nb = patches_list[ip].neighbor_ipatch
            dst = getattr(fields_list[ip], a)
            for b, side in enumerate(_SIDE3):
                if nb[b] < 0:
                    continue
                src = getattr(fields_list[nb[b]], a)
                rng = [_rng_guard(s_, n, ng) for s_, n in zip(side, n3)]
                dst[np.ix_(*[r[0] for r in rng])] = src[np.ix_(*[r[1] for r in rng])]


def sync_currents_3d(fields_list, patches_list, npatches, nx, ny, nz, ng):
    """restates core/patch/sync_fields3d.c:84-348: my interior edge on side s += the neighbour's guard beyond its
    opposite side, which is zeroed; boundaries in Boundary3D order like the reference's sweep"""
    n3 = (nx, ny, nz)
    for a in ("jx", "jy", "jz", "rho"):
        for ip in range(npatches):
            nb = patches_list[ip].neighbor_ipatch
            dst = getattr(fields_list[ip], a)
            for b, side in enumerate(_SIDE3):
                if nb[b] < 0:
                    continue
                src = getattr(fields_list[nb[b]], a)
                rng = [_rng_fold(s_, n, ng) for s_, n in zip(side, n3)]
                si = np.ix_(*[r[1] for r in rng])
                dst[np.ix_(*[r[0] for r in rng])] += src[si]
                src[si] = 0.0


def _classify3(q, bounds):
    """Boundary3D id of every live particle outside the bounds, -1 otherwise (sync_particles_3d.c:78-192)"""
    side = []
    for a, (lo, hi) in zip("xyz", bounds):
        v = getattr(q, a)
        with np.errstate(invalid="ignore"):
            side.append(np.where(v < lo, -1, np.where(v > hi, 1, 0)))
    out = np.array([_boundary3(int(a), int(b), int(c)) for a, b, c in zip(*side)], dtype=np.int64) \
        if q.npart else np.zeros(0, dtype=np.int64)
    out[q.is_dead] = -1
    return out


def sync_particles_3d(patches, ispec, d, attrs=None):
    """restates Patches.sync_particles for one species in 3-D (core/patch/patch.py:739-763 driving
    core/patch/sync_particles_3d.c:365-700): as sync_particles_2d with 26 boundaries; the 3-D
    mark_out_of_bound_as_dead (:324-345) also blanks the positions of slots that are dead already"""
    npatches = patches.npatches
    parts = [p.particles[ispec] for p in patches]
    attrs = attrs or parts[0].attrs
    bounds = [[(getattr(p, ax + "min") - 0.5 * dd, getattr(p, ax + "max") + 0.5 * dd) for ax, dd in zip("xyz", d)]
              for p in patches]
    cls = [_classify3(q, b) for q, b in zip(parts, bounds)]
    gmin = [getattr(patches, ax + "min_global") for ax in "xyz"]
    gmax = [getattr(patches, ax + "max_global") for ax in "xyz"]
    incoming = []
    for ip, p in enumerate(patches):
        rows = []
        for b in range(26):
            nb = p.neighbor_ipatch[b]
            if nb < 0:
                continue
            idx = np.nonzero(cls[nb] == _OPP3[b])[0]
            if idx.size:
                rows.append(np.stack([getattr(parts[nb], a)[idx] for a in attrs], axis=1))
        incoming.append(np.concatenate(rows, axis=0) if rows else np.zeros((0, len(attrs))))
    npart_alive = np.zeros(npatches, dtype=np.int64)
    for ip, (q, buf) in enumerate(zip(parts, incoming)):
        nnew, ndead = buf.shape[0], int(q.is_dead.sum())
        npart_alive[ip] = q.npart - ndead + nnew
        if nnew - ndead > 0:
            q.extend(nnew - ndead + int(q.npart * 0.25))
    for ip, (q, buf) in enumerate(zip(parts, incoming)):
        if buf.shape[0] == 0:
            continue
        buf = buf.copy()
        for k, ax in enumerate("xyz"):
            col = buf[:, attrs.index(ax)]
            orig = col.copy()
            (lo, hi), L = bounds[ip][k], gmax[k] - gmin[k]
            if abs(lo - gmin[k]) < d[k]:
                col[orig > gmax[k]] -= L
            if abs(hi - gmax[k]) < d[k]:
                col[orig < gmin[k]] += L
        slots = np.nonzero(q.is_dead)[0][: buf.shape[0]]
        for k, a in enumerate(attrs):
            getattr(q, a)[slots] = buf[:, k]
        q.is_dead[slots] = False
    for q, b in zip(parts, bounds):
        with np.errstate(invalid="ignore"):
            outside = q.is_dead.copy()
            for ax, (lo, hi) in zip("xyz", b):
                v = getattr(q, ax)
                outside |= (v < lo) | (v > hi)
        q.is_dead[outside] = True
        for ax in "xyz":
            getattr(q, ax)[outside] = np.nan
    return npart_alive


# -------------------------------------------------------------------------------------------------
# one patch that is its own neighbour on every periodic axis (any dimension): what the sync
# functions above reduce to for npatch = 1.  Used by the 3-D engine test; pinned to the reference
# through the 2-D functions above (tests/test_oracle_golden.py::test_periodic_single_patch_twins).
# -------------------------------------------------------------------------------------------------
def periodic_guard_fill(f, attrs):
    ng = f.n_guard
    for a in attrs:
        arr = getattr(f, a)
        conv = np.roll(arr, ng, axis=tuple(range(arr.ndim)))
        inner = conv[tuple(slice(ng, -ng) for _ in range(arr.ndim))]
        arr[...] = np.roll(np.pad(inner, ng, mode="wrap"), -ng, axis=tuple(range(arr.ndim)))


def periodic_current_fold(f):
    ng = f.n_guard
    for a in ("jx", "jy", "jz", "rho"):
        arr = getattr(f, a)
        nd = arr.ndim
        conv = np.roll(arr, ng, axis=tuple(range(nd)))
        n = [s - 2 * ng for s in conv.shape]
        idx = np.ix_(*[(np.arange(s) - ng) % m for s, m in zip(conv.shape, n)])
        out = np.zeros(n)
        np.add.at(out, idx, conv)
        conv[...] = 0.0
        conv[tuple(slice(ng, -ng) for _ in range(nd))] = out
        arr[...] = np.roll(conv, -ng, axis=tuple(range(nd)))


def periodic_fold_positions(p, lo, hi, axes=("x", "y", "z")):
    """what Patches.sync_particles does to a particle that left a patch which is its own periodic neighbour
    (`core/patch/sync_particles_2d.c:168-182`, `sync_particles_3d.c`: the copy that re-enters is shifted by the
    box length): fold the coordinates of the live particles into [lo, hi] per axis.  Pinned by g7 (2-D) and
    g14 (3-D, bit exact)."""
    live = ~p.is_dead
    for a, l, h in zip(axes, lo, hi):
        v = getattr(p, a)
        L = h - l
        v[live & (v > h)] -= L
        v[live & (v < l)] += L

