"""Host-side mirror of λPIC's patch containers (`core/patch/patch.py:70-386,388-907`).

Only the attribute contract the hot path and its callers read is mirrored: geometry
(``x0 y0 nx ny dx dy``), the 8-entry neighbour tables in the reference's ``Boundary2D``
order (`core/patch/patch.py:24-35`: xmin, xmax, ymin, ymax, xminymin, xmaxymin, xminymax,
xmaxymax), ``fields``, ``particles[ispec]`` and the particle ownership bounds
``xmin/xmax/ymin/ymax`` (`core/patch/patch.py:105-148`).  Patches are *views* at the
boundary of the device engine: the GPU holds one contiguous slab per rank.
"""
from __future__ import annotations

from enum import IntEnum

import numpy as np


class Boundary2D(IntEnum):
    XMIN = 0
    XMAX = 1
    YMIN = 2
    YMAX = 3
    XMINYMIN = 4
    XMAXYMIN = 5
    XMINYMAX = 6
    XMAXYMAX = 7


# (di, dj) of each Boundary2D member and its opposite (`core/patch/sync_fields2d.c:31-40`)
OFFSET_2D = {
    Boundary2D.XMIN: (-1, 0), Boundary2D.XMAX: (1, 0),
    Boundary2D.YMIN: (0, -1), Boundary2D.YMAX: (0, 1),
    Boundary2D.XMINYMIN: (-1, -1), Boundary2D.XMAXYMIN: (1, -1),
    Boundary2D.XMINYMAX: (-1, 1), Boundary2D.XMAXYMAX: (1, 1),
}
OPPOSITE_2D = {b: next(o for o in Boundary2D
                       if OFFSET_2D[o] == (-OFFSET_2D[b][0], -OFFSET_2D[b][1]))
               for b in Boundary2D}


class Boundary3D(IntEnum):
    """`core/patch/patch.py:37-69` (must agree with sync_particles_3d.c / sync_fields3d.c): faces, edges, vertices"""
    XMIN = 0
    XMAX = 1
    YMIN = 2
    YMAX = 3
    ZMIN = 4
    ZMAX = 5
    XMINYMIN = 6
    XMINYMAX = 7
    XMINZMIN = 8
    XMINZMAX = 9
    XMAXYMIN = 10
    XMAXYMAX = 11
    XMAXZMIN = 12
    XMAXZMAX = 13
    YMINZMIN = 14
    YMINZMAX = 15
    YMAXZMIN = 16
    YMAXZMAX = 17
    XMINYMINZMIN = 18
    XMINYMINZMAX = 19
    XMINYMAXZMIN = 20
    XMINYMAXZMAX = 21
    XMAXYMINZMIN = 22
    XMAXYMINZMAX = 23
    XMAXYMAXZMIN = 24
    XMAXYMAXZMAX = 25


def _offset_3d(b: Boundary3D):
    name = b.name
    return tuple((-1 if ax + "MIN" in name else (1 if ax + "MAX" in name else 0)) for ax in "XYZ")


# (di, dj, dk) of each Boundary3D member and its opposite (`core/patch/sync_fields3d.c:52-82`)
OFFSET_3D = {b: _offset_3d(b) for b in Boundary3D}
OPPOSITE_3D = {b: next(o for o in Boundary3D if OFFSET_3D[o] == tuple(-v for v in OFFSET_3D[b])) for b in Boundary3D}


class Patch3D:
    """attribute contract of the reference's ``Patch3D`` (`core/patch/patch.py:298-386`) the extensions read"""

    def __init__(self, rank, index, ipatch, origin, n, d):
        self.rank, self.index = rank, index
        self.ipatch_x, self.ipatch_y, self.ipatch_z = ipatch
        self.x0, self.y0, self.z0 = (float(v) for v in origin)
        self.nx, self.ny, self.nz = (int(v) for v in n)
        self.dx, self.dy, self.dz = (float(v) for v in d)
        self.neighbor_index = np.full(len(Boundary3D), -1, dtype=np.int64)
        self.neighbor_rank = np.full(len(Boundary3D), -1, dtype=np.int64)
        self.neighbor_ipatch = np.full(len(Boundary3D), -1, dtype=np.int64)
        self.pml_boundary = []
        self.particles = []
        self.fields = None

    xmin = property(lambda self: self.x0)
    ymin = property(lambda self: self.y0)
    zmin = property(lambda self: self.z0)
    xmax = property(lambda self: self.x0 + (self.nx - 1) * self.dx)
    ymax = property(lambda self: self.y0 + (self.ny - 1) * self.dy)
    zmax = property(lambda self: self.z0 + (self.nz - 1) * self.dz)

    def set_fields(self, fields):
        self.fields = fields

    def add_particles(self, particles):
        self.particles.append(particles)


class Patch2D:
    def __init__(self, rank, index, ipatch_x, ipatch_y, x0, y0, nx, ny, dx, dy):
        self.rank = rank
        self.index = index
        self.ipatch_x, self.ipatch_y = ipatch_x, ipatch_y
        self.x0, self.y0 = float(x0), float(y0)
        self.nx, self.ny = int(nx), int(ny)
        self.dx, self.dy = float(dx), float(dy)
        self.xaxis = np.arange(self.nx) * self.dx + self.x0
        self.yaxis = np.arange(self.ny) * self.dy + self.y0
        self.neighbor_index = np.full(len(Boundary2D), -1, dtype=np.int64)
        self.neighbor_rank = np.full(len(Boundary2D), -1, dtype=np.int64)
        self.neighbor_ipatch = np.full(len(Boundary2D), -1, dtype=np.int64)
        self.pml_boundary = []
        self.particles = []
        self.fields = None

    # particle ownership bounds; PML shrink (`patch.py:105-148`) arrives with the CPML row
    @property
    def xmin(self):
        return self.x0

    @property
    def xmax(self):
        return self.x0 + (self.nx - 1) * self.dx

    @property
    def ymin(self):
        return self.y0

    @property
    def ymax(self):
        return self.y0 + (self.ny - 1) * self.dy

    def set_fields(self, fields):
        self.fields = fields

    def add_particles(self, particles):
        self.particles.append(particles)


class Patches:
    """List-like container of the patches of one rank (`core/patch/patch.py:388-445`)."""

    def __init__(self, dimension: int = 2):
        self.dimension = dimension
        self.patches = []
        self.species = []
        self.indices = []
        self.xmin_global = self.xmax_global = None
        self.ymin_global = self.ymax_global = None

    def __getitem__(self, i):
        return self.patches[i]

    def __len__(self):
        return len(self.patches)

    def __iter__(self):
        return iter(self.patches)

    def append(self, patch):
        self.patches.append(patch)
        self.indices.append(patch.index)

    @property
    def npatches(self):
        return len(self.patches)

    @property
    def nx(self):
        return self.patches[0].fields.nx

    @property
    def ny(self):
        return self.patches[0].fields.ny

    @property
    def dx(self):
        return self.patches[0].fields.dx

    @property
    def dy(self):
        return self.patches[0].fields.dy

    @property
    def n_guard(self):
        return self.patches[0].fields.n_guard

    def init_rect_neighbor_index_2d(self, npatch_x, npatch_y, *, boundary_conditions):
        """neighbour tables of a rectangular patch grid; periodic faces wrap, others stay -1
        (`core/patch/patch.py:446-507`).  Single-rank: neighbor_ipatch == neighbor_index."""
        where = {(p.ipatch_x, p.ipatch_y): k for k, p in enumerate(self.patches)}
        for p in self.patches:
            p.neighbor_index.fill(-1)        # a rebuild (patches relabelled by a moving window) starts clean
            p.neighbor_ipatch.fill(-1)
            p.neighbor_rank.fill(-1)
            for b, (di, dj) in OFFSET_2D.items():
                ni, nj = p.ipatch_x + di, p.ipatch_y + dj
                if ni < 0:
                    if boundary_conditions["xmin"] != "periodic":
                        continue
                    ni = npatch_x - 1
                elif ni >= npatch_x:
                    if boundary_conditions["xmax"] != "periodic":
                        continue
                    ni = 0
                if nj < 0:
                    if boundary_conditions["ymin"] != "periodic":
                        continue
                    nj = npatch_y - 1
                elif nj >= npatch_y:
                    if boundary_conditions["ymax"] != "periodic":
                        continue
                    nj = 0
                k = where[(ni, nj)]
                p.neighbor_index[b] = self.patches[k].index
                p.neighbor_ipatch[b] = k
                p.neighbor_rank[b] = self.patches[k].rank if self.patches[k].rank is not None else 0


def init_rect_neighbor_index_3d(patches, npatch, boundary_conditions):
    """neighbour tables of a rectangular 3-D patch grid (`core/patch/patch.py:509-590`): periodic faces wrap, the others
    stay -1.  Single rank: neighbor_ipatch == neighbor_index."""
    where = {(p.ipatch_x, p.ipatch_y, p.ipatch_z): k for k, p in enumerate(patches)}
    for p in patches:
        p.neighbor_index.fill(-1)
        p.neighbor_ipatch.fill(-1)
        for b, off in OFFSET_3D.items():
            pos, ok = [], True
            for ax, i, o, n in zip("xyz", (p.ipatch_x, p.ipatch_y, p.ipatch_z), off, npatch):
                j = i + o
                if j < 0:
                    if boundary_conditions[ax + "min"] != "periodic":
                        ok = False
                        break
                    j = n - 1
                elif j >= n:
                    if boundary_conditions[ax + "max"] != "periodic":
                        ok = False
                        break
                    j = 0
                pos.append(j)
            if not ok:
                continue
            k = where[tuple(pos)]
            p.neighbor_index[b] = patches[k].index
            p.neighbor_ipatch[b] = k
            p.neighbor_rank[b] = 0


def make_patches_3d(n, d, npatch, n_guard=3, boundary_conditions=None, nspecies=1):
    """the rectangular 3-D patch set of the reference's ``Simulation3D.create_patches`` (`simulation/simulation.py:1143-1292`
    for the in-scope parts): ``n`` = global cells, ``npatch`` = patches per axis, x fastest in the patch index"""
    from .fields import Fields3D
    from .particles import ParticlesBase

    bc = boundary_conditions or {k: "periodic" for k in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")}
    assert all(a % b == 0 for a, b in zip(n, npatch))
    npp = [a // b for a, b in zip(n, npatch)]
    patches = Patches(dimension=3)
    for k in range(npatch[2]):
        for j in range(npatch[1]):
            for i in range(npatch[0]):
                org = (i * npp[0] * d[0], j * npp[1] * d[1], k * npp[2] * d[2])
                p = Patch3D(0, len(patches.patches), (i, j, k), org, npp, d)
                p.set_fields(Fields3D(*npp, *d, *org, n_guard))
                for _ in range(nspecies):
                    p.add_particles(ParticlesBase(ipatch=p.index, rank=0))
                patches.append(p)
    init_rect_neighbor_index_3d(patches.patches, npatch, bc)
    for a, ax in enumerate("xyz"):
        setattr(patches, ax + "min_global", -d[a] / 2)
        setattr(patches, ax + "max_global", n[a] * d[a] - d[a] / 2)
    return patches


def make_patches_2d(nx, ny, dx, dy, npatch_x, npatch_y, n_guard=3, boundary_conditions=None,
                    nspecies=1):
    """Build the rectangular patch set the reference's ``Simulation.create_patches`` +
    ``_init_fields`` produce (`simulation/simulation.py:432-448,467-502`)."""
    from .fields import Fields2D
    from .particles import ParticlesBase

    bc = boundary_conditions or {k: "periodic" for k in ("xmin", "xmax", "ymin", "ymax")}
    assert nx % npatch_x == 0 and ny % npatch_y == 0
    nxp, nyp = nx // npatch_x, ny // npatch_y
    Lx, Ly = nx * dx, ny * dy
    patches = Patches(dimension=2)
    for j in range(npatch_y):
        for i in range(npatch_x):
            p = Patch2D(rank=0, index=i + j * npatch_x, ipatch_x=i, ipatch_y=j,
                        x0=i * Lx / npatch_x, y0=j * Ly / npatch_y, nx=nxp, ny=nyp, dx=dx, dy=dy)
            p.set_fields(Fields2D(nxp, nyp, dx, dy, p.x0, p.y0, n_guard))
            for _ in range(nspecies):
                p.add_particles(ParticlesBase(ipatch=p.index, rank=0))
            patches.append(p)
    patches.init_rect_neighbor_index_2d(npatch_x, npatch_y, boundary_conditions=bc)
    patches.xmin_global, patches.xmax_global = -dx / 2, Lx - dx / 2
    patches.ymin_global, patches.ymax_global = -dy / 2, Ly - dy / 2
    return patches
