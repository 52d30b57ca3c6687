"""Host-side mirror of the λPIC particle container (layout contract only).

Attribute contract of the reference's ``ParticlesBase`` (`core/particles.py:8-217`):
fifteen ``float64[npart]`` arrays named in ``attrs`` (``z`` exists in 2-D too), a
``bool[npart]`` ``is_dead`` flag, ``npart`` (slots including dead ones).  ``_id`` holds a
``uint64`` bit pattern viewed as ``float64`` (14 bit rank | 18 bit patch | 32 bit serial,
`core/particles.py:50-51,91-116`).  Freshly extended slots are NaN with ``w = 0`` and
``is_dead = True`` (`core/particles.py:141-168`).
"""
from __future__ import annotations

import numpy as np

PARTICLE_ATTRS = (
    "x", "y", "z", "w", "ux", "uy", "uz", "inv_gamma",
    "ex_part", "ey_part", "ez_part", "bx_part", "by_part", "bz_part", "_id",
)


class ParticlesBase:
    def __init__(self, ipatch: int | None = None, rank: int | None = None):
        self.attrs = list(PARTICLE_ATTRS)
        self.rank = 0 if rank is None else int(rank)
        self.ipatch = 0 if ipatch is None else int(ipatch)
        if not (0 <= self.rank < 1 << 14 and 0 <= self.ipatch < 1 << 18):
            raise AssertionError("rank must be < 2^14 and ipatch < 2^18")
        self._id_prefix = np.uint64((self.rank << 50) | (self.ipatch << 32))
        self._npart_created = 0
        self._npart_alive = 0
        self.extended = False
        self.npart = 0

    def _new_ids(self, count: int) -> np.ndarray:
        start = self._npart_created
        if start + count > 1 << 32:
            raise AssertionError("more than 2^32 particles created in one patch")
        serial = np.arange(start, start + count, dtype=np.uint64)
        self._npart_created += count
        return (serial | self._id_prefix).view(np.float64)

    def initialize(self, npart: int) -> None:
        assert npart >= 0
        self.npart = int(npart)
        for name in self.attrs:
            setattr(self, name, np.zeros(self.npart, dtype=np.float64))
        self.inv_gamma[:] = 1.0
        self.is_dead = np.zeros(self.npart, dtype=np.bool_)
        self._id[:] = self._new_ids(self.npart)

    def extend(self, n: int) -> None:
        if n <= 0:
            return
        total = self.npart + int(n)
        for name in self.attrs:
            grown = np.full(total, np.nan, dtype=np.float64)
            grown[: self.npart] = getattr(self, name)[: self.npart]
            setattr(self, name, grown)
        self.w[self.npart:] = 0.0
        self._id[self.npart:] = self._new_ids(int(n))
        dead = np.ones(total, dtype=np.bool_)
        dead[: self.npart] = self.is_dead[: self.npart]
        self.is_dead = dead
        self.npart = total
        self.extended = True

    def prune(self, extra_buff: float = 0.1):
        n_alive = int(self.is_alive.sum())
        keep = int(n_alive * (1 + extra_buff))
        if keep >= self.npart:
            return None
        order = np.argsort(self.is_dead, kind="stable")
        for name in self.attrs:
            setattr(self, name, getattr(self, name)[order][:keep].copy())
        self.is_dead = self.is_dead[order][:keep].copy()
        self.npart = keep
        self.extended = True
        return order

    @property
    def id(self) -> np.ndarray:
        return self._id.view(np.uint64)

    @property
    def is_alive(self) -> np.ndarray:
        return ~self.is_dead
