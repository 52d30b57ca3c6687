"""Free-slot reuse of the slab migration (`lpa_free_slots`, `lpa_migrate_pack_edges_x`, `lpa_migrate_unpack_tiled`):
one process plays rank 0 of a periodic 2-slab ring whose neighbour is a translated copy of itself (the wire is a
device copy), so every kernel of the multi-rank path runs.  With the stacks most arrivals must land in the slots
the leavers of their tile freed -- inside the tile-ordered range -- and the physics must not notice."""
import numpy as np
import pytest
import torch

from lambdapic_amd import constants
from lambdapic_amd._lib import LPA_MIG_NATTR
from lambdapic_amd.dist import SlabComm
from lambdapic_amd.engine import PicEngine2D

pytestmark = pytest.mark.gpu
C = 299792458.0


class _Mirror(SlabComm):
    def __init__(self, slab_width, cap):
        super().__init__(None, periodic=True, single=True)
        self.size, self.rank, self.left, self.right = 2, 0, 1, 1
        self.shift = float(slab_width)

    def exchange(self, send_lo, send_hi, recv_lo, recv_hi, wait=True):
        recv_lo.copy_(send_hi)
        recv_hi.copy_(send_lo)
        return []

    def arrival_shift(self, box_length):        # the neighbour is this slab's copy one slab further
        return -self.shift, self.shift

    def exchange_many(self, sets):
        for s_ in sets:
            self.exchange(*s_)
        return []


def _run(reuse, nsteps=18, cap=8192):
    nx, ny, ppc = 128, 64, 8
    dx = dy = 0.8e-6 / 20
    dt = 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2))
    eng = PicEngine2D(nx, ny, dx, dy, device="cuda:0", comm=_Mirror(nx * dx, cap), sort_interval=9,
                      block_particles=2048, migrate_capacity=cap)
    eng.reuse_slots = reuse
    eng.overflow_sort_fraction = 0      # the fixed sort schedule: the test watches the arrival area fill up over an interval
    n = nx * ny * ppc
    q, m = -constants.E_CHARGE, constants.M_E
    eng.add_species(q, m, capacity=2 * n)
    s = eng.species[0].cset
    g = torch.Generator(device="cuda:0").manual_seed(3)
    cell = torch.arange(n, device="cuda:0") // ppc
    r = lambda: torch.rand(n, device="cuda:0", dtype=torch.float64, generator=g)
    s.arr("x")[:n] = ((cell // ny).double() + r() - 0.5) * dx
    s.arr("y")[:n] = ((cell % ny).double() + r() - 0.5) * dy
    for a in ("ux", "uy", "uz"):
        s.arr(a)[:n] = torch.randn(n, device="cuda:0", dtype=torch.float64, generator=g) * 0.3   # hot: many leavers
    s.arr("inv_gamma")[:n] = 1 / torch.sqrt(1 + s.arr("ux")[:n] ** 2 + s.arr("uy")[:n] ** 2 + s.arr("uz")[:n] ** 2)
    s.arr("w")[:n] = 1.7e27 * dx * dy / ppc
    s.id[:n] = torch.arange(n, device="cuda:0")
    eng.species[0].n = n
    in_area = []
    for it in range(nsteps):
        eng.step(dt)
        ws = eng._sort_ws(eng.species[0])
        in_area.append(int(ws["counters"][1].item()))       # arrivals parked in the arrival area since the sort
    return eng, n, in_area


def test_arrivals_take_the_slots_their_tile_freed():
    a, n, area_on = _run(True)
    b, _, area_off = _run(False)
    da, db = a.diagnostics(), b.diagnostics()
    assert da["nalive"][0] == n == db["nalive"][0]                       # nobody lost, nobody doubled
    assert da["charge"] == pytest.approx(db["charge"], rel=1e-12)
    assert da["field_energy"] == pytest.approx(db["field_energy"], rel=1e-10)
    assert da["kinetic"][0] == pytest.approx(db["kinetic"][0], rel=1e-12)
    for name in ("ex", "ey", "bz", "rho"):
        va, vb = a.grid.view(name), b.grid.view(name)
        assert (va - vb).abs().max().item() <= 1e-9 * vb.abs().max().item(), name
    # without the stacks every arrival waits in the area; with them most find a slot in their tile
    peak_on, peak_off = max(area_on), max(area_off)
    assert peak_off > 300, (area_on, area_off)
    assert peak_on < 0.5 * peak_off, (area_on, area_off)
    # ids are still unique: a reused slot holds exactly one particle
    sp = a.species[0]
    x = sp.cset.arr("x")[: sp.n]
    ids = sp.cset.id[: sp.n][~torch.isnan(x)]
    assert ids.numel() == n and torch.unique(ids).numel() == n


def test_send_side_overflow_is_reported():
    """more leavers per step than ``migrate_capacity``: the surplus stays outside the slab (it is not lost), and the
    engine says so at its next sort instead of letting it deposit through the torus wrap on the wrong side"""
    from lambdapic_amd._lib import LpaError
    with pytest.raises(LpaError, match="migration message overflow"):
        _run(True, nsteps=12, cap=16)
