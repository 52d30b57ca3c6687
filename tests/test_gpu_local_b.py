"""B on the x guard planes without a message (``LPA_STEP_B_EXT_LO / _HI``, include/lambdapic_amd.h): a slab with neighbours
lets its B half steps advance the ng low and ng - 1 high x guard planes themselves.  A B update reads E at its node and one
node up; with current E guard planes the values are the ones the neighbour computes for its interior -- BIT FOR BIT.  The
check uses a box that is periodic along x (its own neighbour): after a guard sync the guard planes hold the periodic images;
one B half step through ``lpa_step`` with the flags set (and x NOT wrapped) must leave every advanced guard plane equal to the
image of the freshly updated interior, the B psi rows of the y / z CPML layers included.  The mirrored-slab tests
(test_gpu_native_slab.py) run whole steps on top of this; the reference has no counterpart (it exchanges B:
simulation/simulation.py:954-960, 1103-1108)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CL = 299792458.0


def _engine(dim, cpml):
    from lambdapic_amd.engine import PicEngine2D
    from lambdapic_amd.engine3d import PicEngine3D
    lam = 0.8e-6
    if dim == 2:
        bc = {"xmin": "periodic", "xmax": "periodic", "ymin": "pml" if cpml else "periodic", "ymax": "pml" if cpml else "periodic"}
        eng = PicEngine2D(24, 40, lam / 20, lam / 16, device="cuda:0", boundary_conditions=bc, cpml_thickness=6)
        dt = 0.95 / (CL * np.sqrt(eng.dx ** -2 + eng.dy ** -2))
    else:
        side = "pml" if cpml else "periodic"
        bc = {"xmin": "periodic", "xmax": "periodic", "ymin": side, "ymax": side, "zmin": side, "zmax": side}
        eng = PicEngine3D(16, 20, 24, lam / 20, lam / 16, lam / 12, 3, boundary_conditions=bc, cpml_thickness=6)
        dt = 0.95 / (CL * np.sqrt(sum(v ** -2 for v in eng.d)))
    return eng, dt


def _view(eng, a):
    return eng.grid.view(a) if eng.dim == 2 else eng.view(a)


def _fill(eng, seed):
    """random E, B everywhere, then the guard sync (periodic images along x, and along y / z where they are periodic);
    random psi in the y / z layers with image rows in their x guard rows"""
    from lambdapic_amd.engine import psi_rows
    gen = torch.Generator(device="cuda:0").manual_seed(seed)
    for a in ("ex", "ey", "ez", "bx", "by", "bz"):
        v = _view(eng, a)
        v.copy_(torch.randn(v.shape, dtype=torch.float64, device="cuda:0", generator=gen))
    if eng.dim == 2:
        eng.sync_guard_fields(("ex", "ey", "ez", "bx", "by", "bz"))
    else:
        eng.sync_guard_fields(3)
    ng, nx = eng.ng, eng.n_x_local()
    # whole x planes, open y / z guards included, as a slab-to-slab exchange delivers them (the wrap above leaves the
    # x-guard / open-guard corners alone)
    for a in ("ex", "ey", "ez", "bx", "by", "bz"):
        v = _view(eng, a)
        v[:ng] = v[nx:nx + ng]
        v[ng + nx:] = v[ng:2 * ng]
    if eng.pml is not None:
        for ly in eng.pml.layers:
            if ly["axis"] == 0:
                continue
            assert ly["xpad"] == ng
            for k in ("psi_a", "psi_b"):
                v = psi_rows(ly, k, guards=True)
                v[ng:ng + nx] = torch.randn((nx, v.shape[1]), dtype=torch.float64, device="cuda:0", generator=gen)
                v[:ng] = v[nx:nx + ng]
                v[ng + nx:] = v[ng:2 * ng]


def _b_half_step(eng, dt, flags, wrap_x):
    """one LPA_STAGE_B1 through lpa_step with a hand-made descriptor (no slab section: nothing is exchanged)"""
    from lambdapic_amd import _lib
    d = _lib.lpa_step_desc()
    d.grid = eng._grid_struct()
    d.dim, d.dt, d.eps0 = eng.dim, dt, eng.eps0
    d.local_axes = eng.local_axes if wrap_x else (eng.local_axes & ~1)
    d.flags = flags
    if eng.pml is not None:
        for fld, arr in ((True, d.e_axes), (False, d.b_axes)):
            for a, ax in enumerate(eng._cpml_axes(fld, 0.5 * dt)):
                arr[a] = C.pointer(ax)
    d.nspecies = 0
    _lib.check(eng.L.lpa_step(C.byref(d), _lib.LPA_STAGE_B1, _lib.LPA_STAGE_B1, torch.cuda.current_stream().cuda_stream), "lpa_step")
    torch.cuda.synchronize()


@pytest.mark.parametrize("cpml", [False, True])
@pytest.mark.parametrize("dim", [2, 3])
def test_b_on_the_x_guard_planes_is_the_neighbours_b(dim, cpml):
    from lambdapic_amd import _lib
    from lambdapic_amd.engine import psi_rows
    eng, dt = _engine(dim, cpml)
    assert (eng.pml is not None) == cpml and (eng.local_axes & 1)
    ng, nx = eng.ng, eng.n_x_local()
    # (a) the plain periodic sweep (x wrapped by the sweep itself): the reference result
    _fill(eng, 7)
    _b_half_step(eng, dt, 0, wrap_x=True)
    want = {a: _view(eng, a).clone() for a in ("bx", "by", "bz")}
    want_psi = [psi_rows(ly, k, guards=True).clone() for ly in (eng.pml.layers if cpml else []) if ly["axis"] and not ly["e"]
                for k in ("psi_a", "psi_b")]
    # (b) the same state, x treated as split: the sweep advances the guard planes from the E it finds there
    _fill(eng, 7)
    _b_half_step(eng, dt, _lib.LPA_STEP_B_EXT_LO | _lib.LPA_STEP_B_EXT_HI, wrap_x=False)
    for a in ("bx", "by", "bz"):
        got = _view(eng, a)
        assert torch.equal(got[ng:ng + nx], want[a][ng:ng + nx]), a                    # interior: unchanged arithmetic
        assert torch.equal(got[:ng], want[a][:ng]), a                                  # all ng low guard planes
        assert torch.equal(got[ng + nx:ng + nx + ng - 1], want[a][ng + nx:ng + nx + ng - 1]), a     # ng - 1 high ones
        # (the outermost high plane would need E one node beyond the guard: it keeps its old value)
        assert not torch.equal(got[ng + nx + ng - 1], want[a][ng + nx + ng - 1])
    if cpml:
        got_psi = [psi_rows(ly, k, guards=True) for ly in eng.pml.layers if ly["axis"] and not ly["e"] for k in ("psi_a", "psi_b")]
        assert len(got_psi) == (4 if dim == 2 else 8)
        for g_, w_ in zip(got_psi, want_psi):
            assert torch.equal(g_[ng:ng + nx], w_[ng:ng + nx])
            # the guard rows follow their interior images: rows [0, ng) <-> [nx, nx + ng), [ng + nx, ...) <-> [ng, ...)
            assert torch.equal(g_[:ng], g_[nx:nx + ng])
            assert torch.equal(g_[ng + nx:ng + nx + ng - 1], g_[ng:2 * ng - 1])
            assert g_.abs().max() > 0


def test_one_face_only():
    """a chain end: only the face with a neighbour is advanced"""
    from lambdapic_amd import _lib
    eng, dt = _engine(2, False)
    ng, nx = eng.ng, eng.n_x_local()
    _fill(eng, 3)
    before = {a: _view(eng, a).clone() for a in ("bx", "by", "bz")}
    _b_half_step(eng, dt, _lib.LPA_STEP_B_EXT_HI, wrap_x=False)
    for a in ("by", "bz"):       # (bx has no x derivative: dEz/dy only -- it changes wherever Ez varies along y)
        got = _view(eng, a)
        assert torch.equal(got[:ng], before[a][:ng])                                   # low guard untouched
        assert not torch.equal(got[ng + nx:ng + nx + 2], before[a][ng + nx:ng + nx + 2])


def test_the_flags_need_a_split_x():
    from lambdapic_amd import _lib
    eng, dt = _engine(2, False)
    _fill(eng, 1)
    with pytest.raises(_lib.LpaError):
        _b_half_step(eng, dt, _lib.LPA_STEP_B_EXT_LO, wrap_x=True)
