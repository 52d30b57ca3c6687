"""Host-side laser profiles against source rows recorded from the reference's laser classes
(tests/golden/g11_laser_profiles.npz, written by gen_golden.py:g11_laser_profiles)."""
import json
import types

import numpy as np
import pytest

from lambdapic_amd import laser as L


def _y(g, las):
    ny = int(g["ny"])
    return g["yaxis"][:ny] - float(g["dy"]) / 2 - (las.y0 or float(g["Ly"]) / 2)


def test_profiles_match_reference_rows(golden):
    g = golden("g11_laser_profiles")
    cases = json.loads(str(g["cases"]))
    ny = int(g["ny"])
    lasers = {k: getattr(L, cls)(**kw) for k, (cls, kw) in cases.items()}
    for name, las in lasers.items():
        for it, tm in enumerate(g["times"]):
            sim = types.SimpleNamespace(time=float(tm), dx=float(g["dx"]), cpml_thickness=int(g["thickness"]))
            ey, ez = las.source_fields(sim, _y(g, las))
            if g[f"{name}_t{it}_ey"].size == 0:          # the reference returned (None, None)
                assert ey is None and ez is None
                continue
            for got, key in ((ey, "ey"), (ez, "ez")):
                want = g[f"{name}_t{it}_{key}"][:ny]
                scale = max(np.abs(want).max(), 1.0)
                assert np.abs(got - want).max() <= 1e-13 * scale + 1e-300, (name, it, key)
    # a sum of two lasers
    both = lasers["simple"] + lasers["gauss"]
    assert isinstance(both, L.CombinedLaser2D) and both.tstop == max(lasers["simple"].tstop, lasers["gauss"].tstop)
    for it, tm in enumerate(g["times"]):
        sim = types.SimpleNamespace(time=float(tm), dx=float(g["dx"]), cpml_thickness=int(g["thickness"]))
        a = lasers["simple"].source_fields(sim, _y(g, lasers["simple"]))
        b = lasers["gauss"].source_fields(sim, _y(g, lasers["gauss"]))
        want = g[f"sum_t{it}_ey"][:ny]
        got = b[0] if a[0] is None else a[0] + b[0]
        assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()


def test_parameter_validation():
    """error behaviour of the reference constructors (callback/laser.py:318-333,446-461)"""
    with pytest.raises(ValueError):
        L.SimpleLaser2D(a0=-1, w0=1e-6, ctau=1e-6)
    with pytest.raises(NotImplementedError):
        L.SimpleLaser2D(a0=1, w0=1e-6, ctau=1e-6, side="xmax")
    with pytest.raises(ValueError):
        L.SimpleLaser2D(a0=1, w0=1e-6, ctau=1e-6, angle_y=2.0)
    with pytest.raises(ValueError):
        L.GaussianLaser2D(a0=1, l0=1e-6, w0=1e-6, ctau=1e-6, ellipticity=1.5)
    with pytest.raises(ValueError):
        L.GaussianLaser2D(a0=1, l0=1e-6, w0=1e-6, ctau=1e-6, p=-1)
    with pytest.raises(ValueError):
        L.GaussianLaser2D(a0=1, l0=1e-6, w0=1e-6, ctau=1e-6, l=0.5)
    with pytest.raises(TypeError):
        L.SimpleLaser2D(a0=1, w0=1e-6, ctau=1e-6) + 3


def test_3d_profiles_match_reference_rows(golden):
    g = golden("g11_laser_profiles")
    cases = json.loads(str(g["cases3"]))
    ny, nz = int(g["ny3"]), int(g["nz3"])
    eng = types.SimpleNamespace(n=(8, ny, nz), d=(float(g["dx"]), float(g["dy"]), float(g["dz3"])))
    for name, (cls, kw) in cases.items():
        las = getattr(L, cls)(**kw)
        for it, tm in enumerate(g["times"]):
            sim = types.SimpleNamespace(time=float(tm), dx=float(g["dx"]), cpml_thickness=int(g["thickness"]),
                                        Ly=float(g["Ly3"]), Lz=float(g["Lz3"]), engine=eng)
            ey, ez = las.source_fields(sim, *las.boundary_yz(sim))
            want_y = g[f"{name}_t{it}_ey"]
            if want_y.size == 0:
                assert ey is None
                continue
            for got, want in ((ey, want_y), (ez, g[f"{name}_t{it}_ez"])):
                want = want.reshape(want.shape[-2], want.shape[-1])[:ny, :nz]
                assert np.abs(got - want).max() <= 1e-13 * max(np.abs(want).max(), 1.0) + 1e-300, (name, it)
