"""Host logic of the HDF5 writers (no GPU): the ``np.s_`` selections (`callback/hdf5.py:14-160`; the accepted forms,
errors and text forms are those the reference's `tests/test_hdf5_callback.py:429-498,573-652` asks for) and the file
layer ``h5lite`` (h5py's interface on libhdf5 when h5py is not installed), cross-read with a real h5py where the
image has one for another interpreter."""
import os
import shutil
import subprocess
import types

import numpy as np
import pytest

from lambdapic_amd import h5lite
from lambdapic_amd.slices import normalize_slice, part_in_range, selected_shape, slab_selection, slice_text

needs_hdf5 = pytest.mark.skipif(not h5lite.available(), reason="neither h5py nor libhdf5 in this image")


@pytest.mark.parametrize("user, dims, want, text", [
    (np.s_[:, 5], (32, 32), (slice(0, 32, 1), slice(5, 6, 1)), "[:, 5]"),
    (np.s_[:, -1], (32, 32), (slice(0, 32, 1), slice(31, 32, 1)), "[:, 31]"),
    (np.s_[::2, ::3], (32, 32), (slice(0, 32, 2), slice(0, 32, 3)), "[::2, ::3]"),
    (np.s_[16:, :], (32, 32), (slice(16, 32, 1), slice(0, 32, 1)), "[16:, :]"),
    (np.s_[16:, :, :], (32, 32, 32), (slice(16, 32, 1),) + (slice(0, 32, 1),) * 2, "[16:, :, :]"),
    (np.s_[:, np.int64(5)], (32, 32), (slice(0, 32, 1), slice(5, 6, 1)), "[:, 5]"),
    (np.s_[4:-4:5, :100, 7], (32, 24, 16), (slice(4, 28, 5), slice(0, 24, 1), slice(7, 8, 1)), "[4:28:5, :, 7]"),
    (np.s_[-8:, 2:9], (32, 32), (slice(24, 32, 1), slice(2, 9, 1)), "[24:, 2:9]"),
])
def test_normalised_forms(user, dims, want, text):
    norm = normalize_slice(len(dims), user, dims)
    assert norm == want
    assert slice_text(norm, dims) == text
    assert selected_shape(norm) == np.empty(dims)[norm].shape
    assert normalize_slice(len(dims), None, dims) is None


@pytest.mark.parametrize("user, ndim", [
    (([0, 1], slice(None)), 2),          # a list is not an index
    (np.s_[..., 10], 3),
    (np.s_[::-1, :, :], 3),
    (np.s_[:, :], 3),                    # axis count
    (np.s_[0:0, :, :], 3),
    (np.s_[None, :, :], 3),
    (np.s_[::0, :, :], 3),
    (np.s_[:, :, 32], 3),
    (np.s_[:, -33], 2),
    (np.s_[40:, :], 2),                  # clamps to an empty range
])
def test_rejected_forms(user, ndim):
    with pytest.raises(ValueError):
        normalize_slice(ndim, user, (32,) * ndim)


def test_parts_of_a_range_tile_it():
    rng = np.random.default_rng(0)
    for _ in range(300):
        n = int(rng.integers(1, 60))
        lo = int(rng.integers(0, n))
        s = slice(lo, int(rng.integers(lo + 1, n + 1)), int(rng.integers(1, 9)))
        whole = np.arange(n)[s]
        pieces, size = np.full(len(whole), -1), int(rng.integers(1, 17))
        for off in range(0, n, size):
            part = part_in_range(s, off, min(size, n - off))
            if part is None:
                assert not len([i for i in whole if off <= i < off + size])
                continue
            loc, k0, cnt = part
            got = np.arange(off, min(off + size, n))[loc]
            assert len(got) == cnt
            pieces[k0: k0 + cnt] = got
        assert (pieces == whole).all()


@pytest.mark.parametrize("nranks", [1, 2, 4])
def test_slab_shares_assemble_the_selection(nranks):
    box = np.random.default_rng(1).random((32, 12, 8))
    for user in (None, np.s_[:, :, 3], np.s_[::3, 1::2, :], np.s_[17:, :, :], np.s_[5, :, ::7], np.s_[7:9, 2, 1]):
        norm = normalize_slice(3, user, box.shape)
        out, shape = None, None
        for r in range(nranks):
            sim = types.SimpleNamespace(nx=32, ny=12, nz=8, dimension=3, comm=types.SimpleNamespace(rank=r, size=nranks))
            local, oidx, shape = slab_selection(sim, norm)
            out = np.full(shape, np.nan) if out is None else out
            if local is not None:
                n = 32 // nranks
                out[oidx] = box[r * n: (r + 1) * n][local]
        want = box if user is None else box[tuple(norm)]
        assert shape == want.shape and (out == want).all()


@needs_hdf5
def test_file_round_trip_and_foreign_reader(tmp_path):
    name = tmp_path / "t.h5"
    a = np.arange(24.0).reshape(4, 6)
    with h5lite.File(name, "w") as f:
        d = f.create_dataset("ex", shape=(4, 6), dtype="f8", chunks=(2, 3))
        assert (d[:] == 0).all()                                   # fill value
        d[0:2, 0:3] = a[0:2, 0:3]
        d[2:4] = a[2:4]
        f.create_dataset("id", data=np.arange(5), dtype="u8")
        f.create_dataset("none", shape=(0,), dtype="f8")
        f.attrs["nx"], f.attrs["dx"], f.attrs["slice"], f.attrs["time"] = 4, 0.1, "[:, 5]", np.float64(1.5)
        f.attrs["nx"] = 5                                          # overwrite
    with h5lite.File(name, "a") as f:
        f["ex"][0:2, 3:6] = 7.0
        f.attrs["itime"] = 3
    want = a.copy()
    want[0:2, 3:6] = 7
    with h5lite.File(name, "r") as f:
        assert sorted(f.keys()) == ["ex", "id", "none"] and "ex" in f and "ey" not in f
        assert f["ex"].shape == (4, 6) and f["ex"].dtype == np.float64 and len(f["id"]) == 5
        assert f["id"].dtype == np.uint64 and f["none"].shape == (0,)
        assert (f["ex"][:] == want).all() and (f["ex"][:, 0] == want[:, 0]).all()
        assert (f["ex"][::2, ::3] == want[::2, ::3]).all() and (f["ex"][1:3, 2:5] == want[1:3, 2:5]).all()
        assert f.attrs["nx"] == 5 and f.attrs["dx"] == 0.1 and f.attrs["slice"] == "[:, 5]"
        assert f.attrs["time"] == 1.5 and f.attrs["itime"] == 3
        assert "slice" in f.attrs and "species" not in f.attrs
        with pytest.raises(KeyError):
            f["ey"]
    # a real h5py (the image carries one for another interpreter) reads the same file the same way
    other = shutil.which("python3.9") or "/opt/conda/bin/python3.9"
    probe = "import h5py, numpy"
    if h5lite._h5py is not None or not os.path.exists(other) or subprocess.run([other, "-c", probe]).returncode:
        pytest.skip("no second interpreter with h5py")
    code = f"""
import h5py, numpy as np
with h5py.File({str(name)!r}, "r") as f:
    assert sorted(f) == ["ex", "id", "none"]
    assert f["ex"].shape == (4, 6) and f["ex"].chunks == (2, 3) and f["ex"].dtype == np.float64
    assert f["id"].dtype == np.uint64 and list(f["id"][:]) == [0, 1, 2, 3, 4]
    assert f["ex"][0, 3] == 7.0 and f["ex"][3, 5] == 23.0
    assert f.attrs["nx"] == 5 and f.attrs["slice"] == "[:, 5]" and isinstance(f.attrs["slice"], str)
    assert f.attrs["time"] == 1.5 and f.attrs["itime"] == 3 and f.attrs["dx"] == 0.1
"""
    subprocess.run([other, "-c", code], check=True)


def test_no_library_fails_loudly(monkeypatch):
    if h5lite._h5py is not None:
        pytest.skip("h5py present")
    monkeypatch.setattr(h5lite, "_LIB", None)
    monkeypatch.setattr(h5lite, "_CANDIDATES", ("/nonexistent/libhdf5.so",))
    monkeypatch.setattr(h5lite.ctypes.util, "find_library", lambda _n: None)
    monkeypatch.delenv("LPA_HDF5_LIB", raising=False)
    with pytest.raises(ImportError):
        h5lite.require()
