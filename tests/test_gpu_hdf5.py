"""The HDF5 writers on device state, driven like the reference's own tests of them (`tests/test_hdf5_callback.py`: a real
``Simulation`` / ``Simulation3D``, 32^2 / 32^3 cells, 2 x 2 (x 2) patches, files read back) and compared with the
resident arrays instead of with host patches.  The reference holds no recorded file, so the FILE CONTENT is pinned to
the device arrays (bit exact: a writer only moves data) and the layout -- names, shapes, types, attributes, ``slice``
text -- to what the reference's tests assert."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lambdapic_amd import h5lite
from lambdapic_amd.callbacks import ExtractSpeciesDensity
from lambdapic_amd.hdf5 import LoadParticles, SaveFieldsToHDF5, SaveParticlesToHDF5, SaveSpeciesDensityToHDF5
from lambdapic_amd.simulation import Simulation, Species
from lambdapic_amd.simulation3d import Simulation3D

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not h5lite.available(), reason="neither h5py nor libhdf5")]

PERIODIC2 = {k: "periodic" for k in ("xmin", "xmax", "ymin", "ymax")}
PERIODIC3 = dict(PERIODIC2, zmin="periodic", zmax="periodic")


def _sim2(species=True, **kw):
    sim = Simulation(nx=32, ny=32, dx=1e-7, dy=1e-7, npatch_x=2, npatch_y=2, dt_cfl=0.95, random_seed=1,
                     boundary_conditions=PERIODIC2, **kw)
    e = None
    if species:
        e = Species("electrons", charge=-1, mass=1, density=lambda x, y: 1.0e25 * (1 + (x > 1.6e-6)), ppc=4,
                    momentum_sigma=0.05)
        sim.add_species([e])
    return sim, e


def _sim3(species=True, **kw):
    sim = Simulation3D(nx=32, ny=32, nz=32, dx=1e-7, dy=1e-7, dz=1e-7, npatch_x=2, npatch_y=2, npatch_z=2, random_seed=1,
                       boundary_conditions=PERIODIC3, **kw)
    e = None
    if species:
        e = Species("electrons", charge=-1, mass=1, density=lambda x, y, z: 1.0e25 * (1 + (z > 1.6e-6)), ppc=4,
                    momentum_sigma=0.05)
        sim.add_species([e])
    return sim, e


def _pattern(sim, name):
    """a field with a different value in every cell, written into the resident array; returns the interior"""
    dims = (sim.nx, sim.ny) + ((sim.nz,) if sim.dimension == 3 else ())
    eng, g = sim.engine, sim.engine.ng
    view = eng.view(name) if sim.dimension == 3 else eng.grid.view(name)
    view.copy_(torch.arange(view.numel(), dtype=torch.float64, device=view.device).reshape(view.shape) * 0.5 + 0.25)
    inner = view[(slice(g, -g),) * len(dims)].cpu().numpy()
    assert inner.shape == dims
    return inner


# ---- fields ---------------------------------------------------------------------------------------------------
def test_field_files_of_a_run_2d(tmp_path):
    """`test_hdf5_callback.py:14-47`"""
    sim, _ = _sim2()
    out = tmp_path / "field_output"
    sim.run(21, callbacks=[SaveFieldsToHDF5(prefix=str(out), interval=10, components=["ex", "ey"])])
    assert sorted(os.listdir(out)) == ["000000.h5", "000010.h5", "000020.h5"]
    g = sim.engine.ng
    with h5lite.File(out / "000020.h5", "r") as f:
        assert sorted(f.keys()) == ["ex", "ey"]
        assert f.attrs["nx"] == 32 and f.attrs["ny"] == 32 and f.attrs["itime"] == 20
        assert f.attrs["dx"] == sim.dx and f.attrs["Ly"] == sim.Ly and "slice" not in f.attrs
        assert f.attrs["time"] == pytest.approx(20 * sim.dt, rel=1e-12)
        # stage 'end' of step 20 = the arrays as they are now (the run stopped after that step)
        assert np.array_equal(f["ey"][:], sim.engine.grid.view("ey")[g:-g, g:-g].cpu().numpy())
        assert np.abs(f["ey"][:]).max() > 0


@pytest.mark.parametrize("user, shape, text", [
    (None, (32, 32), None), (np.s_[:, 5], (32, 1), "[:, 5]"), (np.s_[::2, ::3], (16, 11), "[::2, ::3]"),
    (np.s_[16:, :], (16, 32), "[16:, :]"), (np.s_[:, -1], (32, 1), "[:, 31]"), (np.s_[:, np.int64(5)], (32, 1), "[:, 5]"),
])
def test_field_slices_2d(tmp_path, user, shape, text):
    """`test_hdf5_callback.py:251-335,573-637`"""
    sim, _ = _sim2(species=False)
    sim.initialize()
    ref = _pattern(sim, "ex")
    cb = SaveFieldsToHDF5(prefix=str(tmp_path / "out"), interval=1, slice=user, components=["ex"])
    cb._call(sim)
    with h5lite.File(tmp_path / "out" / "000000.h5", "r") as f:
        assert f["ex"].shape == shape
        assert np.array_equal(f["ex"][:], ref if user is None else ref[cb._normalized_slice])
        assert ("slice" not in f.attrs) if text is None else f.attrs["slice"] == text


@pytest.mark.parametrize("user, shape, text", [
    (None, (32, 32, 32), None), (np.s_[:, :, 10], (32, 32, 1), "[:, :, 10]"),
    (np.s_[::2, ::2, ::5], (16, 16, 7), "[::2, ::2, ::5]"), (np.s_[16:, :, :], (16, 32, 32), "[16:, :, :]"),
])
def test_field_slices_3d(tmp_path, user, shape, text):
    """`test_hdf5_callback.py:338-426`"""
    sim, _ = _sim3(species=False)
    sim.initialize()
    ref = {c: _pattern(sim, c) for c in ("ex", "bz", "rho")}
    cb = SaveFieldsToHDF5(prefix=str(tmp_path / "out"), interval=1, slice=user, components=list(ref))
    cb._call(sim)
    with h5lite.File(tmp_path / "out" / "000000.h5", "r") as f:
        for c, a in ref.items():
            assert f[c].shape == shape
            assert np.array_equal(f[c][:], a if user is None else a[cb._normalized_slice])
        assert f.attrs["nz"] == 32 and f.attrs["dz"] == sim.dz and f.attrs["Lz"] == sim.Lz
        assert ("slice" not in f.attrs) if text is None else f.attrs["slice"] == text


def test_all_components_by_default_and_bad_arguments(tmp_path):
    """`callback/hdf5.py:324-333`; `test_hdf5_callback.py:429-498`"""
    sim, _ = _sim3(species=False)
    sim.initialize()
    cb = SaveFieldsToHDF5(prefix=str(tmp_path / "o"), interval=1)
    cb(sim)
    with h5lite.File(tmp_path / "o" / "000000.h5", "r") as f:
        assert sorted(f.keys()) == sorted(["ex", "ey", "ez", "bx", "by", "bz", "jx", "jy", "jz", "rho"])
    with pytest.raises(ValueError):
        SaveFieldsToHDF5(prefix=str(tmp_path / "o"), components=["ex", "phi"])
    for bad in (([0, 1], slice(None), slice(None)), np.s_[..., 10], np.s_[::-1, :, :], np.s_[:, :], np.s_[0:0, :, :],
                np.s_[None, :, :], np.s_[::0, :, :], np.s_[:, :, 32]):
        with pytest.raises(ValueError):
            SaveFieldsToHDF5(prefix=str(tmp_path / "o"), interval=1, slice=bad, components=["ex"])._call(sim)


# ---- species density ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dim, user, shape, text", [
    (2, None, (32, 32), None), (2, np.s_[:, 5], (32, 1), "[:, 5]"), (2, np.s_[::2, ::3], (16, 11), "[::2, ::3]"),
    (2, np.s_[:, -1], (32, 1), "[:, 31]"),
    (3, None, (32, 32, 32), None), (3, np.s_[:, :, 10], (32, 32, 1), "[:, :, 10]"),
    (3, np.s_[::2, ::2, ::5], (16, 16, 7), "[::2, ::2, ::5]"), (3, np.s_[16:, :, :], (16, 32, 32), "[16:, :, :]"),
])
def test_density_files_and_extracted_density(tmp_path, dim, user, shape, text):
    """`test_hdf5_callback.py:49-187,501-620`: the file, the in-memory callback with the same selection and the
    full-box callback agree; the density itself = the species' rho over its charge"""
    sim, e = _sim2() if dim == 2 else _sim3()
    p = Species("ions", charge=2, mass=3672.0, density=(lambda *x: np.full_like(x[0], 0.5e25)), ppc=2)
    sim.add_species([p])                 # a second species, so that the "rho before this species" branch runs too
    out = tmp_path / "out"
    sim.initialize()
    cbs = [SaveSpeciesDensityToHDF5(species=s, prefix=str(out), interval=1, slice=user) for s in (e, p)]
    full = [ExtractSpeciesDensity(sim, s, interval=1) for s in (e, p)]
    part = [ExtractSpeciesDensity(sim, s, interval=1, slice=user) for s in (e, p)]
    sim.run(2, callbacks=cbs + full + part)
    cell = sim.dx * sim.dy * (sim.dz if dim == 3 else 1.0)
    eng = sim.engine
    w_of = (lambda i: eng.species[i].download()["w"].sum()) if dim == 2 else (lambda i: eng.download_species(i)["w"].sum())
    for s, fu, pa in ((e, full[0], part[0]), (p, full[1], part[1])):
        with h5lite.File(out / f"{s.name}_000001.h5", "r") as f:
            assert f["density"].shape == shape and f.attrs["species"] == s.name and f.attrs["itime"] == 1
            assert ("slice" not in f.attrs) if text is None else f.attrs["slice"] == text
            got = f["density"][:]
        want = fu.density if user is None else fu.density[pa._normalized_slice]
        assert fu.density.shape == (32,) * dim and pa.density.shape == shape
        assert np.array_equal(got, want) and np.array_equal(pa.gather(sim), want)
        assert fu.density.sum() * cell == pytest.approx(w_of(s.ispec), rel=1e-9)      # every particle is in it
    assert os.path.exists(out / "electrons_000000.h5")


# ---- particles ------------------------------------------------------------------------------------------------------
def test_particle_files_2d(tmp_path):
    """`test_hdf5_callback.py:189-245`"""
    sim, e = _sim2()
    out = tmp_path / "particles_output"
    sim.run(21, callbacks=[SaveParticlesToHDF5(species=e, prefix=str(out), interval=10, attrs=["x", "y", "w", "id"])])
    assert sorted(os.listdir(out)) == [f"electrons_particles_{k:06d}.h5" for k in (0, 10, 20)]
    with h5lite.File(out / "electrons_particles_000000.h5", "r") as f:
        assert f.attrs["time"] == 0.0 and f.attrs["itime"] == 0
    with h5lite.File(out / "electrons_particles_000020.h5", "r") as f:
        assert sorted(f.keys()) == ["id", "w", "x", "y"]
        assert f["id"].dtype == np.uint64 and f["x"].dtype == np.float64
        n = sim.nx * sim.ny * e.ppc
        assert len(f["x"]) == len(f["y"]) == len(f["w"]) == len(f["id"]) == n
        ids, x, w = f["id"][:], f["x"][:], f["w"][:]
    assert len(np.unique(ids)) == n
    d = sim.engine.species[0].download()           # the run stopped after step 20's 'end' stage
    o, oo = np.argsort(d["_id"].view(np.uint64)), np.argsort(ids)
    assert np.array_equal(ids[oo], d["_id"].view(np.uint64)[o])
    assert np.array_equal(x[oo], d["x"][o]) and np.array_equal(w[oo], d["w"][o])


def test_particle_file_carries_the_fields_the_last_push_gathered_2d(tmp_path):
    """attrs=None writes every attribute, ex_part ... bz_part included (`callback/hdf5.py:660-663`): the push before the
    writer must store them although the writer itself is device native -- the same numbers a host callback at the same
    stage sees in the mirrors (which makes every push write them)"""
    files = {}
    for host in (False, True):
        sim, e = _sim2()
        seen = {}

        def probe(s, seen=seen):                       # a host callback: forces the E/B write-back in every push
            q = [p.particles[0] for p in s.patches]
            seen["ex"] = np.concatenate([v.ex_part[~v.is_dead] for v in q])
            seen["bz"] = np.concatenate([v.bz_part[~v.is_dead] for v in q])
            seen["id"] = np.concatenate([v.id[~v.is_dead] for v in q])
        probe.stage, probe.interval = "end", 10
        out = tmp_path / ("host" if host else "native")
        cbs = [SaveParticlesToHDF5(species=e, prefix=str(out), interval=10)] + ([probe] if host else [])
        sim.run(11, callbacks=cbs)
        with h5lite.File(out / "electrons_particles_000010.h5", "r") as f:
            assert {"ex_part", "ey_part", "ez_part", "bx_part", "by_part", "bz_part"} <= set(f.keys())
            o = np.argsort(f["id"][:])
            files[host] = {a: f[a][:][o] for a in ("ex_part", "bz_part", "x")}
        if host:
            oo = np.argsort(seen["id"].view(np.uint64))
            assert np.array_equal(files[True]["ex_part"], seen["ex"][oo]) and np.array_equal(files[True]["bz_part"], seen["bz"][oo])
    assert np.abs(files[False]["ex_part"]).max() > 0                                   # not the zeros of a lazily allocated row
    scale = np.abs(files[True]["ex_part"]).max()
    # (two runs of the same problem: the order of the FP64 atomics differs, nothing else)
    assert np.abs(files[False]["ex_part"] - files[True]["ex_part"]).max() <= 1e-9 * scale
    assert np.abs(files[False]["bz_part"] - files[True]["bz_part"]).max() <= 1e-9 * np.abs(files[True]["bz_part"]).max()


def test_particle_files_default_attributes_3d_and_dead_particles(tmp_path):
    """attrs=None = everything the store holds (`callback/hdf5.py:660-663`); dead slots are not written (`:686-691`)"""
    bc = dict(PERIODIC3, xmin="pml", xmax="pml")
    sim = Simulation3D(nx=32, ny=16, nz=16, dx=1e-7, dy=1e-7, dz=1e-7, npatch_x=2, random_seed=1, boundary_conditions=bc,
                       cpml_thickness=4)
    e = Species("hot", charge=-1, mass=1, density=lambda x, y, z: 1.0e24 * (x > 2.0e-6), ppc=2, momentum_sigma=1.5)
    sim.add_species([e])
    cb = SaveParticlesToHDF5(species=e, prefix=str(tmp_path), interval=12)
    sim.run(13, callbacks=[cb])
    d = sim.engine.download_species(0)
    with h5lite.File(tmp_path / "hot_particles_000000.h5", "r") as f:
        assert 0 < len(d["x"]) < len(f["id"])       # some left through the open x faces
    with h5lite.File(tmp_path / "hot_particles_000012.h5", "r") as f:
        assert sorted(f.keys()) == sorted(["x", "y", "z", "ux", "uy", "uz", "inv_gamma", "w", "_id", "id"])
        ids = f["id"][:]
        assert len(ids) == len(d["x"]) and np.array_equal(f["_id"][:].view(np.uint64), ids)
        o, oo = np.argsort(d["_id"].view(np.uint64)), np.argsort(ids)
        for a in ("x", "z", "ux", "inv_gamma", "w"):
            assert np.array_equal(f[a][:][oo], d[a][o]), a
    with pytest.raises(ValueError):
        SaveParticlesToHDF5(species=e, prefix=str(tmp_path), attrs=["x", "chi"])._call(sim)


# ---- particles back in -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dim", [2, 3])
def test_load_particles_round_trip(tmp_path, dim):
    """`tests/test_load_particles.py:11-124,243-...`: save, load into a simulation that starts empty, compare the sets --
    and the loaded particles are good citizens of the resident store (fresh unique ids, a step runs, charge adds up)"""
    src, e = _sim2() if dim == 2 else _sim3()
    src.initialize()
    names = ["x", "y", "w", "ux", "uz"] + (["z"] if dim == 3 else [])
    SaveParticlesToHDF5(species=e, prefix=str(tmp_path), interval=1, attrs=names)(src)
    file = tmp_path / "electrons_particles_000000.h5"
    dst, e2 = _sim2(species=False) if dim == 2 else _sim3(species=False)
    e2 = Species("electrons", charge=-1, mass=1, density=None, ppc=0)
    dst.add_species([e2])
    dst.initialize()
    get = (lambda sim: sim.engine.species[0].download()) if dim == 2 else (lambda sim: sim.engine.download_species(0))
    assert len(get(dst)["x"]) == 0
    cb = LoadParticles(species=e2, file=str(file))
    assert cb.stage == "init" and cb.interval(dst)
    cb(dst)
    a, b = get(src), get(dst)
    assert len(b["x"]) == len(a["x"]) == 32 ** dim * 4
    for k in names:
        assert np.array_equal(np.sort(a[k]), np.sort(b[k])), k
    o, oo = np.argsort(a["x"]), np.argsort(b["x"])
    assert np.array_equal(a["ux"][o], b["ux"][oo])                       # attributes stayed together
    assert np.array_equal(b["uy"], np.zeros_like(b["uy"]))               # not in the file
    assert np.allclose(b["inv_gamma"], 1 / np.sqrt(1 + b["ux"] ** 2 + b["uz"] ** 2), rtol=1e-15)
    assert len(np.unique(b["_id"].view(np.uint64))) == len(b["x"])
    dst.run(3)
    d = dst.engine.diagnostics()
    assert d["nalive"][0] == len(a["x"])
    assert d["charge"] == pytest.approx(-1.602176634e-19 * a["w"].sum(), rel=1e-9)


def test_load_particles_edge_cases(tmp_path):
    """`tests/test_load_particles.py:126-240`: a file without weights, an empty file, no file; particles outside the box"""
    sim, _ = _sim2(species=False)
    e = Species("electrons", charge=-1, mass=1, density=None, ppc=0)
    sim.add_species([e])
    sim.initialize()
    rng = np.random.default_rng(0)
    with h5lite.File(tmp_path / "incomplete.h5", "w") as f:
        f.create_dataset("x", data=rng.uniform(0, 3.2e-6, 100))
        f.create_dataset("y", data=rng.uniform(0, 3.2e-6, 100))
    with pytest.raises(ValueError):
        LoadParticles(species=e, file=str(tmp_path / "incomplete.h5"))(sim)
    with h5lite.File(tmp_path / "empty.h5", "w") as f:
        for k in ("x", "y", "w"):
            f.create_dataset(k, data=np.array([]))
        f.create_dataset("id", data=np.array([], dtype=np.uint64))
    LoadParticles(species=e, file=str(tmp_path / "empty.h5"))(sim)
    assert sim.engine.species[0].n == 0
    with pytest.raises(FileNotFoundError):
        LoadParticles(species=e, file=str(tmp_path / "nonexistent.h5"))(sim)
    # the box spans [-dx/2, 31.5 dx): what lies outside belongs to no patch and is dropped
    x = np.array([-0.6e-7, -0.4e-7, 1.0e-6, 31.4e-7, 31.6e-7, 1.0e-6])
    y = np.array([1.0e-6, 1.0e-6, 1.0e-6, 1.0e-6, 1.0e-6, 40e-7])
    with h5lite.File(tmp_path / "edge.h5", "w") as f:
        f.create_dataset("x", data=x), f.create_dataset("y", data=y), f.create_dataset("w", data=np.arange(6.0) + 1)
        f.create_dataset("chi", data=np.zeros(6))          # not an attribute of the store: ignored
    cb = LoadParticles(species=e, file=str(tmp_path / "edge.h5"))
    cb._batch_size = 4                                      # two batches
    cb(sim)
    assert sorted(sim.engine.species[0].download()["w"]) == [2.0, 3.0, 4.0]


# ---- two ranks --------------------------------------------------------------------------------------------------------
def _free_port():
    # below the kernel's ephemeral range (32768-60999): an outgoing gloo connection of an earlier test cannot sit on it
    # (a port taken from bind(0) was, once in a few hundred launches, in use again by the time the store listened)
    import random
    for _ in range(200):
        p = random.randrange(20000, 30000)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", p))
        except OSError:
            continue
        finally:
            s.close()
        return p
    raise RuntimeError("no free port")


def _worker(rank, world, port, tmp, q):
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from lambdapic_amd.dist import SlabComm
    comm = SlabComm(None, periodic=True) if world > 1 else None
    # (patches of 16 x 16 cells whatever the number of ranks: the loading is a function of the patch origins)
    sim = Simulation(nx=64, ny=32, dx=1e-7, dy=1e-7, npatch_x=4 // world, npatch_y=2, random_seed=4,
                     boundary_conditions=PERIODIC2, comm=comm, sort_interval=4)
    e = Species("e", charge=-1, mass=1, density=lambda x, y: 1.0e25 * (1 + (x > 3.2e-6)), ppc=4, momentum_sigma=0.3)
    sim.add_species([e])
    out = os.path.join(tmp, f"w{world}")
    sel = np.s_[3::5, ::2]
    dens = ExtractSpeciesDensity(sim, e, interval=1, slice=np.s_[40:, :])      # rank 0 holds nothing of it at 2 ranks
    sim.run(9, callbacks=[SaveFieldsToHDF5(prefix=out, interval=4, components=["ey", "bz", "rho"]),
                          SaveFieldsToHDF5(prefix=out + "/sel", interval=8, components=["jx"], slice=sel),
                          SaveSpeciesDensityToHDF5(e, prefix=out, interval=8, slice=np.s_[:, 7]),
                          SaveParticlesToHDF5(e, prefix=out, interval=8, attrs=["x", "ux"]), dens])
    got = dens.gather(sim)
    q.put((rank, None if got is None else got.shape, dens.density.shape))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_ranks_write_the_files_of_one(tmp_path):
    ctx = mp.get_context("spawn")
    shapes = {}
    for world in (1, 2):
        q, port = ctx.Queue(), _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
        for p in procs:
            p.daemon = True
            p.start()
        try:
            shapes[world] = sorted(q.get(timeout=240) for _ in range(world))
            for p in procs:
                p.join(timeout=60)
                assert p.exitcode == 0
        finally:
            for p in procs:
                if p.is_alive():
                    p.terminate()
    assert shapes[1] == [(0, (24, 32), (24, 32))]
    assert shapes[2] == [(0, (24, 32), (0,)), (1, None, (24, 32))]
    one, two = tmp_path / "w1", tmp_path / "w2"
    names = sorted(os.path.relpath(os.path.join(r, n), one) for r, _, fs in os.walk(one) for n in fs)
    assert names == sorted(os.path.relpath(os.path.join(r, n), two) for r, _, fs in os.walk(two) for n in fs)
    assert names == ["000000.h5", "000004.h5", "000008.h5", "e_000000.h5", "e_000008.h5", "e_particles_000000.h5",
                     "e_particles_000008.h5", "sel/000000.h5", "sel/000008.h5"]
    for n in names:
        with h5lite.File(one / n, "r") as a, h5lite.File(two / n, "r") as b:
            assert sorted(a.keys()) == sorted(b.keys()) and sorted(a.attrs.keys()) == sorted(b.attrs.keys())
            for k in a.attrs.keys():
                assert a.attrs[k] == b.attrs[k], (n, k)
            if "particles" in n:        # rank by rank: another order and other ids (the rank is part of an id), same particles
                assert len(np.unique(b["id"][:])) == len(a["id"]) == 64 * 32 * 4
                for k in ("x", "ux"):
                    x, y = np.sort(a[k][:]), np.sort(b[k][:])
                    assert np.abs(x - y).max() <= 1e-8 * np.abs(x).max(), (n, k)
                continue
            for k in a.keys():
                x, y = a[k][:], b[k][:]
                assert x.shape == y.shape and (np.abs(x).max() > 0 or k == "bz")      # (B is still zero after one step)
                assert np.abs(x - y).max() <= 1e-9 * max(np.abs(x).max(), 1e-300), (n, k)
    with h5lite.File(two / "sel" / "000008.h5", "r") as f:
        assert f["jx"].shape == (13, 16) and f.attrs["slice"] == "[3::5, ::2]"


@pytest.mark.parametrize("dim", [2, 3])
def test_a_sparse_density_diagnostic_costs_one_real_deposit_per_trigger(dim):
    """a 'current_deposition' callback needs per-species rho only in the steps it runs in: those deposit rho for real, the
    others keep advancing it with the continuity equation -- and the density it extracts is the one an every-step
    diagnostic sees at the same step"""
    def run(interval):
        sim, e = _sim2() if dim == 2 else _sim3()
        sim.add_species([Species("ions", charge=1, mass=1836.0, density=(lambda *x: np.full_like(x[0], 1.5e25)), ppc=2)])
        sim.initialize()
        sim.engine.overflow_sort_fraction = 0
        cb = ExtractSpeciesDensity(sim, e, interval=interval)
        sim.run(13, callbacks=[cb])
        return cb.density, dict(sim.engine.rho_steps)
    often, steps_often = run(1)
    sparse, steps_sparse = run(6)                  # triggers at itime 0, 6, 12
    assert steps_often["continuity"] == 0
    assert steps_sparse["continuity"] >= 8 and steps_sparse["anchor"] <= 5      # 3 triggers + the sorts
    assert np.abs(sparse - often).max() <= 1e-10 * np.abs(often).max()
