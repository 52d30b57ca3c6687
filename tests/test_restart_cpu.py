"""Host-side logic of RestartDump (reference `callback/restart.py:13-160`) without a GPU: shard paths, the
itime / time bookkeeping of ``load`` (the reference's own test, `tests/test_restart.py:20-74`), ``keep`` trimming,
the signal flag, and the pickling rules of the pieces that hold no device memory (SlabComm drops its process
groups like `core/mpi/mpi_manager.py:35-46`; the patch-list facade survives dill's attribute probing)."""
import signal

import dill
import numpy as np
import pytest

from lambdapic_amd.dist import SlabComm
from lambdapic_amd.restart import RestartDump
from lambdapic_amd.simulation import DevicePatches, MPIFacade


class _FakeSim:
    """what RestartDump touches: mpi.comm / mpi.rank, itime, time, dt, update_lists"""

    def __init__(self):
        self.comm = SlabComm(None, periodic=True, single=True)
        self.mpi = MPIFacade(self)
        self.dt, self.itime, self.time = 0.125, 0, 0.0
        self.payload = np.arange(7.0)
        self.lists_updated = 0
        self.density = lambda x, y: x + y        # dill pickles what pickle cannot (species density profiles)

    def update_lists(self):
        self.lists_updated += 1


def test_paths_and_load_bookkeeping(tmp_path):
    sim = _FakeSim()
    sim.itime, sim.time = 5, 5 * sim.dt
    ckpt = RestartDump(tmp_path / "out", interval=5)
    assert ckpt.stage == "end" and ckpt.interval == 5 and ckpt.dump_signals == []
    ckpt._call(sim)
    assert ckpt._ckpt_dir(5) == tmp_path / "out" / "ckpt_000005"
    assert ckpt._rank_shard_path(5, 3).name == "rank_000003.pkl"
    assert (tmp_path / "out" / "ckpt_000005" / "rank_000000.pkl").is_file()
    loaded = RestartDump.load(ckpt._ckpt_dir(5))
    assert loaded is not sim and loaded.lists_updated == 1
    assert loaded.itime == sim.itime + 1                                   # tests/test_restart.py:69
    np.testing.assert_allclose(loaded.time, loaded.itime * loaded.dt, rtol=1e-15)
    assert np.array_equal(loaded.payload, sim.payload) and loaded.density(2, 3) == 5
    assert loaded.mpi.rank == 0 and loaded.mpi.size == 1 and loaded.comm.group is None


def test_keep_trims_old_checkpoints(tmp_path):
    sim = _FakeSim()
    ckpt = RestartDump(tmp_path, interval=1, keep=2)
    for it in (3, 7, 11, 15):
        sim.itime = it
        ckpt(sim)
    assert sorted(p.name for p in tmp_path.iterdir()) == ["ckpt_000011", "ckpt_000015"]


def test_signal_requests_a_dump(tmp_path):
    old = signal.getsignal(signal.SIGUSR2)
    try:
        ckpt = RestartDump(tmp_path, dump_signals=[signal.SIGUSR2])
        assert not ckpt._dump_requested
        signal.raise_signal(signal.SIGUSR2)
        assert ckpt._dump_requested
        assert RestartDump(tmp_path, dump_signals=False).dump_signals == []
    finally:
        signal.signal(signal.SIGUSR2, old)


def test_slabcomm_pickles_without_groups():
    c = SlabComm(None, periodic=False, single=True)
    c.group = c.p2p_group = object()              # stand-ins for ProcessGroup handles (not picklable state)
    d = dill.loads(dill.dumps(c))
    assert (d.rank, d.size, d.periodic, d.left, d.right) == (0, 1, False, -1, -1)
    assert d.group is None and d.p2p_group is None
    d.rebind("g", "p")
    assert d.group == "g" and d.p2p_group == "p"
    c.size, c.rank = 2, 1                          # a shard of a 2-rank run cannot load into a lone process
    with pytest.raises(RuntimeError):
        dill.loads(dill.dumps(c))


def test_patch_facade_survives_pickling():
    from lambdapic_amd.patch import make_patches_2d
    m = make_patches_2d(8, 8, 1.0, 1.0, 2, 2)
    sim = _FakeSim()
    sim.species = []
    p = DevicePatches(sim, m)
    q = dill.loads(dill.dumps(p))
    assert len(q) == 4 and q.nx == 4 and q[3].ipatch_x == 1
    with pytest.raises(AttributeError):
        q._nope
