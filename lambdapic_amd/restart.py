"""RestartDump -- checkpoints of the device-resident simulation, mirror of the reference's callback
(`callback/restart.py:13-160`).

Same protocol: stage ``end``; every ``interval`` (steps | seconds of simulated time | predicate) each rank pickles
its ``Simulation`` / ``Simulation3D`` with ``dill`` into ``out_dir/ckpt_<itime:06d>/rank_<rank:06d>.pkl``
(`restart.py:80-107`), rank 0 trims all but the ``keep`` newest checkpoint directories (`:109-127`), a POSIX
signal listed in ``dump_signals`` requests one dump at the end of the current step after which ``run`` returns
(`:68-78`, `simulation/simulation.py:889-894,1124-1127`).  ``RestartDump.load(ckpt_dir)`` unpickles this rank's
shard, calls ``update_lists()``, advances ``itime`` by one (the dump happens before the loop's increment) and
re-derives ``time`` (`restart.py:130-160`).

What is device specific: the engines hold HBM tensors, ctypes descriptors, a library handle, streams and a process
group.  Their ``__getstate__`` (engine.py, engine3d.py, device.py, dist.py) moves the grid, the slots in use of
every particle store (ids included) and the CPML psi layers to host arrays and drops the rest; ``__setstate__``
re-allocates on the device (``device=`` of ``load`` overrides the saved one), rebuilds the descriptors and binds
the communicator to the loading process' default group -- the reference re-creates its communicators the same way
(`core/mpi/mpi_manager.py:35-46`).  The tile order of the particle stores is not part of a checkpoint: the first
push after a load sorts, exactly like the first push of a run.

The on-disk format is a dill pickle of THIS package's classes, not the reference's (its Simulation cannot be
imported here: mpi4py / numba / loguru are absent), so the file format is *parity unpinned*; what is pinned by
the tests is resume = no-op: the state after ``load`` equals the state at the dump bit for bit, and N + M steps equal
N steps + dump + load + M steps to the run-to-run reproducibility of the atomics (tests/test_gpu_restart.py).
"""
from __future__ import annotations

import signal
from pathlib import Path

import dill

from . import device as _device


class RestartDump:
    DEFAULT_STAGE = "end"
    device_native = True            # reads the device state directly: no mirror refresh around it
    reads_part_eb = True            # the dumped stores carry ex_part ... bz_part: the push before a dump writes them

    def __init__(self, out_dir, interval=1000, keep=None, dump_signals=False):
        self.stage, self.interval, self.keep = self.DEFAULT_STAGE, interval, keep
        self.out_dir = Path(out_dir)
        self.out_dir.mkdir(parents=True, exist_ok=True)
        # True = the two signals a batch scheduler sends before it kills a job; a sequence = exactly those
        if isinstance(dump_signals, bool):
            dump_signals = (signal.SIGINT, signal.SIGTERM) if dump_signals else ()
        self.dump_signals = list(dump_signals)
        self._dump_requested = False
        for signum in self.dump_signals:
            signal.signal(signum, self._dump_handler)

    def _dump_handler(self, sig, frame):
        self._dump_requested = True        # honoured by run() at the end of the current step

    # ---- where the shards live (`restart.py:80-85`) ---------------------------------------------------------------
    def _ckpt_dir(self, itime: int) -> Path:
        return self.out_dir / ("ckpt_%06d" % itime)

    def _rank_shard_path(self, itime: int, rank: int) -> Path:
        return self._ckpt_dir(itime) / ("rank_%06d.pkl" % rank)

    # ---- callback entry (`restart.py:88-107`): directory by rank 0, one shard per rank, trim, three barriers -------
    def _call(self, sim):
        mpi = sim.mpi
        if mpi.rank == 0:
            self._ckpt_dir(sim.itime).mkdir(parents=True, exist_ok=True)
        mpi.comm.Barrier()
        shard = self._rank_shard_path(sim.itime, mpi.rank)
        tmp = shard.with_suffix(".tmp")
        with tmp.open("wb") as fh:          # written under another name first: a killed dump leaves no half shard
            dill.dump(sim, fh, byref=True, recurse=True)
        tmp.replace(shard)
        mpi.comm.Barrier()
        if mpi.rank == 0 and self.keep:
            self._gc_old_checkpoints(int(self.keep))
        mpi.comm.Barrier()

    __call__ = _call

    def _gc_old_checkpoints(self, keep: int) -> None:
        """all but the ``keep`` newest ``ckpt_*`` directories go (`restart.py:109-127`; names sort by itime)"""
        import shutil
        have = sorted(d for d in self.out_dir.glob("ckpt_*") if d.is_dir())
        for stale in have[:-keep] if keep > 0 else []:
            try:
                shutil.rmtree(stale)
            except OSError as err:             # the reference logs and carries on
                print(f"RestartDump: could not remove {stale}: {err}")

    # ---- loader (`restart.py:130-160`) ------------------------------------------------------------------------
    @staticmethod
    def load(ckpt_dir, comm=None, device=None):
        """``comm``: a ``SlabComm`` (default: rank / world of the initialised ``torch.distributed`` default group,
        rank 0 of 1 without one); the loaded simulation's communicator is re-bound to its groups.  ``device``:
        device to restore onto (default: the one the checkpoint was written from)."""
        import torch.distributed as dist
        if comm is not None:
            rank = comm.rank
        else:
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        _device.RESTORE_DEVICE = device
        try:
            with (Path(ckpt_dir) / ("rank_%06d.pkl" % rank)).open("rb") as fh:
                sim = dill.load(fh)
        finally:
            _device.RESTORE_DEVICE = None
        if comm is not None:
            sim.comm.rebind(comm.group, comm.p2p_group)
        if device is not None:
            sim.device = device
        sim.update_lists()
        sim.itime += 1                     # the dump ran before the loop's increment ...
        sim.time = sim.itime * sim.dt      # ... and time follows itime (`restart.py:153-156`)
        sim.mpi.comm.Barrier()
        return sim
