"""Host-side logic of the facade that needs no GPU: callback interval rule, id allocation, slab communicator
bookkeeping, engine argument checks."""
import numpy as np
import pytest

from lambdapic_amd.simulation import Callback, Simulation, callback


def _sim():
    s = Simulation(64, 32, 1e-8, 1e-8, boundary_conditions={k: "periodic" for k in ("xmin", "xmax", "ymin", "ymax")})
    return s


def test_interval_rules_match_the_reference():
    """`callback/callback.py:22-45`: int -> itime % interval == 0; float (seconds) -> (time % interval) < dt;
    callable -> bool(interval(sim))"""
    s = _sim()
    dt = s.dt

    @callback("start", interval=3)
    def every3(sim):
        pass

    @callback("start", interval=2.5 * dt)
    def every_2p5_dt(sim):
        pass

    @callback("start", interval=lambda sim: sim.itime in (1, 4))
    def picky(sim):
        pass

    hits = {"every3": [], "every_2p5_dt": [], "picky": []}
    for it in range(12):
        s.itime, s.time = it, it * dt
        for cb in s._triggered([every3, every_2p5_dt, picky]):
            hits[cb.__name__].append(it)
    assert hits["every3"] == [0, 3, 6, 9]
    assert hits["picky"] == [1, 4]
    assert hits["every_2p5_dt"] == [it for it in range(12) if ((it * dt) % (2.5 * dt)) < dt]
    assert every3.stage == "start" and every3.interval == 3


def test_interval_validation_and_class_style_callbacks():
    """`callback/callback.py:11-19,111-145`; the reference's `tests/test_callback.py:170-260`"""
    for bad, err in (("10", TypeError), (None, TypeError), (0, ValueError), (-3, ValueError), (0.0, ValueError),
                     (1.0, ValueError), (1.5, ValueError), (-0.1, ValueError)):
        with pytest.raises(err):
            callback("end", interval=bad)
    for good in (1, 7, np.int64(3), 1e-15, 0.999, lambda sim: True):
        callback("end", interval=good)(lambda sim: None)
    s = _sim()
    s.initialized = True
    with pytest.raises(ValueError):
        bad = lambda sim: None                                       # noqa: E731
        bad.stage, bad.interval = "end", 0
        s.run(1, callbacks=[bad])

    class Count(Callback):
        stage = "start"

        def __init__(self, interval):
            self.interval, self.seen = interval, []

        def _call(self, sim):
            self.seen.append(sim.itime)
            return "ran"

    c = Count(4)
    for it in range(10):
        s.itime = it
        assert c(s) == ("ran" if it % 4 == 0 else None)
    assert c.seen == [0, 4, 8] and c.stage == "start"
    with pytest.raises(NotImplementedError):
        Callback()(s)


def test_particle_ids_never_collide():
    """one running counter per rank and species (rank above bit 50, `core/particles.py:91-116`): blocks of any size,
    any number of window shifts"""
    s = _sim()
    seen = set()
    for ispec in (0, 1):
        for n in (5, 1 << 21, 7, 1 << 22):      # blocks larger than 2^20 particles included
            first = s._next_ids(ispec)
            assert first >> 50 == s.comm.rank
            rng = (first, first + n)
            assert all(rng[1] <= a or rng[0] >= b for a, b, sp in seen if sp == ispec)
            seen.add((rng[0], rng[1], ispec))
            s._id_next[ispec] += n
    assert s._next_ids(0) == (1 << 21) + (1 << 22) + 12


def test_bad_configurations_are_refused():
    with pytest.raises(ValueError):
        Simulation(64, 32, 1e-8, 1e-8, dt_cfl=1.5)
    with pytest.raises(ValueError):
        Simulation(63, 32, 1e-8, 1e-8, npatch_x=2)
    s = _sim()
    with pytest.raises(ValueError):
        s.initialized = True      # (no GPU needed up to the stage table check)
        s.run(1, callbacks=[callback("no_such_stage")(lambda sim: None)])
    with pytest.raises(ValueError):
        s.run(nsteps=1, sim_time=1.0)


# ---- round 4: E half steps merged across the step boundary -- the rule that decides where ------------------------------
def test_defer_rule_blocks_on_every_possible_reader():
    """Simulation._can_defer_e2: the second E half step of a step may wait for the next step's first one only when nothing
    can read or move E in between"""
    from lambdapic_amd.simulation import MovingWindow
    s = _sim()
    s.nx_per_patch = s.nx // s.npatch_x              # (set by initialize(), which needs the GPU)
    s.itime, s.time = 5, 5 * s.dt
    ok = lambda table, last=False, stop=True: s._can_defer_e2(table, last, stop)
    assert ok({})
    assert not ok({}, last=True)                     # the last step of a run() is a plain one
    assert not ok({}, stop=False)                    # a user's stop_callback may look at anything

    @callback("end", interval=1)
    def at_end(sim):
        pass

    @callback("maxwell_2", interval=5)
    def at_m2(sim):
        pass

    @callback("start", interval=3)
    def at_start(sim):
        pass

    @callback("start", interval=lambda sim: False)
    def moody(sim):
        pass

    @callback("maxwell_1", interval=1)
    def inner(sim):
        pass

    assert not ok({"end": [at_end]})
    assert not ok({"maxwell_2": [at_m2]})            # itime 5 % 5 == 0: fires in this step
    s.itime = 6
    assert ok({"maxwell_2": [at_m2]})
    assert ok({"maxwell_1": [inner]})                # after E1 of the next step: no reader in between
    s.itime = 4
    assert ok({"start": [at_start]})                 # next step is 5: 5 % 3 != 0
    s.itime = 5
    assert not ok({"start": [at_start]})             # next step is 6: fires, and does not say it leaves the fields alone
    assert not ok({"start": [moody]})                # an interval function cannot be asked about the next step
    s.defer_e2 = False
    assert not ok({})
    s.defer_e2 = True
    # the moving window: fires at every 'start', touches the fields only when it removes the layers or shifts
    s.itime, s.time = 10, 10 * s.dt
    win = MovingWindow(velocity=299792458.0, start_time=100 * s.dt)
    assert ok({"start": [win]})                      # not started
    win.start_time = 0.0
    assert not ok({"start": [win]})                  # its first active call removes the x layers
    win.num_shifts, win.total_shift = 3, 0.0
    patch = s.nx_per_patch * s.dx
    win.patch_this_shift = 0.2 * patch
    assert ok({"start": [win]})
    win.patch_this_shift = patch - 0.5 * 299792458.0 * s.dt
    assert not ok({"start": [win]})                  # the next call shifts
    back = MovingWindow(velocity=-299792458.0, start_time=0.0)
    back.num_shifts, back.patch_this_shift = 3, -patch + 0.5 * 299792458.0 * s.dt
    assert not ok({"start": [back]})
    back.patch_this_shift = 0.0
    assert ok({"start": [back]})


def test_interval_functions_are_asked_once_per_step():
    s = _sim()
    asked = []

    @callback("start", interval=lambda sim: asked.append(sim.itime) or True)
    def cb(sim):
        pass

    for it in range(3):
        s.itime = it
        for _ in range(4):                               # the stage loop asks several times per step
            assert s._triggered([cb]) == [cb]
    assert asked == [0, 1, 2]


# ---- round 4: the transport objects need no GPU to exist ---------------------------------------------------------------------
def test_loopback_communicator_without_a_gpu():
    import ctypes as C
    from lambdapic_amd import _lib
    from lambdapic_amd.dist import LoopbackComm
    L = _lib.lib()
    for size in (1, 2):
        comm = LoopbackComm(1.0e-6, size)
        kind, rank, csize, left, right, version = comm.native_info()
        assert (kind, rank, csize, version) == (_lib.LPA_COMM_LOOPBACK, 0, size, 0) and left == right == size - 1
        assert comm.arrival_shift(123.0) == (-1.0e-6, 1.0e-6)
        assert (comm.has_left, comm.has_right) == (size == 2, size == 2)
        assert L.lpa_comm_exchange(comm.native, None, 0, None) == 0          # an empty round is legal
        bad = (_lib.lpa_face_msg * 1)()
        bad[0].n_send_lo = -1
        assert L.lpa_comm_exchange(comm.native, bad, 1, None) == -1 and b"negative count" in L.lpa_last_error()
        comm.close()
        assert comm.native is None
    h = C.c_void_p()
    assert L.lpa_comm_create_loopback(C.byref(h), 3, 1) == -1          # a ring of 1, or of 2 (slab + its copy)
    assert L.lpa_comm_create_loopback(C.byref(h), 2, 0) == -1          # loopback rings are periodic


def test_slab_communicators_pickle_without_their_handles():
    import pickle
    from lambdapic_amd.dist import SlabComm
    c = SlabComm(None, single=True)
    c.native = object()                  # (whatever it is, it must not travel)
    assert pickle.loads(pickle.dumps(c)).native is None


def test_bench_watchdog_prints_what_it_has_and_exits_nonzero():
    """bench.py's Watchdog in a child process: an overrun section -> rank 0 prints the partial line with the section marked
    "timeout", exit code 3"""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "d = bench.Watchdog(0); d.partial = {'metric': 'm', 'value': 1.0}\n"
            "d.arm('leg c4', 0.2); time.sleep(30)\n" % str(root))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 1.0 and line["extra"] == [{"workload": "leg c4", "value": None, "error": "timeout"}]
    assert "exceeded its time budget" in r.stderr


def test_particle_message_window_follows_the_counts():
    """dist.MigrateWindowMixin: the part of the fixed-size particle face message that travels -- MARGIN x the largest
    count the ranks saw, a power of two between MIN and the capacity; grows at once, halves at most per retune; a step
    that overflowed the window sends every rank back to the full capacity instead of raising"""
    from lambdapic_amd.dist import MigrateWindowMixin, SlabComm

    class Eng(MigrateWindowMixin):
        migrate_capacity = 262144
        comm = SlabComm(None, periodic=True)

    e = Eng()
    assert e.migrate_window == 262144                      # nothing known yet: the whole buffer
    e._mig_seen = 5000
    assert e._mig_request() == 5000.0 and e._mig_seen == 0
    e._mig_apply(5000.0)
    assert e.migrate_window == 131072                      # wants 32 768 (4 x 5000 -> next power of two), halves per retune
    e._mig_apply(5000.0)
    e._mig_apply(5000.0)
    assert e.migrate_window == 32768
    e._mig_apply(0.0)
    e._mig_apply(0.0)
    e._mig_apply(0.0)
    assert e.migrate_window == 8192 == Eng.MIGRATE_WINDOW_MIN
    e._mig_apply(40000.0)
    assert e.migrate_window == 262144                      # grows at once (4 x 40 000 -> 262 144 = the capacity)
    e._mig_apply(100.0)
    assert e.migrate_window == 131072
    # an overflow of the window is handled, one of the full capacity is the caller's to raise
    assert e._mig_surplus(3) and e._mig_request() == 262144.0
    e._mig_apply(262144.0)
    # (the counters run from sort to sort: what they hold right after the widening was counted against the narrow window)
    assert e.migrate_window == 262144 and e._mig_surplus(3)
    assert e._mig_request() == 0.0 and not e._mig_surplus(3)        # one retune later an overflow is the capacity's
    assert not e._mig_surplus(0)
    # the travelling part of the buffers is a prefix: header + NATTR attributes x window
    import torch
    from lambdapic_amd._lib import LPA_MIG_NATTR
    e._mig_apply(0.0)
    m = {k: torch.zeros(1 + LPA_MIG_NATTR * 262144, dtype=torch.float64) for k in ("s_lo", "s_hi", "r_lo", "r_hi")}
    v = e._mig_views(m)
    assert all(t.numel() == 1 + LPA_MIG_NATTR * 131072 and t.data_ptr() == m[k].data_ptr() for k, t in v.items())
    e.adaptive_migrate_window = False
    assert e.migrate_window == 262144 and e._mig_request() is None and e._mig_views(m) is m
    # allmin of a list: element-wise (one all-reduce carries the sort interval and the window request)
    assert SlabComm(None, periodic=True).allmin([3, -7.5]) == [3.0, -7.5] and SlabComm(None, periodic=True).allmin(4) == 4.0
