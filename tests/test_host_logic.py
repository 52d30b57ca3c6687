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
