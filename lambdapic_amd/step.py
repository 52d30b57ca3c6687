"""One host call per step: ``lpa_step`` (csrc/lpa_step.hip) enqueues the no-callback stage sequence of a step -- what the
per-stage methods of the engines issue one ctypes call at a time (the reference walks the same stages from Python,
`simulation/simulation.py:937-1122`).  On launch-bound configs (C3: a 4 M-particle laser-target on 2 M cells, ~20 launches
of 5-40 us per step) the Python stage loop issues a step no faster than the GPU executes it; through ``lpa_step`` the host
cost per launch is the HIP launch itself.

Slab ranks (``comm.size > 1``):

* with the library's own transport (``SlabComm.attach_rccl``, ``LoopbackComm``) the descriptor carries a slab section and
  the SAME single call also moves the x faces (ncclSend / ncclRecv groups issued from C on the step's stream): two message
  rounds per step -- J + rho + every species' leavers | E (E2 of this step and E1 of the next when the caller runs steps
  back to back, ``defer_e2``); B does not travel (``local_b_guards``; four rounds without it: + B1, B2 with the jx plane of
  the continuity update);
* without it (gloo rehearsals, ranks sharing a GPU in the tests, torch's own nccl group) the engines call ``step_stages``
  for the ranges between two exchanges and move the faces from Python (engines: ``_step_segments``).

The descriptor is rebuilt for every call from the engine's CURRENT stores (a few microseconds per species): nothing that
holds a device address is cached across steps, so a sort, a re-allocation, a window shift or an upload between two steps
cannot leave a stale pointer behind.  What stays in Python: the sort (it needs the live count on the host) and everything
a callback does.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check


class FusedStepMixin:
    """needs from the engine: ``dim``, ``_grid_struct()``, ``_species_entries()`` (yielding, per species, a dict with
    ``pc`` (lpa_particles), ``tiling`` (or None), ``n_sorted``, ``pp`` (push params), ``overflow`` / ``count`` tensors,
    ``after`` (hook run after the push) and -- slab ranks -- ``mig`` (see ``_slab_species``)), ``_slab_fill(slab)``,
    ``_cpml_axes``, ``pml``, ``fused_cpml``, ``local_axes``, ``eps0``, ``absorb`` and the rho mixin"""

    fused_step = True          # steps go through lpa_step (False: the per-stage calls)
    _e2_pending = False        # the last step left its second E half step to the next one's first (run_steps)
    _e2_dt = 0.0

    def _flush_e2(self):
        """complete a deferred second E half step now (with its guard stage): whoever is about to read or advance the
        fields outside ``run_steps`` sees the state a plain ``step()`` leaves"""
        if self._e2_pending:
            self._e2_pending = False
            self.update_efield(0.5 * self._e2_dt)
            self.sync_guard_fields(("ex", "ey", "ez") if self.dim == 2 else 1)
    fuse_species = True        # 3-D: every tile-ordered species in ONE launch (lpa_push_deposit_tiled_multi_3d)

    # Slab ranks: the B half steps of ``lpa_step`` advance the x guard planes at the faces with a neighbour themselves (ng
    # planes low, ng - 1 high: a B update reads E at its node and one node up, and the E guard planes are current after
    # every E guard stage) -- the values the neighbour computes for its interior, bit for bit -- so B never travels: two
    # message rounds per step (E | J + rho + particles) instead of four.  What it takes: psi rows for the x guard planes in
    # the y / z CPML layers (``psi_ptr``), all ng B guard planes current whenever B was changed outside lpa_step (initial
    # fields, window shifts and ``sync_guard_fields`` leave them so), no injection within ng + 1 nodes of a shared face.
    local_b_guards = True
    _b_guards_current = False
    # overlapped native slab steps: the E guard planes (and the rows of the B half step that read them) travel on the second
    # stream as well -- both message rounds of a step behind the interior push (LPA_STEP_E_ROUND_IN_LINE turns it off)
    overlap_e_round = True
    fused_fold_unpack = True   # native slab steps: the J / rho fold and every species' arrivals in one launch
    fused_rest_pack = True     # ... and the leaver pack inside the launch that pushes the overflow list + arrival area
    fused_sweep_tails = True   # the current reset rides in the first B sweep's launch, the rho continuity update in the second's

    def local_b(self):
        if not (self.local_b_guards and self.comm.size > 1 and self.can_fuse()):
            return False
        # (the antenna of a laser sits cpml_thickness + 2 nodes inside the rank that owns the x-min layer: clear of its
        # high face by more than the guard)
        return self.n_x_local() >= getattr(self, "cpml_thickness", 0) + 2 + 2 * self.ng + 2

    _event_pool = None

    def reserve_kernel_events(self, pairs):
        """create ``pairs`` timer event pairs now (a torch event gets its HIP event at its first record: two records on the
        stream, ~10 us of stream time): a bench that times K steps reserves them before its timed region, so that a timed
        step carries only the two records around its kernel.  Reserve what the run needs and no more: recorded timing
        events that stay alive slow the kernels down (C2's K1: 1.64 ms with 64 pairs alive, 1.67 with 512, 1.70 with 4096)"""
        import os
        if os.environ.get("LPA_NO_EVENT_POOL"):       # (A/B)
            return
        pairs = int(os.environ.get("LPA_EVENT_POOL_N", pairs))
        stream = torch.cuda.current_stream(self.device)
        pool = []
        for _ in range(int(pairs)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            b.record(stream)
            pool.append((a, b))
        self._event_pool = pool

    def can_fuse(self):
        return self.fused_step and (self.pml is None or self.fused_cpml)

    def native_slab(self):
        """is this a slab rank whose faces travel through the library's own transport?"""
        return self.comm.size > 1 and self.comm.native is not None

    def one_call_step(self):
        """can ``lpa_step`` enqueue the whole step (exchanges included)?"""
        return self.can_fuse() and (self.comm.size == 1 or self.comm.native is not None)

    def step_stages(self, dt, first, last, defer_e2=False):
        """enqueue stages ``first .. last`` (LPA_STAGE_*) of one step in one call.  The stores must be sorted when
        they are due BEFORE LPA_STAGE_RESET is reached (``sort_due_species``); the rho mode of the step is decided
        when LPA_STAGE_RESET is part of the range and closed when LPA_STAGE_FOLD is.  ``defer_e2``: LPA_STAGE_E2 is left
        to the next step's LPA_STAGE_E1, which then applies both half steps in one sweep (LPA_STEP_DEFER_E2 /
        LPA_STEP_E1_DOUBLE: the caller runs another step next and nothing reads E in between)."""
        if not self.can_fuse():
            raise _lib.LpaError("lpa_step drives the fused CPML sweeps")
        native = self.native_slab()
        if self.comm.size > 1 and not native and first <= _lib.LPA_STAGE_FOLD <= last:
            raise _lib.LpaError("slab ranks without a native transport fold their currents from Python (sync_currents)")
        if first <= _lib.LPA_STAGE_B1 and not self._b_guards_current and self.local_b():
            # once: whatever wrote the initial B (a test, a loader) may have left the x guard planes behind -- from here on
            # the B sweeps carry them along
            self._b_guards_current = True
            self.sync_guard_fields(("bx", "by", "bz") if self.dim == 2 else 2)
        self._dt_hint = dt        # (the first sort of a store sizes its sort interval from the particles' speed)
        d = _lib.lpa_step_desc()
        d.grid = self._grid_struct()
        d.dim, d.local_axes, d.dt, d.eps0 = self.dim, self.local_axes, dt, self.eps0
        # a deferred second E half step (run_steps): left out at the end of one step, done together with the first half
        # step of the next one in a single sweep
        d.flags = 0
        if defer_e2 and last >= _lib.LPA_STAGE_E2:
            d.flags |= _lib.LPA_STEP_DEFER_E2
        if first <= _lib.LPA_STAGE_E1 and self._e2_pending:
            d.flags |= _lib.LPA_STEP_E1_DOUBLE
        if self.local_b():
            d.flags |= (_lib.LPA_STEP_B_EXT_LO if self.comm.has_left else 0) | (_lib.LPA_STEP_B_EXT_HI if self.comm.has_right else 0)
        if not self.overlap_e_round:
            d.flags |= _lib.LPA_STEP_E_ROUND_IN_LINE
        if not self.fused_fold_unpack:
            d.flags |= _lib.LPA_STEP_SEPARATE_UNPACK
        if not self.fused_rest_pack:
            d.flags |= _lib.LPA_STEP_SEPARATE_PACK
        if not self.fused_sweep_tails:
            d.flags |= _lib.LPA_STEP_SEPARATE_TAILS
        keep = []
        if self.pml is not None:
            for fld, arr in ((True, d.e_axes), (False, d.b_axes)):
                axes = self._cpml_axes(fld, 0.5 * dt)
                for a, ax in enumerate(axes):
                    arr[a] = C.pointer(ax)
        if first <= _lib.LPA_STAGE_RESET <= last:
            for i in self.sort_due_species():
                self.sort(i)
            self._decide_phase()
        d.continuity = int(self._no_rho)
        d.fuse_species = int(self.dim == 3 and self.fuse_species and self.fuse_worthwhile())
        if self.absorb and self.rho_continuity and self._rho_available():
            lst, cnt, cap = self._absorbed_bufs()
            d.absorbed, d.absorbed_count, d.absorbed_capacity = lst.data_ptr(), cnt.data_ptr(), cap
            d.absorbed_spill = self._absorbed_spill().data_ptr()
        push = first <= _lib.LPA_STAGE_PUSH <= last
        fold = first <= _lib.LPA_STAGE_FOLD <= last
        timed = self.kernel_events is not None and push
        self.kernel_events_step = False    # (fused species: one launch, one event pair -- on the first tiled species)
        # the species table is needed by the push and, on slab ranks, by the fold (leavers / arrivals)
        # (slab ranks: with the migration bookkeeping -- the fold's leavers / arrivals, and the arrival cursor that bounds
        # the loose range of the push)
        slab_rank = self.comm.size > 1
        entries = list(self._species_entries(dt, slab_rank, pushed=not push)) if (push or (native and fold)) else []
        arr = (_lib.lpa_step_species * max(len(entries), 1))()
        stream = torch.cuda.current_stream(self.device)
        edge_events = []
        for k, ent in enumerate(entries):
            e = arr[k]
            tiling, n_sorted = ent["tiling"], ent["n_sorted"]
            e.p, e.n_sorted, e.pp = ent["pc"], n_sorted, ent["pp"]
            e.t = C.pointer(tiling) if tiling is not None else None
            if ent["overflow"] is not None:
                e.overflow, e.overflow_count = ent["overflow"].data_ptr(), ent["count"].data_ptr()
            mig = ent.get("mig")
            if mig is not None:
                m = mig["bufs"]
                e.mig.s_lo, e.mig.s_hi, e.mig.r_lo, e.mig.r_hi = (m[k_].data_ptr() for k_ in ("s_lo", "s_hi", "r_lo", "r_hi"))
                e.mig.cursor, e.mig.surplus = mig["cursor"].data_ptr(), mig["surplus"].data_ptr()
                e.mig.fs = C.pointer(mig["fs"]) if mig["fs"] is not None else None
                e.mig.area_capacity, e.mig.edge_cols = mig["area"], mig["cols"]
                if mig.get("overflow_edge") is not None:       # (overlapped steps: the edge part's own list and counter)
                    e.mig.overflow_edge, e.mig.overflow_edge_count = mig["overflow_edge"].data_ptr(), mig["edge_count"].data_ptr()
            if timed and tiling is not None and n_sorted > 0 and not (d.fuse_species and self.kernel_events_step):
                def pair():
                    if self._event_pool:       # (made ahead of the timed region: reserve_kernel_events)
                        return self._event_pool.pop()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream)           # (creates the HIP events; lpa_step records them again around the launch)
                    b.record(stream)
                    return a, b
                e0, e1 = pair()
                e.ev_start, e.ev_stop = e0.cuda_event, e1.cuda_event
                self.kernel_events.append((e0, e1))
                self.kernel_events_step = True
                if mig is not None and mig.get("overflow_edge") is not None:     # the edge part is a launch of its own
                    f0, f1 = pair()
                    e.mig.ev_edge_start, e.mig.ev_edge_stop = f0.cuda_event, f1.cuda_event
                    edge_events.append((f0, f1))
            keep.append(ent)
        d.nspecies, d.species = len(entries), arr
        if native:
            slab = _lib.lpa_step_slab()
            slab.comm = self.comm.native
            keep.append(self._slab_fill(slab))
            d.slab = C.pointer(slab)
            keep.append(slab)
        check(self.L.lpa_step(C.byref(d), first, last, stream.cuda_stream), "lpa_step")
        if edge_events and native and push and fold and slab.overlap_cols > 0:
            self.kernel_events.extend(edge_events)        # (the edge and the interior launch of a species add up)
        if d.flags & _lib.LPA_STEP_E1_DOUBLE:
            self._e2_pending = False
        if d.flags & _lib.LPA_STEP_DEFER_E2:
            self._e2_pending, self._e2_dt = True, dt
        self._step_keep = (d, arr, keep)      # alive until the next call (the launches copy what they need)
        if push:
            self._dt_step = dt
            for ent in entries:
                ent["after"]()
        if fold:
            self._end_of_fold()

    def step_fused(self, dt, laser=None, defer_e2=False):
        """one whole step; ``laser``: optional callable(engine, dt) run at the '_laser' stage"""
        if laser is None:
            self.step_stages(dt, _lib.LPA_STAGE_E1, _lib.LPA_STAGE_E2, defer_e2)
            return
        self.step_stages(dt, _lib.LPA_STAGE_E1, _lib.LPA_STAGE_B2)
        laser(self, dt)
        self.step_stages(dt, _lib.LPA_STAGE_B2_GUARD, _lib.LPA_STAGE_E2, defer_e2)

    def run_steps(self, nsteps, dt, laser=None):
        """``nsteps`` steps back to back with nothing reading the fields in between: the second E half step of a step and
        the first one of the next are ONE sweep (same B, same J; a cell's E update reads no other cell's E: two sequential
        updates in registers = the two sweeps bit for bit) followed by one guard stage -- one field sweep, one launch and,
        between slabs, one message round less per step.  The last step is a plain one: the state after the call is that
        of ``nsteps`` ``step()``s."""
        try:
            for k in range(int(nsteps)):
                self.step(dt, laser=laser, defer_e2=k < nsteps - 1) if laser is not None else \
                    self.step(dt, defer_e2=k < nsteps - 1)
        finally:
            self._flush_e2()          # (an exception in the middle must not leave E half a step behind)
