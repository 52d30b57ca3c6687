"""One host call per step: ``lpa_step`` (csrc/lpa_step.hip) enqueues the no-callback stage sequence of a single-slab
step -- what the per-stage methods of the engines issue one ctypes call at a time (the reference walks the same stages
from Python, `simulation/simulation.py:937-1122`).  On launch-bound configs (C3: a 4 M-particle laser-target on 2 M
cells, ~20 launches of 5-40 us per step) the Python stage loop issues a step no faster than the GPU executes it; through
``lpa_step`` the host cost per launch is the HIP launch itself.

The descriptor is rebuilt for every call from the engine's CURRENT stores (a few microseconds per species): nothing that
holds a device address is cached across steps, so a sort, a re-allocation, a window shift or an upload between two steps
cannot leave a stale pointer behind.  What stays in Python: the sort (it needs the live count on the host), everything
a callback does, and slab-to-slab exchanges (``torch.distributed``) -- engines with more than one rank keep the
per-stage path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check


class FusedStepMixin:
    """needs from the engine: ``dim``, ``_grid_struct()``, ``_species_entries()`` (yielding, per species,
    (lpa_particles, tiling | None, n_sorted, push-params filler, overflow tensor, counter tensor, after-push hook)),
    ``_cpml_axes``, ``pml``, ``fused_cpml``, ``local_axes``, ``eps0``, ``absorb`` and the rho mixin"""

    fused_step = True          # single-slab steps go through lpa_step (False: the per-stage calls)
    fuse_species = True        # 3-D: every tile-ordered species in ONE launch (lpa_push_deposit_tiled_multi_3d)

    def can_fuse(self):
        return self.fused_step and self.comm.size == 1 and (self.pml is None or self.fused_cpml)

    def step_stages(self, dt, first, last):
        """enqueue stages ``first .. last`` (LPA_STAGE_*) of one step in one call.  The stores must be sorted when
        they are due BEFORE LPA_STAGE_RESET is reached (``sort_due_species``); the rho mode of the step is decided
        when LPA_STAGE_RESET is part of the range and closed when LPA_STAGE_FOLD is."""
        if not self.can_fuse():
            raise _lib.LpaError("lpa_step drives a single slab (and the fused CPML sweeps)")
        self._dt_hint = dt        # (the first sort of a store sizes its sort interval from the particles' speed)
        d = _lib.lpa_step_desc()
        d.grid = self._grid_struct()
        d.dim, d.local_axes, d.dt, d.eps0 = self.dim, self.local_axes, dt, self.eps0
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
        timed = self.kernel_events is not None and first <= _lib.LPA_STAGE_PUSH <= last
        self.kernel_events_step = False    # (fused species: one launch, one event pair -- on the first tiled species)
        entries = list(self._species_entries(dt)) if first <= _lib.LPA_STAGE_PUSH <= last else []
        arr = (_lib.lpa_step_species * max(len(entries), 1))()
        stream = torch.cuda.current_stream(self.device)
        for k, (pc, tiling, n_sorted, pp, ovf, cnt_t, _) in enumerate(entries):
            e = arr[k]
            e.p, e.n_sorted, e.pp = pc, n_sorted, pp
            e.t = C.pointer(tiling) if tiling is not None else None
            if ovf is not None:
                e.overflow, e.overflow_count = ovf.data_ptr(), cnt_t.data_ptr()
            if timed and tiling is not None and n_sorted > 0 and not (d.fuse_species and self.kernel_events_step):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)          # (creates the HIP events; lpa_step records them again around the launch)
                e1.record(stream)
                e.ev_start, e.ev_stop = e0.cuda_event, e1.cuda_event
                self.kernel_events.append((e0, e1))
                self.kernel_events_step = True
            keep.append((pc, tiling, pp, ovf, cnt_t))
        d.nspecies, d.species = len(entries), arr
        check(self.L.lpa_step(C.byref(d), first, last, stream.cuda_stream), "lpa_step")
        self._step_keep = (d, arr, keep)      # alive until the next call (the launches copy what they need)
        if entries:
            self._dt_step = dt
            for ent in entries:
                ent[6]()
        if first <= _lib.LPA_STAGE_FOLD <= last:
            self._phase = "idle"

    def step_fused(self, dt, laser=None):
        """one whole step; ``laser``: optional callable(engine, dt) run at the '_laser' stage"""
        if laser is None:
            self.step_stages(dt, _lib.LPA_STAGE_E1, _lib.LPA_STAGE_E2)
            return
        self.step_stages(dt, _lib.LPA_STAGE_E1, _lib.LPA_STAGE_B2)
        laser(self, dt)
        self.step_stages(dt, _lib.LPA_STAGE_B2_GUARD, _lib.LPA_STAGE_E2)
