#!/usr/bin/env python3
"""bench.py -- particle-updates/s of the MI355X PIC inner loop on BASELINE.json's config C2
(2-D uniform thermal plasma, 1024x1024 cells, 64 ppc, 1 species, periodic; push + deposit + FDTD +
guard handling + the periodic tile sort), synthetic inputs generated in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver as ``python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...`` (one rank per GPU, RCCL): WEAK scaling -- every rank owns a 1024x1024 slab of a
(1024*N)x1024 periodic box, x faces exchanged with the ring neighbours.

Rank 0 prints ONE JSON line: metric/value (whole-job particle-updates/s), a ``roofline`` object
for the dominant kernel (the tiled push+deposit kernel, timed live with HIP events on its stream)
and, at N = 1, a ``cpu_baseline`` object (the oracle port timed on the host cores on a bounded
sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C_LIGHT = 299792458.0
LAMBDA0 = 0.8e-6
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
BYTES_PER_PARTICLE = 105.0     # 7 f64 + is_dead read, 6 f64 written
GATHER_SCATTER_BYTES_PER_CELL = 112.0  # 6 arrays read + 4 arrays RMW


def build_engine(args, comm, device, nx=None):
    """``nx``: cells of THIS rank's slab along x (default: args.nx, the weak-scaling slab)"""
    from lambdapic_amd import constants
    from lambdapic_amd.engine import PicEngine2D

    nx, ny, ppc = (args.nx if nx is None else nx), args.ny, args.ppc
    dx = dy = LAMBDA0 / 20                                  # example/ring.py:34-40
    dt = 0.95 / (C_LIGHT * np.sqrt(dx ** -2 + dy ** -2))    # simulation.py:219
    q, m = -constants.E_CHARGE, constants.M_E
    omega = 2 * np.pi * C_LIGHT / LAMBDA0
    n_c = constants.EPSILON_0 * m * omega ** 2 / q ** 2     # tests/test_numerical_heating.py:16,86
    u_th = getattr(args, "uth", 0.0442)                     # 1 keV electrons (C2); --uth: off-benchmark sweeps
    n = nx * ny * ppc
    from lambdapic_amd import _lib
    padded = getattr(args, "order", "striped") == "padded"
    eng = PicEngine2D(nx, ny, dx, dy, n_guard=3, device=device, comm=comm,
                      sort_interval=args.sort_interval, block_particles=args.block_particles,
                      order=_lib.LPA_ORDER_PADDED if padded else _lib.LPA_ORDER_STRIPED)
    if getattr(args, "fixed_sort", False):
        eng.overflow_sort_fraction = 0
    if getattr(args, "lookahead_cold", False):
        eng.sort_lookahead_cold = True
    # (the padded order stores a few per cent of holes and rounds every tile to 64 slots)
    eng.add_species(q, m, capacity=int(1.12 * n) + 65536 if padded else n + 4096)
    s = eng.species[0].cset
    gen = torch.Generator(device=device).manual_seed(20260722 + comm.rank)
    chunk = 1 << 24
    for lo in range(0, n, chunk):                           # cell by cell, ppc per cell (patch/cpu.py:36-44)
        hi = min(lo + chunk, n)
        cell = torch.arange(lo, hi, device=device) // ppc
        r = lambda: torch.rand(hi - lo, device=device, dtype=torch.float64, generator=gen)
        g = lambda: torch.randn(hi - lo, device=device, dtype=torch.float64, generator=gen)
        s.arr("x")[lo:hi] = eng.x0 + ((cell // ny).double() + r() - 0.5) * dx
        s.arr("y")[lo:hi] = ((cell % ny).double() + r() - 0.5) * dy
        ux, uy, uz = g() * u_th + getattr(args, "drift", 0.0), g() * u_th, g() * u_th
        s.arr("ux")[lo:hi], s.arr("uy")[lo:hi], s.arr("uz")[lo:hi] = ux, uy, uz
        s.arr("inv_gamma")[lo:hi] = 1.0 / torch.sqrt(1 + ux * ux + uy * uy + uz * uz)
        s.arr("w")[lo:hi] = n_c * dx * dy / ppc
        s.id[lo:hi] = torch.arange(lo, hi, device=device) + (comm.rank << 40)
        del cell, ux, uy, uz
    eng.species[0].n = n
    return eng, dt, n


def parity_c1(nsteps=200):
    """SURVEY 8(d)'s accuracy figures on config C1 (256 x 256 cells, 16 ppc, periodic thermal plasma): the
    same seeded particles through the CPU port and through the GPU engine; relative error of the field
    energy, the kinetic energy and the total charge after ``nsteps`` steps (the CPU port is the checker)."""
    import oracle
    from oracle import driver
    from lambdapic_amd.engine import PicEngine2D
    from lambdapic_amd.patch import make_patches_2d

    nx = ny = 256
    dx = dy = LAMBDA0 / 20
    dt = 0.95 / (C_LIGHT * np.sqrt(dx ** -2 + dy ** -2))
    q, m = -oracle.E_CHARGE, oracle.M_E
    n_c = oracle.EPSILON_0 * m * (2 * np.pi * C_LIGHT / LAMBDA0) ** 2 / q ** 2
    P = make_patches_2d(nx, ny, dx, dy, 8, 8)
    driver.load_uniform_plasma(P, 0, 16, n_c, 0.0442, np.random.default_rng(20260722))
    eng = PicEngine2D(nx, ny, dx, dy, n_guard=3, device="cuda:0", sort_interval=8)
    n = sum(p.particles[0].npart for p in P)
    eng.add_species(q, m, capacity=int(1.2 * n) + 1024)
    eng.species[0].upload([p.particles[0] for p in P])
    ks = driver.oracle_kernels()
    for _ in range(nsteps):
        driver.step(P, ks, dt, [(q, m)], do_sort=False)
        eng.step(dt)
    d = eng.diagnostics()
    rel = lambda a, b: abs(a - b) / abs(b) if b else abs(a - b)
    return {"config": "C1: 256x256 cells, 16 ppc, periodic", "steps": nsteps,
            "field_energy_rel_err": rel(d["field_energy"], driver.field_energy(P)),
            "kinetic_energy_rel_err": rel(d["kinetic"][0], driver.kinetic_energy(P, 0, m)),
            "charge_rel_err": rel(d["charge"], driver.total_charge(P))}


def _reference_kernel():
    """the reference's own fused 2-D kernel (unified_boris_pusher_cpu_2d of core/pusher/unified/unified_pusher_2d.c, compiled
    in the build container by oracle/Makefile into oracle/_ref -- a binary of this repo's making that travels with the
    snapshot).  A child process tries it first on a tiny case (the binary was built on another host: an illegal instruction
    must cost the baseline's kind, not the run): ('native' | 'portable', module) or (None, None)."""
    import subprocess
    import oracle
    if not oracle.ref_available():
        return None, None
    probe = ("import numpy as np, oracle, sys\n"
             "from oracle import driver\n"
             "from lambdapic_amd.patch import make_patches_2d\n"
             "m = oracle.ref_module('pusher', 'unified_pusher_2d', portable=sys.argv[1] == 'portable')\n"
             "P = make_patches_2d(32, 32, 4e-8, 4e-8, 1, 1)\n"
             "driver.load_uniform_plasma(P, 0, 4, 1e27, 0.05, np.random.default_rng(1))\n"
             "m.unified_boris_pusher_cpu_2d([p.particles[0] for p in P], [p.fields for p in P], P.npatches, 1e-17, -1.6e-19, 9.1e-31)\n")
    for flavour in ("native", "portable"):
        try:
            r = subprocess.run([sys.executable, "-c", probe, flavour], cwd=ROOT, timeout=120, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL)
            if r.returncode == 0:
                return flavour, oracle.ref_module("pusher", "unified_pusher_2d", portable=flavour == "portable")
        except (subprocess.TimeoutExpired, OSError):
            pass
    return None, None


def cpu_baseline(args):
    """CPU path on a bounded sample of the same workload: same physics and cell size, smaller box; fused push+deposit
    (OpenMP over patches) + the four FDTD half steps + guard copies / current fold; no sort, no particle migration.
    The fused kernel -- 95 % of the CPU step -- is the REFERENCE's own compiled one when its binary (oracle/_ref) runs on
    this host (kind "reference"); the field half steps and guard copies around it are the oracle port's C (the reference's
    are numba JIT functions, which cannot run here); otherwise everything is the port (kind "port":
    oracle/picoracle.c rebuilt -O3 -march=native on this host)."""
    import oracle
    from oracle import driver
    from lambdapic_amd.patch import make_patches_2d

    # the box gives one GPU a share of the host cores: use the cores this process may run on,
    # at most 16 (the per-GPU share), and say how many
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    L = oracle.lib(native=True)
    L.orc_set_num_threads(max(1, min(avail, args.cpu_threads)))
    threads = int(L.orc_num_threads())
    nx = ny = args.cpu_cells
    ppc, npx = args.ppc, max(1, args.cpu_cells // 32)
    dx = dy = LAMBDA0 / 20
    dt = 0.95 / (C_LIGHT * np.sqrt(dx ** -2 + dy ** -2))
    q, m = -oracle.E_CHARGE, oracle.M_E
    n_c = oracle.EPSILON_0 * m * (2 * np.pi * C_LIGHT / LAMBDA0) ** 2 / q ** 2
    P = make_patches_2d(nx, ny, dx, dy, npx, npx)
    driver.load_uniform_plasma(P, 0, ppc, n_c, 0.0442, np.random.default_rng(1))
    fl, pl = [p.fields for p in P], list(P)
    parts = [p.particles[0] for p in P]
    npat, ng = P.npatches, 3
    import ctypes as C
    ftab = oracle._tab([getattr(f, a) for f in fl for a in oracle.FIELD_ORDER])

    def fdtd_e(h):
        L.orc_fdtd_e_2d_patches(C.c_long(npat), ftab, C.c_long(P.nx), C.c_long(P.ny), C.c_long(ng),
                                C.c_double(dx), C.c_double(dy), C.c_double(h), C.c_double(oracle.EPSILON_0))

    def fdtd_b(h):
        L.orc_fdtd_b_2d_patches(C.c_long(npat), ftab, C.c_long(P.nx), C.c_long(P.ny), C.c_long(ng),
                                C.c_double(dx), C.c_double(dy), C.c_double(h))

    def guards(attrs):
        oracle.sync_guard_fields_2d_c(fl, pl, attrs, npat, P.nx, P.ny, ng, native=True)

    flavour, ref = _reference_kernel()

    def one_step():
        E, B = ["ex", "ey", "ez"], ["bx", "by", "bz"]
        fdtd_e(0.5 * dt); guards(E)
        fdtd_b(0.5 * dt); guards(B)
        oracle.reset_current(fl, npat)
        if ref is not None:
            ref.unified_boris_pusher_cpu_2d(parts, fl, npat, dt, q, m)
        else:
            oracle.unified_boris_pusher_cpu_2d(parts, fl, npat, dt, q, m, native=True)
        oracle.sync_currents_2d_c(fl, pl, npat, P.nx, P.ny, ng, native=True)
        fdtd_b(0.5 * dt); guards(B)
        fdtd_e(0.5 * dt); guards(E)

    one_step()                      # warm-up (page faults, thread pool)
    n = nx * ny * ppc
    t0 = time.perf_counter()
    steps = 0
    while steps < 2 or (time.perf_counter() - t0 < args.cpu_seconds and steps < 50):
        one_step()
        steps += 1
    el = time.perf_counter() - t0
    # how the port compares with the reference's own compiled kernel: measured in the build container on the same cores
    # (python -m oracle.cpu_ratio -> profiles/r03_cpu_port_vs_reference.txt, which names the sha256 of the oracle source it
    # timed).  Refused when oracle/picoracle.c changed since.
    ratio, ratio_src = None, "profiles/r03_cpu_port_vs_reference.txt missing"
    try:
        import hashlib
        with open(os.path.join(ROOT, "profiles", "r03_cpu_port_vs_reference.txt")) as fh:
            txt = fh.read()
        with open(os.path.join(ROOT, "oracle", "picoracle.c"), "rb") as fh:
            sha = hashlib.sha256(fh.read()).hexdigest()[:16]
        rec = [ln.split("=")[-1].strip() for ln in txt.splitlines() if ln.startswith("oracle/picoracle.c sha256")]
        if rec and rec[0] == sha:
            ratio = float(txt.strip().splitlines()[-1].split("=")[-1])
            ratio_src = (f"recorded in the build container (python -m oracle.cpu_ratio -> "
                         f"profiles/r03_cpu_port_vs_reference.txt @ oracle/picoracle.c {sha}: both kernels on the same cores)")
        else:
            ratio_src = "recorded ratio is for another oracle/picoracle.c: re-run python -m oracle.cpu_ratio"
    except (OSError, ValueError) as e:
        ratio_src = f"no recorded ratio ({e.__class__.__name__})"
    kernel = ("the reference's unified_boris_pusher_cpu_2d (oracle/_ref, " +
              ("-march=native of the build host" if flavour == "native" else "-march=x86-64-v3 build") +
              "; field half steps / guard copies: oracle/picoracle.c)") if ref is not None else \
        "oracle/picoracle.c -O3 -march=native"
    return {"value": n * steps / el, "unit": "particle-updates/s", "cores": threads,
            "kind": "reference" if ref is not None else "port",
            "port_over_reference_same_cores": ratio, "port_over_reference_source": ratio_src,
            "parity_c1": parity_c1(),
            "sample": f"{nx}x{ny} cells, {ppc} ppc ({n} particles, {npat} patches of 32x32), {steps} steps "
                      f"of push+deposit+FDTD+guard sync (no sort / migration), {kernel}, {threads} OpenMP threads"}


def source_hash():
    """sha256 of the K1 sources: a recorded profile (traffic, port / reference ratio) is only quoted for the code it
    was measured on"""
    import hashlib
    h = hashlib.sha256()
    for f in ("lambdapic_amd/csrc/lpa_particles.hip", "lambdapic_amd/csrc/lpa_common.hpp"):
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def recorded_traffic(args, kernel_prefix):
    """HBM bytes of one K1 launch from the committed rocprofv3 --pmc passes (tools/prof_pmc.sh ->
    tools/make_traffic_json.py): counters cannot be read from inside this process.  Refused (None + the reason)
    when the file was recorded for another kernel, another workload or other kernel sources."""
    path = os.path.join(ROOT, "profiles", "r04_k1_traffic.json")
    try:
        with open(path) as fh:
            tr = json.load(fh)
    except Exception as e:   # noqa: BLE001
        return None, f"no traffic profile ({e.__class__.__name__})"
    if tr.get("config") != {"nx": args.nx, "ny": args.ny, "ppc": args.ppc}:
        return None, "traffic profile is for another workload"
    if not str(tr.get("kernel", "")).startswith(kernel_prefix):
        return None, "traffic profile is for another kernel"
    if tr.get("source_sha256_16") != source_hash():
        return None, "traffic profile predates the current kernel sources: re-run tools/prof_pmc.sh"
    return tr["traffic_bytes_per_launch"], f"recorded: profiles/r04_k1_traffic.json @ sources {tr['source_sha256_16']}"


class StageTimer:
    """HIP events around the engine's own entry points, grouped into stages (timed on the stream they run on)"""

    def __init__(self, eng, groups):
        self.ev = {g: [] for g in groups}
        for g, names in groups.items():
            for name in names:
                if hasattr(eng, name):
                    setattr(eng, name, self._wrap(getattr(eng, name), g, eng.device))

    def _wrap(self, fn, g, device):
        def timed(*a, **k):
            st = torch.cuda.current_stream(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            r = fn(*a, **k)
            e1.record(st)
            self.ev[g].append((e0, e1))
            return r
        return timed

    def reset(self):
        for v in self.ev.values():
            v.clear()

    def ms_per_step(self, steps):
        return {g: float(sum(a.elapsed_time(b) for a, b in v)) / steps for g, v in self.ev.items()}


STAGES_2D = {"fdtd_cpml": ["update_efield", "update_bfield", "laser_inject"], "guards": ["sync_guard_fields"],
             "push_deposit": ["push_deposit"], "sort": ["sort"], "fold_migrate": ["sync_currents", "sync_particles"],
             "reset": ["reset_current"], "window": ["shift_window"]}


def extra_c3(steps=400, warm=160):
    """BASELINE config C3 (2-D laser-target, `example/laser-target.py:28-66`): 2048 x 1024 cells at lambda / 50,
    1 um slab of e- + p at 32 ppc each, CPML on all sides, GaussianLaser2D a0 = 10, tile sort + a moving window
    that shifts inside the timed region -- through the Simulation stage loop, one GPU."""
    from lambdapic_amd import constants
    from lambdapic_amd.laser import GaussianLaser2D
    from lambdapic_amd.simulation import MovingWindow, Simulation, Species
    nx, ny, ppc = 2048, 1024, 32
    dx = dy = LAMBDA0 / 50
    nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C_LIGHT / LAMBDA0) ** 2 / constants.E_CHARGE ** 2
    sim = Simulation(nx, ny, dx, dy, npatch_x=nx // 64, npatch_y=ny // 64, random_seed=1, sort_interval=20)
    Lx = nx * dx
    dens = lambda x, y: np.where((x > Lx / 2) & (x < Lx / 2 + 1e-6), 10 * nc, 0.0)
    sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=ppc, momentum_sigma=0.01))
    sim.add_species(Species("p", charge=1, mass=1836.0, density=dens, ppc=ppc))
    sim.initialize()
    eng = sim.engine
    cbs = [GaussianLaser2D(a0=10.0, l0=LAMBDA0, w0=2e-6, ctau=2e-6, x0=4e-6),
           # the window starts inside the warm-up, so that its first shift (one-time work: the x layers are dropped,
           # torch modules load) is not timed; the timed region sees steady-state shifts (one per 64 cells at c)
           MovingWindow(velocity=C_LIGHT, start_time=40 * sim.dt)]
    sim.run(warm, callbacks=cbs)
    # timed region: the stage loop as a user runs it -- one lpa_step call per step (no callback between the stages
    # once the window has removed the laser's layer), only the K1 launches carry HIP events
    eng.kernel_events = []
    eng.reserve_kernel_events(2 * steps + 8)     # (no event is created inside the timed region; no more than needed: live timer events slow K1)
    torch.cuda.synchronize()
    n0 = sum(eng.diagnostics()["nalive"])
    t0 = time.perf_counter()
    sim.run(steps, callbacks=cbs)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n1 = sum(eng.diagnostics()["nalive"])
    alive = 0.5 * (n0 + n1)
    k1_ms = float(sum(a.elapsed_time(b) for a, b in eng.kernel_events)) / steps
    ms = 1e3 * el / steps
    shifts, rho_steps = int(getattr(sim, "window_shifts", 0)), dict(eng.rho_steps)
    # stage breakdown: a second, shorter pass walked stage by stage through the facades with HIP events around every
    # engine call (the events and the per-stage host calls cost ~0.15 ms per step themselves: not in the timed region)
    eng.kernel_events = None
    eng.fused_step = False
    timer = StageTimer(eng, STAGES_2D)
    staged_steps = 120
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sim.run(staged_steps, callbacks=cbs)
    torch.cuda.synchronize()
    staged_ms = 1e3 * (time.perf_counter() - t1) / staged_steps
    stage = timer.ms_per_step(staged_steps)
    t = eng.cpml_thickness
    psi_cells = 2 * t * (nx + ny)                       # layer cells (the x layers leave when the window starts)
    b_fields = 336.0 * nx * ny + 4 * 32.0 * psi_cells   # FDTD 336 B/cell/step + psi_a, psi_b RMW per half step
    b_k1 = BYTES_PER_PARTICLE * alive + GATHER_SCATTER_BYTES_PER_CELL * nx * ny
    dom = max(stage, key=stage.get)
    dom_bytes = {"fdtd_cpml": b_fields, "push_deposit": b_k1}.get(dom)
    return {"workload": "C3: 2-D laser-target 2048x1024 cells (lambda/50), e- + p 32 ppc each in a 1 um slab, CPML, "
                        "GaussianLaser2D a0=10, tile sort every 20 steps, moving window at c (Simulation stage loop)",
            "value": alive * steps / el, "unit": "particle-updates/s", "ms_per_step": ms, "steps": steps,
            "alive": int(alive), "window_shifts": shifts,
            "rho": eng.rho_mode(), "rho_steps": rho_steps,
            "host_calls_per_step": "1 (lpa_step)",
            "stage_ms_per_step": {k: round(v, 4) for k, v in stage.items()},
            "stage_sum_ms": round(sum(stage.values()), 4),
            "stage_pass": f"separate pass of {staged_steps} steps walked stage by stage with HIP events around every "
                          f"engine call: {staged_ms:.3f} ms/step with that instrumentation",
            "roofline": {"bound": "hbm", "scope": "step", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "algorithmic_bytes_per_step": b_fields + b_k1,
                         "achieved": (b_fields + b_k1) / (ms * 1e-3) / 1e9,
                         "frac": (b_fields + b_k1) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "dominant_stage": dom,
                         "dominant_stage_frac": (dom_bytes / (stage[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS) if dom_bytes else None,
                         "k1_kernel_ms": k1_ms,
                         "note": "4 M particles on 2.1 M cells: the step is field / launch bound, not particle bound"}}


def extra_c5(steps=40, warm=12):
    """one GPU's slab of BASELINE config C5 (3-D laser-target, `example/laser-target-3d.py:26-60`): 64 x 256 x 256
    cells, e- + p at 8 ppc each for x > 1 um, CPML on six faces, GaussianLaser3D -- Simulation3D stage loop"""
    from lambdapic_amd import constants
    from lambdapic_amd.laser import GaussianLaser3D
    from lambdapic_amd.simulation3d import Simulation3D, Species
    nx, ny, nz, ppc = 64, 256, 256, 8
    dx, dy, dz = LAMBDA0 / 20, LAMBDA0 / 10, LAMBDA0 / 10
    nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C_LIGHT / LAMBDA0) ** 2 / constants.E_CHARGE ** 2
    sim = Simulation3D(nx, ny, nz, dx, dy, dz, npatch_x=nx // 32, npatch_y=ny // 64, npatch_z=nz // 64,
                       random_seed=1, sort_interval=10)
    dens = lambda x, y, z: np.where(x > 1e-6, nc, 0.0)
    sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=ppc, momentum_sigma=0.01))
    sim.add_species(Species("p", charge=1, mass=1836.0, density=dens, ppc=ppc))
    sim.initialize()
    eng = sim.engine
    cbs = [GaussianLaser3D(a0=10.0, l0=LAMBDA0, w0=2e-6, ctau=3e-6, x0=6e-6)]
    sim.run(warm, callbacks=cbs)
    eng.kernel_events = []
    eng.reserve_kernel_events(2 * steps + 8)     # (no event is created inside the timed region; no more than needed: live timer events slow K1)
    torch.cuda.synchronize()
    n0 = sum(eng.diagnostics()["nalive"])
    t0 = time.perf_counter()
    sim.run(steps, callbacks=cbs)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n1 = sum(eng.diagnostics()["nalive"])
    alive = 0.5 * (n0 + n1)
    k_ms = float(sum(a.elapsed_time(b) for a, b in eng.kernel_events)) / steps     # both species' launches
    ms = 1e3 * el / steps
    b_k1 = 121.0 * alive                                  # 3-D particle-update
    return {"workload": "C5 slab: one GPU's 64x256x256 cells of the 3-D laser-target (dx=lambda/20, dy=dz=lambda/10), "
                        "e- + p 8 ppc each for x > 1 um, CPML on 6 faces, GaussianLaser3D a0=10, tile sort every 10 "
                        "steps (Simulation3D stage loop)",
            "value": alive * steps / el, "unit": "particle-updates/s", "ms_per_step": ms, "steps": steps,
            "alive": int(alive), "rho": eng.rho_mode(), "rho_steps": dict(eng.rho_steps),
            "species_launches": "e- and p in ONE K1-3D launch per step (lpa_push_deposit_tiled_multi_3d)"
                                if eng.fuse_species else "one K1-3D launch per species",
            "roofline": {"bound": "hbm", "kernel": "k_push_deposit_tiled_3d", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch_set": b_k1,
                         "achieved": b_k1 / (k_ms * 1e-3) / 1e9, "frac": b_k1 / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         **recorded_traffic_3d()}}


def recorded_traffic_3d():
    """HBM bytes per algorithmic byte of the K1-3D launch of THIS leg (e- + p in one launch) from the committed counter
    passes (tools/prof_pmc_c5.sh on tools/bench_c5leg.py -> profiles/r04_k13d_c5_traffic.json); refused for other sources"""
    import hashlib
    try:
        with open(os.path.join(ROOT, "profiles", "r04_k13d_c5_traffic.json")) as fh:
            tr = json.load(fh)
        h = hashlib.sha256()
        for f in ("lambdapic_amd/csrc/lpa_particles3d.hip", "lambdapic_amd/csrc/lpa_common.hpp"):
            with open(os.path.join(ROOT, f), "rb") as fh:
                h.update(fh.read())
        if tr.get("source_sha256_16") != h.hexdigest()[:16]:
            return {"traffic_per_algorithmic_byte": None, "traffic_source": "3-D traffic profile predates the kernel sources"}
        return {"traffic_per_algorithmic_byte": tr["traffic_per_algorithmic_byte"],
                "traffic_source": f"recorded: profiles/r04_k13d_c5_traffic.json (this leg, tools/prof_pmc_c5.sh) @ sources {tr['source_sha256_16']}"}
    except Exception as e:   # noqa: BLE001
        return {"traffic_per_algorithmic_byte": None, "traffic_source": f"no 3-D traffic profile ({e.__class__.__name__})"}


# ---- N > 1: the configs BASELINE.json defines on several GPUs, fixed-size problems cut into N slabs (strong scaling) ----
def _allsum(vals):
    """sum of a list of python numbers over the ranks (control plane: the gloo default group)"""
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in vals], dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
    return t.tolist()


def _allmax(v):
    import torch.distributed as dist
    t = torch.tensor([float(v)], dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _gather(v):
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [v]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, v)
    return out


def _timed(comm, device, step, nsteps, batch=None):
    """EXACTLY ``nsteps`` steps -- ``nsteps`` calls of ``step()``, or one call of ``batch(nsteps)`` -- between barrier +
    synchronize on both sides; max over the ranks"""
    torch.cuda.synchronize(device)
    comm.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    if batch is not None:
        batch(nsteps)
    for _ in range(nsteps if batch is None else 0):
        step()
    torch.cuda.synchronize(device)
    comm.barrier()
    torch.cuda.synchronize(device)
    return _allmax(time.perf_counter() - t0)


def _k1_roofline(eng, nsteps, bytes_local, kernel):
    """roofline object of a slab leg: the ranks' algorithmic bytes over the SLOWEST rank's K1 time (edge + interior
    launches add up), against N x the per-GPU peak"""
    ev = eng.kernel_events or []
    k_ms = float(sum(a.elapsed_time(b) for a, b in ev)) / nsteps if ev else float("nan")
    k_all = _gather(k_ms)
    tot_bytes = _allsum([bytes_local])[0]
    worst = max(k_all)
    n = len(k_all)
    ach = tot_bytes / (worst * 1e-3) / 1e9 if worst == worst and worst > 0 else float("nan")
    return {"bound": "hbm", "kernel": kernel, "kernel_ms_per_rank": [round(v, 4) for v in k_all],
            "algorithmic_bytes_per_step_all_ranks": tot_bytes, "achieved": ach, "peak": HBM_PEAK_GBS * n,
            "unit": "GB/s", "frac": ach / (HBM_PEAK_GBS * n)}


def _live_2d(eng):
    """(sum q w, sum |q| w, count) of this rank's live particles"""
    qw = qa = 0.0
    n = 0
    for sp in eng.species:
        s_ = sp.cset
        ok = ~torch.isnan(s_.arr("x")[: sp.n])
        w = s_.arr("w")[: sp.n][ok].sum().item()
        qw, qa, n = qw + sp.q * w, qa + abs(sp.q) * w, n + int(ok.sum().item())
    return qw, qa, n


def _live_3d(eng):
    qw = qa = 0.0
    n = 0
    for sp in eng.species:
        d = sp["data"][:, : sp["n"]]
        ok = ~torch.isnan(d[0])
        w = d[7][ok].sum().item()
        qw, qa, n = qw + sp["q"] * w, qa + abs(sp["q"]) * w, n + int(ok.sum().item())
    return qw, qa, n


def _charge_check(sim, live, rho_sum, cell_volume, cbs, shifts=lambda: 0):
    """one more step, bracketed: the charge of the padded rho arrays of all ranks (x guard planes between slabs are
    zero after the fold: counted once) equals sum q w of the particles that were alive when the step deposited, and
    the live count can only have dropped by what stood next to an absorbing bound.  A step in which the window
    shifted is skipped (it drops / injects at stage 'start').  Returns (rel. charge error, particles absorbed)"""
    for _ in range(4):
        s0 = shifts()
        qw, qa, n0 = _allsum(list(live(sim.engine)))
        sim.run(1, callbacks=cbs)
        if shifts() != s0:
            continue
        charge = _allsum([rho_sum(sim.engine) * cell_volume])[0]
        n1 = _allsum([live(sim.engine)[2]])[0]
        return abs(charge - qw) / qa, int(n0 - n1)
    return float("nan"), -1


def leg_c2_strong(args, comm, device, steps, warm):
    """config C2's 1024 x 1024 box at a FIXED size, cut into N x-slabs (strong scaling), next to the weak headline"""
    nx_loc = 1024 // comm.size
    if nx_loc * comm.size != 1024 or nx_loc % 8:
        return {"workload": "C2 strong", "value": None, "error": f"1024 cells do not split into {comm.size} tile-aligned slabs"}
    a = argparse.Namespace(**vars(args))
    a.ny, a.ppc = 1024, 64
    eng, dt, n_local = build_engine(a, comm, device, nx=nx_loc)
    eng.rho_continuity = args.rho == "continuity"
    for _ in range(warm):
        eng.step(dt)
    eng.kernel_events = []
    eng.reserve_kernel_events(2 * steps + 8)     # (no event is created inside the timed region; no more than needed: live timer events slow K1)
    el = _timed(comm, device, None, steps, batch=lambda n: eng.run_steps(n, dt))
    d = eng.diagnostics(reduce=True)
    w = float(eng.species[0].cset.arr("w")[0].item())
    n_tot = n_local * comm.size
    from lambdapic_amd import constants
    charge_err = abs(d["charge"] / (n_tot * w * -constants.E_CHARGE) - 1)
    assert d["nalive"][0] == n_tot and charge_err < 1e-10, (d["nalive"], n_tot, charge_err)
    out = {"workload": f"C2 strong: 2-D uniform thermal plasma 1024x1024 cells, 64 ppc, periodic, as {comm.size} x-slabs "
                       f"of {nx_loc}x1024 (fixed problem size)",
           "scaling": "strong", "value": n_tot * steps / el, "unit": "particle-updates/s", "ms_per_step": 1e3 * el / steps,
           "steps": steps, "alive_per_rank": _gather(eng.diagnostics()["nalive"][0]), "charge_rel_err": charge_err,
           "rho": eng.rho_mode(), "overlap": bool(eng.overlap), "particle_message_window": [eng.migrate_window, eng.migrate_capacity],
           "one_call_step": eng.one_call_step(), "e_half_steps": "merged across the step boundary (engine.run_steps)",
           "roofline": _k1_roofline(eng, steps, BYTES_PER_PARTICLE * n_local + GATHER_SCATTER_BYTES_PER_CELL * nx_loc * 1024,
                                    "k_push_deposit_tiled_2d")}
    del eng
    return out


def leg_c4(args, comm, device, steps, warm, make_comm):
    """BASELINE config C4 (2-D LWFA, `example/lwfa.py:30-76`): 4096 x 512 cells, 16 ppc, ne = 0.01 nc for x > 1 um with
    1 um vacuum margins in y, CPML on all sides, SimpleLaser2D a0 = 2, moving window at c with injection -- as N slabs
    of (4096 / N) x 512 through the Simulation stage loop (chain: the end ranks own the x layers)"""
    from lambdapic_amd import constants
    from lambdapic_amd.dist import SlabComm
    from lambdapic_amd.laser import SimpleLaser2D
    from lambdapic_amd.simulation import MovingWindow, Simulation, Species
    nx, ny, ppc = 4096, 512, 16
    dx = dy = LAMBDA0 / 20
    nxl = nx // comm.size
    if nxl * comm.size != nx or nxl % 64:
        return {"workload": "C4", "value": None, "error": f"4096 cells do not split into {comm.size} slabs of 64-cell patches"}
    nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C_LIGHT / LAMBDA0) ** 2 / constants.E_CHARGE ** 2
    chain = make_comm(periodic=False)
    sim = Simulation(nx, ny, dx, dy, npatch_x=nxl // 64, npatch_y=ny // 64, random_seed=1, sort_interval=20, comm=chain,
                     device=str(device))
    Ly = ny * dy
    dens = lambda x, y: np.where((x > 1e-6) & (y > 1e-6) & (y < Ly - 1e-6), 0.01 * nc, 0.0)
    sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=ppc))
    sim.initialize()
    eng = sim.engine
    eng.rho_continuity = args.rho == "continuity"
    ledger = {"dropped": 0, "injected": 0, "shifts": 0}
    shift, append = eng.shift_window, eng.append_particles_device

    def shift_window(ncells):
        before = _live_2d(eng)[2]
        shift(ncells)
        ledger["dropped"] += before - _live_2d(eng)[2]      # (what a slab hands to its left neighbour arrives there:
        ledger["shifts"] += 1                                # the sum over the ranks is what left the chain)

    def append_particles_device(ispec, dev):       # (only the injection of fresh columns comes through here)
        ledger["injected"] += int(dev["x"].numel())
        append(ispec, dev)

    eng.shift_window, eng.append_particles_device = shift_window, append_particles_device
    cbs = [SimpleLaser2D(a0=2.0, w0=5e-6, ctau=5e-6, l0=LAMBDA0), MovingWindow(velocity=C_LIGHT, start_time=0.03 * sim.Lx / C_LIGHT)]
    if getattr(args, "stall_rank", -1) == comm.rank:      # rehearsal of the watchdog: this rank never joins the next exchange
        time.sleep(1e6)
    sim.run(5, callbacks=cbs)          # the cells loaded inside the x-max layer are absorbed by the first step
    n_init = _allsum([_live_2d(eng)[2]])[0]
    ledger.update(dropped=0, injected=0)
    sim.run(warm, callbacks=cbs)
    eng.kernel_events = []
    eng.reserve_kernel_events(2 * steps + 8)     # (no event is created inside the timed region; no more than needed: live timer events slow K1)
    a0 = _allsum([_live_2d(eng)[2]])[0]
    el = _timed(comm, device, lambda: sim.run(1, callbacks=cbs), steps)
    a1 = _allsum([_live_2d(eng)[2]])[0]
    alive = 0.5 * (a0 + a1)
    roof = _k1_roofline(eng, steps, BYTES_PER_PARTICLE * _live_2d(eng)[2] + GATHER_SCATTER_BYTES_PER_CELL * nxl * ny,
                        "k_push_deposit_tiled_2d")
    eng.kernel_events = None
    g = eng.grid
    err, absorbed = _charge_check(sim, _live_2d, lambda e: e.grid.view("rho").sum().item(), g.dx * g.dy, cbs,
                                  lambda: ledger["shifts"])
    shifts = _gather(ledger["shifts"])
    dropped, injected = _allsum([ledger["dropped"], ledger["injected"]])
    n_end = _allsum([_live_2d(eng)[2]])[0]
    # nothing doubled, nothing from nowhere: the live count follows the ledger up to what the open low-x edge absorbs
    ok = n_end <= n_init - dropped + injected and n_end >= n_init - dropped + injected - 0.01 * n_init
    assert ok and err <= 1e-10, (n_init, dropped, injected, n_end, err)
    return _closing(chain, {"workload": f"C4: 2-D LWFA 4096x512 cells (lambda/20), e- 16 ppc at 0.01 nc, CPML, SimpleLaser2D a0=2, moving "
                        f"window at c with injection, tile sort every 20 steps, as {comm.size} x-slabs of {nxl}x512 "
                        f"(Simulation stage loop)",
            "scaling": "strong", "value": alive * steps / el, "unit": "particle-updates/s", "ms_per_step": 1e3 * el / steps,
            "steps": steps, "alive": int(alive), "alive_per_rank": _gather(_live_2d(eng)[2]), "window_shifts": shifts[0],
            "ledger": {"initial": int(n_init), "dropped": int(dropped), "injected": int(injected), "final": int(n_end)},
            "charge_rel_err": err, "absorbed_in_checked_step": absorbed, "rho": eng.rho_mode(),
            "overlap": bool(eng.overlap), "particle_message_window": [eng.migrate_window, eng.migrate_capacity], "one_call_step": eng.one_call_step(), "roofline": roof})


def _closing(chain, result):
    """a leg's own communicator is destroyed when the leg is done (the device work has been synchronised by then)"""
    torch.cuda.synchronize()
    chain.close()
    return result


def leg_c5(args, comm, device, steps, warm, make_comm):
    """BASELINE config C5 (3-D laser-target, `example/laser-target-3d.py:26-60`): 512 x 256 x 256 cells, e- + p at 8 ppc
    each for x > 1 um, CPML on six faces, GaussianLaser3D a0 = 10 -- as N slabs of (512 / N) x 256 x 256 through the
    Simulation3D stage loop, the J / rho guard planes travelling behind the interior tiles (overlap on)"""
    from lambdapic_amd import constants
    from lambdapic_amd.dist import SlabComm
    from lambdapic_amd.laser import GaussianLaser3D
    from lambdapic_amd.simulation3d import Simulation3D, Species
    nx, ny, nz, ppc = 512, 256, 256, 8
    dx, dy, dz = LAMBDA0 / 20, LAMBDA0 / 10, LAMBDA0 / 10
    nxl = nx // comm.size
    if nxl * comm.size != nx or nxl % 32:
        return {"workload": "C5", "value": None, "error": f"512 cells do not split into {comm.size} slabs of 32-cell patches"}
    nc = constants.EPSILON_0 * constants.M_E * (2 * np.pi * C_LIGHT / LAMBDA0) ** 2 / constants.E_CHARGE ** 2
    chain = make_comm(periodic=False)
    sim = Simulation3D(nx, ny, nz, dx, dy, dz, npatch_x=nxl // 32, npatch_y=ny // 64, npatch_z=nz // 64, random_seed=1,
                       sort_interval=10, comm=chain, device=str(device))
    dens = lambda x, y, z: np.where(x > 1e-6, nc, 0.0)
    sim.add_species(Species("e", charge=-1, mass=1, density=dens, ppc=ppc, momentum_sigma=0.01))
    sim.add_species(Species("p", charge=1, mass=1836.0, density=dens, ppc=ppc))
    sim.initialize()
    eng = sim.engine
    eng.overlap = True
    eng.rho_continuity = args.rho == "continuity"
    cbs = [GaussianLaser3D(a0=10.0, l0=LAMBDA0, w0=2e-6, ctau=3e-6, x0=6e-6)]
    sim.run(warm, callbacks=cbs)
    eng.kernel_events = []
    eng.reserve_kernel_events(2 * steps + 8)     # (no event is created inside the timed region; no more than needed: live timer events slow K1)
    a0 = _allsum([_live_3d(eng)[2]])[0]
    el = _timed(comm, device, lambda: sim.run(1, callbacks=cbs), steps)
    a1 = _allsum([_live_3d(eng)[2]])[0]
    alive = 0.5 * (a0 + a1)
    roof = _k1_roofline(eng, steps, 121.0 * _live_3d(eng)[2], "k_push_deposit_tiled_3d")
    eng.kernel_events = None
    err, absorbed = _charge_check(sim, _live_3d, lambda e: e.view("rho").sum().item(), dx * dy * dz, cbs)
    assert err <= 1e-10 and 0 <= absorbed <= 0.01 * alive, (err, absorbed)
    return _closing(chain, {"workload": f"C5: 3-D laser-target 512x256x256 cells (dx=lambda/20, dy=dz=lambda/10), e- + p 8 ppc each for "
                        f"x > 1 um, CPML on 6 faces, GaussianLaser3D a0=10, tile sort every 10 steps, as {comm.size} "
                        f"x-slabs of {nxl}x256x256 (Simulation3D stage loop)",
            "scaling": "strong", "value": alive * steps / el, "unit": "particle-updates/s", "ms_per_step": 1e3 * el / steps,
            "steps": steps, "alive": int(alive), "alive_per_rank": _gather(_live_3d(eng)[2]), "charge_rel_err": err,
            "absorbed_in_checked_step": absorbed, "rho": eng.rho_mode(), "overlap": bool(eng.overlap), "particle_message_window": [eng.migrate_window, eng.migrate_capacity],
            "one_call_step": eng.one_call_step(), "roofline": roof})


class Watchdog:
    """bounded waits for the multi-rank run: a collective (or a first contact with RCCL) that hangs cannot be cancelled from
    inside the process, so when a section overruns its budget every rank that notices says so on stderr, rank 0 prints the
    JSON line it has so far (the headline survives a hanging leg; the leg is reported as {"error": "timeout"}) and the
    process exits non-zero -- the launcher then takes the other ranks down.  Never a re-exec."""

    def __init__(self, rank):
        import threading
        self.rank, self.deadline, self.label, self.partial = rank, None, "", None
        self._lock = threading.Lock()
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def arm(self, label, seconds):
        with self._lock:
            self.label, self.deadline = label, time.monotonic() + seconds

    def disarm(self):
        with self._lock:
            self.deadline = None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self._lock:
                late = self.deadline is not None and time.monotonic() > self.deadline
                label = self.label
            if late:
                print(f"[bench] rank {self.rank}: '{label}' exceeded its time budget -- giving up", file=sys.stderr, flush=True)
                if self.rank == 0 and self.partial is not None:
                    out = dict(self.partial)
                    out.setdefault("extra", []).append({"workload": label, "value": None, "error": "timeout"})
                    print(json.dumps(out), flush=True)
                if self.rank != 0:
                    time.sleep(3.0)      # (rank 0's own watchdog gets to print its line before the launcher tears the job down)
                os._exit(3)


def preflight_native():
    """child process of one rank (``bench.py --preflight``): can the library's own RCCL transport serve this job?  Creates the
    communicator and runs one ring exchange, nothing else; the parent waits with a timeout, so a first contact that hangs
    costs a fallback, not the run.  Exit code 0 = yes."""
    import torch.distributed as dist
    from lambdapic_amd.dist import SlabComm
    local_rank = 0 if os.environ.get("LPA_BENCH_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    dist.init_process_group("gloo")
    comm = SlabComm(None).attach_rccl()
    t = [torch.full((1024,), float(comm.rank), dtype=torch.float64, device=device) for _ in range(4)]
    for _ in range(3):
        comm.exchange(t[0], t[1], t[2], t[3])
    torch.cuda.synchronize(device)
    ok = t[2][0].item() == comm.left and t[3][0].item() == comm.right
    dist.barrier()
    comm.close()
    if ok:
        try:
            preflight_engines(lambda width: SlabComm(None, periodic=True).attach_rccl(), device)
        except Exception as e:      # noqa: BLE001 -- whatever it is, the run falls back
            print(f"pre-flight engines: {e!r}", file=sys.stderr, flush=True)
            ok = False
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 5)


def preflight_engines(make_ring, device):
    """the second half of the pre-flight: a few steps of a small 2-D and a small 3-D slab ring through the native transport
    in every form the run will use -- two message rounds per step with B at home, the particle-message window, in line and
    overlapped (both rounds on the communicator's second stream) -- with the bookkeeping checked over the ranks.  Whatever a
    first contact with RCCL does to one of them, it does it here, in a child with a bounded wait."""
    import types
    from lambdapic_amd import constants
    from lambdapic_amd.engine3d import PicEngine3D
    import torch.distributed as dist
    ranks = lambda c: c.size if dist.is_initialized() else 1      # (the self test plays a ring of two with one process)
    comm = make_ring(64 * LAMBDA0 / 20)      # (slab width: what a loopback ring needs to know)
    args = types.SimpleNamespace(nx=64, ny=64, ppc=8, sort_interval=5, block_particles=1024, uth=0.2)
    eng, dt, n = build_engine(args, comm, device)
    eng.MIGRATE_WINDOW_MIN = 256
    for overlap in (False, True, False):
        eng.overlap = overlap
        eng.run_steps(7, dt)
        for _ in range(3):
            eng.step(dt)
    d = eng.diagnostics(reduce=True)
    if d["nalive"][0] != n * ranks(comm) or not np.isfinite(d["field_energy"]) or not eng.one_call_step():
        raise RuntimeError(f"2-D ring: {d['nalive'][0]} of {n * ranks(comm)} particles, field energy {d['field_energy']}")
    if not 256 <= eng.migrate_window < eng.migrate_capacity:
        raise RuntimeError(f"2-D ring: message window {eng.migrate_window}")
    torch.cuda.synchronize(device)
    comm.close()
    del eng
    comm = make_ring(32 * LAMBDA0 / 20)
    n3, d3 = (32, 16, 32), (LAMBDA0 / 20, LAMBDA0 / 10, LAMBDA0 / 10)
    dt = 0.95 / (C_LIGHT * np.sqrt(sum(v ** -2 for v in d3)))
    eng = PicEngine3D(*n3, *d3, 3, sort_interval=4, comm=comm, migrate_capacity=16384)
    gen = torch.Generator(device=device).manual_seed(77 + comm.rank)
    n = n3[0] * n3[1] * n3[2] * 4
    cell = torch.arange(n, device=device) // 4
    r = lambda: torch.rand(n, device=device, dtype=torch.float64, generator=gen)
    data = torch.full((8, 2 * n + eng.arrival_area() + 1024), float("nan"), dtype=torch.float64, device=device)
    data[0, :n] = eng.x0 + ((cell // (n3[1] * n3[2])).double() + r() - 0.5) * d3[0]
    data[1, :n] = (((cell // n3[2]) % n3[1]).double() + r() - 0.5) * d3[1]
    data[2, :n] = ((cell % n3[2]).double() + r() - 0.5) * d3[2]
    for k in range(3):
        data[3 + k, :n] = torch.randn(n, device=device, dtype=torch.float64, generator=gen) * 0.2
    data[6, :n] = 1.0 / torch.sqrt(1 + (data[3:6, :n] ** 2).sum(0))
    data[7, :n] = 1e27 * d3[0] * d3[1] * d3[2] / 4
    eng.add_species_device(-constants.E_CHARGE, constants.M_E, data, n)
    for overlap in (True, False):
        eng.overlap = overlap
        eng.run_steps(5, dt)
        for _ in range(3):
            eng.step(dt)
    d = eng.diagnostics(reduce=True)
    if d["nalive"][0] != n * ranks(comm) or not np.isfinite(d["field_energy"]) or not eng.one_call_step():
        raise RuntimeError(f"3-D ring: {d['nalive'][0]} of {n * ranks(comm)} particles, field energy {d['field_energy']}")
    torch.cuda.synchronize(device)
    comm.close()


def native_preflight_ok(timeout=200):
    """run ``preflight_native`` in a child of THIS rank, on a rendezvous of its own (MASTER_PORT + 1); True when it exited 0
    in time.  Called before this process touches the GPU."""
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 1)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--preflight"], env=env, timeout=timeout,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        if r.returncode != 0:
            tail = r.stderr.decode(errors="replace").strip().splitlines()[-3:]
            print(f"[bench] rank {os.environ.get('RANK')}: native RCCL pre-flight failed (exit {r.returncode}): {' | '.join(tail)}",
                  file=sys.stderr, flush=True)
        return r.returncode == 0
    except subprocess.TimeoutExpired:
        print(f"[bench] rank {os.environ.get('RANK')}: native RCCL pre-flight timed out", file=sys.stderr, flush=True)
        return False


def main():
    if "--preflight" in sys.argv:
        preflight_native()
    if "--preflight-selftest" in sys.argv:
        # the engine half of the pre-flight on ONE GPU: a ring of two identical slabs over a one-rank RCCL communicator
        from lambdapic_amd.dist import LoopbackComm
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        preflight_engines(lambda width: LoopbackComm(width, 2, rccl=True), dev)
        print("pre-flight engines ok")
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--nx", type=int, default=1024)
    ap.add_argument("--ny", type=int, default=1024)
    ap.add_argument("--ppc", type=int, default=64)
    ap.add_argument("--sort-interval", type=int, default=20)
    ap.add_argument("--block-particles", type=int, default=8192)
    ap.add_argument("--cpu-cells", type=int, default=512)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the C3 / C5-slab legs (N = 1) / the C2-strong, C4 "
                                                            "and C5 slab legs (N > 1)")
    ap.add_argument("--leg-steps", type=int, default=0, help="timed steps of the N > 1 legs (default: 200 / 200 / 20)")
    ap.add_argument("--legs", default="c2s,c4,c5", help="which N > 1 legs to run")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the multi-rank path with ranks sharing GPUs (buffers staged "
                         "through the host); the driver's runs use nccl (RCCL)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: all ranks use cuda:0")
    ap.add_argument("--transport", default="auto", choices=["auto", "native", "torch"],
                    help="N > 1 with --backend nccl: 'native' = the library's own RCCL communicator (whole steps, exchanges "
                         "included, in one lpa_step call); 'torch' = torch.distributed's nccl group, faces moved from Python "
                         "between lpa_step sub-ranges; 'auto' = native when its pre-flight (a child process, bounded wait) "
                         "succeeds on every rank, else torch, else host-staged gloo")
    ap.add_argument("--no-overlap-pass", action="store_true", help="N > 1: skip the second, overlapped pass of the headline")
    ap.add_argument("--force-overlap-pass", action="store_true", help="rehearsals: run that pass without the native transport too")
    ap.add_argument("--stall-rank", type=int, default=-1, help="rehearsal of the watchdog: this rank stops inside the C4 leg")
    ap.add_argument("--leg-timeout", type=float, default=420.0, help="time budget of one N > 1 leg in seconds")
    ap.add_argument("--no-defer", action="store_true", help="A/B: deposit cell-crossers' tail cells inline")
    ap.add_argument("--order", default="striped", choices=["striped", "padded"],
                    help="padded = LPA_ORDER_PADDED store + cooperative deposit")
    ap.add_argument("--reseat", action="store_true", help="A/B: with the in-kernel cell-index sort (off by default)")
    ap.add_argument("--lookahead-cold", action="store_true", help="experiment: bin every sort half an interval ahead")
    ap.add_argument("--fixed-sort", action="store_true", help="off-benchmark: sort on the fixed interval only "
                                                              "(engine.overflow_sort_fraction = 0)")
    ap.add_argument("--drift", type=float, default=0.0, help="mean u_x of the plasma (off-benchmark: a relativistic flow)")
    ap.add_argument("--uth", type=float, default=0.0442, help="thermal momentum spread per axis (C2: 0.0442 = 1 keV); "
                                                              "other values: off-benchmark sweeps (tools/sweep_uth2d.sh)")
    ap.add_argument("--inv-gamma", default="recomputed", choices=["recomputed", "streamed"],
                    help="1 / gamma of a particle: recomputed from its momenta by the fused kernels (default; two of the "
                         "thirteen attribute streams go) or loaded and stored like the reference's kernel")
    ap.add_argument("--per-step", action="store_true",
                    help="time K engine.step() calls instead of engine.run_steps(K) (the default): run_steps does the second E "
                         "half step of a step and the first one of the next in ONE sweep (same B, same J, cell-local update: "
                         "the two sweeps bit for bit) with one guard stage -- same state after K steps, one field sweep, one "
                         "launch and, between slabs, one message round less per step")
    ap.add_argument("--run-steps", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--rho", default="continuity", choices=["continuity", "deposited"],
                    help="rho between two sorts: advanced with the discrete continuity equation (default; the fused "
                         "kernel skips its rho atomics, a real deposit re-anchors rho on every sort step) or deposited "
                         "in every step like the reference's kernel")
    args = ap.parse_args()

    import torch.distributed as dist
    from lambdapic_amd.dist import SlabComm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: lambdapic_amd has no CPU path")
    # first contact with the library's own RCCL transport happens in a CHILD process with a bounded wait, before this
    # process touches the GPU: a hang there costs the fallback, not the run
    native_ok = False
    if world > 1 and args.backend == "nccl" and args.transport in ("auto", "native"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            os.environ["LPA_BENCH_SHARE_GPU"] = "1"       # (rehearsal: RCCL refuses two ranks on one GPU -> the fallback runs)
        native_ok = native_preflight_ok()
    device = torch.device("cuda:0" if args.share_gpu else f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    rank = int(os.environ.get("RANK", "0"))
    dog = Watchdog(rank)
    p2p, comm_note, native = None, "single rank", False
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane (barrier, max of the elapsed time) on gloo; the halo / migration messages of the step travel
        # device-to-device: through the library's own RCCL communicator (native), else through torch's nccl group, else
        # staged through the host on gloo -- the line says which
        dog.arm("process group", 300)
        dist.init_process_group("gloo")
        flag = torch.tensor([1 if native_ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        native = int(flag.item()) == 1
        comm_note = "gloo (host-staged faces)"
        if native:
            comm_note = "native rccl (lpa_comm: ncclSend / ncclRecv groups inside lpa_step), gloo control"
        elif args.backend == "nccl" and args.transport != "native":
            ok = 1
            try:
                import datetime
                # (no device_id: with a gloo default group there is no parent communicator to split from;
                # the communicators are created by the first exchange below, on the device set above)
                p2p = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=300))
                one = torch.ones(1, dtype=torch.float64, device=device)
                dist.all_reduce(one, group=p2p)          # creates the group-wide communicator collectively
                if int(one.item()) != world:
                    raise RuntimeError("RCCL all-reduce returned a wrong sum")
                probe = SlabComm(None, p2p_group=p2p)
                t = [torch.full((16,), float(probe.rank), dtype=torch.float64, device=device) for _ in range(4)]
                probe.exchange(t[0], t[1], t[2], t[3])
                torch.cuda.synchronize(device)
                if t[2][0].item() != probe.left or t[3][0].item() != probe.right:
                    ok = 0
            except Exception as e:   # noqa: BLE001 -- any RCCL failure: fall back, visibly
                print(f"[bench] rank {os.environ.get('RANK')}: RCCL face exchange unavailable ({e!r})", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                comm_note = "rccl p2p (torch batch_isend_irecv between lpa_step sub-ranges), gloo control"
            else:
                p2p = None
        dog.disarm()

    def make_comm(periodic=True):
        """the communicator of one engine: ring (periodic) or chain; with the native transport every communicator is an RCCL
        communicator of its own (rank 0 makes the id, gloo broadcasts it)"""
        c = SlabComm(None, periodic=periodic, p2p_group=p2p)
        if native and c.size > 1:
            c.attach_rccl()
        return c

    dog.arm("communicator", 300)
    comm = make_comm()
    dog.disarm()
    rccl_version = comm.native_info()[5] if comm.native is not None else None
    assert comm.size == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N"

    if args.order == "padded" or args.reseat:
        # the cooperative deposit / the in-kernel re-seating live in the variants build only (A/B use)
        from lambdapic_amd import _lib
        with _lib.use_variants():
            eng, dt, n_local = build_engine(args, comm, device)
    else:
        eng, dt, n_local = build_engine(args, comm, device)
    eng.defer_crossers = not args.no_defer
    eng.reseat = args.reseat
    eng.rho_continuity = args.rho == "continuity"
    eng.lazy_inv_gamma = args.inv_gamma == "recomputed"
    if comm.size > 1:
        # the headline is measured with every exchange IN LINE on the step's stream first -- the path with the fewest moving
        # parts -- and printed at once; the overlapped form (both message rounds on the communicator's second stream beside
        # the interior tiles) is measured right after it as a second pass of exactly --steps steps and replaces the headline
        # only if it ran and was faster.  A first contact with RCCL from two streams cannot take the headline along.
        eng.overlap = False
    dog.arm("headline (weak-scaled C2)", 600)
    for _ in range(args.warmup):
        eng.step(dt)
    # timed region: EXACTLY --steps steps between barrier + synchronize on both sides
    eng.kernel_events = []
    eng.reserve_kernel_events(2 * args.steps + 8)     # (no event is created inside the timed region; no more than needed: live timer events slow K1)
    torch.cuda.synchronize(device)
    comm.barrier()
    torch.cuda.synchronize(device)
    run_steps = not args.per_step
    t0 = time.perf_counter()
    if run_steps:
        eng.run_steps(args.steps, dt)
    else:
        for _ in range(args.steps):
            eng.step(dt)
    torch.cuda.synchronize(device)
    comm.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: tiled push+deposit, HIP events recorded on its stream inside the timed region
    ev = eng.kernel_events
    # (on a slab ring the launch is split into an edge and an interior part: their times add up)
    k_ms = float(np.sum([a.elapsed_time(b) for a, b in ev])) / args.steps if ev else float("nan")
    d = eng.diagnostics()
    alive = d["nalive"][0]
    alg_bytes = BYTES_PER_PARTICLE * n_local + GATHER_SCATTER_BYTES_PER_CELL * args.nx * args.ny
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if ev else float("nan")
    traffic, traffic_source = recorded_traffic(args, "k_push_deposit_tiled_2d")
    n_total = n_local * comm.size
    out = {
        "metric": "particle-updates/sec", "value": n_total * args.steps / elapsed,
        "unit": "particle-updates/s", "n_gpus": comm.size, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C2: 2-D uniform thermal plasma {args.nx}x{args.ny} cells per GPU, "
                               f"{args.ppc} ppc, 1 species (e-), periodic, dt_cfl 0.95, push+deposit+FDTD"
                               f"+guards, tile sort every {args.sort_interval} steps",
                   "particles_per_gpu": n_local, "alive_rank0": alive,
                   "decomposition": f"{comm.size} x-slabs" if comm.size > 1 else "single slab",
                   "comm": comm_note, "world": world, "rccl_version": rccl_version,
                   "message_rounds_per_step": (2 if eng.local_b() else 4) if comm.size > 1 else 0,
                   "particle_message_window": [eng.migrate_window, eng.migrate_capacity] if comm.size > 1 else None,
                   "overlap": bool(eng.overlap),
                   "e_half_steps": "engine.run_steps: the second E half step of a step and the first of the next are one "
                                   "sweep (two sequential updates per cell) + one guard stage; the last step is a plain one"
                                   if run_steps else "two sweeps, two guard stages per step (engine.step)",
                   "part_eb_writeback": False,
                   # rho between two sorts (lambdapic_amd/rho.py): "continuity" = advanced from the folded currents,
                   # re-anchored by a real deposit on every sort step; "deposited" = the reference's kernel
                   "rho": eng.rho_mode(), "rho_steps": dict(eng.rho_steps),
                   # algorithmic bytes stay SURVEY 8(d)'s 105 B per particle-update (the reference's data contract); with
                   # LPA_PUSH_NO_IG the kernel itself streams 89 of them (inv_gamma neither loaded nor stored)
                   "inv_gamma": "recomputed from the momenta in the fused kernels (LPA_PUSH_NO_IG): 89 of the 105 "
                                "algorithmic bytes are streamed" if eng._noig() else "streamed",
                   "dead_particles": "x = NaN (the resident store has no is_dead array; the 105 B of SURVEY 8(d) "
                                     "count one byte for it)"},
        "roofline": {"bound": "hbm", "kernel": "k_push_deposit_tiled_2d", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes, "traffic": traffic,
                     "traffic_source": traffic_source,
                     # context, not the roofline: what a plain streaming kernel with K1's own access pattern reaches
                     # (tools/ubench/stream_soa.hip, recorded in profiles/r03_k1_streams.txt; boxes: 5.1-6.1 TB/s)
                     "counted_traffic_over_plain_stream_6070GBps":
                         (traffic / (k_ms * 1e-3) / 1e9 / 6070.0) if (traffic and ev) else None},
    }
    dog.disarm()
    if comm.rank == 0 and comm.size > 1:
        # the headline, at once: a leg that hangs below cannot take it along (the full line follows at the end)
        print(json.dumps(out), flush=True)
        dog.partial = out
    want_overlap_pass = comm.size > 1 and not args.no_overlap_pass and (comm.native is not None or args.force_overlap_pass)

    def overlapped_pass():
        """the headline's second pass: the same K steps with both message rounds behind the interior tiles (engine.overlap).
        Runs LAST: whatever a first contact with RCCL from two streams does, the in-line headline and the legs are out"""
        in_line_ms = out["ms_per_step"]
        passes = {"in_line_ms_per_step": in_line_ms, "overlapped_ms_per_step": None}
        out["config"]["passes"] = passes
        if comm.rank == 0:
            dog.partial = out
        dog.arm("headline, overlapped pass", 300)
        try:
            eng.overlap = True
            eng.kernel_events = None
            for _ in range(max(2, args.warmup // 2)):
                eng.step(dt)
            torch.cuda.synchronize(device)
            comm.barrier()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            if run_steps:
                eng.run_steps(args.steps, dt)
            else:
                for _ in range(args.steps):
                    eng.step(dt)
            torch.cuda.synchronize(device)
            comm.barrier()
            torch.cuda.synchronize(device)
            el2 = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([el2], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el2 = float(t.item())
            passes["overlapped_ms_per_step"] = 1e3 * el2 / args.steps
            if el2 < elapsed:
                out["value"], out["ms_per_step"] = n_total * args.steps / el2, 1e3 * el2 / args.steps
                out["config"]["overlap"] = True
                out["roofline"]["note"] = "kernel time, achieved and frac are those of the in-line pass (one K1 launch per step)"
        except Exception as e:   # noqa: BLE001 -- the in-line headline stands
            passes["overlapped_error"] = repr(e)
        dog.disarm()
        if comm.rank == 0:
            print(json.dumps(out), flush=True)
            dog.partial = out
    if comm.size > 1 and not args.no_extra:
        # the configs BASELINE.json defines on several GPUs are FIXED-size problems: C2's 1024^2 box (north_star quotes
        # ">= 6x at 8 GPUs" on it), C4 (4096 x 512 LWFA with window) and C5 (512 x 256 x 256 laser-target) cut into N
        # slabs -- strong scaling, reported next to the weak-scaled headline.  Every rank runs them; each leg asserts
        # its bookkeeping (live counts against the ledger, total charge of the padded arrays) before it reports.
        keep_engine = want_overlap_pass
        if not keep_engine:
            del eng
        torch.cuda.empty_cache()
        legs = []
        want = set(args.legs.split(","))
        todo = [("c2s", lambda: leg_c2_strong(args, comm, device, args.leg_steps or 200, 10)),
                ("c4", lambda: leg_c4(args, comm, device, args.leg_steps or 200, 40, make_comm)),
                ("c5", lambda: leg_c5(args, comm, device, args.leg_steps or 20, 12, make_comm))]
        for name, leg in todo:
            if name not in want:
                continue
            dog.arm(f"leg {name}", args.leg_timeout)       # (a stuck rank: rank 0 prints what it has and everybody exits != 0)
            try:
                r = leg()
            except Exception as e:   # noqa: BLE001 -- reported; the other ranks run the same code and fail alike
                r = {"workload": name, "value": None, "error": repr(e)}
            dog.disarm()
            r.setdefault("comm", comm_note)
            legs.append(r)
            if comm.rank == 0:
                dog.partial = dict(out, extra=list(legs))
            torch.cuda.empty_cache()
        out["extra"] = legs
    if want_overlap_pass:
        overlapped_pass()
    if comm.rank == 0 and comm.size == 1 and not args.no_extra:
        # north_star: "uniform-plasma and laser-target configs": the other single-GPU configs, bounded legs
        del eng
        torch.cuda.empty_cache()
        out["extra"] = []
        for leg in (extra_c3, extra_c5):
            try:
                out["extra"].append(leg())
            except Exception as e:   # noqa: BLE001 -- the headline number stands on its own
                out["extra"].append({"workload": leg.__name__, "value": None, "error": repr(e)})
            torch.cuda.empty_cache()
    if comm.rank == 0 and comm.size == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args)
        except Exception as e:  # the GPU number stands on its own
            out["cpu_baseline"] = {"value": None, "unit": "particle-updates/s", "cores": 0, "kind": "port",
                                   "sample": f"failed: {e!r}"}
    if comm.rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
