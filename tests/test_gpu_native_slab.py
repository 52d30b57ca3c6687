"""The slab path with the library's OWN transport (csrc/lpa_comm.hip) -- whole steps, exchanges included, enqueued by one
``lpa_step`` call -- on one GPU: a slab that is rank 0 of a periodic 2-slab ring whose other slab is its translated copy
(``LoopbackComm``) must reproduce one half of a single-slab run of the doubled periodic box.  Every kernel of the N > 1
path runs (E planes straight from the arrays, B advanced on the x guard planes by the sweeps themselves -- or, with
``local_b_guards`` off, exchanged like E --, J / rho fold, leaver pack, arrival unpack with free slots, arrival area, the jx
plane of the continuity update formed from what travels with J -- or riding with the B planes); the wire is a copy kernel
(``loopback``), a one-rank RCCL
communicator sending to itself (``rccl``: real ncclSend / ncclRecv groups) or Python-side copies between ``lpa_step``
sub-ranges (``python``: the path torch.distributed transports take).  Tolerance 1e-10: summation order only."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CL = 299792458.0
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _comm(transport, width, cap):
    from lambdapic_amd.dist import LoopbackComm
    if transport == "python":
        from bench_mirror_comm import MirrorComm
        return MirrorComm(width, cap)
    return LoopbackComm(width, 2, rccl=transport == "rccl")


# ---- the transport itself ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rccl", [False, True])
def test_exchange_pairs_high_face_with_low_face(rccl):
    from lambdapic_amd.dist import LoopbackComm
    comm = LoopbackComm(1.0, 2, rccl=rccl)
    kind, rank, size, left, right, version = comm.native_info()
    assert (rank, left >= 0, right >= 0) == (0, True, True)
    assert (version > 20000) == rccl
    dev = torch.device("cuda:0")
    mk = lambda n, v: torch.full((n,), float(v), dtype=torch.float64, device=dev)
    # three messages in one round, different sizes per direction (the window shift sends a big block one way only)
    sets = [(mk(7, 1), mk(5, 2), mk(5, -1), mk(7, -1)), (mk(1000, 3), mk(1000, 4), mk(1000, -1), mk(1000, -1)),
            (mk(1, 5), mk(33, 6), mk(33, -1), mk(1, -1))]
    comm.exchange_many(sets)
    torch.cuda.synchronize()
    for s_lo, s_hi, r_lo, r_hi in sets:
        assert torch.equal(r_lo, s_hi) and torch.equal(r_hi, s_lo)
    a, b, c, d = mk(9, 7), mk(9, 8), mk(9, 0), mk(9, 0)
    comm.exchange(a, b, c, d)
    torch.cuda.synchronize()
    assert torch.equal(c, b) and torch.equal(d, a)
    with pytest.raises(ValueError):
        comm.exchange(a.float(), b, c, d)
    comm.close()


def test_loopback_refuses_counts_that_do_not_pair():
    from lambdapic_amd import _lib
    from lambdapic_amd.dist import LoopbackComm
    comm = LoopbackComm(1.0, 2)
    t = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for n in (4, 5, 6, 4)]
    with pytest.raises(_lib.LpaError, match="counts differ"):
        comm.exchange(*t)


# ---- 2-D: a mirrored slab == one half of the doubled box -----------------------------------------------------------------
NX, NY, PPC, NSTEPS = 64, 64, 6, 24


def _problem2d():
    lam = 0.8e-6
    dx = dy = lam / 20
    dt = 0.95 / (CL * np.sqrt(dx ** -2 + dy ** -2))
    rng = np.random.default_rng(5)
    n = NX * NY * PPC
    cell = np.arange(n) // PPC
    x = ((cell // NY) + rng.uniform(-0.5, 0.5, n)) * dx
    y = ((cell % NY) + rng.uniform(-0.5, 0.5, n)) * dy
    u = rng.normal(size=(3, n)) * 0.3          # hot: plenty of slab crossings within a few steps
    w = np.full(n, 1.7e27 * dx * dy / PPC)
    return dx, dy, dt, x, y, u, w


def _engine2d(nx_cells, comm, copies, run_steps=False, rho=True, overlap=False, local_b=True, ypml=False):
    from lambdapic_amd.engine import PicEngine2D
    dx, dy, dt, x, y, u, w = _problem2d()
    bc = {"xmin": "periodic", "xmax": "periodic", "ymin": "pml", "ymax": "pml"} if ypml else None
    eng = PicEngine2D(nx_cells, NY, dx, dy, device="cuda:0", comm=comm, sort_interval=5, block_particles=1024,
                      migrate_capacity=4096, boundary_conditions=bc, cpml_thickness=6)
    eng.rho_continuity = rho
    eng.overlap = overlap
    eng.local_b_guards = local_b
    if ypml:
        # a field blob per slab copy that reaches the y layers within the run (their psi recursions run on the slab AND,
        # for B, on its x guard planes)
        ii, jj = np.meshgrid(np.arange(NX) - NX / 2, np.arange(NY) - NY / 2, indexing="ij")
        blob = np.tile(np.exp(-(ii ** 2 + jj ** 2) / 150.0), (nx_cells // NX, 1))
        for a, amp in (("ez", 2e11), ("ey", 1e11), ("by", 500.0), ("bz", -300.0)):
            eng.grid.view(a)[3:3 + nx_cells, 3:3 + NY] = torch.from_numpy(amp * blob).cuda()
        eng.sync_guard_fields(("ex", "ey", "ez", "bx", "by", "bz"))
    n = x.size * copies
    eng.add_species(-1.602176634e-19, 9.1093837139e-31, capacity=2 * n + 20000)
    s = eng.species[0].cset
    cat = lambda a: np.concatenate([a] * copies)
    xs = np.concatenate([x + k * NX * dx for k in range(copies)])
    ig = 1 / np.sqrt(1 + (u ** 2).sum(0))
    for name, arr in (("x", xs), ("y", cat(y)), ("ux", cat(u[0])), ("uy", cat(u[1])), ("uz", cat(u[2])),
                      ("inv_gamma", cat(ig)), ("w", cat(w))):
        s.arr(name)[:n] = torch.from_numpy(arr).cuda()
    s.id[:n] = torch.arange(n, device="cuda:0")
    eng.species[0].n = n
    trace = []
    if run_steps:
        for _ in range(NSTEPS // 4):
            eng.run_steps(4, dt)
            d = eng.diagnostics()
            trace.append([d["field_energy"], d["charge"], d["kinetic"][0], d["nalive"][0]])
    else:
        for _ in range(NSTEPS):
            eng.step(dt)
            d = eng.diagnostics()
            trace.append([d["field_energy"], d["charge"], d["kinetic"][0], d["nalive"][0]])
    g = eng.grid
    fields = {a: g.view(a)[:, 3:3 + NY].cpu().numpy() for a in ("ex", "ey", "ez", "bx", "by", "bz", "rho", "jx")}
    return np.array(trace), fields, eng


@pytest.fixture(scope="module")
def doubled2d():
    tr, f, _ = _engine2d(2 * NX, None, 2)
    return tr, f


def _close(a, b, tol=1e-10):
    scale = max(np.max(np.abs(b)), 1e-300)
    return np.max(np.abs(a - b)) / scale < tol


def _rows(eng, a, nx):
    """x rows of field ``a`` a slab must share with the doubled box: its interior; for E and B also the guard planes (the
    neighbour's edge) -- all of them, except that a slab advancing B on its guard planes itself keeps ng - 1 high ones"""
    if a in ("rho", "jx"):
        return 3, 3 + nx
    return 0, nx + 6 - (1 if a[0] == "b" and eng.local_b() else 0)


@pytest.mark.parametrize("local_b", [True, False])
@pytest.mark.parametrize("transport", ["loopback", "rccl", "python"])
def test_mirrored_slab_is_half_of_the_doubled_box_2d(doubled2d, transport, local_b):
    tr2, f2 = doubled2d
    comm = _comm(transport, NX * _problem2d()[0], 4096)
    tr, f, eng = _engine2d(NX, comm, 1, local_b=local_b)
    assert eng.one_call_step() == (transport != "python") and eng.local_b() == local_b
    assert np.array_equal(tr[:, 3] * 2, tr2[:, 3])                       # nobody lost, nobody doubled
    for k in range(3):
        assert _close(2 * tr[:, k], tr2[:, k]), (transport, k)
    for a in f:
        # interior of the slab + its x guard planes (E / B: the neighbour's edge) == the doubled box's left half
        lo, hi = _rows(eng, a, NX)
        assert _close(f[a][lo:hi], f2[a][lo:hi]), (transport, a)
    assert eng.rho_steps["continuity"] > eng.rho_steps["anchor"] > 0
    if local_b and transport != "python":
        # the jx guard plane at node -1 holds the left neighbour's folded jx at its last node (what the continuity update
        # of node 0 reads): in the mirrored ring that neighbour's last node is the doubled box's node 2 NX - 1
        assert _close(f["jx"][2, :], f2["jx"][3 + 2 * NX - 1, :])


@pytest.mark.parametrize("overlap", [False, True])
def test_per_stage_path_over_the_native_transport_2d(doubled2d, overlap):
    """what Simulation walks when a callback sits between the stages: the engine's per-stage methods (fused_step off), their
    exchanges through ``SlabComm.exchange`` -> ``lpa_comm_exchange`` (the T1 overlap: edge part + J exchange on torch's side
    stream)"""
    tr2, f2 = doubled2d
    from lambdapic_amd.engine import PicEngine2D
    comm = _comm("loopback", NX * _problem2d()[0], 4096)
    saved = PicEngine2D.fused_step
    PicEngine2D.fused_step = False
    try:
        tr, f, eng = _engine2d(NX, comm, 1, overlap=overlap)
    finally:
        PicEngine2D.fused_step = saved
    assert eng.comm.native is not None          # (fused_step was off while the engine ran: every stage a call of its own)
    assert np.array_equal(tr[:, 3] * 2, tr2[:, 3])
    for k in range(3):
        assert _close(2 * tr[:, k], tr2[:, k]), k
    for a in f:
        lo, hi = _rows(eng, a, NX)
        assert _close(f[a][lo:hi], f2[a][lo:hi]), a


@pytest.mark.parametrize("e_round", [True, False])
@pytest.mark.parametrize("transport", ["loopback", "rccl"])
def test_overlapped_mirrored_slab_2d(doubled2d, transport, e_round, monkeypatch):
    """lpa_step_slab.overlap_cols: edge tile columns, leaver pack and the whole exchange on the communicator's second stream
    beside the interior tiles (north_star: halo exchange overlapped with interior work on a second HIP stream); ``e_round``:
    the E guard planes and the rows of the B half step that read them travel there as well (both rounds hidden)"""
    from lambdapic_amd.engine import PicEngine2D
    monkeypatch.setattr(PicEngine2D, "overlap_e_round", e_round)
    tr2, f2 = doubled2d
    comm = _comm(transport, NX * _problem2d()[0], 4096)
    tr, f, eng = _engine2d(NX, comm, 1, overlap=True)
    assert eng.one_call_step() and eng.edge_columns(_problem2d()[2]) > 0
    assert np.array_equal(tr[:, 3] * 2, tr2[:, 3])
    for k in range(3):
        assert _close(2 * tr[:, k], tr2[:, k]), (transport, k)
    for a in f:
        lo, hi = _rows(eng, a, NX)
        assert _close(f[a][lo:hi], f2[a][lo:hi]), (transport, a)


def test_run_steps_equals_steps_on_a_mirrored_slab(doubled2d):
    """deferred E2 guards (one message round and one launch less per step): the state after run_steps(n) is the state after
    n step()s"""
    comm = _comm("loopback", NX * _problem2d()[0], 4096)
    tr, f, eng = _engine2d(NX, comm, 1)
    comm2 = _comm("loopback", NX * _problem2d()[0], 4096)
    tr_r, f_r, _ = _engine2d(NX, comm2, 1, run_steps=True)
    # (two runs of the same problem differ in the order of their FP64 atomics: tolerance, not bit equality)
    assert np.array_equal(tr[3::4, 3], tr_r[:, 3]) and _close(tr[3::4, :3], tr_r[:, :3])
    for a in f:
        assert _close(f[a], f_r[a]), a


def test_run_steps_equals_steps_single_slab():
    tr, f, _ = _engine2d(NX, None, 1)
    tr_r, f_r, _ = _engine2d(NX, None, 1, run_steps=True)
    assert np.array_equal(tr[3::4, 3], tr_r[:, 3]) and _close(tr[3::4, :3], tr_r[:, :3])
    for a in f:
        assert _close(f[a], f_r[a]), a


def test_mirrored_slab_with_deposited_rho(doubled2d):
    """rho deposited in every step (the reference's kernel): no jx plane travels"""
    tr2, f2 = doubled2d
    comm = _comm("loopback", NX * _problem2d()[0], 4096)
    tr, f, eng = _engine2d(NX, comm, 1, rho=False)
    for k in range(3):
        assert _close(2 * tr[:, k], tr2[:, k]), k
    assert _close(f["rho"][3:3 + NX], f2["rho"][3:3 + NX])


# ---- 3-D twin ---------------------------------------------------------------------------------------------------------------
N3 = (32, 16, 32)


def _engine3d(nx_cells, comm, copies, nsteps=12, overlap=False, local_b=True):
    from lambdapic_amd import constants
    from lambdapic_amd.engine3d import PicEngine3D
    lam = 0.8e-6
    d = (lam / 20, lam / 10, lam / 10)
    dt = 0.95 / (CL * np.sqrt(sum(v ** -2 for v in d)))
    ppc = 4
    rng = np.random.default_rng(11)
    n = N3[0] * N3[1] * N3[2] * ppc
    cell = np.arange(n) // ppc
    pos = [((cell // (N3[1] * N3[2])) + rng.uniform(-0.5, 0.5, n)) * d[0],
           (((cell // N3[2]) % N3[1]) + rng.uniform(-0.5, 0.5, n)) * d[1],
           ((cell % N3[2]) + rng.uniform(-0.5, 0.5, n)) * d[2]]
    u = rng.normal(size=(3, n)) * 0.3
    eng = PicEngine3D(nx_cells, N3[1], N3[2], *d, 3, sort_interval=5, comm=comm, migrate_capacity=8192)
    eng.overlap = overlap
    eng.local_b_guards = local_b
    ntot = n * copies
    data = torch.full((8, 2 * ntot + eng.arrival_area() + 1024), float("nan"), dtype=torch.float64, device="cuda:0")
    cat = lambda a: torch.from_numpy(np.concatenate([a] * copies)).cuda()
    data[0, :ntot] = torch.from_numpy(np.concatenate([pos[0] + k * N3[0] * d[0] for k in range(copies)])).cuda()
    data[1, :ntot], data[2, :ntot] = cat(pos[1]), cat(pos[2])
    for k in range(3):
        data[3 + k, :ntot] = cat(u[k])
    data[6, :ntot] = cat(1 / np.sqrt(1 + (u ** 2).sum(0)))
    data[7, :ntot] = 1.7e27 * d[0] * d[1] * d[2] / ppc
    eng.add_species_device(-constants.E_CHARGE, constants.M_E, data, ntot)
    trace = []
    for _ in range(nsteps):
        eng.step(dt)
        dg = eng.diagnostics()
        trace.append([dg["field_energy"], dg["charge"], dg["kinetic"][0], dg["nalive"][0]])
    fields = {a: eng.view(a)[:, 3:-3, 3:-3].cpu().numpy() for a in ("ex", "ey", "ez", "bx", "by", "bz", "rho")}
    return np.array(trace), fields, eng, d


@pytest.fixture(scope="module")
def doubled3d():
    tr, f, _, _ = _engine3d(2 * N3[0], None, 2)
    return tr, f


@pytest.mark.parametrize("transport,overlap,local_b", [("loopback", False, True), ("rccl", False, True), ("python", False, True),
                                                       ("loopback", True, True), ("rccl", True, True),
                                                       ("loopback", False, False), ("python", False, False),
                                                       ("loopback", True, False), ("loopback", True, None)])
def test_mirrored_slab_is_half_of_the_doubled_box_3d(doubled3d, transport, overlap, local_b, monkeypatch):
    tr2, f2 = doubled3d
    if local_b is None:         # (B at home, but the E round in line: LPA_STEP_E_ROUND_IN_LINE)
        from lambdapic_amd.engine3d import PicEngine3D
        monkeypatch.setattr(PicEngine3D, "overlap_e_round", False)
        local_b = True
    lam = 0.8e-6
    comm = _comm(transport, N3[0] * lam / 20, 8192)
    tr, f, eng, d = _engine3d(N3[0], comm, 1, overlap=overlap, local_b=local_b)
    assert eng.one_call_step() == (transport != "python") and eng.local_b() == local_b
    if overlap:
        assert eng.edge_columns(0.95 / (CL * np.sqrt(sum(v ** -2 for v in d)))) > 0
    assert np.array_equal(tr[:, 3] * 2, tr2[:, 3])
    for k in range(3):
        assert _close(2 * tr[:, k], tr2[:, k]), (transport, k)
    for a in f:
        lo, hi = _rows(eng, a, N3[0])
        assert _close(f[a][lo:hi], f2[a][lo:hi]), (transport, a)


# ---- run_steps with CPML layers: the doubled E half step carries the psi recursions of both half steps ---------------------
@pytest.mark.parametrize("dim", [2, 3])
def test_run_steps_equals_steps_with_cpml(dim):
    from lambdapic_amd import constants
    lam = 0.8e-6
    rng = np.random.default_rng(3)

    def build():
        if dim == 2:
            from lambdapic_amd.engine import PicEngine2D
            nx, ny, dx = 64, 64, lam / 16
            dt = 0.95 / (CL * np.sqrt(2) / dx)
            bc = {k: "pml" for k in ("xmin", "xmax", "ymin", "ymax")}
            eng = PicEngine2D(nx, ny, dx, dx, device="cuda:0", boundary_conditions=bc, cpml_thickness=6, sort_interval=4,
                              block_particles=1024)
            view = eng.grid.view
            shape = (nx, ny)
        else:
            from lambdapic_amd.engine3d import PicEngine3D
            nx, ny, dx = 32, 32, lam / 10
            dt = 0.95 / (CL * np.sqrt(3) / dx)
            bc = {k: "pml" for k in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")}
            eng = PicEngine3D(nx, ny, 32, dx, dx, dx, 3, sort_interval=4, boundary_conditions=bc, cpml_thickness=4)
            view = eng.view
            shape = (nx, ny, 32)
        g = np.random.default_rng(9)
        inner = tuple(slice(3, 3 + m) for m in shape)
        # a smooth blob of E / B that reaches the layers within the run
        grids = np.meshgrid(*[np.arange(m) - m / 2 for m in shape], indexing="ij")
        blob = np.exp(-sum(v ** 2 for v in grids) / 60.0)
        for a, amp in (("ez", 1e11), ("ey", 5e10), ("by", 300.0), ("bz", -200.0)):
            view(a)[inner] = torch.from_numpy(amp * blob * (1 + 0.1 * g.normal(size=shape))).cuda()
        n = 20000
        pos = [rng.uniform(0.3, 0.7, n) * m * dx for m in shape]
        u = rng.normal(size=(3, n)) * 0.2
        w = np.full(n, 1e27 * dx ** dim / 4)
        if dim == 2:
            eng.add_species(-constants.E_CHARGE, constants.M_E, capacity=2 * n)
            cs = eng.species[0].cset
            for name, arr in (("x", pos[0]), ("y", pos[1]), ("ux", u[0]), ("uy", u[1]), ("uz", u[2]),
                              ("inv_gamma", 1 / np.sqrt(1 + (u ** 2).sum(0))), ("w", w)):
                cs.arr(name)[:n] = torch.from_numpy(arr).cuda()
            cs.id[:n] = torch.arange(n, device="cuda:0")
            eng.species[0].n = n
        else:
            data = torch.full((8, 2 * n), float("nan"), dtype=torch.float64, device="cuda:0")
            data[:, :n] = torch.from_numpy(np.concatenate([np.stack(pos), u, (1 / np.sqrt(1 + (u ** 2).sum(0)))[None], w[None]])).cuda()
            eng.add_species_device(-constants.E_CHARGE, constants.M_E, data, n)
        return eng, dt, view

    out = []
    for batched in (False, True):
        rng = np.random.default_rng(3)
        eng, dt, view = build()
        if batched:
            for _ in range(4):
                eng.run_steps(4, dt)
        else:
            for _ in range(16):
                eng.step(dt)
        torch.cuda.synchronize()
        psi = torch.cat([torch.cat([l["psi_a"].reshape(-1), l["psi_b"].reshape(-1)]) for l in eng.pml.layers]).cpu().numpy()
        out.append(({a: view(a).cpu().numpy() for a in ("ex", "ey", "ez", "bx", "by", "bz", "rho")}, psi,
                    eng.diagnostics()))
    (fa, pa, da), (fb, pb, db) = out
    assert np.abs(pa).max() > 0 and da["nalive"] == db["nalive"]
    # (two runs of the same problem: the order of the FP64 deposit atomics differs, nothing else)
    assert _close(pb, pa, 1e-9)
    for a in fa:
        assert _close(fb[a], fa[a], 1e-9), a


# ---- the particle face message sends a window of its buffer (dist.MigrateWindowMixin) ----------------------------------------
@pytest.mark.parametrize("transport", ["loopback", "rccl", "python"])
def test_particle_message_window(doubled2d, transport, monkeypatch):
    """with a floor of 64 slots the window settles at 4 x the ~30 leavers a face sees per step (128-256 of the 4096 slots the
    buffers hold): the stride of the message changes under the running exchange and nothing else does"""
    from lambdapic_amd.engine import PicEngine2D
    tr2, f2 = doubled2d
    monkeypatch.setattr(PicEngine2D, "MIGRATE_WINDOW_MIN", 64)
    comm = _comm(transport, NX * _problem2d()[0], 4096)
    tr, f, eng = _engine2d(NX, comm, 1)
    assert 64 <= eng.migrate_window <= 512 and eng.migrate_capacity == 4096
    assert np.array_equal(tr[:, 3] * 2, tr2[:, 3])
    for k in range(3):
        assert _close(2 * tr[:, k], tr2[:, k]), (transport, k)
    for a in f:
        lo, hi = _rows(eng, a, NX)
        assert _close(f[a][lo:hi], f2[a][lo:hi]), (transport, a)


def test_an_overflowing_window_is_not_an_error(doubled2d, monkeypatch):
    """a window smaller than a step's leavers: those that do not fit wait for the next step, the surplus counter sends the
    window back to the full capacity at the next retune, nobody is lost and charge is conserved; only an overflow of the
    FULL capacity raises"""
    from lambdapic_amd import _lib
    from lambdapic_amd.engine import PicEngine2D
    tr2, f2 = doubled2d
    monkeypatch.setattr(PicEngine2D, "MIGRATE_WINDOW_MIN", 8)
    monkeypatch.setattr(PicEngine2D, "_mig_window", 16)     # (every engine starts with room for 16 of its ~30 leavers per step)
    comm = _comm("loopback", NX * _problem2d()[0], 4096)
    seen = []
    orig = PicEngine2D._mig_apply
    monkeypatch.setattr(PicEngine2D, "_mig_apply", lambda self, v: (orig(self, v), seen.append(self.migrate_window))[0])
    with pytest.warns(RuntimeWarning, match="did not fit the message window"):
        tr, f, eng = _engine2d(NX, comm, 1)
    # (the retune of the very first step has seen no message yet; the next one finds the surplus counter set)
    assert seen[0] < 64 and 4096 in seen[1:3] and 64 <= seen[-1] < 4096, seen
    assert np.array_equal(tr[:, 3] * 2, tr2[:, 3])          # nobody lost, nobody doubled
    # (a leaver kept waiting for several steps drifts beyond the guard planes its deposit can reach: 3e-6 of the charge here,
    # with room for 8-16 of 30 leavers per step over five steps -- the price of an overflow that used to stop the run)
    assert _close(2 * tr[:, 1], tr2[:, 1], 1e-4)            # total charge
    # the full capacity too small: the error the engines always raised
    monkeypatch.setattr(PicEngine2D, "adaptive_migrate_window", False)
    from lambdapic_amd.dist import LoopbackComm
    dx, dy, dt, x, y, u, w = _problem2d()
    small = PicEngine2D(NX, NY, dx, dy, device="cuda:0", comm=LoopbackComm(NX * dx, 2), sort_interval=5, block_particles=1024,
                        migrate_capacity=8)
    n = x.size
    small.add_species(-1.602176634e-19, 9.1093837139e-31, capacity=2 * n + 20000)
    s = small.species[0].cset
    ig = 1 / np.sqrt(1 + (u ** 2).sum(0))
    for name, arr in (("x", x), ("y", y), ("ux", u[0]), ("uy", u[1]), ("uz", u[2]), ("inv_gamma", ig), ("w", w)):
        s.arr(name)[:n] = torch.from_numpy(arr).cuda()
    s.id[:n] = torch.arange(n, device="cuda:0")
    small.species[0].n = n
    with pytest.raises(_lib.LpaError, match="migration message overflow"):
        for _ in range(12):
            small.step(dt)
        small.diagnostics()


# ---- CPML layers on the local axis of a slab ring: the y layers' B psi runs on the x guard planes too -------------------------
@pytest.mark.parametrize("transport,local_b,overlap", [("loopback", True, False), ("rccl", True, False), ("python", True, False),
                                                       ("loopback", False, False), ("loopback", True, True), ("rccl", True, True)])
def test_mirrored_slab_with_y_layers(transport, local_b, overlap):
    from lambdapic_amd.engine import psi_rows
    tr2, f2, eng2 = _engine2d(2 * NX, None, 2, ypml=True)
    comm = _comm(transport, NX * _problem2d()[0], 4096)
    tr, f, eng = _engine2d(NX, comm, 1, local_b=local_b, ypml=True, overlap=overlap)
    assert eng.local_b() == local_b and eng.pml is not None and len(eng.pml.layers) == 4
    assert np.array_equal(tr[:, 3] * 2, tr2[:, 3]) and tr[-1, 3] < tr[0, 3]       # (the y layers absorb: some are gone)
    for k in range(3):
        assert _close(2 * tr[:, k], tr2[:, k], 1e-9), (transport, k)
    for a in f:
        lo, hi = _rows(eng, a, NX)
        assert _close(f[a][lo:hi], f2[a][lo:hi], 1e-9), (transport, a)
    # psi of the y layers: the slab's rows == the doubled box's left half; with B at home also the x guard rows of the
    # B layers (low: all ng, high: ng - 1) == the neighbouring rows of the doubled box
    for ly, ly2 in zip(eng.pml.layers, eng2.pml.layers):
        for k in ("psi_a", "psi_b"):
            a, b = psi_rows(ly, k, guards=True).cpu().numpy(), psi_rows(ly2, k, guards=True).cpu().numpy()
            assert np.abs(b).max() > 0
            assert _close(a[3:3 + NX], b[3:3 + NX], 1e-9), (ly["key"], k)
            if local_b and not ly["e"]:
                assert _close(a[3 + NX:3 + NX + 2], b[3 + NX:3 + NX + 2], 1e-9) and _close(a[0:3], b[2 * NX:2 * NX + 3], 1e-9)


def test_bench_preflight_engines_on_one_gpu():
    """``bench.py --gpus N`` tries the native transport in a child process first: a ring exchange, then a few steps of a small
    2-D and a small 3-D slab ring in every form the run uses (in line, overlapped, window retunes).  The engine half of that
    child, on a ring of two identical slabs over a one-rank RCCL communicator"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--preflight-selftest"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "pre-flight engines ok" in r.stdout, r.stderr[-2000:]


def test_2d_engine_chooses_the_overlap_by_the_edge_fraction():
    """``PicEngine2D.overlap`` left at None: on behind a real RCCL communicator when the edge tile columns are at most a tenth
    of the slab, off on narrow slabs and without a wire"""
    from lambdapic_amd.dist import LoopbackComm
    from lambdapic_amd.engine import PicEngine2D
    dx, dy, dt = _problem2d()[:3]
    seen = {}
    for nx, rccl in ((512, True), (128, True), (512, False)):
        eng = PicEngine2D(nx, 64, dx, dy, device="cuda:0", comm=LoopbackComm(nx * dx, 2, rccl=rccl), sort_interval=5)
        eng.add_species(-1.602176634e-19, 9.1093837139e-31, capacity=4096)
        assert eng.overlap is False             # (no step yet: no dt to size the edge columns from)
        eng._dt_hint = dt
        seen[(nx, rccl)] = (eng.overlap, eng.edge_columns(dt))
        eng.overlap = True
        assert eng.overlap is True
        eng.overlap = None
        eng.comm.close()
    assert seen[(512, True)][0] is True and seen[(128, True)][0] is False and seen[(512, False)][0] is False, seen
