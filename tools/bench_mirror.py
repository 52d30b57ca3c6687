#!/usr/bin/env python3
"""Compute-side cost of the slab (multi-rank) code path without a network: ONE process plays rank 0 of a
periodic 2-slab ring whose other slab is an exact translated copy of itself, so what a neighbour would send
through a face is what this rank sends through the opposite one (particle messages are translated by one
slab width first).  Everything the N > 1 path does on the device runs -- edge / interior split of K1, halo and
current pack / unpack, leaver scan, migration unpack, arrival area -- only the wire is replaced by a device
copy.  Compare ms/step with bench.py at N = 1: the difference is what weak scaling loses before any
communication cost.  (A tool: not part of the product path.)"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_mirror_comm import MirrorComm


ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=1024); ap.add_argument("--ny", type=int, default=1024)
ap.add_argument("--ppc", type=int, default=64); ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--warmup", type=int, default=8); ap.add_argument("--sort-interval", type=int, default=20)
ap.add_argument("--block-particles", type=int, default=8192)
ap.add_argument("--overlap", action="store_true", help="edge tiles + J exchange on a second stream beside the interior")
ap.add_argument("--transport", default="loopback", choices=["loopback", "rccl", "python"],
                help="loopback: the library's own transport with the wire replaced by one copy kernel per round (whole step = "
                     "one lpa_step call); rccl: the same through a one-rank RCCL communicator sending to itself (real "
                     "ncclSend / ncclRecv groups); python: the faces moved from Python between lpa_step sub-ranges")
ap.add_argument("--run-steps", action="store_true", help="engine.run_steps: E guards once per step (deferred E2 guards)")
ap.add_argument("--uth", type=float, default=0.0442, help="thermal momentum spread (C2: 0.0442 = 1 keV; hotter: more leavers per step)")
ap.add_argument("--b-messages", action="store_true", help="the B guard planes travel (local_b_guards off): four message rounds per step")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dx = bench.LAMBDA0 / 20
from lambdapic_amd.dist import LoopbackComm
comm = MirrorComm(a.nx * dx, 32768) if a.transport == "python" else LoopbackComm(a.nx * dx, 2, rccl=a.transport == "rccl")
eng, dt, n = bench.build_engine(a, comm, dev)
assert eng.migrate_capacity == 32768
eng.overlap = a.overlap
eng.local_b_guards = not a.b_messages
for _ in range(a.warmup):
    eng.step(dt)
eng.kernel_events = []
eng.reserve_kernel_events(2 * a.steps + 8)
torch.cuda.synchronize(); t0 = time.perf_counter()
if a.run_steps:
    eng.run_steps(a.steps, dt)
else:
    for _ in range(a.steps):
        eng.step(dt)
torch.cuda.synchronize(); el = time.perf_counter() - t0
k_ms = float(np.sum([x.elapsed_time(y) for x, y in eng.kernel_events])) / a.steps
sp0 = eng.species[0]
ws0 = eng._sort_ws(sp0)
print("counters [overflow, arrival cursor, overflow edge, surplus, leavers]:", ws0["counters"].tolist()[:5], "steps since sort", sp0.steps_since_sort,
      "fs" , None if ws0.get("fs") is None else (ws0["fs"].edge_cols, int(ws0["fs_count"].sum().item()), int(ws0["fs_count"].max().item())), file=sys.stderr)
d = eng.diagnostics()
w = float(eng.species[0].cset.arr("w")[0].item())
print(json.dumps({"what": f"rank 0 of a mirrored 2-slab ring, transport {a.transport}" + (", B messages" if a.b_messages else "") + (", overlapped" if a.overlap else "") +
                          (", run_steps" if a.run_steps else ""), "ms_per_step": 1e3 * el / a.steps,
                  "particle_updates_per_s_per_gpu": n * a.steps / el, "k1_edge_plus_interior_ms": k_ms,
                  "alive": d["nalive"][0], "particles": n, "message_window": eng.migrate_window,
                  "charge_rel_err": abs(d["charge"] / (d["nalive"][0] * w * -1.602176634e-19) - 1)}))
