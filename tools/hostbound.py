#!/usr/bin/env python3
"""is a small slab's step host bound?  Enqueue time of N steps (no sync) against the time until the GPU has finished them:
python tools/hostbound.py NX NY PPC [steps] [mirror]   (mirror: rank 0 of a mirrored 2-slab ring over the loopback transport)"""
import sys, time, json, types
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch
import bench
from lambdapic_amd.dist import SlabComm

nx, ny, ppc = (int(v) for v in sys.argv[1:4])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
mirror = len(sys.argv) > 5
args = types.SimpleNamespace(nx=nx, ny=ny, ppc=ppc, sort_interval=20, block_particles=8192)
dev = torch.device("cuda:0")
if mirror:
    from lambdapic_amd.dist import LoopbackComm
    comm = LoopbackComm(nx * bench.LAMBDA0 / 20, 2)
else:
    comm = SlabComm(None, periodic=True)
eng, dt, n = bench.build_engine(args, comm, dev)
eng.run_steps(40, dt)
torch.cuda.synchronize()
out = {}
for label, fn in (("run_steps", lambda: eng.run_steps(steps, dt)), ("step", lambda: [eng.step(dt) for _ in range(steps)])):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out[label] = {"enqueue_us_per_step": 1e6 * (t1 - t0) / steps, "total_us_per_step": 1e6 * (t2 - t0) / steps}
# the host's own cost of a step: median over single step() calls (sort steps -- a host read -- fall out of the median)
torch.cuda.synchronize()
one = []
for _ in range(60):
    t0 = time.perf_counter(); eng.step(dt); one.append(time.perf_counter() - t0)
    if len(one) % 6 == 0:
        torch.cuda.synchronize()          # (never more than a few steps ahead: no back-pressure from a full queue)
out["host_us_per_step_call_median"] = 1e6 * sorted(one)[len(one) // 2]
print(json.dumps({"nx": nx, "ny": ny, "ppc": ppc, "mirror": mirror, **out}))
