#!/usr/bin/env python3
"""which Python lines of the package launch small torch kernels / copies inside a C5-slab step (or a C3 step: --c3)?
torch.profiler with stacks over a few steps after the warm-up, aggregated by the innermost lambdapic_amd frame."""
import collections, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity
c3 = "--c3" in sys.argv
# build the leg's simulation by running the leg for a few steps, then keep stepping it under the profiler
holder = {}
import lambdapic_amd.simulation3d as s3, lambdapic_amd.simulation as s2
cls = s2.Simulation if c3 else s3.Simulation3D
orig_run = cls.run
def run(self, *a, **k):
    holder["sim"], holder["cbs"] = self, k.get("callbacks")
    return orig_run(self, *a, **k)
cls.run = run
(bench.extra_c3 if c3 else bench.extra_c5)(steps=12, warm=12)
cls.run = orig_run
sim, cbs = holder["sim"], holder["cbs"]
sim.run(3, callbacks=cbs)
torch.cuda.synchronize()
N = 10
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    sim.run(N, callbacks=cbs)
    torch.cuda.synchronize()
agg = collections.Counter(); tim = collections.Counter()
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue      # top-level aten ops only
    fr = [f for f in (ev.stack or []) if "lambdapic_amd" in f or "bench.py" in f]
    key = (ev.name, fr[0].split("lambdapic_amd/")[-1] if fr else "?")
    agg[key] += 1; tim[key] += ev.cpu_time_total
print("top-level aten ops per step (count / step, host us / step), by innermost package frame:")
for key, n in sorted(agg.items(), key=lambda kv: -tim[kv[0]])[:40]:
    print(f"{n / N:6.1f}  {tim[key] / N:8.1f} us  {key[0]:28s} {key[1]}")
