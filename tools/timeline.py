#!/usr/bin/env python3
"""per-launch timeline of the last steps of a rocprofv3 --kernel-trace CSV: kernel, duration, gap to the previous kernel's
end; then per-step totals (a step starts at each k_fdtd_e* launch pair E2 -> E1 ... simplest: split at the tiled push kernel)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
# split into steps at the tiled push kernel
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_push_deposit_tiled") or "k_push_deposit_tiled" in r["Kernel_Name"]]
if len(idx) < nsteps + 2:
    print("too few steps", len(idx)); sys.exit(1)
a, b = idx[-nsteps - 1], idx[-1]
seg = rows[a:b]
busy = collections.defaultdict(float); cnt = collections.Counter(); gap_after = collections.defaultdict(float)
prev_end = None
tot_gap = 0.0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:60]
    busy[name] += (e - s) / 1e3; cnt[name] += 1
    if prev_end is not None:
        g = max(0, s - prev_end) / 1e3
        gap_after[name] += g; tot_gap += g
    prev_end = max(prev_end or 0, e)
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
print(f"{nsteps} steps: span {span / nsteps:.1f} us/step, busy {sum(busy.values()) / nsteps:.1f} us/step, gaps {tot_gap / nsteps:.1f} us/step, launches/step {len(seg) / nsteps:.1f}")
for k in sorted(busy, key=lambda k: -busy[k] - gap_after[k]):
    print(f"  {k:60s} x{cnt[k] / nsteps:5.1f}/step  busy {busy[k] / nsteps:7.2f} us  gap-before {gap_after[k] / nsteps:7.2f} us")
if len(sys.argv) > 3:
    print("--- one step")
    a = idx[-2]; b = idx[-1]
    prev_end = None
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  {r['Kernel_Name'].split('(')[0][:50]:50s} dur {(e - s) / 1e3:7.2f}  gap {((s - prev_end) / 1e3 if prev_end else 0):7.2f}")
        prev_end = e
