#!/usr/bin/env python3
"""profiles/<round>_k1_traffic.json (LPA_ROUND, default r04) from the 'fetch' and 'write' passes of tools/prof_pmc.sh.

usage: tools/make_traffic_json.py <pmc-outdir> [2d|3d]   (run where the passes were collected, or on their merged
gpurun_out copy; 3d = the passes of tools/prof_pmc3d.sh -> profiles/r03_k13d_traffic.json).  HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both in KB; on gfx950
FETCH_SIZE counts half of a wide coalesced read, MI355X_MICROARCH.md, HBM section).  The file carries the hash of
the kernel sources it was measured on; bench.py refuses it for any other."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
d, mode = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "2d")
flt = "k_push_deposit_tiled_3d" if mode in ("3d", "c5") else "k_push_deposit_tiled_2d"
SRC = {"2d": ("lambdapic_amd/csrc/lpa_particles.hip", "lambdapic_amd/csrc/lpa_common.hpp"),
       "3d": ("lambdapic_amd/csrc/lpa_particles3d.hip", "lambdapic_amd/csrc/lpa_common.hpp"),
       "c5": ("lambdapic_amd/csrc/lpa_particles3d.hip", "lambdapic_amd/csrc/lpa_common.hpp")}[mode]
vals, name = {}, None
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            name = r["Kernel_Name"]
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
mean = lambda k: sum(vals[k]) / len(vals[k])
import hashlib
h = hashlib.sha256()
for f in SRC:
    h.update(open(os.path.join(ROOT, f), "rb").read())
out = {"kernel": name.replace("void ", "").split("(")[0].replace(", ", ","),
       "config": {"nx": 1024, "ny": 1024, "ppc": 64} if mode == "2d" else
                 {"nx": 64, "ny": 256, "nz": 256, "ppc": 8, "particles": 33554432, "algorithmic_bytes_per_launch": 121.0 * 33554432},
       "FETCH_SIZE_KB_mean": mean("FETCH_SIZE"), "WRITE_SIZE_KB_mean": mean("WRITE_SIZE"),
       "launches": len(vals["FETCH_SIZE"]),
       "launch_mix": "mean over every launch of the passes: the first push after the sort deposits rho (real deposit), "
                     "the others run with LPA_PUSH_NO_RHO (rho from the continuity equation)",
       "correction": "gfx950: FETCH_SIZE x 2 for wide coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
       "traffic_bytes_per_launch": (2 * mean("FETCH_SIZE") + mean("WRITE_SIZE")) * 1024.0,
       "source_sha256_16": h.hexdigest()[:16],
       "source": ("tools/prof_pmc.sh passes 'fetch' and 'write' (rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE / --pmc WRITE_SIZE), "
                  "bench.py --no-cpu-baseline --no-extra --steps 4 --warmup 2") if mode == "2d" else
                 "tools/prof_pmc3d.sh passes 'fetch' and 'write', tools/bench3d.py --steps 4 --warmup 2 (uniform 8 ppc slab)"}
if mode == "3d":
    out["traffic_per_algorithmic_byte"] = out["traffic_bytes_per_launch"] / out["config"]["algorithmic_bytes_per_launch"]
if mode == "c5":
    # the C5-slab leg (tools/bench_c5leg.py): e- + p in ONE launch; live particles from the leg's own JSON line
    leg = None
    for f in glob.glob(d + "/*.log"):
        for ln in open(f):
            if ln.startswith("{") and "alive" in ln:
                leg = json.loads(ln)
    alive = leg["alive"]
    out["config"] = {"workload": "C5 slab leg of bench.py (tools/bench_c5leg.py 6 12): 64x256x256 cells, e- + p 8 ppc each, "
                                 "both species in one launch", "alive": alive, "algorithmic_bytes_per_launch": 121.0 * alive}
    out["source"] = "tools/prof_pmc_c5.sh passes 'fetch' and 'write', tools/bench_c5leg.py 6 12"
    out["traffic_per_algorithmic_byte"] = out["traffic_bytes_per_launch"] / (121.0 * alive)
rnd = os.environ.get("LPA_ROUND", "r04")
name = {"2d": f"{rnd}_k1_traffic.json", "3d": f"{rnd}_k13d_traffic.json", "c5": f"{rnd}_k13d_c5_traffic.json"}[mode]
json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
print(json.dumps(out, indent=1))
