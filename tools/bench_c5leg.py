#!/usr/bin/env python3
"""bench.py's C5-slab leg on its own (64 x 256 x 256 cells, e- + p at 8 ppc each, CPML x 6, GaussianLaser3D, Simulation3D
stage loop, both species in one K1-3D launch): the command the counter passes of tools/prof_pmc_c5.sh profile.
    python tools/bench_c5leg.py [steps] [warmup] [striped|column]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 12
if len(sys.argv) > 3 and sys.argv[3] == "column":
    from lambdapic_amd import _lib
    from lambdapic_amd.engine3d import PicEngine3D
    PicEngine3D.DEFAULT_ORDER = _lib.LPA_ORDER_COLUMN
print(json.dumps(bench.extra_c5(steps=steps, warm=warm)))
