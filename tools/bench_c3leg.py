#!/usr/bin/env python3
"""bench.py's C3 leg on its own (2-D laser-target 2048 x 1024, e- + p 32 ppc each in a 1 um slab, CPML, GaussianLaser2D, moving
window): the command for kernel lists / counters of that leg.     python tools/bench_c3leg.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.extra_c3()))
