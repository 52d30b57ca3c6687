#!/usr/bin/env python3
"""C3 / C5 legs of bench.py with and without the overflow-driven early sorts (device.OverflowMonitor: its bounded host
run-ahead costs one event wait per species and step).     python tools/ab_adaptive_c3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lambdapic_amd.engine import PicEngine2D
from lambdapic_amd.engine3d import PicEngine3D
def patch(cls, frac):
    orig = cls.__init__
    def init(self, *a, **k):
        orig(self, *a, **k)
        self.overflow_sort_fraction = frac
    cls.__init__ = init
    return orig
for rep in range(2):
    for frac in (0.001, 0.0):
        o2, o3 = patch(PicEngine2D, frac), patch(PicEngine3D, frac)
        c3, c5 = bench.extra_c3(), bench.extra_c5()
        PicEngine2D.__init__, PicEngine3D.__init__ = o2, o3
        print("overflow_sort_fraction", frac, "C3 %.4f ms/step" % c3["ms_per_step"], c3["rho_steps"], "C5 leg %.4f ms/step" % c5["ms_per_step"], c5["rho_steps"], flush=True)
