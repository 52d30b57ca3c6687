#!/usr/bin/env python3
"""C3 leg of bench.py at several work-block sizes (K1 launches of 2 M particles: few blocks per CU)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lambdapic_amd import engine
for bp in (8192, 4096, 2048, 1024):
    orig = engine.PicEngine2D.__init__
    def init(self, *a, _bp=bp, **k):
        orig(self, *a, **k)
        self.block_particles = _bp
    engine.PicEngine2D.__init__ = init
    r = bench.extra_c3(steps=300, warm=160)
    engine.PicEngine2D.__init__ = orig
    print(bp, round(r["ms_per_step"], 4), r["stage_ms_per_step"], round(r["roofline"]["k1_kernel_ms"], 4), flush=True)
