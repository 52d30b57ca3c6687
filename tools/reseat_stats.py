#!/usr/bin/env python3
"""per-step counters of the in-kernel cell-index sort on config C2 (diagnostic)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, argparse
a = argparse.Namespace(nx=1024, ny=1024, ppc=64, sort_interval=20, block_particles=8192)
from lambdapic_amd.dist import SlabComm
eng, dt, n = bench.build_engine(a, SlabComm(None), torch.device("cuda:0"))
eng.reseat = True
eng.reseat_stats = True
prev = [0, 0, 0, 0]
for it in range(24):
    eng.kernel_events = []
    eng.step(dt)
    torch.cuda.synchronize()
    ms = sum(x.elapsed_time(y) for x, y in eng.kernel_events)
    st = eng._sort_ws(eng.species[0]).get("reloc_stats")
    cur = st.tolist() if st is not None else prev
    d = [c - p for c, p in zip(cur, prev)]
    prev = cur
    print(f"step {it:2d} K1 {ms:.3f} ms  parked {d[0]/n:.4f}  movers {d[1]/n:.4f}  unmatched {d[3]/n:.5f} (pool {d[2]/n:.5f})  overflow {eng._sort_ws(eng.species[0])['counters'][0].item()}")
