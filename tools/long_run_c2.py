import sys, json, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from lambdapic_amd.dist import SlabComm
class A: pass
a = A(); a.nx=1024; a.ny=1024; a.ppc=64; a.sort_interval=20; a.block_particles=8192
eng, dt, n = bench.build_engine(a, SlabComm(None, single=True), torch.device("cuda:0"))
rows=[]
for it in range(1001):
    eng.step(dt)
    if it % 100 == 0:
        d = eng.diagnostics()
        ws = eng._sort_ws(eng.species[0])
        rows.append((it, d["field_energy"], d["kinetic"][0], d["charge"], d["nalive"][0], int(ws["counters"][0].item())))
e0 = rows[0][1] + rows[0][2]
for r in rows:
    print(r[0], "tot/tot0-1 = %.3e" % ((r[1]+r[2])/e0-1), "field %.4e kin %.6e" % (r[1], r[2]), "charge_rel %.1e" % (r[3]/rows[0][3]-1), r[4], "overflow", r[5])
