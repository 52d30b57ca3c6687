"""the loop-back communicator of tools/bench_mirror*.py (see bench_mirror.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lambdapic_amd._lib import LPA_MIG_NATTR
from lambdapic_amd.dist import SlabComm


class MirrorComm(SlabComm):
    def __init__(self, slab_width, migrate_capacity):
        super().__init__(None, periodic=True, single=True)
        self.size, self.rank, self.left, self.right = 2, 0, 1, 1
        self.shift = float(slab_width)
        self.Lx = 2 * self.shift

    def exchange(self, send_lo, send_hi, recv_lo, recv_hi, wait=True):
        recv_lo.copy_(send_hi)          # the left neighbour's high face == my own high face
        recv_hi.copy_(send_lo)
        return []

    def arrival_shift(self, box_length):
        # the neighbour is this slab's copy one slab width further (LoopbackComm.arrival_shift)
        return -self.shift, self.shift

    def exchange_many(self, sets):
        for s_ in sets:
            self.exchange(*s_)
        return []

    def allmin(self, v):
        return [float(x) for x in v] if isinstance(v, (list, tuple)) else float(v)

    def any(self, flag):
        return bool(flag)

    def barrier(self):
        pass

    def reduce_diagnostics(self, d):
        return d
