#!/bin/bash
# A/B inside one gpurun call: the non-temporal hint on the particle attribute loads (l) / stores (s) / both of the tiled
# push+deposit kernels.  2-D on bench.py's C2; 3-D on the C5 leg and the uniform slab.  The product build is 3-D: both,
# 2-D: none.  Build the alternatives first (build container):
#   python -c "from lambdapic_amd import build as b; [b.build_variant(n, [f]) for n, f in (('nt2l','LPA_NT_PARTICLES_2D=1'),
#     ('nt2s','LPA_NT_PARTICLES_2D=2'),('nt2','LPA_NT_PARTICLES_2D=3'),('nt3n','LPA_NT_PARTICLES_3D=0'),
#     ('nt3l','LPA_NT_PARTICLES_3D=1'),('nt3s','LPA_NT_PARTICLES_3D=2'))]"
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/lambdapic_amd/csrc/build/liblambdapic_amd
for rep in 1 2; do
for lib in product nt2l nt2s nt2; do
  if [ $lib = product ]; then unset LPA_LIB_PATH; else export LPA_LIB_PATH=${B}_$lib.so; fi
  python3 $ROOT/bench.py --no-extra --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('2-D %-8s' % '$lib', 'C2 k1=%.3f ms step=%.3f ms' % (r.get('kernel_ms', 0), d['ms_per_step']))"
done
for lib in nt3n nt3l nt3s product; do
  if [ $lib = product ]; then unset LPA_LIB_PATH; else export LPA_LIB_PATH=${B}_$lib.so; fi
  python3 $ROOT/tools/bench_c5leg.py 40 12 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('3-D %-8s' % '$lib', 'c5leg k1=%.3f ms step=%.3f ms' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
  python3 $ROOT/tools/bench3d.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('3-D %-8s' % '$lib', 'slab  k1=%.3f ms step=%.3f ms' % (d['k1_3d_ms'], d['ms_per_step']))"
done
done
