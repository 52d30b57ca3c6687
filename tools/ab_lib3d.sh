#!/bin/bash
# A/B inside one gpurun call: K1-3D (C5 leg + uniform slab) with alternative builds of the library
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for lib in product "$@"; do
  if [ $lib = product ]; then unset LPA_LIB_PATH; else export LPA_LIB_PATH=$ROOT/lambdapic_amd/csrc/build/liblambdapic_amd_$lib.so; fi
  python3 $ROOT/tools/bench_c5leg.py 40 12 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-8s' % '$lib', 'c5leg k1=%.3f ms step=%.3f ms' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
  python3 $ROOT/tools/bench3d.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-8s' % '$lib', 'slab  k1=%.3f ms step=%.3f ms' % (d['k1_3d_ms'], d['ms_per_step']))"
done
done
