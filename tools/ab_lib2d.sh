#!/bin/bash
# A/B inside one gpurun call: bench.py's C2 with alternative builds of the library (csrc/build/liblambdapic_amd_<name>.so)
#   [BENCH_ARGS="--no-defer"] bash tools/ab_lib2d.sh name1 [name2 ...]
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for lib in product "$@"; do
  if [ $lib = product ]; then unset LPA_LIB_PATH; else export LPA_LIB_PATH=$ROOT/lambdapic_amd/csrc/build/liblambdapic_amd_$lib.so; fi
  python3 $ROOT/bench.py --no-extra --no-cpu-baseline --steps 60 --warmup 10 $BENCH_ARGS 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-10s' % '$lib', '$BENCH_ARGS', 'C2 k1=%.3f ms step=%.3f ms' % (r.get('kernel_ms', 0), d['ms_per_step']))"
done
done
