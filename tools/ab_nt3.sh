#!/bin/bash
# A/B inside one gpurun call: K1-3D with the non-temporal hint on the particle attribute loads / stores
# (-DLPA_NT_PARTICLES_3D=1) against without it: time of the C5 leg, time of the uniform slab, and the HBM-side fetch /
# write traffic of the C5 leg (separate --pmc passes).   bash tools/ab_nt3.sh <alt-lib-name>
ALT=${1:-nt3}
mkdir -p gpurun_out/ab_$ALT
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_$ALT
for lib in product $ALT product $ALT; do
  if [ $lib = product ]; then unset LPA_LIB_PATH; else export LPA_LIB_PATH=$ROOT/lambdapic_amd/csrc/build/liblambdapic_amd_$lib.so; fi
  python3 $ROOT/tools/bench_c5leg.py 40 12 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', 'c5leg k1=%.3f ms frac=%.4f step=%.3f ms' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['ms_per_step']))"
  python3 $ROOT/tools/bench3d.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', 'slab  k1=%.3f ms step=%.3f ms' % (d['k1_3d_ms'], d['ms_per_step']))"
done
for lib in product $ALT; do
  if [ $lib = product ]; then unset LPA_LIB_PATH; else export LPA_LIB_PATH=$ROOT/lambdapic_amd/csrc/build/liblambdapic_amd_$lib.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${lib}_$c -- python3 $ROOT/tools/bench_c5leg.py 6 12 > $OUT/${lib}_$c.log 2>&1
    python3 $ROOT/tools/pmc_summary.py $OUT/${lib}_$c k_push_deposit_tiled_3d | grep -A1 "true, false" | tail -1 | sed "s/^/$lib /"
  done
done
