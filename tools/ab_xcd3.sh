#!/bin/bash
# A/B inside one gpurun call: plain tile order against an XCD-contiguous run of tiles in the whole-tile mode of K1-3D:
# time of the C5 leg and its HBM fetch traffic (separate --pmc pass)
mkdir -p gpurun_out/ab_xcd3
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for lib in product xcd3 product xcd3; do
  if [ $lib = product ]; then unset LPA_LIB_PATH; else export LPA_LIB_PATH=$ROOT/lambdapic_amd/csrc/build/liblambdapic_amd_$lib.so; fi
  python3 $ROOT/tools/bench_c5leg.py 40 12 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', 'k1=%.3f ms frac=%.4f step=%.3f ms' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['ms_per_step']))"
done
for lib in product xcd3; do
  if [ $lib = product ]; then unset LPA_LIB_PATH; else export LPA_LIB_PATH=$ROOT/lambdapic_amd/csrc/build/liblambdapic_amd_$lib.so; fi
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/ab_xcd3/$lib -- python3 $ROOT/tools/bench_c5leg.py 6 12 > $ROOT/gpurun_out/ab_xcd3/$lib.log 2>&1
  python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/ab_xcd3/$lib k_push_deposit_tiled_3d | grep -A1 "true, false" | tail -1 | sed "s/^/$lib /"
done
