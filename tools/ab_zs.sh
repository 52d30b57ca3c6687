#!/bin/bash
# A/B inside one gpurun call: z stride of the no-rho J image of K1-3D, 32 (product) against 24
mkdir -p gpurun_out
for rep in 1 2; do
for lib in product zs24; do
  if [ $lib = product ]; then unset LPA_LIB_PATH; else export LPA_LIB_PATH=$PWD/lambdapic_amd/csrc/build/liblambdapic_amd_$lib.so; fi
  python tools/bench3d.py --steps 40 --warmup 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', 'k1=%.3f ms frac=%.4f step=%.3f ms' % (d['k1_3d_ms'], d['k1_3d_frac_of_hbm'], d['ms_per_step']), d['charge_rel_err'])"
done; done | tee gpurun_out/r03_ab_zs.txt
