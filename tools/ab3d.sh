#!/bin/bash
# usage: tools/ab3d.sh <variant...>  ("product" = in-tree library): K1-3D time of tools/bench3d.py, same box
for v in "$@"; do
  if [ $v = product ]; then unset LPA_LIB_PATH; else export LPA_LIB_PATH=$PWD/lambdapic_amd/csrc/build/liblambdapic_amd_$v.so; fi
  echo -n "== $v: "
  timeout -k 10 200 python tools/bench3d.py --steps 20 $BENCH3D_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],3), 'k1_3d_ms', round(d['k1_3d_ms'],3), 'frac', round(d['k1_3d_frac_of_hbm'],4), 'charge_err', d['charge_rel_err'])"
done
