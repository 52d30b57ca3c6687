#!/bin/bash
# usage: tools/ab_libs.sh <variant...>  ("product" = in-tree library): bench.py C2 K1 / step time, same box, twice each
for rep in 1 2; do
for v in "$@"; do
  if [ $v = product ]; then unset LPA_LIB_PATH; else export LPA_LIB_PATH=$PWD/lambdapic_amd/csrc/build/liblambdapic_amd_$v.so; fi
  echo -n "== $v: "
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 8 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"
done
done
