#!/bin/bash
# usage: tools/ab_variants.sh <sort_interval> <variant...>   ("product" = the in-tree library)
si=$1; shift
for v in "$@"; do
  if [ $v = product ]; then unset LPA_LIB_PATH; else export LPA_LIB_PATH=$PWD/lambdapic_amd/csrc/build/liblambdapic_amd_$v.so; fi
  for fl in "--reseat" ""; do
    echo -n "== $v si=$si $fl: "
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 8 --sort-interval $si $fl 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3))"
  done
done
